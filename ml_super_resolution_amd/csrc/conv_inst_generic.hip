// Generic (runtime filter size, streamed weights) forward / dgrad: any KHxKW, Cin,Cout <= 64.
#include "launchers.h"
namespace srx {
#define SRX_GENERIC_CASE(CINP, NCH, WT)                                                                          \
    if (k.cinp == CINP && k.nch == NCH && k.wt == WT) {                                                          \
        auto kern = conv_mfma_generic_kernel<CINP, NCH, WT>;                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        if (e == hipSuccess) {                                                                                   \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, k.kh, k.kw);                              \
            e = hipGetLastError();                                                                               \
        }                                                                                                        \
        *err = e;                                                                                                \
        return true;                                                                                             \
    }
bool launch_conv_generic(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_GENERIC_CASE(4, 1, false) SRX_GENERIC_CASE(4, 2, false) SRX_GENERIC_CASE(4, 4, false)
    SRX_GENERIC_CASE(32, 1, false) SRX_GENERIC_CASE(32, 2, false) SRX_GENERIC_CASE(32, 4, false)
    SRX_GENERIC_CASE(64, 1, false) SRX_GENERIC_CASE(64, 2, false) SRX_GENERIC_CASE(64, 4, false)
    SRX_GENERIC_CASE(4, 1, true) SRX_GENERIC_CASE(4, 2, true) SRX_GENERIC_CASE(4, 4, true)
    SRX_GENERIC_CASE(32, 1, true) SRX_GENERIC_CASE(32, 2, true) SRX_GENERIC_CASE(32, 4, true)
    SRX_GENERIC_CASE(64, 1, true) SRX_GENERIC_CASE(64, 2, true) SRX_GENERIC_CASE(64, 4, true)
    return false;
}
}  // namespace srx
