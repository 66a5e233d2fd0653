// 3x3, 64 input channels: the VDSR / EnhanceNet body (fwd and dgrad).
#include "launchers.h"
namespace srx {
bool launch_conv_k3c64(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_CONV_CASE(3, 3, 64, 4, false, 2)
    SRX_CONV_CASE(3, 3, 64, 4, true, 2)
    SRX_CONV_CASE(3, 3, 64, 2, false, 2)
    SRX_CONV_CASE(3, 3, 64, 2, true, 2)
    SRX_CONV_CASE(3, 3, 64, 1, false, 2)
    SRX_CONV_CASE(3, 3, 64, 1, true, 2)
    return false;
}
}  // namespace srx
