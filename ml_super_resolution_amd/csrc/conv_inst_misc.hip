// 1x1 (SRCNN non-linear mapping, EnhanceNet residual blocks) and 5x5x32 (SRCNN reconstruction).
#include "launchers.h"
namespace srx {
bool launch_conv_misc(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_CONV_CASE(1, 1, 64, 4, false, 2)
    SRX_CONV_CASE(1, 1, 64, 4, true, 2)
    SRX_CONV_CASE(1, 1, 64, 2, false, 2)
    SRX_CONV_CASE(1, 1, 64, 2, true, 2)
    SRX_CONV_CASE(1, 1, 32, 4, false, 2)
    SRX_CONV_CASE(1, 1, 32, 4, true, 2)
    SRX_CONV_CASE(1, 1, 32, 2, false, 2)
    SRX_CONV_CASE(1, 1, 32, 2, true, 2)
    SRX_CONV_CASE(5, 5, 32, 1, false, 1)
    SRX_CONV_CASE(5, 5, 32, 2, false, 1)
    return false;
}
}  // namespace srx
