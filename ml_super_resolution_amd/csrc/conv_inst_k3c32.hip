// 3x3, <=32 input channels (ESPCN f3 and the dgrads of 32-wide layers).
#include "launchers.h"
namespace srx {
bool launch_conv_k3c32(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_CONV_CASE(3, 3, 32, 4, false, 2)
    SRX_CONV_CASE(3, 3, 32, 4, true, 2)
    SRX_CONV_CASE(3, 3, 32, 2, false, 2)
    SRX_CONV_CASE(3, 3, 32, 2, true, 2)
    SRX_CONV_CASE(3, 3, 32, 1, false, 2)
    SRX_CONV_CASE(3, 3, 32, 1, true, 2)
    return false;
}
}  // namespace srx
