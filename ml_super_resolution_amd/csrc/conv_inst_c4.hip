// <=4 input channels (RGB first layers; dgrad of RGB-output last layers): 3x3, 5x5, 9x9.
#include "launchers.h"
namespace srx {
bool launch_conv_c4(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_CONV_CASE(3, 3, 4, 4, false, 2)
    SRX_CONV_CASE(3, 3, 4, 4, true, 2)
    SRX_CONV_CASE(3, 3, 4, 2, false, 2)
    SRX_CONV_CASE(3, 3, 4, 2, true, 2)
    SRX_CONV_CASE(3, 3, 4, 1, false, 2)
    SRX_CONV_CASE(5, 5, 4, 4, false, 2)
    SRX_CONV_CASE(5, 5, 4, 4, true, 2)
    SRX_CONV_CASE(5, 5, 4, 2, false, 2)
    SRX_CONV_CASE(5, 5, 4, 2, true, 2)
    SRX_CONV_CASE(9, 9, 4, 4, false, 2)
    SRX_CONV_CASE(9, 9, 4, 4, true, 2)
    return false;
}
}  // namespace srx
