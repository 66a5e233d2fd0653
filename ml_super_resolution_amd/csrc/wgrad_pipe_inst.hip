// Conv2DBackpropFilter, one-workgroup-per-CU pipelined linear-walk kernel: the 3x3 64 -> 64 body layers.
#include "launchers.h"
namespace srx {
bool launch_wgrad_pipe(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_WGRAD_PIPE_CASE(3, 3, 64, 4)
    return false;
}
}  // namespace srx
