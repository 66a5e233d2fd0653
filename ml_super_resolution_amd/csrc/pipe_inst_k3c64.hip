// Pipelined (one wave per SIMD) 3x3, 64 input channels: the VDSR / EnhanceNet body.
#include "launchers.h"
namespace srx {
// (no instance for a single 16-channel chunk: it would need rows of >= 64 pixels, which the planner column-tiles)
bool launch_pipe_k3c64(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_PIPE_CASE_FWD(3, 3, 64, 4)
    SRX_PIPE_CASE_DGRAD(3, 3, 64, 4)
    SRX_PIPE_CASE_FWD(3, 3, 64, 2)
    SRX_PIPE_CASE_DGRAD(3, 3, 64, 2)
    return false;
}
}  // namespace srx
