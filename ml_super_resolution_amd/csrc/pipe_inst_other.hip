// Pipelined (one wave per SIMD) instances for 32-channel 3x3 layers.
#include "launchers.h"
namespace srx {
bool launch_pipe_other(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_PIPE_CASE_FWD(3, 3, 32, 4)
    SRX_PIPE_CASE_DGRAD(3, 3, 32, 4)
    SRX_PIPE_CASE_FWD(3, 3, 32, 2)
    SRX_PIPE_CASE_DGRAD(3, 3, 32, 2)
    return false;
}
}  // namespace srx
