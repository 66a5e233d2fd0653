// Pipelined (one wave per SIMD) instances for 32-channel 3x3 layers, 1x1 layers and 5x5x32.
#include "launchers.h"
namespace srx {
bool launch_pipe_other(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_PIPE_CASE(3, 3, 32, 4, false)
    SRX_PIPE_CASE(3, 3, 32, 4, true)
    SRX_PIPE_CASE(3, 3, 32, 2, false)
    SRX_PIPE_CASE(3, 3, 32, 2, true)
    SRX_PIPE_CASE(3, 3, 32, 1, false)
    SRX_PIPE_CASE(3, 3, 32, 1, true)
    SRX_PIPE_CASE(1, 1, 64, 4, false)
    SRX_PIPE_CASE(1, 1, 64, 4, true)
    SRX_PIPE_CASE(1, 1, 64, 2, false)
    SRX_PIPE_CASE(1, 1, 64, 2, true)
    SRX_PIPE_CASE(1, 1, 32, 4, false)
    SRX_PIPE_CASE(1, 1, 32, 4, true)
    SRX_PIPE_CASE(1, 1, 32, 2, false)
    SRX_PIPE_CASE(1, 1, 32, 2, true)
    SRX_PIPE_CASE(5, 5, 32, 1, false)
    SRX_PIPE_CASE(5, 5, 32, 2, false)
    return false;
}
}  // namespace srx
