// Pipelined (one wave per SIMD) 3x3, 64 -> 64 channels on column-strip tiles: wide images (W >= 60).
#include "launchers.h"
namespace srx {
bool launch_pipe_strip(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_PIPE_STRIP_CASE(3, 3, 64, 4)
    SRX_PIPE_STRIP_CASE_FWD2(3, 3, 64, 2)
    SRX_PIPE_STRIP_CASE_D2S(3, 3, 32, 2)
    return false;
}
}  // namespace srx
