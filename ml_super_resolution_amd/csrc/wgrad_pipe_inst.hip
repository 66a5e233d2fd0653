// Conv2DBackpropFilter, one-workgroup-per-CU pipelined linear-walk kernel: the 3x3 64 -> 64 body layers.
#include "launchers.h"
namespace srx {
bool launch_wgrad_pipe(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_WGRAD_PIPE_CASE(3, 3, 64, 4)
    return false;
}
// full-width tiles of 41-pixel rows (VDSR patches), exact rows: one window of 31 steps per 3-row unit
bool launch_wgrad_rows_full(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    if (k.kh == 3 && k.kw == 3 && k.cinp == 64 && k.nch == 4 && a.OW == 41 && a.Cout == 64) {
        *err = launch_with_lds(wgrad_rows_full_kernel<3, 3, 64, 4, 41>, a, grid, lds, s);
        return true;
    }
    return false;
}
// column strips, exact rows (windows of two 32-column strip rows)
bool launch_wgrad_rows_strip(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, bool nt, hipStream_t s, hipError_t* err) {
    if (k.kh == 3 && k.kw == 3 && k.cinp == 64 && k.nch == 4) {
        *err = nt ? launch_with_lds(wgrad_rows_strip_kernel<3, 3, 64, 4, true>, a, grid, lds, s)
                  : launch_with_lds(wgrad_rows_strip_kernel<3, 3, 64, 4, false>, a, grid, lds, s);
        return true;
    }
    return false;
}
}  // namespace srx
