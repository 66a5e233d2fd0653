// Conv2DBackpropFilter, linear-walk kernel (full-width tiles): the 3x3 layers with 32 / 64 staged channels.
#include "launchers.h"
namespace srx {
bool launch_wgrad_lin(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_WGRAD_LIN_CASE(3, 3, 64, 4, 2)
    SRX_WGRAD_LIN_CASE(3, 3, 64, 2, 2)
    SRX_WGRAD_LIN_CASE(3, 3, 64, 1, 2)
    SRX_WGRAD_LIN_CASE(3, 3, 32, 4, 2)
    SRX_WGRAD_LIN_CASE(3, 3, 32, 2, 2)
    SRX_WGRAD_LIN_CASE(3, 3, 32, 1, 2)
    return false;
}
// RGB-input layers, packed (kw, ci) rows: SRCNN 9x9 3 -> 64, ESPCN 5x5 3 -> 64
bool launch_wgrad_lin_pack3(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    if (a.Cin != 3 || k.nch != 4) return false;
    if (k.kh == 9 && k.kw == 9) { *err = launch_with_lds(wgrad_lin_pack3_kernel<9, 9, 4, 2>, a, grid, lds, s); return true; }
    if (k.kh == 5 && k.kw == 5) { *err = launch_with_lds(wgrad_lin_pack3_kernel<5, 5, 4, 2>, a, grid, lds, s); return true; }
    return false;
}
// column strips (wide images): the 64 -> 64 body layers
bool launch_wgrad_lin_strip(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_WGRAD_LIN_STRIP_CASE(3, 3, 64, 4, 2)
    return false;
}
// layers wider than 64 channels: all block pairs in one launch
bool launch_wgrad_lin_pairs(const ConvKey& k, const WgradPairs& q, int grid, int pairs, bool strips, size_t lds, hipStream_t s, hipError_t* err) {
    if (k.kh == 3 && k.kw == 3 && k.cinp == 64 && k.nch == 4) {
        *err = strips ? launch_with_lds(wgrad_lin_pairs_kernel<3, 3, 64, 4, 2, true>, q, grid, lds, s, pairs)
                      : launch_with_lds(wgrad_lin_pairs_kernel<3, 3, 64, 4, 2, false>, q, grid, lds, s, pairs);
        return true;
    }
    return false;
}
}  // namespace srx
