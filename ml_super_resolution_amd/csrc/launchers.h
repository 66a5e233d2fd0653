// launchers.h -- internal: typed launch entry points for the kernel instances, one per
// translation unit so the instances compile in parallel.
#pragma once
#include "conv_kernels.hip.h"

namespace srx {

struct ConvKey {
    int kh, kw, cinp, nch;
    bool wt;
};

// Each returns true if the key belongs to that unit (then *err holds the launch status).
bool launch_conv_k3c64(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_conv_k3c32(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_conv_c4(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_conv_misc(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_pipe_k3c64(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_pipe_other(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_narrow(const ConvKey& k, const WgradArgs& a, int grid, hipStream_t s, hipError_t* err);
bool launch_conv_narrow(const ConvKey& k, const ConvArgs& a, hipStream_t s, hipError_t* err);
bool launch_conv_1x1(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err);
bool launch_conv_rows3x3(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err);
bool launch_conv_pack3(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err);
bool launch_conv_kwrows(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err);
bool launch_pipe_strip(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_conv_generic(const ConvKey& k, const ConvArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_generic(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_lin(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_lin_pack3(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_lin_strip(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_pipe(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_kwcols(const ConvKey& k, const WgradArgs& a, int max_partials, long min_pixels, int* n_partials, hipStream_t s, hipError_t* err);
bool launch_wgrad_1x1(const ConvKey& k, const WgradArgs& a, int max_partials, int* n_partials, hipStream_t s, hipError_t* err);
bool launch_wgrad_rows_full(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err);
bool launch_wgrad_rows_strip(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, bool nt, hipStream_t s, hipError_t* err);
bool launch_wgrad_lin_pairs(const ConvKey& k, const WgradPairs& q, int grid, int pairs, bool strips, size_t lds, hipStream_t s, hipError_t* err);

hipError_t launch_reduce_partials(const float* part, int G, int stride, int wn, int cout, float* dw, float* dbias,
                                  const float* w, float wd, hipStream_t s);
// `pairs` problems at once: partials [pair][G][stride] -> dw [pair][wn], dbias [cob][cout] (taken from the pairs ib == 0)
hipError_t launch_reduce_partials_pairs(const float* part, int G, int stride, int wn, int cout, float* dw, float* dbias,
                                        int pairs, int cob, hipStream_t s);

template <typename K, typename A>
inline hipError_t launch_with_lds(K kernel, const A& a, int grid, size_t lds, hipStream_t s, int grid_y = 1) {
    // > 64 KiB of dynamic LDS needs the attribute.  It is raised once per kernel to the largest size the
    // planner can ask for (160 KiB), outside any stream capture of later launches.
    static thread_local const void* configured[64];
    static thread_local int n_configured = 0;
    const void* fn = reinterpret_cast<const void*>(kernel);
    bool known = false;
    for (int i = 0; i < n_configured; ++i) known |= (configured[i] == fn);
    if (!known) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (n_configured < 64) configured[n_configured++] = fn;
    }
    hipLaunchKernelGGL(kernel, dim3(grid, grid_y), dim3(256), lds, s, a);
    return hipGetLastError();
}

#define SRX_CONV_CASE(KH, KW, CINP, NCH, WT, MINW)                                                        \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH && k.wt == WT) {                       \
        if (a.skip || a.mask)                                                                             \
            *err = launch_with_lds(conv_mfma_kernel<KH, KW, CINP, NCH, WT, MINW, true>, a, grid, lds, s); \
        else                                                                                              \
            *err = launch_with_lds(conv_mfma_kernel<KH, KW, CINP, NCH, WT, MINW, false>, a, grid, lds, s);\
        return true;                                                                                      \
    }
// Pipelined one-wave-per-SIMD kernel (>= 16 input channels): one workgroup per CU, two LDS buffers.
// (forward launches carry at most a residual operand, dgrad launches at most a ReluGrad mask)
#define SRX_PIPE_CASE_FWD(KH, KW, CINP, NCH)                                                              \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH && !k.wt) {                            \
        if (a.skip)                                                                                       \
            *err = launch_with_lds(conv_pipe_kernel<KH, KW, CINP, NCH, false, 2>, a, grid, lds, s);       \
        else                                                                                              \
            *err = launch_with_lds(conv_pipe_kernel<KH, KW, CINP, NCH, false, 0>, a, grid, lds, s);       \
        return true;                                                                                      \
    }
#define SRX_PIPE_CASE_DGRAD(KH, KW, CINP, NCH)                                                            \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH && k.wt) {                             \
        if (a.mask)                                                                                       \
            *err = launch_with_lds(conv_pipe_kernel<KH, KW, CINP, NCH, true, 1>, a, grid, lds, s);        \
        else                                                                                              \
            *err = launch_with_lds(conv_pipe_kernel<KH, KW, CINP, NCH, true, 0>, a, grid, lds, s);        \
        return true;                                                                                      \
    }
#define SRX_PIPE_STRIP_CASE(KH, KW, CINP, NCH)                                                            \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH && !k.wt) {                            \
        if (a.skip)                                                                                       \
            *err = launch_with_lds(conv_pipe_strip_kernel<KH, KW, CINP, NCH, false, 2>, a, grid, lds, s); \
        else                                                                                              \
            *err = launch_with_lds(conv_pipe_strip_kernel<KH, KW, CINP, NCH, false, 0>, a, grid, lds, s); \
        return true;                                                                                      \
    }                                                                                                     \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH && k.wt) {                             \
        if (a.mask)                                                                                       \
            *err = launch_with_lds(conv_pipe_strip_kernel<KH, KW, CINP, NCH, true, 1>, a, grid, lds, s);  \
        else                                                                                              \
            *err = launch_with_lds(conv_pipe_strip_kernel<KH, KW, CINP, NCH, true, 0>, a, grid, lds, s);  \
        return true;                                                                                      \
    }
// forward only, 32 output channels (two chunks), no aux operand: none / ReLU, or tanh (ESPCN's f2 on whole images)
#define SRX_PIPE_STRIP_CASE_FWD2(KH, KW, CINP, NCH)                                                       \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH && !k.wt && !a.skip && !a.mask) {      \
        if (a.act == ACT_TANH)                                                                            \
            *err = launch_with_lds(conv_pipe_strip_kernel<KH, KW, CINP, NCH, false, 3>, a, grid, lds, s); \
        else                                                                                              \
            *err = launch_with_lds(conv_pipe_strip_kernel<KH, KW, CINP, NCH, false, 0>, a, grid, lds, s); \
        return true;                                                                                      \
    }
#define SRX_PIPE_STRIP_CASE_D2S(KH, KW, CINP, NCH)                                                        \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH && !k.wt && !a.skip && !a.mask && a.d2s_r) { \
        *err = launch_with_lds(conv_pipe_strip_kernel<KH, KW, CINP, NCH, false, 4>, a, grid, lds, s);     \
        return true;                                                                                      \
    }
#define SRX_WGRAD_LIN_CASE(KH, KW, CINP, NCH, MINW)                                                       \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH) {                                     \
        *err = launch_with_lds(wgrad_lin_kernel<KH, KW, CINP, NCH, MINW>, a, grid, lds, s);               \
        return true;                                                                                      \
    }
#define SRX_WGRAD_LIN_STRIP_CASE(KH, KW, CINP, NCH, MINW)                                                 \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH) {                                     \
        *err = launch_with_lds(wgrad_lin_strip_kernel<KH, KW, CINP, NCH, MINW>, a, grid, lds, s);         \
        return true;                                                                                      \
    }
#define SRX_WGRAD_PIPE_CASE(KH, KW, CINP, NCH)                                                            \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH) {                                     \
        *err = launch_with_lds(wgrad_pipe_kernel<KH, KW, CINP, NCH>, a, grid, lds, s);                    \
        return true;                                                                                      \
    }
#define SRX_WGRAD_CASE(KH, KW, CINP, NCH, MINW)                                                           \
    if (k.kh == KH && k.kw == KW && k.cinp == CINP && k.nch == NCH) {                                     \
        *err = launch_with_lds(wgrad_mfma_kernel<KH, KW, CINP, NCH, MINW>, a, grid, lds, s);              \
        return true;                                                                                      \
    }

}  // namespace srx
