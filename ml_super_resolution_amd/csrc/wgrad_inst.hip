// Conv2DBackpropFilter instances.
#include "launchers.h"
namespace srx {
// out = sum over the G per-workgroup partials, in a fixed order (deterministic):
//   j <  wn : dw[j]      = sum_g part[g][j] (+ wd * w[j])
//   j >= wn : dbias[j-wn] = sum_g part[g][j]
// A block owns 64 consecutive outputs; its 256 threads are 16 float4 columns x 16 partial groups.
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ part, int G, int stride, int wn, int n,
                                                     float* __restrict__ dw, float* __restrict__ dbias,
                                                     const float* __restrict__ w, float wd) {
    __shared__ f32x4 sh[16][17];
    const int c = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int j = blockIdx.x * 64 + 4 * c;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (j < n) {
        const float* p = part + j;
        // all of a thread's loads in flight at once (the launch is two waves of blocks: its time is memory round
        // trips, not bytes); the order of the additions stays g = pg, pg+16, ...: results are unchanged
        int g = pg;
        for (; g + 7 * 16 < G; g += 8 * 16) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(g + 16 * i) * stride);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        for (; g < G; g += 16) s += *reinterpret_cast<const f32x4*>(p + (size_t)g * stride);
    }
    sh[pg][c] = s;
    __syncthreads();
    if (pg == 0 && j < n) {
        f32x4 t = sh[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sh[k][c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int jj = j + e;
            if (jj < wn)
                dw[jj] = t[e] + (w ? wd * w[jj] : 0.f);
            else if (jj < wn + (n - wn) && dbias)
                dbias[jj - wn] = t[e];
        }
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int G, int stride,
                                                              int wn, int n, float* __restrict__ dw,
                                                              float* __restrict__ dbias, const float* __restrict__ w,
                                                              float wd) {
    reduce_partials_body(part, G, stride, wn, n, dw, dbias, w, wd);
}
// blockIdx.y = pair (ib * cob + ob) of a layer wider than 64 channels; the bias gradient of output block ob is the one
// the pair (0, ob) produced (every pair ib, ob sums the same dpre[ob])
__global__ __launch_bounds__(256) void reduce_partials_pairs_kernel(const float* __restrict__ part, int G, int stride,
                                                                    int wn, int n, float* __restrict__ dw,
                                                                    float* __restrict__ dbias, int cob) {
    const int pair = blockIdx.y;
    float* db = (dbias && pair < cob) ? dbias + (size_t)pair * (n - wn) : nullptr;
    reduce_partials_body(part + (size_t)pair * G * stride, G, stride, wn, n, dw + (size_t)pair * wn, db, nullptr, 0.f);
}

hipError_t launch_reduce_partials_pairs(const float* part, int G, int stride, int wn, int cout, float* dw, float* dbias,
                                        int pairs, int cob, hipStream_t s) {
    const int n = wn + cout;
    hipLaunchKernelGGL(reduce_partials_pairs_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)pairs), dim3(256), 0, s, part, G,
                       stride, wn, n, dw, dbias, cob);
    return hipGetLastError();
}

hipError_t launch_reduce_partials(const float* part, int G, int stride, int wn, int cout, float* dw, float* dbias,
                                  const float* w, float wd, hipStream_t s) {
    const int n = wn + cout;   // stride is a multiple of 4 >= n, so the float4 loads stay in the row
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, part, G, stride, wn, n,
                       dw, dbias, w, wd);
    return hipGetLastError();
}

// Shapes outside the tuned set: passes of 9 taps over runtime KH x KW (wgrad_generic_kernel)
bool launch_wgrad_generic(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    if (a.stride != 1) return false;
    const int taps = k.kh * k.kw;
    *err = hipSuccess;
    for (int tap0 = 0; tap0 < taps && *err == hipSuccess; tap0 += 9) {
        const int nt = taps - tap0 < 9 ? taps - tap0 : 9;
#define SRX_WGRAD_GENERIC(CI, NC)                                                                                        \
        if (k.cinp == CI && k.nch == NC) {                                                                               \
            hipLaunchKernelGGL((wgrad_generic_kernel<CI, NC>), dim3(grid), dim3(256), lds, s, a, k.kh, k.kw, tap0, nt); \
            *err = hipGetLastError();                                                                                    \
            continue;                                                                                                    \
        }
        SRX_WGRAD_GENERIC(64, 4) SRX_WGRAD_GENERIC(64, 2) SRX_WGRAD_GENERIC(64, 1)
        SRX_WGRAD_GENERIC(32, 4) SRX_WGRAD_GENERIC(32, 2) SRX_WGRAD_GENERIC(32, 1)
        SRX_WGRAD_GENERIC(4, 4) SRX_WGRAD_GENERIC(4, 2) SRX_WGRAD_GENERIC(4, 1)
#undef SRX_WGRAD_GENERIC
        return false;
    }
    return true;
}

bool launch_wgrad(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_WGRAD_CASE(3, 3, 64, 4, 2)
    SRX_WGRAD_CASE(3, 3, 64, 2, 2)
    SRX_WGRAD_CASE(3, 3, 64, 1, 2)
    SRX_WGRAD_CASE(3, 3, 32, 4, 2)
    SRX_WGRAD_CASE(3, 3, 32, 2, 2)
    SRX_WGRAD_CASE(3, 3, 32, 1, 2)
    SRX_WGRAD_CASE(3, 3, 4, 4, 2)
    SRX_WGRAD_CASE(3, 3, 4, 2, 2)
    SRX_WGRAD_CASE(5, 5, 4, 4, 2)
    SRX_WGRAD_CASE(9, 9, 4, 4, 2)
    SRX_WGRAD_CASE(1, 1, 64, 4, 2)
    SRX_WGRAD_CASE(1, 1, 64, 2, 2)
    SRX_WGRAD_CASE(1, 1, 32, 4, 2)
    SRX_WGRAD_CASE(5, 5, 32, 1, 2)
    return false;
}
}  // namespace srx
