// texture_gram.hip -- EnhanceNet's texture-matching statistics (enet/enet/model_enet.py:34-41, 225-259) in one pass:
//   normalize(x) = x / (mean over channels + 1e-6), per pixel;
//   tf.extract_image_patches(16x16, stride 16) + reshape [-1, h*w/256, 256, C];
//   gram = patches^T patches   ([.., C, C], a sum over the 256 pixels of a patch),
// and the gradient of all three.  As separate launches (srx_channel_normalize, srx_extract_patches16, srx_gemm) every
// step writes and re-reads a feature-sized tensor -- 268 MB for block1_conv1 at 64 x 128 x 128 x 64 -- and the GEMM
// streams its operands from L2; here a workgroup reads its patch once, normalises it on the way into LDS and runs the
// products on exact-fp32 MFMA out of LDS.  Bound: HBM (the feature tensor is read once per 64 gram rows).
//
// Forward: workgroup = (patch, block of 64 gram rows); wave = 16 rows x all C columns (C/16 accumulators); K = the
// patch's 256 pixels, 4 per MFMA, staged 128 (C <= 128) or 64 (C = 256) pixels at a time (<= 74 KB of LDS: two to four
// workgroups per CU, one filling while another multiplies).
// Backward (dgram symmetric, as the difference of two gram matrices is): dn = alpha * n . dgram  (alpha = 2:
// d(n^T n) -> n (dG + dG^T)), then the gradient of normalize: with m = mean + eps, t = sum_c dn_c x_c:
// dx_c = dn_c / m - t / (C m^2).  Workgroup = (patch, 64 of its pixels); wave = 16 pixels x all C channels; K = the C
// gram rows, staged 64 at a time.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>

#include "../../include/srx.h"

namespace srx {
int set_error(int code, const char* fmt, ...);

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

// sum over the LPP consecutive lanes that hold one pixel (LPP = 16, 32 or 64)
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPP / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Patch p of x [N,H,W,C]: pointer to its pixel (0, 0); pixel (r, c) is (r * W + c) * C floats further.
__device__ __forceinline__ size_t patch_origin(int p, int H, int W, int C) {
    const int pw = W >> 4, ph = H >> 4;
    const int n = p / (ph * pw), q = p - n * (ph * pw);
    const int py = q / pw, px = q - py * pw;
    return (((size_t)n * H + py * 16) * W + px * 16) * C;
}

// KP pixels [first, first + KP) of the patch -> LDS rows of LS floats, normalised; mrow (nullable): the divisor per pixel
template <int C, int KP, int LS>
__device__ __forceinline__ void fill_normalised(float* tile, float* mrow, const float* xp, int W, int first, float eps, int tid) {
    constexpr int LPP = C / 4;             // lanes per pixel, one float4 each
    constexpr int PPI = 256 / LPP;         // pixels per iteration
    const int c4 = tid % LPP, sub = tid / LPP;
#pragma unroll 4
    for (int it = 0; it < KP / PPI; ++it) {
        const int k = it * PPI + sub;
        const int idx = first + k;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + ((size_t)(idx >> 4) * W + (idx & 15)) * C + 4 * c4);
        const float m = group_sum<LPP>((v[0] + v[1]) + (v[2] + v[3])) / (float)C + eps;
        const float im = 1.0f / m;            // (one division per lane and pixel, not four: the fill is VALU work beside the MFMAs)
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] * im;
        *reinterpret_cast<f32x4*>(tile + k * LS + 4 * c4) = o;
        if (mrow && c4 == 0) mrow[k] = m;
    }
}

template <int C>
__global__ __launch_bounds__(256) void texture_gram_kernel(const float* __restrict__ x, float* __restrict__ gram, int H, int W,
                                                           float eps) {
    constexpr int LS = C + 16;                       // (+16: the four pixel rows of a k-step fall on two bank halves)
    constexpr int KP = (C <= 128) ? 128 : 64;        // pixels per LDS pass: <= 74 KB, two to four workgroups per CU
    constexpr int RB = C / 64, NJ = C / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x / RB, rb = blockIdx.x - p * RB;
    const float* xp = x + patch_origin(p, H, W, C);
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int first = 0; first < 256; first += KP) {
        if (first) __syncthreads();
        fill_normalised<C, KP, LS>(lds, nullptr, xp, W, first, eps, tid);
        __syncthreads();
        const float* arow = lds + kq * LS + rb * 64 + 16 * wave + li;
        const float* brow = lds + kq * LS + li;
        // operands of k-step s + 1 are read while the MFMAs of k-step s run
        float an = arow[0], bn[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bn[j] = brow[16 * j];
#pragma unroll 2
        for (int s = 0; s < KP / 4; ++s) {
            const float a = an;
            float b[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = bn[j];
            const int s1 = (s + 1 < KP / 4) ? s + 1 : s;
            an = arow[4 * s1 * LS];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bn[j] = brow[4 * s1 * LS + 16 * j];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[j], 0, 0, 0);
        }
    }
    float* g = gram + (size_t)p * C * C + (size_t)(rb * 64 + 16 * wave + 4 * kq) * C + li;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[(size_t)r * C + 16 * j] = acc[j][r];
}

template <int C>
__global__ __launch_bounds__(256) void texture_gram_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dgram,
                                                               float* __restrict__ dx, int H, int W, float eps, float alpha) {
    constexpr int LSN = C + 4;                       // normalised pixels (read down a column of 16 pixels per k-step)
    constexpr int LSG = C + 16;                      // gram rows (read along 16 columns)
    constexpr int NJ = C / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tn = lds;                                 // [64][LSN]
    float* tm = tn + 64 * LSN;                       // [64]
    float* tg = tm + 64;                             // [64][LSG]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int p = blockIdx.x >> 2, pb = blockIdx.x & 3;
    const size_t origin = patch_origin(p, H, W, C);
    fill_normalised<C, 64, LSN>(tn, tm, x + origin, W, pb * 64, eps, tid);
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* gp = dgram + (size_t)p * C * C;
    for (int jc = 0; jc < C / 64; ++jc) {
        __syncthreads();                             // the previous rows are consumed (first time: tn is complete)
        for (int i = tid; i < 64 * (C / 4); i += 256) {
            const int row = i / (C / 4), c4 = i - row * (C / 4);
            *reinterpret_cast<f32x4*>(tg + row * LSG + 4 * c4) =
                *reinterpret_cast<const f32x4*>(gp + (size_t)(64 * jc + row) * C + 4 * c4);
        }
        __syncthreads();
        const float* arow = tn + (16 * wave + li) * LSN + 64 * jc + kq;
        const float* brow = tg + kq * LSG + li;
        float an = arow[0], bn[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bn[j] = brow[16 * j];
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const float a = an;
            float b[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = bn[j];
            const int s1 = (s + 1 < 16) ? s + 1 : s;
            an = arow[4 * s1];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bn[j] = brow[4 * s1 * LSG + 16 * j];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[j], 0, 0, 0);
        }
    }
    // the lane holds dn[pixel 16 wave + 4 kq + r][channel 16 j + li] / alpha
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 16 * wave + 4 * kq + r;
        const float m = tm[k];
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) t += (alpha * acc[j][r]) * (tn[k * LSN + 16 * j + li] * m);
        t = group_sum<16>(t);
        const float im = 1.0f / m;
        const float kk = t * (im * im) * (1.0f / (float)C);
        const int idx = pb * 64 + k;
        float* o = dx + origin + ((size_t)(idx >> 4) * W + (idx & 15)) * C + li;
#pragma unroll
        for (int j = 0; j < NJ; ++j) o[16 * j] = (alpha * acc[j][r]) * im - kk;
    }
}

// more than 64 KiB of dynamic LDS needs the attribute: raised once per kernel and thread (outside any later stream capture)
template <typename K>
hipError_t configure_lds(K kernel, size_t bytes) {
    static thread_local bool done = false;           // (one instance of this function per kernel type)
    if (done || bytes <= 64 * 1024) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done = (e == hipSuccess);
    return e;
}

int check_dims(const void* a, const void* b, const void* c, int N, int H, int W, int C) {
    if (!a || !b || !c) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || (H & 15) || (W & 15)) return set_error(SRX_ERR_BAD_ARG, "texture_gram: H, W must be multiples of 16");
    if (C != 64 && C != 128 && C != 256) return set_error(SRX_ERR_UNSUPPORTED, "texture_gram: 64, 128 or 256 channels (VGG-19 block1/2/3_conv1), got %d", C);
    if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15u) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    if ((long)N * (H / 16) * (W / 16) * (C / 64) >= (1L << 31) / 4) return set_error(SRX_ERR_UNSUPPORTED, "texture_gram: too many patches");
    return SRX_OK;
}

}  // namespace
}  // namespace srx

using namespace srx;

#define SRX_TG_LAUNCH(KERNEL, GRID, LDS, ...)                                                                          \
    do {                                                                                                               \
        hipError_t e_ = configure_lds(KERNEL, LDS);                                                                    \
        if (e_ == hipSuccess) {                                                                                        \
            hipLaunchKernelGGL(KERNEL, dim3((unsigned)(GRID)), dim3(256), LDS, (hipStream_t)stream, __VA_ARGS__);      \
            e_ = hipGetLastError();                                                                                    \
        }                                                                                                              \
        if (e_ != hipSuccess) return set_error(SRX_ERR_LAUNCH, "texture_gram launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

extern "C" int srx_texture_gram(const float* x, float* gram, int N, int H, int W, int C, float eps, srx_stream_t stream) {
    const int rc = check_dims(x, gram, gram, N, H, W, C);
    if (rc) return rc;
    const long patches = (long)N * (H / 16) * (W / 16);
    if (C == 64) SRX_TG_LAUNCH(texture_gram_kernel<64>, patches, (size_t)128 * 80 * 4, x, gram, H, W, eps);
    else if (C == 128) SRX_TG_LAUNCH(texture_gram_kernel<128>, patches * 2, (size_t)128 * 144 * 4, x, gram, H, W, eps);
    else SRX_TG_LAUNCH(texture_gram_kernel<256>, patches * 4, (size_t)64 * 272 * 4, x, gram, H, W, eps);
    return SRX_OK;
}

extern "C" int srx_texture_gram_bwd(const float* x, const float* dgram, float* dx, int N, int H, int W, int C, float eps,
                                    float alpha, srx_stream_t stream) {
    const int rc = check_dims(x, dgram, dx, N, H, W, C);
    if (rc) return rc;
    if (x == dx) return set_error(SRX_ERR_BAD_ARG, "texture_gram_bwd: dx cannot alias x (four workgroups read each patch)");
    const long patches = (long)N * (H / 16) * (W / 16);
    const size_t lds = ((size_t)64 * (C + 4) + 64 + (size_t)64 * (C + 16)) * 4;
    if (C == 64) SRX_TG_LAUNCH(texture_gram_bwd_kernel<64>, patches * 4, lds, x, dgram, dx, H, W, eps, alpha);
    else if (C == 128) SRX_TG_LAUNCH(texture_gram_bwd_kernel<128>, patches * 4, lds, x, dgram, dx, H, W, eps, alpha);
    else SRX_TG_LAUNCH(texture_gram_bwd_kernel<256>, patches * 4, lds, x, dgram, dx, H, W, eps, alpha);
    return SRX_OK;
}
