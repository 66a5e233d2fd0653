// wgrad_1x1.hip -- filter gradient of a 1x1 convolution: dW[ci][co] = sum over pixels p of x[p][ci] * dpre[p][co]
// (SRCNN's non-linear mapping layer 64 -> 32, srcnn/srcnn.py:111-119; EnhanceNet's residual blocks 64 -> 64,
// enet/enet/model_enet.py:8-31).  A plain GEMM with K = every pixel of the batch and a tiny M x N: 384 / 512 bytes of traffic
// per pixel against 4 / 8 kFLOP -- HBM-bound (bound: **HBM**).  On wgrad_mfma_kernel (LDS tile staged between barriers, per-lane
// pixel cursors) SRCNN's layer ran at 1.44 TB/s (942 us for batch 64 of 243 x 243).  Here nothing goes through LDS: a wave
// reads 4 pixels per step straight into MFMA operand layout -- lane (li, kq) takes CIN/16 consecutive input channels and COUT/16
// consecutive output channels of pixel kq, i.e. whole 128- / 256-byte pixels per 16 lanes -- and issues (CIN/16) x (COUT/16)
// MFMAs on element pairs; MFMA row m of tile e is channel (CIN/16) m + e: a fixed permutation, undone when the partial is
// written.  Three steps of loads in flight per wave, 16 waves per CU.  Exact fp32; the 4 pixels of a step are added inside
// the MFMA, steps in order, the workgroup's 8 waves in a fixed tree, workgroups by reduce_partials_kernel: deterministic.
#include "launchers.h"
namespace srx {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V> struct VecOf;
template <> struct VecOf<2> { typedef f32x2 type; };
template <> struct VecOf<4> { typedef f32x4 type; };

template <int V>
__device__ __forceinline__ typename VecOf<V>::type load_vec(__amdgpu_buffer_rsrc_t rs, int voff) {
    if constexpr (V == 4) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    else return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0));
}

constexpr int kWaves1x1 = 8;

template <int CIN, int COUT>
__global__ __launch_bounds__(64 * kWaves1x1, 2) void wgrad_1x1_kernel(const float* __restrict__ x, const float* __restrict__ dpre,
                                                                      float* __restrict__ part, int part_stride, long pixels) {
    constexpr int VA = CIN / 16, VB = COUT / 16;          // channels per lane: x / dpre
    constexpr int DEPTH = 3;                              // steps of loads in flight per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    // this workgroup's pixels: a contiguous range, a whole number of 4-pixel steps except at the very end
    const long steps_total = (pixels + 3) >> 2;
    const long s0 = steps_total * blockIdx.x / gridDim.x, s1 = steps_total * (blockIdx.x + 1) / gridDim.x;
    const long p0 = s0 << 2;
    const long pend = (s1 << 2) < pixels ? (s1 << 2) : pixels;
    const int nsteps = (int)(s1 - s0);
    // buffer resources over the range: pixels past its end (the tensor's ragged last step) read as zeros
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + p0 * CIN, 0, (int)((pend - p0) * CIN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dpre) + p0 * COUT, 0, (int)((pend - p0) * COUT * 4), 0x00020000);
    const int xoff = (kq * CIN + VA * li) * 4, boff = (kq * COUT + VB * li) * 4;

    f32x4 acc[VA][VB];
#pragma unroll
    for (int i = 0; i < VA; ++i)
#pragma unroll
        for (int j = 0; j < VB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs[VB];
#pragma unroll
    for (int j = 0; j < VB; ++j) bs[j] = 0.f;

    typename VecOf<VA>::type av[DEPTH];
    typename VecOf<VB>::type bv[DEPTH];
    // steps wave, wave + kWaves, ... of the range; out-of-range steps have out-of-range offsets (zeros: they add nothing)
    auto fetch = [&](int d, int s) {
        const int ok = s < nsteps;
        av[d] = load_vec<VA>(xrs, ok ? xoff + s * (4 * CIN * 4) : kOobOffset);
        bv[d] = load_vec<VB>(brs, ok ? boff + s * (4 * COUT * 4) : kOobOffset);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, wave + d * kWaves1x1);
    for (int s = wave; s < nsteps; s += DEPTH * kWaves1x1) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const typename VecOf<VA>::type a = av[d];
            const typename VecOf<VB>::type b = bv[d];
            fetch(d, s + (d + DEPTH) * kWaves1x1);
#pragma unroll
            for (int j = 0; j < VB; ++j) bs[j] += b[j];
#pragma unroll
            for (int i = 0; i < VA; ++i)
#pragma unroll
                for (int j = 0; j < VB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- the workgroup's 8 waves: a fixed tree through LDS (waves 4..7 -> 0..3, 2..3 -> 0..1, 1 -> 0)
    constexpr int ACCF = VA * VB * 4;                     // accumulator floats per lane
    constexpr int REGION = 64 * (ACCF + VB);              // one wave's accumulators and bias sums; wave w + half writes region w
#pragma unroll
    for (int half = kWaves1x1 / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* dst = lds + (size_t)(wave - half) * REGION + lane * (ACCF + VB);
#pragma unroll
            for (int i = 0; i < VA; ++i)
#pragma unroll
                for (int j = 0; j < VB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(i * VB + j) * 4 + r] = acc[i][j][r];
#pragma unroll
            for (int j = 0; j < VB; ++j) dst[ACCF + j] = bs[j];
        }
        __syncthreads();
        if (wave < half) {
            const float* src = lds + (size_t)wave * REGION + lane * (ACCF + VB);
#pragma unroll
            for (int i = 0; i < VA; ++i)
#pragma unroll
                for (int j = 0; j < VB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += src[(i * VB + j) * 4 + r];
#pragma unroll
            for (int j = 0; j < VB; ++j) bs[j] += src[ACCF + j];
        }
        __syncthreads();
    }
    if (wave == 0) {
        // D layout: lane (li, kq) holds rows 4 kq + r (r = 0..3) of column li -> channels ci = VA (4 kq + r) + i, co = VB li + j
        float* pw = part + (size_t)blockIdx.x * part_stride;
#pragma unroll
        for (int i = 0; i < VA; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = VA * (4 * kq + r) + i;
                float v[VB];
#pragma unroll
                for (int j = 0; j < VB; ++j) v[j] = acc[i][j][r];
                if constexpr (VB == 4) *reinterpret_cast<f32x4*>(pw + ci * COUT + VB * li) = f32x4{v[0], v[1], v[2], v[3 % VB]};
                else *reinterpret_cast<f32x2*>(pw + ci * COUT + VB * li) = f32x2{v[0], v[1 % VB]};
            }
        // bias gradient: the lane's dpre sums over its pixels kq, kq + 4, ...: add the four lane groups
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            float t = bs[j];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (kq == 0) pw[CIN * COUT + VB * li + j] = t;
        }
    }
}

}  // namespace

// Returns true when this route took the launch (then *n_partials workgroups wrote one partial each).
bool launch_wgrad_1x1(const ConvKey& k, const WgradArgs& a, int max_partials, int* n_partials, hipStream_t s, hipError_t* err) {
    if (k.kh != 1 || k.kw != 1 || a.stride != 1 || a.OH != a.H || a.OW != a.W) return false;
    if (!((a.Cin == 64 || a.Cin == 32) && (a.Cout == 64 || a.Cout == 32))) return false;
    const long pixels = (long)a.N * a.OH * a.OW;
    const long steps = (pixels + 3) >> 2;
    int grid = max_partials < 512 ? max_partials : 512;          // two workgroups of 8 waves per CU
    if (steps < grid) grid = (int)steps;
    if (grid < 1) return false;
    // a workgroup's range must stay below 2^31 bytes of either tensor (32-bit buffer offsets)
    if ((steps / grid + 2) * 4 * (long)(a.Cin > a.Cout ? a.Cin : a.Cout) * 4 >= (1L << 31) - 4096) return false;
    const size_t lds = (size_t)4 * 64 * ((a.Cin / 16) * (a.Cout / 16) * 4 + a.Cout / 16) * sizeof(float);
#define SRX_1X1(CI, CO)                                                                                                   \
    if (a.Cin == CI && a.Cout == CO) {                                                                                    \
        static thread_local bool configured = false;                                                                      \
        if (!configured && lds > 48 * 1024) {                                                                             \
            *err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_1x1_kernel<CI, CO>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
            if (*err != hipSuccess) return true;                                                                          \
            configured = true;                                                                                            \
        }                                                                                                                 \
        hipLaunchKernelGGL((wgrad_1x1_kernel<CI, CO>), dim3(grid), dim3(64 * kWaves1x1), lds, s, a.x, a.dpre, a.part, a.part_stride, pixels); \
    }
    SRX_1X1(64, 32) SRX_1X1(64, 64) SRX_1X1(32, 32) SRX_1X1(32, 64)
#undef SRX_1X1
    *err = hipGetLastError();
    *n_partials = grid;
    return true;
}
}  // namespace srx
