// conv_1x1.hip -- forward and data gradient of a 1x1 convolution on large inputs (SRCNN's non-linear mapping layer 64 -> 32,
// srcnn/srcnn.py:111-119, and its data gradient 32 -> 64 with the ReLU gradient of the layer input; the 64 -> 64 / 32 -> 32
// shapes come for free).  Per pixel: 256 + 128 bytes of traffic against 4 kFLOP -- bound: **HBM**.  conv_mfma_kernel stages
// such a layer through an LDS tile between barriers (3.6 TB/s at 720 x 1280, 4.0-4.4 TB/s in SRCNN's train step); here, as in
// wgrad_1x1_kernel, nothing goes through LDS: a wave owns 16 pixels per step and reads them straight into MFMA operand layout
// -- lane (li, kq) takes channels 16 g + 4 kq .. + 3 of pixel li, one 16-byte load per group g of 16 input channels: whole
// 64-byte pieces of a pixel per 4 lanes --, the COUT x CIN filter is stationary in registers (A operand: lane (li, kq) holds
// w[16 g + 4 kq + s][16 c + li] for k-step s of group g and output chunk c), the accumulators start as the bias, and the D
// layout (lane (li, kq): channels 16 c + 4 kq .. + 3 of pixel li) is stored with one 16-byte store per chunk.  Two steps of
// loads in flight per wave, 16 waves per CU.  Same products in the same order as conv_mfma_kernel (per accumulator: groups g
// ascending, k-steps s ascending; bias first): bit-identical to it.
#include "launchers.h"
namespace srx {
namespace {

constexpr int kWaves = 8;

// WT = false: forward, w is [CIN][COUT] (HWIO of a 1x1 layer), epilogue bias + none / ReLU.
// WT = true:  data gradient of a layer [COUT_layer = CIN here][...]: x is dpre, w is the FORWARD layer's [COUT][CIN] array (this
//             kernel's output channel m is the layer's input channel), no bias, epilogue: ReLU-gradient mask on the layer input.
template <int CIN, int COUT, bool WT>
__global__ __launch_bounds__(64 * kWaves, 2) void conv_1x1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const float* __restrict__ mask,
                                                                  float* __restrict__ y, long pixels, int act) {
    constexpr int NG = CIN / 16, NC = COUT / 16;
    constexpr int DEPTH = 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    // this workgroup's steps of 16 pixels: a contiguous range
    const long steps_total = (pixels + 15) >> 4;
    const long s0 = steps_total * blockIdx.x / gridDim.x, s1 = steps_total * (blockIdx.x + 1) / gridDim.x;
    const int nsteps = (int)(s1 - s0);
    const long p0 = s0 << 4;
    const long pend = (s1 << 4) < pixels ? (s1 << 4) : pixels;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + p0 * CIN, 0, (int)((pend - p0) * CIN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y + p0 * COUT, 0, (int)((pend - p0) * COUT * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mask ? mask : x) + (mask ? p0 * COUT : 0), 0,
                                                                         mask ? (int)((pend - p0) * COUT * 4) : 0, 0x00020000);

    // stationary filter: wr[c][4 g + s]
    float wr[NC][CIN / 4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * g + 4 * kq + s, m = 16 * c + li;
                wr[c][4 * g + s] = WT ? w[(size_t)m * CIN + k] : w[(size_t)k * COUT + m];
            }
    f32x4 bias4[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        bias4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!WT && bias) bias4[c] = *reinterpret_cast<const f32x4*>(bias + 16 * c + 4 * kq);
    }
    const float slope = act_slope(act);

    const int xoff = (li * CIN + 4 * kq) * 4, yoff = (li * COUT + 4 * kq) * 4;
    f32x4 bv[DEPTH][NG], mv[DEPTH][WT ? NC : 1];
    // steps wave, wave + kWaves, ... of the range; steps past its end get out-of-range offsets (zeros in, nothing out)
    auto fetch = [&](int d, int s) {
        const bool ok = s < nsteps;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            bv[d][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xoff + s * (16 * CIN * 4) + 64 * g : kOobOffset, 0, 0));
        if (WT) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                mv[d][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, ok ? yoff + s * (16 * COUT * 4) + 64 * c : kOobOffset, 0, 0));
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, wave + d * kWaves);
    for (int s = wave; s < nsteps; s += DEPTH * kWaves) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            f32x4 b[NG], m[WT ? NC : 1];
#pragma unroll
            for (int g = 0; g < NG; ++g) b[g] = bv[d][g];
            if (WT) {
#pragma unroll
                for (int c = 0; c < NC; ++c) m[c] = mv[d][c];
            }
            const int sd = s + d * kWaves;
            fetch(d, sd + DEPTH * kWaves);
            f32x4 acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = bias4[c];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[c][4 * g + e], b[g][e], acc[c], 0, 0, 0);
            const int so = (sd < nsteps) ? yoff + sd * (16 * COUT * 4) : kOobOffset;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 v = acc[c];
                if (WT) {
                    if (mask) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = m[c][e] > 0.f ? v[e] : 0.0f;
                    }
                } else {
                    v = act_apply4(v, act, slope);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrs, so == kOobOffset ? kOobOffset : so + 64 * c, 0, 0);
            }
        }
    }
}

}  // namespace

// Returns true when this route took the launch.  min_pixels: below it the layer stays on conv_mfma_kernel.
bool launch_conv_1x1(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err) {
    if (k.kh != 1 || k.kw != 1 || a.stride != 1 || a.skip || a.d2s_r || a.post_relu || a.OH != a.H || a.OW != a.W) return false;
    if (!((a.Cin == 64 || a.Cin == 32) && (a.Cout == 64 || a.Cout == 32))) return false;
    if (k.wt) {
        if (a.bias || a.act != ACT_NONE || (a.mask && a.mask_act != ACT_RELU)) return false;
    } else {
        if (a.mask || !(a.act == ACT_NONE || a.act == ACT_RELU)) return false;
        if (a.bias && ((uintptr_t)a.bias & 15u)) return false;
    }
    const long pixels = (long)a.N * a.OH * a.OW;
    if (pixels < min_pixels) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)(a.mask ? a.mask : a.x)) & 15u) return false;
    const long steps = (pixels + 15) >> 4;
    int grid = 512;                                                // two workgroups of 8 waves per CU
    if (steps < grid) grid = (int)steps;
    // a workgroup's range must stay below 2^31 bytes of either tensor (32-bit buffer offsets)
    if ((steps / grid + 2) * 16 * 64 * 4 >= (1L << 31) - 4096) return false;
#define SRX_C1(CI, CO)                                                                                                   \
    if (a.Cin == CI && a.Cout == CO) {                                                                                   \
        if (k.wt) hipLaunchKernelGGL((conv_1x1_kernel<CI, CO, true>), dim3(grid), dim3(64 * kWaves), 0, s, a.x, a.w, a.bias, a.mask, a.y, pixels, a.act); \
        else hipLaunchKernelGGL((conv_1x1_kernel<CI, CO, false>), dim3(grid), dim3(64 * kWaves), 0, s, a.x, a.w, a.bias, a.mask, a.y, pixels, a.act); \
    }
    SRX_C1(64, 32) SRX_C1(32, 64) SRX_C1(64, 64) SRX_C1(32, 32)
#undef SRX_C1
    *err = hipGetLastError();
    return true;
}
}  // namespace srx
