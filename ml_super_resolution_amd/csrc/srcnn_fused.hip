// srcnn_fused.hip -- SRCNN 9-1-5 inference in ONE launch (BASELINE configs[0]: one 256x256 image, 243x243 as the reference
// crops it):
//   t1 = relu(conv9x9(x; 3 -> 64) + b1), t2 = relu(conv1x1(t1; 64 -> 32) + b2), y = tanh(conv5x5(t2; 32 -> 3) + b3), all VALID
// (srcnn/srcnn.py:100-130).  [N,H,W,3] -> [N,H-12,W-12,3].
//
// As three launches the 243x243 image takes ~89 us: three latency-bound ramps (filters -> tile -> MFMA chains -> store), the
// 9x9x3 layer on the generic-filter kernel.  Here a workgroup owns a tile of <= 15x15 OUTPUT pixels and chains the layers
// through LDS: the (T+12)^2 input halo (4 floats per pixel), t1 on the tile grown by 2 ((T+4)^2 x 64 channels), t2 on the
// same pixels (1x1 layer, x 32 channels, in the LDS space the input halo no longer needs), then the 5x5 layer straight to
// global memory.  VALID geometry: no padding anywhere, every halo pixel lies inside the image.  231 = 15.4 tiles of 15,
// evened out to 16 x 16 tiles = 256 workgroups: one per CU, one round.  Halo pixels of t1 / t2 are recomputed by every tile
// that needs them ((T+4)^2 / T^2 = 1.6 at T = 15): the kernel is for latency-bound sizes, the host keeps the per-layer
// launches for large batches.
// All three filter slices of a wave (81 + 16 + 200 registers) are loaded once per workgroup (one workgroup per CU, 512
// registers), exact fp32 on v_mfma_f32_16x16x4_f32, the same products in the same order as the per-layer kernels
// (bias as the initial accumulator; taps in (kh, kw) order; per tap the channel groups of 16, four k-steps each, k-step s
// of group g covering channels 16 g + 4 q + s of lane group q) and the same activation code: bit-identical to them.
#include <stdarg.h>

#include "../../include/srx.h"
#include "launchers.h"

namespace srx {
int set_error(int code, const char* fmt, ...);

namespace {

struct SrcnnArgs {
    const float *x, *w1, *b1, *w2, *b2, *w3, *b3;
    float* y;
    int N, H, W, OH, OW;
    int T;                       // tile edge (<= 15)
    int tiles_y, tiles_x, units;
};

constexpr int kT = 15;                       // largest tile edge
constexpr int kP1 = 68, kP2 = 36;            // LDS pixel strides of t1 (64 + 4) and t2 (32 + 4) in floats
constexpr int kT1 = (kT + 4) * (kT + 4) * kP1;
constexpr int kX0 = (kT + 12) * (kT + 12) * 4;           // input halo, 4 floats per pixel
constexpr int kT2 = (kT + 4) * (kT + 4) * kP2;
constexpr int kShared = kX0 > kT2 ? kX0 : kT2;           // t2 reuses the input halo's space (dead after the first layer)

__device__ __forceinline__ f32x4 relu4(f32x4 v) {        // as act_apply4 (conv_kernels.hip.h): x & ~(x >> 31)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float f = v[e];
        const int b = __float_as_int(f);
        v[e] = __int_as_float(b & ~(b >> 31));
    }
    return v;
}

// four MFMAs that share the A operand (a weight in a VGPR), accumulators in VGPRs; the leading s_nop covers a VALU write of
// any operand right before the statement (hipcc pads no hazards across an inline-asm boundary)
__device__ __forceinline__ void mfma4_guarded(f32x4 (&c)[4], float w, float b0, float b1, float b2, float b3) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\t" "v_mfma_f32_16x16x4_f32 %1, %4, %6, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %2, %4, %7, %2\n\t" "v_mfma_f32_16x16x4_f32 %3, %4, %8, %3"
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "v"(w), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
}

__global__ __launch_bounds__(256, 1) void srcnn_fused_kernel(const SrcnnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* T1 = lds;
    float* X0 = lds + kT1;
    float* T2 = lds + kT1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- this wave's filter slices, stationary for the whole kernel
    // f1: 9x9, chunk = wave (16 of the 64 channels); k index = input channel kq (3 -> 4: channel 3 is zero)
    float w1r[81];
#pragma unroll
    for (int t = 0; t < 81; ++t) w1r[t] = (kq < 3) ? a.w1[(t * 3 + kq) * 64 + 16 * wave + li] : 0.f;
    const f32x4 b1r = *reinterpret_cast<const f32x4*>(a.b1 + 16 * wave + 4 * kq);
    // f2: 1x1, chunk = wave & 1 (16 of the 32 channels); the two waves of a chunk split the sub-tiles
    const int ch2 = wave & 1, half2 = wave >> 1;
    float w2r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w2r[j] = a.w2[(16 * (j / 4) + 4 * kq + (j % 4)) * 32 + 16 * ch2 + li];
    const f32x4 b2r = *reinterpret_cast<const f32x4*>(a.b2 + 16 * ch2 + 4 * kq);
    // f3: 5x5, one chunk (3 of its 16 channels are real); the four waves split the sub-tiles
    float w3r[200];
#pragma unroll
    for (int t = 0; t < 25; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) w3r[t * 8 + j] = (li < 3) ? a.w3[(t * 32 + 16 * (j / 4) + 4 * kq + (j % 4)) * 3 + li] : 0.f;
    float b3r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b3r[e] = (4 * kq + e < 3) ? a.b3[4 * kq + e] : 0.f;
    const f32x4 b3v = {b3r[0], b3r[1], b3r[2], b3r[3]};
    // f3's weights live in accumulation registers from here on: 297 weights do not fit the 256 architectural VGPRs -- round 3's
    // build kept the overflow in AGPRs behind v_accvgpr_read copies and read every LDS fragment right before its MFMA.  They are
    // defined as "a" values here and consumed by the "a" operands of the asm MFMA statements of the f3 phase.
#pragma unroll
    for (int i = 0; i < 200; ++i) { const float t = w3r[i]; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(w3r[i]) : "v"(t)); }

    for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
        const int tx_i = u % a.tiles_x, t2_ = u / a.tiles_x;
        const int ty_i = t2_ % a.tiles_y, n = t2_ / a.tiles_y;
        const int oy = ty_i * a.T, ox = tx_i * a.T;
        const int th = (a.OH - oy < a.T) ? (a.OH - oy) : a.T;
        const int tw = (a.OW - ox < a.T) ? (a.OW - ox) : a.T;
        const int w0 = tw + 12, w1 = tw + 4;                         // region widths: input halo, t1 / t2
        const int n0 = (th + 12) * w0, n1 = (th + 4) * w1, n3 = th * tw;

        // ---- stage the input halo, 4 floats per pixel (VALID: every pixel of it is inside the image)
        __syncthreads();
        for (int p = tid; p < n0; p += 256) {
            const int r = p / w0, c = p - r * w0;
            const float* px = a.x + (((size_t)n * a.H + (oy + r)) * a.W + (ox + c)) * 3;
            f32x4 v = {px[0], px[1], px[2], 0.f};
            *reinterpret_cast<f32x4*>(X0 + p * 4) = v;
        }
        __syncthreads();

        // ---- f1: 9x9, 3 -> 64, relu, on the tile grown by 2; wave = channel chunk, all sub-tiles
        for (int s0 = 0; s0 * 16 < n1; s0 += 4) {
            int la[4];
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (s0 + i) + li;
                const int tt = t < n1 ? t : 0;
                const int r = tt / w1, c = tt - r * w1;
                la[i] = (r * w0 + c) * 4 + kq;
                acc[i] = b1r;
            }
            // (the LDS reads of tap t + 1 come before the MFMAs of tap t)
            float b[4], bn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = X0[la[i]];
#pragma unroll
            for (int t = 0; t < 81; ++t) {
                if (t + 1 < 81) {
                    const int kh1 = (t + 1) / 9, kw1 = (t + 1) % 9;
#pragma unroll
                    for (int i = 0; i < 4; ++i) bn[i] = X0[la[i] + (kh1 * w0 + kw1) * 4];
                }
                mfma4_guarded(acc, w1r[t], b[0], b[1], b[2], b[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = bn[i];
            }
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (s0 + i) + li;
                if (t < n1) *reinterpret_cast<f32x4*>(T1 + t * kP1 + 16 * wave + 4 * kq) = relu4(acc[i]);
            }
        }
        __syncthreads();          // t1 complete; the input halo is dead: its space becomes t2

        // ---- f2: 1x1, 64 -> 32, relu, same pixels; wave = (chunk, half of the sub-tiles)
        for (int s0 = 4 * half2; s0 * 16 < n1; s0 += 8) {
            int la[4];
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (s0 + i) + li;
                la[i] = (t < n1 ? t : 0) * kP1 + 4 * kq;
                acc[i] = b2r;
            }
            f32x4 b[4], bn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f32x4*>(T1 + la[i]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bn[i] = *reinterpret_cast<const f32x4*>(T1 + la[i] + 16 * (g + 1));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[4 * g + e], b[i][e], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = bn[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (s0 + i) + li;
                if (t < n1) *reinterpret_cast<f32x4*>(T2 + t * kP2 + 16 * ch2 + 4 * kq) = relu4(acc[i]);
            }
        }
        __syncthreads();

        // ---- f3: 5x5, 32 -> 3, tanh, to global memory; the four waves split the sub-tiles
        float* y_img = a.y + (size_t)n * a.OH * a.OW * 3;
        for (int s0 = 4 * wave; s0 * 16 < n3; s0 += 16) {
            int la[4];
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (s0 + i) + li;
                const int tt = t < n3 ? t : 0;
                const int r = tt / tw, c = tt - r * tw;
                la[i] = (r * w1 + c) * kP2 + 4 * kq;
            }
            // asm MFMA statements (weights in accumulation registers, the first one defines the accumulators with C = bias);
            // the fragments of block t + 1 are read before the MFMAs of block t
            f32x4 b[4], bn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f32x4*>(T2 + la[i]);
#pragma unroll
            for (int t = 0; t < 50; ++t) {
                if (t + 1 < 50) {
                    const int tap1 = (t + 1) / 2, g1 = (t + 1) % 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        bn[i] = *reinterpret_cast<const f32x4*>(T2 + la[i] + ((tap1 / 5) * w1 + (tap1 % 5)) * kP2 + 16 * g1);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (t == 0 && e == 0) mfma_sub_a_first(acc, b3v, w3r[0], b[0][0], b[1][0], b[2][0], b[3][0]);
                    else mfma_sub_a(acc, w3r[4 * t + e], b[0][e], b[1][e], b[2][e], b[3][e]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = bn[i];
            }
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            if (kq == 0) {       // lane group 0 holds output channels 0..3: three of them exist
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = 16 * (s0 + i) + li;
                    if (t < n3) {
                        const int r = t / tw, c = t - r * tw;
                        float* o = y_img + ((size_t)(oy + r) * a.OW + (ox + c)) * 3;
                        const f32x4 v = act_transcendental4(acc[i], ACT_TANH);
                        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
                    }
                }
            }
        }
    }
}

}  // namespace
}  // namespace srx

using namespace srx;

extern "C" int srx_srcnn_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                 const float* w3, const float* b3, float* y, int N, int H, int W, srx_stream_t stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !y) return set_error(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (N <= 0 || H < 13 || W < 13) return set_error(SRX_ERR_BAD_ARG, "srcnn_forward: the 9-1-5 VALID chain needs images of at least 13 x 13");
    if (((uintptr_t)b1 | (uintptr_t)b2) & 15u) return set_error(SRX_ERR_ALIGN, "bias pointers must be 16-byte aligned");
    if ((long)N * H * W * 3L >= (1L << 31)) return set_error(SRX_ERR_UNSUPPORTED, "srcnn_forward: tensor beyond 32-bit offsets");
    SrcnnArgs a;
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.y = y;
    a.N = N; a.H = H; a.W = W; a.OH = H - 12; a.OW = W - 12;
    // tile edge: the largest that fits, evened out over the image (231 -> 16 tiles of 15, not 15 of 15 + one of 6)
    const int ny = (a.OH + kT - 1) / kT, nx = (a.OW + kT - 1) / kT;
    int T = (a.OH + ny - 1) / ny;
    const int Tx = (a.OW + nx - 1) / nx;
    if (Tx > T) T = Tx;
    a.T = T;
    a.tiles_y = (a.OH + T - 1) / T; a.tiles_x = (a.OW + T - 1) / T;
    const long units = (long)N * a.tiles_y * a.tiles_x;
    if (units >= (1L << 31)) return set_error(SRX_ERR_UNSUPPORTED, "srcnn_forward: too many tiles");
    a.units = (int)units;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0 || cus > 256) cus = 256;
    const int grid = (int)(units < (long)cus ? units : (long)cus);
    const size_t lds = (size_t)(kT1 + kShared) * 4;
    const hipError_t e = launch_with_lds(srcnn_fused_kernel, a, grid, lds, (hipStream_t)stream);
    if (e != hipSuccess) return set_error(SRX_ERR_LAUNCH, "srcnn_forward launch failed: %s", hipGetErrorString(e));
    return SRX_OK;
}
