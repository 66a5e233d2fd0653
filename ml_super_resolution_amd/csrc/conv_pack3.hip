// conv_pack3.hip -- forward convolution of the RGB-INPUT layers on large inputs: SRCNN's patch extraction 9x9 3 -> 64
// (srcnn/srcnn.py:100-109) and ESPCN's f1 5x5 3 -> 64 (espcn/espcn/model_espcn.py:30-38).  conv_mfma_kernel stages such an
// input as 4-float pixels (3 channels + a zero) and spends one MFMA per tap: its K = 4 slots hold 3 useful products, 81 / 25
// MFMAs per 16 pixels and 16 output channels.  Here the LDS pixel is 3 floats, so the 3 KW values (kw, ci) of one filter row are
// CONSECUTIVE floats and K simply runs along them, 4 at a time: ceil(3 KW / 4) MFMAs per filter row -- 63 instead of 81 for 9x9,
// 20 instead of 25 for 5x5 -- with the B fragment of MFMA j a plain `ds_read_b32` at (pixel, 4 j + lane's k) and the last
// MFMA's unused slots weighted zero.  Exact fp32.  The products of an output reach its accumulator in the same (kh, kw, ci)
// order as in conv_mfma_kernel, cut into groups of four at other places -- and the results are BIT-IDENTICAL to that kernel's
// on every shape tested (tests/test_gpu_ops.py: test_conv_rgb_input_packed_k_route_vs_oracle): v_mfma_f32_16x16x4_f32 adds its
// four products to the accumulator one after the other in k order, and a zero-weighted slot adds +0 -- for FINITE inputs: the
// slot past a filter row multiplies the next pixel's first channel by zero, so an Inf / NaN there (one column right of the
// window) would reach this output too; images are finite (uint8-derived), conv path 0 has no such slot.  Thresholds
// (srx_api.hip): 9x9 from 4,096 output pixels, 5x5 from 60,000 (smaller ESPCN inputs take the one-launch kernel).
//
// (End of round 4: the same kernel as the DATA GRADIENT of SRCNN's 5x5 32 -> 3 layer -- a 5x5 "forward" from the 3 channels of dpre
// to 32 channels with the filter flipped and transposed and the ReLU gradient of the layer input as epilogue: template WT.)
//
// One workgroup of 8 waves per CU; a tile is 32 output rows x 64 output columns (its input halo: <= 40 x 72 pixels = 34.6 KB);
// wave = (16-channel chunk, row parity): it walks its rows, four 16-pixel sub-tiles (= the strip's 64 columns) at a time,
// MFMAs pinned in blocks of 4 (one per sub-tile, one weight register) with the LDS reads of the next block issued first;
// the bias is the accumulators' initial value, the activation and the 16-byte stores follow each row.
#include "launchers.h"
namespace srx {
namespace {

constexpr int kTH = 32, kTW = 64, kNW = 8;

__device__ __forceinline__ void mfma4_shared_a(f32x4 (&c)[4], float w, float b0, float b1, float b2, float b3) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\t" "v_mfma_f32_16x16x4_f32 %1, %4, %6, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %2, %4, %7, %2\n\t" "v_mfma_f32_16x16x4_f32 %3, %4, %8, %3"
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "v"(w), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

// COUT = 64: 4 channel chunks x 2 row phases; COUT = 32: 2 chunks x 4 row phases.
// WT: the DATA GRADIENT of a k x k layer with 3 output channels (SRCNN's reconstruction layer, srcnn/srcnn.py:122-130): x is dpre
// [N,H,W,3], w the forward layer's HWIO array [KH,KW,COUT,3] read flipped and transposed, no bias, the ReLU gradient of the
// layer input (a.mask) as the epilogue.
template <int KH, int KW, int COUT = 64, bool WT = false>
__global__ __launch_bounds__(64 * kNW, 1) void conv_pack3_kernel(const ConvArgs a, int units_total) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KR = 3 * KW;                  // (kw, ci) values of one filter row
    constexpr int NJ = (KR + 3) / 4;            // MFMAs per filter row
    constexpr int NBLK = KH * NJ;
    constexpr int RSW = kTW + KW - 1;           // input columns of a tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    constexpr int NCHK = COUT / 16, PH = kNW / NCHK;
    const int chunk = wave % NCHK, par = wave / NCHK;
    const int co = 16 * chunk + li;             // A operand: the lane's output channel

    // stationary weights: wr[kh][j] = w[kh][k / 3][k % 3][co], k = 4 j + kq (zero past the filter row)
    float wr[NBLK];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int kk = 4 * j + kq;
            const bool ok = kk < KR && co < a.Cout;
            size_t wi = ((size_t)(kh * KW) * 3 + kk) * a.Cout + co;                         // ((kh KW + kw) 3 + ci) = kh KW 3 + k
            if (WT) wi = ((size_t)((KH - 1 - kh) * KW + (KW - 1 - kk / 3)) * a.Cout + co) * 3 + kk % 3;   // w_f[KH-1-kh][KW-1-kw][co][c]
            const float v = a.w[ok ? wi : 0];
            wr[kh * NJ + j] = ok ? v : 0.0f;
        }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    const int cb = 16 * chunk + 4 * kq;         // D layout: the lane's four output channels
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cb + e < a.Cout) bias4[e] = a.bias[cb + e];
    }
    const float slope = act_slope(a.act);

    const int u0 = (int)(((long)blockIdx.x * units_total) / gridDim.x), u1 = (int)(((long)(blockIdx.x + 1) * units_total) / gridDim.x);
    for (int u = u0; u < u1;) {
        // units are the output rows of the column strips (a contiguous range per workgroup); a tile = up to kTH of them
        const int h0 = u % a.OH;
        const int t2 = u / a.OH;
        const int tx = t2 % a.NTX, n = t2 / a.NTX;
        const int ow0 = tx * kTW;
        int th = a.OH - h0 < kTH ? a.OH - h0 : kTH;
        if (u1 - u < th) th = u1 - u;
        const int tw = a.OW - ow0 < kTW ? a.OW - ow0 : kTW;
        lds_barrier();
        {
            // the input halo as 3-float pixels: float f of tile row r <-> image float (ow0 - pad_l) 3 + f of row h0 - pad_t + r
            const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)n * a.H * a.W * 3, 0, a.H * a.W * 3 * 4, 0x00020000);
            const int rows = th + KH - 1, rowf = RSW * 3;
            const int c0 = (ow0 - a.pad_l) * 3;
            for (int i = tid; i < rows * rowf; i += 64 * kNW) {
                const int r = i / rowf, f = i - r * rowf;
                const int ih = h0 - a.pad_t + r, cf = c0 + f;
                const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)cf < (unsigned)(a.W * 3));
                lds[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? (ih * a.W * 3 + cf) * 4 : kOobOffset, 0, 0));
            }
            // the zero-weighted k slots of the tile's very last pixel lie past the staged floats: they must hold NUMBERS (0 x NaN
            // is NaN; whatever an earlier kernel left in LDS is there otherwise -- found by scripts/fuzz_round4.py, round 4)
#ifndef SRX_TEST_NO_PAD_ZERO
            if (tid < 4) lds[rows * rowf + tid] = 0.0f;
#endif
        }
        lds_barrier();
        float* yn = a.y + ((size_t)n * a.OH + h0) * a.OW * a.Cout;
        const float* mn = (WT && a.mask) ? a.mask + ((size_t)n * a.OH + h0) * a.OW * a.Cout : nullptr;
        for (int r = par; r < th; r += PH) {
            // B operand of sub-tile g (columns 16 g + li), block (kh, j): float ((r + kh) RSW + 16 g + li) 3 + 4 j + kq
            const float* px = lds + (r * RSW + li) * 3 + kq;
            f32x4 acc[4] = {bias4, bias4, bias4, bias4};
            // (data gradient: the row's mask values are requested here and used after the MFMA blocks)
            f32x4 mrow[WT ? 4 : 1];
            if (WT && mn && cb < a.Cout) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool in = 16 * g + li < tw;
                    mrow[g] = in ? *reinterpret_cast<const f32x4*>(mn + ((size_t)r * a.OW + ow0 + li + 16 * g) * a.Cout + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            float c0 = px[0], c1 = px[48], c2 = px[96], c3 = px[144];
#pragma unroll
            for (int t = 0; t < NBLK; ++t) {
                float n0 = c0, n1 = c1, n2 = c2, n3 = c3;
                if (t + 1 < NBLK) {
                    const int kh1 = (t + 1) / NJ, j1 = (t + 1) % NJ;
                    const float* q = px + kh1 * RSW * 3 + 4 * j1;
                    n0 = q[0]; n1 = q[48]; n2 = q[96]; n3 = q[144];
                }
                mfma4_shared_a(acc, wr[t], c0, c1, c2, c3);
                c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            }
            // MFMA results are read by VALU code next: software covers the result latency (the MFMAs above are asm)
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            if (cb < a.Cout) {
                float* yo = yn + ((size_t)r * a.OW + ow0 + li) * a.Cout + cb;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (16 * g + li < tw) {
                        f32x4 v = acc[g];
                        if (WT) {
                            if (mn) {
                                const f32x4 m = mrow[WT ? g : 0];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.0f;
                            }
                        } else {
                            v = act_apply4(v, a.act, slope);
                        }
                        *reinterpret_cast<f32x4*>(yo + (size_t)16 * g * a.Cout) = v;
                    }
            }
        }
        u += th;
    }
}

}  // namespace

// Returns true when this route took the launch.  min_pixels: below it the layer stays on conv_mfma_kernel.
bool launch_conv_pack3(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err) {
    if (a.skip || a.d2s_r || a.stride != 1 || a.post_relu) return false;
    if (k.wt) {
        // data gradient of a 5x5 layer with 3 output channels and 32 input channels (dx: 32 channels), ReLU mask or none
        if (a.Cin != 3 || a.Cout != 32 || k.kh != 5 || k.kw != 5 || a.bias || a.act != ACT_NONE || (a.mask && a.mask_act != ACT_RELU)) return false;
        if (((uintptr_t)a.y | (uintptr_t)(a.mask ? a.mask : a.y)) & 15u) return false;
    } else {
        if (a.mask) return false;
        if (a.Cin != 3 || a.Cout != 64 || !((k.kh == 9 && k.kw == 9) || (k.kh == 5 && k.kw == 5))) return false;
    }
    if ((long)a.N * a.OH * a.OW < min_pixels) return false;
    if ((long)a.H * a.W * 3 * 4 >= (1L << 31) - 4096) return false;
    ConvArgs b = a;
    b.NTX = (a.OW + kTW - 1) / kTW;
    const long units = (long)a.N * b.NTX * a.OH;
    if (units >= (1L << 31)) return false;
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else (void)hipGetLastError();
    }
    // every CU gets a range of strip rows (at least one per row phase of the waves); a tile is up to kTH of them
    const int phases = kNW / (a.Cout / 16);                     // row phases of the waves: 2 (64 channels) or 4 (32)
    const long ranges = (units + phases - 1) / phases;
    const int grid = (int)(ranges < cus ? ranges : cus);
    const size_t lds = (size_t)(kTH + k.kh - 1) * (kTW + k.kw - 1) * 3 * sizeof(float) + 16;     // (+ the zero-weighted slots past the last pixel)
    if (k.wt) hipLaunchKernelGGL((conv_pack3_kernel<5, 5, 32, true>), dim3(grid), dim3(64 * kNW), lds, s, b, (int)units);
    else if (k.kh == 9) hipLaunchKernelGGL((conv_pack3_kernel<9, 9>), dim3(grid), dim3(64 * kNW), lds, s, b, (int)units);
    else hipLaunchKernelGGL((conv_pack3_kernel<5, 5>), dim3(grid), dim3(64 * kNW), lds, s, b, (int)units);
    *err = hipGetLastError();
    return true;
}
}  // namespace srx
