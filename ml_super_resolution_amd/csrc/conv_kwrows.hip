// conv_kwrows.hip -- forward convolution into a FEW output channels with many input channels and a large filter:
// SRCNN's reconstruction layer, 5x5 32 -> 3, tanh (srcnn/srcnn.py:122-130).  On conv_mfma_kernel the MFMA's 16 rows are
// output channels: 16 computed, 3 used -- 200 MFMAs per 16 pixels, 15 useful TFLOP/s and the input read at 0.4 TB/s
// (profiles/r03_time_srcnn_image.txt).  Here the rows are (kw, co) PAIRS, 5 x 3 = 15 of 16:
//
//     P[kw*3+co][q] = sum over (kh, ci) of w[kh][kw][ci][co] * x[row + kh][q][ci]        q = an INPUT column
//     y[c][co]      = act(bias[co] + sum over kw of P[kw*3+co][c + kw])                  c = an output column
//
// i.e. the MFMA's K dimension runs over (kh, ci) only -- 5 x 32 / 4 = 40 MFMAs per 16 input columns instead of 200 per 16
// output pixels -- and the five kw partial sums of an output are added in the epilogue, through LDS (a wave writes its
// row's P, [input column][16 rows], and reads it back shifted by kw).  Exact fp32 like every other kernel of the
// library; the summation ORDER differs from conv_mfma_kernel's (there: one chain over all 25 taps), so the results
// agree with it to rounding, not bit for bit -- which is why the launcher takes this route only for problems beyond the
// window of the one-launch SRCNN kernel (whose tests demand bit-equality with the per-layer MFMA launches).
//
// A persistent workgroup (one per CU, 128 KiB of LDS) owns tiles of 8 output rows x 60 output columns: the (8 + KH - 1) x 64
// input pixels are staged once (stage_tile: zero padding by selects), each wave then computes whole output rows: 4 blocks
// of 16 input columns x 40 MFMAs, P through LDS, epilogue by one lane per output pixel (12-byte stores).
#include "launchers.h"
namespace srx {

template <int KH, int KW, int CIN, int CO>
__global__ __launch_bounds__(256, 1) void conv_kwrows_kernel(const ConvArgs a, int units_total, int tiles_per_col) {
    static_assert(KW * CO <= 16, "the (kw, co) pairs are the 16 rows of the MFMA");
    static_assert(CIN % 16 == 0 && CIN >= 16, "input channels in groups of 16 (one ds_read_b128 per lane and group)");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CIN>::PS;         // padded pixel stride (floats)
    constexpr int NG = CIN / 16;
    constexpr int TH = 8, TWO = 64 - (KW - 1), RSW = 64;      // output rows / output columns / input columns of a tile
    constexpr int PST = 20;                  // floats per input column in the P buffer (16 rows + 4: 16-byte aligned, fewer bank conflicts than 16)
    constexpr int NB = RSW / 16;             // blocks of 16 input columns per row
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    float* tile = lds;
    float* pbuf = lds + (TH + KH - 1) * RSW * PS + wave * (RSW * PST);

    // stationary A operands: row m = li <-> (kw, co) = (li / CO, li % CO); k = channel 16 g + 4 kq + s of tap row kh
    float wa[KH][NG][4];
    {
        const int kw = li / CO, co = li - kw * CO;
        const bool row_ok = li < KW * CO;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int ci = 16 * g + 4 * kq + s;
                    const float v = a.w[row_ok ? (((size_t)(kh * KW + kw) * CIN + ci) * CO + co) : 0];
                    wa[kh][g][s] = row_ok ? v : 0.0f;
                }
    }
    float bias[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) bias[c] = a.bias ? a.bias[c] : 0.0f;

    // a contiguous range of tiles per workgroup: it walks down a column strip, and the KH - 1 halo rows a tile shares with
    // the one above it were read by the same CU a moment ago (L2 hits)
    const int u0 = (int)(((long)blockIdx.x * units_total) / gridDim.x), u1 = (int)(((long)(blockIdx.x + 1) * units_total) / gridDim.x);
    for (int u = u0; u < u1; ++u) {
        // unit -> (image, column strip, tile row), the tile row running fastest
        const int ti = u % tiles_per_col;
        const int t2 = u / tiles_per_col;
        const int tx = t2 % a.NTX, n = t2 / a.NTX;
        const int h0 = ti * TH, ow0 = tx * TWO;
        const int th = a.OH - h0 < TH ? a.OH - h0 : TH;
        const int tw = a.OW - ow0 < TWO ? a.OW - ow0 : TWO;
        lds_barrier();          // every wave is done with the previous tile
        stage_tile<CIN>(tile, a.x, n, a.H, a.W, a.Cin, h0 - a.pad_t, ow0 - a.pad_l, RSW, 1.0f / (float)RSW, (th + KH - 1) * RSW, tid);
        lds_barrier();
        for (int r = wave; r < th; r += 4) {
            // ---- P of output row r: NB blocks of 16 input columns, two at a time (two independent MFMA chains)
#pragma unroll
            for (int b = 0; b < NB; b += 2) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* px = tile + ((r * RSW + 16 * b + li) * PS + 4 * kq);
#pragma unroll
                for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(px + kh * RSW * PS + 16 * g);
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(px + kh * RSW * PS + 16 * PS + 16 * g);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kh][g][s], x0[s], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[kh][g][s], x1[s], acc1, 0, 0, 0);
                        }
                    }
                // lane (li, kq) holds rows 4 kq .. 4 kq + 3 of input column 16 b + li
                *reinterpret_cast<f32x4*>(pbuf + (16 * b + li) * PST + 4 * kq) = acc0;
                *reinterpret_cast<f32x4*>(pbuf + (16 * (b + 1) + li) * PST + 4 * kq) = acc1;
            }
            // (the P buffer belongs to this wave alone: its own LDS operations complete in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- epilogue: lane c = output column c of the strip
            if (lane < tw) {
                float o[CO];
#pragma unroll
                for (int c = 0; c < CO; ++c) o[c] = bias[c];
#pragma unroll
                for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                    for (int c = 0; c < CO; ++c) o[c] += pbuf[(lane + kw) * PST + kw * CO + c];
                float* yo = a.y + (((size_t)n * a.OH + h0 + r) * a.OW + ow0 + lane) * CO;
#pragma unroll
                for (int c = 0; c < CO; ++c) yo[c] = act_apply(o[c], a.act);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads above, before the next row's P overwrites them
        }
    }
}

// Returns true when this route took the launch.  min_pixels: below it the layer stays on conv_mfma_kernel.
bool launch_conv_kwrows(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err) {
    if (k.wt || a.skip || a.mask || a.d2s_r || a.stride != 1 || a.post_relu) return false;
    if (!(k.kh == 5 && k.kw == 5 && a.Cin == 32 && a.Cout == 3)) return false;
    if ((long)a.N * a.OH * a.OW < min_pixels) return false;
    if ((long)a.H * a.W * a.Cin * 4 >= (1L << 31) - 4096) return false;       // stage_tile's 32-bit in-image offsets
    ConvArgs b = a;
    constexpr int TH = 8, TWO = 60;
    b.NTX = (a.OW + TWO - 1) / TWO;
    const int tiles_per_col = (a.OH + TH - 1) / TH;
    const long units = (long)a.N * b.NTX * tiles_per_col;
    if (units >= (1L << 31)) return false;
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else (void)hipGetLastError();
    }
    const int grid = (int)(units < cus ? units : cus);
    const size_t lds = ((size_t)(TH + 4) * 64 * Lds<32>::PS + 4 * 64 * 20) * sizeof(float);
    // (> 64 KiB of dynamic LDS needs the attribute: raised once per host thread, outside any stream capture of later launches)
    static thread_local bool configured = false;
    if (!configured) {
        *err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kwrows_kernel<5, 5, 32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (*err != hipSuccess) return true;
        configured = true;
    }
    hipLaunchKernelGGL((conv_kwrows_kernel<5, 5, 32, 3>), dim3(grid), dim3(256), lds, s, b, (int)units, tiles_per_col);
    *err = hipGetLastError();
    return true;
}
}  // namespace srx
