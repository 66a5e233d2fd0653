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
// input pixels are staged once (bounds-checked buffer loads: zero padding for free; fetched into registers one tile
// ahead), each wave then computes whole output rows: 4 blocks of 16 input columns x 40 MFMAs, P through LDS, epilogue by
// one lane per output pixel (12-byte stores).
#include "launchers.h"
namespace srx {

// 8 MFMAs of one fragment pair, pinned: two independent accumulator chains, alternating.  (Leading s_nop: an operand may have
// been written by a VALU instruction right before the block -- the accumulators' zero initialisation.)
__device__ __forceinline__ void mfma8_kw(f32x4& a0, f32x4& a1, const float (&w)[4], const f32x4 x0, const f32x4 x1) {
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %2, %6, %0\n\t" "v_mfma_f32_16x16x4_f32 %1, %2, %10, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\t" "v_mfma_f32_16x16x4_f32 %1, %3, %11, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0\n\t" "v_mfma_f32_16x16x4_f32 %1, %4, %12, %1\n\t"
                 "v_mfma_f32_16x16x4_f32 %0, %5, %9, %0\n\t" "v_mfma_f32_16x16x4_f32 %1, %5, %13, %1"
                 : "+v"(a0), "+v"(a1)
                 : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(x0[0]), "v"(x0[1]), "v"(x0[2]), "v"(x0[3]),
                   "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]));
}

template <int KH, int KW, int CIN, int CO, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv_kwrows_kernel(const ConvArgs a, int units_total) {
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
    // The NEXT tile's input is fetched into registers while the current one is computed (24 x 16 bytes per thread: the
    // kernel owns the CU's whole register file) and written to LDS at the top of the next iteration -- one LDS buffer,
    // the global-memory latency of a tile (~4 of 10 us per tile when it was staged between the barriers) hidden.
    // Slot s of a tile <-> (row s / 64, column s % 64); out-of-image slots carry an out-of-range offset: the buffer load
    // returns the zero padding by itself.
    constexpr int TPP = CIN / 4, PPP = 64 * NW / TPP, NPASS = (TH + KH - 1) * RSW / PPP;
    static_assert(((TH + KH - 1) * RSW) % PPP == 0, "whole staging passes");
    const int c4 = tid % TPP, sp = tid / TPP;
    f32x4 pre[NPASS];
    // units are the output rows of the strips (a contiguous range per workgroup, balanced to a row); a tile = up to TH of them
    auto tile_of = [&](int u, int& n, int& h0, int& ow0, int& th, int& tw) {
        h0 = u % a.OH;
        const int t2 = u / a.OH;
        const int tx = t2 % a.NTX;
        n = t2 / a.NTX;
        ow0 = tx * TWO;
        th = a.OH - h0 < TH ? a.OH - h0 : TH;
        if (u1 - u < th) th = u1 - u;
        tw = a.OW - ow0 < TWO ? a.OW - ow0 : TWO;
    };
    auto fetch = [&](int u) {
        int n, h0, ow0, th, tw;
        tile_of(u, n, h0, ow0, th, tw);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)n * a.H * a.W * CIN, 0, a.H * a.W * CIN * 4, 0x00020000);
        const int hin = h0 - a.pad_t, win = ow0 - a.pad_l;
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int sl = sp + j * PPP;
            const int ih = hin + (sl >> 6), iw = win + (sl & 63);
            const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & ((sl >> 6) < th + KH - 1);
            pre[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? ((ih * a.W + iw) * CIN + 4 * c4) * 4 : kOobOffset, 0, 0));
        }
    };
    if (u0 < u1) fetch(u0);
    for (int u = u0; u < u1;) {
        int n, h0, ow0, th, tw;
        tile_of(u, n, h0, ow0, th, tw);
        const int un = u + th;
        lds_barrier();          // every wave is done with the previous tile
#pragma unroll
        for (int j = 0; j < NPASS; ++j) *reinterpret_cast<f32x4*>(tile + (sp + j * PPP) * PS + 4 * c4) = pre[j];
        lds_barrier();
        if (un < u1) fetch(un);
        for (int r = wave; r < th; r += NW) {
            // ---- P of output row r: NB blocks of 16 input columns, two at a time (two independent MFMA chains)
#pragma unroll
            for (int b = 0; b < NB; b += 2) {
                if (16 * b >= tw + KW - 1) continue;        // (a narrow last strip: no input columns there; wave-uniform)
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* px = tile + ((r * RSW + 16 * b + li) * PS + 4 * kq);
                // software pipeline over the KH * NG fragment pairs: the LDS reads of pair t + 1 are issued before the MFMAs
                // of pair t (hipcc on its own reuses one register set and exposes the LDS latency every 8 MFMAs: 63 %)
                constexpr int NT = KH * NG;
                f32x4 c0 = *reinterpret_cast<const f32x4*>(px), c1 = *reinterpret_cast<const f32x4*>(px + 16 * PS);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 n0 = c0, n1 = c1;
                    if (t + 1 < NT) {
                        const int kh1 = (t + 1) / NG, g1 = (t + 1) % NG;
                        n0 = *reinterpret_cast<const f32x4*>(px + kh1 * RSW * PS + 16 * g1);
                        n1 = *reinterpret_cast<const f32x4*>(px + kh1 * RSW * PS + 16 * PS + 16 * g1);
                    }
                    mfma8_kw(acc0, acc1, wa[t / NG][t % NG], c0, c1);
                    c0 = n0; c1 = n1;
                }
                // MFMA results are read by the LDS writes next: software covers the result latency (the MFMAs above are asm)
                asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));
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
        u = un;
    }
}

// ---------------------------------------------------------------------------------------------
// The same layer's FILTER GRADIENT with (kw, co) pairs as the MFMA's COLUMNS (srcnn/srcnn.py:122-130 under
// AdamOptimizer.minimize, :155-157):  dW[kh][kw][ci][co] = sum over pixels of x[r + kh][c + kw][ci] dpre[r][c][co]
//                                                          = sum over (r, q) of x[r + kh][q][ci] dpre[r][q - kw][co]    (q = an INPUT column)
// i.e. M = (kh, ci) = 5 x 32 = 160 rows (10 MFMA tiles, A operand = x straight from the staged tile, one float per lane),
// N = (kw, co) = 15 of 16 columns (B operand = the dpre row, zero-padded by KW - 1 pixels on both sides, read shifted by kw),
// K = the input columns of all rows.  wgrad_mfma_kernel has the 800 (tap, ci) pairs as rows and 16 columns for 3 output
// channels: 50 MFMAs per 4 pixels; here 10 per 4 input columns.  Same tiling and register prefetch as the forward kernel
// above; each of the 8 waves owns whole rows of a tile and keeps its own 10 accumulators, added in a fixed tree at the end;
// one partial per workgroup for reduce_partials_kernel.  The bias gradient is the sum of the B operands of the kw = 0 columns.
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CIN, int CO, int NW>
__global__ __launch_bounds__(64 * NW, 1) void wgrad_kwcols_kernel(const WgradArgs a, int units_total) {
    static_assert(KW * CO <= 16 && CIN % 16 == 0, "see conv_kwrows_kernel");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CIN>::PS;
    constexpr int NG = CIN / 16;
    constexpr int NT = KH * NG;                                // MFMA tiles: (kh, 16-channel group)
    constexpr int TH = 8, TWO = 64 - (KW - 1), RSW = 64;
    constexpr int DROW = ((RSW + KW - 1) * CO + 3) / 4 * 4;     // floats of a padded dpre row: KW - 1 zero pixels, the strip's columns, zeros
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    float* tile = lds;
    float* drows = lds + (TH + KH - 1) * RSW * PS;

    const int u0 = (int)(((long)blockIdx.x * units_total) / gridDim.x), u1 = (int)(((long)(blockIdx.x + 1) * units_total) / gridDim.x);
    constexpr int TPP = CIN / 4, PPP = 64 * NW / TPP, NPASS = (TH + KH - 1) * RSW / PPP;
    const int c4 = tid % TPP, sp = tid / TPP;
    f32x4 pre[NPASS];
    auto tile_of = [&](int u, int& n, int& h0, int& ow0, int& th, int& tw) {
        h0 = u % a.OH;
        const int t2 = u / a.OH;
        const int tx = t2 % a.NTX;
        n = t2 / a.NTX;
        ow0 = tx * TWO;
        th = a.OH - h0 < TH ? a.OH - h0 : TH;
        if (u1 - u < th) th = u1 - u;
        tw = a.OW - ow0 < TWO ? a.OW - ow0 : TWO;
    };
    auto fetch = [&](int u) {
        int n, h0, ow0, th, tw;
        tile_of(u, n, h0, ow0, th, tw);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)n * a.H * a.W * CIN, 0, a.H * a.W * CIN * 4, 0x00020000);
        const int hin = h0 - a.pad_t, win = ow0 - a.pad_l;
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int sl = sp + j * PPP;
            const int ih = hin + (sl >> 6), iw = win + (sl & 63);
            const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & ((sl >> 6) < th + KH - 1);
            pre[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? ((ih * a.W + iw) * CIN + 4 * c4) * 4 : kOobOffset, 0, 0));
        }
    };
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // B operand of lane (li = (kw, co), kq) at step st: dpre-row float (4 st + kq - kw + KW - 1) CO + co  (column 15: anything finite)
    const int kwl = li / CO, col = li - kwl * CO;
    const int boff = ((kq - (li < KW * CO ? kwl : 0) + KW - 1) * CO + (li < KW * CO ? col : 0));
    const int aoff = kq * PS + li;                              // A operand: pixel kq of the step, channel li of the tile's group

    if (u0 < u1) fetch(u0);
    for (int u = u0; u < u1;) {
        int n, h0, ow0, th, tw;
        tile_of(u, n, h0, ow0, th, tw);
        const int un = u + th;
        lds_barrier();          // every wave is done with the previous tile
#pragma unroll
        for (int j = 0; j < NPASS; ++j) *reinterpret_cast<f32x4*>(tile + (sp + j * PPP) * PS + 4 * c4) = pre[j];
        // the tile's dpre rows, zero-padded: float e of row r <-> pixel e / CO - (KW - 1), channel e % CO
        for (int i = tid; i < TH * DROW; i += 64 * NW) {
            const int r = i / DROW, e = i - r * DROW;
            const int px = e / CO - (KW - 1), co = e % CO;
            float v = 0.f;
            if (r < th && px >= 0 && px < tw) v = a.dpre[(((size_t)n * a.OH + h0 + r) * a.OW + ow0 + px) * CO + co];
            drows[i] = v;
        }
        lds_barrier();
        if (un < u1) fetch(un);
        for (int r = wave; r < th; r += NW) {
            const float* ax = tile + r * RSW * PS + aoff;
            const float* bx = drows + r * DROW + boff;
#pragma unroll
            for (int st = 0; st < RSW / 4; ++st) {
                const float b = bx[st * 4 * CO];
                if (li < CO) bsum += b;                         // the kw = 0 columns see every pixel of the row exactly once
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[((t / NG) * RSW + 4 * st) * PS + 16 * (t % NG)], b, acc[t], 0, 0, 0);
            }
        }
        u = un;
    }
    // ---- the workgroup's waves: a fixed tree through LDS (the tile area is free now)
    __syncthreads();
    constexpr int ACCF = NT * 4 + 1;
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* dst = lds + (size_t)(wave - half) * 64 * ACCF + lane * ACCF;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) dst[t * 4 + rr] = acc[t][rr];
            dst[NT * 4] = bsum;
        }
        __syncthreads();
        if (wave < half) {
            const float* src = lds + (size_t)wave * 64 * ACCF + lane * ACCF;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) acc[t][rr] += src[t * 4 + rr];
            bsum += src[NT * 4];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* pw = a.part + (size_t)blockIdx.x * a.part_stride;
        // D layout: lane (li = (kw, co), kq) holds rows 4 kq + rr of tile t: (kh, ci) = (t / NG, 16 (t % NG) + 4 kq + rr)
        if (li < KW * CO) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int kh = t / NG, ci = 16 * (t % NG) + 4 * kq + rr;
                    pw[(((size_t)kh * KW + kwl) * CIN + ci) * CO + col] = acc[t][rr];
                }
        }
        float tb = bsum;
        tb += __shfl_xor(tb, 16);
        tb += __shfl_xor(tb, 32);
        if (kq == 0 && li < CO) pw[(size_t)KH * KW * CIN * CO + li] = tb;
    }
}

// Returns true when this route took the launch (then *n_partials workgroups wrote one partial each).
bool launch_wgrad_kwcols(const ConvKey& k, const WgradArgs& a, int max_partials, long min_pixels, int* n_partials, hipStream_t s, hipError_t* err) {
    if (!(k.kh == 5 && k.kw == 5 && a.Cin == 32 && a.Cout == 3) || a.stride != 1) return false;
    if ((long)a.N * a.OH * a.OW < min_pixels) return false;
    if ((long)a.H * a.W * a.Cin * 4 >= (1L << 31) - 4096) return false;
    WgradArgs b = a;
    constexpr int TH = 8, TWO = 60, kWaves = 8;
    b.NTX = (a.OW + TWO - 1) / TWO;
    const long units = (long)a.N * b.NTX * a.OH;
    if (units >= (1L << 31)) return false;
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else (void)hipGetLastError();
    }
    const long tiles = (units + TH - 1) / TH;
    long grid = tiles < cus ? tiles : cus;
    if (grid > max_partials) grid = max_partials;
    if (grid < 1) return false;
    constexpr int DROW = ((64 + 4) * 3 + 3) / 4 * 4;
    const size_t lds = ((size_t)(TH + 4) * 64 * Lds<32>::PS + TH * DROW) * sizeof(float);
    static thread_local bool configured = false;
    if (!configured) {
        *err = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kwcols_kernel<5, 5, 32, 3, kWaves>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (*err != hipSuccess) return true;
        configured = true;
    }
    hipLaunchKernelGGL((wgrad_kwcols_kernel<5, 5, 32, 3, kWaves>), dim3((unsigned)grid), dim3(64 * kWaves), lds, s, b, (int)units);
    *err = hipGetLastError();
    *n_partials = (int)grid;
    return true;
}

// Returns true when this route took the launch.  min_pixels: below it the layer stays on conv_mfma_kernel.
bool launch_conv_kwrows(const ConvKey& k, const ConvArgs& a, long min_pixels, hipStream_t s, hipError_t* err) {
    if (k.wt || a.skip || a.mask || a.d2s_r || a.stride != 1 || a.post_relu) return false;
    if (!(k.kh == 5 && k.kw == 5 && a.Cin == 32 && a.Cout == 3)) return false;
    if ((long)a.N * a.OH * a.OW < min_pixels) return false;
    if ((long)a.H * a.W * a.Cin * 4 >= (1L << 31) - 4096) return false;       // stage_tile's 32-bit in-image offsets
    ConvArgs b = a;
    constexpr int TH = 8, TWO = 60, kWaves = 8;      // (8 waves: two per SIMD, one's epilogue under the other's MFMAs)
    b.NTX = (a.OW + TWO - 1) / TWO;
    const long units = (long)a.N * b.NTX * a.OH;           // strip rows
    if (units >= (1L << 31)) return false;
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else (void)hipGetLastError();
    }
    const long ranges = (units + 1) / 2;                   // (mid-size inputs: every CU gets rows, a tile is then shorter than TH)
    const int grid = (int)(ranges < cus ? ranges : cus);
    const size_t lds = ((size_t)(TH + 4) * 64 * Lds<32>::PS + kWaves * 64 * 20) * sizeof(float);
    // (> 64 KiB of dynamic LDS needs the attribute: raised once per host thread, outside any stream capture of later launches)
    static thread_local bool configured = false;
    if (!configured) {
        *err = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kwrows_kernel<5, 5, 32, 3, kWaves>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (*err != hipSuccess) return true;
        configured = true;
    }
    hipLaunchKernelGGL((conv_kwrows_kernel<5, 5, 32, 3, kWaves>), dim3(grid), dim3(64 * kWaves), lds, s, b, (int)units);
    *err = hipGetLastError();
    return true;
}
}  // namespace srx
