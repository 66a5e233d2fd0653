// conv_wide.hip -- 3x3 SAME stride-1 convolution (forward / data gradient) of layers WIDER than 64 channels, on
// channel-blocked tensors: x [SB][N,H,W,64] -> y [PB][N,H,W,64], filters [CIB][COB][3][3][64][64] (blocked.py).
// VGG-19's blocks 2-5 and the discriminator's 128-512-channel layers (enet/enet/model_vgg.py:65-99,
// enet/enet/model_enet.py:118-146).
//
// The 64-channel kernels serve such a layer one block pair per launch, the running sum going through HBM between
// launches: 64 launches per 512 -> 512 layer, each a few microseconds of work at VGG's 8x8 / 16x16 maps.  Here ONE
// launch covers the layer: a work unit is (a tile of <= 128 pixels of one image, one produced block of 64
// channels); its accumulators (8 sub-tiles of 16 pixels x 16 channels per wave) stay in registers while the unit
// walks the staged blocks -- per block: the wave's 3x3x64x16 filter slice (144 registers, from L2), the tile with
// its halo through LDS (stage_tile), 9 x 16 k-steps of exact-fp32 MFMA per sub-tile (the software-pipelined
// group body of conv_mfma_kernel).  Bound: fp32 MFMA; each filter slice is loaded once per 128 pixels.
// dgrad = the same walk over the OUTPUT blocks with the filters read flipped + transposed (WT).
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/srx.h"
#include "launchers.h"

namespace srx {
int set_error(int code, const char* fmt, ...);

namespace {

struct WideArgs {
    const float* x;      // staged tensor, SB blocks of [N,H,W,64]
    const float* w;      // blocked filters [CIB][COB][9][64][64] of the FORWARD layer
    const float* bias;   // [PB*64] or null
    float* y;            // produced tensor, PB blocks of [N,H,W,64]
    const float* mask;   // produced-shaped or null: y *= act'(mask) with mask_act (the activation gradient of the layer below)
    int N, H, W, SB, PB;
    int TH, TW, RS, tiles_y, tiles_x, units_total;
    float inv_rs;
    int act;             // fused after the sum over the staged blocks: NONE / RELU / LRELU
    int mask_act;
};

constexpr int kPS = 68;          // LDS pixel stride in floats (64 + 4: conflict-free ds_read_b128 over 16 pixels)
constexpr int kMaxSub = 8;       // sub-tiles of 16 pixels per unit (two groups of four accumulators)

__device__ __forceinline__ void wide_group(f32x4 (&acc)[4], const int (&laddr)[4], const float (&wr)[144], const float* lds,
                                           int row_stride) {
    constexpr int NBLK = 36;     // 9 taps x 4 groups of 16 input channels
    f32x4 cur[4], nxt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i]);
#pragma unroll
    for (int t = 0; t < NBLK; ++t) {
        if (t + 1 < NBLK) {
            const int t1 = t + 1;
            const int kh1 = (t1 / 4) / 3, kw1 = (t1 / 4) % 3, g1 = t1 % 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                nxt[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i] + kh1 * row_stride + kw1 * kPS + 16 * g1);
        }
        const int wb = (t / 4) * 16 + 4 * (t % 4);
        // GUARD: built with more than 256 registers (one workgroup per CU), so an operand may reach the block through
        // a v_accvgpr_read -- a VALU write that needs wait states before an MFMA reads it
        mfma_block<true>(acc, wr[wb], wr[wb + 1], wr[wb + 2], wr[wb + 3], cur);
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
}

template <bool WT>
__global__ __launch_bounds__(256, 1) void conv_wide_kernel(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int cout0 = wave * 16;
    const size_t blk_elems = (size_t)a.N * a.H * a.W * 64;
    const int row_stride = a.RS * kPS;
    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    __builtin_amdgcn_s_setprio(2);
    for (int u = u0; u < u1; ++u) {
        // units of one tile are adjacent (the produced blocks re-read the same staged tiles: L2 serves them)
        const int pb = u % a.PB;
        int tile = u / a.PB;
        const int tx = tile % a.tiles_x; tile /= a.tiles_x;
        const int n = tile / a.tiles_y;
        const int h0 = (tile % a.tiles_y) * a.TH;
        const int ox = tx * a.TW;
        const int th = (a.H - h0 < a.TH) ? (a.H - h0) : a.TH;
        const int tw = (a.W - ox < a.TW) ? (a.W - ox) : a.TW;
        const int npx = th * tw;
        const int n_sub = (npx + 15) >> 4;
        const int n_need = (th + 2) * a.RS + 2;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + pb * 64 + cout0 + 4 * kq);
        f32x4 acc0[4], acc1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc0[i] = bias4; acc1[i] = bias4; }
        const float inv_w = 1.0f / (float)tw;
        // lane's pixel of sub-tile i -> LDS address of its (tap 0,0) slot (sub-tiles past the tile compute on pixel 0 and
        // are not stored).  Recomputed per group rather than kept: the kernel sits at the 256-register limit of two
        // waves per SIMD (144 filter registers + 32 accumulators + 32 LDS fragments), and a dozen integer
        // instructions per 576 MFMAs cost nothing.
        auto group_addresses = [&](int first, int (&la)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (first + i) + li;
                const int tt = (t < npx) ? t : 0;
                const int orow = fdiv_small(tt, inv_w, tw);
                const int ocol = tt - orow * tw;
                la[i] = (orow * a.RS + ocol) * kPS + 4 * kq;
            }
        };
        for (int sb = 0; sb < a.SB; ++sb) {
            float wr[144];
            ConvArgs wa;
            const int wblock = WT ? (pb * a.SB + sb) : (sb * a.PB + pb);
            wa.w = a.w + (size_t)wblock * (9 * 64 * 64);
            wa.Cin = 64;
            wa.Cout = 64;
            // One workgroup per CU: 144 filter registers + 32 accumulators + the staging loads in flight do not fit the
            // 256 registers of two waves per SIMD (tried: 28-138 spilled registers).  The filter loads are issued
            // first and land while the tile is staged.
            load_stationary_weights<9, 64, WT>(wr, wa, cout0, li, kq);
            lds_barrier();                                                     // every wave is done with the previous tile
            stage_tile<64>(lds, a.x + (size_t)sb * blk_elems, n, a.H, a.W, 64, h0 - 1, ox - 1, a.RS, a.inv_rs, n_need, tid);
            lds_barrier();
            __builtin_amdgcn_s_setprio(0);
            {
                int la[4];
                group_addresses(0, la);
                wide_group(acc0, la, wr, lds, row_stride);
            }
            if (n_sub > 4) {
                int la[4];
                group_addresses(4, la);
                wide_group(acc1, la, wr, lds, row_stride);
            }
            __builtin_amdgcn_s_setprio(2);
        }
        // MFMA results are read by VALU code next: software covers the result latency
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        const size_t img_off = (size_t)pb * blk_elems + (size_t)n * a.H * a.W * 64 + cout0 + 4 * kq;
        float* yb = a.y + img_off;
        const float* mb = a.mask ? a.mask + img_off : nullptr;
        const float slope = act_slope(a.act), mslope = act_slope(a.mask_act);
        auto store = [&](int first, const f32x4 (&acc)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (first + i) + li;
                if (t < npx) {
                    const int orow = fdiv_small(t, inv_w, tw);
                    const int ocol = t - orow * tw;
                    const size_t o = ((size_t)(h0 + orow) * a.W + ox + ocol) * 64;
                    f32x4 v = act_apply4(acc[i], a.act, slope);
                    if (mb) v = act_grad4(v, *reinterpret_cast<const f32x4*>(mb + o), a.mask_act, mslope);
                    *reinterpret_cast<f32x4*>(yb + o) = v;
                }
            }
        };
        store(0, acc0);
        if (n_sub > 4) store(4, acc1);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same walk, software-pipelined.  In conv_wide_kernel every (unit, staged block) step starts with its filter
// slice and its tile exposed: ~9 us of loads and barriers in front of 17.4 us of MFMA work.  Here a step's MFMA
// groups carry the NEXT step's loads:
//   * two LDS buffers; the next tile's loads are issued before a group and written to the other buffer after it
//     (half the passes per group: <= 36 registers in flight), one barrier per step;
//   * the next filter slice replaces the current one IN PLACE during the step's last group: the four registers
//     of block t are dead once block t has issued, so their loads go out right behind it -- no second register set.
//     They are consumed in the same order by the next step's first group, so only the first few must have landed
//     when it starts (the compiler's counted waits do the rest).
// Steps are numbered flat, s = unit * SB + staged block, so that "next" needs no special cases.
constexpr int kPasses = 17;                 // staging passes of 16 pixel slots (256 threads, 16 per pixel)
constexpr int kBufSlots = 16 * kPasses;     // slots per LDS buffer: every pass writes, the tail slots are never read

struct WideStep {                           // wave-uniform
    int n, h0, ox, th, tw, pb, sb;
};

// The four filter values of block t (tap t / 4, input channels 16 (t % 4) ..) of the 64 x 64 slice behind `wrs`.
template <bool WT>
__device__ __forceinline__ void wide_load_w(float (&wr)[144], int t, __amdgpu_buffer_rsrc_t wrs, int wvlane) {
    const int tap = t / 4, g = t % 4;
    if constexpr (!WT) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            wr[tap * 16 + 4 * g + e] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wvlane, ((tap * 64 + 16 * g + e) * 64) * 4, 0));
    } else {
        const f32x4 v = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvlane, (((8 - tap) * 64) * 64 + 16 * g) * 4, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[tap * 16 + 4 * g + e] = v[e];
    }
}

// staging pass j of a tile (16 pixel slots, 16 threads x 16 bytes each): the load, bounds-checked -- slots outside the
// image or past the tile read as zero ...
struct WideStage {                          // what a tile's staging needs (wave-uniform except sp / c4)
    __amdgpu_buffer_rsrc_t xrs;
    float* nbuf;
    int h_in0, w_in0, H, W, RS, n_need, sp, c4;
    float inv_rs;
};
__device__ __forceinline__ f32x4 wide_issue_pass(const WideStage& g, int j) {
    const int s = g.sp + 16 * j;
    const int r = fdiv_small(s, g.inv_rs, g.RS);
    const int c = s - r * g.RS;
    const int ih = g.h_in0 + r, iw = g.w_in0 + c;
    const bool ok = ((unsigned)ih < (unsigned)g.H) & ((unsigned)iw < (unsigned)g.W) & (s < g.n_need);
    return __builtin_bit_cast(
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(g.xrs, ok ? ((ih * g.W + iw) * 64 + 4 * g.c4) * 4 : kOobOffset, 0, 0));
}
// ... and its LDS write
__device__ __forceinline__ void wide_commit_pass(const WideStage& g, int j, const f32x4 v) {
    *reinterpret_cast<f32x4*>(g.nbuf + (g.sp + 16 * j) * kPS + 4 * g.c4) = v;
}

// One group of four sub-tiles: 36 blocks (tap x 16 input channels) of four k-steps, each k-step one statement of four
// MFMAs with the filter value in an accumulation register.  Everything else is dealt out over the gaps between the
// k-steps, so that the matrix pipe never waits for a run of other instructions:
//   gaps 0..2 of a block:   the LDS fragments of the next block;
//   WNEXT:                  the next step's filter value(s) into the register(s) the k-step just consumed;
//   gap 1 / gap 2:          one staging load / one staging LDS write of the next tile, per MODE:
//       0: passes 0..8  issued in blocks 0..8, written in blocks 18..26        (first group of a two-group step)
//       1: passes 9..16 likewise                                               (second group)
//       2: passes 0..8 in blocks 0..8 / 9..17, passes 9..16 in blocks 18..25 / 27..34   (one-group step)
template <bool WT, bool WNEXT, int MODE>
__device__ __forceinline__ void wide_group_pipe(f32x4 (&acc)[4], const int (&laddr)[4], float (&wr)[144], const float* lds,
                                                int row_stride, __amdgpu_buffer_rsrc_t wrs, int wvlane, WideStage g) {
    constexpr int NBLK = 36;
    constexpr int NA = (MODE == 1) ? kPasses - 9 : 9;
    constexpr int JA = (MODE == 1) ? 9 : 0;
    constexpr int CA = (MODE == 2) ? 9 : 18;          // first writing block of the first batch
    constexpr int NB = (MODE == 2) ? kPasses - 9 : 0; // second batch (one-group steps)
    // (opaque: the slot geometry is a dozen instructions per pass; hoisted out of the step loop it would pin two
    // registers per pass for the whole kernel)
    asm volatile("" : "+v"(g.sp));
    f32x4 stg[9];
    f32x4 cur[4], nxt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i]);
#pragma unroll
    for (int t = 0; t < NBLK; ++t) {
        const int wb = (t / 4) * 16 + 4 * (t % 4);
        const int t1 = t + 1;
        const int kh1 = (t1 / 4) / 3, kw1 = (t1 / 4) % 3, g1 = t1 % 4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            mfma_sub_a(acc, wr[wb + ks], cur[0][ks], cur[1][ks], cur[2][ks], cur[3][ks]);
            // ---- gap ks
            // (fragments in gaps 0, 1, 2: the last one has a whole k-step to land before the next block needs it)
            auto frag = [&](int i) {
                nxt[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i] + kh1 * row_stride + kw1 * kPS + 16 * g1);
            };
            if (t1 < NBLK && ks < 2) frag(ks);
            if (t1 < NBLK && ks == 2) { frag(2); frag(3); }
            // next step's filter values, in place: the register of k-step ks is dead once the k-step has issued
            if constexpr (WNEXT && !WT)
                wr[wb + ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                            wrs, wvlane, (((t / 4) * 64 + 16 * (t % 4) + ks) * 64) * 4, 0));
            if constexpr (WNEXT && WT) {
                if (ks == 3) {
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                  wrs, wvlane, (((8 - t / 4) * 64) * 64 + 16 * (t % 4)) * 4, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) wr[wb + e] = v[e];
                }
            }
            if (ks == 1) {
                if (t < NA) stg[t] = wide_issue_pass(g, JA + t);
                if (NB > 0 && t >= 18 && t < 18 + NB) stg[t - 18] = wide_issue_pass(g, 9 + t - 18);
            }
            if (ks == 2) {
                if (t >= CA && t < CA + NA) wide_commit_pass(g, JA + t - CA, stg[t - CA]);
                if (NB > 0 && t >= 27 && t < 27 + NB) wide_commit_pass(g, 9 + t - 27, stg[t - 27]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
}

// staging passes [J0, J1) of a tile: loads (bounds-checked: slots outside the image or past the tile read as zero) ...
template <int J0, int J1>
__device__ __forceinline__ void wide_issue(f32x4 (&v)[J1 - J0], __amdgpu_buffer_rsrc_t xrs, int h_in0, int w_in0, int H, int W,
                                           int RS, float inv_rs, int n_need, int sp, int c4) {
    // (opaque: the slot geometry below is a dozen instructions per pass; hoisted out of the step loop it would pin
    // two registers per pass for the whole kernel)
    asm volatile("" : "+v"(sp));
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int s = sp + 16 * j;
        const int r = fdiv_small(s, inv_rs, RS);
        const int c = s - r * RS;
        const int ih = h_in0 + r, iw = w_in0 + c;
        const bool ok = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W) & (s < n_need);
        v[j - J0] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? ((ih * W + iw) * 64 + 4 * c4) * 4 : kOobOffset, 0, 0));
    }
}
// ... and LDS writes
template <int J0, int J1>
__device__ __forceinline__ void wide_commit(float* buf, const f32x4 (&v)[J1 - J0], int sp, int c4) {
#pragma unroll
    for (int j = J0; j < J1; ++j) *reinterpret_cast<f32x4*>(buf + (sp + 16 * j) * kPS + 4 * c4) = v[j - J0];
}

// MASK (template): the launch has a mask operand.  A compile-time property: a conditional load makes hipcc's wait-count
// bookkeeping assume the load may be pending, and every later wait then drains the whole queue -- filter loads and
// output stores included.  For the same reason the bias comes from LDS, and the epilogue below is loads-free.
// ONE (template): tiles of <= 64 pixels, one group per step; otherwise every step runs two groups (on an edge tile of
// <= 64 pixels the second one computes on pixel 0 and stores nothing).  Not a run-time branch: where the two forms
// joined, the compiler kept the next filter slice in different registers on either side and drained the queue to
// move it.
template <bool WT, bool MASK, bool ONE>
__global__ __launch_bounds__(256, 1) void conv_wide_pipe_kernel(const WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int JH = 9;                                   // passes carried by the first group (the rest by the second)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int cout0 = wave * 16;
    const int c4 = tid & 15, sp = tid >> 4;
    const size_t blk_elems = (size_t)a.N * a.H * a.W * 64;
    const int img_bytes = a.H * a.W * 64 * 4;
    const int row_stride = a.RS * kPS;
    const long G_ = gridDim.x;
    const int s_begin = (int)(((long)blockIdx.x * a.units_total) / G_) * a.SB;
    const int s_end = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_) * a.SB;
    if (s_begin >= s_end) return;
    const int wvlane = WT ? ((cout0 + li) * 64 + 4 * kq) * 4 : (4 * kq * 64 + cout0 + li) * 4;

    auto decode = [&](int s, WideStep& d) {
        const int u = s / a.SB;
        d.sb = s - u * a.SB;
        d.pb = u % a.PB;
        int tile = u / a.PB;
        const int tx = tile % a.tiles_x;
        tile /= a.tiles_x;
        d.n = tile / a.tiles_y;
        d.h0 = (tile % a.tiles_y) * a.TH;
        d.ox = tx * a.TW;
        d.th = (a.H - d.h0 < a.TH) ? (a.H - d.h0) : a.TH;
        d.tw = (a.W - d.ox < a.TW) ? (a.W - d.ox) : a.TW;
    };
    // resources of a step's operands; `live` false: zero records, every load reads as zero without touching memory
    auto x_rsrc = [&](const WideStep& d, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)d.sb * blk_elems + (size_t)d.n * a.H * a.W * 64, 0,
                                                 live ? img_bytes : 0, 0x00020000);
    };
    auto w_rsrc = [&](const WideStep& d, bool live) {
        const int wblock = WT ? (d.pb * a.SB + d.sb) : (d.sb * a.PB + d.pb);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w) + (size_t)wblock * (9 * 64 * 64), 0,
                                                 live ? 9 * 64 * 64 * 4 : 0, 0x00020000);
    };

    float* lds_bias = lds + 2 * kBufSlots * kPS;             // [PB * 64], behind the two tile buffers
    for (int i = tid; i < a.PB * 64; i += 256) lds_bias[i] = a.bias ? a.bias[i] : 0.0f;
    const int chb = (cout0 + 4 * kq) * 4;                    // the lane's four output channels, bytes inside a pixel
    WideStep d;
    decode(s_begin, d);
    float wr[144];
    {
        const __amdgpu_buffer_rsrc_t wrs = w_rsrc(d, true);
#pragma unroll
        for (int t = 0; t < 36; ++t) wide_load_w<WT>(wr, t, wrs, wvlane);
        const __amdgpu_buffer_rsrc_t xrs = x_rsrc(d, true);
        const int n_need = (d.th + 2) * a.RS + 2;
        {
            f32x4 v0[JH];
            wide_issue<0, JH>(v0, xrs, d.h0 - 1, d.ox - 1, a.H, a.W, a.RS, a.inv_rs, n_need, sp, c4);
            wide_commit<0, JH>(lds, v0, sp, c4);
        }
        f32x4 v1[kPasses - JH];
        wide_issue<JH, kPasses>(v1, xrs, d.h0 - 1, d.ox - 1, a.H, a.W, a.RS, a.inv_rs, n_need, sp, c4);
        wide_commit<JH, kPasses>(lds, v1, sp, c4);
    }
    lds_barrier();
    int cur_buf = 0;
    f32x4 acc0[4], acc1[4];
    for (int s = s_begin; s < s_end; ++s) {
        const bool has_next = s + 1 < s_end;
        WideStep dn = d;
        if (has_next) decode(s + 1, dn);
        const __amdgpu_buffer_rsrc_t xrs = x_rsrc(dn, has_next);
        const __amdgpu_buffer_rsrc_t wrs = w_rsrc(dn, has_next);
        const int n_need = (dn.th + 2) * a.RS + 2;
        const float* buf = lds + cur_buf * (kBufSlots * kPS);
        float* nbuf = lds + (cur_buf ^ 1) * (kBufSlots * kPS);
        const int npx = d.th * d.tw;
        const float inv_w = 1.0f / (float)d.tw;
        if (d.sb == 0) {
            const f32x4 bias4 = *reinterpret_cast<const f32x4*>(lds_bias + d.pb * 64 + cout0 + 4 * kq);
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc0[i] = bias4; acc1[i] = bias4; }
        }
        // the unit's output pixels: byte offset of (sub-tile m, this lane) inside the image of the produced block, or
        // the out-of-range offset (sub-tiles past the tile; every step but the unit's last) -- such stores are dropped,
        // such loads read zero
        const bool last_sb = d.sb == a.SB - 1;
        int eo[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) eo[m] = kOobOffset;
        if (last_sb) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int t = 16 * m + li;
                const int orow = fdiv_small(t, inv_w, d.tw);
                const int ocol = t - orow * d.tw;
                if (t < npx) eo[m] = (((d.h0 + orow) * a.W + d.ox + ocol) * 64) * 4 + chb;
            }
        }
        const size_t unit_base = (size_t)d.pb * blk_elems + (size_t)d.n * a.H * a.W * 64;
        // the mask operand of the whole unit, fetched before the step's last group (lands during it); straight-line:
        // on the other steps the offsets are out of range
        f32x4 mk[MASK ? 8 : 1];
        auto fetch_mask = [&]() {
            if constexpr (MASK) {
                const __amdgpu_buffer_rsrc_t mrs =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask) + unit_base, 0, img_bytes, 0x00020000);
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    mk[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, eo[m], 0, 0));
            }
        };
        auto group_addresses = [&](int first, int (&la)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * (first + i) + li;
                const int tt = (t < npx) ? t : 0;
                const int orow = fdiv_small(tt, inv_w, d.tw);
                const int ocol = tt - orow * d.tw;
                la[i] = (orow * a.RS + ocol) * kPS + 4 * kq;
            }
        };
        WideStage sg;
        sg.xrs = xrs; sg.nbuf = nbuf; sg.h_in0 = dn.h0 - 1; sg.w_in0 = dn.ox - 1; sg.H = a.H; sg.W = a.W; sg.RS = a.RS;
        sg.n_need = n_need; sg.sp = sp; sg.c4 = c4; sg.inv_rs = a.inv_rs;
        if constexpr (!ONE) {
            {
                int la[4];
                group_addresses(0, la);
                wide_group_pipe<WT, false, 0>(acc0, la, wr, buf, row_stride, wrs, wvlane, sg);
            }
            fetch_mask();
            int la[4];
            group_addresses(4, la);
            wide_group_pipe<WT, true, 1>(acc1, la, wr, buf, row_stride, wrs, wvlane, sg);
        } else {
            fetch_mask();
            int la[4];
            group_addresses(0, la);
            wide_group_pipe<WT, true, 2>(acc0, la, wr, buf, row_stride, wrs, wvlane, sg);
        }
        lds_barrier();                   // the next tile is complete; nobody reads this one any more
        cur_buf ^= 1;
        if (last_sb) {
            // MFMA results are read by VALU code next: software covers the result latency
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + unit_base, 0, img_bytes, 0x00020000);
            const float slope = act_slope(a.act), mslope = act_slope(a.mask_act);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                f32x4 v = act_apply4(m < 4 ? acc0[m & 3] : acc1[m & 3], a.act, slope);
                if constexpr (MASK) v = act_grad4(v, mk[m], a.mask_act, mslope);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yrs, eo[m], 0, 0);
            }
        }
        d = dn;
    }
}

}  // namespace
}  // namespace srx

using namespace srx;

extern "C" int srx_conv3x3_blocked(const float* x, const float* w, const float* bias, const float* mask, int mask_act,
                                   float* y, int N, int H, int W, int staged_blocks, int produced_blocks, int act,
                                   int transpose_filters, srx_stream_t stream) {
    if (!x || !w || !y) return set_error(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (N <= 0 || H <= 0 || W <= 0 || staged_blocks <= 0 || produced_blocks <= 0)
        return set_error(SRX_ERR_BAD_ARG, "non-positive dimension");
    if (act != SRX_ACT_NONE && act != SRX_ACT_RELU && act != SRX_ACT_LRELU)
        return set_error(SRX_ERR_UNSUPPORTED, "conv3x3_blocked: activation must be none, relu or leaky relu");
    if (mask && (mask_act < SRX_ACT_NONE || mask_act > SRX_ACT_SIGMOID)) return set_error(SRX_ERR_BAD_ARG, "bad mask_act");
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)mask) & 15u)
        return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    if (mask == y) return set_error(SRX_ERR_BAD_ARG, "conv3x3_blocked: the mask operand cannot alias the output");
    if ((long)N * H * W * 64 >= (1L << 31) / 4) return set_error(SRX_ERR_UNSUPPORTED, "conv3x3_blocked: block beyond 32-bit offsets");
    WideArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.mask = mask; a.mask_act = mask ? mask_act : 0;
    a.N = N; a.H = H; a.W = W; a.SB = staged_blocks; a.PB = produced_blocks;
    // <= 128 pixels per unit.  Rows of up to 64 pixels: full-width tiles (the zero column left of row r+1 doubles as the
    // right padding of row r); wider images: column strips of 64 with their own halo columns.
    int tw = W <= 64 ? W : 64;
    if (W > 64) {                          // even the strips out: 130 -> 44 + 43 + 43, not 64 + 64 + 2
        const int nx = (W + 63) / 64;
        tw = (W + nx - 1) / nx;
    }
    int th = (16 * kMaxSub) / tw;
    if (th < 1) th = 1;
    if (th > H) th = H;
    a.TH = th; a.TW = tw;
    a.RS = W <= 64 ? W + 1 : tw + 2;
    a.inv_rs = 1.0f / (float)a.RS;
    a.tiles_y = (H + th - 1) / th;
    a.tiles_x = (W + tw - 1) / tw;
    const long units = (long)N * a.tiles_y * a.tiles_x * produced_blocks;
    if (units >= (1L << 31)) return set_error(SRX_ERR_UNSUPPORTED, "conv3x3_blocked: too many work units");
    a.units_total = (int)units;
    a.act = act;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0 || cus > 256) cus = 256;
    const int grid = (int)(units < (long)cus ? units : (long)cus);      // one persistent workgroup per CU
    const int n_need = (th + 2) * a.RS + 2;
    hipError_t e;
    // SRX_WIDE_PIPE=0: the unpipelined kernel (A/B)
    static const bool use_pipe = [] { const char* v = getenv("SRX_WIDE_PIPE"); return !(v && v[0] == '0'); }();
    const size_t lds_pipe = ((size_t)2 * kBufSlots * kPS + (size_t)produced_blocks * 64) * 4;
    if (use_pipe && n_need <= kBufSlots && lds_pipe <= 160 * 1024) {
        const hipStream_t st = (hipStream_t)stream;
        const int variant = (transpose_filters ? 4 : 0) | (mask ? 2 : 0) | (th * tw <= 64 ? 1 : 0);
        switch (variant) {
            case 0: e = launch_with_lds(conv_wide_pipe_kernel<false, false, false>, a, grid, lds_pipe, st); break;
            case 1: e = launch_with_lds(conv_wide_pipe_kernel<false, false, true>, a, grid, lds_pipe, st); break;
            case 2: e = launch_with_lds(conv_wide_pipe_kernel<false, true, false>, a, grid, lds_pipe, st); break;
            case 3: e = launch_with_lds(conv_wide_pipe_kernel<false, true, true>, a, grid, lds_pipe, st); break;
            case 4: e = launch_with_lds(conv_wide_pipe_kernel<true, false, false>, a, grid, lds_pipe, st); break;
            case 5: e = launch_with_lds(conv_wide_pipe_kernel<true, false, true>, a, grid, lds_pipe, st); break;
            case 6: e = launch_with_lds(conv_wide_pipe_kernel<true, true, false>, a, grid, lds_pipe, st); break;
            default: e = launch_with_lds(conv_wide_pipe_kernel<true, true, true>, a, grid, lds_pipe, st); break;
        }
    } else {
        const size_t lds = (size_t)n_need * kPS * 4;
        if (transpose_filters) e = launch_with_lds(conv_wide_kernel<true>, a, grid, lds, (hipStream_t)stream);
        else e = launch_with_lds(conv_wide_kernel<false>, a, grid, lds, (hipStream_t)stream);
    }
    if (e != hipSuccess) return set_error(SRX_ERR_LAUNCH, "conv3x3_blocked launch failed: %s", hipGetErrorString(e));
    return SRX_OK;
}
