// conv_kernels.hip.h -- gfx950 (MI355X) implicit-GEMM convolution kernels, exact fp32 on
// v_mfma_f32_16x16x4_f32.
//
// Design:
//  * NHWC activations, HWIO filters, stride 1.  A persistent workgroup (4 waves) owns a contiguous range
//    of output rows ("units").  A tile's input halo (TH+KH-1 rows) is staged ONCE into LDS with a padded
//    pixel stride (Cin+4 floats) and explicit zero padding, then read KH*KW times; tap offsets become
//    ds_read immediates.
//  * forward / dgrad: the layer's weights for a wave's 16 output channels are STATIONARY IN REGISTERS for
//    the whole kernel (3x3x64 -> 144 registers); MFMA A = weights (rows = Cout), B = pixels (cols), so
//    each lane ends up holding 4 consecutive output channels of one pixel -> one 16-byte store.
//    Activation / residual add / upstream activation-gradient mask are fused in the epilogue.
//      - conv_pipe_kernel (the 3x3 body layers): one workgroup per CU, weights in AGPRs, the tile
//        double-buffered, everything that is not an MFMA done by scalar and memory instructions;
//        conv_pipe_strip_kernel: the same on column strips, for images too wide for full-width tiles;
//      - conv_mfma_kernel / conv_mfma_generic_kernel (all other shapes): two workgroups per CU that stage
//        between their MFMA phases;
//      - the 64 <-> 3 channel layers have plain-FMA kernels of their own in conv_narrow.hip.
//  * wgrad: the dW accumulators (144 VGPRs for 3x3x64 per 16 Cout) are stationary for the whole kernel;
//    x comes from the same LDS halo tile, dpre straight from global (each element is used by exactly one
//    wave).  Per-workgroup partials are reduced by a second, fixed-order kernel.  wgrad_pipe_kernel /
//    wgrad_lin_kernel walk the padded tile positions (VALU-free K loop; one double-buffered workgroup per CU /
//    two workgroups per CU), wgrad_mfma_kernel the real pixels with per-lane cursors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace srx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_LRELU = 3, ACT_SIGMOID = 4 };

struct ConvArgs {
    const float* x;     // input  [N,H,W,Cin]
    const float* w;     // filters, HWIO of the FORWARD layer
    const float* bias;  // [Cout] or null
    const float* skip;  // output-shaped or null (added after the activation)
    const float* mask;  // output-shaped or null: out *= act'(mask) with mask_act  (dgrad)
    float* y;           // output [N,OH,OW,Cout]
    int N, H, W, OH, OW, Cin, Cout;
    int pad_t, pad_l;
    int TH, TW, NTX, RS;  // output tile, #column tiles, LDS row stride (slots)
    int units_total;      // N*NTX*OH
    int act, post_relu, mask_act;
    float inv_rs;         // 1/RS
    int stagger;          // s_sleep(127) count for the second resident workgroup (0 = off)
    int dbg;              // diagnostic timing knobs, compiled in only with -DSRX_TRACE (SRX_DBG: 1 = stage only the first tile, 2 = no stores)
    unsigned long long* trace;  // diagnostic build (-DSRX_TRACE) only: per-wave cycle stamps
    int buf_floats;       // pipelined kernel: floats per LDS tile buffer (two buffers)
    int* tile_counter;    // dynamic scheduling (two-workgroup kernels): next tile to hand out, preset to gridDim.x; null = static
    int tiles_total, tiles_per_col;   // N*NTX*tiles_per_col tiles of TH rows (the last of a column may be shorter)
    int lds_sched_slot;   // float index in LDS of the 4-byte mailbox used to broadcast the tile index
    int stride;           // two-workgroup kernels, forward only: 1, or 2 (tiles with their own halo: RS = (TW-1)*2 + KW slots per row)
    int d2s_r, d2s_rc;    // sub-pixel store mode (two-workgroup kernels): r > 1 -> y is [N,OH*r,OW*r,Cout/(r*r)] and channel
                          // ch of LR pixel (h,w) is stored at HR row h*r + ch/rc, column offset w*rc + ch%rc, rc = r*C
};

struct WgradArgs {
    const float* x;     // layer input [N,H,W,Cin]
    const float* dpre;  // gradient wrt pre-activation output [N,OH,OW,Cout]
    float* part;        // [G][part_stride]: KH*KW*Cin*Cout floats of dW then Cout floats of dbias
    int part_stride;
    int N, H, W, OH, OW, Cin, Cout;
    int pad_t, pad_l;
    int TH, TW, NTX, RS;
    int units_total;
    float inv_rs;
    int stagger;
    int zero_slot;      // index of a pixel slot past the tile that the kernel keeps zeroed
    int stride;         // cursor kernel (wgrad_mfma_kernel) only: 1, or 2 (output pixel (r, c) reads the tile at (2r, 2c))
    unsigned long long* trace;  // diagnostic build (-DSRX_TRACE) only: per-wave cycle stamps
};

template <int CINP>
struct Lds {
    static constexpr int PS = (CINP == 4) ? 4 : CINP + 4;  // pixel stride in floats
};

// tanh, branch-free on the hardware's v_exp_f32 / v_rcp_f32 (round 4).  libdevice's tanhf is ~28 VALU instructions in two
// EXEC-masked branches; every VALU instruction of an epilogue adds to the MFMA time (section 3.0 of DESIGN.md), and ESPCN's two
// tanh layers spent 69 of 651 us at 720 x 1280 in it.  |x| < 0.55: x + x^3 P(x^2), P a degree-4 minimax fit of (tanh x - x) / x^3
// (relative error 1e-9 before rounding); otherwise 1 - 2 / (2^(2 |x| log2 e) + 1).  Measured against float64 tanh over 2.2 M
// points of [-12, 12] and a log sweep down to 1e-8: max relative error 1.9e-7, max absolute error 1e-7; exact limits (+-1 for
// large |x|, +-0 at +-0, NaN through).  Every fused multiply-add is spelled out, so that all kernels round identically.
__device__ __forceinline__ float srx_tanhf(float x) {
    const float ax = fabsf(x);
    const float x2 = ax * ax;
    float p = fmaf(x2, -0.006274174898862839f, 0.021071631461381912f);
    p = fmaf(x2, p, -0.053852297365665436f);
    p = fmaf(x2, p, 0.13332585990428925f);
    p = fmaf(x2, p, -0.33333316445350647f);
    const float small = fmaf(x2 * ax, p, ax);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);
    const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    return copysignf(ax < 0.55f ? small : big, x);
}

// The same function on two values at once: every multiply / add / fma below is ONE packed instruction for the pair
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32; gfx90a+), 23 VALU instructions per pair instead of 2 x 16 -- the VALU work of an
// epilogue adds to the fp32 MFMA time, so the count is what matters.  Same operations in the same order as
// srx_tanhf: the same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 srx_tanhf2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 x2 = ax * ax;
    f32x2 p = __builtin_elementwise_fma(x2, f32x2{-0.006274174898862839f, -0.006274174898862839f}, f32x2{0.021071631461381912f, 0.021071631461381912f});
    p = __builtin_elementwise_fma(x2, p, f32x2{-0.053852297365665436f, -0.053852297365665436f});
    p = __builtin_elementwise_fma(x2, p, f32x2{0.13332585990428925f, 0.13332585990428925f});
    p = __builtin_elementwise_fma(x2, p, f32x2{-0.33333316445350647f, -0.33333316445350647f});
    const f32x2 small = __builtin_elementwise_fma(x2 * ax, p, ax);
    const f32x2 t = ax * 2.8853900817779268f;
    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = e + 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    const f32x2 big = __builtin_elementwise_fma(r, f32x2{-2.0f, -2.0f}, f32x2{1.0f, 1.0f});
    f32x2 o;
    o[0] = copysignf(ax[0] < 0.55f ? small[0] : big[0], x[0]);
    o[1] = copysignf(ax[1] < 0.55f ? small[1] : big[1], x[1]);
    return o;
}
__device__ __forceinline__ f32x4 srx_tanhf4(f32x4 v) {
    const f32x2 a = srx_tanhf2(f32x2{v[0], v[1]}), b = srx_tanhf2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_TANH: return srx_tanhf(v);
        case ACT_LRELU: return v > 0.0f ? v : 0.2f * v;
        case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

__device__ __forceinline__ float act_grad_from_y(float y, int act) {
    switch (act) {
        case ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case ACT_TANH: return 1.0f - y * y;
        case ACT_LRELU: return y > 0.0f ? 1.0f : 0.2f;
        case ACT_SIGMOID: return y * (1.0f - y);
        default: return 1.0f;
    }
}

// exact for 0 <= s < 2^20, d < 2^12
__device__ __forceinline__ int fdiv_small(int s, float inv_d, int d) {
    int q = (int)(((float)s + 0.5f) * inv_d);
    // one correction step keeps it exact even where the float product rounds across an integer
    q -= (q * d > s);
    q += ((q + 1) * d <= s);
    return q;
}

#ifdef SRX_TRACE
#define SRX_STAMP() __builtin_amdgcn_s_memtime()
#else
#define SRX_STAMP() 0ull
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// make every tile boundary wait for the previous tile's output stores to retire.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Two workgroups share a CU.  Launched together they would stage their tiles at the same time and
// leave the MFMA pipe idle; delaying the one whose waves sit in odd hardware wave slots by about
// half a tile keeps one of them computing while the other stages.  Pure scheduling hint: results
// never depend on it (each wave decides for itself; the first barrier re-joins the workgroup).
__device__ __forceinline__ void stagger_second_workgroup(int sleeps) {
    // HW_REG_HW_ID (id 4), bits [3:0] = wave slot on the SIMD
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    if (slot & 1) {
        for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
}

// Stage one tile's input halo into LDS.  Slot s <-> (r = s / RS, c = s % RS) <-> input pixel
// (h_in0 + r, w_in0 + c); out-of-image slots are written as zeros (TF SAME zero padding).
// Written to cost few VALU instructions: branch-free, 32-bit offsets inside the image, loads
// always issued (clamped to the image base) and zeroed by a select, NB loads in flight.
template <int CINP>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ x, int n, int H, int W,
                                           int Cin, int h_in0, int w_in0, int RS, float inv_rs,
                                           int n_need, int tid) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TPP = CINP / 4;    // threads per pixel (each moves 16 B)
    constexpr int PPP = 256 / TPP;   // pixels per pass
    constexpr int NB = 8;            // 16-B loads in flight per thread
    const int c4 = tid % TPP;
    const int sp = tid / TPP;
    const float* xn = x + (size_t)n * H * W * Cin;      // wave-uniform base; offsets below fit 32 bits
    const bool ch_ok = 4 * c4 < Cin;
    if ((Cin & 3) == 0) {
        for (int s0 = sp; s0 < n_need; s0 += NB * PPP) {
            f32x4 v[NB];
            bool ok[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int s = s0 + j * PPP;
                const int r = fdiv_small(s, inv_rs, RS);
                const int c = s - r * RS;
                const int ih = h_in0 + r, iw = w_in0 + c;
                ok[j] = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W) & (s < n_need) & ch_ok;
                const unsigned off = ok[j] ? (unsigned)((ih * W + iw) * Cin + 4 * c4) : 0u;
                v[j] = *reinterpret_cast<const f32x4*>(xn + off);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int s = s0 + j * PPP;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (s < n_need) *reinterpret_cast<f32x4*>(lds + s * PS + 4 * c4) = ok[j] ? v[j] : z;
            }
        }
    } else {
        // channel counts that are not a multiple of 4 (RGB inputs, 27-channel sub-pixel tensors)
        for (int s = sp; s < n_need; s += PPP) {
            const int r = fdiv_small(s, inv_rs, RS);
            const int c = s - r * RS;
            const int ih = h_in0 + r, iw = w_in0 + c;
            const bool ok = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool oke = ok & (4 * c4 + e < Cin);
                const float t = xn[oke ? (unsigned)((ih * W + iw) * Cin + 4 * c4 + e) : 0u];
                v[e] = oke ? t : 0.f;
            }
            *reinterpret_cast<f32x4*>(lds + s * PS + 4 * c4) = v;
        }
    }
}

// Epilogue shared by all forward / dgrad variants: lane holds channels cb..cb+3 of pixel (orow, ocol).
// The skip / mask operand (the API never passes both) is fetched BEFORE the MFMA loop by
// conv_prefetch_aux so its latency hides behind the matrix work.  Offsets are 32-bit element
// offsets from the image's base (the host rejects images of 2^31 elements or more).
// AUX (template): the launch has a skip or mask operand.  It is a compile-time property because a
// conditional load in the epilogue makes hipcc's wait-count bookkeeping assume the load may be
// pending, and the counted s_waitcnt it inserts then waits for the group's own output STORES.
template <int G, bool AUX>
__device__ __forceinline__ void conv_prefetch_aux(f32x4 (&aux)[G], const unsigned (&off)[G], const ConvArgs& a,
                                                  size_t img_base, bool vec) {
    const float* src = a.mask ? a.mask : a.skip;
#pragma unroll
    for (int i = 0; i < G; ++i) aux[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (AUX && vec) {
        src += img_base;
#pragma unroll
        for (int i = 0; i < G; ++i) aux[i] = *reinterpret_cast<const f32x4*>(src + off[i]);
    }
}

// Branch-free forms for the piecewise-linear activations (none / relu / leaky-relu):
//   y = max(v, slope * v),  dy/dv seen through y: (y > 0) ? 1 : slope     with slope = 1, 0, 0.2
// tanh / sigmoid take one wave-uniform branch per accumulator.
__device__ __forceinline__ float act_slope(int act) {
    return act == ACT_RELU ? 0.0f : (act == ACT_LRELU ? 0.2f : 1.0f);
}

// tanh / sigmoid live OUT OF LINE: inlined, their code (x4 elements x every accumulator x every
// group body) pushed the 3x3x64 kernel to 60 KB, the size of the instruction cache two CUs share,
// and the VALU-dense staging / epilogue code then ran at instruction-fetch speed.
// (Round 4: inlined again.  With the 18-instruction branch-free srx_tanhf the code is a few KB per kernel, and the call was the
// expensive part: `act` arrived in a VGPR, so both activations ran under EXEC masks, and the caller shuffled its live
// registers around every call -- swapping libdevice's tanhf for srx_tanhf inside the out-of-line function changed nothing
// (ESPCN f1 at 720 x 1280: 133 -> 135 us); inlining it did.)
__device__ __forceinline__ f32x4 act_transcendental4(f32x4 v, int act) {
    if (act == ACT_TANH) return srx_tanhf4(v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + __expf(-v[e]));
    return v;
}

__device__ __forceinline__ f32x4 act_apply4(f32x4 v, int act, float slope) {
    if (act == ACT_TANH || act == ACT_SIGMOID) {
        v = act_transcendental4(v, act);
    } else if (act == ACT_RELU) {
        // relu as integer ops (x & ~(x >> 31)): fp32 VALU instructions share the datapath the fp32 MFMA
        // runs on and cost it issue time; integer ones do not.  Negative inputs (and -0) give +0.
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = v[e];                 // (bit_cast straight on a vector element miscompiles)
            const int b = __float_as_int(f);
            v[e] = __int_as_float(b & ~(b >> 31));
        }
    } else if (act == ACT_NONE) {
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
    }
    return v;
}

__device__ __forceinline__ f32x4 act_grad4(f32x4 v, f32x4 m, int act, float slope) {
    if (act == ACT_TANH || act == ACT_SIGMOID) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= (act == ACT_TANH) ? (1.0f - m[e] * m[e]) : m[e] * (1.0f - m[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : slope * v[e];
    }
    return v;
}

template <int G, bool AUX>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[G], const f32x4 (&aux)[G], const bool (&valid)[G],
                                              const unsigned (&off)[G], const ConvArgs& a, size_t img_base, int cb,
                                              bool vec) {
    float* yb = a.y + img_base;
    const float slope = act_slope(a.act), mslope = act_slope(a.mask_act);
    if (vec) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            f32x4 v = act_apply4(acc[i], a.act, slope);
            if (AUX) {
                if (a.skip) v += aux[i];
                if (a.post_relu) v = act_apply4(v, a.post_relu, act_slope(a.post_relu));   // ReLU or leaky ReLU after the add
                if (a.mask) v = act_grad4(v, aux[i], a.mask_act, mslope);
            }
#ifdef SRX_TRACE
            if ((a.dbg & 2) && v[0] != 12345.678f) continue;      // diagnostic builds: no stores
#endif
            if (valid[i]) *reinterpret_cast<f32x4*>(yb + off[i]) = v;
        }
    } else {
        // ragged channel counts (Cout = 3, 27, ...): scalar tail
        const float* sk = (AUX && a.skip) ? a.skip + img_base : nullptr;
        const float* mk = (AUX && a.mask) ? a.mask + img_base : nullptr;
        // Sub-pixel store mode (espcn/espcn/experiment_test.py:171-177 fused into the f3 layer): channel ch of an LR
        // pixel goes to HR row +ch/rc, element +ch%rc of the pixel's r*C-float segment.  The lane's four channels are
        // the same for the whole kernel, so these are four small constants (host: no skip / mask in this mode).
        int eo[4] = {0, 1, 2, 3};
        if (a.d2s_r) {
            const int hr_row = a.OW * a.d2s_rc;      // floats per HR row: OW*r pixels of C channels
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = cb + e, dy = ch / a.d2s_rc;
                eo[e] = dy * hr_row + (ch - dy * a.d2s_rc);
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (!valid[i]) continue;
            const f32x4 v = act_apply4(acc[i], a.act, slope);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (cb + e < a.Cout) {
                    float t = v[e];
                    if (sk) t += sk[off[i] + e];
                    if (AUX && a.post_relu) t = act_apply(t, a.post_relu);
                    if (mk) t *= act_grad_from_y(mk[off[i] + e], a.mask_act);
                    yb[off[i] + eo[e]] = t;
                }
            }
        }
    }
}

#define SRX_MFMA(ACC, A, B) "v_mfma_f32_16x16x4_f32 %" #ACC ", %" #A ", %" #B ", %" #ACC "\n\t"

// GUARD: the kernel is built with > 256 registers, so operands may reach the block through a
// v_accvgpr_read (a VALU write) that needs wait states before an MFMA reads it.
template <bool GUARD>
__device__ __forceinline__ void mfma_block(f32x4 (&c)[1], float w0, float w1, float w2, float w3, const f32x4 (&b)[1]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 1, 5) SRX_MFMA(0, 2, 6) SRX_MFMA(0, 3, 7) SRX_MFMA(0, 4, 8)
                     : "+v"(c[0])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]));
    else
        asm volatile(SRX_MFMA(0, 1, 5) SRX_MFMA(0, 2, 6) SRX_MFMA(0, 3, 7) SRX_MFMA(0, 4, 8)
                     : "+v"(c[0])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]));
}
template <bool GUARD>
__device__ __forceinline__ void mfma_block(f32x4 (&c)[2], float w0, float w1, float w2, float w3, const f32x4 (&b)[2]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 2, 6) SRX_MFMA(1, 2, 10) SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13)
                     : "+v"(c[0]), "+v"(c[1])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]));
    else
        asm volatile(SRX_MFMA(0, 2, 6) SRX_MFMA(1, 2, 10) SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13)
                     : "+v"(c[0]), "+v"(c[1])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]));
}
template <bool GUARD>
__device__ __forceinline__ void mfma_block(f32x4 (&c)[3], float w0, float w1, float w2, float w3, const f32x4 (&b)[3]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(2, 3, 15) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3]));
    else
        asm volatile(SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(2, 3, 15) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3]));
}
template <bool GUARD>
__device__ __forceinline__ void mfma_block(f32x4 (&c)[4], float w0, float w1, float w2, float w3, const f32x4 (&b)[4]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(3, 4, 20) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(3, 5, 21) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18) SRX_MFMA(3, 6, 22) SRX_MFMA(0, 7, 11) SRX_MFMA(1, 7, 15) SRX_MFMA(2, 7, 19) SRX_MFMA(3, 7, 23)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3]), "v"(b[3][0]), "v"(b[3][1]), "v"(b[3][2]), "v"(b[3][3]));
    else
        asm volatile(SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(3, 4, 20) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(3, 5, 21) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18) SRX_MFMA(3, 6, 22) SRX_MFMA(0, 7, 11) SRX_MFMA(1, 7, 15) SRX_MFMA(2, 7, 19) SRX_MFMA(3, 7, 23)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                     : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3]), "v"(b[3][0]), "v"(b[3][1]), "v"(b[3][2]), "v"(b[3][3]));
}

// Same blocks with the stationary weights in ACCUMULATION registers ("a"): the one-wave-per-SIMD
// kernel owns the whole 512-entry register file, keeps its 144+ weights in AGPRs (MFMA reads A/B
// operands from either file) and leaves the 256 VGPRs to everything else.  "memory" clobber: the
// staging loads / LDS writes placed between blocks must stay there.
template <bool GUARD>
__device__ __forceinline__ void mfma_block_a(f32x4 (&c)[1], float w0, float w1, float w2, float w3, const f32x4 (&b)[1]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 1, 5) SRX_MFMA(0, 2, 6) SRX_MFMA(0, 3, 7) SRX_MFMA(0, 4, 8)
                     : "+v"(c[0])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3])
                     : "memory");
    else
        asm volatile(SRX_MFMA(0, 1, 5) SRX_MFMA(0, 2, 6) SRX_MFMA(0, 3, 7) SRX_MFMA(0, 4, 8)
                     : "+v"(c[0])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3])
                     : "memory");
}
template <bool GUARD>
__device__ __forceinline__ void mfma_block_a(f32x4 (&c)[2], float w0, float w1, float w2, float w3, const f32x4 (&b)[2]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 2, 6) SRX_MFMA(1, 2, 10) SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13)
                     : "+v"(c[0]), "+v"(c[1])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3])
                     : "memory");
    else
        asm volatile(SRX_MFMA(0, 2, 6) SRX_MFMA(1, 2, 10) SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13)
                     : "+v"(c[0]), "+v"(c[1])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3])
                     : "memory");
}
template <bool GUARD>
__device__ __forceinline__ void mfma_block_a(f32x4 (&c)[3], float w0, float w1, float w2, float w3, const f32x4 (&b)[3]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(2, 3, 15) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3])
                     : "memory");
    else
        asm volatile(SRX_MFMA(0, 3, 7) SRX_MFMA(1, 3, 11) SRX_MFMA(2, 3, 15) SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3])
                     : "memory");
}
template <bool GUARD>
__device__ __forceinline__ void mfma_block_a(f32x4 (&c)[4], float w0, float w1, float w2, float w3, const f32x4 (&b)[4]) {
    if constexpr (GUARD)
        asm volatile("s_nop 1\n\t" SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(3, 4, 20) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(3, 5, 21) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18) SRX_MFMA(3, 6, 22) SRX_MFMA(0, 7, 11) SRX_MFMA(1, 7, 15) SRX_MFMA(2, 7, 19) SRX_MFMA(3, 7, 23)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3]), "v"(b[3][0]), "v"(b[3][1]), "v"(b[3][2]), "v"(b[3][3])
                     : "memory");
    else
        asm volatile(SRX_MFMA(0, 4, 8) SRX_MFMA(1, 4, 12) SRX_MFMA(2, 4, 16) SRX_MFMA(3, 4, 20) SRX_MFMA(0, 5, 9) SRX_MFMA(1, 5, 13) SRX_MFMA(2, 5, 17) SRX_MFMA(3, 5, 21) SRX_MFMA(0, 6, 10) SRX_MFMA(1, 6, 14) SRX_MFMA(2, 6, 18) SRX_MFMA(3, 6, 22) SRX_MFMA(0, 7, 11) SRX_MFMA(1, 7, 15) SRX_MFMA(2, 7, 19) SRX_MFMA(3, 7, 23)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                     : "a"(w0), "a"(w1), "a"(w2), "a"(w3), "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3]), "v"(b[3][0]), "v"(b[3][1]), "v"(b[3][2]), "v"(b[3][3])
                     : "memory");
}

// Stationary-weight fetch for one wave: wr[tap*KSPT + j] = W_eff[k = channel ci(j,kq)][cout0 + li].
// Exact-fit layers (Cin == CINP, Cout a multiple of 16: every VDSR / EnhanceNet body layer) take a fast
// path: the lane's part of the index is one VGPR, the (tap, j) part is wave-uniform and goes into the
// buffer load's SGPR offset -- one instruction per weight (one b128 per 4 weights for the transposed
// dgrad read) instead of ~10 (bounds compares, selects, 64-bit address arithmetic).
template <int TAPS, int CINP, bool WT>
__device__ __forceinline__ void load_stationary_weights(float (&wr)[TAPS * (CINP / 4)], const ConvArgs& a, int cout0,
                                                        int li, int kq) {
    constexpr int KSPT = CINP / 4;
    const int co = cout0 + li;
    if (CINP >= 16 && a.Cin == CINP && (a.Cout & 15) == 0 && (long)TAPS * a.Cin * a.Cout * 4 < (1L << 31)) {
        __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, TAPS * a.Cin * a.Cout * 4, 0x00020000);
        if (!WT) {
            const int vlane = (4 * kq * a.Cout + co) * 4;              // bytes
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int j = 0; j < KSPT; ++j) {
                    const int soff = ((tap * a.Cin + 16 * (j / 4) + (j % 4)) * a.Cout) * 4;   // wave-uniform
                    wr[tap * KSPT + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vlane, soff, 0));
                }
        } else {
            const int vlane = (co * a.Cin + 4 * kq) * 4;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int g = 0; g < KSPT / 4; ++g) {
                    const int soff = (((TAPS - 1 - tap) * a.Cout) * a.Cin + 16 * g) * 4;
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vlane, soff, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) wr[tap * KSPT + 4 * g + e] = v[e];
                }
        }
        return;
    }
    // general path: branch-free, out-of-range (padded) channels read element 0 and are zeroed by a select
    const bool co_ok = co < a.Cout;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
        for (int j = 0; j < KSPT; ++j) {
            const int ci = (CINP >= 16) ? 16 * (j / 4) + 4 * kq + (j % 4) : kq;
            const bool ok = co_ok & (ci < a.Cin);
            const unsigned idx = !WT ? (unsigned)((tap * a.Cin + ci) * a.Cout + co)
                                     : (unsigned)(((TAPS - 1 - tap) * a.Cout + co) * a.Cin + ci);
            const float v = a.w[ok ? idx : 0u];
            wr[tap * KSPT + j] = ok ? v : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CINP, int G, bool GUARD, bool AUX>
__device__ __forceinline__ void conv_group(const float* lds, const float (&wr)[KH * KW * (CINP / 4)],
                                           const f32x4 bias4, const ConvArgs& a, int n, int h, int ow0,
                                           int th, int tw, float inv_tw, int m_first, int m_step, int n_active,
                                           int li, int kq, int cout0, unsigned long long& t_mfma, unsigned long long& t_pro,
                                           unsigned long long& t_epi) {
    // n_active <= G: sub-tiles beyond it are dummies (computed on pixel 0, never stored), so that only
    // two group bodies per kernel instance have to exist
    constexpr int PS = Lds<CINP>::PS;
    constexpr int NG = (CINP >= 16) ? CINP / 16 : 1;
    const unsigned long long ts_pro = SRX_STAMP();
    const int npx = th * tw;
    const int cb = cout0 + 4 * kq;
    const bool vec = ((a.Cout & 3) == 0) && (cb + 3 < a.Cout) && !a.d2s_r;
    const size_t img_base = (size_t)n * a.OH * a.OW * a.Cout;   // wave-uniform
    int laddr[G];       // float index of this lane's pixel (tap 0,0) in LDS
    unsigned off[G];    // element offset of this lane's 4 output channels inside image n
    bool valid[G];
    f32x4 acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m_first + i * m_step) + li;
        const bool live = (t < npx) & (i < n_active);
        valid[i] = live & (cb < a.Cout);
        const int tt = live ? t : 0;
        const int orow = fdiv_small(tt, inv_tw, tw);
        const int ocol = tt - orow * tw;
        laddr[i] = ((orow * a.RS + ocol) * PS << (a.stride - 1)) + ((CINP >= 16) ? 4 * kq : kq);   // (stride 2: tile position (2 orow, 2 ocol))
        // (sub-pixel store mode: the offset of HR pixel (h*r, w*r), channel 0 -- conv_epilogue adds the channel part)
        off[i] = !valid[i] ? 0u
                 : (a.d2s_r ? (unsigned)(((h + orow) * a.d2s_r * a.OW + ow0 + ocol) * a.d2s_rc)
                            : (unsigned)(((h + orow) * a.OW + ow0 + ocol) * a.Cout + cb));
        acc[i] = bias4;
    }
    f32x4 aux[G];
    conv_prefetch_aux<G, AUX>(aux, off, a, img_base, vec);
    const int row_stride = a.RS * PS;
    if constexpr (CINP >= 16) {
        // Software pipeline over the KH*KW*NG k-groups: the LDS fragments of group t+1 are requested
        // before the MFMAs of group t are issued.
        constexpr int NBLK = KH * KW * NG;
        // Everything outside the MFMA stream (staging, address set-up, epilogue) runs at raised
        // priority: next to a partner wave that issues MFMAs back to back, priority-0 VALU / memory
        // instructions get roughly one issue slot per MFMA period.
        __builtin_amdgcn_s_setprio(0);
        const unsigned long long ts0 = SRX_STAMP();
        t_pro += ts0 - ts_pro;
        f32x4 cur[G], nxt[G];
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i]);
#pragma unroll
        for (int t = 0; t < NBLK; ++t) {
            if (t + 1 < NBLK) {
                const int t1 = t + 1;
                const int kh1 = (t1 / NG) / KW, kw1 = (t1 / NG) % KW, g1 = t1 % NG;
#pragma unroll
                for (int i = 0; i < G; ++i)
                    nxt[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i] + kh1 * row_stride + kw1 * PS + 16 * g1);
            }
            const int tap = t / NG, g = t % NG;
            const int wb = tap * (CINP / 4) + 4 * g;
            // the accumulators were initialised by VALU moves that the compiler may sink right in front
            // of the first block: that block carries its own leading s_nop (as every GUARD block does)
            if (t == 0)
                mfma_block<true>(acc, wr[wb], wr[wb + 1], wr[wb + 2], wr[wb + 3], cur);
            else
                mfma_block<GUARD>(acc, wr[wb], wr[wb + 1], wr[wb + 2], wr[wb + 3], cur);
#pragma unroll
            for (int i = 0; i < G; ++i) cur[i] = nxt[i];
        }
        // MFMA results are read by VALU code next: software must cover the result latency
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_s_setprio(2);
        t_mfma += SRX_STAMP() - ts0;
    } else {
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int tap = kh * KW + kw;
                float av[G];
#pragma unroll
                for (int i = 0; i < G; ++i) av[i] = lds[laddr[i] + kh * row_stride + kw * PS];
#pragma unroll
                for (int i = 0; i < G; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[tap], av[i], acc[i], 0, 0, 0);
            }
        }
    }
    const unsigned long long ts_epi = SRX_STAMP();
    conv_epilogue<G, AUX>(acc, aux, valid, off, a, img_base, cb, vec);
    t_epi += SRX_STAMP() - ts_epi;
}

// KH,KW: filter; CINP: padded input channels held per LDS pixel (4, 32 or 64); NCH: number of
// 16-wide output-channel chunks (1, 2 or 4 -> 4/NCH waves share the pixels of a chunk);
// WT: read the filters transposed + flipped (dgrad); MINW: waves per SIMD for launch bounds.
template <int KH, int KW, int CINP, int NCH, bool WT, int MINW, bool AUX>
__global__ __launch_bounds__(256, MINW) void conv_mfma_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TAPS = KH * KW;
    constexpr int KSPT = CINP / 4;
    constexpr int NPART = 4 / NCH;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, part = wave / NCH;
    const int cout0 = chunk * 16;
#ifdef SRX_TRACE
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_setprio(2);

    // ---- stationary weights: wr[tap][j], k index of lane = channel ci(j, kq)
    float wr[TAPS * KSPT];
    load_stationary_weights<TAPS, CINP, WT>(wr, a, cout0, li, kq);
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cout0 + 4 * kq + e < a.Cout) bias4[e] = a.bias[cout0 + 4 * kq + e];
    }

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    if (a.stagger) stagger_second_workgroup(a.stagger);
    [[maybe_unused]] unsigned long long t_mfma = 0, t_stage = 0, t_bar1 = 0, t_load = 0, t_pro = 0, t_epi = 0;   // (trace builds)
    [[maybe_unused]] const unsigned long long t_begin = SRX_STAMP();
    // Work distribution.  Static: a contiguous range of output rows per workgroup.  Dynamic (tile_counter
    // set): fixed tiles of TH rows handed out through one atomic counter -- the wave that an fp32-MFMA
    // partner starves falls behind, and a static split then leaves one workgroup per CU idle at the end.
    // Which workgroup computes a tile never changes a result.
    const bool dyn = a.tile_counter != nullptr;
    int u = u0;
    int tile = blockIdx.x;             // dynamic mode: the first tile is the workgroup's own index
    int* mailbox = reinterpret_cast<int*>(lds + a.lds_sched_slot);
    while (dyn ? (tile < a.tiles_total) : (u < u1)) {
        int h, n, tx, th;
        if (dyn) {
            const int ti = tile % a.tiles_per_col;
            const int t = tile / a.tiles_per_col;
            tx = t % a.NTX;
            n = t / a.NTX;
            h = ti * a.TH;
            th = (a.OH - h < a.TH) ? (a.OH - h) : a.TH;
        } else {
            h = u % a.OH;
            const int t = u / a.OH;
            tx = t % a.NTX;
            n = t / a.NTX;
            th = a.TH;
            if (a.OH - h < th) th = a.OH - h;
            if (u1 - u < th) th = u1 - u;
        }
        const int ow0 = tx * a.TW;
        const int tw = (a.OW - ow0 < a.TW) ? (a.OW - ow0) : a.TW;
        const int n_need = ((th - 1) * a.stride + KH) * a.RS + (KW - 1);
        [[maybe_unused]] const bool first_tile = dyn ? (tile == (int)blockIdx.x) : (u == u0);

        const unsigned long long ts_stage = SRX_STAMP();
        lds_barrier();
        const unsigned long long ts_b1 = SRX_STAMP();
        if (dyn && tid == 0) *mailbox = atomicAdd(a.tile_counter, 1);     // the NEXT tile, fetched early
#ifdef SRX_TRACE
        if (!(a.dbg & 1) || first_tile)                       // diagnostic builds: stage only the first tile
#endif
            stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h * a.stride - a.pad_t, ow0 * a.stride - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        const unsigned long long ts_ld = SRX_STAMP();
        lds_barrier();
        const int next_tile = dyn ? *mailbox : 0;
        t_stage += SRX_STAMP() - ts_stage;
        t_bar1 += ts_b1 - ts_stage;
        t_load += ts_ld - ts_b1;
        const int n_sub = (th * tw + 15) >> 4;
        const int cnt = (n_sub - part + NPART - 1) / NPART;  // subtiles of this wave
        const float inv_tw = 1.0f / (float)tw;
        if (cnt > 0) {
            // instances with an aux operand also hold its prefetched registers: 3 accumulators per group.
            // Two bodies only (MAXG and MAXG-1; smaller groups run the small body with dummy sub-tiles):
            // code size matters, see act_transcendental4.
            constexpr int MAXG = AUX ? 3 : 4;
            const int ng = (cnt + MAXG - 1) / MAXG;
            const int base = cnt / ng, rem = cnt % ng;
            int idx = 0;
            for (int gi = 0; gi < ng; ++gi) {
                const int gs = base + (gi < rem ? 1 : 0);
                const int m_first = part + idx * NPART;
                if (gs == MAXG)
                    conv_group<KH, KW, CINP, MAXG, (MINW < 2), AUX>(lds, wr, bias4, a, n, h, ow0, th, tw, inv_tw, m_first, NPART, gs, li, kq, cout0, t_mfma, t_pro, t_epi);
                else
                    conv_group<KH, KW, CINP, MAXG - 1, (MINW < 2), AUX>(lds, wr, bias4, a, n, h, ow0, th, tw, inv_tw, m_first, NPART, gs, li, kq, cout0, t_mfma, t_pro, t_epi);
                idx += gs;
            }
        }
        u += th;
        tile = next_tile;
    }
#ifdef SRX_TRACE
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 12;
        tr[8] = t_bar1; tr[9] = t_load; tr[10] = t_pro; tr[11] = t_epi;
        tr[0] = t_begin; tr[1] = SRX_STAMP(); tr[2] = t_mfma; tr[3] = t_stage;
        tr[4] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID[15:0]
        tr[5] = rt_entry; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = t_entry;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// One-wave-per-SIMD pipelined forward / dgrad (srx_set_conv_path(1) / SRX_PIPE=1).
//
// What gfx950 charges a wave for instructions placed between its own fp32 MFMAs (measured,
// scripts/shadow2_ubench.hip, shadow3_ubench.hip; an MFMA alone issues every 32 cycles):
//     any VALU instruction (v_add, v_cndmask, v_mov, even v_nop) ...... +13 cycles for the first of a run,
//                                                                        +8 for each further one
//     SALU instruction ................................................ ~0 (up to ~4 per MFMA)
//     ds_read_b128 / buffer_load / buffer_store / ds_write / s_waitcnt . ~0..3 (bandwidth permitting)
// and a wave streaming fp32 MFMAs starves the other wave of its SIMD completely (coissue_ubench).  So
// fp32 MFMA time and VALU time simply add up, on one wave or two, and the only lever is the NUMBER of
// VALU instructions.  This kernel therefore keeps one 4-wave workgroup per CU (weights in AGPRs, the
// input tile double-buffered in LDS, 2 x 80 KiB) and does everything that is not an MFMA with scalar
// or memory instructions:
//   * staging of the next tile: the tile is cut into passes of PPP slots inside ONE tile row, so the
//     lane part of every address is a constant (tid*16 bytes in the image row, a constant in LDS) and
//     the pass part is wave-uniform -> SGPR offset of a bounds-checked buffer load (rows outside the image:
//     an out-of-range scalar offset, the hardware returns zeros), lanes outside the row: an EXEC mask
//     set by s_mov.  0 VALU to issue a pass, 1 (LDS address) to commit it;
//   * sub-tile LDS addresses: a table (sub-tile, lane) -> byte address kept in the 16 pad bytes of the
//     LDS pixel slots, fetched with ds_read_b32 during the previous group;
//   * output addresses: lane part constant, sub-tile part in the buffer resource's base / num_records
//     (which also drops the pixels past the end of a short last sub-tile);
//   * bias: the C operand of each accumulator's first MFMA;
//   * the PREVIOUS group's epilogue runs inside the current group from a parked copy of its
//     accumulators, so no s_nop covers the MFMA result latency.
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));

constexpr int kOobOffset = 0x7fffffff;    // lane offset beyond every buffer resource used here (host-checked < 2^31)

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// a buffer resource that provably lives in SGPRs (it is an "s" operand of asm statements)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float* p, int num_records) {
    const unsigned long long q = uniform64((unsigned long long)reinterpret_cast<uintptr_t>(p));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(static_cast<uintptr_t>(q)), 0,
                                             __builtin_amdgcn_readfirstlane(num_records), 0x00020000);
}
struct StageGeo {      // wave-uniform constants of the staging sequence
    int JP1;           // passes per tile row, minus 1
    int rowfix_g, rowfix_l;  // extra bytes (beyond the constant pass step) when the next pass starts a new row: image, LDS
    unsigned long long m_first, m_last;   // EXEC masks of a row's first / last pass (pad columns / row end)
    unsigned long long m_row;             // column-strip tiles only: lanes of a row's last pass that lie inside the tile row
    unsigned long long m_mid;             // strips whose invalid columns may start before the last pass (wgrad's narrow
                                          // last strip): valid lanes of the passes between the first and the last
};
// The scalar offset is UNSIGNED to the hardware (measured: a negative one puts the whole pass out of range),
// so the buffer resource starts pad_l pixels before the image (those bytes are never touched: the lanes
// that would are masked) and offsets count from there.  Rows below the image need no test: their offsets
// are beyond num_records and the range check returns the zero padding by itself.  For rows above it the
// scalar offset is replaced by a large one: they are the tile's first passes, those with more than `thr`
// passes left.
//
// The cursor lives in SGPRs and advances on the SALU.  It is written as inline asm with "s" operands
// because hipcc otherwise moves such wave-uniform add/select chains to the VALU under SGPR pressure
// (v_add + v_readfirstlane per step -- each a VALU slot in the MFMA stream).
struct StageSeq {      // wave-uniform cursor: the next pass
    int j;             // pass index inside the tile row
    int off;           // issue cursor: byte offset in the image of (tile row, column PPP*j - pad_l), may be negative;
                       // commit cursor: LDS byte address of slot (tile row, PPP*j) in the destination buffer
    int left;          // passes left in this tile
    int thr;           // issue cursor: the passes done while left > thr lie in rows above the image
};

// One pass = four small statements, so that the group code can deal them out over different gaps of the
// MFMA stream (about 4 SALU instructions per gap are free, 20 in one gap are not).
// (1) EXEC mask of the cursor's pass, first half: row-start / row-end masks
__device__ __forceinline__ void stage_mask_a(const StageSeq& q, const StageGeo& G, unsigned long long& m, unsigned long long& t) {
    asm volatile("s_cmp_eq_u32 %2, 0\n\t"
                 "s_cselect_b64 %0, %4, -1\n\t"
                 "s_cmp_eq_u32 %2, %3\n\t"
                 "s_cselect_b64 %1, %5, -1"
                 : "=&s"(m), "=&s"(t) : "s"(q.j), "s"(G.JP1), "s"(G.m_first), "s"(G.m_last) : "scc");
}
// (2) second half: combine, nothing once the tile is complete; and the scalar offset (rows above the image)
__device__ __forceinline__ void stage_mask_b(const StageSeq& q, unsigned long long& m, unsigned long long t, int& so) {
    asm volatile("s_and_b64 %0, %0, %2\n\t"
                 "s_cmp_gt_i32 %3, 0\n\t"
                 "s_cselect_b64 %0, %0, 0\n\t"
                 "s_cmp_gt_i32 %3, %4\n\t"
                 "s_cselect_b32 %1, 0x7ff00000, %5"
                 : "+s"(m), "=&s"(so) : "s"(t), "s"(q.left), "s"(q.thr), "s"(q.off) : "scc");
}
// EXEC mask of a cursor's pass in one piece (commit side of kernels that do not keep the issue-time mask)
__device__ __forceinline__ unsigned long long stage_mask_full(const StageSeq& q, const StageGeo& G) {
    unsigned long long m, t;
    asm volatile("s_cmp_eq_u32 %2, 0\n\t"
                 "s_cselect_b64 %0, %4, -1\n\t"
                 "s_cmp_eq_u32 %2, %3\n\t"
                 "s_cselect_b64 %1, %5, -1\n\t"
                 "s_and_b64 %0, %0, %1\n\t"
                 "s_cmp_gt_i32 %6, 0\n\t"
                 "s_cselect_b64 %0, %0, 0"
                 : "=&s"(m), "=&s"(t) : "s"(q.j), "s"(G.JP1), "s"(G.m_first), "s"(G.m_last), "s"(q.left) : "scc");
    return m;
}
// (3) the load: 16 bytes per active lane, no VALU
__device__ __forceinline__ f32x4 stage_fire(unsigned long long m, int so, __amdgpu_buffer_rsrc_t rsrc, int voff_lane) {
    f32x4 v;
    asm volatile("s_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_mov_b64 exec, -1"
                 : "=&v"(v) : "v"(voff_lane), "s"(rsrc), "s"(so), "s"(m) : "memory");
    return v;
}
// ---- column-strip tiles ("Z" variants).  A strip carries real halo columns on both sides, except where it touches
// the image edge: there the halo slots must read as zero, and since a buffer holds strips of different columns
// in turn they cannot be zeroed once.  So a pass issues TWO loads into the same register: the valid lanes from the
// image, the edge-halo lanes with an out-of-range scalar offset (they return 0); one LDS write covers both.
// m_first / m_last are then the VALID lanes of a row's first / last pass for the strip being staged (set per
// tile), m_row the lanes inside the tile row.
template <bool MID = false>
__device__ __forceinline__ void stage_mask_az(const StageSeq& q, const StageGeo& G, unsigned long long& m, unsigned long long& t,
                                              unsigned long long& r) {
    if constexpr (MID) {
        // three classes of passes: first (m_first, already combined with m_last when a row is one pass), last, between
        asm volatile("s_cmp_eq_u32 %3, %4\n\t"
                     "s_cselect_b64 %1, %6, %8\n\t"
                     "s_cselect_b64 %2, %7, -1\n\t"
                     "s_cmp_eq_u32 %3, 0\n\t"
                     "s_cselect_b64 %0, %5, %1\n\t"
                     "s_mov_b64 %1, -1"
                     : "=&s"(m), "=&s"(t), "=&s"(r)
                     : "s"(q.j), "s"(G.JP1), "s"(G.m_first), "s"(G.m_last), "s"(G.m_row), "s"(G.m_mid) : "scc");
        return;
    }
    asm volatile("s_cmp_eq_u32 %3, 0\n\t"
                 "s_cselect_b64 %0, %5, -1\n\t"
                 "s_cmp_eq_u32 %3, %4\n\t"
                 "s_cselect_b64 %1, %6, -1\n\t"
                 "s_cselect_b64 %2, %7, -1"
                 : "=&s"(m), "=&s"(t), "=&s"(r) : "s"(q.j), "s"(G.JP1), "s"(G.m_first), "s"(G.m_last), "s"(G.m_row) : "scc");
}
// after this: r = lanes the pass writes to LDS (none once the tile is complete), m = those of them loaded from the image
__device__ __forceinline__ void stage_mask_bz(const StageSeq& q, unsigned long long& m, unsigned long long t, unsigned long long& r,
                                              int& so) {
    asm volatile("s_and_b64 %0, %0, %3\n\t"
                 "s_cmp_gt_i32 %4, 0\n\t"
                 "s_cselect_b64 %1, %1, 0\n\t"
                 "s_and_b64 %0, %0, %1\n\t"
                 "s_cmp_gt_i32 %4, %5\n\t"
                 "s_cselect_b32 %2, 0x7ff00000, %6"
                 : "+s"(m), "+s"(r), "=&s"(so) : "s"(t), "s"(q.left), "s"(q.thr), "s"(q.off) : "scc");
}
__device__ __forceinline__ f32x4 stage_fire_z(unsigned long long m, unsigned long long r, int so, __amdgpu_buffer_rsrc_t rsrc,
                                              int voff_lane) {
    f32x4 v;
    unsigned long long z;
    const int oob = 0x7ff00000;
    asm volatile("s_andn2_b64 %1, %6, %5\n\t"
                 "s_mov_b64 exec, %5\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\t"
                 "s_mov_b64 exec, %1\n\tbuffer_load_dwordx4 %0, %2, %3, %7 offen\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&v"(v), "=&s"(z) : "v"(voff_lane), "s"(rsrc), "s"(so), "s"(m), "s"(r), "s"(oob) : "memory", "scc");
    return v;
}
// the lanes a cursor's pass writes to LDS, from the commit cursor alone
__device__ __forceinline__ unsigned long long stage_mask_row(const StageSeq& q, const StageGeo& G) {
    unsigned long long r;
    asm volatile("s_cmp_eq_u32 %1, %2\n\ts_cselect_b64 %0, %3, -1\n\ts_cmp_gt_i32 %4, 0\n\ts_cselect_b64 %0, %0, 0"
                 : "=&s"(r) : "s"(q.j), "s"(G.JP1), "s"(G.m_row), "s"(q.left) : "scc");
    return r;
}
// (4) advance a cursor by one pass
template <int STEP>
__device__ __forceinline__ void stage_next(StageSeq& q, int JP1, int rowfix) {
    int tmp;
    asm volatile("s_cmp_eq_u32 %0, %4\n\t"          // last pass of the row?
                 "s_cselect_b32 %3, %5, 0\n\t"
                 "s_cselect_b32 %0, -1, %0\n\t"
                 "s_add_i32 %1, %1, %3\n\t"
                 "s_add_i32 %1, %1, %6\n\t"
                 "s_add_i32 %0, %0, 1\n\t"
                 "s_sub_i32 %2, %2, 1"
                 : "+s"(q.j), "+s"(q.off), "+s"(q.left), "=&s"(tmp) : "s"(JP1), "s"(rowfix), "n"(STEP) : "scc");
}
// the LDS write of a pass (mask from its issue): wait until at most `pend` younger VMEM operations are
// outstanding, one VALU (LDS address)
#define SRX_COMMIT_ASM(N)                                                                                    \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_add_u32 %0, %3, %4\n\ts_mov_b64 exec, %2\n\tds_write_b128 %0, %1\n\ts_mov_b64 exec, -1" \
                 : "=&v"(addr) : "v"(v), "s"(m), "s"(q.off), "v"(wl_lane) : "memory")
__device__ __forceinline__ void stage_commit(int pend, const StageSeq& q, unsigned long long m, int wl_lane, const f32x4 v) {
    int addr;
    // (pend is a constant after unrolling: the chain folds to one statement)
    if (pend >= 14) SRX_COMMIT_ASM(14);
    else if (pend >= 12) SRX_COMMIT_ASM(12);
    else if (pend >= 10) SRX_COMMIT_ASM(10);
    else if (pend >= 8) SRX_COMMIT_ASM(8);
    else if (pend >= 6) SRX_COMMIT_ASM(6);
    else if (pend == 5) SRX_COMMIT_ASM(5);
    else if (pend == 4) SRX_COMMIT_ASM(4);
    else if (pend == 3) SRX_COMMIT_ASM(3);
    else if (pend == 2) SRX_COMMIT_ASM(2);
    else if (pend == 1) SRX_COMMIT_ASM(1);
    else SRX_COMMIT_ASM(0);
}
// a whole pass at once (prologue / drain loops)
template <int CINP>
__device__ __forceinline__ void stage_pass_now(StageSeq& qi, StageSeq& qc, const StageGeo& G, __amdgpu_buffer_rsrc_t rsrc,
                                               int voff_lane, int wl_lane) {
    constexpr int PPP = 256 / (CINP / 4);
    unsigned long long m, t;
    int so;
    stage_mask_a(qi, G, m, t);
    stage_mask_b(qi, m, t, so);
    const f32x4 v = stage_fire(m, so, rsrc, voff_lane);
    stage_next<PPP * CINP * 4>(qi, G.JP1, G.rowfix_g);
    stage_commit(0, qc, m, wl_lane, v);
    stage_next<PPP * Lds<CINP>::PS * 4>(qc, G.JP1, G.rowfix_l);
}

// a whole tile with NB loads in flight (kernels that stage between their MFMA phases)
template <int CINP, int NB, bool Z = false, bool MID = false>
__device__ __forceinline__ void stage_tile_scalar(StageSeq& qi, StageSeq& qc, const StageGeo& G, __amdgpu_buffer_rsrc_t rsrc,
                                                  int voff_lane, int wl_lane) {
    constexpr int PPP = 256 / (CINP / 4);
    while (qi.left > 0) {
        f32x4 v[NB];
        unsigned long long mk[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            unsigned long long t;
            int so;
            if constexpr (Z) {
                unsigned long long m;
                stage_mask_az<MID>(qi, G, m, t, mk[i]);
                stage_mask_bz(qi, m, t, mk[i], so);
                v[i] = stage_fire_z(m, mk[i], so, rsrc, voff_lane);
            } else {
                stage_mask_a(qi, G, mk[i], t);
                stage_mask_b(qi, mk[i], t, so);
                v[i] = stage_fire(mk[i], so, rsrc, voff_lane);
            }
            stage_next<PPP * CINP * 4>(qi, G.JP1, G.rowfix_g);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            stage_commit((Z ? 2 : 1) * (NB - 1 - i), qc, mk[i], wl_lane, v[i]);
            stage_next<PPP * Lds<CINP>::PS * 4>(qc, G.JP1, G.rowfix_l);
        }
    }
}

// zero-cost ordering fence: code that uses x afterwards cannot be scheduled above this point, code that
// produced x cannot sink below it (volatile asm statements keep their relative order)
#define SRX_PIN(x) asm volatile("" : "+v"(x))

template <int MAXG, int AUX>
struct PipePend {      // a finished group waiting for its epilogue
    f32x4 acc[MAXG];
    f32x4 aux[(AUX == 1 || AUX == 2) ? MAXG : 1];
    __amdgpu_buffer_rsrc_t yrs;   // wave-uniform: the unit's output pixels
    int m_first, gs;
};
struct PipeEpi {
    int lo, lo2;       // wave-uniform clamps
    int cout4;         // Cout * 4   (AUX 4: the bytes from one LR pixel to the next inside an HR row, 3 r * 4)
    int sh, rowb;      // column strips: log2(sub-tiles per strip row), bytes of one output row (AUX 4: of r HR rows)
    int vo1, vo2, vo3; // AUX 4 only, PER LANE: byte offsets of the lane's output elements 1..3 (element 0: vst)
};
// AUX (template): 0 no aux operand, 1 ReluGrad mask (dgrad), 2 residual add, 3 no aux operand + tanh (ESPCN's f2 on whole images),
// 4 no aux operand, no activation, stored through the sub-pixel map (ESPCN's f3 on whole images; column strips only)
template <int AUX>
__device__ __forceinline__ PipeEpi pipe_epi_setup(const ConvArgs& a) {
    PipeEpi e;
    e.lo = (a.act == ACT_RELU) ? 0 : (int)0x80000000;
    e.lo2 = (AUX == 2 && a.post_relu) ? 0 : (int)0x80000000;
    e.cout4 = (AUX == 4) ? 3 * a.d2s_r * 4 : a.Cout * 4;
    e.sh = (a.TW >= 32) ? 1 : 0;
    e.rowb = (AUX == 4) ? a.d2s_r * (a.OW * a.d2s_r * 3) * 4 : a.OW * a.Cout * 4;
    e.vo1 = e.vo2 = e.vo3 = 0;
    return e;
}
__device__ __forceinline__ f32x4 clamp_lo4(f32x4 v, int lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float f = v[e];                 // (bit_cast straight on a vector element miscompiles)
        const int b = __float_as_int(f);
        v[e] = __int_as_float(b > lo ? b : lo);
    }
    return v;
}
// Output / aux addressing of sub-tile m of a unit: ONE buffer resource per unit (base = the unit's first pixel,
// num_records = the bytes of its pixels), the sub-tile as the scalar offset (it takes part in the range check, so
// the pixels past the end of a short last sub-tile are dropped), a constant lane offset.  Dummy sub-tiles get an
// out-of-range scalar offset.
// (Column strips, Z: the strip is 16 << sh pixels wide, so sub-tile m is the 16 pixels from column 16 (m mod 2^sh) of
// strip row m >> sh; the unit's resource starts at the strip's first pixel and rows are a whole image row apart.)
template <bool Z>
__device__ __forceinline__ int subtile_soffset(int m, bool live, const PipeEpi& ep) {
    if constexpr (Z) {
        const int r = m >> ep.sh, c = m - (r << ep.sh);
        return live ? r * ep.rowb + c * 16 * ep.cout4 : 0x7ff00000;
    } else {
        return live ? m * 16 * ep.cout4 : 0x7ff00000;
    }
}
// Epilogue of ONE parked sub-tile.  The host sends only these forms to the pipelined kernel:
//   no aux:  y = max_int(acc, lo)                    lo = 0 (ReLU) or INT_MIN (no activation)
//   mask:    y = (mask > 0) ? acc : 0                ReluGrad on the saved activation (dgrad)
//   skip:    y = max_int(max_int(acc, lo) + skip, lo2)      residual add, optional ReLU after it
template <int MAXG, int AUX, int NPART, bool Z = false>
__device__ __forceinline__ void pipe_epilogue_one(const PipePend<MAXG, AUX>& pd, int i, const PipeEpi& ep, int vst) {
    f32x4 v = pd.acc[i];
    if (AUX == 3) {
        v = srx_tanhf4(v);
    } else if (AUX == 1) {
        const f32x4 m = pd.aux[AUX ? i : 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mf = m[e], f = v[e];
            v[e] = (__float_as_int(mf) > 0) ? f : 0.0f;
        }
    } else if (AUX == 2) {
        v = clamp_lo4(clamp_lo4(v, ep.lo) + pd.aux[AUX ? i : 0], ep.lo2);
    } else if (AUX != 4) {
        v = clamp_lo4(v, ep.lo);
    }
    const int so = subtile_soffset<Z>(pd.m_first + i * NPART, i < pd.gs, ep);
    if (AUX == 4) {
        // channel c of LR pixel (h, w) is HR element (h r + c / (3 r), 3 r w + c % (3 r)): four 4-byte stores per lane, their
        // per-lane offsets fixed for the kernel (channels past the layer's 3 r^2: out of range)
        float f0 = v[0], f1 = v[1], f2 = v[2], f3 = v[3];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(f0), pd.yrs, vst, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(f1), pd.yrs, ep.vo1, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(f2), pd.yrs, ep.vo2, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(f3), pd.yrs, ep.vo3, so, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), pd.yrs, vst, so, 0);
    }
}

// One k-step (4 input channels of one tap) for G accumulators: G MFMAs, weight from an AGPR.
#define SRX_MFMA_AV(ACC, B) "v_mfma_f32_16x16x4_f32 %" #ACC ", %4, %" #B ", %" #ACC "\n\t"
__device__ __forceinline__ void mfma_sub_a(f32x4 (&c)[4], float w, float b0, float b1, float b2, float b3) {
    asm volatile(SRX_MFMA_AV(0, 5) SRX_MFMA_AV(1, 6) SRX_MFMA_AV(2, 7) SRX_MFMA_AV(3, 8)
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "a"(w), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
}
#define SRX_MFMA_AV3(ACC, B) "v_mfma_f32_16x16x4_f32 %" #ACC ", %3, %" #B ", %" #ACC "\n\t"
__device__ __forceinline__ void mfma_sub_a(f32x4 (&c)[3], float w, float b0, float b1, float b2, float) {
    asm volatile(SRX_MFMA_AV3(0, 4) SRX_MFMA_AV3(1, 5) SRX_MFMA_AV3(2, 6)
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]) : "a"(w), "v"(b0), "v"(b1), "v"(b2) : "memory");
}
// the group's very first k-step: the accumulators are DEFINED here, C operand = bias (no VALU initialisation)
#define SRX_MFMA_AVC(ACC, B) "v_mfma_f32_16x16x4_f32 %" #ACC ", %4, %" #B ", %9\n\t"
__device__ __forceinline__ void mfma_sub_a_first(f32x4 (&c)[4], f32x4 bias4, float w, float b0, float b1, float b2, float b3) {
    asm volatile(SRX_MFMA_AVC(0, 5) SRX_MFMA_AVC(1, 6) SRX_MFMA_AVC(2, 7) SRX_MFMA_AVC(3, 8)
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                 : "a"(w), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(bias4) : "memory");
}
#define SRX_MFMA_AVC3(ACC, B) "v_mfma_f32_16x16x4_f32 %" #ACC ", %3, %" #B ", %7\n\t"
__device__ __forceinline__ void mfma_sub_a_first(f32x4 (&c)[3], f32x4 bias4, float w, float b0, float b1, float b2, float) {
    asm volatile(SRX_MFMA_AVC3(0, 4) SRX_MFMA_AVC3(1, 5) SRX_MFMA_AVC3(2, 6)
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2])
                 : "a"(w), "v"(b0), "v"(b1), "v"(b2), "v"(bias4) : "memory");
}

struct PipeUnit {      // wave-uniform description of the running unit
    __amdgpu_buffer_rsrc_t yrs, auxrs;   // its output pixels / the aux operand's
};

// One group of G (<= MAXG) sub-tiles m_first, m_first+NPART, ...  On entry `cur` holds the LDS fragments of
// the group's first block and la[] its sub-tiles' LDS byte addresses; on exit the group is parked in `pd`
// and la / cur describe the unit's next group (tcur is the per-lane cursor into the address table).
//
// Instruction placement: a block (one tap x 16 input channels) is four k-step statements of G MFMAs; the
// other work is cut into pieces and dealt out over the gaps between them:
//                  gap 0..2 of every block: the LDS reads of the next block;
//                  blocks 0..NST-1, gap 0: one staging pass (scalar cursor + masked bounds-checked load);
//                  blocks NBLK/2.., gap 1: the LDS write of that pass;
//                  blocks E0.., gap 1: the previous group's epilogue, one sub-tile per block;
//                  block A0, gap 3: the next group's LDS addresses (table reads).
// No branches in here: every taken branch stalls the stream for an instruction refetch.
template <int KH, int KW, int CINP, int G, int MAXG, int AUX, int NPART, bool Z = false>
__device__ __forceinline__ void conv_group_pipe(const char* ldsb, const float (&wr)[KH * KW * (CINP / 4)],
                                                const f32x4 bias4, int row_stride_b, StageSeq& qi, StageSeq& qc,
                                                const StageGeo& SG, __amdgpu_buffer_rsrc_t xrs, int voff_lane, int wl_lane,
                                                int (&la)[MAXG], f32x4 (&cur)[MAXG], int& tcur,
                                                PipePend<MAXG, AUX>& pd, const PipeUnit& un, int m_first, int gs,
                                                const PipeEpi& ep, int vst, unsigned long long (&tt)[4]) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int NG = CINP / 16;
    constexpr int NBLK = KH * KW * NG;
    // (a strip tile has 6 (TH + 2) / TH passes per group of 4 sub-tiles of the WORKGROUP; with two waves per chunk a wave's
    // group is half of that work, so it carries 12 passes -- with 8, a third of every tile was left to the exposed drain)
    constexpr int NSTW = Z ? (NPART == 1 ? 8 : 12) : 6;
    constexpr int NST = (NBLK / 2 < NSTW) ? NBLK / 2 : NSTW;   // staging passes threaded through this group
    constexpr int E0 = NST;                               // first epilogue block
    constexpr int A0 = NST + MAXG;                        // address block
    constexpr int TSTEP = 16 * PS * 4 * NPART;            // table bytes from one sub-tile of the wave to the next
    constexpr int PPP = 256 / (CINP / 4);
    static_assert(NBLK >= NST + MAXG + 2, "group too short for the dealt-out schedule");
    const unsigned long long ts_m = SRX_STAMP();
    f32x4 acc[G];
    constexpr bool HASAUX = (AUX == 1 || AUX == 2);
    f32x4 aux[HASAUX ? G : 1];
    if (HASAUX) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            aux[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                un.auxrs, vst, subtile_soffset<Z>(m_first + i * NPART, i < gs, ep), 0));
        }
    }
    int lan[MAXG];
#pragma unroll
    for (int i = 0; i < MAXG; ++i) lan[i] = 0;
    f32x4 stg[NST];
    unsigned long long mk[Z ? 1 : NST], mt, mv = 0, mr = 0;   // (strips: the write mask comes from the commit cursor)
    int so;
    f32x4 nxt[MAXG];
    f32x4 cg[MAXG];
#pragma unroll
    for (int i = 0; i < MAXG; ++i) cg[i] = cur[i];
#pragma unroll
    for (int t = 0; t < NBLK; ++t) {
        const int tap = t / NG, gg = t % NG;
        const int wb = tap * (CINP / 4) + 4 * gg;
        const bool last = (t + 1 == NBLK);
        const int t1 = t + 1;
        const int kh1 = (t1 / NG) / KW, kw1 = (t1 / NG) % KW, g1 = t1 % NG;
        // LDS fragment i of the next block -- or, in the last block, of the next group's first block
        auto frag = [&](int i) {
            nxt[i] = last ? *reinterpret_cast<const f32x4*>(ldsb + lan[i])
                          : *reinterpret_cast<const f32x4*>(ldsb + la[i] + kh1 * row_stride_b + (kw1 * PS + 16 * g1) * 4);
        };
        const int nfr = last ? MAXG : G;     // fragments to fetch during this block
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (t == 0 && ks == 0)
                mfma_sub_a_first(acc, bias4, wr[wb + ks], cg[0][ks], cg[1][ks], cg[2][ks], cg[G - 1][ks]);
            else
                mfma_sub_a(acc, wr[wb + ks], cg[0][ks], cg[1][ks], cg[2][ks], cg[G - 1][ks]);
            // ---- gap ks
            if (ks == 0) {
                frag(0);
                if (t < NST) {
                    if constexpr (Z) stage_mask_az(qi, SG, mv, mt, mr);
                    else stage_mask_a(qi, SG, mk[t], mt);
                }
            } else if (ks == 1) {
                frag(1);
                if (t < NST) {
                    if constexpr (Z) stage_mask_bz(qi, mv, mt, mr, so);
                    else stage_mask_b(qi, mk[t], mt, so);
                }
                if (t >= NBLK / 2 && t < NBLK / 2 + NST) {
                    if constexpr (Z)
                        stage_commit(2 * (NST - 1 - (t - NBLK / 2)), qc, stage_mask_row(qc, SG), wl_lane, stg[t - NBLK / 2]);
                    else
                        stage_commit(NST - 1 - (t - NBLK / 2), qc, mk[t - NBLK / 2], wl_lane, stg[t - NBLK / 2]);
                }
                if (t >= E0 && t < E0 + MAXG) {
                    SRX_PIN(pd.acc[t - E0]);
                    pipe_epilogue_one<MAXG, AUX, NPART, Z>(pd, t - E0, ep, vst);
                }
            } else if (ks == 2) {
#pragma unroll
                for (int i = 2; i < nfr; ++i) frag(i);
                if (t < NST) {
                    if constexpr (Z) stg[t] = stage_fire_z(mv, mr, so, xrs, voff_lane);
                    else stg[t] = stage_fire(mk[t], so, xrs, voff_lane);
                }
                if (t >= NBLK / 2 && t < NBLK / 2 + NST) stage_next<PPP * PS * 4>(qc, SG.JP1, SG.rowfix_l);
            } else {
                if (t < NST) stage_next<PPP * CINP * 4>(qi, SG.JP1, SG.rowfix_g);
                if (t == A0) {
                    // the next group's LDS addresses: its sub-tiles are the next MAXG table entries of this lane
                    tcur += gs * TSTEP;
#pragma unroll
                    for (int i = 0; i < MAXG; ++i) lan[i] = *reinterpret_cast<const int*>(ldsb + tcur + i * TSTEP);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < nfr; ++i) cg[i] = nxt[i];
    }
#pragma unroll
    for (int i = 0; i < MAXG; ++i) { cur[i] = cg[i]; la[i] = lan[i]; }
    // park the group: software must cover the MFMA result latency before anything reads the accumulators
    // (here only the compiler's register moves; the arithmetic happens inside the next group)
    if constexpr (G == 4)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    else
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        pd.acc[i] = (i < G) ? acc[i < G ? i : 0] : bias4;
        if (HASAUX) pd.aux[i] = (i < G) ? aux[i < G ? i : 0] : bias4;
    }
    pd.yrs = un.yrs; pd.m_first = m_first; pd.gs = gs;
    tt[1] += SRX_STAMP() - ts_m;
}

// Z = false: full-width tiles (a tile is TH whole rows of the image, units are output rows);
// Z = true:  column strips of TW = 16 or 32 output columns with their own halo columns (units are the rows of a
//            strip; the last strip of an image is shifted left to end at the image's edge, so every strip is TW wide
//            and the columns two strips share are computed twice with identical results).
template <int KH, int KW, int CINP, int NCH, bool WT, int AUX, bool Z>
__device__ __forceinline__ void conv_pipe_body(const ConvArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TAPS = KH * KW;
    constexpr int KSPT = CINP / 4;
    constexpr int NPART = 4 / NCH;
    constexpr int TPP = CINP / 4, PPP = 256 / TPP;
    constexpr int PS = Lds<CINP>::PS;
    constexpr int MAXG = 4;
    constexpr int TSTEP = 16 * PS * 4 * NPART;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, part = wave / NCH;
    const int cout0 = chunk * 16;
    const int cb = cout0 + 4 * kq;
    const int c4 = tid % TPP, sp = tid / TPP;
    char* ldsb = reinterpret_cast<char*>(lds);

#ifdef SRX_TRACE
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    // the weight loads are issued first and collected last (after the first tile is staged): their latency overlaps
    // the LDS set-up and the first tile's loads
    float wr[TAPS * KSPT];
    load_stationary_weights<TAPS, CINP, WT>(wr, a, cout0, li, kq);
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cb + e < a.Cout) bias4[e] = a.bias[cb + e];
    }

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    if (u0 >= u1) return;
    // The host sends only shapes this scheme covers: exact-fit channels (Cin == CINP), full-width tiles
    // (NTX == 1, tw == OW), Cout a multiple of 4, images below 2^31 bytes, epilogue forms as listed above.
    const int buf_bytes = a.buf_floats * 4;
    const int rows_full = a.TH + KH - 1;
    const int row_stride_b = a.RS * PS * 4;

    // ---- one-time LDS set-up in BOTH buffers: the pad columns (zero for good: no pass ever writes them) and
    // the sub-tile address table in the slots' pad bytes: entry (m, L) = LDS byte address (in that buffer) of
    // pixel 16m + (L & 15) of a tile, channel 4 * (L >> 4); it sits in pad word (L & 3) of slot 16m + (L >> 2).
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; !Z && i < (rows_full + 1) * a.pad_l * TPP; i += 256) {
            const int q = i / TPP, ch = i % TPP;
            const int r = q / a.pad_l, c = q % a.pad_l;
            const int off = ((r * a.RS + c) * PS + 4 * ch) * 4;
            if ((r * a.RS + c) * PS * 4 < buf_bytes) {
                *reinterpret_cast<f32x4*>(ldsb + off) = z;
                *reinterpret_cast<f32x4*>(ldsb + buf_bytes + off) = z;
            }
        }
        const int tw_ = Z ? a.TW : a.OW;
        const int n_sub_max = (a.TH * tw_ + 15) >> 4;
        for (int i = tid; i < n_sub_max * 64; i += 256) {
            const int m = i >> 6, L = i & 63;
            const int t = 16 * m + (L & 15);
            int addr = 0;
            if (t < a.TH * tw_) {
                const int orow = t / tw_, ocol = t - orow * tw_;
                addr = ((orow * a.RS + ocol) * PS + 4 * (L >> 4)) * 4;
            }
            const int where = ((16 * m + (L >> 2)) * PS + CINP + (L & 3)) * 4;
            *reinterpret_cast<int*>(ldsb + where) = addr;
            *reinterpret_cast<int*>(ldsb + buf_bytes + where) = buf_bytes + addr;
        }
    }
    // ---- per-lane constants
    const int voff_lane = tid * 16;                         // image bytes of (slot sp, chunk c4) inside a pass
    const int wl_lane = (sp * PS + 4 * c4) * 4;             // the same inside LDS
    int vst = (cb < a.Cout) ? (li * a.Cout + cb) * 4 : kOobOffset;   // output bytes inside a sub-tile
    const int tlane = (((lane >> 2) + 16 * part) * PS + CINP + (lane & 3)) * 4;   // this lane's table entry of sub-tile `part`
    StageGeo SG;
    const int JP = (a.RS + PPP - 1) / PPP;
    SG.JP1 = __builtin_amdgcn_readfirstlane(JP - 1);
    SG.rowfix_g = __builtin_amdgcn_readfirstlane((a.W - JP * PPP) * CINP * 4);
    SG.rowfix_l = __builtin_amdgcn_readfirstlane((a.RS - JP * PPP) * PS * 4);
    SG.m_first = uniform64(__ballot(sp >= a.pad_l));
    SG.m_last = uniform64(__ballot((JP - 1) * PPP + sp < a.RS));
    SG.m_row = SG.m_last;
    PipeEpi ep = pipe_epi_setup<AUX>(a);
    if constexpr (AUX == 4) {
        const int rc = 3 * a.d2s_r, hr_row = a.OW * a.d2s_r * 3;       // floats
        int vo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cb + e, dy = c / rc;
            vo[e] = (c < a.Cout) ? (li * rc + dy * hr_row + (c - dy * rc)) * 4 : kOobOffset;
        }
        vst = vo[0]; ep.vo1 = vo[1]; ep.vo2 = vo[2]; ep.vo3 = vo[3];
    }
    const float* auxp = a.mask ? a.mask : a.skip;

    // tile descriptor of unit u (wave-uniform)
    auto tile_of = [&](int u, int& n, int& h, int& th, int& ow0) {
        h = u % a.OH;
        n = u / a.OH;
        ow0 = 0;
        if constexpr (Z) {
            const int tx = n % a.NTX;
            n = n / a.NTX;
            ow0 = tx * a.TW;
            if (ow0 > a.OW - a.TW) ow0 = a.OW - a.TW;
        }
        th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - u < th) th = u1 - u;
    };
    // cursors of a tile's staging: qi walks the image (issue), qc the LDS buffer (commit)
    auto stage_setup = [&](StageSeq& qi, StageSeq& qc, int h, int th, int ow0, int buf, bool active) {
        // (readfirstlane: once per tile, so that the cursors provably start out in SGPRs)
        const int left = __builtin_amdgcn_readfirstlane(active ? (th + KH - 1) * JP : 0);
        qi.j = 0; qc.j = 0;
        qi.off = __builtin_amdgcn_readfirstlane(((h - a.pad_t) * a.W + ow0) * CINP * 4);
        if constexpr (Z) {
            // image column of tile slot c is ow0 - pad_l + c
            SG.m_first = uniform64(__ballot(ow0 - a.pad_l + sp >= 0));
            const int c = (JP - 1) * PPP + sp;
            SG.m_last = uniform64(__ballot(c < a.RS && ow0 - a.pad_l + c < a.W));
        }
        qc.off = __builtin_amdgcn_readfirstlane(buf * buf_bytes);
        qi.left = left; qc.left = left;
        const int above = (a.pad_t > h) ? (a.pad_t - h) * JP : 0;      // passes in rows above the image
        qi.thr = __builtin_amdgcn_readfirstlane(left - above);
        qc.thr = 0;
    };

    unsigned long long tt[4] = {0, 0, 0, 0};   // trace: unit prologue, group sections, -, drain+barrier
    int n, h, th, ow0;
    tile_of(u0, n, h, th, ow0);
    [[maybe_unused]] const unsigned long long t_begin = SRX_STAMP();
    __syncthreads();     // (set-up writes above vs. the first tile's writes below touch different bytes; this orders them with the reads)
    {
        StageSeq qi, qc;
        stage_setup(qi, qc, h, th, ow0, 0, true);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + ((size_t)n * a.H * a.W - a.pad_l) * CINP, 0, (a.H * a.W + a.pad_l) * CINP * 4, 0x00020000);
        stage_tile_scalar<CINP, 6, Z>(qi, qc, SG, xrs, voff_lane, wl_lane);   // (6 loads in flight: the first tile is the only one whose latency is exposed)
    }
    lds_barrier();
    // now move the weights into the accumulation-register file for good: they are defined as "a" values here and
    // only ever consumed by "a" operands of the MFMA statements.  (One asm per weight right after its own load
    // would serialise 144 global-load round trips.)
#pragma unroll
    for (int i = 0; i < TAPS * KSPT; ++i) {
        float t = wr[i];
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(wr[i]) : "v"(t));
    }
    [[maybe_unused]] const unsigned long long t_first = SRX_STAMP();

    PipePend<MAXG, AUX> pd;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) pd.acc[i] = bias4;
    if (AUX == 1 || AUX == 2) {
#pragma unroll
        for (int i = 0; i < MAXG; ++i) pd.aux[i] = bias4;
    }
    pd.yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000); pd.m_first = 0; pd.gs = 0;

    int cur_buf = 0;
    int u = u0;
    while (u < u1) {
        const unsigned long long ts_u = SRX_STAMP();
        tile_of(u, n, h, th, ow0);
        const int un_ = u + th;
        const bool has_next = un_ < u1;
        int n2 = n, h2 = h, th2 = th, ow2 = ow0;
        if (has_next) tile_of(un_, n2, h2, th2, ow2);
        StageSeq qi, qc;
        stage_setup(qi, qc, h2, th2, ow2, cur_buf ^ 1, has_next);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + ((size_t)n2 * a.H * a.W - a.pad_l) * CINP, 0, (a.H * a.W + a.pad_l) * CINP * 4, 0x00020000);

        PipeUnit un;
        const int npx = th * (Z ? a.TW : a.OW);
        // (AUX 4: the unit's HR elements: r HR rows of 3 r OW floats per LR row, 3 r floats per LR pixel)
        const size_t unit_off = (AUX == 4) ? (((size_t)n * a.OH + h) * a.d2s_r * a.OW + ow0) * (size_t)(3 * a.d2s_r)
                                           : (((size_t)n * a.OH + h) * a.OW + ow0) * a.Cout;
        // (strips: the unit's bytes run from its first pixel to the last pixel of its last row)
        const int unit_bytes = (AUX == 4) ? ((th * a.d2s_r - 1) * a.OW + a.TW) * ep.cout4
                                          : (Z ? (th - 1) * a.OW + a.TW : npx) * ep.cout4;
        un.yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + unit_off, 0, unit_bytes, 0x00020000);
        un.auxrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((AUX == 1 || AUX == 2) ? auxp + unit_off : a.x), 0,
                                                     (AUX == 1 || AUX == 2) ? unit_bytes : 0, 0x00020000);

        const int n_sub = (npx + 15) >> 4;
        const int cnt = (n_sub - part + NPART - 1) / NPART;
        if (cnt > 0) {
            const int ng = (cnt + MAXG - 1) / MAXG;
            const int base = cnt / ng, rem = cnt % ng;
            int tcur = cur_buf * buf_bytes + tlane;
            int la[MAXG];
#pragma unroll
            for (int i = 0; i < MAXG; ++i) la[i] = *reinterpret_cast<const int*>(ldsb + tcur + i * TSTEP);
            f32x4 cur[MAXG];
#pragma unroll
            for (int i = 0; i < MAXG; ++i) cur[i] = *reinterpret_cast<const f32x4*>(ldsb + la[i]);
            tt[0] += SRX_STAMP() - ts_u;
            int idx = 0;
            for (int gi = 0; gi < ng; ++gi) {
                const int gs = base + (gi < rem ? 1 : 0);
                const int m_first = part + idx * NPART;
                if (gs == MAXG)
                    conv_group_pipe<KH, KW, CINP, MAXG, MAXG, AUX, NPART, Z>(ldsb, wr, bias4, row_stride_b, qi, qc, SG, xrs, voff_lane,
                                                                          wl_lane, la, cur, tcur, pd, un, m_first, gs, ep, vst, tt);
                else
                    conv_group_pipe<KH, KW, CINP, MAXG - 1, MAXG, AUX, NPART, Z>(ldsb, wr, bias4, row_stride_b, qi, qc, SG, xrs, voff_lane,
                                                                              wl_lane, la, cur, tcur, pd, un, m_first, gs, ep, vst, tt);
                idx += gs;
            }
        }
        // drain: whatever part of the next tile the groups did not cover
        const unsigned long long ts_d = SRX_STAMP();
        stage_tile_scalar<CINP, 6, Z>(qi, qc, SG, xrs, voff_lane, wl_lane);
        lds_barrier();
        tt[3] += SRX_STAMP() - ts_d;
        cur_buf ^= 1;
        u = un_;
    }
#pragma unroll
    for (int i = 0; i < MAXG; ++i) pipe_epilogue_one<MAXG, AUX, NPART, Z>(pd, i, ep, vst);
#ifdef SRX_TRACE
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 12;
        tr[0] = t_begin; tr[1] = SRX_STAMP(); tr[2] = tt[1]; tr[3] = tt[3];
        tr[4] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        tr[5] = rt_entry; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = t_entry;
        tr[8] = t_first - t_begin; tr[9] = 0; tr[10] = tt[0]; tr[11] = tt[2];
    }
#endif
}

template <int KH, int KW, int CINP, int NCH, bool WT, int AUX>
__global__ __launch_bounds__(256, 1) void conv_pipe_kernel(const ConvArgs a) {
    conv_pipe_body<KH, KW, CINP, NCH, WT, AUX, false>(a);
}
// the same for images too wide for full-width tiles: column strips
template <int KH, int KW, int CINP, int NCH, bool WT, int AUX>
__global__ __launch_bounds__(256, 1) void conv_pipe_strip_kernel(const ConvArgs a) {
    static_assert(NCH == 4 || NCH == 2, "strip tiles: 64 or 32 output channels");
    static_assert(AUX != 4 || (!WT && NCH == 2), "sub-pixel store: forward, <= 32 output channels");
    conv_pipe_body<KH, KW, CINP, NCH, WT, AUX, true>(a);
}

// ---------------------------------------------------------------------------------------------
// Generic forward / dgrad for filter shapes outside the tuned set (runtime KH, KW; weights are
// streamed from global/L1 per MFMA instead of living in registers).  Same tiling, same
// epilogue, same numerics; slower.
// ---------------------------------------------------------------------------------------------
template <int CINP, bool WT, int G>
__device__ __forceinline__ void conv_group_generic(const float* lds, const f32x4 bias4, const ConvArgs& a, int KH,
                                                   int KW, int n, int h, int ow0, int th, int tw, float inv_tw,
                                                   int m_first, int m_step, int li, int kq, int cout0) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int NG = (CINP >= 16) ? CINP / 16 : 1;
    const int TAPS = KH * KW;
    const int npx = th * tw;
    const int cb = cout0 + 4 * kq;
    const bool vec = ((a.Cout & 3) == 0) && (cb + 3 < a.Cout) && !a.d2s_r;
    const size_t img_base = (size_t)n * a.OH * a.OW * a.Cout;   // wave-uniform
    int laddr[G];       // float index of this lane's pixel (tap 0,0) in LDS
    unsigned off[G];    // element offset of this lane's 4 output channels inside image n
    bool valid[G];
    f32x4 acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m_first + i * m_step) + li;
        valid[i] = (t < npx) & (cb < a.Cout);
        const int tt = (t < npx) ? t : 0;
        const int orow = fdiv_small(tt, inv_tw, tw);
        const int ocol = tt - orow * tw;
        laddr[i] = ((orow * a.RS + ocol) * PS << (a.stride - 1)) + ((CINP >= 16) ? 4 * kq : kq);   // (stride 2: tile position (2 orow, 2 ocol))
        // (sub-pixel store mode: the offset of HR pixel (h*r, w*r), channel 0 -- conv_epilogue adds the channel part)
        off[i] = !valid[i] ? 0u
                 : (a.d2s_r ? (unsigned)(((h + orow) * a.d2s_r * a.OW + ow0 + ocol) * a.d2s_rc)
                            : (unsigned)(((h + orow) * a.OW + ow0 + ocol) * a.Cout + cb));
        acc[i] = bias4;
    }
    f32x4 aux[G];
    // (this kernel is always built with AUX: without a skip / mask operand there is nothing to fetch -- the
    // unconditional prefetch would read through a null pointer)
    conv_prefetch_aux<G, true>(aux, off, a, img_base, vec && (a.mask != nullptr || a.skip != nullptr));
    const int row_stride = a.RS * PS;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;
    for (int kh = 0; kh < KH; ++kh) {
        for (int kw = 0; kw < KW; ++kw) {
            const int tap = kh * KW + kw;
            const int wtap = WT ? (TAPS - 1 - tap) : tap;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f32x4 wv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int ci = (CINP >= 16) ? 16 * g + 4 * kq + s : kq;
                    if (ci < a.Cin && co_ok && (CINP >= 16 || s == 0))
                        wv[s] = WT ? a.w[((size_t)wtap * a.Cout + co) * a.Cin + ci]
                                   : a.w[((size_t)wtap * a.Cin + ci) * a.Cout + co];
                }
                if constexpr (CINP >= 16) {
                    f32x4 av[G];
#pragma unroll
                    for (int i = 0; i < G; ++i)
                        av[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i] + kh * row_stride + kw * PS + 16 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int i = 0; i < G; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], av[i][s], acc[i], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < G; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], lds[laddr[i] + kh * row_stride + kw * PS],
                                                                      acc[i], 0, 0, 0);
                }
            }
        }
    }
    conv_epilogue<G, true>(acc, aux, valid, off, a, img_base, cb, vec);
}

template <int CINP, int NCH, bool WT>
__global__ __launch_bounds__(256, 2) void conv_mfma_generic_kernel(const ConvArgs a, int KH, int KW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NPART = 4 / NCH;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, part = wave / NCH;
    const int cout0 = chunk * 16;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cout0 + 4 * kq + e < a.Cout) bias4[e] = a.bias[cout0 + 4 * kq + e];
    }
    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    if (a.stagger) stagger_second_workgroup(a.stagger);
    int u = u0;
    while (u < u1) {
        const int h = u % a.OH;
        const int t = u / a.OH;
        const int tx = t % a.NTX;
        const int n = t / a.NTX;
        int th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - u < th) th = u1 - u;
        const int ow0 = tx * a.TW;
        const int tw = (a.OW - ow0 < a.TW) ? (a.OW - ow0) : a.TW;
        const int n_need = ((th - 1) * a.stride + KH) * a.RS + (KW - 1);
        lds_barrier();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h * a.stride - a.pad_t, ow0 * a.stride - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        lds_barrier();
        const int n_sub = (th * tw + 15) >> 4;
        const float inv_tw = 1.0f / (float)tw;
        for (int m = part; m < n_sub; m += 2 * NPART) {
            if (m + NPART < n_sub)
                conv_group_generic<CINP, WT, 2>(lds, bias4, a, KH, KW, n, h, ow0, th, tw, inv_tw, m, NPART, li, kq, cout0);
            else
                conv_group_generic<CINP, WT, 1>(lds, bias4, a, KH, KW, n, h, ow0, th, tw, inv_tw, m, NPART, li, kq, cout0);
        }
        u += th;
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad: dW[tap][ci][co] += sum_p x[p + tap][ci] * dpre[p][co]
// Rows of the (tap, ci) space are flattened as R = tap*CINP + ci; one b128 LDS read by lane (i, kq)
// covers R = 64q + 4i + g (g = 0..3) for pixel kq of the step, feeding 4 MFMAs.
// ---------------------------------------------------------------------------------------------
// 4 MFMAs of one LDS fragment: acc[g] += x[g] (rows = input channels) * b (cols = output channels)
__device__ __forceinline__ void mfma4_wgrad(f32x4 (&c)[4], const f32x4 x, float b) {
    asm volatile(SRX_MFMA(0, 4, 8) SRX_MFMA(1, 5, 8) SRX_MFMA(2, 6, 8) SRX_MFMA(3, 7, 8)
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(b));
}

// per-lane cursor over the tile's pixels, 4 pixels per step (lane group kq takes pixel 4*step + kq)
struct WgCursor {
    int p;      // pixel index inside the tile
    int c;      // its column
    int xaddr;  // float index of its (tap 0,0) slot in LDS
    int boff;   // element offset of dpre[pixel][0] from the tile's first pixel
};

template <int KH, int KW, int CINP, int NCH, int MINW>
__global__ __launch_bounds__(256, MINW) void wgrad_mfma_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TAPS = KH * KW;
    constexpr int ROWS = TAPS * CINP;
    constexpr int Q = (ROWS + 63) / 64;
    constexpr int NQP = 4 / NCH;             // waves sharing a cout chunk split the q's
    constexpr int QW = (Q + NQP - 1) / NQP;  // q's (LDS fragments per step) of this wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, qpart = wave / NCH;
    const int cout0 = chunk * 16;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;
    const int co_c = co_ok ? co : a.Cout - 1;   // clamped: columns >= Cout are never written out

    __builtin_amdgcn_s_setprio(2);   // see conv_group: only the MFMA stream runs at priority 0
    // per-lane LDS offset (floats) of the tap/channel this lane feeds for each of its q's
    int toff[QW];
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        int R = 64 * q + 4 * li;  // first of the lane's 4 rows (all 4 share the tap since CINP % 4 == 0)
        if (q >= Q || R >= ROWS) R = 0;
        const int tap = R / CINP, ci = R % CINP;
        toff[k] = ((tap / KW) * a.RS + (tap % KW)) * PS + ci;
    }
    // one zeroed pixel slot after the tile: the operand of pixels beyond the tile's end
    const int zaddr = a.zero_slot * PS + (4 * li) % CINP;
    if (tid < PS) lds[a.zero_slot * PS + tid] = 0.f;

    f32x4 acc[QW][4];
#pragma unroll
    for (int k = 0; k < QW; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[k][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    if (a.stagger) stagger_second_workgroup(a.stagger);
    int u = u0;
    while (u < u1) {
        const int h = u % a.OH;
        const int t = u / a.OH;
        const int tx = t % a.NTX;
        const int n = t / a.NTX;
        int th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - u < th) th = u1 - u;
        const int ow0 = tx * a.TW;
        const int tw = (a.OW - ow0 < a.TW) ? (a.OW - ow0) : a.TW;
        const int n_need = ((th - 1) * a.stride + KH) * a.RS + (KW - 1);

        lds_barrier();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h * a.stride - a.pad_t, ow0 * a.stride - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        lds_barrier();

        const int npx = th * tw;
        const int nsteps = (npx + 3) >> 2;
        const float* dbase = a.dpre + (((size_t)n * a.OH + h) * a.OW + ow0) * a.Cout + co_c;  // + 32-bit offsets
        const int x_step = 4 * PS * a.stride, x_wrap = (a.RS - tw) * PS * a.stride;   // (stride 2: pixel (r, c) sits at tile slot (2r, 2c))
        const int b_step = 4 * a.Cout, b_wrap = (a.OW - tw) * a.Cout;

        auto advance = [&](WgCursor& cu) {
            cu.p += 4; cu.c += 4; cu.xaddr += x_step; cu.boff += b_step;
            while (cu.c >= tw) { cu.c -= tw; cu.xaddr += x_wrap; cu.boff += b_wrap; }   // once, unless tw < 4
        };
        auto load_b = [&](const WgCursor& cu) -> float {
            // always a valid address: pixels past the end re-read the tile's first pixel (their x operand is 0)
            return dbase[cu.p < npx ? cu.boff : 0];
        };
        auto read_x = [&](const WgCursor& cu, int k) -> f32x4 {
            return *reinterpret_cast<const f32x4*>(lds + (cu.p < npx ? cu.xaddr + toff[k] : zaddr));
        };

        WgCursor cur;
        {
            const int r0 = fdiv_small(kq < npx ? kq : 0, 1.0f / (float)tw, tw);
            cur.p = kq; cur.c = (kq < npx ? kq : 0) - r0 * tw;
            cur.xaddr = (r0 * a.RS + cur.c) * PS * a.stride; cur.boff = (r0 * a.OW + cur.c) * a.Cout;
        }
        WgCursor pf = cur;                      // dpre prefetch cursor, 3 steps ahead
        float bq[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { bq[j] = load_b(pf); advance(pf); }

        // ring of 3 LDS fragments running LA fragments ahead of the MFMAs, across step boundaries
        // (fragment f of the stream = (step f / QW, k = f % QW); with one fragment per step the
        // look-ahead is 1 so that only the next step's cursor is needed)
        constexpr int LA = (QW >= 2) ? 2 : 1;
        f32x4 ring[3];
        WgCursor nxt = cur;
        advance(nxt);
        ring[0] = read_x(cur, 0);
        if (LA == 2) ring[1] = read_x(cur, 1);

        __builtin_amdgcn_s_setprio(0);
        for (int s0 = 0; s0 < nsteps; s0 += 3) {
#pragma unroll
            for (int uu = 0; uu < 3; ++uu) {
                if (s0 + uu < nsteps) {
                    const float b = bq[uu];
                    bq[uu] = load_b(pf);
                    advance(pf);
                    bsum += (cur.p < npx && co_ok) ? b : 0.f;
#pragma unroll
                    for (int k = 0; k < QW; ++k) {
                        const int idx = (uu * QW + k) % 3;
                        const int kk = k + LA;
                        ring[(idx + LA) % 3] = (kk < QW) ? read_x(cur, kk) : read_x(nxt, kk - QW);
                        mfma4_wgrad(acc[k], ring[idx], b);
                    }
                    cur = nxt;
                    advance(nxt);
                }
            }
        }
        __builtin_amdgcn_s_setprio(2);
        u += th;
    }

    // MFMA results are read by VALU / stores next
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // ---- write this workgroup's partial
    float* pw = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        if (q >= Q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = 64 * q + 4 * (4 * kq + r) + g;
                const int tap = R / CINP, ci = R % CINP;
                if (R < ROWS && ci < a.Cin && co_ok) pw[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[k][g][r];
            }
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (qpart == 0 && kq == 0 && co_ok) pw[(size_t)TAPS * a.Cin * a.Cout + co] = bsum;
}


// ---------------------------------------------------------------------------------------------
// Generic filter gradient for filter shapes outside the tuned set (runtime KH, KW -- the counterpart of
// conv_mfma_generic_kernel, so that the C ABI accepts the same layers in all three directions).  The cursor kernel above with
// the (tap, ci) rows cut into passes of at most 9 taps (the accumulators of 9 x CINP rows are what fits the register file):
// pass `tap0` accumulates the rows of taps tap0 .. tap0 + ntaps - 1 over all pixels and writes them into the workgroup's
// partial; the host launches ceil(KH KW / 9) passes with the same grid, then the usual reduction.  Same numerics; slower (the
// inputs are read once per pass).  Stride 1.
// ---------------------------------------------------------------------------------------------
template <int CINP, int NCH>
__global__ __launch_bounds__(256, 2) void wgrad_generic_kernel(const WgradArgs a, int KH, int KW, int tap0, int ntaps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TPASS = 9;
    constexpr int ROWS = TPASS * CINP;
    constexpr int Q = (ROWS + 63) / 64;
    constexpr int NQP = 4 / NCH;
    constexpr int QW = (Q + NQP - 1) / NQP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, qpart = wave / NCH;
    const int cout0 = chunk * 16;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;
    const int co_c = co_ok ? co : a.Cout - 1;
    int toff[QW];
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        int R = 64 * q + 4 * li;
        int tap = tap0 + R / CINP;
        if (q >= Q || R >= ROWS || tap >= KH * KW) { R = 0; tap = tap0; }      // (rows of no tap: any finite operand, never written out)
        toff[k] = ((tap / KW) * a.RS + (tap % KW)) * PS + R % CINP;
    }
    const int zaddr = a.zero_slot * PS + (4 * li) % CINP;
    if (tid < PS) lds[a.zero_slot * PS + tid] = 0.f;
    f32x4 acc[QW][4];
#pragma unroll
    for (int k = 0; k < QW; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[k][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    int u = u0;
    while (u < u1) {
        const int h = u % a.OH;
        const int t = u / a.OH;
        const int tx = t % a.NTX;
        const int n = t / a.NTX;
        int th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - u < th) th = u1 - u;
        const int ow0 = tx * a.TW;
        const int tw = (a.OW - ow0 < a.TW) ? (a.OW - ow0) : a.TW;
        const int n_need = (th - 1 + KH) * a.RS + (KW - 1);
        lds_barrier();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        lds_barrier();
        const int npx = th * tw;
        const int nsteps = (npx + 3) >> 2;
        const float* dbase = a.dpre + (((size_t)n * a.OH + h) * a.OW + ow0) * a.Cout + co_c;
        // per-lane cursor: lane group kq takes pixel 4 step + kq of the tile
        int p = kq, c = 0, xaddr = 0, boff = 0;
        {
            const int r0 = fdiv_small(kq < npx ? kq : 0, 1.0f / (float)tw, tw);
            c = (kq < npx ? kq : 0) - r0 * tw;
            xaddr = (r0 * a.RS + c) * PS; boff = (r0 * a.OW + c) * a.Cout;
        }
        for (int s = 0; s < nsteps; ++s) {
            const bool live = p < npx;
            const float b = live ? dbase[boff] : 0.f;
            bsum += co_ok ? b : 0.f;
#pragma unroll
            for (int k = 0; k < QW; ++k) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(lds + (live ? xaddr + toff[k] : zaddr));
                mfma4_wgrad(acc[k], xv, b);
            }
            p += 4; c += 4; xaddr += 4 * PS; boff += 4 * a.Cout;
            while (c >= tw) { c -= tw; xaddr += (a.RS - tw) * PS; boff += (a.OW - tw) * a.Cout; }
        }
        u += th;
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    float* pw = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        if (q >= Q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = 64 * q + 4 * (4 * kq + r) + g;
                const int lt = R / CINP, ci = R % CINP, tap = tap0 + lt;
                if (R < ROWS && lt < ntaps && tap < KH * KW && ci < a.Cin && co_ok) pw[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[k][g][r];
            }
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (tap0 == 0 && qpart == 0 && kq == 0 && co_ok) pw[(size_t)KH * KW * a.Cin * a.Cout + co] = bsum;
}

// ---------------------------------------------------------------------------------------------
// wgrad, linear walk (full-width tiles).  Same data flow as wgrad_mfma_kernel -- accumulators
// stationary, x from the LDS halo tile, dpre from global -- but the K loop walks the PADDED positions
// p' = orow*RS + ocol of the tile, 4 per step, instead of the real pixels.  In that space the x operand
// of (position, tap) is LDS slot p' + kh*RS + kw: per lane a constant plus step*const, i.e. ds_read
// immediates inside an unrolled window of U steps and one v_add per fragment stream per window.  The
// dpre operand of the fake positions (ocol >= OW) must be 0 and the real ones sit at pixel
// p' - orow*(RS-OW): a scalar state machine (column of the step's first position) classifies the step's
// four positions -- before the row end, fake, in the next row -- as lane masks, and one bounds-checked
// buffer load per step takes its lane offset through two selects on them (next-row lanes start
// RS-OW pixels earlier, fake lanes are sent out of range -> 0); the step part of the address is the
// scalar offset.  VALU instructions per 36-MFMA step: 3 (two selects, bias-gradient add) + 9/U, against ~46 in
// wgrad_mfma_kernel (cursor advances, end-of-tile selects, address adds).  See conv_pipe_kernel for why
// that is what matters.
// ---------------------------------------------------------------------------------------------
struct DpreSeq {       // wave-uniform state of the dpre stream: the next step to load
    int c0;            // column (0..RS-1) of the step's first position
    int soff;          // byte offset of that position's real pixel from the unit's first pixel, channel 0
};
__device__ __forceinline__ void dpre_counts(const DpreSeq& q, int tw, int RS, int& nA, int& nAF) {
    asm volatile("s_sub_i32 %0, %3, %2\n\ts_max_i32 %0, %0, 0\n\ts_min_i32 %0, %0, 4\n\t"
                 "s_sub_i32 %1, %4, %2\n\ts_min_i32 %1, %1, 4"
                 : "=&s"(nA), "=&s"(nAF) : "s"(q.c0), "s"(tw), "s"(RS) : "scc");
}
// lanes of the first n (0..4) 16-lane groups
__device__ __forceinline__ unsigned long long low_groups_mask(int n) {
    unsigned long long m;
    int w;
    asm volatile("s_lshl_b32 %1, %2, 4\n\ts_bfm_b64 %0, %1, 0\n\ts_cmp_eq_u32 %2, 4\n\ts_cselect_b64 %0, -1, %0"
                 : "=&s"(m), "=&s"(w) : "s"(n) : "scc");
    return m;
}
__device__ __forceinline__ void dpre_masks(unsigned long long mA, unsigned long long mAF, unsigned long long& mF,
                                           unsigned long long& mB) {
    asm volatile("s_andn2_b64 %0, %3, %2\n\ts_not_b64 %1, %3" : "=&s"(mF), "=&s"(mB) : "s"(mA), "s"(mAF) : "scc");
}
// The load itself is an ordinary (compiler-visible) buffer load: the compiler tracks its latency, and the
// prefetch ring may live across loop back-edges.  Only the lane offset is patched, by two selects on the
// scalar masks (2 VALU per step): next-row lanes start pad pixels earlier, fake lanes go out of range.
__device__ __forceinline__ float dpre_fire(unsigned long long mF, unsigned long long mB, __amdgpu_buffer_rsrc_t rsrc,
                                           int voff, int voff_next_row, int soff) {
    int v;
    asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(v) : "v"(voff), "v"(voff_next_row), "s"(mB));
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v) : "v"(kOobOffset), "s"(mF));
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, v, soff, 0));
}
__device__ __forceinline__ void dpre_next(DpreSeq& q, int RS, int padb, int stepb) {
    int t, t2;
    asm volatile("s_add_i32 %0, %0, 4\n\ts_add_i32 %1, %1, %6\n\ts_cmp_ge_i32 %0, %4\n\t"
                 "s_cselect_b32 %2, %4, 0\n\ts_cselect_b32 %3, %5, 0\n\ts_sub_i32 %0, %0, %2\n\ts_sub_i32 %1, %1, %3"
                 : "+s"(q.c0), "+s"(q.soff), "=&s"(t), "=&s"(t2) : "s"(RS), "s"(padb), "s"(stepb) : "scc");
}
struct DpreGeo {       // wave-uniform constants of the dpre stream
    int tw, RS, padb, stepb;
};
// all pieces of one step at once (prologue of a unit)
__device__ __forceinline__ float dpre_step_now(DpreSeq& q, const DpreGeo& D, __amdgpu_buffer_rsrc_t rsrc, int voff,
                                               int voff_next_row) {
    int nA, nAF;
    unsigned long long mF, mB;
    dpre_counts(q, D.tw, D.RS, nA, nAF);
    const unsigned long long mA = low_groups_mask(nA), mAF = low_groups_mask(nAF);
    dpre_masks(mA, mAF, mF, mB);
    const float b = dpre_fire(mF, mB, rsrc, voff, voff_next_row, q.soff);
    dpre_next(q, D.RS, D.padb, D.stepb);
    return b;
}

// Z = true: column strips (images too wide for full-width tiles).  Strips are disjoint (the last one of an image
// may be narrower); the x tile is staged with the strip variant of the scalar stager (edge halo columns loaded as
// zeros), the dpre walk only needs the strip's width as the number of real positions per tile row and the image's
// row pitch in the row-wrap correction (which then has the other sign: a tile row is shorter than an image row).
// PACK3 (RGB-input layers: SRCNN 9x9 3 -> 64, ESPCN 5x5 3 -> 64; CINP must be 4): the LDS pixel is 3 floats and the MFMA's rows are
// (kh, kw * 3 + ci) -- 27 per filter row for 9x9: 243 rows in 4 groups of 64 instead of 81 x 4 = 324 in 6 (a quarter of them the
// zero fourth channel): 16 instead of 24 MFMAs per step.  A lane's four rows are four consecutive (kw, ci) values -- across a
// filter-row end they are not consecutive floats --, so a fragment is four ds_read_b32 at four per-lane bases.
template <int KH, int KW, int CINP, int NCH, int MINW, bool Z, bool PACK3 = false>
__device__ __forceinline__ void wgrad_lin_body(const WgradArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(!PACK3 || (CINP == 4 && !Z), "PACK3: 3-channel input staged as 3-float pixels, full-width tiles");
    constexpr int PS = PACK3 ? 3 : Lds<CINP>::PS;
    constexpr int TAPS = KH * KW;
    constexpr int ROWS = PACK3 ? KH * KW * 3 : TAPS * CINP;
    constexpr int NXB = PACK3 ? 4 : 1;           // per-lane bases of a fragment stream
    constexpr int Q = (ROWS + 63) / 64;
    constexpr int NQP = 4 / NCH;             // waves sharing a cout chunk split the q's
    constexpr int QW = (Q + NQP - 1) / NQP;  // q's (LDS fragments per step) of this wave
    constexpr int TPP = CINP / 4, PPP = 256 / TPP;
    constexpr int U = 12;                    // steps per unrolled window (U*QW is a multiple of 3, U of 4)
    constexpr int XSTEP = 4 * PS * 4;        // LDS bytes from one step to the next
    constexpr int PF = 3;                    // dpre prefetch distance in steps
    constexpr int LA = (QW >= 2) ? 2 : 1;    // LDS fragments in flight ahead of the MFMAs (3 or 5 measured no faster)
    constexpr int RN = LA + 1;               // fragment ring
    static_assert((U * QW) % RN == 0 && U % 4 == 0, "window must keep the register rings in phase");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, qpart = wave / NCH;
    const int cout0 = chunk * 16;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;
    const int co_c = co_ok ? co : a.Cout - 1;   // clamped: columns >= Cout are never written out
    const int c4 = tid % TPP, sp = tid / TPP;
    char* ldsb = reinterpret_cast<char*>(lds);

    // per-lane LDS byte offset of (position kq of a step, the lane's tap / channels) for each of its q's
    int xb[QW][NXB];
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        if constexpr (PACK3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int R = 64 * q + 4 * li + g;
                if (q >= Q || R >= ROWS) R = 0;          // (rows past the filter: any finite operand, never written out)
                const int kh = R / (3 * KW), rem = R % (3 * KW);
                xb[k][g] = ((kq + kh * a.RS) * PS + rem) * 4;
            }
        } else {
            int R = 64 * q + 4 * li;  // first of the lane's 4 rows (all 4 share the tap since CINP % 4 == 0)
            if (q >= Q || R >= ROWS) R = 0;
            const int tap = R / CINP, ci = R % CINP;
            xb[k][0] = ((kq + (tap / KW) * a.RS + (tap % KW)) * PS + ci) * 4;
        }
    }
    // the whole tile buffer starts out as zeros: pad columns (never written by the scalar staging below) and
    // the slots past a short tile that the last step may touch (their dpre operand is 0, they must be finite)
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const int n16 = (a.zero_slot + 4) * (PACK3 ? 4 : PS) / 4;     // (the host allocates 4 slots past the largest tile; PACK3: sized for 4-float slots)
        for (int i = tid; i < n16; i += 256) reinterpret_cast<f32x4*>(lds)[i] = z;
    }
    // scalar staging (see conv_pipe_kernel) when the channels fit exactly, else the generic stager
    const bool lean = Z || ((a.Cin == CINP) && (a.RS >= PPP));   // (the host sends strips only when this holds)
    const int voff_lane = tid * 16;
    const int wl_lane = (sp * PS + 4 * c4) * 4;
    StageGeo SG;
    const int JP = (a.RS + PPP - 1) / PPP;
    SG.JP1 = __builtin_amdgcn_readfirstlane(JP - 1);
    SG.rowfix_g = __builtin_amdgcn_readfirstlane((a.W - JP * PPP) * CINP * 4);
    SG.rowfix_l = __builtin_amdgcn_readfirstlane((a.RS - JP * PPP) * PS * 4);
    SG.m_first = uniform64(__ballot(sp >= a.pad_l));
    SG.m_last = uniform64(__ballot((JP - 1) * PPP + sp < a.RS));
    SG.m_row = SG.m_last;
    SG.m_mid = ~0ull;
    DpreGeo DG;
    DG.tw = __builtin_amdgcn_readfirstlane(Z ? a.TW : a.OW);
    DG.RS = __builtin_amdgcn_readfirstlane(a.RS);
    // row-wrap correction: a tile row has RS positions, an image row OW pixels (negative for strips)
    const int padb = (a.RS - a.OW) * a.Cout * 4;
    DG.padb = __builtin_amdgcn_readfirstlane(padb);
    DG.stepb = __builtin_amdgcn_readfirstlane(16 * a.Cout);
    // lane offsets count from `bias_b` bytes BEFORE the unit's first pixel (the buffer resource starts there; those
    // bytes are never touched), so that both variants are non-negative: offsets are unsigned to the hardware
    const int bias_b = padb > 0 ? padb : 0;
    const int voff_b = (kq * a.Cout + co_c) * 4 + bias_b;            // lane in the row of the step's first position
    const int voff_bn = voff_b - padb;                               // lane whose position lies in the next tile row

    f32x4 acc[QW][4];
#pragma unroll
    for (int k = 0; k < QW; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[k][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    if (a.stagger) stagger_second_workgroup(a.stagger);
    int u = u0;
    while (u < u1) {
        const int h = u % a.OH;
        int n = u / a.OH;
        int ow0 = 0, tw = a.OW;
        if constexpr (Z) {
            const int tx = n % a.NTX;
            n = n / a.NTX;
            ow0 = tx * a.TW;
            tw = a.OW - ow0 < a.TW ? a.OW - ow0 : a.TW;
        }
        int th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - u < th) th = u1 - u;

        lds_barrier();
        if (lean) {
            StageSeq qi, qc;
            const int left = __builtin_amdgcn_readfirstlane((th + KH - 1) * JP);
            const int above = (a.pad_t > h) ? (a.pad_t - h) * JP : 0;
            qi.j = 0; qc.j = 0;
            qi.off = __builtin_amdgcn_readfirstlane(((h - a.pad_t) * a.W + ow0) * CINP * 4);
            qc.off = 0;
            if constexpr (Z) {
                // image column of tile slot c is ow0 - pad_l + c; a narrow last strip has invalid columns before the
                // last pass too (the host keeps a tile row to at most three passes: first / between / last)
                const int iw0 = ow0 - a.pad_l + sp;
                const unsigned long long v0 = __ballot(iw0 >= 0 && iw0 < a.W && sp < a.RS);
                const int cl = (JP - 1) * PPP + sp;
                const unsigned long long vl = __ballot(cl < a.RS && iw0 + (JP - 1) * PPP >= 0 && iw0 + (JP - 1) * PPP < a.W);
                SG.m_first = uniform64(JP == 1 ? (v0 & vl) : v0);
                SG.m_last = uniform64(vl);
                SG.m_mid = uniform64(__ballot(iw0 + PPP >= 0 && iw0 + PPP < a.W));
            }
            qi.left = left; qc.left = left;
            qi.thr = __builtin_amdgcn_readfirstlane(left - above);
            qc.thr = 0;
            const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + ((size_t)n * a.H * a.W - a.pad_l) * CINP, 0, (a.H * a.W + a.pad_l) * CINP * 4, 0x00020000);
            stage_tile_scalar<CINP, 6, Z, Z>(qi, qc, SG, xrs, voff_lane, wl_lane);
        } else if constexpr (PACK3) {
            // 3-float pixels: slot s <-> (row s / RS, column s % RS) of the tile; out-of-image slots are zeros
            const int n_need = (th + KH - 1) * a.RS + (KW - 1);
            const float* xn = a.x + (size_t)n * a.H * a.W * 3;
            for (int s0 = tid; s0 < n_need; s0 += 4 * 256) {
                float v[4][3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sl = s0 + j * 256;
                    const int r = fdiv_small(sl < n_need ? sl : 0, a.inv_rs, a.RS);
                    const int c = sl - r * a.RS;
                    const int ih = h - a.pad_t + r, iw = c - a.pad_l;
                    const bool ok = (sl < n_need) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                    const unsigned off = ok ? (unsigned)((ih * a.W + iw) * 3) : 0u;
#pragma unroll
                    for (int e = 0; e < 3; ++e) { const float t = xn[off + e]; v[j][e] = ok ? t : 0.f; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sl = s0 + j * 256;
                    if (sl < n_need) {
#pragma unroll
                        for (int e = 0; e < 3; ++e) lds[sl * 3 + e] = v[j][e];
                    }
                }
            }
        } else {
            stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, -a.pad_l, a.RS, a.inv_rs,
                             (th + KH - 1) * a.RS + (KW - 1), tid);
        }
        lds_barrier();

        const int nsteps = (th * a.RS + 3) >> 2;
        // the unit's pixels: th rows, the last one ending after its tw real positions
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<float*>(reinterpret_cast<char*>(const_cast<float*>(a.dpre) + (((size_t)n * a.OH + h) * a.OW + ow0) * a.Cout) - bias_b), 0,
            ((th - 1) * a.OW + tw) * a.Cout * 4 + bias_b, 0x00020000);
        const int tw_u = __builtin_amdgcn_readfirstlane(tw);
        DpreSeq dq;
        dq.c0 = __builtin_amdgcn_readfirstlane(0);
        dq.soff = __builtin_amdgcn_readfirstlane(0);
        float bq[4];
        DG.tw = tw_u;
#pragma unroll
        for (int j = 0; j < PF; ++j) bq[j] = dpre_step_now(dq, DG, brs, voff_b, voff_bn);
        bq[3] = 0.f;

        int xw[QW][NXB];      // window base of each fragment stream
#pragma unroll
        for (int k = 0; k < QW; ++k)
#pragma unroll
            for (int g = 0; g < NXB; ++g) xw[k][g] = xb[k][g];
        auto read_x = [&](int k, int uu) -> f32x4 {
            if constexpr (PACK3) {
                f32x4 v;
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const float*>(ldsb + xw[k][g] + uu * XSTEP);
                return v;
            } else {
                return *reinterpret_cast<const f32x4*>(ldsb + xw[k][0] + uu * XSTEP);
            }
        };
        f32x4 ring[RN];
#pragma unroll
        for (int f = 0; f < LA; ++f) ring[f] = read_x(f % QW, f / QW);

        for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                if (s0 + uu < nsteps) {
                    const float b = bq[uu % 4];      // dpre of this step, loaded PF steps ago
                    bsum += b;
                    int nA = 0, nAF = 0;
                    unsigned long long mA = 0, mAF = 0, mF = 0, mB = 0;
#pragma unroll
                    for (int k = 0; k < QW; ++k) {
                        const int idx = (uu * QW + k) % RN;
                        const int kk = k + LA;
                        ring[(idx + LA) % RN] = read_x(kk % QW, uu + kk / QW);
                        mfma4_wgrad(acc[k], ring[idx], b);
                        // the dpre stream of step + PF: six small scalar pieces dealt out over the step's gaps
#pragma unroll
                        for (int pc = 0; pc < 6; ++pc) {
                            if ((pc * QW) / 6 != k) continue;
                            if (pc == 0) dpre_counts(dq, tw_u, DG.RS, nA, nAF);
                            if (pc == 1) mA = low_groups_mask(nA);
                            if (pc == 2) mAF = low_groups_mask(nAF);
                            if (pc == 3) dpre_masks(mA, mAF, mF, mB);
                            if (pc == 4) bq[(uu + PF) % 4] = dpre_fire(mF, mB, brs, voff_b, voff_bn, dq.soff);
                            if (pc == 5) dpre_next(dq, DG.RS, DG.padb, DG.stepb);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < QW; ++k)
#pragma unroll
                for (int g = 0; g < NXB; ++g) xw[k][g] += U * XSTEP;
        }
        u += th;
    }

    // MFMA results are read by VALU / stores next
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // ---- write this workgroup's partial
    float* pw = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        if (q >= Q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = 64 * q + 4 * (4 * kq + r) + g;
                // (PACK3: row R = (kh, kw * 3 + ci) -> the same HWIO index: ((kh KW + kw) 3 + ci) = R)
                const int tap = PACK3 ? R / 3 : R / CINP, ci = PACK3 ? R % 3 : R % CINP;
                if (R < ROWS && ci < a.Cin && co_ok) pw[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[k][g][r];
            }
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (qpart == 0 && kq == 0 && co_ok) pw[(size_t)TAPS * a.Cin * a.Cout + co] = bsum;
}

template <int KH, int KW, int CINP, int NCH, int MINW>
__global__ __launch_bounds__(256, MINW) void wgrad_lin_kernel(const WgradArgs a) {
    wgrad_lin_body<KH, KW, CINP, NCH, MINW, false>(a);
}
template <int KH, int KW, int CINP, int NCH, int MINW>
__global__ __launch_bounds__(256, MINW) void wgrad_lin_strip_kernel(const WgradArgs a) {
    wgrad_lin_body<KH, KW, CINP, NCH, MINW, true>(a);
}
// RGB-input layers (Cin = 3) with the packed row space, see wgrad_lin_body
template <int KH, int KW, int NCH, int MINW>
__global__ __launch_bounds__(256, MINW) void wgrad_lin_pack3_kernel(const WgradArgs a) {
    wgrad_lin_body<KH, KW, 4, NCH, MINW, false, true>(a);
}
// The filter gradients of a layer wider than 64 channels on channel-blocked tensors: every (input block, output block)
// pair is an independent 64 -> 64 problem of the same geometry; blockIdx.y picks the pair, the body never looks at it.
struct WgradPairs {
    WgradArgs a;                 // the pair (0, 0)
    long x_pair_stride;          // floats from one input block to the next
    long d_pair_stride;          // floats from one output block to the next
    long part_pair_stride;       // floats of partials per pair
    int cob;                     // output blocks: pair = ib * cob + ob
};
// STRIP: the column-strip body (rows too wide for one tile: the discriminator's 128 / 256-channel layers on 128- and 64-pixel rows)
template <int KH, int KW, int CINP, int NCH, int MINW, bool STRIP>
__global__ __launch_bounds__(256, MINW) void wgrad_lin_pairs_kernel(const WgradPairs q) {
    WgradArgs a = q.a;
    const int pair = blockIdx.y, ib = pair / q.cob, ob = pair - ib * q.cob;
    a.x += ib * q.x_pair_stride;
    a.dpre += ob * q.d_pair_stride;
    a.part += pair * q.part_pair_stride;
    wgrad_lin_body<KH, KW, CINP, NCH, MINW, STRIP>(a);
}


// ---------------------------------------------------------------------------------------------
// wgrad, linear walk, ONE workgroup per CU (the 64 -> 64 body layers).  Same K loop as wgrad_lin_kernel, but
// the tile is double-buffered in LDS and the NEXT tile's staging passes ride on the steps of the running
// one, as in conv_pipe_kernel; the dpre operands are fetched a whole window (16 steps) ahead.
//
// All loads of the step loop are issued from inline asm and waited for by hand (mixing them with
// compiler-visible loads would make the compiler's counted waits cover the youngest asm loads too).  An
// asm-issued load is asynchronous behind the compiler's back, so:
//   * its consumer sits in the same straight-line code (the unrolled window): no such value is live across
//     a branch or a loop back-edge, where the compiler could copy the register before the data has arrived
//     (at window / unit boundaries the values pass through an s_waitcnt statement that names them);
//   * it is parked in AGPRs: the 144 accumulators fill the VGPRs, and an "=a" / "=v" constraint only fixes
//     the register class AT the statement -- with spare VGPRs the compiler moved such a value to a VGPR
//     right after issue (seen in the smaller instances, which therefore stay on wgrad_lin_kernel);
//   * scripts/check_async_regs.py (make check) rejects any build in which an instruction touches such a
//     register between the load and its consumer.
// Distances: a staging load is written to LDS 8 steps (~9 k cycles, ~4 us) after its issue, a dpre value is
// used 8..23 steps after its issue -- activations come from HBM here, a batch of staging loads was measured
// to take ~2.5 us in wgrad_lin_kernel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 stage_fire_a(unsigned long long m, int so, __amdgpu_buffer_rsrc_t rsrc, int voff_lane) {
    f32x4 v;
    asm volatile("s_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_mov_b64 exec, -1"
                 : "=&a"(v) : "v"(voff_lane), "s"(rsrc), "s"(so), "s"(m) : "memory");
    return v;
}
#define SRX_COMMIT_ASM_A(N)                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_add_u32 %0, %3, %4\n\ts_mov_b64 exec, %2\n\tds_write_b128 %0, %1\n\ts_mov_b64 exec, -1" \
                 : "=&v"(addr) : "a"(v), "s"(m), "s"(q.off), "v"(wl_lane) : "memory")
__device__ __forceinline__ void stage_commit_a(int pend, const StageSeq& q, unsigned long long m, int wl_lane, const f32x4 v) {
    int addr;
    if (pend >= 14) SRX_COMMIT_ASM_A(14);
    else if (pend == 12) SRX_COMMIT_ASM_A(12);
    else if (pend == 10) SRX_COMMIT_ASM_A(10);
    else if (pend == 8) SRX_COMMIT_ASM_A(8);
    else if (pend == 6) SRX_COMMIT_ASM_A(6);
    else if (pend == 4) SRX_COMMIT_ASM_A(4);
    else if (pend == 2) SRX_COMMIT_ASM_A(2);
    else SRX_COMMIT_ASM_A(0);
}
// dpre load with a ready lane offset (wgrad_pipe_kernel looks it up in an LDS table keyed by the column of the step's
// first position)
__device__ __forceinline__ float dpre_fire_tab(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    float b;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&a"(b) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return b;
}

// the strip stager's load with an accumulation-register destination (see stage_fire_z)
__device__ __forceinline__ f32x4 stage_fire_za(unsigned long long m, unsigned long long r, int so, __amdgpu_buffer_rsrc_t rsrc,
                                               int voff_lane) {
    f32x4 v;
    unsigned long long z;
    const int oob = 0x7ff00000;
    asm volatile("s_andn2_b64 %1, %6, %5\n\t"
                 "s_mov_b64 exec, %5\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\t"
                 "s_mov_b64 exec, %1\n\tbuffer_load_dwordx4 %0, %2, %3, %7 offen\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&a"(v), "=&s"(z) : "v"(voff_lane), "s"(rsrc), "s"(so), "s"(m), "s"(r), "s"(oob) : "memory", "scc");
    return v;
}
// NT: the dpre stream is read exactly once, by exactly one wave -- marked non-temporal so that it does not push the x
// tile's halo rows (re-read by the same workgroup one tile later) out of the L2
template <bool NT>
__device__ __forceinline__ float dpre_fire_tab_nt(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    float b;
    if constexpr (NT)
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen nt" : "=&a"(b) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&a"(b) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return b;
}

template <int KH, int KW, int CINP, int NCH>
__global__ __launch_bounds__(256, 1) void wgrad_pipe_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TAPS = KH * KW;
    constexpr int ROWS = TAPS * CINP;
    constexpr int Q = (ROWS + 63) / 64;
    constexpr int NQP = 4 / NCH;
    constexpr int QW = (Q + NQP - 1) / NQP;
    constexpr int TPP = CINP / 4, PPP = 256 / TPP;
    constexpr int U = 14;                    // steps per unrolled window (the host picks TH so that a full unit is a whole number of windows)
    constexpr int NSW = U / 2;               // staging passes per full window: loads in its first half, LDS writes in its second
    constexpr int XSTEP = 4 * PS * 4;        // LDS bytes from one step to the next
    constexpr int LA = 2, RN = 3;            // LDS fragments in flight ahead of the MFMAs / fragment ring
    static_assert(QW * 16 >= 128, "AGPR parking assumes the accumulators fill the VGPRs (see above)");
    static_assert((U * QW) % RN == 0, "window must keep the fragment ring in phase");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, qpart = wave / NCH;
    const int cout0 = chunk * 16;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;
    const int co_c = co_ok ? co : a.Cout - 1;
    const int c4 = tid % TPP, sp = tid / TPP;
    char* ldsb = reinterpret_cast<char*>(lds);
    const int buf_bytes = (a.zero_slot + 4) * PS * 4;   // (the host allocates 4 slots past the largest tile, per buffer)

    // per-lane LDS byte address of (position kq of the running window's first step, the lane's tap / channels)
    // (absolute 32-bit LDS addresses, read through address_space(3) pointers: with generic pointers or offsets from
    // `lds` the compiler re-adds the LDS base in front of every read once the bases are opaque to it)
    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    const int lds_base = (int)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
    int xw[QW];
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        int R = 64 * q + 4 * li;
        if (q >= Q || R >= ROWS) R = 0;
        const int tap = R / CINP, ci = R % CINP;
        xw[k] = lds_base + ((kq + (tap / KW) * a.RS + (tap % KW)) * PS + ci) * 4;
    }
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 2 * buf_bytes / 16; i += 256) reinterpret_cast<f32x4*>(lds)[i] = z;
    }
    __syncthreads();
    // dpre lane-offset table, keyed by the column c0 of a step's first position, in the pad words of the first RS
    // slots of buffer 0 (staging writes only the data floats): entry (c0, kq) = the offset of lane group kq's position
    // c0 + kq -- in this row (biased by RS-OW pixels, see below), fake (out of range), or in the next row
    for (int i = tid; i < a.RS * 4; i += 256) {
        const int c0 = i >> 2, g = i & 3, col = c0 + g;
        const int v = col < a.OW ? (g * a.Cout + (a.RS - a.OW) * a.Cout) * 4 : (col < a.RS ? 0x7ff00000 : g * a.Cout * 4);
        *reinterpret_cast<int*>(ldsb + (c0 * PS + CINP + g) * 4) = v;
    }
    const int voff_lane = tid * 16;
    const int wl_lane = (sp * PS + 4 * c4) * 4;
    StageGeo SG;
    const int JP = (a.RS + PPP - 1) / PPP;
    SG.JP1 = __builtin_amdgcn_readfirstlane(JP - 1);
    SG.rowfix_g = __builtin_amdgcn_readfirstlane((a.W - JP * PPP) * CINP * 4);
    SG.rowfix_l = __builtin_amdgcn_readfirstlane((a.RS - JP * PPP) * PS * 4);
    SG.m_first = uniform64(__ballot(sp >= a.pad_l));
    SG.m_last = uniform64(__ballot((JP - 1) * PPP + sp < a.RS));
    DpreGeo DG;
    DG.tw = __builtin_amdgcn_readfirstlane(a.OW);
    DG.RS = __builtin_amdgcn_readfirstlane(a.RS);
    DG.padb = __builtin_amdgcn_readfirstlane((a.RS - a.OW) * a.Cout * 4);
    DG.stepb = __builtin_amdgcn_readfirstlane(16 * a.Cout);
    const int voff_bn = (kq * a.Cout + co_c) * 4;
    const int voff_b = voff_bn + (a.RS - a.OW) * a.Cout * 4;
    const int tbl_lane = lds_base + (CINP + kq) * 4;      // this lane's word of a table entry
    const int co4 = co_c * 4;

    f32x4 acc[QW][4];
#pragma unroll
    for (int k = 0; k < QW; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[k][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);

    auto tile_of = [&](int uu_, int& n, int& h, int& th) {
        h = uu_ % a.OH;
        n = uu_ / a.OH;
        th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - uu_ < th) th = u1 - uu_;
    };
    auto stage_setup = [&](StageSeq& qi, StageSeq& qc, int h, int th, int buf, bool active) {
        const int left = __builtin_amdgcn_readfirstlane(active ? (th + KH - 1) * JP : 0);
        const int above = (a.pad_t > h) ? (a.pad_t - h) * JP : 0;
        qi.j = 0; qc.j = 0;
        qi.off = __builtin_amdgcn_readfirstlane((h - a.pad_t) * a.W * CINP * 4);
        qc.off = __builtin_amdgcn_readfirstlane(buf * buf_bytes);
        qi.left = left; qc.left = left;
        qi.thr = __builtin_amdgcn_readfirstlane(left - above);
        qc.thr = 0;
    };
    auto x_rsrc = [&](int n) {
        return uniform_rsrc(a.x + ((size_t)n * a.H * a.W - a.pad_l) * CINP, (a.H * a.W + a.pad_l) * CINP * 4);
    };
    auto b_rsrc = [&](int n, int h, int th) {
        return uniform_rsrc(a.dpre + (((size_t)n * a.OH + h) * a.OW - (a.RS - a.OW)) * a.Cout,
                            (th * a.OW + (a.RS - a.OW)) * a.Cout * 4);
    };

    // dpre operands: bcur[] = the running window's (VGPRs), bnext[] = the next window's (in flight, AGPRs)
    float bcur[U], bnext[U];
    DpreSeq dq;
    // a unit's first window, all at once: ordinary (compiler-visible) loads -- these ARE in flight across the
    // staging drain loop and the barrier
    auto dpre_window_now = [&](__amdgpu_buffer_rsrc_t brs) {
        dq.c0 = __builtin_amdgcn_readfirstlane(0);
        dq.soff = __builtin_amdgcn_readfirstlane(0);
#pragma unroll
        for (int j = 0; j < U; ++j) bnext[j] = dpre_step_now(dq, DG, brs, voff_b, voff_bn);
    };
#define SRX_TAKE_OVER_B()                                                                                          \
    do {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(0) ; %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13"                          \
                     : "+a"(bnext[0]), "+a"(bnext[1]), "+a"(bnext[2]), "+a"(bnext[3]), "+a"(bnext[4]), "+a"(bnext[5]), \
                       "+a"(bnext[6]), "+a"(bnext[7]), "+a"(bnext[8]), "+a"(bnext[9]), "+a"(bnext[10]),            \
                       "+a"(bnext[11]), "+a"(bnext[12]), "+a"(bnext[13]));                                         \
        _Pragma("unroll") for (int j = 0; j < U; ++j)                                                              \
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(bcur[j]) : "a"(bnext[j]));   /* (here, not next to the MFMA that uses it) */ \
    } while (0)
    static_assert(U == 14, "SRX_TAKE_OVER_B names 14 registers");

    int u = u0;
    int n, h, th;
    tile_of(u, n, h, th);
    [[maybe_unused]] unsigned long long t_loop = 0, t_end = 0, t_full = 0;   // (trace builds)
    [[maybe_unused]] const unsigned long long t_begin = SRX_STAMP();
    __syncthreads();
    {
        StageSeq qi, qc;
        stage_setup(qi, qc, h, th, 0, true);
        stage_tile_scalar<CINP, 6>(qi, qc, SG, x_rsrc(n), voff_lane, wl_lane);
        dpre_window_now(b_rsrc(n, h, th));
        lds_barrier();
        SRX_TAKE_OVER_B();
    }
    int cur_buf = 0;
    while (u < u1) {
        tile_of(u, n, h, th);
        const int un_ = u + th;
        const bool has_next = un_ < u1;
        int n2 = n, h2 = h, th2 = th;
        if (has_next) tile_of(un_, n2, h2, th2);
        StageSeq qi, qc;
        stage_setup(qi, qc, h2, th2, cur_buf ^ 1, has_next);
        const __amdgpu_buffer_rsrc_t xrs = x_rsrc(n2);
        const __amdgpu_buffer_rsrc_t brs = b_rsrc(n, h, th);
        const __amdgpu_buffer_rsrc_t brs_next = b_rsrc(n2, h2, has_next ? th2 : 0);

        // Every window is a FULL window: the steps are padded to a multiple of U (their dpre operand is out of
        // range -> 0, their x operand whatever finite data lies behind the tile), and the host picks TH so that
        // a full unit needs no padding.  The step loop then has no branch but its back-edge: each branch -- taken
        // or not -- costs the MFMA stream 20-35 cycles (shadow3_ubench), six of them per step cost 13 %.
        const int nsteps = (((th * a.RS + 3) >> 2) + U - 1) / U * U;
        auto read_x = [&](int k, int uu) -> f32x4 { return *(lds_f32x4*)(uintptr_t)(unsigned)(xw[k] + uu * XSTEP); };
        f32x4 ring[RN];
#pragma unroll
        for (int f = 0; f < LA; ++f) ring[f] = read_x(f % QW, f / QW);

        const unsigned long long ts_l = SRX_STAMP();
        __amdgpu_buffer_rsrc_t brs_pf = brs;
        for (int s0 = 0; s0 < nsteps; s0 += U) {
            if (s0 + U >= nsteps) {
                // the unit's last window prefetches the NEXT unit's first dpre window (nothing real when there is none)
                brs_pf = brs_next;
                dq.c0 = __builtin_amdgcn_readfirstlane(0);
                dq.soff = __builtin_amdgcn_readfirstlane(0);
            }
            // Extra work of a window: the next window's dpre loads (two steps' worth in each of the first U/2
            // steps) and NSW staging passes of the next tile (loads in the first half, LDS writes in the second).
            // Every asm load has its consumer in the same straight-line code, and the number of memory operations
            // between the two, which the s_waitcnt of an LDS write relies on, is known.
            f32x4 stg[NSW];
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                const float b = bcur[uu];
                bsum += b;
                int tv = 0;
                unsigned long long smk = 0, smt = 0;
                int sso = 0;
#pragma unroll
                for (int k = 0; k < QW; ++k) {
                    const int idx = (uu * QW + k) % RN;
                    const int kk = k + LA;
                    ring[(idx + LA) % RN] = read_x(kk % QW, uu + kk / QW);
                    mfma4_wgrad(acc[k], ring[idx], b);
                    if (k == 0 && uu >= NSW) {
                        // LDS write of the pass issued NSW steps ago.  Younger memory operations that certainly
                        // count: the two dpre loads of each first-half step after it.
                        const unsigned long long m = stage_mask_full(qc, SG);
                        stage_commit_a(2 * (NSW - 1 - (uu - NSW)), qc, m, wl_lane, stg[uu - NSW]);
                    }
                    if (k == 4 && uu >= NSW) stage_next<PPP * PS * 4>(qc, SG.JP1, SG.rowfix_l);
                    if (k == 2 && uu == NSW) stage_next<PPP * CINP * 4>(qi, SG.JP1, SG.rowfix_g);      // (the cursor step of pass NSW-1)
                    if (uu < NSW) {
                        // dpre of steps 2uu, 2uu+1 of the next window: eight small pieces over the step's gaps
#pragma unroll
                        for (int pc = 0; pc < 8; ++pc) {
                            if (pc + (pc >= 4 ? 1 : 0) != k) continue;      // gaps 0..3 and 5..8
                            const int p4 = pc % 4;
                            if (p4 == 0) tv = *(__attribute__((address_space(3))) const int*)(uintptr_t)(unsigned)(tbl_lane + dq.c0 * (PS * 4));
                            if (p4 == 2) bnext[2 * uu + pc / 4] = dpre_fire_tab(brs_pf, tv + co4, dq.soff);
                            if (p4 == 3) dpre_next(dq, DG.RS, DG.padb, DG.stepb);
                        }
                        // one pass of the next tile, in pieces
                        if (k == 4) { stage_mask_a(qi, SG, smk, smt); stage_mask_b(qi, smk, smt, sso); }
                        if (k == 6) stg[uu] = stage_fire_a(smk, sso, xrs, voff_lane);
                        if (k == 2) { if (uu > 0) stage_next<PPP * CINP * 4>(qi, SG.JP1, SG.rowfix_g); }
                    }
                }
            }
            SRX_TAKE_OVER_B();
#pragma unroll
            for (int k = 0; k < QW; ++k) { xw[k] += U * XSTEP; SRX_PIN(xw[k]); }
        }
        const unsigned long long ts_e = SRX_STAMP();
        t_loop += ts_e - ts_l;
        {
            // back to step 0, in the other buffer
            const int nwin = nsteps / U;
            const int back = __builtin_amdgcn_readfirstlane((cur_buf ? -buf_bytes : buf_bytes) - nwin * U * XSTEP);
#pragma unroll
            for (int k = 0; k < QW; ++k) { xw[k] += back; SRX_PIN(xw[k]); }
        }
        stage_tile_scalar<CINP, 6>(qi, qc, SG, xrs, voff_lane, wl_lane);
        lds_barrier();
        cur_buf ^= 1;
        u = un_;
        t_end += SRX_STAMP() - ts_e;
    }
#undef SRX_TAKE_OVER_B
#ifdef SRX_TRACE
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 12;
        tr[0] = t_begin; tr[1] = SRX_STAMP(); tr[2] = t_loop; tr[3] = t_end; tr[4] = t_full;
    }
#endif

    // MFMA results are read by VALU / stores next
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    float* pw = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        if (q >= Q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = 64 * q + 4 * (4 * kq + r) + g;
                const int tap = R / CINP, ci = R % CINP;
                if (R < ROWS && ci < a.Cin && co_ok) pw[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[k][g][r];
            }
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (qpart == 0 && kq == 0 && co_ok) pw[(size_t)TAPS * a.Cin * a.Cout + co] = bsum;
}

// ---------------------------------------------------------------------------------------------
// wgrad on column strips, EXACT rows (round 4).  A strip is 32 output columns = 8 steps of 4 positions, so the K loop can
// walk the REAL pixels of a strip row and still have every address a compile-time constant: a window is 16 steps = two
// strip rows, step s of it reads the x operand at LDS slot (s / 8) * RS + 4 (s % 8) (+ tap) -- ds_read immediates -- and the
// dpre operand of pixel (row s / 8, column 4 (s % 8) + lane group): a per-lane offset that depends on s % 8 only (8
// registers, set up per unit) plus the row as the buffer load's scalar offset.  Against the padded walk of
// the padded-position walk of wgrad_lin_strip_kernel: no fake positions (34 -> 32 MFMA columns per row: -5.9 % MFMAs), no lane-offset table
// (2 VALU + 1 LDS read per step), no dpre cursor (7 SALU per step).  The narrower last strip of an image masks its
// missing columns in those 8 registers (out-of-range offset -> 0), so any width >= 1 is covered; rows past a short
// unit fall beyond the unit's buffer resource (-> 0).  Staging of the next tile, double buffering, AGPR parking and the
// hand-counted waits are those of wgrad_pipe_kernel.
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CINP, int NCH, bool NT>
__global__ __launch_bounds__(256, 1) void wgrad_rows_strip_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TAPS = KH * KW;
    constexpr int ROWS = TAPS * CINP;
    constexpr int Q = (ROWS + 63) / 64;
    constexpr int NQP = 4 / NCH;
    constexpr int QW = (Q + NQP - 1) / NQP;
    constexpr int TPP = CINP / 4, PPP = 256 / TPP;
    constexpr int TWZ = 32;                  // strip width
    constexpr int RSZ = TWZ + KW - 1;        // slots per tile row (the host plans exactly this)
    constexpr int SPR = TWZ / 4;             // steps per strip row
    constexpr int U = 2 * SPR;               // steps per unrolled window: two strip rows
    constexpr int NSW = U / 2;               // staging passes per window: loads in its first half, LDS writes in its second
    constexpr int LA = 2, RN = 3;            // LDS fragments in flight ahead of the MFMAs / fragment ring
    constexpr int WINB = 2 * RSZ * PS * 4;   // LDS bytes from one window to the next
    static_assert(QW * 16 >= 128, "AGPR parking assumes the accumulators fill the VGPRs (see wgrad_pipe_kernel)");
    static_assert((U * QW) % RN == 0, "window must keep the fragment ring in phase");
    static_assert(U == 16, "SRX_TAKE_OVER_B names the 16 registers of a window");
    static_assert(QW >= 9, "the dealt-out schedule of a step uses gaps 0..8");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, qpart = wave / NCH;
    const int cout0 = chunk * 16;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;
    const int co_c = co_ok ? co : a.Cout - 1;
    const int c4 = tid % TPP, sp = tid / TPP;
    char* ldsb = reinterpret_cast<char*>(lds);
    const int buf_bytes = (a.zero_slot + 4) * PS * 4;   // (the host allocates 4 slots past the largest tile, per buffer)

    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    const int lds_base = (int)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
    int xw[QW];      // LDS byte address of (position kq of the running window's first step, the lane's tap / channels)
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        int R = 64 * q + 4 * li;
        if (q >= Q || R >= ROWS) R = 0;
        const int tap = R / CINP, ci = R % CINP;
        xw[k] = lds_base + ((kq + (tap / KW) * RSZ + (tap % KW)) * PS + ci) * 4;
    }
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 2 * buf_bytes / 16; i += 256) reinterpret_cast<f32x4*>(lds)[i] = z;
    }
    const int voff_lane = tid * 16;
    const int wl_lane = (sp * PS + 4 * c4) * 4;
    StageGeo SG;
    constexpr int JP = (RSZ + PPP - 1) / PPP;
    SG.JP1 = __builtin_amdgcn_readfirstlane(JP - 1);
    SG.rowfix_g = __builtin_amdgcn_readfirstlane((a.W - JP * PPP) * CINP * 4);
    SG.rowfix_l = __builtin_amdgcn_readfirstlane((RSZ - JP * PPP) * PS * 4);
    SG.m_first = ~0ull; SG.m_last = ~0ull; SG.m_mid = ~0ull;     // (per tile, stage_setup)
    SG.m_row = uniform64(__ballot((JP - 1) * PPP + sp < RSZ));
    const int rowb = __builtin_amdgcn_readfirstlane(a.OW * a.Cout * 4);      // bytes of one image row of dpre
    const int lane_b = (kq * a.Cout + co_c) * 4;                            // lane part of a dpre offset, column 0 of a step
    const int colb = 4 * a.Cout * 4;                                        // bytes from one step's first column to the next

    f32x4 acc[QW][4];
#pragma unroll
    for (int k = 0; k < QW; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[k][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);

    auto tile_of = [&](int uu_, int& n, int& h, int& th, int& ow0, int& tw) {
        h = uu_ % a.OH;
        n = uu_ / a.OH;
        const int tx = n % a.NTX;
        n = n / a.NTX;
        ow0 = tx * TWZ;
        tw = a.OW - ow0 < TWZ ? a.OW - ow0 : TWZ;
        th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - uu_ < th) th = u1 - uu_;
    };
    auto stage_setup = [&](StageSeq& qi, StageSeq& qc, int h, int th, int ow0, int buf, bool active) {
        const int left = __builtin_amdgcn_readfirstlane(active ? (th + KH - 1) * JP : 0);
        const int above = (a.pad_t > h) ? (a.pad_t - h) * JP : 0;
        qi.j = 0; qc.j = 0;
        qi.off = __builtin_amdgcn_readfirstlane(((h - a.pad_t) * a.W + ow0) * CINP * 4);
        qc.off = __builtin_amdgcn_readfirstlane(buf * buf_bytes);
        // image column of tile slot c is ow0 - pad_l + c; a narrow last strip has invalid columns before the last pass too
        const int iw0 = ow0 - a.pad_l + sp;
        const unsigned long long v0 = __ballot(iw0 >= 0 && iw0 < a.W && sp < RSZ);
        const int cl = (JP - 1) * PPP + sp;
        const unsigned long long vl = __ballot(cl < RSZ && iw0 + (JP - 1) * PPP >= 0 && iw0 + (JP - 1) * PPP < a.W);
        SG.m_first = uniform64(JP == 1 ? (v0 & vl) : v0);
        SG.m_last = uniform64(vl);
        SG.m_mid = uniform64(__ballot(iw0 + PPP >= 0 && iw0 + PPP < a.W));
        qi.left = left; qc.left = left;
        qi.thr = __builtin_amdgcn_readfirstlane(left - above);
        qc.thr = 0;
    };
    auto x_rsrc = [&](int n) {
        return uniform_rsrc(a.x + ((size_t)n * a.H * a.W - a.pad_l) * CINP, (a.H * a.W + a.pad_l) * CINP * 4);
    };
    // the unit's pixels: th rows, the last one ending after its tw columns (th == 0: nothing)
    auto b_rsrc = [&](int n, int h, int th, int ow0, int tw) {
        return uniform_rsrc(a.dpre + (((size_t)n * a.OH + h) * a.OW + ow0) * a.Cout, th > 0 ? ((th - 1) * a.OW + tw) * a.Cout * 4 : 0);
    };
    // per-lane dpre offsets of the 8 steps of a strip row, for a strip of tw columns (columns >= tw: out of range -> 0)
    int vpf[SPR];
    auto set_vpf = [&](int tw) {
#pragma unroll
        for (int j = 0; j < SPR; ++j) vpf[j] = (4 * j + kq < tw) ? lane_b + j * colb : kOobOffset;
    };

    // dpre operands: bcur[] = the running window's (VGPRs), bnext[] = the next window's (in flight, AGPRs)
    float bcur[U], bnext[U];
#define SRX_TAKE_OVER_B()                                                                                          \
    do {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(0) ; %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14 %15"                  \
                     : "+a"(bnext[0]), "+a"(bnext[1]), "+a"(bnext[2]), "+a"(bnext[3]), "+a"(bnext[4]), "+a"(bnext[5]), \
                       "+a"(bnext[6]), "+a"(bnext[7]), "+a"(bnext[8]), "+a"(bnext[9]), "+a"(bnext[10]),            \
                       "+a"(bnext[11]), "+a"(bnext[12]), "+a"(bnext[13]), "+a"(bnext[14]), "+a"(bnext[15]));       \
        _Pragma("unroll") for (int j = 0; j < U; ++j)                                                              \
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(bcur[j]) : "a"(bnext[j]));   /* (here, not next to the MFMA that uses it) */ \
    } while (0)

    int u = u0;
    int n, h, th, ow0, tw;
    tile_of(u, n, h, th, ow0, tw);
    [[maybe_unused]] unsigned long long t_loop = 0, t_end = 0;   // (trace builds)
    [[maybe_unused]] const unsigned long long t_begin = SRX_STAMP();
    __syncthreads();
    {
        StageSeq qi, qc;
        stage_setup(qi, qc, h, th, ow0, 0, true);
        stage_tile_scalar<CINP, 6, true, true>(qi, qc, SG, x_rsrc(n), voff_lane, wl_lane);
        // the first unit's first window, all at once: ordinary (compiler-visible) loads
        set_vpf(tw);
        const __amdgpu_buffer_rsrc_t brs0 = b_rsrc(n, h, th, ow0, tw);
#pragma unroll
        for (int j = 0; j < U; ++j)
            bnext[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs0, vpf[j % SPR], (j / SPR) * rowb, 0));
        lds_barrier();
        SRX_TAKE_OVER_B();
    }
    int cur_buf = 0;
    while (u < u1) {
        tile_of(u, n, h, th, ow0, tw);
        const int un_ = u + th;
        const bool has_next = un_ < u1;
        int n2 = n, h2 = h, th2 = th, ow2 = ow0, tw2 = tw;
        if (has_next) tile_of(un_, n2, h2, th2, ow2, tw2);
        StageSeq qi, qc;
        stage_setup(qi, qc, h2, th2, ow2, cur_buf ^ 1, has_next);
        const __amdgpu_buffer_rsrc_t xrs = x_rsrc(n2);
        const __amdgpu_buffer_rsrc_t brs = b_rsrc(n, h, th, ow0, tw);
        const __amdgpu_buffer_rsrc_t brs_next = b_rsrc(n2, h2, has_next ? th2 : 0, ow2, tw2);

        const int nwin = (th + 1) >> 1;      // (an odd last row pair: its second row lies beyond the unit's resource -> 0)
        auto read_x = [&](int k, int uu) -> f32x4 {
            return *(lds_f32x4*)(uintptr_t)(unsigned)(xw[k] + ((uu / SPR) * RSZ + 4 * (uu % SPR)) * PS * 4);
        };
        f32x4 ring[RN];
#pragma unroll
        for (int f = 0; f < LA; ++f) ring[f] = read_x(f % QW, f / QW);

        const unsigned long long ts_l = SRX_STAMP();
        __amdgpu_buffer_rsrc_t brs_pf = brs;
        int spf0 = __builtin_amdgcn_readfirstlane(2 * rowb), spf1 = __builtin_amdgcn_readfirstlane(3 * rowb);   // rows of the window being prefetched
        for (int wi = 0; wi < nwin; ++wi) {
            if (wi + 1 == nwin) {
                // the unit's last window prefetches the NEXT unit's first dpre window (nothing real when there is none)
                brs_pf = brs_next;
                spf0 = __builtin_amdgcn_readfirstlane(0);
                spf1 = rowb;
                set_vpf(tw2);
            }
            f32x4 stg[NSW];
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                const float b = bcur[uu];
                bsum += b;
                unsigned long long smk = 0, smt = 0, smr = 0;
                int sso = 0;
#pragma unroll
                for (int k = 0; k < QW; ++k) {
                    const int idx = (uu * QW + k) % RN;
                    const int kk = k + LA;
                    ring[(idx + LA) % RN] = read_x(kk % QW, uu + kk / QW);
                    mfma4_wgrad(acc[k], ring[idx], b);
                    if (uu >= NSW) {
                        // LDS write of the pass issued NSW steps ago.  Younger memory operations that certainly
                        // count: the two dpre loads of each first-half step after it.
                        if (k == 0) stage_commit_a(2 * (NSW - 1 - (uu - NSW)), qc, stage_mask_row(qc, SG), wl_lane, stg[uu - NSW]);
                        if (k == 4) stage_next<PPP * PS * 4>(qc, SG.JP1, SG.rowfix_l);
                        if (k == 2 && uu == NSW) stage_next<PPP * CINP * 4>(qi, SG.JP1, SG.rowfix_g);      // (the cursor step of pass NSW-1)
                    } else {
                        // dpre of steps 2uu, 2uu+1 of the next window: row uu / 4 of its two rows
                        if (k == 1) bnext[2 * uu] = dpre_fire_tab_nt<NT>(brs_pf, vpf[(2 * uu) % SPR], (2 * uu) / SPR ? spf1 : spf0);
                        if (k == 7) bnext[2 * uu + 1] = dpre_fire_tab_nt<NT>(brs_pf, vpf[(2 * uu + 1) % SPR], (2 * uu + 1) / SPR ? spf1 : spf0);
                        // one pass of the next tile, in pieces
                        if (k == 2) { if (uu > 0) stage_next<PPP * CINP * 4>(qi, SG.JP1, SG.rowfix_g); }
                        if (k == 3) stage_mask_az<true>(qi, SG, smk, smt, smr);
                        if (k == 4) stage_mask_bz(qi, smk, smt, smr, sso);
                        if (k == 6) stg[uu] = stage_fire_za(smk, smr, sso, xrs, voff_lane);
                    }
                }
            }
            SRX_TAKE_OVER_B();
            spf0 += 2 * rowb; spf1 += 2 * rowb;
#pragma unroll
            for (int k = 0; k < QW; ++k) { xw[k] += WINB; SRX_PIN(xw[k]); }
        }
        const unsigned long long ts_e = SRX_STAMP();
        t_loop += ts_e - ts_l;
        {
            // back to the first window, in the other buffer
            const int back = __builtin_amdgcn_readfirstlane((cur_buf ? -buf_bytes : buf_bytes) - nwin * WINB);
#pragma unroll
            for (int k = 0; k < QW; ++k) { xw[k] += back; SRX_PIN(xw[k]); }
        }
        stage_tile_scalar<CINP, 6, true, true>(qi, qc, SG, xrs, voff_lane, wl_lane);
        lds_barrier();
        cur_buf ^= 1;
        u = un_;
        t_end += SRX_STAMP() - ts_e;
    }
#undef SRX_TAKE_OVER_B
#ifdef SRX_TRACE
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 12;
        tr[0] = t_begin; tr[1] = SRX_STAMP(); tr[2] = t_loop; tr[3] = t_end; tr[4] = 0;
    }
#endif

    // MFMA results are read by VALU / stores next
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    float* pw = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        if (q >= Q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = 64 * q + 4 * (4 * kq + r) + g;
                const int tap = R / CINP, ci = R % CINP;
                if (R < ROWS && ci < a.Cin && co_ok) pw[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[k][g][r];
            }
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (qpart == 0 && kq == 0 && co_ok) pw[(size_t)TAPS * a.Cin * a.Cout + co] = bsum;
}

// ---------------------------------------------------------------------------------------------
// wgrad on full-width tiles, EXACT rows, for image widths OWC = 4 SPR + 1 (41-pixel VDSR patches: SPR = 10).  The padded walk
// of wgrad_pipe_kernel pays, per 36-MFMA step, a lane-offset table lookup (2 VALU + 1 LDS read) and a dpre cursor (7 SALU), and
// one fake position in 42.  Here a unit of THC = 3 rows is ONE straight-line window of 3 SPR + 1 = 31 steps over real pixels:
//   * THC x SPR "row steps": step (row r, j) takes columns 4 j .. 4 j + 3 of row r -- x fragments at ds_read immediates
//     (r RS + 4 j), the dpre operand at a per-lane base (3 registers) + immediate + the row as scalar offset;
//   * ONE "column step": the lane groups take the LAST column of rows 0 .. 3 (lane group kq: row kq; row 3 lies beyond a
//     3-row unit's buffer resource and reads 0) -- a second per-lane base for the x fragments (+ kq (RS - 1) slots) and one
//     lane offset for dpre.
// Every address of the window is a compile-time constant; rows past a short unit lie beyond the unit's resource (-> 0), so
// there is one body and no branch but the unit loop's.  (Units of 4 rows -- 41 steps, no idle lane group in the column step --
// leave a ONE-row unit at the end of a 41-row image: as a full window that costs 451 steps per image against 434 here and
// now; as a short second body, chosen before or inside the window, it made hipcc copy in-flight AGPRs at the merge /
// spill the 41 dpre operands -- both built, both rejected by `make check`.)  41 rows = 13 units of 3 + one of 2: 14 x 31 =
// 434 steps, the padded walk's count; the gain is the per-step work.  Staging (15 passes: loads in steps 0..14, LDS writes
// in steps 16..30), double buffering, AGPR parking and the hand-counted waits as in wgrad_pipe_kernel.
// ---------------------------------------------------------------------------------------------
template <int IMM>
__device__ __forceinline__ float dpre_fire_imm(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    float b;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=&a"(b) : "v"(voff), "s"(rsrc), "s"(soff), "n"(IMM) : "memory");
    return b;
}

template <int KH, int KW, int CINP, int NCH, int OWC>
__global__ __launch_bounds__(256, 1) void wgrad_rows_full_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TAPS = KH * KW;
    constexpr int ROWS = TAPS * CINP;
    constexpr int Q = (ROWS + 63) / 64;
    constexpr int NQP = 4 / NCH;
    constexpr int QW = (Q + NQP - 1) / NQP;
    constexpr int TPP = CINP / 4, PPP = 256 / TPP;
    constexpr int SPR = OWC / 4;             // whole steps per image row
    constexpr int THC = 3;                   // rows per unit
    constexpr int RSZ = OWC + 1;             // slots per tile row (one pad column: the host sends SAME 3x3 layers only)
    constexpr int COUTC = 64;                // output channels (the dpre loads' immediates are multiples of 16 COUTC bytes; host-checked)
    constexpr int U = THC * SPR + 1;         // steps per window
    constexpr int JP = (RSZ + PPP - 1) / PPP;
    constexpr int NPASSES = (THC + KH - 1) * JP;   // staging passes of a full tile
    constexpr int C0 = (U + 1) / 2;          // first step with an LDS write of the staging (all loads are out by then)
    constexpr int LA = 2, RN = 3;
    static_assert(OWC % 4 == 1, "one leftover column per row: the column step takes it for up to four rows at once");
    static_assert(QW * 16 >= 128 && QW >= 9, "accumulators fill the VGPRs; the dealt-out schedule of a step uses gaps 0..8");
    static_assert(2 * NPASSES <= U && NPASSES < C0 && C0 + NPASSES <= U, "staging loads in the first half (beside two dpre loads each), LDS writes in the second");
    static_assert(U == 31, "SRX_TAKE_OVER_B names the 31 registers of a window");
    static_assert((((THC + 1) * RSZ + 4 * SPR) * PS * 4) < 65536, "ds_read immediates");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, qpart = wave / NCH;
    const int cout0 = chunk * 16;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;
    const int co_c = co_ok ? co : a.Cout - 1;
    const int c4 = tid % TPP, sp = tid / TPP;
    char* ldsb = reinterpret_cast<char*>(lds);
    const int buf_bytes = (a.zero_slot + 4) * PS * 4;    // (the host allocates a tile of THC + 3 rows: the column step's idle lane group reads row 3 + KH - 1)

    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    const int lds_base = (int)(uintptr_t)(__attribute__((address_space(3))) char*)ldsb;
    int xw[QW];      // LDS byte address of (tile slot kq, the lane's tap / channels) in the running buffer
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        int R = 64 * q + 4 * li;
        if (q >= Q || R >= ROWS) R = 0;
        const int tap = R / CINP, ci = R % CINP;
        xw[k] = lds_base + ((kq + (tap / KW) * RSZ + (tap % KW)) * PS + ci) * 4;
    }
    const int dX = kq * (RSZ - 1) * PS * 4;       // column step: lane group kq reads row kq instead of column kq
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 2 * buf_bytes / 16; i += 256) reinterpret_cast<f32x4*>(lds)[i] = z;
    }
    const int voff_lane = tid * 16;
    const int wl_lane = (sp * PS + 4 * c4) * 4;
    StageGeo SG;
    SG.JP1 = __builtin_amdgcn_readfirstlane(JP - 1);
    SG.rowfix_g = __builtin_amdgcn_readfirstlane((a.W - JP * PPP) * CINP * 4);
    SG.rowfix_l = __builtin_amdgcn_readfirstlane((RSZ - JP * PPP) * PS * 4);
    SG.m_first = uniform64(__ballot(sp >= a.pad_l));
    SG.m_last = uniform64(__ballot((JP - 1) * PPP + sp < RSZ));
    SG.m_row = SG.m_last; SG.m_mid = ~0ull;
    // dpre addressing: a unit's pixels are contiguous (full-width rows); lane part of a row step = column 4 j + kq
    const int colb = 16 * COUTC;                                        // bytes from one step's first column to the next
    const int lane_b = (kq * COUTC + co_c) * 4;
    const int vb0 = lane_b, vb1 = lane_b + 4 * colb, vb2 = lane_b + 8 * colb;      // + (j % 4) colb as the immediate (< 4096)
    const int vX = ((kq * OWC + OWC - 1) * COUTC + co_c) * 4;           // column step: last column of row kq
    const int rowb = __builtin_amdgcn_readfirstlane(OWC * COUTC * 4);
    const int srow1 = rowb, srow2 = __builtin_amdgcn_readfirstlane(2 * rowb);

    f32x4 acc[QW][4];
#pragma unroll
    for (int k = 0; k < QW; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[k][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);

    auto tile_of = [&](int uu_, int& n, int& h, int& th) {
        h = uu_ % a.OH;
        n = uu_ / a.OH;
        th = THC;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - uu_ < th) th = u1 - uu_;
    };
    auto stage_setup = [&](StageSeq& qi, StageSeq& qc, int h, int th, int buf, bool active) {
        const int left = __builtin_amdgcn_readfirstlane(active ? (th + KH - 1) * JP : 0);
        const int above = (a.pad_t > h) ? (a.pad_t - h) * JP : 0;
        qi.j = 0; qc.j = 0;
        qi.off = __builtin_amdgcn_readfirstlane((h - a.pad_t) * a.W * CINP * 4);
        qc.off = __builtin_amdgcn_readfirstlane(buf * buf_bytes);
        qi.left = left; qc.left = left;
        qi.thr = __builtin_amdgcn_readfirstlane(left - above);
        qc.thr = 0;
    };
    auto x_rsrc = [&](int n) {
        return uniform_rsrc(a.x + ((size_t)n * a.H * a.W - a.pad_l) * CINP, (a.H * a.W + a.pad_l) * CINP * 4);
    };
    auto b_rsrc = [&](int n, int h, int th) {
        return uniform_rsrc(a.dpre + ((size_t)n * a.OH + h) * a.OW * COUTC, th * a.OW * COUTC * 4);
    };

    float bcur[U], bnext[U];
    // a unit's first dpre window all at once: ordinary (compiler-visible) loads.  Window position i: rows first (i / SPR,
    // i % SPR), the column step last.
    auto dpre_window_now = [&](__amdgpu_buffer_rsrc_t brs) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (i == U - 1) {
                bnext[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, vX, 0, 0));
            } else {
                const int r = i / SPR, j = i % SPR;
                const int vb = j < 4 ? vb0 : (j < 8 ? vb1 : vb2);
                bnext[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, vb + (j % 4) * colb, r * rowb, 0));
            }
        }
    };
    // (inline asm takes at most 30 operands: the wait names the registers in two statements)
#define SRX_TAKE_OVER_B()                                                                                          \
    do {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(0) ; %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14 %15"                  \
                     : "+a"(bnext[0]), "+a"(bnext[1]), "+a"(bnext[2]), "+a"(bnext[3]), "+a"(bnext[4]), "+a"(bnext[5]), \
                       "+a"(bnext[6]), "+a"(bnext[7]), "+a"(bnext[8]), "+a"(bnext[9]), "+a"(bnext[10]),            \
                       "+a"(bnext[11]), "+a"(bnext[12]), "+a"(bnext[13]), "+a"(bnext[14]), "+a"(bnext[15]));       \
        asm volatile("; %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14"                                         \
                     : "+a"(bnext[16]), "+a"(bnext[17]), "+a"(bnext[18]), "+a"(bnext[19]), "+a"(bnext[20]),        \
                       "+a"(bnext[21]), "+a"(bnext[22]), "+a"(bnext[23]), "+a"(bnext[24]), "+a"(bnext[25]),        \
                       "+a"(bnext[26]), "+a"(bnext[27]), "+a"(bnext[28]), "+a"(bnext[29]), "+a"(bnext[30]));       \
        _Pragma("unroll") for (int j = 0; j < U; ++j)                                                              \
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(bcur[j]) : "a"(bnext[j]));                             \
    } while (0)

    int u = u0;
    int n, h, th;
    tile_of(u, n, h, th);
    [[maybe_unused]] unsigned long long t_loop = 0, t_end = 0;   // (trace builds)
    [[maybe_unused]] const unsigned long long t_begin = SRX_STAMP();
    __syncthreads();
    {
        StageSeq qi, qc;
        stage_setup(qi, qc, h, th, 0, true);
        stage_tile_scalar<CINP, 6>(qi, qc, SG, x_rsrc(n), voff_lane, wl_lane);
        dpre_window_now(b_rsrc(n, h, th));
        lds_barrier();
        SRX_TAKE_OVER_B();
    }
    int cur_buf = 0;
    while (u < u1) {
        tile_of(u, n, h, th);
        const int un_ = u + th;
        const bool has_next = un_ < u1;
        int n2 = n, h2 = h, th2 = th;
        if (has_next) tile_of(un_, n2, h2, th2);
        StageSeq qi, qc;
        stage_setup(qi, qc, h2, th2, cur_buf ^ 1, has_next);
        const __amdgpu_buffer_rsrc_t xrs = x_rsrc(n2);
        const __amdgpu_buffer_rsrc_t brs_pf = b_rsrc(n2, h2, has_next ? th2 : 0);

        // x fragment of window position i (i == U: the position after the window -- prefetched by the ring, never used)
        auto xaddr = [&](int k, int i) -> unsigned {
            if (i == U - 1) return (unsigned)(xw[k] + dX + (OWC - 1) * PS * 4);
            const int r = i >= U ? THC : i / SPR, j = i >= U ? 0 : i % SPR;
            return (unsigned)(xw[k] + (r * RSZ + 4 * j) * PS * 4);
        };
        auto read_x = [&](int k, int i) -> f32x4 { return *(lds_f32x4*)(uintptr_t)xaddr(k, i); };
        f32x4 ring[RN];
#pragma unroll
        for (int f = 0; f < LA; ++f) ring[f] = read_x(f % QW, f / QW);

        const unsigned long long ts_l = SRX_STAMP();
        f32x4 stg[NPASSES];
        // one step of the window (position i a compile-time constant: hipcc does not unroll a 31 x 9 loop nest of this size by
        // itself): 9 fragments x 4 MFMAs, and the step's share of the extra work dealt out over the gaps
        auto step = [&](auto IC) {
            constexpr int i = decltype(IC)::value;
            const float b = bcur[i];
            bsum += b;
            unsigned long long smk = 0, smt = 0;
            int sso = 0;
#pragma unroll
            for (int k = 0; k < QW; ++k) {
                const int idx = (i * QW + k) % RN;
                const int kk = k + LA;
                ring[(idx + LA) % RN] = read_x(kk % QW, i + kk / QW);
                mfma4_wgrad(acc[k], ring[idx], b);
                // dpre of the NEXT unit's window positions 2 i (gap 1) and 2 i + 1 (gap 7)
                if ((k == 1 && 2 * i < U) || (k == 7 && 2 * i + 1 < U)) {
                    const int p = 2 * i + (k == 7 ? 1 : 0);
                    if (p == U - 1) {
                        bnext[p] = dpre_fire_imm<0>(brs_pf, vX, 0);
                    } else {
                        const int r = p / SPR, j = p % SPR;
                        const int vb = j < 4 ? vb0 : (j < 8 ? vb1 : vb2);
                        const int so = r == 0 ? 0 : (r == 1 ? srow1 : srow2);
                        switch (j % 4) {
                            case 0: bnext[p] = dpre_fire_imm<0>(brs_pf, vb, so); break;
                            case 1: bnext[p] = dpre_fire_imm<16 * COUTC>(brs_pf, vb, so); break;
                            case 2: bnext[p] = dpre_fire_imm<32 * COUTC>(brs_pf, vb, so); break;
                            default: bnext[p] = dpre_fire_imm<48 * COUTC>(brs_pf, vb, so); break;
                        }
                    }
                }
                // one staging pass of the next tile: load in steps 0 .. NPASSES-1, LDS write in steps C0 .. C0+NPASSES-1
                if constexpr (i < NPASSES) {
                    if (k == 2) { if (i > 0) stage_next<PPP * CINP * 4>(qi, SG.JP1, SG.rowfix_g); }
                    if (k == 3) stage_mask_a(qi, SG, smk, smt);
                    if (k == 4) stage_mask_b(qi, smk, smt, sso);
                    if (k == 6) stg[i < NPASSES ? i : 0] = stage_fire_a(smk, sso, xrs, voff_lane);
                }
                if constexpr (i == NPASSES) { if (k == 2) stage_next<PPP * CINP * 4>(qi, SG.JP1, SG.rowfix_g); }     // (the cursor step of the last pass)
                if constexpr (i >= C0 && i < C0 + NPASSES) {
                    constexpr int jj = i - C0;
                    // memory operations certainly younger than the pass's load: its step's second dpre load, three per later
                    // staging step, the dpre loads issued after the last staging step
                    constexpr int after = 1 + 3 * (NPASSES - 1 - jj) + (U - 2 * NPASSES);
                    if (k == 0) stage_commit_a(after >= 14 ? 14 : (after & ~1), qc, stage_mask_full(qc, SG), wl_lane, stg[jj < NPASSES ? jj : 0]);
                    if (k == 4) stage_next<PPP * PS * 4>(qc, SG.JP1, SG.rowfix_l);
                }
            }
        };
#define SRX_ONE_STEP(I) step(std::integral_constant<int, I>{});
        SRX_ONE_STEP(0) SRX_ONE_STEP(1) SRX_ONE_STEP(2) SRX_ONE_STEP(3) SRX_ONE_STEP(4) SRX_ONE_STEP(5) SRX_ONE_STEP(6) SRX_ONE_STEP(7)
        SRX_ONE_STEP(8) SRX_ONE_STEP(9) SRX_ONE_STEP(10) SRX_ONE_STEP(11) SRX_ONE_STEP(12) SRX_ONE_STEP(13) SRX_ONE_STEP(14) SRX_ONE_STEP(15)
        SRX_ONE_STEP(16) SRX_ONE_STEP(17) SRX_ONE_STEP(18) SRX_ONE_STEP(19) SRX_ONE_STEP(20) SRX_ONE_STEP(21) SRX_ONE_STEP(22) SRX_ONE_STEP(23)
        SRX_ONE_STEP(24) SRX_ONE_STEP(25) SRX_ONE_STEP(26) SRX_ONE_STEP(27) SRX_ONE_STEP(28) SRX_ONE_STEP(29) SRX_ONE_STEP(30)
#undef SRX_ONE_STEP
        SRX_TAKE_OVER_B();
        const unsigned long long ts_e = SRX_STAMP();
        t_loop += ts_e - ts_l;
        {
            const int flip = __builtin_amdgcn_readfirstlane(cur_buf ? -buf_bytes : buf_bytes);
#pragma unroll
            for (int k = 0; k < QW; ++k) { xw[k] += flip; SRX_PIN(xw[k]); }
        }
        stage_tile_scalar<CINP, 6>(qi, qc, SG, xrs, voff_lane, wl_lane);
        lds_barrier();
        cur_buf ^= 1;
        u = un_;
        t_end += SRX_STAMP() - ts_e;
    }
#undef SRX_TAKE_OVER_B
#ifdef SRX_TRACE
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * 4 + wave) * 12;
        tr[0] = t_begin; tr[1] = SRX_STAMP(); tr[2] = t_loop; tr[3] = t_end; tr[4] = 0;
    }
#endif

    // MFMA results are read by VALU / stores next
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    float* pw = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        if (q >= Q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int R = 64 * q + 4 * (4 * kq + r) + g;
                const int tap = R / CINP, ci = R % CINP;
                if (R < ROWS && ci < a.Cin && co_ok) pw[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[k][g][r];
            }
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (qpart == 0 && kq == 0 && co_ok) pw[(size_t)TAPS * a.Cin * a.Cout + co] = bsum;
}

}  // namespace srx
