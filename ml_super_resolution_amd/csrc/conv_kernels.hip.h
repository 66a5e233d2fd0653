// conv_kernels.hip.h -- gfx950 (MI355X) implicit-GEMM convolution kernels, exact fp32 on
// v_mfma_f32_16x16x4_f32.
//
// Design:
//  * NHWC activations, HWIO filters, stride 1.  One persistent workgroup (4 waves) owns a
//    contiguous range of output rows ("units"); two workgroups are resident per CU (<= 80 KiB
//    LDS, <= 256 VGPRs each) so one stages its next tile while the other issues MFMAs.
//  * A tile's input halo (TH+KH-1 rows) is staged ONCE into LDS with a padded pixel stride
//    (Cin+4 floats) and explicit zero padding, then read KH*KW times; tap offsets become
//    ds_read immediates.
//  * forward / dgrad: the layer's weights for a wave's 16 output channels are STATIONARY IN
//    REGISTERS for the whole kernel (3x3x64 -> 144 VGPRs); MFMA A = weights (rows = Cout),
//    B = pixels (cols), so each lane ends up holding 4 consecutive output channels of one pixel
//    -> one 16-byte store.  Bias is the accumulator's initial value; activation / residual add /
//    upstream activation-gradient mask are fused in the epilogue.
//  * wgrad: the dW accumulators (144 VGPRs for 3x3x64 per 16 Cout) are stationary for the whole
//    kernel; x comes from the same LDS halo tile, dpre straight from global (each element is used
//    by exactly one wave).  Per-workgroup partials are reduced by a second, fixed-order kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    int dbg;              // diagnostic timing knobs (SRX_DBG): 1 = stage only the first tile, 2 = no stores
    unsigned long long* trace;  // diagnostic build (-DSRX_TRACE) only: per-wave cycle stamps
    int buf_floats;       // pipelined kernel: floats per LDS tile buffer (two buffers)
    int* tile_counter;    // dynamic scheduling (two-workgroup kernels): next tile to hand out, preset to gridDim.x; null = static
    int tiles_total, tiles_per_col;   // N*NTX*tiles_per_col tiles of TH rows (the last of a column may be shorter)
    int lds_sched_slot;   // float index in LDS of the 4-byte mailbox used to broadcast the tile index
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
};

template <int CINP>
struct Lds {
    static constexpr int PS = (CINP == 4) ? 4 : CINP + 4;  // pixel stride in floats
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_TANH: return tanhf(v);
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
__device__ __attribute__((noinline)) f32x4 act_transcendental4(f32x4 v, int act) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (act == ACT_TANH) ? tanhf(v[e]) : 1.0f / (1.0f + __expf(-v[e]));
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
                if (a.post_relu) v = act_apply4(v, ACT_RELU, 0.0f);
                if (a.mask) v = act_grad4(v, aux[i], a.mask_act, mslope);
            }
            if ((a.dbg & 2) && v[0] != 12345.678f) continue;
            if (valid[i]) *reinterpret_cast<f32x4*>(yb + off[i]) = v;
        }
    } else {
        // ragged channel counts (Cout = 3, 27, ...): scalar tail
        const float* sk = (AUX && a.skip) ? a.skip + img_base : nullptr;
        const float* mk = (AUX && a.mask) ? a.mask + img_base : nullptr;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (!valid[i]) continue;
            const f32x4 v = act_apply4(acc[i], a.act, slope);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (cb + e < a.Cout) {
                    float t = v[e];
                    if (sk) t += sk[off[i] + e];
                    if (AUX && a.post_relu) t = fmaxf(t, 0.f);
                    if (mk) t *= act_grad_from_y(mk[off[i] + e], a.mask_act);
                    yb[off[i] + e] = t;
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
    const bool vec = ((a.Cout & 3) == 0) && (cb + 3 < a.Cout);
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
        laddr[i] = (orow * a.RS + ocol) * PS + ((CINP >= 16) ? 4 * kq : kq);
        off[i] = valid[i] ? (unsigned)(((h + orow) * a.OW + ow0 + ocol) * a.Cout + cb) : 0u;
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
    unsigned long long t_mfma = 0, t_stage = 0, t_bar1 = 0, t_load = 0, t_pro = 0, t_epi = 0;
    const unsigned long long t_begin = SRX_STAMP();
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
        const int n_need = (th + KH - 1) * a.RS + (KW - 1);
        const bool first_tile = dyn ? (tile == (int)blockIdx.x) : (u == u0);

        const unsigned long long ts_stage = SRX_STAMP();
        lds_barrier();
        const unsigned long long ts_b1 = SRX_STAMP();
        if (dyn && tid == 0) *mailbox = atomicAdd(a.tile_counter, 1);     // the NEXT tile, fetched early
        if (!(a.dbg & 1) || first_tile)
            stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs, n_need, tid);
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
// One-wave-per-SIMD pipelined forward / dgrad (the main path for >= 16 input channels).
//
// Measured on gfx950 (scripts/coissue_ubench.hip, scripts/shadow_ubench.hip): a wave that streams
// v_mfma_f32_16x16x4_f32 back to back starves the OTHER wave on its SIMD completely (VALU, LDS and
// VMEM probes all finish only after the MFMA wave does; s_setprio does not help), so two resident
// workgroups cannot hide each other's staging.  Within ONE wave, integer VALU / memory instructions
// issued between its own MFMAs are almost free.  Hence: one 4-wave workgroup per CU, the input tile
// double-buffered in LDS (2 x 80 KiB), and the NEXT tile's staging (address math, global loads, LDS
// writes) threaded through the MFMA blocks of the CURRENT tile.
// ---------------------------------------------------------------------------------------------
// Staging cursor of one thread over the NEXT tile's slots s = sp, sp+PPP, sp+2*PPP, ...  Everything is
// incremental 32-bit integer arithmetic (integer VALU is what hides in an fp32-MFMA shadow), and the
// load is a bounds-checked buffer load: an out-of-image slot is given an out-of-range offset and the
// hardware returns zeros, so the zero padding needs no select and no branch.
template <int CINP>
struct StageCtx {
    __amdgpu_buffer_rsrc_t rsrc;  // the next tile's image, H*W*Cin*4 bytes
    int H, W, RS, n_need;
    int h_in0, w_in0;
    int step_bytes, wrap_bytes;   // PPP*Cin*4 ; (W-RS)*Cin*4
    int pass, n_pass;             // wave-uniform: next pass to issue, number of passes the tile needs
    int s, r, c;                  // next slot of this thread and its (row, col) in the tile
    int voff;                     // byte offset of that slot's 16 B in the image (may be negative: padding)
    int loff;                     // float index of that slot's 16 B in the LDS buffer
    bool active;                  // there is a next tile and its channel count allows 16-B loads
    bool ch_ok;                   // this thread's 4 channels exist (else it writes the zero padding up to CINP)
};

template <int CINP>
__device__ __forceinline__ void stage_begin(StageCtx<CINP>& sc, const float* xn, int H, int W, int Cin, int h_in0,
                                            int w_in0, int RS, float inv_rs, int n_need, int sp, int c4, bool active) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int PPP = 256 / (CINP / 4);
    sc.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, H * W * Cin * 4, 0x00020000);
    sc.H = H; sc.W = W; sc.RS = RS; sc.n_need = n_need; sc.h_in0 = h_in0; sc.w_in0 = w_in0;
    sc.step_bytes = PPP * Cin * 4;
    sc.wrap_bytes = (W - RS) * Cin * 4;
    sc.pass = 0;
    sc.n_pass = active ? (n_need + PPP - 1) / PPP : 0;
    sc.s = sp;
    sc.r = fdiv_small(sp, inv_rs, RS);
    sc.c = sp - sc.r * RS;
    sc.voff = (((h_in0 + sc.r) * W + w_in0 + sc.c) * Cin + 4 * c4) * 4;
    sc.loff = sp * PS + 4 * c4;
    sc.active = active;
    sc.ch_ok = 4 * c4 < Cin;
}

// issue the load of the cursor's slot
template <int CINP>
__device__ __forceinline__ f32x4 stage_issue(const StageCtx<CINP>& sc) {
    const int ih = sc.h_in0 + sc.r, iw = sc.w_in0 + sc.c;
    const bool ok = ((unsigned)ih < (unsigned)sc.H) & ((unsigned)iw < (unsigned)sc.W) & (sc.s < sc.n_need) & sc.active & sc.ch_ok;
    const int off = ok ? sc.voff : 0x7fffffff;   // beyond num_records -> the load returns 0
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sc.rsrc, off, 0, 0));
}

// write a previously loaded slot (the one `back` passes behind the cursor) to the LDS buffer, then nothing else
template <int CINP>
__device__ __forceinline__ void stage_commit(const StageCtx<CINP>& sc, float* lnxt, int back, const f32x4 v) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int PPP = 256 / (CINP / 4);
    const int s = sc.s - back * PPP;
    if (sc.active && s < sc.n_need) *reinterpret_cast<f32x4*>(lnxt + sc.loff - back * PPP * PS) = v;
}

template <int CINP>
__device__ __forceinline__ void stage_advance(StageCtx<CINP>& sc) {
    // one wrap at most: the kernel enables shadow staging only when RS >= PPP
    constexpr int PS = Lds<CINP>::PS;
    constexpr int PPP = 256 / (CINP / 4);
    sc.pass += 1;
    sc.s += PPP; sc.loff += PPP * PS;
    const int c1 = sc.c + PPP;
    const bool wrap = c1 >= sc.RS;
    sc.c = wrap ? c1 - sc.RS : c1;
    sc.r += wrap ? 1 : 0;
    sc.voff += sc.step_bytes + (wrap ? sc.wrap_bytes : 0);
}

// single MFMA statements for the pipelined kernel: weights from AGPRs; "memory" keeps the staging
// loads / LDS accesses that are threaded between them in place
#ifdef SRX_EXP_NOMEM
#define SRX_MEMCLOB
#else
#define SRX_MEMCLOB : "memory"
#endif
__device__ __forceinline__ void mfma1_a(f32x4& c, float w, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(b) SRX_MEMCLOB);
}
__device__ __forceinline__ void mfma1_a_guard(f32x4& c, float w, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "a"(w), "v"(b) : "memory");
}
// zero-cost ordering fence: code that uses x afterwards cannot be scheduled above this point, code that
// produced x cannot sink below it (volatile asm statements keep their relative order)
#define SRX_PIN(x) asm volatile("" : "+v"(x))

template <int KH, int KW, int CINP, int G, bool AUX>
__device__ __forceinline__ void conv_group_pipe(const float* lds, float* lnxt, const float (&wr)[KH * KW * (CINP / 4)],
                                                const f32x4 bias4, const ConvArgs& a, StageCtx<CINP>& sc, int n, int h,
                                                int ow0, int th, int tw, float inv_tw, int m_first, int m_step,
                                                int n_active, int li, int kq, int cout0, int sp, int c4,
                                                unsigned long long (&tt)[4]) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int NG = CINP / 16;
    constexpr int NBLK = KH * KW * NG;
    const unsigned long long ts_p = SRX_STAMP();
#ifdef SRX_EXP_NOST
    constexpr int NST = 0;
#else
    constexpr int NST = (NBLK / 2 < 6) ? NBLK / 2 : 6;   // staging passes threaded through this group
#endif
    const int npx = th * tw;
    const int cb = cout0 + 4 * kq;
    const bool vec = ((a.Cout & 3) == 0) && (cb + 3 < a.Cout);
    const size_t img_base = (size_t)n * a.OH * a.OW * a.Cout;
    int laddr[G];
    unsigned off[G];
    bool valid[G];
    f32x4 acc[G];
    // pixel (row, col) of this lane in the group's first sub-tile by one division; the following sub-tiles
    // (stride 16*m_step pixels) by integer add-and-wrap when a single wrap suffices
    const int t0 = 16 * m_first + li;
    int orow = fdiv_small(t0 < npx ? t0 : 0, inv_tw, tw);
    int ocol = (t0 < npx ? t0 : 0) - orow * tw;
    const int dstep = 16 * m_step;
    const bool inc_ok = dstep <= tw;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = t0 + i * dstep;
        const bool live = (t < npx) & (i < n_active);
        valid[i] = live & (cb < a.Cout);
        if (i > 0) {
            if (inc_ok) {
                const int c1 = ocol + dstep;
                const bool wrap = c1 >= tw;
                ocol = wrap ? c1 - tw : c1;
                orow += wrap ? 1 : 0;
            } else {
                const int tt = t < npx ? t : 0;
                orow = fdiv_small(tt, inv_tw, tw);
                ocol = tt - orow * tw;
            }
        }
        const int prow = live ? orow : 0, pcol = live ? ocol : 0;
        laddr[i] = (prow * a.RS + pcol) * PS + 4 * kq;
        off[i] = valid[i] ? (unsigned)(((h + prow) * a.OW + ow0 + pcol) * a.Cout + cb) : 0u;
        acc[i] = bias4;
    }
    f32x4 aux[G];
    conv_prefetch_aux<G, AUX>(aux, off, a, img_base, vec);
    const int row_stride = a.RS * PS;

    f32x4 stg[NST > 0 ? NST : 1];
    f32x4 cur[G], nxt[G];
    const int pass_base = sc.pass;
    const unsigned long long ts_m = SRX_STAMP();
    tt[0] += ts_m - ts_p;
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i]);
    // Per block (one tap x 16 input channels) ONE asm statement of 4G MFMAs: every extra instruction in
    // this stream costs MFMA issue time (about 3 cycles each, measured), and separate statements make the
    // compiler re-emit an s_waitcnt per statement.  Between blocks: the LDS reads of the next block and,
    // in blocks 0..NST-1 / NBLK/2..NBLK/2+NST-1, one lean staging slot (issue / LDS write) for the next tile.
#pragma unroll
    for (int t = 0; t < NBLK; ++t) {
        if (t + 1 < NBLK) {
            const int t1 = t + 1;
            const int kh1 = (t1 / NG) / KW, kw1 = (t1 / NG) % KW, g1 = t1 % NG;
#pragma unroll
            for (int i = 0; i < G; ++i)
                nxt[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i] + kh1 * row_stride + kw1 * PS + 16 * g1);
        }
        // (wave-uniform tests: passes beyond the tile's last one cost nothing)
        if (NST > 0 && t < NST && pass_base + t < sc.n_pass) {
            stg[t % (NST > 0 ? NST : 1)] = stage_issue<CINP>(sc);
            stage_advance<CINP>(sc);
        }
        if (NST > 0 && t >= NBLK / 2 && t < NBLK / 2 + NST) {
            const int j = t - NBLK / 2;
            if (pass_base + j < sc.n_pass)
                stage_commit<CINP>(sc, lnxt, sc.pass - (pass_base + j), stg[j % (NST > 0 ? NST : 1)]);
        }
        const int tap = t / NG, g = t % NG;
        const int wb = tap * (CINP / 4) + 4 * g;
        if (t == 0)
            mfma_block_a<true>(acc, wr[wb], wr[wb + 1], wr[wb + 2], wr[wb + 3], cur);
        else
            mfma_block_a<false>(acc, wr[wb], wr[wb + 1], wr[wb + 2], wr[wb + 3], cur);
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    const unsigned long long ts_e = SRX_STAMP();
    tt[1] += ts_e - ts_m;
    conv_epilogue<G, AUX>(acc, aux, valid, off, a, img_base, cb, vec);
    tt[2] += SRX_STAMP() - ts_e;
}

template <int KH, int KW, int CINP, int NCH, bool WT, bool AUX>
__global__ __launch_bounds__(256, 1) void conv_pipe_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TAPS = KH * KW;
    constexpr int KSPT = CINP / 4;
    constexpr int NPART = 4 / NCH;
    constexpr int TPP = CINP / 4, PPP = 256 / TPP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, part = wave / NCH;
    const int cout0 = chunk * 16;
    const int c4 = tid % TPP, sp = tid / TPP;

#ifdef SRX_TRACE
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    float wr[TAPS * KSPT];
    {
        load_stationary_weights<TAPS, CINP, WT>(wr, a, cout0, li, kq);
        // all loads are in flight; now move the weights into the accumulation-register file for good: they
        // are defined as "a" values here and only ever consumed by "a" operands of the MFMA blocks.
        // (One asm per weight right after its own load would serialise 144 global-load round trips.)
#pragma unroll
        for (int i = 0; i < TAPS * KSPT; ++i) {
            float t = wr[i];
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(wr[i]) : "v"(t));
        }
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cout0 + 4 * kq + e < a.Cout) bias4[e] = a.bias[cout0 + 4 * kq + e];
    }

    const long G_ = gridDim.x;
    const int u0 = (int)(((long)blockIdx.x * a.units_total) / G_);
    const int u1 = (int)(((long)(blockIdx.x + 1) * a.units_total) / G_);
    if (u0 >= u1) return;
    const int buf_floats = a.buf_floats;
    const bool vec_in = (a.Cin & 3) == 0;

    // tile descriptor of unit u (wave-uniform)
    auto tile_of = [&](int u, int& n, int& h, int& th, int& ow0, int& tw) {
        h = u % a.OH;
        const int t = u / a.OH;
        const int tx = t % a.NTX;
        n = t / a.NTX;
        th = a.TH;
        if (a.OH - h < th) th = a.OH - h;
        if (u1 - u < th) th = u1 - u;
        ow0 = tx * a.TW;
        tw = (a.OW - ow0 < a.TW) ? (a.OW - ow0) : a.TW;
    };

    unsigned long long tt[4] = {0, 0, 0, 0};   // trace: group prologue, MFMA+shadow section, epilogue, drain+barrier
    int n, h, th, ow0, tw;
    tile_of(u0, n, h, th, ow0, tw);
    const unsigned long long t_begin = SRX_STAMP();
    stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs,
                     (th + KH - 1) * a.RS + (KW - 1), tid);
    lds_barrier();
    const unsigned long long t_first = SRX_STAMP();

    int cur = 0;
    int u = u0;
    while (u < u1) {
        tile_of(u, n, h, th, ow0, tw);
        const int un = u + th;
        const bool has_next = un < u1;
        int n2 = n, h2 = h, th2 = th, ow02 = ow0, tw2 = tw;
        if (has_next) tile_of(un, n2, h2, th2, ow02, tw2);
        StageCtx<CINP> sc;
        stage_begin<CINP>(sc, a.x + (size_t)n2 * a.H * a.W * a.Cin, a.H, a.W, a.Cin, h2 - a.pad_t, ow02 - a.pad_l, a.RS,
                          a.inv_rs, (th2 + KH - 1) * a.RS + (KW - 1), sp, c4, has_next && vec_in && a.RS >= PPP && !(a.dbg & 4));
        const float* lcur = lds + cur * buf_floats;
        float* lnxt = lds + (cur ^ 1) * buf_floats;

        const int n_sub = (th * tw + 15) >> 4;
        const int cnt = (n_sub - part + NPART - 1) / NPART;
        const float inv_tw = 1.0f / (float)tw;
        if (cnt > 0) {
            constexpr int MAXG = AUX ? 3 : 4;
            const int ng = (cnt + MAXG - 1) / MAXG;
            const int base = cnt / ng, rem = cnt % ng;
            int idx = 0;
            for (int gi = 0; gi < ng; ++gi) {
                const int gs = base + (gi < rem ? 1 : 0);
                const int m_first = part + idx * NPART;
                if (gs == MAXG)
                    conv_group_pipe<KH, KW, CINP, MAXG, AUX>(lcur, lnxt, wr, bias4, a, sc, n, h, ow0, th, tw, inv_tw, m_first,
                                                             NPART, gs, li, kq, cout0, sp, c4, tt);
                else
                    conv_group_pipe<KH, KW, CINP, MAXG - 1, AUX>(lcur, lnxt, wr, bias4, a, sc, n, h, ow0, th, tw, inv_tw,
                                                                 m_first, NPART, gs, li, kq, cout0, sp, c4, tt);
                idx += gs;
            }
        }
        // drain: whatever part of the next tile the groups did not cover (short tiles, ragged channels)
        const unsigned long long ts_d = SRX_STAMP();
        if (has_next) {
            if (vec_in && a.RS >= PPP && !(a.dbg & 4)) {
                while (sc.pass < sc.n_pass) {
                    const f32x4 v = stage_issue<CINP>(sc);
                    stage_advance<CINP>(sc);
                    stage_commit<CINP>(sc, lnxt, 1, v);
                }
            } else {
                stage_tile<CINP>(lnxt, a.x, n2, a.H, a.W, a.Cin, h2 - a.pad_t, ow02 - a.pad_l, a.RS, a.inv_rs, sc.n_need, tid);
            }
        }
        lds_barrier();
        tt[3] += SRX_STAMP() - ts_d;
        cur ^= 1;
        u = un;
    }
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
    const bool vec = ((a.Cout & 3) == 0) && (cb + 3 < a.Cout);
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
        laddr[i] = (orow * a.RS + ocol) * PS + ((CINP >= 16) ? 4 * kq : kq);
        off[i] = valid[i] ? (unsigned)(((h + orow) * a.OW + ow0 + ocol) * a.Cout + cb) : 0u;
        acc[i] = bias4;
    }
    f32x4 aux[G];
    conv_prefetch_aux<G, true>(aux, off, a, img_base, vec);
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
        const int n_need = (th + KH - 1) * a.RS + (KW - 1);
        lds_barrier();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs, n_need, tid);
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
        const int n_need = (th + KH - 1) * a.RS + (KW - 1);

        lds_barrier();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        lds_barrier();

        const int npx = th * tw;
        const int nsteps = (npx + 3) >> 2;
        const float* dbase = a.dpre + (((size_t)n * a.OH + h) * a.OW + ow0) * a.Cout + co_c;  // + 32-bit offsets
        const int x_step = 4 * PS, x_wrap = (a.RS - tw) * PS;
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
            cur.xaddr = (r0 * a.RS + cur.c) * PS; cur.boff = (r0 * a.OW + cur.c) * a.Cout;
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

}  // namespace srx
