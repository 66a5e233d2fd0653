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
};

struct WgradArgs {
    const float* x;     // layer input [N,H,W,Cin]
    const float* dpre;  // gradient wrt pre-activation output [N,OH,OW,Cout]
    float* part_dw;     // [G][KH*KW*Cin*Cout]
    float* part_db;     // [G][Cout]
    int N, H, W, OH, OW, Cin, Cout;
    int pad_t, pad_l;
    int TH, TW, NTX, RS;
    int units_total;
    float inv_rs;
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

// Stage one tile's input halo into LDS.  Slot s <-> (r = s / RS, c = s % RS) <-> input pixel
// (h_in0 + r, w_in0 + c); out-of-image slots are written as zeros (TF SAME zero padding).
template <int CINP>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ x, int n, int H, int W,
                                           int Cin, int h_in0, int w_in0, int RS, float inv_rs,
                                           int n_need, int tid) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TPP = CINP / 4;    // threads per pixel (each moves 16 B)
    constexpr int PPP = 256 / TPP;   // pixels per pass
    const int c4 = tid % TPP;
    const int sp = tid / TPP;
    const bool vec = (Cin & 3) == 0;
    const float* xn = x + (size_t)n * H * W * Cin;
    for (int s0 = 0; s0 < n_need; s0 += 4 * PPP) {
        f32x4 v[4];
        int sl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s0 + j * PPP + sp;
            sl[j] = s;
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s < n_need) {
                const int r = fdiv_small(s, inv_rs, RS);
                const int c = s - r * RS;
                const int ih = h_in0 + r, iw = w_in0 + c;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
                    const float* p = xn + ((size_t)ih * W + iw) * Cin + 4 * c4;
                    if (vec) {
                        if (4 * c4 < Cin) v[j] = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * c4 + e < Cin) v[j][e] = p[e];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (sl[j] < n_need) *reinterpret_cast<f32x4*>(lds + (size_t)sl[j] * PS + 4 * c4) = v[j];
    }
}

// Epilogue shared by all forward / dgrad variants: lane holds channels cb..cb+3 of pixel (orow, ocol).
template <int G>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[G], const bool (&valid)[G], const int (&orow)[G],
                                              const int (&ocol)[G], const ConvArgs& a, int n, int h, int ow0, int kq,
                                              int cout0) {
    // ---- epilogue: lane holds channels cb..cb+3 of pixel (orow, ocol)
    const int cb = cout0 + 4 * kq;
    const bool vec = ((a.Cout & 3) == 0) && (cb + 3 < a.Cout);
#pragma unroll
    for (int i = 0; i < G; ++i) {
        if (!valid[i] || cb >= a.Cout) continue;
        const size_t off = (((size_t)n * a.OH + h + orow[i]) * a.OW + ow0 + ocol[i]) * a.Cout + cb;
        f32x4 v = acc[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], a.act);
        if (vec) {
            if (a.skip) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(a.skip + off);
                v += s;
            }
            if (a.post_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (a.mask) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(a.mask + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_grad_from_y(m[e], a.mask_act);
            }
            *reinterpret_cast<f32x4*>(a.y + off) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (cb + e < a.Cout) {
                    float t = v[e];
                    if (a.skip) t += a.skip[off + e];
                    if (a.post_relu) t = fmaxf(t, 0.f);
                    if (a.mask) t *= act_grad_from_y(a.mask[off + e], a.mask_act);
                    a.y[off + e] = t;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CINP, int G>
__device__ __forceinline__ void conv_group(const float* lds, const float (&wr)[KH * KW * (CINP / 4)],
                                           const f32x4 bias4, const ConvArgs& a, int n, int h, int ow0,
                                           int th, int tw, float inv_tw, int m_first, int m_step,
                                           int li, int kq, int cout0) {
    constexpr int PS = Lds<CINP>::PS;
    constexpr int NG = (CINP >= 16) ? CINP / 16 : 1;
    const int npx = th * tw;
    int laddr[G];   // float index of this lane's pixel (tap 0,0) in LDS
    int orow[G], ocol[G];
    bool valid[G];
    f32x4 acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m_first + i * m_step) + li;
        valid[i] = t < npx;
        const int tt = valid[i] ? t : 0;
        orow[i] = fdiv_small(tt, inv_tw, tw);
        ocol[i] = tt - orow[i] * tw;
        laddr[i] = (orow[i] * a.RS + ocol[i]) * PS + ((CINP >= 16) ? 4 * kq : kq);
        acc[i] = bias4;
    }
    const int row_stride = a.RS * PS;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            constexpr int dummy = 0;
            (void)dummy;
            const int tap = kh * KW + kw;
            if constexpr (CINP >= 16) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    f32x4 av[G];
#pragma unroll
                    for (int i = 0; i < G; ++i)
                        av[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i] + kh * row_stride + kw * PS + 16 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
#pragma unroll
                        for (int i = 0; i < G; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[tap * (CINP / 4) + 4 * g + s], av[i][s],
                                                                          acc[i], 0, 0, 0);
                    }
                }
            } else {
                float av[G];
#pragma unroll
                for (int i = 0; i < G; ++i) av[i] = lds[laddr[i] + kh * row_stride + kw * PS];
#pragma unroll
                for (int i = 0; i < G; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[tap], av[i], acc[i], 0, 0, 0);
            }
        }
    }
    conv_epilogue<G>(acc, valid, orow, ocol, a, n, h, ow0, kq, cout0);
}

// KH,KW: filter; CINP: padded input channels held per LDS pixel (4, 32 or 64); NCH: number of
// 16-wide output-channel chunks (1, 2 or 4 -> 4/NCH waves share the pixels of a chunk);
// WT: read the filters transposed + flipped (dgrad); MINW: waves per SIMD for launch bounds.
template <int KH, int KW, int CINP, int NCH, bool WT, int MINW>
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

    // ---- stationary weights: wr[tap][j], k index of lane = channel ci(j, kq)
    float wr[TAPS * KSPT];
    {
        const int co = cout0 + li;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
            for (int j = 0; j < KSPT; ++j) {
                const int ci = (CINP >= 16) ? 16 * (j / 4) + 4 * kq + (j % 4) : kq;
                float v = 0.f;
                if (ci < a.Cin && co < a.Cout) {
                    if (!WT)
                        v = a.w[((size_t)tap * a.Cin + ci) * a.Cout + co];
                    else
                        v = a.w[((size_t)(TAPS - 1 - tap) * a.Cout + co) * a.Cin + ci];
                }
                wr[tap * KSPT + j] = v;
            }
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

        __syncthreads();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        __syncthreads();

        const int n_sub = (th * tw + 15) >> 4;
        const int cnt = (n_sub - part + NPART - 1) / NPART;  // subtiles of this wave
        const float inv_tw = 1.0f / (float)tw;
        if (cnt > 0) {
            const int ng = (cnt + 3) >> 2;
            const int base = cnt / ng, rem = cnt % ng;
            int idx = 0;
            for (int gi = 0; gi < ng; ++gi) {
                const int gs = base + (gi < rem ? 1 : 0);
                const int m_first = part + idx * NPART;
                switch (gs) {
                    case 4: conv_group<KH, KW, CINP, 4>(lds, wr, bias4, a, n, h, ow0, th, tw, inv_tw, m_first, NPART, li, kq, cout0); break;
                    case 3: conv_group<KH, KW, CINP, 3>(lds, wr, bias4, a, n, h, ow0, th, tw, inv_tw, m_first, NPART, li, kq, cout0); break;
                    case 2: conv_group<KH, KW, CINP, 2>(lds, wr, bias4, a, n, h, ow0, th, tw, inv_tw, m_first, NPART, li, kq, cout0); break;
                    default: conv_group<KH, KW, CINP, 1>(lds, wr, bias4, a, n, h, ow0, th, tw, inv_tw, m_first, NPART, li, kq, cout0); break;
                }
                idx += gs;
            }
        }
        u += th;
    }
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
    const int npx = th * tw;
    const int TAPS = KH * KW;
    int laddr[G], orow[G], ocol[G];
    bool valid[G];
    f32x4 acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m_first + i * m_step) + li;
        valid[i] = t < npx;
        const int tt = valid[i] ? t : 0;
        orow[i] = fdiv_small(tt, inv_tw, tw);
        ocol[i] = tt - orow[i] * tw;
        laddr[i] = (orow[i] * a.RS + ocol[i]) * PS + ((CINP >= 16) ? 4 * kq : kq);
        acc[i] = bias4;
    }
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
    conv_epilogue<G>(acc, valid, orow, ocol, a, n, h, ow0, kq, cout0);
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
        __syncthreads();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        __syncthreads();
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
template <int KH, int KW, int CINP, int NCH, int MINW>
__global__ __launch_bounds__(256, MINW) void wgrad_mfma_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = Lds<CINP>::PS;
    constexpr int TAPS = KH * KW;
    constexpr int ROWS = TAPS * CINP;
    constexpr int Q = (ROWS + 63) / 64;
    constexpr int NQP = 4 / NCH;             // waves sharing a cout chunk split the q's
    constexpr int QW = (Q + NQP - 1) / NQP;  // q's per wave
    constexpr int PD = (QW >= 4) ? 2 : 8;    // dpre prefetch depth (steps)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int chunk = wave % NCH, qpart = wave / NCH;
    const int cout0 = chunk * 16;
    const int co = cout0 + li;
    const bool co_ok = co < a.Cout;

    // per-lane LDS offset (floats) of the tap/channel this lane feeds for each of its q's
    int toff[QW];
    bool qok[QW];
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = qpart + k * NQP;
        int R = 64 * q + 4 * li;  // first of the lane's 4 rows (all 4 share tap since CINP % 4 == 0)
        qok[k] = (q < Q) && (R < ROWS);
        if (!qok[k]) R = 0;
        const int tap = R / CINP, ci = R % CINP;
        const int kh = tap / KW, kw = tap % KW;
        toff[k] = (kh * a.RS + kw) * PS + ci;
    }

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
        const int n_need = (th + KH - 1) * a.RS + (KW - 1);

        __syncthreads();
        stage_tile<CINP>(lds, a.x, n, a.H, a.W, a.Cin, h - a.pad_t, ow0 - a.pad_l, a.RS, a.inv_rs, n_need, tid);
        __syncthreads();

        const int npx = th * tw;
        const int nsteps = (npx + 3) >> 2;
        const float inv_tw = 1.0f / (float)tw;
        const float* dbase = a.dpre + (((size_t)n * a.OH + h) * a.OW + ow0) * a.Cout + co;

        auto load_b = [&](int s) -> float {
            const int p = 4 * s + kq;
            float v = 0.f;
            if (p < npx && co_ok) {
                const int r = fdiv_small(p, inv_tw, tw);
                const int c = p - r * tw;
                v = dbase[((size_t)r * a.OW + c) * a.Cout];
            }
            return v;
        };
        auto x_addr = [&](int s) -> int {
            int p = 4 * s + kq;
            if (p >= npx) p = npx - 1;  // finite data, multiplied by b = 0
            const int r = fdiv_small(p, inv_tw, tw);
            const int c = p - r * tw;
            return (r * a.RS + c) * PS;
        };

        float bq[PD];
#pragma unroll
        for (int j = 0; j < PD; ++j) bq[j] = load_b(j);

        for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const int s = s0 + j;
                const float b = bq[j];
                bq[j] = load_b(s + PD);
                if (s < nsteps) {
                    bsum += b;
                    const int xa = x_addr(s);
#pragma unroll
                    for (int k = 0; k < QW; ++k) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(lds + xa + toff[k]);
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[k][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g], b, acc[k][g], 0, 0, 0);
                    }
                }
            }
        }
        u += th;
    }

    // ---- write this workgroup's partial
    float* pw = a.part_dw + (size_t)blockIdx.x * TAPS * a.Cin * a.Cout;
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
    if (a.part_db) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (qpart == 0 && kq == 0 && co_ok) a.part_db[(size_t)blockIdx.x * a.Cout + co] = bsum;
    }
}

}  // namespace srx
