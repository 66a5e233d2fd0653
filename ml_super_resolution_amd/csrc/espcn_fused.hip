// espcn_fused.hip -- ESPCN inference in ONE launch (BASELINE configs[1]: 3x, batch 32 of 17x17 LR patches):
//   t1 = tanh(conv5x5(x; 3 -> 64) + b1), t2 = tanh(conv3x3(t1; 64 -> 32) + b2), y = conv3x3(t2; 32 -> 3 r^2) + b3,
//   hr[n, h r + dy, w r + dx, c] = y[n, h, w, (dy r + dx) 3 + c]
// (espcn/espcn/model_espcn.py:117-134 + espcn/espcn/experiment_test.py:171-177).
//
// At that size the three layers are 0.57 GFLOP and three launches are three latency-bound ramps (9 + 14 + 9 us of
// kernel time plus the gaps between dependent launches).  Here a workgroup owns a tile of <= 16x16 LR output pixels and
// chains the layers through LDS: the input halo (tile grown by 4), t1 on the tile grown by 2 (<= 20x20 x 64 channels),
// t2 on the tile grown by 1 (<= 18x18 x 32), then the output straight through the sub-pixel map.  The host cuts the
// image into the tile shape that needs the fewest rounds of workgroups and the least MFMA work per tile ([32,17,17]:
// 2 x 3 tiles of 9 x 6 = 192 workgroups; round 2 used 2 x 2 of 9 x 9 = 128: 29.5 -> 22.7 us).  Positions of t1 / t2 outside the image
// are stored as zeros (SAME padding pads the LAYER INPUT).  Halo pixels are computed by every tile that needs them
// (f1 x2.1, f2 x1.5 at 9x9 tiles): worth it only while the problem is latency-bound -- the host uses this kernel for
// small problems and the three-launch path otherwise.  All three filter slices of a wave (25 + 144 + 72 registers) are
// loaded once per workgroup, exact fp32 on v_mfma_f32_16x16x4_f32; same products and order of accumulation as the
// per-layer kernels, so the result is bit-identical to them.
#include <stdarg.h>

#include "../../include/srx.h"
#include "launchers.h"

namespace srx {
int set_error(int code, const char* fmt, ...);

namespace {

struct EspcnArgs {
    const float *x, *w1, *b1, *w2, *b2, *w3, *b3;
    float* hr;
    float *t1k, *t2k, *yk;       // KEEP (training forward): the activations t1 [N,H,W,64], t2 [N,H,W,32] and y [N,H,W,C3] in sub-pixel space
    int N, H, W, r, C3;          // C3 = 3 r^2
    int TY, TX;                  // tile (<= 16 x 16 LR output pixels)
    int tiles_y, tiles_x, units;
};

constexpr int kT = 16;                      // largest tile edge (round 4: 9 -> 16, see srx_espcn_forward)
constexpr int kP1 = 68, kP2 = 36;           // LDS pixel strides of t1 (64 + 4) and t2 (32 + 4) in floats
constexpr int kX0 = (kT + 8) * (kT + 8) * 4;            // input halo, 4 floats per pixel
constexpr int kT1 = (kT + 4) * (kT + 4) * kP1;
constexpr int kT2 = (kT + 2) * (kT + 2) * kP2;
// LDS: t1, then ONE region shared by the input halo and t2 -- the halo is dead once f1 has run (barrier), t2 is dead when the
// next tile's halo is staged (barrier at the top of the tile loop): 108.8 + 46.7 KB at 16 x 16 tiles (of 160)
constexpr int kShared = kT2 > kX0 ? kT2 : kX0;
static_assert((size_t)(kT1 + kShared) * 4 <= 160 * 1024, "tile does not fit the LDS");

__device__ __forceinline__ f32x4 tanh4(f32x4 v) { return srx_tanhf4(v); }

// A wave's sub-tiles of a phase are m = first, first + step, ... (count of them); they are processed in groups of G <= 4
// accumulators, the groups evened out (9 sub-tiles: 3 + 3 + 3, not 4 + 4 + 1 with three idle accumulators' worth of
// MFMAs): the group bodies below are templates on G, picked by a wave-uniform switch outside the tap loops.
#define SRX_ESPCN_GROUPS(BODY)                                                                                 \
    {                                                                                                          \
        const int ng_ = (count + 3) >> 2;                                                                      \
        const int base_ = ng_ ? count / ng_ : 0, rem_ = ng_ ? count % ng_ : 0;                                 \
        int m_ = first;                                                                                        \
        for (int gi_ = 0; gi_ < ng_; ++gi_) {                                                                  \
            const int g_ = base_ + (gi_ < rem_ ? 1 : 0);                                                       \
            switch (g_) {                                                                                      \
                case 1: BODY(1) break;                                                                         \
                case 2: BODY(2) break;                                                                         \
                case 3: BODY(3) break;                                                                         \
                default: BODY(4) break;                                                                        \
            }                                                                                                  \
            m_ += g_ * step;                                                                                   \
        }                                                                                                      \
    }

// ---- f1: 5x5, 3 -> 64, tanh, on the tile grown by 2
template <int G, bool KEEP>
__device__ __forceinline__ void espcn_f1_group(const EspcnArgs& a, const float* X0, float* T1, const float (&w1r)[25], f32x4 b1r,
                                               int m0, int step, int n1, int w0, int w1, int oy, int ox, int wave, int li, int kq,
                                               float* t1_img, int th, int tw) {
    int la[G];
    f32x4 acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m0 + i * step) + li;
        const int tt = t < n1 ? t : 0;
        const int r = tt / w1, c = tt - r * w1;
        la[i] = (r * w0 + c) * 4 + kq;
        acc[i] = b1r;
    }
    // (the LDS reads of tap t + 1 are issued before the MFMAs of tap t: left to itself the compiler reads each value right
    // before its MFMA and waits out the LDS latency 25 times per group)
    float b[G], bn[G];
#pragma unroll
    for (int i = 0; i < G; ++i) b[i] = X0[la[i]];
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        if (t + 1 < 25) {
            const int kh1 = (t + 1) / 5, kw1 = (t + 1) % 5;
#pragma unroll
            for (int i = 0; i < G; ++i) bn[i] = X0[la[i] + (kh1 * w0 + kw1) * 4];
        }
#pragma unroll
        for (int i = 0; i < G; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[t], b[i], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < G; ++i) b[i] = bn[i];
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m0 + i * step) + li;
        if (t < n1) {
            const int r = t / w1, c = t - r * w1;
            const bool in_img = (unsigned)(oy - 2 + r) < (unsigned)a.H && (unsigned)(ox - 2 + c) < (unsigned)a.W;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v = in_img ? tanh4(acc[i]) : z;
            *reinterpret_cast<f32x4*>(T1 + t * kP1 + 16 * wave + 4 * kq) = v;
            // KEEP: the tile's own pixels (not the halo other tiles compute as well) also go to global memory, once
            if (KEEP && (unsigned)(r - 2) < (unsigned)th && (unsigned)(c - 2) < (unsigned)tw)
                *reinterpret_cast<f32x4*>(t1_img + ((size_t)(oy - 2 + r) * a.W + (ox - 2 + c)) * 64 + 16 * wave + 4 * kq) = v;
        }
    }
}

// ---- f2: 3x3, 64 -> 32, tanh, on the tile grown by 1
template <int G, bool KEEP>
__device__ __forceinline__ void espcn_f2_group(const EspcnArgs& a, const float* T1, float* T2, const float (&w2r)[144], f32x4 b2r,
                                               int m0, int step, int n2, int w1, int w2, int oy, int ox, int ch2, int li, int kq,
                                               float* t2_img, int th, int tw) {
    int la[G];
    f32x4 acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m0 + i * step) + li;
        const int tt = t < n2 ? t : 0;
        const int r = tt / w2, c = tt - r * w2;
        la[i] = (r * w1 + c) * kP1 + 4 * kq;
    }
    // Fragments of block t + 1 are read before the MFMAs of block t.  Groups of 3 / 4 sub-tiles issue their MFMAs as asm
    // statements (weights in accumulation registers, the order of LDS reads and MFMAs pinned); the rare groups of 1 / 2
    // (a tile of <= 2 sub-tiles per wave) go through the builtin.
    f32x4 b[G], bn[G];
#pragma unroll
    for (int i = 0; i < G; ++i) b[i] = *reinterpret_cast<const f32x4*>(T1 + la[i]);
#pragma unroll
    for (int t = 0; t < 36; ++t) {
        if (t + 1 < 36) {
            const int tap1 = (t + 1) / 4, g1 = (t + 1) % 4;
#pragma unroll
            for (int i = 0; i < G; ++i)
                bn[i] = *reinterpret_cast<const f32x4*>(T1 + la[i] + ((tap1 / 3) * w1 + (tap1 % 3)) * kP1 + 16 * g1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (G >= 3) {
                if (t == 0 && e == 0) mfma_sub_a_first(acc, b2r, w2r[0], b[0][0], b[1][0], b[2][0], b[G - 1][0]);
                else mfma_sub_a(acc, w2r[4 * t + e], b[0][e], b[1][e], b[2][e], b[G - 1][e]);
            } else {
#pragma unroll
                for (int i = 0; i < G; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[4 * t + e], b[i][e], (t == 0 && e == 0) ? b2r : acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) b[i] = bn[i];
    }
    if constexpr (G == 4) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    else if constexpr (G == 3) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m0 + i * step) + li;
        if (t < n2) {
            const int r = t / w2, c = t - r * w2;
            const bool in_img = (unsigned)(oy - 1 + r) < (unsigned)a.H && (unsigned)(ox - 1 + c) < (unsigned)a.W;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v = in_img ? tanh4(acc[i]) : z;
            *reinterpret_cast<f32x4*>(T2 + t * kP2 + 16 * ch2 + 4 * kq) = v;
            if (KEEP && (unsigned)(r - 1) < (unsigned)th && (unsigned)(c - 1) < (unsigned)tw)
                *reinterpret_cast<f32x4*>(t2_img + ((size_t)(oy - 1 + r) * a.W + (ox - 1 + c)) * 32 + 16 * ch2 + 4 * kq) = v;
        }
    }
}

// ---- f3: 3x3, 32 -> 3 r^2, stored through the sub-pixel map
template <int G, bool KEEP>
__device__ __forceinline__ void espcn_f3_group(const EspcnArgs& a, const float* T2, float* hr_img, const float (&w3r)[72],
                                               const float (&b3r)[4], const int (&eo)[4], int m0, int step, int n3, int tw, int w2,
                                               int oy, int ox, int rc, int ch3, int li, int kq) {
    int la[G];
    f32x4 acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m0 + i * step) + li;
        const int tt = t < n3 ? t : 0;
        const int r = tt / tw, c = tt - r * tw;
        la[i] = (r * w2 + c) * kP2 + 4 * kq;
    }
    const f32x4 b3v = {b3r[0], b3r[1], b3r[2], b3r[3]};
    f32x4 b[G], bn[G];
#pragma unroll
    for (int i = 0; i < G; ++i) b[i] = *reinterpret_cast<const f32x4*>(T2 + la[i]);
#pragma unroll
    for (int t = 0; t < 18; ++t) {
        if (t + 1 < 18) {
            const int tap1 = (t + 1) / 2, g1 = (t + 1) % 2;
#pragma unroll
            for (int i = 0; i < G; ++i)
                bn[i] = *reinterpret_cast<const f32x4*>(T2 + la[i] + ((tap1 / 3) * w2 + (tap1 % 3)) * kP2 + 16 * g1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (G >= 3) {
                if (t == 0 && e == 0) mfma_sub_a_first(acc, b3v, w3r[0], b[0][0], b[1][0], b[2][0], b[G - 1][0]);
                else mfma_sub_a(acc, w3r[4 * t + e], b[0][e], b[1][e], b[2][e], b[G - 1][e]);
            } else {
#pragma unroll
                for (int i = 0; i < G; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3r[4 * t + e], b[i][e], (t == 0 && e == 0) ? b3v : acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) b[i] = bn[i];
    }
    if constexpr (G == 4) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    else if constexpr (G == 3) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int t = 16 * (m0 + i * step) + li;
        if (t < n3) {
            const int r = t / tw, c = t - r * tw;
            if (KEEP) {          // (training: y stays in sub-pixel space, [N,H,W,C3]; hr_img is y's image here)
                float* o = hr_img + ((size_t)(oy + r) * a.W + (ox + c)) * a.C3 + 16 * ch3 + 4 * kq;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (16 * ch3 + 4 * kq + e < a.C3) o[e] = acc[i][e];
            } else {
                float* o = hr_img + ((size_t)(oy + r) * a.r * a.W + (ox + c)) * rc;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (16 * ch3 + 4 * kq + e < a.C3) o[eo[e]] = acc[i][e];
            }
        }
    }
}

template <int NCH3, bool KEEP = false>
__global__ __launch_bounds__(256, 1) void espcn_fused_kernel(const EspcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* T1 = lds;
    float* X0 = lds + kT1;
    float* T2 = lds + kT1;          // (shares the halo's bytes: see kShared)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- this wave's filter slices, stationary for the whole kernel
    // f1: chunk = wave (16 of the 64 channels); k index = input channel kq (3 -> 4: channel 3 is zero)
    float w1r[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) w1r[t] = (kq < 3) ? a.w1[(t * 3 + kq) * 64 + 16 * wave + li] : 0.f;
    f32x4 b1r = *reinterpret_cast<const f32x4*>(a.b1 + 16 * wave + 4 * kq);
    // f2: chunk = wave & 1 (16 of the 32 channels); the two waves of a chunk take the even / the odd sub-tiles
    const int ch2 = wave & 1, half2 = wave >> 1;
    float w2r[144];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) w2r[t * 16 + j] = a.w2[(t * 64 + 16 * (j / 4) + 4 * kq + (j % 4)) * 32 + 16 * ch2 + li];
    f32x4 b2r = *reinterpret_cast<const f32x4*>(a.b2 + 16 * ch2 + 4 * kq);
    // f3: chunk = wave % NCH3 of ceil(3 r^2 / 16); the W3 waves of a chunk take every W3-th sub-tile
    constexpr int W3 = 4 / NCH3;             // waves per chunk (NCH3 = 3: one each, the fourth wave idles in this phase)
    const int ch3 = wave % NCH3, part3 = wave / NCH3;
    const bool on3 = part3 < W3;
    float w3r[72];
    {
        const int co = 16 * ch3 + li;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w3r[t * 8 + j] = (co < a.C3) ? a.w3[(t * 32 + 16 * (j / 4) + 4 * kq + (j % 4)) * a.C3 + co] : 0.f;
    }
    float b3r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b3r[e] = (16 * ch3 + 4 * kq + e < a.C3) ? a.b3[16 * ch3 + 4 * kq + e] : 0.f;
    // f2's and f3's weights live in accumulation registers from here on (216 of them: the 256 architectural VGPRs cannot
    // hold 241 weights AND double-buffered LDS fragments -- round 3's build read every fragment right before its MFMA).
    // They are defined as "a" values here and consumed by the "a" operands of the asm MFMA statements.
#pragma unroll
    for (int i = 0; i < 144; ++i) { const float t = w2r[i]; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(w2r[i]) : "v"(t)); }
#pragma unroll
    for (int i = 0; i < 72; ++i) { const float t = w3r[i]; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(w3r[i]) : "v"(t)); }
    // sub-pixel store: channel ch -> HR row + ch / (3 r), element + ch % (3 r)   (srx_conv_desc.subpixel_r)
    const int rc = 3 * a.r, hr_row = a.W * rc;
    int eo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = 16 * ch3 + 4 * kq + e, dy = ch / rc;
        eo[e] = dy * hr_row + (ch - dy * rc);
    }

    for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
        const int tx_i = u % a.tiles_x, t2_ = u / a.tiles_x;
        const int ty_i = t2_ % a.tiles_y, n = t2_ / a.tiles_y;
        const int oy = ty_i * a.TY, ox = tx_i * a.TX;
        const int th = (a.H - oy < a.TY) ? (a.H - oy) : a.TY;
        const int tw = (a.W - ox < a.TX) ? (a.W - ox) : a.TX;
        const int w0 = tw + 8, w1 = tw + 4, w2 = tw + 2;            // region widths: input halo, t1, t2
        const int n0 = (th + 8) * w0, n1 = (th + 4) * w1, n2 = (th + 2) * w2, n3 = th * tw;

        // ---- stage the input halo (zero outside the image), 4 floats per pixel
        __syncthreads();
        for (int p = tid; p < n0; p += 256) {
            const int r = p / w0, c = p - r * w0;
            const int ih = oy - 4 + r, iw = ox - 4 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
                const float* px = a.x + (((size_t)n * a.H + ih) * a.W + iw) * 3;
                v[0] = px[0]; v[1] = px[1]; v[2] = px[2];
            }
            *reinterpret_cast<f32x4*>(X0 + p * 4) = v;
        }
        __syncthreads();

        {   // f1: wave = channel chunk, all sub-tiles
            const int first = 0, step = 1, count = (n1 + 15) >> 4;
            float* t1_img = KEEP ? a.t1k + (size_t)n * a.H * a.W * 64 : nullptr;
#define SRX_F1(G) espcn_f1_group<G, KEEP>(a, X0, T1, w1r, b1r, m_, step, n1, w0, w1, oy, ox, wave, li, kq, t1_img, th, tw);
            SRX_ESPCN_GROUPS(SRX_F1)
#undef SRX_F1
        }
        __syncthreads();
        {   // f2: wave = (chunk, even / odd sub-tiles)
            const int nsub = (n2 + 15) >> 4;
            const int first = half2, step = 2, count = (nsub - half2 + 1) >> 1;
            float* t2_img = KEEP ? a.t2k + (size_t)n * a.H * a.W * 32 : nullptr;
#define SRX_F2(G) espcn_f2_group<G, KEEP>(a, T1, T2, w2r, b2r, m_, step, n2, w1, w2, oy, ox, ch2, li, kq, t2_img, th, tw);
            SRX_ESPCN_GROUPS(SRX_F2)
#undef SRX_F2
        }
        __syncthreads();
        if (on3) {   // f3: wave = (chunk, every W3-th sub-tile)
            float* hr_img = KEEP ? a.yk + (size_t)n * a.H * a.W * a.C3 : a.hr + (size_t)n * a.H * a.r * hr_row;
            const int nsub = (n3 + 15) >> 4;
            const int first = part3, step = W3, count = (nsub - part3 + W3 - 1) / W3;
#define SRX_F3(G) espcn_f3_group<G, KEEP>(a, T2, hr_img, w3r, b3r, eo, m_, step, n3, tw, w2, oy, ox, rc, ch3, li, kq);
            SRX_ESPCN_GROUPS(SRX_F3)
#undef SRX_F3
        }
    }
}

}  // namespace
}  // namespace srx

using namespace srx;

static int espcn_launch(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* w3, const float* b3, float* hr, float* t1k, float* t2k, float* yk, int N, int H, int W, int r,
                        srx_stream_t stream) {
    const bool keep = yk != nullptr;
    if (!x || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || (!keep && !hr) || (keep && (!t1k || !t2k)))
        return set_error(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (keep && (((uintptr_t)t1k | (uintptr_t)t2k) & 15u)) return set_error(SRX_ERR_ALIGN, "t1 / t2 must be 16-byte aligned");
    if (N <= 0 || H <= 0 || W <= 0) return set_error(SRX_ERR_BAD_ARG, "non-positive dimension");
    if (r < 2 || r > 4) return set_error(SRX_ERR_UNSUPPORTED, "espcn_forward: scaling factor %d (2..4 are built)", r);
    if (((uintptr_t)b1 | (uintptr_t)b2) & 15u) return set_error(SRX_ERR_ALIGN, "bias pointers must be 16-byte aligned");
    if ((long)N * H * W * 3L * r * r >= (1L << 31)) return set_error(SRX_ERR_UNSUPPORTED, "espcn_forward: output beyond 32-bit offsets");
    EspcnArgs a;
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.hr = hr;
    a.t1k = t1k; a.t2k = t2k; a.yk = yk;
    a.N = N; a.H = H; a.W = W; a.r = r; a.C3 = 3 * r * r;
    // Tile shape: <= 16 x 16; among all shapes the one with the least MFMA time: rounds of workgroups over the CUs x the
    // work of one tile (every phase costs a tile its sub-tiles of 16 pixels: 25 / 144 / 72 MFMAs each, shared by 4 / 2 / W3
    // waves) + a fixed cost per tile (staging, barriers, ramps).  [32,17,17]: 2 x 3 or 2 x 4 tiles per patch = 192 / 256
    // workgroups in one round (round 2 used 2 x 2 tiles of 9 x 9: 29.5 -> 22.7 us); a 256 x 256 image: 256 tiles of 16 x 16 in
    // one round (with the 9 x 9 limit of round 3: 841 tiles in four rounds).  The tiling does not change a bit of the result.
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0 || cus > 256) cus = 256;
    const int nch3 = (a.C3 + 15) / 16, waves3 = nch3 == 3 ? 1 : 4 / nch3;
    long best_cost = -1;
    a.TY = a.TX = kT;
    for (int ty = 1; ty <= kT && ty <= H; ++ty)
        for (int tx = 1; tx <= kT && tx <= W; ++tx) {
            // (only the even cuts: a tile edge that is not ceil(extent / number of tiles) just leaves a smaller last tile)
            if (ty != (H + (H + ty - 1) / ty - 1) / ((H + ty - 1) / ty) || tx != (W + (W + tx - 1) / tx - 1) / ((W + tx - 1) / tx)) continue;
            const long tiles = (long)N * ((H + ty - 1) / ty) * ((W + tx - 1) / tx);
            const long rounds = (tiles + cus - 1) / cus;
            const long s1 = ((ty + 4) * (tx + 4) + 15) / 16, s2 = ((ty + 2) * (tx + 2) + 15) / 16, s3 = (ty * tx + 15) / 16;
            const long cost = rounds * (25 * s1 + 144 * ((s2 + 1) / 2) + 72 * ((s3 + waves3 - 1) / waves3) + 150);   // (+150: staging, barriers)
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; a.TY = ty; a.TX = tx; }
        }
    a.tiles_y = (H + a.TY - 1) / a.TY; a.tiles_x = (W + a.TX - 1) / a.TX;
    const long units = (long)N * a.tiles_y * a.tiles_x;
    if (units >= (1L << 31)) return set_error(SRX_ERR_UNSUPPORTED, "espcn_forward: too many tiles");
    a.units = (int)units;
    const int grid = (int)(units < (long)cus ? units : (long)cus);
    const size_t lds = (size_t)(kT1 + kShared) * 4;
    hipError_t e;
    if (keep) {
        if (nch3 == 1) e = launch_with_lds(espcn_fused_kernel<1, true>, a, grid, lds, (hipStream_t)stream);
        else if (nch3 == 2) e = launch_with_lds(espcn_fused_kernel<2, true>, a, grid, lds, (hipStream_t)stream);
        else e = launch_with_lds(espcn_fused_kernel<3, true>, a, grid, lds, (hipStream_t)stream);
    } else {
        if (nch3 == 1) e = launch_with_lds(espcn_fused_kernel<1>, a, grid, lds, (hipStream_t)stream);
        else if (nch3 == 2) e = launch_with_lds(espcn_fused_kernel<2>, a, grid, lds, (hipStream_t)stream);
        else e = launch_with_lds(espcn_fused_kernel<3>, a, grid, lds, (hipStream_t)stream);
    }
    if (e != hipSuccess) return set_error(SRX_ERR_LAUNCH, "espcn_forward launch failed: %s", hipGetErrorString(e));
    return SRX_OK;
}

extern "C" int srx_espcn_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                 const float* w3, const float* b3, float* hr, int N, int H, int W, int r,
                                 srx_stream_t stream) {
    return espcn_launch(x, w1, b1, w2, b2, w3, b3, hr, nullptr, nullptr, nullptr, N, H, W, r, stream);
}

extern "C" int srx_espcn_forward_keep(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                      const float* w3, const float* b3, float* t1, float* t2, float* y, int N, int H, int W, int r,
                                      srx_stream_t stream) {
    if (!y) return set_error(SRX_ERR_BAD_ARG, "null tensor pointer");
    return espcn_launch(x, w1, b1, w2, b2, w3, b3, nullptr, t1, t2, y, N, H, W, r, stream);
}
