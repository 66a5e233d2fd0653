// Forward convolution into THREE output channels (the RGB output layer of VDSR / EnhanceNet: 3x3 64->3).  On the
// MFMA kernels such a layer pays for 16 output channels and uses 3; its real bound is the read of the input tensor
// (256 B per pixel against 3,456 FLOP).  Here 16 lanes share one output pixel, four input channels each: a
// wavefront's load instruction covers four whole pixels (4 x 256 contiguous bytes), bounds-checked (taps outside the
// image carry an out-of-range offset and read as zero, no divergence).  The lane's 9 x 4 x 3 weights stay in
// registers for the whole workgroup; the 16 partial sums of a pixel are added with four DPP steps.  Every input
// byte is read nine times, by the lanes of neighbouring pixels: L1 / L2 hits.  (A first version with one lane per
// pixel issued loads of 64 scattered 16-byte pieces and was slower than the MFMA kernel.)
#include "launchers.h"
namespace srx {
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x3v __attribute__((ext_vector_type(3)));

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(moved);
}

// Output position of a 16-lane group, advanced by 16 positions per step without divisions.
struct PixWalk {
    int n, oh, ow;
    __device__ __forceinline__ void init(unsigned p, int OH, int OW) {
        const unsigned t = p / (unsigned)OW;
        ow = (int)(p - t * (unsigned)OW);
        n = (int)(t / (unsigned)OH);
        oh = (int)(t - (unsigned)n * (unsigned)OH);
    }
    __device__ __forceinline__ void step16(int OH, int OW) { step(16, OH, OW); }
    __device__ __forceinline__ void step(int d, int OH, int OW) {
        ow += d;
        while (ow >= OW) {          // (at most once per step unless the image is narrower than 16 pixels)
            ow -= OW;
            oh += 1;
            if (oh == OH) { oh = 0; n += 1; }
        }
    }
};
// Byte offsets of the KH x KW input pixels under an output position (pixel size PB bytes, plus `lane_b`), or the
// out-of-range offset where the tap lies outside the image / the position is past the end: 3 + 3 range tests, then
// one add and one select per tap.
template <int KH, int KW>
__device__ __forceinline__ void tap_offsets(int (&off)[KH * KW], const PixWalk& q, bool live, int H, int W, int pad_t, int pad_l,
                                            int PB, int lane_b) {
    const int centre = ((q.n * H + q.oh - pad_t) * W + (q.ow - pad_l)) * PB + lane_b;
    bool rok[KH], cok[KW];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) rok[kh] = live && (unsigned)(q.oh + kh - pad_t) < (unsigned)H;
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) cok[kw] = (unsigned)(q.ow + kw - pad_l) < (unsigned)W;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
            off[kh * KW + kw] = (rok[kh] && cok[kw]) ? centre + (kh * W + kw) * PB : kOobOffset;
}

template <int KH, int KW, int CIN, int CO>
__global__ __launch_bounds__(256) void conv_narrow_kernel(const ConvArgs a, long total_px, int iters) {
    static_assert(CIN == 64, "16 lanes x 4 channels per pixel");
    constexpr int TAPS = KH * KW;
    const int c4 = threadIdx.x & 15;
    const int sub = threadIdx.x >> 4;
    float wr[TAPS][4][CO];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CO; ++c) wr[t][e][c] = a.w[((size_t)t * CIN + 4 * c4 + e) * CO + c];
    float bias[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) bias[c] = a.bias ? a.bias[c] : 0.0f;
    // one wave-uniform resource over the whole tensor: byte offsets fit 31 bits (checked by the host)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0,
                                                                         a.N * a.H * a.W * CIN * 4, 0x00020000);
    const unsigned p0 = blockIdx.x * (unsigned)iters * 16u + sub;      // (pixel indices fit 31 bits: host-checked)
    PixWalk q;
    q.init(p0 < (unsigned)total_px ? p0 : 0u, a.OH, a.OW);
    for (int it = 0; it < iters; ++it, q.step16(a.OH, a.OW)) {
        const unsigned p = p0 + (unsigned)it * 16u;
        const bool live = p < (unsigned)total_px;  // (uniform over the 16 lanes of a pixel)
        int off[TAPS];
        tap_offsets<KH, KW>(off, q, live, a.H, a.W, a.pad_t, a.pad_l, CIN * 4, 16 * c4);
        f32x4 v[TAPS];
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
            v[tp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off[tp], 0, 0));
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < CO; ++c) acc[c] = fmaf(v[tp][e], wr[tp][e][c], acc[c]);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float r = acc[c];
            r = dpp_add<0xB1>(r);     // quad_perm [1,0,3,2]
            r = dpp_add<0x4E>(r);     // quad_perm [2,3,0,1]
            r = dpp_add<0x124>(r);    // row_ror 4
            r = dpp_add<0x128>(r);    // row_ror 8: every lane of the 16 now holds the pixel's sum
            acc[c] = r;
        }
        if (live && c4 < CO) {
            // lane c of the 16 writes output channel c
            float r = (c4 == 0) ? acc[0] : (c4 == 1 ? acc[1 < CO ? 1 : 0] : acc[2 < CO ? 2 : 0]);
            r = act_apply(r + ((c4 == 0) ? bias[0] : (c4 == 1 ? bias[1 < CO ? 1 : 0] : bias[2 < CO ? 2 : 0])), a.act);
            const size_t o = (size_t)p * CO + c4;
            if (a.skip) r += a.skip[o];
            if (a.post_relu) r = act_apply(r, a.post_relu);
            a.y[o] = r;
        }
    }
}

// Filter gradient of the same two layer shapes (3x3, 64 <-> 3 channels).  Same lane layout (16 lanes per output
// position, four channels of the 64-channel tensor each); the 9 x 4 x 3 products of a position go into 108
// per-lane accumulators, the bias gradient into 3 or 4 more.  A workgroup walks a contiguous range of positions,
// then adds its 16 lane groups (xor-shuffles inside a wave, LDS across the four waves) and writes one partial
// filter in the layout reduce_partials_kernel sums: HWIO gradient, then the bias gradient.
// NOUT = true: Cin 64, Cout 3 (x is the wide tensor); false: Cin 3, Cout 64 (dpre is the wide tensor).
template <int KH, int KW, bool NOUT, int CW = 64>
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(const WgradArgs a, long total_px, int iters) {
    constexpr int TAPS = KH * KW, CN = 3;
    constexpr int LPP = CW / 4;                    // lanes per position (16, or 8 for the 32-channel side), four channels each
    constexpr int PPI = 256 / LPP;                 // positions per workgroup iteration
    constexpr int NB = NOUT ? CN : 4;              // bias-gradient values per lane
    constexpr int NV = TAPS * 4 * CN + NB;         // values per lane
    __shared__ float red[4 * NV * LPP];
    const int c4 = threadIdx.x & (LPP - 1);
    const int sub = threadIdx.x / LPP;
    const int wave = threadIdx.x >> 6;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, a.N * a.H * a.W * (NOUT ? CW : CN) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dpre), 0, a.N * a.OH * a.OW * (NOUT ? CN : CW) * 4, 0x00020000);
    const unsigned p0 = blockIdx.x * (unsigned)iters * (unsigned)PPI + sub;
    PixWalk q;
    q.init(p0 < (unsigned)total_px ? p0 : 0u, a.OH, a.OW);
    for (int it = 0; it < iters; ++it, q.step(PPI, a.OH, a.OW)) {
        const unsigned p = p0 + (unsigned)it * (unsigned)PPI;
        const bool live = p < (unsigned)total_px;
        int off[TAPS];
        tap_offsets<KH, KW>(off, q, live, a.H, a.W, a.pad_t, a.pad_l, (NOUT ? CW : CN) * 4, NOUT ? 16 * c4 : 0);
        if constexpr (NOUT) {
            const f32x3 d = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(drs, live ? (int)(p * CN * 4) : kOobOffset, 0, 0));
            f32x4 v[TAPS];
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
                v[tp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off[tp], 0, 0));
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < CN; ++c) acc[(tp * 4 + e) * CN + c] = fmaf(v[tp][e], d[c], acc[(tp * 4 + e) * CN + c]);
#pragma unroll
            for (int c = 0; c < CN; ++c) acc[TAPS * 4 * CN + c] += d[c];
        } else {
            const f32x4 d = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                drs, live ? (int)((p * CW + 4 * c4) * 4) : kOobOffset, 0, 0));
            f32x3 v[TAPS];
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
                v[tp] = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(xrs, off[tp], 0, 0));
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int cn = 0; cn < CN; ++cn)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[(tp * CN + cn) * 4 + e] = fmaf(v[tp][cn], d[e], acc[(tp * CN + cn) * 4 + e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[TAPS * 4 * CN + e] += d[e];
        }
    }
    // the wave's 64 / LPP lane groups, then the four waves
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float r = acc[i];
#pragma unroll
        for (int o = LPP; o < 64; o <<= 1) r += __shfl_xor(r, o);
        if (sub % (64 / LPP) == 0) red[(wave * NV + i) * LPP + c4] = r;
    }
    __syncthreads();
    float* part = a.part + (size_t)blockIdx.x * a.part_stride;
    const int wn = TAPS * CW * CN;
    for (int q = threadIdx.x; q < NV * LPP; q += 256) {
        const int i = q / LPP, l = q % LPP;
        const float r = (red[(0 * NV + i) * LPP + l] + red[(1 * NV + i) * LPP + l]) +
                        (red[(2 * NV + i) * LPP + l] + red[(3 * NV + i) * LPP + l]);
        if (i < TAPS * 4 * CN) {
            int idx;
            if constexpr (NOUT) {                  // i = (tap * 4 + e) * 3 + c  ->  dW[tap][4l + e][c]
                const int c = i % CN, te = i / CN, e = te & 3, tp = te >> 2;
                idx = (tp * CW + 4 * l + e) * CN + c;
            } else {                               // i = (tap * 3 + cn) * 4 + e  ->  dW[tap][cn][4l + e]
                const int e = i & 3, tc = i >> 2;
                idx = tc * CW + 4 * l + e;
            }
            part[idx] = r;
        } else {
            const int b = i - TAPS * 4 * CN;
            if constexpr (NOUT) {
                if (l == 0) part[wn + b] = r;      // (every lane of a pixel added the same three values)
            } else {
                part[wn + 4 * l + b] = r;
            }
        }
    }
}

bool launch_wgrad_narrow(const ConvKey& k, const WgradArgs& a, int grid, hipStream_t s, hipError_t* err) {
    if (k.kh != 3 || k.kw != 3) return false;
    // 64 -> 3 (output layers), 3 -> 64 (input layers), 3 -> 32 (the first layer of EnhanceNet's discriminator: on the
    // cursor MFMA kernel its filter gradient took 470 us at 128 x 128 x 128 -- 4.5 % of a discriminator run)
    const bool nout = a.Cin == 64 && a.Cout == 3, nin = a.Cin == 3 && a.Cout == 64, nin32 = a.Cin == 3 && a.Cout == 32;
    if (!nout && !nin && !nin32) return false;
    const long total = (long)a.N * a.OH * a.OW;
    if (total >= (1L << 31) - 4096 || (long)a.N * a.H * a.W * 64 * 4 >= (1L << 31) - 4096 || total * 64 * 4 >= (1L << 31) - 4096)
        return false;
    const long ppi = nin32 ? 32 : 16;          // positions per workgroup iteration
    const int iters = (int)((total + ppi * grid - 1) / (ppi * grid));
    if (nout)
        hipLaunchKernelGGL((wgrad_narrow_kernel<3, 3, true>), dim3(grid), dim3(256), 0, s, a, total, iters);
    else if (nin)
        hipLaunchKernelGGL((wgrad_narrow_kernel<3, 3, false>), dim3(grid), dim3(256), 0, s, a, total, iters);
    else
        hipLaunchKernelGGL((wgrad_narrow_kernel<3, 3, false, 32>), dim3(grid), dim3(256), 0, s, a, total, iters);
    *err = hipGetLastError();
    return true;
}

bool launch_conv_narrow(const ConvKey& k, const ConvArgs& a, hipStream_t s, hipError_t* err) {
    if (k.wt || a.Cout != 3 || a.mask) return false;
    if ((long)a.N * a.H * a.W * a.Cin * 4 >= (1L << 31) - 4096) return false;   // 31-bit byte offsets into the whole tensor
    const long total = (long)a.N * a.OH * a.OW;
    // 16 pixels per workgroup iteration; enough iterations to amortise the weight load, enough workgroups to fill the chip
    int iters = 16;
    while (iters > 1 && total / (16L * iters) < 2048) iters >>= 1;
    const long blocks = (total + 16L * iters - 1) / (16L * iters);
    if (total >= (1L << 31) - 4096) return false;
    if (k.kh == 3 && k.kw == 3 && a.Cin == 64) {
        hipLaunchKernelGGL((conv_narrow_kernel<3, 3, 64, 3>), dim3((unsigned)blocks), dim3(256), 0, s, a, total, iters);
    } else {
        return false;
    }
    *err = hipGetLastError();
    return true;
}
}  // namespace srx
