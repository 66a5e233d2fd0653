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

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(moved);
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
    for (int it = 0; it < iters; ++it) {
        const unsigned p = p0 + (unsigned)it * 16u;
        const bool live = p < (unsigned)total_px;  // (uniform over the 16 lanes of a pixel)
        const unsigned pp = live ? p : 0u;
        const unsigned t = pp / (unsigned)a.OW;
        const int ow = (int)(pp - t * (unsigned)a.OW);
        const int n = (int)(t / (unsigned)a.OH);
        const int oh = (int)(t - (unsigned)n * (unsigned)a.OH);
        const int img = n * a.H * a.W;
        f32x4 v[TAPS];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int ih = oh + kh - a.pad_t, iw = ow + kw - a.pad_l;
                const bool ok = live && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                const int off = ok ? ((img + ih * a.W + iw) * CIN + 4 * c4) * 4 : kOobOffset;
                v[kh * KW + kw] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
            }
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
            if (a.post_relu) r = fmaxf(r, 0.0f);
            a.y[o] = r;
        }
    }
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
