// resample_u8.hip -- the reference's uint8 image resizes on the GPU, byte for byte.
//
// enet/enet/datasets.py:110-111 (training batches) and enet/enet/experiment_resolve.py:78-79 (inference) resize with
// scipy.misc.imresize, which is Pillow's Image.resize on the uint8 image: `imresize(hd, 25)` = BILINEAR down to 25 %
// (antialiased: the filter support grows with the scale), `imresize(sd, 400, 'bicubic')` = BICUBIC (a = -0.5) up 4x.
// Pillow (libImaging/Resample.c; a dependency of the reference, not vendored; the algorithm is stable across the
// versions that still ship scipy.misc.imresize and today's 12.x) resamples in two passes, horizontal then vertical, each
//     out = clip8((2^21 + sum_k kk[k] * in[xmin + k]) >> 22),
// with integer coefficients kk = round(w * 2^22) of the normalised filter weights w computed in double precision
// (precompute_coeffs / normalize_coeffs_8bpc); the intermediate image is uint8.  The coefficient tables are built on
// the HOST by the same double-precision arithmetic in the same order (srx_pil_resample_coeffs: no fused multiply-add on
// the x86-64 baseline), the passes are integer arithmetic on the device: the result does not depend on rounding modes or
// instruction selection.  Pinned by the reference's own output: assets/enet_eagle_bq.png (P5), 0 differing bytes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>

#include "../../include/srx.h"

namespace srx {
int set_error(int code, const char* fmt, ...);

namespace {
constexpr int kPrecisionBits = 32 - 8 - 2;

double bilinear_filter(double x) {
#pragma clang fp contract(off)
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
double bicubic_filter(double x) {
#pragma clang fp contract(off)     // (the products and sums below round one by one, as in Pillow's build)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
double filter_support(int filter) { return filter == SRX_RESAMPLE_BICUBIC ? 2.0 : 1.0; }

// one thread per output byte of [outer][out_size][inner]
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long total,
                                                          int in_size, int out_size, long inner, const int* __restrict__ bounds,
                                                          const int* __restrict__ kk, int ksize) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long i = idx % inner, t = idx / inner;
        const int o = (int)(t % out_size);
        const long outer = t / out_size;
        const int xmin = bounds[2 * o], n = bounds[2 * o + 1];
        const uint8_t* src = in + (outer * in_size + xmin) * inner + i;
        const int* k = kk + (long)o * ksize;
        int ss = 1 << (kPrecisionBits - 1);
        for (int x = 0; x < n; ++x) ss += (int)src[x * inner] * k[x];
        ss >>= kPrecisionBits;                       // (arithmetic shift: negative sums clip to 0 like Pillow's lookup table)
        out[idx] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

// astype(float32) / 127.5 - 1.0 (enet/enet/datasets.py:113-115): two roundings, division then subtraction
__global__ __launch_bounds__(256) void u8_to_pm1_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, size_t n) {
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (float)in[i] / 127.5f - 1.0f;
}

}  // namespace
}  // namespace srx

using namespace srx;

extern "C" int srx_pil_resample_ksize(int in_size, int out_size, int filter) {
    if (in_size <= 0 || out_size <= 0 || (filter != SRX_RESAMPLE_BILINEAR && filter != SRX_RESAMPLE_BICUBIC)) return -1;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(filter_support(filter) * filterscale) * 2 + 1;
}

// HOST arrays: bounds [out_size][2] = (first input index, count), kk [out_size][ksize] (zero past the count)
extern "C" int srx_pil_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk) {
#pragma clang fp contract(off)
    const int ksize = srx_pil_resample_ksize(in_size, out_size, filter);
    if (ksize < 0 || !bounds || !kk) return set_error(SRX_ERR_BAD_ARG, "pil_resample_coeffs: bad sizes, filter or null table");
    double (*const f)(double) = filter == SRX_RESAMPLE_BICUBIC ? bicubic_filter : bilinear_filter;
    double filterscale, scale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = filter_support(filter) * filterscale;
    double* w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            w[x] = f((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) w[x] /= ww;
        int32_t* k = kk + (long)xx * ksize;
        for (x = 0; x < xmax; ++x)
            k[x] = (int32_t)(w[x] < 0 ? -0.5 + w[x] * (1 << kPrecisionBits) : 0.5 + w[x] * (1 << kPrecisionBits));
        for (; x < ksize; ++x) k[x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return SRX_OK;
}

extern "C" int srx_resample_u8(const uint8_t* in, uint8_t* out, long outer, int in_size, int out_size, long inner,
                               const int32_t* bounds, const int32_t* kk, int ksize, srx_stream_t stream) {
    if (!in || !out || !bounds || !kk) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (outer <= 0 || in_size <= 0 || out_size <= 0 || inner <= 0 || ksize <= 0) return set_error(SRX_ERR_BAD_ARG, "bad dims");
    if (in == out) return set_error(SRX_ERR_BAD_ARG, "resample_u8 cannot run in place");
    const long total = outer * out_size * inner;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(resample_u8_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, in, out,
                       total, in_size, out_size, inner, (const int*)bounds, (const int*)kk, ksize);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(SRX_ERR_LAUNCH, "resample_u8 launch failed: %s", hipGetErrorString(e));
    return SRX_OK;
}

extern "C" int srx_u8_to_pm1(const uint8_t* in, float* out, size_t n, srx_stream_t stream) {
    if (!in || !out) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (n == 0) return SRX_OK;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(u8_to_pm1_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(SRX_ERR_LAUNCH, "u8_to_pm1 launch failed: %s", hipGetErrorString(e));
    return SRX_OK;
}
