// Conv2DBackpropFilter instances.
#include "launchers.h"
namespace srx {
// out[j] = sum_g part[g][j] (+ wd * w[j]); fixed summation order -> deterministic.
__global__ void reduce_partials_kernel(const float* __restrict__ part, int G, size_t n, float* __restrict__ out,
                                       const float* __restrict__ w, float wd) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < G; g += 4) {
        s0 += part[(size_t)g * n + j];
        s1 += part[(size_t)(g + 1) * n + j];
        s2 += part[(size_t)(g + 2) * n + j];
        s3 += part[(size_t)(g + 3) * n + j];
    }
    for (; g < G; ++g) s0 += part[(size_t)g * n + j];
    float s = (s0 + s1) + (s2 + s3);
    if (w) s += wd * w[j];
    out[j] = s;
}


hipError_t launch_reduce_partials(const float* part, int G, size_t n, float* out, const float* w, float wd, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, G, n, out, w, wd);
    return hipGetLastError();
}

bool launch_wgrad(const ConvKey& k, const WgradArgs& a, int grid, size_t lds, hipStream_t s, hipError_t* err) {
    SRX_WGRAD_CASE(3, 3, 64, 4, 2)
    SRX_WGRAD_CASE(3, 3, 64, 2, 2)
    SRX_WGRAD_CASE(3, 3, 64, 1, 2)
    SRX_WGRAD_CASE(3, 3, 32, 4, 2)
    SRX_WGRAD_CASE(3, 3, 32, 2, 2)
    SRX_WGRAD_CASE(3, 3, 32, 1, 2)
    SRX_WGRAD_CASE(3, 3, 4, 4, 2)
    SRX_WGRAD_CASE(3, 3, 4, 2, 2)
    SRX_WGRAD_CASE(5, 5, 4, 4, 2)
    SRX_WGRAD_CASE(9, 9, 4, 4, 2)
    SRX_WGRAD_CASE(1, 1, 64, 4, 2)
    SRX_WGRAD_CASE(1, 1, 64, 2, 2)
    SRX_WGRAD_CASE(1, 1, 32, 4, 2)
    SRX_WGRAD_CASE(5, 5, 32, 1, 2)
    return false;
}
}  // namespace srx
