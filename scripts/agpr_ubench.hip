// agpr_ubench.hip -- diagnostic: does v_mfma_f32_16x16x4_f32 issue slower when its A operand comes from an
// AGPR instead of a VGPR?  (one wave per SIMD, 4 independent accumulators, back-to-back MFMAs)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: A in VGPR, B in VGPR; 1: A in AGPR; 2: A and B in AGPR; 3: acc in AGPR, A/B VGPR
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* st, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w[8], wa[8];
    for (int i = 0; i < 8; ++i) w[i] = 0.001f * (lane + i);
    for (int i = 0; i < 8; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(wa[i]) : "v"(w[i]));
    float b = 0.5f, ba;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ba) : "v"(b));
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[u]), "v"(b));
                if (MODE == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(wa[u]), "v"(b));
                if (MODE == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(wa[u]), "a"(ba));
                if (MODE == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(w[u]), "v"(b));
            }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (lane == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}
template <typename K> void run(const char* name, K kern) {
    const int grid = 256, iters = 4000;
    float* out; unsigned long long* st;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&st, grid * 4 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, st, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid * 4); hipMemcpy(h.data(), st, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::vector<double> c; for (auto x : h) c.push_back((double)x); std::sort(c.begin(), c.end());
    printf("%-40s %.2f cycles per MFMA\n", name, c[c.size() / 2] / (iters * 32.0));
}
int main() {
    run("A,B in VGPR, acc VGPR", k<0>);
    run("A in AGPR", k<1>);
    run("A and B in AGPR", k<2>);
    run("acc in AGPR, A/B VGPR", k<3>);
    return 0;
}
