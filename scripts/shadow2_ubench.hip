// shadow2_ubench.hip -- diagnostic: cost of ONE given instruction (x N) issued by a wave between two of
// its own fp32 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles each), by instruction type.  One wave per SIMD.
// Follow-up of shadow_ubench: v_mul_lo_u32 was cheap (+2.3 cycles), v_fma expensive (+13); which class do
// the integer instructions the staging / address code is made of belong to?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ONE(K, TXT)                                                                         \
    if (KIND == K) asm volatile(TXT : "+v"(x0), "+v"(x1), "+s"(s0) : "v"(y0), "v"(y1), "s"(s1));

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* st, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w = 0.001f * lane, b = 0.5f;
    int x0 = lane, x1 = lane * 3, y0 = 7 + lane, y1 = 11;
    int s0 = 5, s1 = 3;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(w), "v"(b));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                ONE(1, "v_add_u32 %0, %3, %0")
                ONE(2, "v_cndmask_b32 %0, %3, %4, vcc")
                ONE(3, "v_and_b32 %0, %3, %0")
                ONE(4, "v_mov_b32 %0, %3")
                ONE(5, "v_lshl_add_u32 %0, %0, 2, %3")
                ONE(6, "v_cmp_lt_i32 vcc, %0, %3")
                ONE(7, "v_max_i32 %0, %0, %3")
                ONE(8, "v_add3_u32 %0, %0, %3, %4")
                ONE(9, "v_mul_lo_u32 %0, %0, %3")
                ONE(10, "v_mul_hi_u32 %0, %0, %3")
                ONE(11, "v_mad_u32_u24 %0, %0, %3, %4")
                ONE(12, "s_add_u32 %2, %2, %5")
                ONE(13, "s_mul_i32 %2, %2, %5")
                ONE(14, "v_exp_f32 %0, %0")
                ONE(15, "v_mov_b32_dpp %0, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                ONE(16, "v_readfirstlane_b32 %2, %0")
                ONE(17, "v_mul_f32 %0, %0, %3")
                ONE(18, "v_add_f32 %0, %0, %3")
                ONE(19, "v_cvt_f32_i32 %0, %0")
                ONE(20, "v_accvgpr_write_b32 a0, %0")
                ONE(22, "v_mul_u32_u24 %0, %0, %3")
                ONE(23, "v_sub_u32 %0, %0, %3")
                ONE(24, "v_xor_b32 %0, %0, %3")
                ONE(25, "s_nop 0")
                ONE(26, "v_nop")
            }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)(x0 + x1 + s0);
    if (lane == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <typename K>
void run(const char* name, K kern) {
    const int grid = 256, iters = 2000;
    float* out; unsigned long long* st;
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&st, grid * 4 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, st, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), st, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::vector<double> c; for (auto x : h) c.push_back((double)x);
    std::sort(c.begin(), c.end());
    printf("%-44s %.2f cycles per MFMA\n", name, c[c.size() / 2] / (iters * 16.0));
    hipFree(out); hipFree(st);
}
#define R(K, NAME) run(NAME " x1", k<K, 1>); run(NAME " x2", k<K, 2>); run(NAME " x4", k<K, 4>);
int main() {
    run("MFMA only", k<0, 0>);
    R(1, "v_add_u32") R(2, "v_cndmask_b32") R(3, "v_and_b32") R(4, "v_mov_b32") R(5, "v_lshl_add_u32")
    R(6, "v_cmp_lt_i32") R(7, "v_max_i32") R(8, "v_add3_u32") R(9, "v_mul_lo_u32") R(10, "v_mul_hi_u32")
    R(11, "v_mad_u32_u24") R(12, "s_add_u32") R(13, "s_mul_i32") R(14, "v_exp_f32") R(15, "v_mov_b32_dpp")
    R(16, "v_readfirstlane") R(17, "v_mul_f32") R(18, "v_add_f32") R(19, "v_cvt_f32_i32") R(20, "v_accvgpr_write")
    R(22, "v_mul_u32_u24") R(23, "v_sub_u32") R(24, "v_xor_b32") R(25, "s_nop 0") R(26, "v_nop")
    return 0;
}
