// shadow_ubench.hip -- diagnostic: how many instructions of its OWN can a wave issue between two
// fp32 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles each) without slowing the MFMA stream?
// One wave per SIMD (256-thread workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: n independent v_fma per MFMA; 1: n v_mul_lo_u32 (quarter rate) per MFMA;
// 2: one ds_read_b128 + n v_fma per MFMA; 3: one global_load_dwordx4 per 4 MFMAs + n v_fma per MFMA;
// 4: one ds_write_b128 per 2 MFMAs + n v_fma; 5: one global_store_dwordx4 per 4 MFMAs + n v_fma
template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* gsrc, float* gdst, unsigned long long* st, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w = 0.001f * lane, b = 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + i;
    unsigned iv[8];
    for (int i = 0; i < 8; ++i) iv[i] = lane + i;
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + threadIdx.x;
    f32x4* lw = reinterpret_cast<f32x4*>(lds) + 1024 + threadIdx.x;
    const f32x4* gp = reinterpret_cast<const f32x4*>(gsrc) + (size_t)blockIdx.x * 65536 + threadIdx.x;
    f32x4* gq = reinterpret_cast<f32x4*>(gdst) + (size_t)blockIdx.x * 65536 + threadIdx.x;
    f32x4 ld = {0, 0, 0, 0}, gl = {0, 0, 0, 0};
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc[u & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (KIND == 2) { ld += lp[((u + it) & 3) * 256]; }
            if (KIND == 3 && (u & 3) == 0) { gl += gp[((it * 4 + (u >> 2)) & 255) * 256]; }
            if (KIND == 4 && (u & 1) == 0) { lw[(u & 2) * 128] = acc[3] ; }
            if (KIND == 5 && (u & 3) == 0) { gq[((it * 4 + (u >> 2)) & 255) * 256] = f32x4{v[0], v[1], v[2], v[3]}; }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 1) iv[j & 7] = iv[j & 7] * 41u + 3u;
                else v[j & 7] = __builtin_fmaf(v[j & 7], 0.999f, 0.001f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + ld[0] + gl[1];
    for (int i = 0; i < 8; ++i) r += v[i] + (float)iv[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (lane == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <typename K>
void run(const char* name, K kern) {
    const int grid = 256, iters = 2000;
    float *out, *g, *gd; unsigned long long* st;
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&g, (size_t)grid * 65536 * 16 + (1 << 22)); hipMemset(g, 0, (size_t)grid * 65536 * 16 + (1 << 22));
    hipMalloc(&gd, (size_t)grid * 65536 * 16 + (1 << 22));
    hipMalloc(&st, grid * 4 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, g, gd, st, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), st, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::vector<double> c; for (auto x : h) c.push_back((double)x);
    std::sort(c.begin(), c.end());
    printf("%-52s %.2f cycles per MFMA\n", name, c[c.size() / 2] / (iters * 16.0));
    hipFree(out); hipFree(g); hipFree(gd); hipFree(st);
}

int main() {
    run("MFMA only", k<0, 0>);
    run("+1 v_fma per MFMA", k<0, 1>);
    run("+2 v_fma", k<0, 2>);
    run("+4 v_fma", k<0, 4>);
    run("+6 v_fma", k<0, 6>);
    run("+8 v_fma", k<0, 8>);
    run("+1 v_mul_lo_u32", k<1, 1>);
    run("+2 v_mul_lo_u32", k<1, 2>);
    run("+4 v_mul_lo_u32", k<1, 4>);
    run("+1 ds_read_b128 (+acc) per MFMA", k<2, 0>);
    run("+1 ds_read_b128 +2 v_fma", k<2, 2>);
    run("+1 global_load_x4 per 4 MFMA", k<3, 0>);
    run("+1 global_load_x4 per 4 MFMA +2 v_fma", k<3, 2>);
    run("+1 ds_write_b128 per 2 MFMA", k<4, 0>);
    run("+1 ds_write_b128 per 2 MFMA +2 v_fma", k<4, 2>);
    run("+1 global_store_x4 per 4 MFMA", k<5, 0>);
    run("+1 global_store_x4 per 4 MFMA +2 v_fma", k<5, 2>);
    return 0;
}
