// shadow3_ubench.hip -- diagnostic: cost of SALU / memory instructions between a wave's own fp32 MFMAs, and a
// functional check that a raw buffer load's bounds check ignores the SGPR offset (soffset).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* gsrc, float* gdst, unsigned long long* st, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w = 0.001f * lane, b = 0.5f;
    int s0 = 5 + blockIdx.x, s1 = 3, s2 = 0;
    unsigned long long m0 = 0xffffffff0000ffffull;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, 1 << 24, 0x00020000);
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(gdst, 0, 1 << 24, 0x00020000);
    int voff = threadIdx.x * 16, laddr = threadIdx.x * 16;
    f32x4 ld = {0, 0, 0, 0};
    int soff = 0;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(w), "v"(b));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1));
                if (KIND == 2) asm volatile("s_cmp_lt_i32 %0, %1\n\ts_cselect_b32 %2, %0, %1" : "+s"(s0), "+s"(s1), "+s"(s2) : : "scc");
                if (KIND == 3) asm volatile("s_mov_b64 exec, %0\n\ts_mov_b64 exec, -1" : : "s"(m0));
                if (KIND == 4) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
                if (KIND == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(ld[0]) : "v"(laddr) : "memory");
                if (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(laddr) : "memory");
                if (KIND == 7) asm volatile("ds_write_b128 %1, %0 offset:16384" : : "v"(ld), "v"(laddr) : "memory");
                if (KIND == 8) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ld) : "v"(voff), "s"(xr), "s"(soff) : "memory");
                if (KIND == 9) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(ld), "v"(voff), "s"(yr), "s"(soff) : "memory");
                if (KIND == 10) asm volatile("s_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_mov_b64 exec, -1" : "=v"(ld) : "v"(voff), "s"(xr), "s"(soff), "s"(m0) : "memory");
                if (KIND == 11) asm volatile("s_waitcnt lgkmcnt(0)");
                if (KIND == 12) asm volatile("s_waitcnt vmcnt(0)");
                if (KIND == 13) asm volatile("s_branch 1f\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n1:");            // taken branch over 4 instructions
                if (KIND == 14) asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc0 1f\n\ts_nop 0\n1:" : : "s"(s1) : "scc");   // not-taken conditional branch
                if (KIND == 15) asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n1:" : : "s"(s1) : "scc");   // taken conditional branch over 20 instructions
            }
            if (KIND == 8 || KIND == 9 || KIND == 10) soff = (soff + 4096) & ((1 << 23) - 1);
        }
        if (KIND >= 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)(s0 + s1 + s2) + ld[0] + ld[3];
    if (lane == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

// functional: bounds check vs soffset.  num_records = 64 bytes.  lane L reads voffset = 16*(L&7) with soffset = 4096:
// lanes 0..3 are in range by voffset alone (16*L+16 <= 64); if soffset took part in the check everything would be 0.
__global__ void kcheck(const float* src, float* res) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 64, 0x00020000);
    const int lane = threadIdx.x;
    f32x4 v;
    int voff = 16 * (lane & 7), soff = 4096;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(voff), "s"(r), "s"(soff) : "memory");
    res[lane] = v[0];
    // store side: same rule? write 7.0 with soffset 8192, num_records 64
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(res + 1024, 0, 64, 0x00020000);
    f32x4 s = {7.f, 7.f, 7.f, 7.f};
    int soff2 = 8192;
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : : "v"(s), "v"(voff), "s"(w), "s"(soff2) : "memory");
}

template <typename K>
void run(const char* name, K kern) {
    const int grid = 256, iters = 1000;
    float *out, *g, *gd; unsigned long long* st;
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&g, 1 << 25); hipMemset(g, 0, 1 << 25);
    hipMalloc(&gd, 1 << 25);
    hipMalloc(&st, grid * 4 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, g, gd, st, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), st, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::vector<double> c; for (auto x : h) c.push_back((double)x);
    std::sort(c.begin(), c.end());
    printf("%-44s %.2f cycles per MFMA\n", name, c[c.size() / 2] / (iters * 16.0));
    hipFree(out); hipFree(g); hipFree(gd); hipFree(st);
}
#define R(K, NAME) run(NAME " x1", k<K, 1>); run(NAME " x2", k<K, 2>); run(NAME " x4", k<K, 4>);
int main() {
    {
        float *src, *res;
        hipMalloc(&src, 1 << 16); hipMalloc(&res, 1 << 16);
        std::vector<float> h(1 << 14);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
        hipMemcpy(src, h.data(), 1 << 16, hipMemcpyHostToDevice);
        hipMemset(res, 0, 1 << 16);
        hipLaunchKernelGGL(kcheck, dim3(1), dim3(64), 0, 0, src, res);
        hipDeviceSynchronize();
        std::vector<float> r(1 << 14);
        hipMemcpy(r.data(), res, 1 << 16, hipMemcpyDeviceToHost);
        printf("load  (num_records 64, soffset 4096): lanes 0..7 got");
        for (int i = 0; i < 8; ++i) printf(" %.0f", r[i]);
        printf("   (expect 1024 1028 1032 1036 0 0 0 0 if soffset is outside the check)\n");
        printf("store (num_records 64, soffset 8192): dst[2048+4*i] =");
        for (int i = 0; i < 8; ++i) printf(" %.0f", r[1024 + 2048 + 4 * i]);
        printf("   (expect 7 7 7 7 0 0 0 0)\n");
    }
    run("MFMA only", k<0, 0>);
    R(1, "s_add_u32") R(2, "s_cmp+s_cselect") R(3, "s_mov exec x2") R(4, "s_mul_i32")
    R(5, "ds_read_b32") R(6, "ds_read_b128") R(7, "ds_write_b128") R(8, "buffer_load_x4 soffset")
    R(9, "buffer_store_x4 soffset") R(10, "exec-masked buffer_load") R(11, "s_waitcnt lgkmcnt(0)") R(12, "s_waitcnt vmcnt(0)")
    R(13, "taken s_branch (+4)") R(14, "not-taken s_cbranch") R(15, "taken s_cbranch (+20)")
    return 0;
}
