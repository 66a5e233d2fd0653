// coissue_ubench.hip -- diagnostic: what does a wave doing VALU / LDS / global-memory work get when
// the OTHER wave on its SIMD streams fp32 MFMAs back to back (and what does that cost the MFMA wave)?
// 512-thread workgroup = 2 waves per SIMD: waves 0-3 stream v_mfma_f32_16x16x4_f32, waves 4-7 run the
// probe.  One workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/coissue_ubench.hip -o scripts/coissue_ubench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PROBE: 0 none, 1 dependent VALU chain, 2 independent VALU, 3 ds_read_b128 loop, 4 global_load loop,
//        5 integer address math (v_mul_lo, cvt) mix
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PROBE, bool MFMA_ON, int PRIO, int YIELD = 0, int SHAPE = 16>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* gsrc, unsigned long long* st, int mfma_iters,
                                           int probe_iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned long long c0 = 0, c1 = 0;
    float res = 0.f;
    if (wave < 4) {
        if (MFMA_ON && SHAPE == 32) {
            f32x16 acc32[2];
            for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
            float w = 0.001f * lane, b = 0.5f;
            c0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, acc32[i], 0, 0, 0);
            }
            c1 = __builtin_amdgcn_s_memtime();
            res = acc32[0][0] + acc32[1][5];
        } else if (MFMA_ON) {
            f32x4 acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            float w = 0.001f * lane, b = 0.5f;
            c0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc[i], 0, 0, 0);
                    if (YIELD == 1 && (u & 3) == 3) asm volatile("s_nop 15");
                    if (YIELD == 2 && (u & 3) == 3) asm volatile("s_sleep 1");
                    if (YIELD == 3 && (u & 1) == 1) asm volatile("s_nop 7");
                    if (YIELD == 4 && u == 7) asm volatile("s_sleep 1");
                    if (YIELD == 5 && (u & 3) == 3) asm volatile("s_nop 15\n\ts_nop 15");
                }
            }
            c1 = __builtin_amdgcn_s_memtime();
            res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        }
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        float x = 1.0f + lane * 1e-3f, y = 0.5f, z = 0.25f, q = 0.125f;
        int ia = lane * 7 + 3;
        c0 = __builtin_amdgcn_s_memtime();
        if (PROBE == 1) {
            for (int it = 0; it < probe_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, 0.999f, 0.001f);
            }
        } else if (PROBE == 2) {
            for (int it = 0; it < probe_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x = __builtin_fmaf(x, 0.999f, 0.001f); y = __builtin_fmaf(y, 0.998f, 0.002f);
                    z = __builtin_fmaf(z, 0.997f, 0.003f); q = __builtin_fmaf(q, 0.996f, 0.004f);
                }
            }
        } else if (PROBE == 3) {
            const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
            f32x4 a = {0, 0, 0, 0};
            for (int it = 0; it < probe_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) a += lp[(u * 64 + it) & 511];
            }
            x = a[0] + a[1] + a[2] + a[3];
        } else if (PROBE == 4) {
            const f32x4* gp = reinterpret_cast<const f32x4*>(gsrc) + (size_t)blockIdx.x * 4096 + lane;
            f32x4 a = {0, 0, 0, 0};
            for (int it = 0; it < probe_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) a += gp[((u + it * 16) & 63) * 64];
            }
            x = a[0] + a[1] + a[2] + a[3];
        } else if (PROBE == 5) {
            for (int it = 0; it < probe_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int t = (int)(((float)ia + 0.5f) * 0.02439f);
                    t -= (t * 41 > ia);
                    ia = ia + t * 3 + 1;
                    ia &= 0xffff;
                }
            }
            x = (float)ia;
        }
        if (PROBE == 6) {          // VALU-free: global loads with fixed address registers, SALU loop
            const f32x4* gp = reinterpret_cast<const f32x4*>(gsrc) + (size_t)blockIdx.x * 4096 + lane;
            f32x4 t0, t1, t2, t3;
            for (int it = 0; it < probe_iters; ++it) {
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                             "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(gp) : "memory");
            }
            x = t0[0] + t1[0] + t2[0] + t3[0];
        } else if (PROBE == 7) {   // VALU-free: ds_read_b128 with a fixed address register
            const unsigned la = (unsigned)(lane * 16);
            f32x4 t0, t1, t2, t3;
            for (int it = 0; it < probe_iters; ++it) {
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                             "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(la) : "memory");
            }
            x = t0[0] + t1[0] + t2[0] + t3[0];
        } else if (PROBE == 8) {   // VALU-free: ds_write_b128
            const unsigned la = (unsigned)(lane * 16 + 8192);
            f32x4 t0 = {x, y, z, q};
            for (int it = 0; it < probe_iters; ++it) {
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:1024\n\t"
                             "ds_write_b128 %0, %1 offset:2048\n\tds_write_b128 %0, %1 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                             :: "v"(la), "v"(t0) : "memory");
            }
        } else if (PROBE == 10) {  // LDS-DMA: global -> LDS directly, 4 B per lane (one 256-B pixel per wave-instruction)
            const float* gp = gsrc + (size_t)blockIdx.x * 16384 + lane;
            float* lbase = lds + 2048 + (wave - 4) * 1024;     // wave-uniform LDS destination
            for (int it = 0; it < probe_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    __builtin_amdgcn_global_load_lds(gp + u * 64, (__attribute__((address_space(3))) void*)(lbase + u * 64), 4, 0, 0);   // loop-invariant addresses: no VALU
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (PROBE == 11) {  // LDS-DMA 16 B per lane (1 KiB per wave-instruction)
            const float* gp = gsrc + (size_t)blockIdx.x * 16384 + lane * 4;
            float* lbase = lds + 2048 + (wave - 4) * 1024;
            for (int it = 0; it < probe_iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    __builtin_amdgcn_global_load_lds(gp + u * 256, (__attribute__((address_space(3))) void*)(lbase + (u & 3) * 256), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (PROBE == 9) {   // SALU only
            unsigned sa = 1;
            for (int it = 0; it < probe_iters; ++it) {
                asm volatile("s_add_u32 %0, %0, 3\n\ts_lshl_b32 %0, %0, 1\n\ts_add_u32 %0, %0, 5\n\ts_lshr_b32 %0, %0, 1" : "+s"(sa));
            }
            x = (float)sa;
        }
        c1 = __builtin_amdgcn_s_memtime();
        res = x + y + z + q;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0) { st[(blockIdx.x * 8 + wave) * 2] = c0; st[(blockIdx.x * 8 + wave) * 2 + 1] = c1; }
}

template <typename K>
void run(const char* name, K kern, int mfma_iters, int probe_iters, int probe_instr) {
    const int grid = 256;
    float *out, *g; unsigned long long* st;
    hipMalloc(&out, grid * 512 * 4); hipMalloc(&g, (size_t)grid * 4096 * 16 + 65536 * 16); hipMemset(g, 0, (size_t)grid * 4096 * 16 + 65536 * 16);
    hipMalloc(&st, grid * 16 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, g, st, mfma_iters, probe_iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid * 16);
    hipMemcpy(h.data(), st, grid * 16 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ma, pr;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) {
        double d = (double)(h[(b * 8 + w) * 2 + 1] - h[(b * 8 + w) * 2]);
        (w < 4 ? ma : pr).push_back(d);
    }
    std::sort(ma.begin(), ma.end()); std::sort(pr.begin(), pr.end());
    double nm = (double)mfma_iters * 32;
    printf("%-46s mfma wave: %9.0f cyc (%.1f cyc/MFMA)   probe wave: %9.0f cyc (%.1f cyc/instr)\n", name, ma[ma.size() / 2],
           nm > 0 ? ma[ma.size() / 2] / nm : 0.0, pr[pr.size() / 2], probe_instr ? pr[pr.size() / 2] / ((double)probe_iters * probe_instr) : 0.0);
    hipFree(out); hipFree(g); hipFree(st);
}

int main() {
    const int MI = 4000, PI = 500;
    run("MFMA alone", k<0, true, 0>, MI, 0, 0);
    run("dep VALU alone", k<1, false, 0>, 0, PI, 16);
    run("dep VALU + MFMA partner", k<1, true, 0>, MI, PI, 16);
    run("dep VALU + MFMA partner, probe prio 3", k<1, true, 3>, MI, PI, 16);
    run("indep VALU alone", k<2, false, 0>, 0, PI, 16);
    run("indep VALU + MFMA partner", k<2, true, 0>, MI, PI, 16);
    run("indep VALU + MFMA partner, probe prio 3", k<2, true, 3>, MI, PI, 16);
    run("ds_read_b128 alone", k<3, false, 0>, 0, PI, 16);
    run("ds_read_b128 + MFMA partner", k<3, true, 0>, MI, PI, 16);
    run("global_load x4 alone", k<4, false, 0>, 0, PI, 16);
    run("global_load x4 + MFMA partner", k<4, true, 0>, MI, PI, 16);
    run("int addr math alone", k<5, false, 0>, 0, PI, 4);
    run("int addr math + MFMA partner", k<5, true, 0>, MI, PI, 4);
    run("int addr math + MFMA partner, prio 3", k<5, true, 3>, MI, PI, 4);
    run("VALU-free global_load x4 alone", k<6, false, 0>, 0, PI, 4);
    run("VALU-free global_load x4 + MFMA partner", k<6, true, 0>, MI, PI, 4);
    run("VALU-free ds_read_b128 x4 alone", k<7, false, 0>, 0, PI, 4);
    run("VALU-free ds_read_b128 x4 + MFMA partner", k<7, true, 0>, MI, PI, 4);
    run("VALU-free ds_write_b128 x4 alone", k<8, false, 0>, 0, PI, 4);
    run("VALU-free ds_write_b128 x4 + MFMA partner", k<8, true, 0>, MI, PI, 4);
    run("32x32x2 MFMA alone (16/iter)", k<0, true, 0, 0, 32>, MI, 0, 0);
    run("indep VALU + 32x32x2 MFMA partner", k<2, true, 0, 0, 32>, MI, PI, 16);
    run("VALU-free global_load + 32x32x2 partner", k<6, true, 0, 0, 32>, MI, PI, 4);
    run("LDS-DMA dword x4 alone", k<10, false, 0>, 0, PI, 4);
    run("LDS-DMA dword x4 + MFMA partner", k<10, true, 0>, MI, PI, 4);
    run("LDS-DMA dwordx4 x4 alone", k<11, false, 0>, 0, PI, 4);
    run("LDS-DMA dwordx4 x4 + MFMA partner", k<11, true, 0>, MI, PI, 4);
    run("SALU only alone", k<9, false, 0>, 0, PI, 4);
    run("SALU only + MFMA partner", k<9, true, 0>, MI, PI, 4);
    run("indep VALU + MFMA(yield s_nop15/16)", k<2, true, 0, 1>, MI, PI, 16);
    run("indep VALU + MFMA(yield s_sleep1/16)", k<2, true, 0, 2>, MI, PI, 16);
    run("indep VALU + MFMA(yield s_nop7/8)", k<2, true, 0, 3>, MI, PI, 16);
    run("indep VALU + MFMA(yield s_sleep1/32)", k<2, true, 0, 4>, MI, PI, 16);
    run("indep VALU + MFMA(yield 2x s_nop15/16)", k<2, true, 0, 5>, MI, PI, 16);
    run("int addr math + MFMA(yield s_sleep1/16)", k<5, true, 0, 2>, MI, PI, 4);
    run("global_load + MFMA(yield s_sleep1/16)", k<4, true, 0, 2>, MI, PI, 16);
    run("ds_read + MFMA(yield s_sleep1/16)", k<3, true, 0, 2>, MI, PI, 16);
    return 0;
}
