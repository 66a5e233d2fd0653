// pipe_ubench.hip -- diagnostic: the pipelined conv kernel's group loop (144 stationary weights, G=4
// accumulators, one ds_read_b128 per accumulator per 16-MFMA block) with ingredients toggled, one wave per
// SIMD.  What does each ingredient cost the fp32-MFMA stream?
//   AG    weights in AGPRs ("a" operands) instead of VGPRs
//   SPLIT a block = 4 asm statements of 4 MFMAs (else one statement of 16)
//   LOADS n bounds-checked buffer_load_dwordx4 per 36-block group + matching ds_write_b128
//   STORES 4 buffer_store_dwordx4 per group
//   LDS2  read from a 160 KiB allocation (two tile buffers) instead of 80 KiB
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));
struct Stamp { unsigned long long clk0, clk1, rt0, rt1; };

#define MF(ACC, A, B) "v_mfma_f32_16x16x4_f32 %" #ACC ", %" #A ", %" #B ", %" #ACC "\n\t"

template <bool AG>
__device__ __forceinline__ void sub4(f32x4 (&c)[4], float w, float b0, float b1, float b2, float b3) {
    if constexpr (AG)
        asm volatile(MF(0, 4, 5) MF(1, 4, 6) MF(2, 4, 7) MF(3, 4, 8)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "a"(w), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
    else
        asm volatile(MF(0, 4, 5) MF(1, 4, 6) MF(2, 4, 7) MF(3, 4, 8)
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "v"(w), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
}
template <bool AG>
__device__ __forceinline__ void blk16(f32x4 (&c)[4], float w0, float w1, float w2, float w3, const f32x4 (&b)[4]) {
#define OPS "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
#define BS "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]), "v"(b[2][0]), "v"(b[2][1]), "v"(b[2][2]), "v"(b[2][3]), "v"(b[3][0]), "v"(b[3][1]), "v"(b[3][2]), "v"(b[3][3])
#define BODY MF(0, 4, 8) MF(1, 4, 12) MF(2, 4, 16) MF(3, 4, 20) MF(0, 5, 9) MF(1, 5, 13) MF(2, 5, 17) MF(3, 5, 21) MF(0, 6, 10) MF(1, 6, 14) MF(2, 6, 18) MF(3, 6, 22) MF(0, 7, 11) MF(1, 7, 15) MF(2, 7, 19) MF(3, 7, 23)
    if constexpr (AG)
        asm volatile(BODY : OPS : "a"(w0), "a"(w1), "a"(w2), "a"(w3), BS : "memory");
    else
        asm volatile(BODY : OPS : "v"(w0), "v"(w1), "v"(w2), "v"(w3), BS : "memory");
}

template <bool AG, bool SPLIT, int LOADS, bool STORES, bool LDS2>
__global__ __launch_bounds__(256, 1) void kpipe(float* out, const float* wsrc, const float* xsrc, float* ydst, Stamp* st, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = 68, NBLK = 36, NW = 144;
    constexpr int BUF = 300 * PS;
    for (int i = threadIdx.x; i < 300 * PS; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    float w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = wsrc[i * 64 + lane];
    if (AG) {
#pragma unroll
        for (int i = 0; i < NW; ++i) { float t = w[i]; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(w[i]) : "v"(t)); }
    }
    f32x4 acc[4];
    int laddr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; laddr[i] = (16 * i + li) * PS + 4 * kq; }
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc), 0, 1 << 24, 0x00020000);
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(ydst, 0, 1 << 24, 0x00020000);
    int voff = (blockIdx.x * 256 + threadIdx.x) * 16 & ((1 << 24) - 16);
    const int woff = (LDS2 ? BUF : 0) + (threadIdx.x >> 4) * PS + 4 * (threadIdx.x & 15);
    const float* lrd = lds;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 cur[4], nxt[4];
        f32x4 stg[LOADS > 0 ? LOADS : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const f32x4*>(lrd + laddr[i]);
#pragma unroll
        for (int t = 0; t < NBLK; ++t) {
            if (t + 1 < NBLK) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    nxt[i] = *reinterpret_cast<const f32x4*>(lrd + laddr[i] + ((t + 1) / 4) * PS + 16 * ((t + 1) % 4));
            }
            if (t < LOADS) {
                stg[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 0, 0));
                voff = (voff + 256 * 64 * 16) & ((1 << 24) - 16);
            }
            if (t >= 18 && t < 18 + LOADS)
                *reinterpret_cast<f32x4*>(lds + woff + (t - 18) * 16 * PS) = stg[t - 18];
            if (STORES && t >= 8 && t < 12) {
                f32x4 v = acc[t - 8];   // (garbage: mid-accumulation values; only the traffic matters)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), yr, voff, 0, 0);
            }
            if (SPLIT) {
#pragma unroll
                for (int s = 0; s < 4; ++s) sub4<AG>(acc, w[4 * t + s], cur[0][s], cur[1][s], cur[2][s], cur[3][s]);
            } else {
                blk16<AG>(acc, w[4 * t], w[4 * t + 1], w[4 * t + 2], w[4 * t + 3], cur);
            }
            if (t + 1 < NBLK) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 t = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, r0, r1};
}

template <typename K>
void run(const char* name, K kern, size_t lds_bytes) {
    const int grid = 256, iters = 200;
    float *out, *w, *x, *y; Stamp* st;
    hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&w, 144 * 64 * 4); hipMemset(w, 0, 144 * 64 * 4);
    hipMalloc(&x, 1 << 24); hipMemset(x, 0, 1 << 24); hipMalloc(&y, 1 << 24);
    hipMalloc(&st, grid * sizeof(Stamp));
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, 0, out, w, x, y, st, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid); hipMemcpy(h.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (auto& s : h) { clk.push_back((double)(s.clk1 - s.clk0) / (double)(s.rt1 - s.rt0) * 100.0); cyc.push_back((double)(s.clk1 - s.clk0)); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    double n = 576.0 * iters;
    printf("%-56s %8.3f ms  clk %5.0f MHz  cycles/MFMA %.2f\n", name, ms, clk[clk.size() / 2], cyc[cyc.size() / 2] / n);
    hipFree(out); hipFree(w); hipFree(x); hipFree(y); hipFree(st);
}

int main() {
    const size_t L1 = 300 * 68 * 4, L2 = 2 * L1;
    run("VGPR weights, 16-MFMA blocks", kpipe<false, false, 0, false, false>, L1);
    run("AGPR weights, 16-MFMA blocks", kpipe<true, false, 0, false, false>, L1);
    run("AGPR weights, 4x4 statements", kpipe<true, true, 0, false, false>, L1);
    run("AGPR, 16-blocks, 160K LDS", kpipe<true, false, 0, false, true>, L2);
    run("AGPR, 16-blocks, +6 loads/ds_writes", kpipe<true, false, 6, false, true>, L2);
    run("AGPR, 16-blocks, +4 stores", kpipe<true, false, 0, true, true>, L2);
    run("AGPR, 16-blocks, +6 loads +4 stores", kpipe<true, false, 6, true, true>, L2);
    run("AGPR, 4x4, +6 loads +4 stores", kpipe<true, true, 6, true, true>, L2);
    return 0;
}
