// mfma_ubench.hip -- diagnostic only (not part of libsrx): what does a bare fp32 MFMA stream reach
// on this MI355X, and at what clock?  Variants: 16x16x4 vs 32x32x2; 1 or 2 waves per SIMD; with or
// without one ds_read_b128 per 4 MFMAs.  Prints TFLOP/s and the in-kernel clock
// (delta s_memtime / delta s_memrealtime * 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_ubench.hip -o /tmp/mfma_ubench && /tmp/mfma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long clk0, clk1, rt0, rt1; };

template <int NACC, bool LDS>
__global__ __launch_bounds__(256, 2) void k16(float* out, Stamp* st, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 68];
    for (int i = threadIdx.x; i < 64 * 68; i += 256) lds[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w[8];
    for (int i = 0; i < 8; ++i) w[i] = (float)(((lane * 8 + i) * 2246822519u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    f32x4 b = {w[1], w[5], w[2], w[7]};
    const float* lp = lds + (lane & 15) * 68 + (lane >> 4) * 4;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (LDS) b = *reinterpret_cast<const f32x4*>(lp + 16 * (u & 3) + 68 * 16 * (u >> 2));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < NACC; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(u + s) & 7], b[s], acc[i], 0, 0, 0);
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 t = acc[0];
    for (int i = 1; i < NACC; ++i) t += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, r0, r1};
}

template <int NACC>
__global__ __launch_bounds__(256, 2) void k32(float* out, Stamp* st, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float w[8];
    for (int i = 0; i < 8; ++i) w[i] = (float)(((lane * 8 + i) * 2246822519u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u], w[(u + 3) & 7], acc[i], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) t += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = t;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, r0, r1};
}

template <typename K>
void run(const char* name, K kern, int grid, int iters, double flop_per_wave_iter) {
    float* out; Stamp* st;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipMalloc(&st, grid * sizeof(Stamp));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, st, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid);
    hipMemcpy(h.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (auto& s : h) { clk.push_back((double)(s.clk1 - s.clk0) / (double)(s.rt1 - s.rt0) * 100.0); cyc.push_back((double)(s.clk1 - s.clk0)); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    double flops = flop_per_wave_iter * iters * 4.0 * grid;
    printf("%-34s grid %4d  %8.3f ms  %7.1f TFLOP/s  clock(median) %6.0f MHz  loop cycles(median) %.0f\n", name, grid, ms,
           flops / (ms * 1e-3) / 1e12, clk[clk.size() / 2], cyc[cyc.size() / 2]);
    hipFree(out); hipFree(st);
}

// sustained: back-to-back launches for ~3 s with random-ish operands, then report the last launch
template <typename K>
void sustained(const char* name, K kern, int grid, int iters, double flop_per_wave_iter, double seconds) {
    float* out; Stamp* st;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipMalloc(&st, grid * sizeof(Stamp));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, st, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    int n = (int)(seconds * 1e3 / ms1) + 1;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, st, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, st, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid);
    hipMemcpy(h.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (auto& s : h) clk.push_back((double)(s.clk1 - s.clk0) / (double)(s.rt1 - s.rt0) * 100.0);
    std::sort(clk.begin(), clk.end());
    double flops = flop_per_wave_iter * iters * 4.0 * grid;
    printf("SUSTAINED %-28s first %.3f ms -> after %.1f s: %.3f ms  %7.1f TFLOP/s  clock(median) %6.0f MHz\n", name, ms1,
           seconds, ms, flops / (ms * 1e-3) / 1e12, clk[clk.size() / 2]);
    hipFree(out); hipFree(st);
}

int main() {
    const int iters = 4000;
    sustained("16x16x4 4acc 2w/SIMD", k16<4, false>, 512, iters, 8 * 4 * 4 * 2048.0, 3.0);
    sustained("16x16x4 4acc +ds_read 2w/SIMD", k16<4, true>, 512, iters, 8 * 4 * 4 * 2048.0, 3.0);
    sustained("32x32x2 4acc 2w/SIMD", k32<4>, 512, iters, 8 * 4 * 4096.0, 3.0);
    // the conv kernels' LDS read density: one ds_read_b128 per 4 MFMAs (1 accumulator per read here)
    sustained("16x16x4 1acc +ds_read/4MFMA 2w/SIMD", k16<1, true>, 512, iters, 8 * 4 * 1 * 2048.0, 3.0);
    sustained("16x16x4 1acc no LDS 2w/SIMD", k16<1, false>, 512, iters, 8 * 4 * 1 * 2048.0, 3.0);
    // per wave per iter: 8 u * 4 s * NACC MFMAs * 2048 flop
    run("16x16x4 4acc 1wave/SIMD", k16<4, false>, 256, iters, 8 * 4 * 4 * 2048.0);
    run("16x16x4 4acc 2waves/SIMD", k16<4, false>, 512, iters, 8 * 4 * 4 * 2048.0);
    run("16x16x4 3acc 2waves/SIMD", k16<3, false>, 512, iters, 8 * 4 * 3 * 2048.0);
    run("16x16x4 1acc 2waves/SIMD", k16<1, false>, 512, iters, 8 * 4 * 1 * 2048.0);
    run("16x16x4 4acc +ds_read 1wave/SIMD", k16<4, true>, 256, iters, 8 * 4 * 4 * 2048.0);
    run("16x16x4 4acc +ds_read 2waves/SIMD", k16<4, true>, 512, iters, 8 * 4 * 4 * 2048.0);
    run("32x32x2 4acc 1wave/SIMD", k32<4>, 256, iters, 8 * 4 * 4096.0);
    run("32x32x2 4acc 2waves/SIMD", k32<4>, 512, iters, 8 * 4 * 4096.0);
    run("32x32x2 2acc 2waves/SIMD", k32<2>, 512, iters, 8 * 2 * 4096.0);
    return 0;
}
