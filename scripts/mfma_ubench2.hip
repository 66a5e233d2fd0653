// mfma_ubench2.hip -- diagnostic: the conv inner loop (G accumulators, NW stationary weight
// registers, one ds_read_b128 per accumulator per 4 MFMAs) with ingredients toggled.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long clk0, clk1, rt0, rt1; };

// MODE bit0: read LDS every block (else reuse registers); bit1: padded stride 68 (else 64: conflicts);
// bit2: rotate the B operand registers (cur/nxt prefetch as in the real kernel)
template <int G, int NW, int MODE>
__global__ __launch_bounds__(256, 2) void kconv(float* out, const float* wsrc, Stamp* st, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PS = (MODE & 2) ? 68 : 64;
    for (int i = threadIdx.x; i < 300 * PS; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    float w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = wsrc[i * 64 + lane];
    f32x4 acc[G];
    int laddr[G];
#pragma unroll
    for (int i = 0; i < G; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; laddr[i] = (16 * i + li) * PS + 4 * kq; }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 cur[G], nxt[G];
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i]);
#pragma unroll
        for (int t = 0; t < NW / 4; ++t) {
            if ((MODE & 1) && t + 1 < NW / 4) {
#pragma unroll
                for (int i = 0; i < G; ++i)
                    nxt[i] = *reinterpret_cast<const f32x4*>(lds + laddr[i] + ((t + 1) / 4) * PS + 16 * ((t + 1) % 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < G; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * t + s], cur[i][s], acc[i], 0, 0, 0);
            if (MODE & 1) {
#pragma unroll
                for (int i = 0; i < G; ++i) cur[i] = nxt[i];
            }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 t = acc[0];
    for (int i = 1; i < G; ++i) t += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, r0, r1};
}

template <typename K>
void run(const char* name, K kern, int grid, int iters, double mfma_per_wave_iter) {
    float *out, *w; Stamp* st;
    hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&w, 144 * 64 * 4); hipMemset(w, 0, 144 * 64 * 4);
    hipMalloc(&st, grid * sizeof(Stamp));
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 300 * 68 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 300 * 68 * 4, 0, out, w, st, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid); hipMemcpy(h.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (auto& s : h) { clk.push_back((double)(s.clk1 - s.clk0) / (double)(s.rt1 - s.rt0) * 100.0); cyc.push_back((double)(s.clk1 - s.clk0)); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    double n = mfma_per_wave_iter * iters;
    printf("%-44s grid %4d %8.3f ms %7.1f TF  clk %5.0f MHz  cycles/MFMA(per SIMD) %.2f\n", name, grid, ms,
           n * 2048.0 * 4 * grid / (ms * 1e-3) / 1e12, clk[clk.size() / 2], cyc[cyc.size() / 2] / n / (grid / 256));
    hipFree(out); hipFree(w); hipFree(st);
}

int main() {
    const int it = 300;
    run("G4 NW144 noLDS", kconv<4, 144, 0>, 512, it, 144.0 * 4);
    run("G4 NW144 LDS stride64 (conflicts)", kconv<4, 144, 1>, 512, it, 144.0 * 4);
    run("G4 NW144 LDS stride68", kconv<4, 144, 3>, 512, it, 144.0 * 4);
    run("G4 NW144 LDS stride68 1wave/SIMD", kconv<4, 144, 3>, 256, it, 144.0 * 4);
    run("G3 NW144 LDS stride68", kconv<3, 144, 3>, 512, it, 144.0 * 3);
    run("G4 NW16 LDS stride68", kconv<4, 16, 3>, 512, it * 9, 16.0 * 4);
    run("G4 NW16 noLDS", kconv<4, 16, 0>, 512, it * 9, 16.0 * 4);
    return 0;
}
