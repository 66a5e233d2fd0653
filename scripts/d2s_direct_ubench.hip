// d2s_direct_ubench.hip -- diagnostic, companion of d2s_ubench.hip: can the sub-pixel map at [256,41,41,27] (r = 3) run
// WITHOUT the LDS pass, i.e. at a copy's occupancy (8 workgroups of 256 per CU) and in ONE round of requests?
//   X1  gather:  a lane owns one OUTPUT float4 = four 4-byte loads at the permuted addresses (the vector L1 merges the
//                lanes of an instruction per cache line), one 16-byte store
//   X2  scatter: a lane owns one INPUT float4 = one 16-byte load, four 4-byte stores at the permuted addresses
//   X3  gather with the loads of ONE float4 at a time in flight (the serialised copy's request pattern)
// every variant is checked against the index map on the host before it is timed.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/d2s_direct_ubench.hip -o scripts/d2s_direct_ubench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned W = 41, RC = 9, R = 3, ROW = W * RC, B = ROW * R;   // 369, 1107

struct Pos { unsigned blkbase, dy, w, j; };
// output float o -> its digits; src = blkbase + w * (R * RC) + dy * RC + j
__device__ __forceinline__ Pos split_out(unsigned o) {
    Pos p;
    const unsigned blk = o / B, oo = o - blk * B;
    p.blkbase = blk * B;
    p.dy = oo / ROW;
    const unsigned rem = oo - p.dy * ROW;
    p.w = rem / RC;
    p.j = rem - p.w * RC;
    return p;
}
// input float i -> digits (w, dy, j); dst = blkbase + dy * ROW + w * RC + j
__device__ __forceinline__ Pos split_in(unsigned i) {
    Pos p;
    const unsigned blk = i / B, ii = i - blk * B;
    p.blkbase = blk * B;
    p.w = ii / (R * RC);
    const unsigned rem = ii - p.w * (R * RC);
    p.dy = rem / RC;
    p.j = rem - p.dy * RC;
    return p;
}

// MODE 0: every load of the thread issued before the first store (counted waits); 1: one float4's loads at a time;
// 2: as 0 with nontemporal loads
template <int KMAX, int MODE>
__global__ __launch_bounds__(256) void gather_x(const float* __restrict__ in, float* __restrict__ out, unsigned nfl) {
    const unsigned bytes = nfl * 4;
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    const unsigned base = blockIdx.x * (KMAX * 1024u) + threadIdx.x * 4;
    u32x4v v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const unsigned o = base + k * 1024u;
        Pos p = split_out(o);
        unsigned src = p.blkbase + p.w * (R * RC) + p.dy * RC + p.j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[k][e] = __builtin_amdgcn_raw_buffer_load_b32(ri, src * 4, 0, MODE == 2 ? 2 : 0);
            // next output float: j carries into w, w into dy, dy into the next block
            ++p.j; ++src;
            if (p.j == RC) { p.j = 0; ++p.w; src += R * RC - RC;
                if (p.w == W) { p.w = 0; ++p.dy; src = p.blkbase + p.dy * RC;
                    if (p.dy == R) { p.dy = 0; p.blkbase += B; src = p.blkbase; } } }
        }
        if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, o * 4, 0, 2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    if (MODE != 1) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, (base + k * 1024u) * 4, 0, 2);
    }
}

// MODE 0: all loads, then all stores; 1: one float4 at a time
template <int KMAX, int MODE>
__global__ __launch_bounds__(256) void scatter_x(const float* __restrict__ in, float* __restrict__ out, unsigned nfl) {
    const unsigned bytes = nfl * 4;
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    const unsigned base = blockIdx.x * (KMAX * 1024u) + threadIdx.x * 4;
    u32x4v v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(ri, (base + k * 1024u) * 4, 0, 2);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const unsigned i = base + k * 1024u;
        Pos p = split_in(i);
        unsigned dst = p.blkbase + p.dy * ROW + p.w * RC + p.j;      // past the tensor: a block past the tensor, dropped
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __builtin_amdgcn_raw_buffer_store_b32(v[k][e], ro, dst * 4, 0, MODE == 2 ? 2 : 0);
            ++p.j; ++dst;
            if (p.j == RC) { p.j = 0; ++p.dy; dst += ROW - RC;
                if (p.dy == R) { p.dy = 0; ++p.w; dst = p.blkbase + p.w * RC;
                    if (p.w == W) { p.w = 0; p.blkbase += B; dst = p.blkbase; } } }
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

__global__ void fill_iota(unsigned* p, unsigned n) { for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = i * 2654435761u; }

int main() {
    const int P = 8;
    const unsigned nfl = 256u * 41 * 41 * 27;
    std::vector<float*> in(P), out(P);
    for (int i = 0; i < P; ++i) { CK(hipMalloc(&in[i], (size_t)nfl * 4)); CK(hipMalloc(&out[i], (size_t)nfl * 4)); hipLaunchKernelGGL(fill_iota, dim3(2048), dim3(256), 0, 0, (unsigned*)in[i], nfl); CK(hipMemset(out[i], 0, (size_t)nfl * 4)); }
    std::vector<unsigned> host(nfl);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto check = [&](const char* name) {
        CK(hipMemcpy(host.data(), out[0], (size_t)nfl * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (unsigned o = 0; o < nfl; ++o) {
            const unsigned blk = o / B, oo = o % B, dy = oo / ROW, rem = oo % ROW, w = rem / RC, j = rem % RC;
            const unsigned src = blk * B + w * R * RC + dy * RC + j;
            if (host[o] != src * 2654435761u) ++bad;
        }
        if (bad) printf("%s: %zu WRONG floats\n", name, bad);
        CK(hipMemset(out[0], 0, (size_t)nfl * 4));
        return bad == 0;
    };
    auto timeit = [&](const char* name, auto launch) {
        launch(0);
        CK(hipDeviceSynchronize());
        if (!check(name)) return;
        for (int i = 0; i < 2 * P; ++i) launch(i % P);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 80; ++i) launch(i % P);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 80 < best) best = ms / 80;
        }
        CK(hipGetLastError());
        printf("%-78s %6.2f us  %5.2f TB/s\n", name, best * 1e3, 2.0 * nfl * 4 / (best * 1e-3) / 1e12);
    };
    auto grid = [&](int k) { return dim3((nfl + k * 1024 - 1) / (k * 1024)); };
#define RUN(name, kern, K) timeit(name, [&](int i) { hipLaunchKernelGGL(kern, grid(K), dim3(256), 0, 0, in[i], out[i], nfl); })
    RUN("X1 gather, 4 float4/thread, all loads then all stores", (gather_x<4, 0>), 4);
    RUN("X1 gather, 6 float4/thread", (gather_x<6, 0>), 6);
    RUN("X1 gather, 8 float4/thread", (gather_x<8, 0>), 8);
    RUN("X1n gather, 6 float4/thread, nontemporal loads", (gather_x<6, 2>), 6);
    RUN("X3 gather, 6 float4/thread, one float4 at a time", (gather_x<6, 1>), 6);
    RUN("X3 gather, 8 float4/thread, one float4 at a time", (gather_x<8, 1>), 8);
    RUN("X2 scatter, 4 float4/thread, all loads then all stores", (scatter_x<4, 0>), 4);
    RUN("X2 scatter, 6 float4/thread", (scatter_x<6, 0>), 6);
    RUN("X2 scatter, 8 float4/thread", (scatter_x<8, 0>), 8);
    RUN("X2n scatter, 6 float4/thread, nontemporal stores", (scatter_x<6, 2>), 6);
    RUN("X2s scatter, 6 float4/thread, one float4's stores at a time", (scatter_x<6, 1>), 6);
    return 0;
}
