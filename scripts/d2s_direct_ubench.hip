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


// ---- is the map's chunking paying for its IMBALANCE?  2,624 chunks over 1,024 persistent workgroups = 3 chunks for 576
// of them and 2 for the rest.  The map's structure as a copy (d2s_ubench's V3/W3: depth-1 prefetch, at most 3 requests
// per wave, optionally through LDS with an identity gather), chunk c of workgroup g = g + c * G (STRIDED) or
// g * per + c (CONTIG: every workgroup exactly `per` chunks).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int KMAX, int THR, bool LDS, bool CONTIG>
__global__ __launch_bounds__(256) void chunked_bal(const float* __restrict__ in, float* __restrict__ out, int nchunks, int chunk_floats, int per) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = gridDim.x;
    int c = CONTIG ? blockIdx.x * per : blockIdx.x;
    const int cend = CONTIG ? min(c + per, nchunks) : nchunks, cstep = CONTIG ? 1 : G;
    if (c >= cend) return;
    const int c4 = chunk_floats >> 2;
    auto throttle = [&]() {
        if (THR == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (THR == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (THR == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (THR == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };
    f32x4 v[KMAX];
    int idx[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { const int i = k * 256 + threadIdx.x; idx[k] = i < c4 ? i : c4 - 1; }
    auto issue = [&](int chunk) {
        const f32x4* src = reinterpret_cast<const f32x4*>(in) + (size_t)chunk * c4;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { v[k] = __builtin_nontemporal_load(src + idx[k]); throttle(); }
    };
    issue(c);
    int p = 0;
    for (; c < cend; c += cstep, p ^= 1) {
        f32x4 w[KMAX];
        if (LDS) {
            f32x4* buf4 = reinterpret_cast<f32x4*>(lds + p * chunk_floats);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) buf4[idx[k]] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) w[k] = v[k];
        }
        if (c + cstep < cend) issue(c + cstep);
        if (LDS) {
            lds_barrier();
            const float* buf = lds + p * chunk_floats;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) w[k][e] = buf[4 * idx[k] + e];
        }
        f32x4* dst = reinterpret_cast<f32x4*>(out) + (size_t)c * c4;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k * 256 + (int)threadIdx.x < c4) __builtin_nontemporal_store(w[k], dst + idx[k]);
            throttle();
        }
    }
}


// plain one-round copy (d2s_ubench's A1) for the alignment question below
template <int KMAX>
__global__ __launch_bounds__(256) void copy_a(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4) {
    const size_t per = (size_t)KMAX * 256;
    for (size_t base = (size_t)blockIdx.x * per; base < n4; base += (size_t)gridDim.x * per) {
        f32x4 v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n4) v[k] = __builtin_nontemporal_load(in + i); }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n4) __builtin_nontemporal_store(v[k], out + i); }
    }
}
// the same with bounds-checked buffer loads whose base is only 4-byte aligned
template <int KMAX>
__global__ __launch_bounds__(256) void copy_buf_src(const float* __restrict__ in, float* __restrict__ out, unsigned bytes) {
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    const unsigned per = KMAX * 4096u;
    for (unsigned base = blockIdx.x * per; base < bytes; base += gridDim.x * per) {
        u32x4v v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { v[k] = __builtin_amdgcn_raw_buffer_load_b128(ri, base + k * 4096 + threadIdx.x * 16, 0, 2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, base + k * 4096 + threadIdx.x * 16, 0, 2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
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
    if (getenv("D2S_DIRECT")) {
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
    }

    {   // imbalance: copies in the map's structure (no permutation: not checked against the map)
        auto timecopy = [&](const char* name, auto launch) {
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
        const int ca = 4608, na = (int)(nfl / ca);                 // 2,521 line-aligned chunks: 2.46 per workgroup, 3 at most
        const size_t la = (size_t)ca * 8 + 4096;
        timecopy("S3  strided chunks of 4608 floats, 1024 wgs (2 or 3 chunks each), no LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<5, 3, false, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca, 0); });
        timecopy("S3l the same through LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<5, 3, true, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca, 0); });
        const int cb = 3776, nb = 3 * 1024;                         // 944 float4 (118 lines) x 3 per workgroup: 99.8 % of the tensor
        timecopy("C3  contiguous: 3 chunks of 3776 floats for EVERY one of 1024 wgs, no LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<4, 3, false, true>), dim3(1024), dim3(256), la, 0, in[i], out[i], nb, cb, 3); });
        timecopy("C3l the same through LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<4, 3, true, true>), dim3(1024), dim3(256), la, 0, in[i], out[i], nb, cb, 3); });
        timecopy("C3s 3 chunks of 3776 floats each, STRIDED over 1024 wgs, no LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<4, 3, false, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], nb, cb, 0); });
        timecopy("C3sl the same through LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<4, 3, true, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], nb, cb, 0); });
        const int cc = 2832, nc = 4 * 1024;                         // 708 float4 x 4 per workgroup (99.8 %)
        timecopy("C4s 4 chunks of 2832 floats each, strided, no LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<3, 3, false, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], nc, cc, 0); });
        timecopy("C4sl the same through LDS", [&](int i) { hipLaunchKernelGGL((chunked_bal<3, 3, true, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], nc, cc, 0); });
        const int cd = 5664, nd = 2 * 1024;                         // 1416 float4 x 2 per workgroup
        const size_t ld = (size_t)cd * 8 + 4096;
        timecopy("C2s 2 chunks of 5664 floats each, strided, no LDS (4 wgs per CU)", [&](int i) { hipLaunchKernelGGL((chunked_bal<6, 3, false, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], nd, cd, 0); });
        (void)ld;

        // 16-byte loads from a source that is only 4-byte aligned (chunks of whole blocks of 4428 bytes start 0/4/8/12 bytes into a float4)
        timecopy("N0 copy A1 (8 float4/thread, one round), global ops, aligned", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], (size_t)nfl / 4 - 8); });
        timecopy("N1 ... source 4 bytes into a float4", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)(in[i] + 1), (f32x4*)out[i], (size_t)nfl / 4 - 8); });
        timecopy("N2 ... source 12 bytes into a float4", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)(in[i] + 3), (f32x4*)out[i], (size_t)nfl / 4 - 8); });
        timecopy("N3 serialised buffer-op copy (6 float4/thread, 1892 wgs), aligned", [&](int i) { hipLaunchKernelGGL(copy_buf_src<6>, dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4 - 128)); });
        timecopy("N4 ... source base 4 bytes into a float4", [&](int i) { hipLaunchKernelGGL(copy_buf_src<6>, dim3(1892), dim3(256), 0, 0, in[i] + 1, out[i], (unsigned)(nfl * 4 - 128)); });
        timecopy("N5 ... source base 8 bytes into a float4", [&](int i) { hipLaunchKernelGGL(copy_buf_src<6>, dim3(1892), dim3(256), 0, 0, in[i] + 2, out[i], (unsigned)(nfl * 4 - 128)); });
    }
    return 0;
}
