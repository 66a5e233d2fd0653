// d2s_ubench.hip -- diagnostic: where do the ~1.8 us between a plain streaming copy (15.6 us) and the sub-pixel map
// (17.4 us) at [256,41,41,27] (46.5 MB in, 46.5 MB out) go?  The map's structure is rebuilt step by step around a copy:
//   A  grid-stride copy, 8 float4 per thread in flight, 2048 workgroups                      (= srx_stream_copy)
//   B  the map's chunking: 1024 persistent workgroups, chunks of 4428 floats (5 float4 slots per thread, the last one
//      ragged), next chunk's loads issued before the current chunk's stores, bounds-checked buffer operations, NO LDS
//   C  B + every chunk written to LDS (b128) and read back with 4 x ds_read_b32 per float4 at the IDENTITY index
//      (the map's LDS traffic and barrier without its index arithmetic and bank pattern)
//   D  C with the map's real gather indices read from a table in global memory (no index arithmetic in the kernel)
//   E  one chunk per workgroup (2624 workgroups), no LDS
// Build: hipcc -O3 --offload-arch=gfx950 scripts/d2s_ubench.hip -o scripts/d2s_ubench.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((__vector_size__(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KMAX, bool NT = true>
__global__ __launch_bounds__(256) void copy_a(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4) {
    const size_t per = (size_t)KMAX * 256;
    for (size_t base = (size_t)blockIdx.x * per; base < n4; base += (size_t)gridDim.x * per) {
        f32x4 v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n4) v[k] = NT ? __builtin_nontemporal_load(in + i) : in[i]; }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < n4) { if (NT) __builtin_nontemporal_store(v[k], out + i); else out[i] = v[k]; } }
    }
}

// MODE 0: no LDS (B / E)   1: LDS, identity gather (C)   2: LDS, gather through a table (D)
template <int KMAX, int MODE, int AUX = 2, bool CONTIG = false>
__global__ __launch_bounds__(256) void chunked(const float* __restrict__ in, float* __restrict__ out, int nfull, int chunk_floats,
                                               const int* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // CONTIG: a workgroup owns a contiguous range of chunks instead of every G-th one
    const int per = (nfull + gridDim.x - 1) / gridDim.x;
    int c = CONTIG ? blockIdx.x * per : blockIdx.x;
    const int cend = CONTIG ? (c + per < nfull ? c + per : nfull) : nfull;
    if (c >= nfull) return;
    const int G = CONTIG ? 1 : gridDim.x;
    const int c4 = chunk_floats >> 2;
    const unsigned chunk_bytes = (unsigned)chunk_floats * 4u;
    const int lane_off = threadIdx.x * 16;
    auto rsrc_at = [&](const float* base, int chunk) {
        const bool ok = chunk < cend;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (size_t)(ok ? chunk : 0) * chunk_floats), 0, ok ? chunk_bytes : 0u, 0x00020000);
    };
    u32x4v v[KMAX];
    auto issue = [&](int chunk) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_at(in, chunk);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + k * 4096, 0, AUX);
    };
    issue(c);
    {
        const __amdgpu_buffer_rsrc_t rs = rsrc_at(out, cend);
        const u32x4v z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < KMAX; ++k) __builtin_amdgcn_raw_buffer_store_b128(z, rs, lane_off + k * 4096, 0, AUX);
    }
    int sidx[KMAX][4];
    if (MODE) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = k * 256 + threadIdx.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) sidx[k][e] = i < c4 ? (MODE == 2 ? table[4 * i + e] : 4 * i + e) : 0;
        }
    }
    const int last_i = (KMAX - 1) * 256 + threadIdx.x;
    const bool last_in = last_i < c4;
    int p = 0;
    for (; c < cend; c += G, p ^= 1) {
        const __amdgpu_buffer_rsrc_t ro = rsrc_at(out, c);
        if (MODE == 0) {
            u32x4v w[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) w[k] = v[k];
            issue(c + G);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) __builtin_amdgcn_raw_buffer_store_b128(w[k], ro, lane_off + k * 4096, 0, AUX);
        } else {
            float* buf = lds + p * chunk_floats;
            u32x4v* buf4 = reinterpret_cast<u32x4v*>(buf);
#pragma unroll
            for (int k = 0; k < KMAX - 1; ++k) buf4[k * 256 + threadIdx.x] = v[k];
            reinterpret_cast<u32x4v*>(lds)[last_in ? p * c4 + last_i : 2 * c4 + threadIdx.x] = v[KMAX - 1];
            issue(c + G);
            lds_barrier();
            float g[KMAX][4];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) g[k][e] = buf[sidx[k][e]];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                f32x4 o; o[0] = g[k][0]; o[1] = g[k][1]; o[2] = g[k][2]; o[3] = g[k][3];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o), ro, lane_off + k * 4096, 0, AUX);
            }
        }
    }
}

// every chunk of the workgroup (at most DEPTH) requested before the first one is touched; MODE as in `chunked`
template <int KMAX, int DEPTH, int MODE, bool CONTIG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void upfront(const float* __restrict__ in, float* __restrict__ out, int nfull, int chunk_floats, const int* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int c = CONTIG ? blockIdx.x * DEPTH : blockIdx.x, G = CONTIG ? 1 : gridDim.x;   // CONTIG: the workgroup's chunks are neighbours
    if (c >= nfull) return;
    const int c4 = chunk_floats >> 2;
    const unsigned chunk_bytes = (unsigned)chunk_floats * 4u;
    const int lane_off = threadIdx.x * 16;
    auto rsrc_at = [&](const float* base, int chunk) {
        const bool ok = chunk < nfull;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (size_t)(ok ? chunk : 0) * chunk_floats), 0, ok ? chunk_bytes : 0u, 0x00020000);
    };
    u32x4v v[DEPTH][KMAX];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_at(in, c + d * G);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v[d][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + k * 4096, 0, 2);
    }
    unsigned short sidx[KMAX][4];
    if (MODE) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = k * 256 + threadIdx.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) sidx[k][e] = (unsigned short)(i < c4 ? (MODE == 2 ? table[4 * i + e] : 4 * i + e) : 0);
        }
    }
    const int last_i = (KMAX - 1) * 256 + threadIdx.x;
    const bool last_in = last_i < c4;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const __amdgpu_buffer_rsrc_t ro = rsrc_at(out, c + d * G);
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) __builtin_amdgcn_raw_buffer_store_b128(v[d][k], ro, lane_off + k * 4096, 0, 2);
        } else {
            const int p = d & 1;
            float* buf = lds + p * chunk_floats;
            u32x4v* buf4 = reinterpret_cast<u32x4v*>(buf);
#pragma unroll
            for (int k = 0; k < KMAX - 1; ++k) buf4[k * 256 + threadIdx.x] = v[d][k];
            reinterpret_cast<u32x4v*>(lds)[last_in ? p * c4 + last_i : 2 * c4 + threadIdx.x] = v[d][KMAX - 1];
            lds_barrier();
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = buf[sidx[k][e]];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o), ro, lane_off + k * 4096, 0, 2);
            }
        }
    }
}

// copy_a with bounds-checked buffer operations instead of global ones (one resource over the whole tensor)
template <int KMAX, int FLAGS>
__global__ __launch_bounds__(256) void copy_buf(const float* __restrict__ in, float* __restrict__ out, unsigned bytes) {
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, bytes, FLAGS);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, FLAGS);
    const unsigned per = KMAX * 4096u;
    for (unsigned base = blockIdx.x * per; base < bytes; base += gridDim.x * per) {
        u32x4v v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(ri, base + k * 4096 + threadIdx.x * 16, 0, 2);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, base + k * 4096 + threadIdx.x * 16, 0, 2);
    }
}

// which side pays for buffer operations?  MODE 1: buffer loads + global stores; 2: global loads + buffer stores
template <int KMAX, int MODE>
__global__ __launch_bounds__(256) void copy_mix(const float* __restrict__ in, float* __restrict__ out, unsigned bytes) {
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    const unsigned per = KMAX * 4096u;
    for (unsigned base = blockIdx.x * per; base < bytes; base += gridDim.x * per) {
        u32x4v v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned off = base + k * 4096 + threadIdx.x * 16;
            if (MODE == 1) v[k] = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 2);
            else v[k] = __builtin_bit_cast(u32x4v, __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(in) + off)));
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned off = base + k * 4096 + threadIdx.x * 16;
            if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, off, 0, 2);
            else __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v[k]), reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off));
        }
    }
}

// How many requests per wave should be in flight?  copy_a (the fastest copy) turns out to be fully serialised by the
// compiler (per-lane predicates -> s_waitcnt vmcnt(0) in front of EVERY load and store).  WAITS 1: the same explicitly
// (one request per wave at a time); 2: all loads, wait, all stores, wait; 3: loads in flight two at a time, stores one
// at a time; 0: counted waits (store k as soon as load k has landed)
template <int KMAX, int WAITS>
__global__ __launch_bounds__(256) void copy_thr(const float* __restrict__ in, float* __restrict__ out, unsigned bytes) {
    const unsigned per = KMAX * 4096u;
    for (unsigned base = blockIdx.x * per; base < bytes; base += gridDim.x * per) {
        f32x4 v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned off = base + k * 4096 + threadIdx.x * 16;
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(in) + off));
            if (WAITS == 1 || (WAITS == 3 && (k & 1))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (WAITS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned off = base + k * 4096 + threadIdx.x * 16;
            __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + off));
            if (WAITS == 1 || WAITS == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (WAITS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// the map's structure (persistent workgroups, 4 per CU, chunks of `chunk_floats`, next chunk's loads issued before the
// current chunk's stores) with the number of requests a wave keeps in flight capped at THR (0 = counted waits only)
template <int KMAX, int THR, bool LDS>
__global__ __launch_bounds__(256) void chunked_thr(const float* __restrict__ in, float* __restrict__ out, int nfull, int chunk_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int c = blockIdx.x;
    if (c >= nfull) return;
    const int G = gridDim.x;
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
    for (int k = 0; k < KMAX; ++k) { const int i = k * 256 + threadIdx.x; idx[k] = i < c4 ? i : c4 - 1; }   // clamped: duplicates, no predicate
    auto issue = [&](int chunk) {
        const f32x4* src = reinterpret_cast<const f32x4*>(in) + (size_t)chunk * c4;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { v[k] = __builtin_nontemporal_load(src + idx[k]); throttle(); }
    };
    issue(c);
    int p = 0;
    for (; c < nfull; c += G, p ^= 1) {
        f32x4 w[KMAX];
        if (LDS) {
            f32x4* buf4 = reinterpret_cast<f32x4*>(lds + p * chunk_floats);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) buf4[idx[k]] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) w[k] = v[k];
        }
        if (c + G < nfull) issue(c + G);
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

int main() {
    const int P = 8;
    const size_t nfl = (size_t)256 * 41 * 41 * 27;
    const int chunk = 4 * 41 * 3 * 9;   // 4428 floats
    const int nfull = (int)(nfl / chunk);
    std::vector<float*> in(P), out(P);
    for (int i = 0; i < P; ++i) { CK(hipMalloc(&in[i], nfl * 4)); CK(hipMalloc(&out[i], nfl * 4)); CK(hipMemset(in[i], 1, nfl * 4)); CK(hipMemset(out[i], 0, nfl * 4)); }
    // the map's gather table of one chunk (depth-to-space, W = 41, rC = 9, r = 3)
    std::vector<int> tab(chunk);
    { const int W = 41, rC = 9, r = 3, B = W * r * rC;
      for (int o = 0; o < chunk; ++o) { int blk = o / B, oo = o % B, dy = oo / (W * rC), rest = oo % (W * rC), w = rest / rC, j = rest % rC; tab[o] = blk * B + w * r * rC + dy * rC + j; } }
    int* dtab; CK(hipMalloc(&dtab, chunk * 4)); CK(hipMemcpy(dtab, tab.data(), chunk * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds2 = (size_t)chunk * 8 + 4096;
    auto timeit = [&](const char* name, auto launch) {
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
        printf("%-70s %6.2f us  %5.2f TB/s\n", name, best * 1e3, 2.0 * nfl * 4 / (best * 1e-3) / 1e12);
    };
    timeit("A  grid-stride copy, 8 float4/thread, 2048 wgs", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(2048), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], nfl / 4); });
    timeit("A' grid-stride copy, 4 float4/thread, 2048 wgs", [&](int i) { hipLaunchKernelGGL(copy_a<4>, dim3(2048), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], nfl / 4); });
    timeit("A'' grid-stride copy, 8 float4/thread, 1024 wgs", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1024), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], nfl / 4); });
    timeit("B  chunked 4428 floats, 1024 persistent wgs, prefetch 1, no LDS", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(1024), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("B' the same with the map's LDS allocation (4 wgs per CU)", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(1024), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("C  B' + through LDS, identity gather", [&](int i) { hipLaunchKernelGGL((chunked<5, 1>), dim3(1024), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("D  B' + through LDS, the map's gather indices from a table", [&](int i) { hipLaunchKernelGGL((chunked<5, 2>), dim3(1024), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("E  one chunk per workgroup (2624 wgs), no LDS", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(nfull), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("E' one chunk per workgroup, 22 KB of LDS each (7 per CU)", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(nfull), dim3(256), 22 * 1024, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("A1 copy, 8 float4/thread, 1419 wgs (exactly one round, no idle wg)", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], nfl / 4); });
    timeit("A2 copy, 8 float4/thread, 2048 wgs, plain (not nontemporal) loads and stores", [&](int i) { hipLaunchKernelGGL((copy_a<8, false>), dim3(2048), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], nfl / 4); });
    timeit("A3 copy, 6 float4/thread, 1892 wgs (one round)", [&](int i) { hipLaunchKernelGGL(copy_a<6>, dim3(1892), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], nfl / 4); });
    timeit("A4 copy, 12 float4/thread, 946 wgs (one round)", [&](int i) { hipLaunchKernelGGL(copy_a<12>, dim3(946), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)out[i], nfl / 4); });
    timeit("B2 B with plain (not nontemporal) accesses", [&](int i) { hipLaunchKernelGGL((chunked<5, 0, 0>), dim3(1024), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("B3 B, each workgroup a contiguous range of chunks", [&](int i) { hipLaunchKernelGGL((chunked<5, 0, 2, true>), dim3(1024), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("B4 B with 875 wgs (3 chunks each)", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(875), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("B5 B with 1312 wgs (2 chunks each, 5.1 per CU)", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(1312), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("D3 D, contiguous ranges", [&](int i) { hipLaunchKernelGGL((chunked<5, 2, 2, true>), dim3(1024), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("U0 all 3 chunks of a workgroup requested up front, 1024 wgs, no LDS", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 0>), dim3(1024), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("U0' the same, 875 wgs (3 chunks each)", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 0>), dim3(875), dim3(256), 0, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("U1 up front + through LDS, identity gather", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 1>), dim3(1024), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("U2 up front + through LDS, the map's gather from a table", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 2>), dim3(1024), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("U2' the same, 875 wgs", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 2>), dim3(875), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    timeit("M1 copy A1 (1419 wgs), source 48 bytes off a 128-byte line", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)(in[i] + 12), (f32x4*)out[i], nfl / 4 - 8); });
    timeit("M2 copy A1, destination 48 bytes off", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)in[i], (f32x4*)(out[i] + 12), nfl / 4 - 8); });
    timeit("M3 copy A1, both 48 bytes off", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)(in[i] + 12), (f32x4*)(out[i] + 12), nfl / 4 - 8); });
    timeit("M4 copy A1, both 64 bytes off", [&](int i) { hipLaunchKernelGGL(copy_a<8>, dim3(1419), dim3(256), 0, 0, (const f32x4*)(in[i] + 16), (f32x4*)(out[i] + 16), nfl / 4 - 8); });
    {   // line-aligned chunks: 4608 floats = 144 lines of 128 bytes (the map's chunks are 138.375 lines)
        const int ca = 4608, na = (int)(nfl / ca);
        const size_t la = (size_t)ca * 8 + 4096;
        timeit("Ba B with line-aligned chunks (4608 floats)", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(1024), dim3(256), 0, 0, in[i], out[i], na, ca, dtab); });
        timeit("Ca C with line-aligned chunks", [&](int i) { hipLaunchKernelGGL((chunked<5, 1>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca, dtab); });
        timeit("Ea E with line-aligned chunks (one chunk per workgroup)", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(na), dim3(256), 0, 0, in[i], out[i], na, ca, dtab); });
        timeit("Ua U0 with line-aligned chunks (all up front, no LDS)", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 0>), dim3(1024), dim3(256), 0, 0, in[i], out[i], na, ca, dtab); });
        timeit("Ua1 U1 with line-aligned chunks (all up front, LDS identity)", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 1>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca, dtab); });
    }
    {
        const int ca = 4608, na = (int)(nfl / ca);
        const size_t la = (size_t)ca * 8 + 4096;
        timeit("Uc all up front, aligned chunks, CONTIGUOUS chunks per wg (841 wgs x 3), no LDS", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 0, true>), dim3((na + 2) / 3), dim3(256), 0, 0, in[i], out[i], na, ca, dtab); });
        timeit("Uc1 the same through LDS (identity gather)", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 1, true>), dim3((na + 2) / 3), dim3(256), la, 0, in[i], out[i], na, ca, dtab); });
        const int cf = 5120, nf = (int)(nfl / cf);     // full slots: 5 float4 per thread, nothing out of range
        const size_t lf = (size_t)cf * 8 + 4096;
        timeit("Uf all up front, chunks of 5120 floats (5 FULL slots), strided chunks, no LDS", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 0>), dim3(1024), dim3(256), 0, 0, in[i], out[i], nf, cf, dtab); });
        timeit("Ufc the same, contiguous chunks per wg (757 wgs x 3)", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 0, true>), dim3((nf + 2) / 3), dim3(256), 0, 0, in[i], out[i], nf, cf, dtab); });
        timeit("Ufc1 the same through LDS (identity gather)", [&](int i) { hipLaunchKernelGGL((upfront<5, 3, 1, true>), dim3((nf + 2) / 3), dim3(256), lf, 0, in[i], out[i], nf, cf, dtab); });
        timeit("Bf  depth-1 chunked, chunks of 5120 floats, 1024 wgs, no LDS", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(1024), dim3(256), 0, 0, in[i], out[i], nf, cf, dtab); });
        timeit("Ef  one chunk of 5120 floats per wg, no LDS", [&](int i) { hipLaunchKernelGGL((chunked<5, 0>), dim3(nf), dim3(256), 0, 0, in[i], out[i], nf, cf, dtab); });
        timeit("Ef1 one chunk of 5120 floats per wg through LDS (22 KB: 7 per CU)", [&](int i) { hipLaunchKernelGGL((chunked<5, 1>), dim3(nf), dim3(256), (size_t)cf * 4 + 4096 + 100, 0, in[i], out[i], nf, cf, dtab); });
    }
    {   // half-size chunks, twice the workgroups: every load of the tensor in flight at once AND few loads per thread
        const int ch = 2304, nh = (int)(nfl / ch);      // 2304 floats = 72 lines; 2.25 slots -> KMAX 3
        const size_t lh = (size_t)ch * 8 + 4096;
        timeit("H0 chunks of 2304 floats, 2048 wgs, 3 chunks up front (<= 9 float4/thread), no LDS", [&](int i) { hipLaunchKernelGGL((upfront<3, 3, 0>), dim3(2048), dim3(256), 0, 0, in[i], out[i], nh, ch, dtab); });
        timeit("H1 the same through LDS (identity gather), 8 wgs per CU", [&](int i) { hipLaunchKernelGGL((upfront<3, 3, 1>), dim3(2048), dim3(256), lh, 0, in[i], out[i], nh, ch, dtab); });
        timeit("H0c contiguous chunks per wg (1682 wgs x 3), no LDS", [&](int i) { hipLaunchKernelGGL((upfront<3, 3, 0, true>), dim3((nh + 2) / 3), dim3(256), 0, 0, in[i], out[i], nh, ch, dtab); });
        timeit("H1c the same through LDS", [&](int i) { hipLaunchKernelGGL((upfront<3, 3, 1, true>), dim3((nh + 2) / 3), dim3(256), lh, 0, in[i], out[i], nh, ch, dtab); });
        const int cq = 2048, nq = (int)(nfl / cq);      // 2 full slots
        const size_t lq = (size_t)cq * 8 + 4096;
        timeit("Q0 chunks of 2048 floats (2 full slots), 1892 wgs x 3 contiguous, no LDS", [&](int i) { hipLaunchKernelGGL((upfront<2, 3, 0, true>), dim3((nq + 2) / 3), dim3(256), 0, 0, in[i], out[i], nq, cq, dtab); });
        timeit("Q1 the same through LDS", [&](int i) { hipLaunchKernelGGL((upfront<2, 3, 1, true>), dim3((nq + 2) / 3), dim3(256), lq, 0, in[i], out[i], nq, cq, dtab); });
        timeit("Q0s chunks of 2048 floats, 2048 wgs strided x 3, no LDS", [&](int i) { hipLaunchKernelGGL((upfront<2, 3, 0>), dim3(2048), dim3(256), 0, 0, in[i], out[i], nq, cq, dtab); });
        timeit("Q1s the same through LDS", [&](int i) { hipLaunchKernelGGL((upfront<2, 3, 1>), dim3(2048), dim3(256), lq, 0, in[i], out[i], nq, cq, dtab); });
    }
    timeit("G1 copy A3 (6 float4/thread, 1892 wgs) with BUFFER loads / stores, resource word 3 = 0x00020000", [&](int i) { hipLaunchKernelGGL((copy_buf<6, 0x00020000>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("G2 the same, resource word 3 = 0x00027000 (dst_sel xyzw)", [&](int i) { hipLaunchKernelGGL((copy_buf<6, 0x00027000>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("G3 copy A (8 float4/thread, 2048 wgs) with buffer operations", [&](int i) { hipLaunchKernelGGL((copy_buf<8, 0x00020000>), dim3(2048), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("G4 copy A3 with buffer LOADS + global stores", [&](int i) { hipLaunchKernelGGL((copy_mix<6, 1>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("G5 copy A3 with global loads + buffer STORES", [&](int i) { hipLaunchKernelGGL((copy_mix<6, 2>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T0 copy 6 float4/thread, 1892 wgs, global ops, counted waits", [&](int i) { hipLaunchKernelGGL((copy_thr<6, 0>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T1 the same, ONE request per wave at a time (vmcnt(0) after every load and store)", [&](int i) { hipLaunchKernelGGL((copy_thr<6, 1>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T2 the same, all loads - wait - all stores - wait", [&](int i) { hipLaunchKernelGGL((copy_thr<6, 2>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T3 the same, loads two at a time, stores one at a time", [&](int i) { hipLaunchKernelGGL((copy_thr<6, 3>), dim3(1892), dim3(256), 0, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T1' one request per wave at a time, 5 float4/thread, 1024 wgs (the map's occupancy: 4 wgs per CU by LDS)", [&](int i) { hipLaunchKernelGGL((copy_thr<5, 1>), dim3(1024), dim3(256), 40000, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T2' all loads - wait - all stores - wait, 5 float4/thread, 1024 wgs, 4 per CU", [&](int i) { hipLaunchKernelGGL((copy_thr<5, 2>), dim3(1024), dim3(256), 40000, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T0' counted waits, 5 float4/thread, 1024 wgs, 4 per CU", [&](int i) { hipLaunchKernelGGL((copy_thr<5, 0>), dim3(1024), dim3(256), 40000, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T1'' one request at a time, 5 float4/thread, 2270 wgs (one iteration each), 7 per CU (22 KB LDS)", [&](int i) { hipLaunchKernelGGL((copy_thr<5, 1>), dim3(2270), dim3(256), 22000, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    timeit("T2'' all loads - wait - all stores, 5 float4/thread, 2270 wgs, 7 per CU", [&](int i) { hipLaunchKernelGGL((copy_thr<5, 2>), dim3(2270), dim3(256), 22000, 0, in[i], out[i], (unsigned)(nfl * 4)); });
    {
        const int ca = 4608, na = (int)(nfl / ca);
        const size_t la = (size_t)ca * 8 + 4096;
        timeit("V0 map structure (aligned chunks, 1024 wgs, 4 per CU), global ops, no LDS, counted waits", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 0, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("V1 ... at most 1 request per wave in flight", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 1, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("V2 ... at most 2", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 2, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("V3 ... at most 3", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 3, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("V4 ... at most 4", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 4, false>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("W0 the same through LDS (identity gather), counted waits", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 0, true>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("W2 through LDS, at most 2 in flight", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 2, true>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("W3 through LDS, at most 3 in flight", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 3, true>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
        timeit("W4 through LDS, at most 4 in flight", [&](int i) { hipLaunchKernelGGL((chunked_thr<5, 4, true>), dim3(1024), dim3(256), la, 0, in[i], out[i], na, ca); });
    }
    timeit("F  D with 2624 wgs (one chunk each)", [&](int i) { hipLaunchKernelGGL((chunked<5, 2>), dim3(nfull), dim3(256), lds2, 0, in[i], out[i], nfull, chunk, dtab); });
    return 0;
}
