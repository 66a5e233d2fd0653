// elementwise.hip -- HBM-bound kernels of the path: sub-pixel index maps, loss, optimizers,
// metrics.  All are grid-stride, 16 B per lane where the layout allows.
#include "elementwise.h"

namespace srx {

// ---------------------------------------------------------------------------------------------
// depth_to_space / space_to_depth.
// For one LR row (n,h) the r HR rows it produces are CONTIGUOUS in the output and cover exactly
// the same flat range [blk*B, (blk+1)*B), B = W*r*r*C, as the LR row does in the input.  So the
// map is one fixed permutation applied independently to every block of B floats:
//   d2s: out[dy*(W*rC) + w*rC + j] = in[w*(r*rC) + dy*rC + j],   rC = r*C
// A workgroup stages RB blocks into LDS with coalesced 16-B loads, gathers from LDS, and writes
// coalesced 16-B stores: HBM traffic is exactly read-once + write-once.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int d2s_src(int oo, int W, int rC, int r, float inv_WrC, float inv_rC) {
    // oo = dy*(W*rC) + w*rC + j  ->  w*(r*rC) + dy*rC + j
    const int WrC = W * rC;
    int dy = (int)(((float)oo + 0.5f) * inv_WrC);
    dy -= (dy * WrC > oo);
    dy += ((dy + 1) * WrC <= oo);
    const int rest = oo - dy * WrC;
    int w = (int)(((float)rest + 0.5f) * inv_rC);
    w -= (w * rC > rest);
    w += ((w + 1) * rC <= rest);
    const int j = rest - w * rC;
    return w * (r * rC) + dy * rC + j;
}

__device__ __forceinline__ int s2d_src(int oo, int W, int rC, int r, float inv_rrC, float inv_rC) {
    // oo = w*(r*rC) + dy*rC + j  ->  dy*(W*rC) + w*rC + j
    const int rrC = r * rC;
    int w = (int)(((float)oo + 0.5f) * inv_rrC);
    w -= (w * rrC > oo);
    w += ((w + 1) * rrC <= oo);
    const int rest = oo - w * rrC;
    int dy = (int)(((float)rest + 0.5f) * inv_rC);
    dy -= (dy * rC > rest);
    dy += ((dy + 1) * rC <= rest);
    const int j = rest - dy * rC;
    return dy * (W * rC) + w * rC + j;
}

// chunk = RB blocks (chunk_floats = RB*B, a multiple of 4).  Every chunk undergoes the SAME
// permutation, so a thread computes the LDS gather indices of its (at most KMAX) output float4s once
// and then only moves data: coalesced 16-B loads -> LDS -> 4 scalar LDS reads -> coalesced 16-B store.
//
// Software pipeline (DB = two LDS buffers): the loads of chunk c+1 are issued right after chunk c has been
// written to LDS, i.e. BEFORE chunk c is gathered and stored, so every wave keeps a chunk of reads in flight
// while it works through LDS; one barrier per chunk (a thread that writes buffer p for chunk c has passed the
// barrier of chunk c-1, which every thread reaches only after its gather of chunk c-2 from the same buffer).
// The first chunk's loads go out before the index computation (~30 integer divisions per thread by
// reciprocal multiplication), which then runs under the memory latency instead of in front of it.
// The barrier is the raw s_barrier behind an LDS-only wait: __syncthreads() would also drain the
// vector-memory queue, i.e. wait for the loads just issued.
//
// Gather without the 4-way bank conflict of "lane l reads output floats 4l+e in round e" (ds_read_b32 banks
// are (a/4) % 32 per 32-lane half: the lanes of a round sit 4 floats apart and hit 8 banks): in round e lane l
// reads element (e + l/8) % 4 of its float4 instead, so the four 8-lane groups of a half start on banks
// 0,1,2,3 (+ the permutation's own offsets), and the float4 is put back in order with two select stages.
template <bool INVERSE, int KMAX, bool DB>
__global__ __launch_bounds__(256) void subpixel_lds_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           size_t total, int B, int chunk_floats, int W, int rC,
                                                           int r) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int c4 = chunk_floats >> 2;
    const size_t stride = (size_t)gridDim.x * chunk_floats;
    size_t c0 = (size_t)blockIdx.x * chunk_floats;
    if (c0 >= total) return;
    auto chunk_len = [&](size_t at) -> int {
        return (int)((total - at < (size_t)chunk_floats) ? (total - at) : (size_t)chunk_floats);
    };
    f32x4 v[KMAX];
    auto issue_loads = [&](size_t at, int n4) {
        const f32x4* gin = reinterpret_cast<const f32x4*>(in + at);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = k * 256 + threadIdx.x;
            if (i < n4) v[k] = __builtin_nontemporal_load(gin + i);   // streamed once: keep it out of L2
        }
    };
    int n = chunk_len(c0);
    issue_loads(c0, n >> 2);

    const float inv_B = 1.0f / (float)B;
    const float inv_a = INVERSE ? 1.0f / (float)(r * rC) : 1.0f / (float)(W * rC);
    const float inv_rC = 1.0f / (float)rC;
    auto src_of = [&](int o) -> int {
        int blk = (int)(((float)o + 0.5f) * inv_B);
        blk -= (blk * B > o);
        blk += ((blk + 1) * B <= o);
        const int oo = o - blk * B;
        return blk * B + (INVERSE ? s2d_src(oo, W, rC, r, inv_a, inv_rC) : d2s_src(oo, W, rC, r, inv_a, inv_rC));
    };
    const int rot = (threadIdx.x >> 3) & 3;
    int sidx[KMAX][4];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int i = k * 256 + threadIdx.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) sidx[k][e] = (i < c4) ? src_of(4 * i + ((e + rot) & 3)) : 0;
    }
    const bool r1 = rot & 1, r2 = rot & 2;
    int p = 0;
    for (;;) {
        const int n4 = n >> 2;
        float* buf = lds + (DB ? p * chunk_floats : 0);
        if (!DB) lds_barrier();   // single buffer: the previous chunk's gathers are done
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = k * 256 + threadIdx.x;
            if (i < n4) reinterpret_cast<f32x4*>(buf)[i] = v[k];
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) buf[i] = in[c0 + i];
        const size_t next = c0 + stride;
        const bool more = next < total;
        const int nn = more ? chunk_len(next) : 0;
        if (more) issue_loads(next, nn >> 2);
        lds_barrier();
        f32x4* gout = reinterpret_cast<f32x4*>(out + c0);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = k * 256 + threadIdx.x;
            if (i < n4) {
                float a0 = buf[sidx[k][0]], a1 = buf[sidx[k][1]], a2 = buf[sidx[k][2]], a3 = buf[sidx[k][3]];
                // a_e = element (e + rot) % 4  ->  element j = a_{(j - rot) % 4}: rotate right by rot
                float b0 = r1 ? a3 : a0, b1 = r1 ? a0 : a1, b2 = r1 ? a1 : a2, b3 = r1 ? a2 : a3;
                f32x4 o;
                o[0] = r2 ? b2 : b0; o[1] = r2 ? b3 : b1; o[2] = r2 ? b0 : b2; o[3] = r2 ? b1 : b3;
                __builtin_nontemporal_store(o, gout + i);
            }
        }
        for (int o = (n4 << 2) + threadIdx.x; o < n; o += 256) out[c0 + o] = buf[src_of(o)];
        if (!more) break;
        c0 = next;
        n = nn;
        p ^= 1;
    }
}

// What a plain streaming copy of the same bytes reaches with the same launch shape (persistent workgroups, KMAX
// nontemporal 16-B loads in flight per thread, nontemporal stores): the ceiling the sub-pixel map is measured
// against (bench.py `subpixel.copy_ceiling_gbps`).
template <int KMAX>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4) {
    const size_t per = (size_t)KMAX * 256;
    for (size_t base = (size_t)blockIdx.x * per; base < n4; base += (size_t)gridDim.x * per) {
        f32x4 v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const size_t i = base + (size_t)k * 256 + threadIdx.x;
            if (i < n4) v[k] = __builtin_nontemporal_load(in + i);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const size_t i = base + (size_t)k * 256 + threadIdx.x;
            if (i < n4) __builtin_nontemporal_store(v[k], out + i);
        }
    }
}

hipError_t launch_stream_copy(const float* in, float* out, size_t bytes, hipStream_t s) {
    const size_t n4 = bytes / 16;
    if (n4 == 0) return hipSuccess;
    const size_t per = 8 * 256;
    size_t nb = (n4 + per - 1) / per;
    const int grid = (int)(nb < 2048 ? nb : 2048);
    hipLaunchKernelGGL(stream_copy_kernel<8>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const f32x4*>(in),
                       reinterpret_cast<f32x4*>(out), n4);
    return hipGetLastError();
}

// Fallback for rows too long for LDS: direct gather (reads stay inside one B-float block).
template <bool INVERSE>
__global__ __launch_bounds__(256) void subpixel_direct_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              size_t total, int B, int W, int rC, int r) {
    const float inv_a = INVERSE ? 1.0f / (float)(r * rC) : 1.0f / (float)(W * rC);
    const float inv_rC = 1.0f / (float)rC;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const size_t blk = o / (size_t)B;
        const int oo = (int)(o - blk * (size_t)B);
        const int src = INVERSE ? s2d_src(oo, W, rC, r, inv_a, inv_rC) : d2s_src(oo, W, rC, r, inv_a, inv_rC);
        out[o] = in[blk * (size_t)B + src];
    }
}

hipError_t launch_subpixel(const float* in, float* out, int N, int H, int W, int C, int r, bool inverse,
                           const SubpixelTune& kn, hipStream_t s) {
    const int rC = r * C;
    const size_t B = (size_t)W * r * rC;
    const size_t total = (size_t)N * H * B;
    if (total == 0) return hipSuccess;
    const size_t lds_cap = 48 * 1024;
    // RB blocks per chunk: chunk must be a multiple of 4 floats so every chunk start is 16-B aligned
    size_t RB = 4;
    if (B % 4 == 0) RB = 1; else if (B % 2 == 0) RB = 2;
    if (RB * B * 4 <= lds_cap && B < (1u << 20)) {
        const size_t target = (size_t)kn.chunk_kb * 1024;
        while (2 * RB * B * 4 <= target) RB *= 2;   // ~16-24 KiB chunks: several workgroups per CU
        const int chunk = (int)(RB * B);
        size_t nchunks = (total + chunk - 1) / chunk;
        // two LDS buffers (loads of the next chunk in flight during the gather) while four workgroups still share a CU
        const bool db = kn.db && (size_t)chunk * 8 <= 40 * 1024;
        // persistent workgroups: the index precomputation is paid once per workgroup
        const size_t cap = (size_t)(kn.grid > 0 ? kn.grid : 1024);
        int grid = (int)(nchunks < cap ? nchunks : cap);
        const int kneed = (chunk / 4 + 255) / 256;
        const size_t lds = (size_t)chunk * 4 * (db ? 2 : 1);
#define SRX_SUBPIXEL_LAUNCH2(INV, K, DBUF)                                                                      \
        hipLaunchKernelGGL((subpixel_lds_kernel<INV, K, DBUF>), dim3(grid), dim3(256), lds, s, in, out, total,  \
                           (int)B, chunk, W, rC, r);
#define SRX_SUBPIXEL_LAUNCH(K)                                                                                   \
        if (inverse) { if (db) { SRX_SUBPIXEL_LAUNCH2(true, K, true) } else { SRX_SUBPIXEL_LAUNCH2(true, K, false) } }    \
        else { if (db) { SRX_SUBPIXEL_LAUNCH2(false, K, true) } else { SRX_SUBPIXEL_LAUNCH2(false, K, false) } }
        if (kneed <= 4) { SRX_SUBPIXEL_LAUNCH(4) }
        else if (kneed <= 8) { SRX_SUBPIXEL_LAUNCH(8) }
        else { SRX_SUBPIXEL_LAUNCH(12) }
#undef SRX_SUBPIXEL_LAUNCH
#undef SRX_SUBPIXEL_LAUNCH2
    } else {
        if (B >= (1u << 22)) return hipErrorInvalidValue;
        size_t nb = (total + 255) / 256;
        int grid = (int)(nb < 8192 ? nb : 8192);
        if (inverse)
            hipLaunchKernelGGL(subpixel_direct_kernel<true>, dim3(grid), dim3(256), 0, s, in, out, total, (int)B, W, rC, r);
        else
            hipLaunchKernelGGL(subpixel_direct_kernel<false>, dim3(grid), dim3(256), 0, s, in, out, total, (int)B, W, rC, r);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// reductions: per-block partials (fixed grid) -> one finishing block, double accumulate.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = (threadIdx.x < (blockDim.x >> 6)) ? sh[threadIdx.x] : 0.f;
    if (wave == 0) t = wave_sum(t);
    return t;  // valid in wave 0
}

// MODE 0: (a-b)^2, also writes dpred = 2*(a-b)*inv ; MODE 1: a^2 (times b as a 0/1 mask when b != null)
template <int MODE>
__global__ __launch_bounds__(256) void sq_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         size_t n, float inv, float* __restrict__ dpred,
                                                         float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 va = reinterpret_cast<const f32x4*>(a)[i];
        if (MODE == 0) {
            const f32x4 vb = reinterpret_cast<const f32x4*>(b)[i];
            va -= vb;
            if (dpred) reinterpret_cast<f32x4*>(dpred)[i] = va * (2.0f * inv);
            acc += va[0] * va[0] + va[1] * va[1] + va[2] * va[2] + va[3] * va[3];
        } else if (b) {
            const f32x4 m = reinterpret_cast<const f32x4*>(b)[i];
            acc += m[0] * va[0] * va[0] + m[1] * va[1] * va[1] + m[2] * va[2] * va[2] + m[3] * va[3] * va[3];
        } else {
            acc += va[0] * va[0] + va[1] * va[1] + va[2] * va[2] + va[3] * va[3];
        }
    }
    if (blockIdx.x == 0) {
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            float d = a[i];
            if (MODE == 0) {
                d -= b[i];
                if (dpred) dpred[i] = d * (2.0f * inv);
            }
            acc += (MODE == 1 && b) ? b[i] * d * d : d * d;
        }
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ partial, int n, float scale,
                                                         float* __restrict__ out, int accumulate) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)(sh[0] * (double)scale);
        out[0] = accumulate ? out[0] + v : v;
    }
}

static int reduce_grid(size_t n) {
    size_t nb = ((n >> 2) + 255) / 256;
    if (nb < 1) nb = 1;
    return (int)(nb < (size_t)kReduceBlocks ? nb : (size_t)kReduceBlocks);
}

hipError_t launch_mse(const float* pred, const float* target, size_t n, float inv, float* loss, int accumulate,
                      float* dpred, float* scratch, hipStream_t s) {
    const int grid = reduce_grid(n);
    hipLaunchKernelGGL(sq_partial_kernel<0>, dim3(grid), dim3(256), 0, s, pred, target, n, inv, dpred, scratch);
    if (loss) hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, scratch, grid, inv, loss, accumulate);
    return hipGetLastError();
}

hipError_t launch_l2(const float* w, const float* mask, size_t n, float scale, float* loss, int accumulate,
                     float* scratch, hipStream_t s) {
    const int grid = reduce_grid(n);
    hipLaunchKernelGGL(sq_partial_kernel<1>, dim3(grid), dim3(256), 0, s, w, mask, n, 0.f,
                       (float*)nullptr, scratch);
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, scratch, grid, 0.5f * scale, loss, accumulate);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// optimizers (flat buffers)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, size_t n,
                                                      float lr_t, float b1, float b2, float eps, float gs) {
    const size_t n4 = n >> 2;
    const float ob1 = 1.0f - b1, ob2 = 1.0f - b2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gs;
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
        f32x4 wv = reinterpret_cast<f32x4*>(w)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mv[e] = b1 * mv[e] + ob1 * gv[e];
            vv[e] = b2 * vv[e] + ob2 * gv[e] * gv[e];
            wv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
        }
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        reinterpret_cast<f32x4*>(w)[i] = wv;
    }
    if (blockIdx.x == 0) {
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const float gi = g[i] * gs;
            const float mi = b1 * m[i] + ob1 * gi;
            const float vi = b2 * v[i] + ob2 * gi * gi;
            m[i] = mi;
            v[i] = vi;
            w[i] -= lr_t * mi / (sqrtf(vi) + eps);
        }
    }
}

__global__ __launch_bounds__(256) void momentum_clip_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                            float* __restrict__ acc, size_t n, float lr, float mom,
                                                            float cap, float gs) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * gs;
        gi = fminf(fmaxf(gi, -cap), cap);
        const float a = mom * acc[i] + gi;
        acc[i] = a;
        w[i] -= lr * a;
    }
}

static int ew_grid(size_t n, int per_thread) {
    size_t nb = (n / per_thread + 255) / 256;
    if (nb < 1) nb = 1;
    return (int)(nb < 2048 ? nb : 2048);
}

hipError_t launch_adam(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                       float eps, float gs, hipStream_t s) {
    hipLaunchKernelGGL(adam_tf_kernel, dim3(ew_grid(n, 4)), dim3(256), 0, s, w, g, m, v, n, lr_t, b1, b2, eps, gs);
    return hipGetLastError();
}

hipError_t launch_momentum(float* w, const float* g, float* acc, size_t n, float lr, float mom, float cap, float gs,
                           hipStream_t s) {
    hipLaunchKernelGGL(momentum_clip_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, s, w, g, acc, n, lr, mom, cap, gs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// misc elementwise
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dpre, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dpre[i] = dy[i] * act_grad_from_y(y[i], act);
}

__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n,
                                                     float a, float b) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
#pragma clang fp contract(off)
        const float t = a * x[i];                    // two roundings, like the reference's separate multiply and add
        out[i] = t + b;
    }
}

__global__ __launch_bounds__(256) void saturate_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out,
                                                          size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // multiply and add rounded separately, as the two TensorFlow ops are (a fused multiply-add rounds once and
        // moves a few values across an integer, i.e. changes the truncated byte)
#pragma clang fp contract(off)
        float v = x[i] * 127.5f;
        v = v + 127.5f;
        v = fminf(fmaxf(v, 0.f), 255.f);
        out[i] = (uint8_t)v;  // truncation toward zero, as tf.saturate_cast
    }
}

__global__ __launch_bounds__(256) void psnr_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ out, size_t per_image, float max_val) {
    __shared__ float sh[4];
    const float* pa = a + (size_t)blockIdx.x * per_image;
    const float* pb = b + (size_t)blockIdx.x * per_image;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < per_image; i += 256) {
        const float d = pa[i] - pb[i];
        acc += d * d;
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0)
        out[blockIdx.x] = 20.0f * log10f(max_val) - 10.0f * log10f(t / (float)per_image);
}

// Pixel replication and its gradient.  VEC floats per thread (4 when C is a multiple of 4: 16-byte accesses, a
// wavefront covers whole pixels of the 64-channel tensors); one index decomposition per vector.
template <int VEC>
__global__ __launch_bounds__(256) void upsample_nearest_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int N, int H, int W, int C, int f) {
    const int CV = C / VEC;
    const size_t total = (size_t)N * H * f * W * f * CV;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int c = (int)(o % CV);
        size_t t = o / CV;
        const int ow = (int)(t % ((size_t)W * f));
        t /= (size_t)W * f;
        const int oh = (int)(t % ((size_t)H * f));
        const int n = (int)(t / ((size_t)H * f));
        const size_t src = (((size_t)n * H + oh / f) * W + ow / f) * CV + c;
        if constexpr (VEC == 4)
            reinterpret_cast<float4*>(out)[o] = reinterpret_cast<const float4*>(in)[src];
        else
            out[o] = in[src];
    }
}

// gradient of the replication: din[n,h,w,c] = sum over the f x f block of dout (rows, then columns, in order)
template <int VEC>
__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                                   int N, int H, int W, int C, int f) {
    const int CV = C / VEC;
    const size_t total = (size_t)N * H * W * CV;
    const size_t orow = (size_t)W * f * CV;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int c = (int)(o % CV);
        size_t t = o / CV;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const size_t base = (((size_t)n * H * f + (size_t)h * f) * W * f + (size_t)w * f) * CV + c;
        if constexpr (VEC == 4) {
            float4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) {
                    const float4 v = reinterpret_cast<const float4*>(dout)[base + dy * orow + (size_t)dx * CV];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            reinterpret_cast<float4*>(din)[o] = acc;
        } else {
            float acc = 0.f;
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) acc += dout[base + dy * orow + (size_t)dx * CV];
            din[o] = acc;
        }
    }
}

// gradient through relu(a_in + b_in) where the two branches' gradients arrive separately:
// out = (y > 0) ? a + b : 0   (residual block: gradient via the skip path + gradient via the conv path)
__global__ __launch_bounds__(256) void add_relu_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ y, float* __restrict__ out, size_t n4,
                                                            size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 va = reinterpret_cast<const float4*>(a)[i], vb = reinterpret_cast<const float4*>(b)[i];
        const float4 vy = reinterpret_cast<const float4*>(y)[i];
        float4 r;
        r.x = vy.x > 0.f ? va.x + vb.x : 0.f; r.y = vy.y > 0.f ? va.y + vb.y : 0.f;
        r.z = vy.z > 0.f ? va.z + vb.z : 0.f; r.w = vy.w > 0.f ? va.w + vb.w : 0.f;
        reinterpret_cast<float4*>(out)[i] = r;
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = y[i] > 0.f ? a[i] + b[i] : 0.f;
}

// SRCNN loss: one block per row computes ||pred-target||_2 of that row; a second pass scales.
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      size_t row_len, float* __restrict__ norms) {
    __shared__ float sh[4];
    const float* pa = a + (size_t)blockIdx.x * row_len;
    const float* pb = b + (size_t)blockIdx.x * row_len;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < row_len; i += 256) {
        const float d = pa[i] - pb[i];
        acc += d * d;
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) norms[blockIdx.x] = sqrtf(t);
}

__global__ __launch_bounds__(256) void rownorm_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           size_t rows, size_t row_len, const float* __restrict__ norms,
                                                           float* __restrict__ dpred) {
    const size_t n = rows * row_len;
    const float inv_rows = 1.0f / (float)rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float nr = norms[i / row_len];
        dpred[i] = nr > 0.f ? (a[i] - b[i]) * inv_rows / nr : 0.f;
    }
}

hipError_t launch_rownorm_loss(const float* pred, const float* target, size_t rows, size_t row_len, float* loss,
                               float* dpred, float* norms, hipStream_t s) {
    hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)rows), dim3(256), 0, s, pred, target, row_len, norms);
    if (loss) hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, s, norms, (int)rows, 1.0f / (float)rows, loss, 0);
    if (dpred)
        hipLaunchKernelGGL(rownorm_grad_kernel, dim3(ew_grid(rows * row_len, 1)), dim3(256), 0, s, pred, target, rows,
                           row_len, norms, dpred);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// SSIM (tf.image.ssim semantics).  One thread per output position and channel evaluates the 11x11
// gaussian-weighted moments directly; evaluation-only code, HBM/L2-bound at image sizes.
// ---------------------------------------------------------------------------------------------
constexpr int kSsimBlocks = 128;   // partial sums per image

__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           int H, int W, int C, float c1, float c2,
                                                           float* __restrict__ partial) {
    __shared__ float sh[4];
    __shared__ float g[11];
    if (threadIdx.x < 11) {
        float sum = 0.f;
        for (int i = 0; i < 11; ++i) sum += __expf(-(float)((i - 5) * (i - 5)) / (2.0f * 1.5f * 1.5f));
        const int d = (int)threadIdx.x - 5;
        g[threadIdx.x] = __expf(-(float)(d * d) / (2.0f * 1.5f * 1.5f)) / sum;
    }
    __syncthreads();
    const int n = blockIdx.y;
    const int OH = H - 10, OW = W - 10;
    const size_t total = (size_t)OH * OW * C;
    const float* pa = a + (size_t)n * H * W * C;
    const float* pb = b + (size_t)n * H * W * C;
    float acc = 0.f;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int c = (int)(o % C);
        const size_t t = o / C;
        const int ow = (int)(t % OW), oh = (int)(t / OW);
        float ma = 0.f, mb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
        for (int i = 0; i < 11; ++i) {
            const float* ra = pa + ((size_t)(oh + i) * W + ow) * C + c;
            const float* rb = pb + ((size_t)(oh + i) * W + ow) * C + c;
            float xa = 0.f, xb = 0.f, xaa = 0.f, xbb = 0.f, xab = 0.f;
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                const float va = ra[(size_t)j * C], vb = rb[(size_t)j * C], w = g[j];
                xa += w * va; xb += w * vb; xaa += w * va * va; xbb += w * vb * vb; xab += w * va * vb;
            }
            const float wi = g[i];
            ma += wi * xa; mb += wi * xb; saa += wi * xaa; sbb += wi * xbb; sab += wi * xab;
        }
        // TF's _ssim_per_channel: luminance * contrast-structure
        const float num0 = 2.0f * ma * mb, den0 = ma * ma + mb * mb;
        const float lum = (num0 + c1) / (den0 + c1);
        const float cs = (2.0f * sab - num0 + c2) / (saa + sbb - den0 + c2);
        acc += lum * cs;
    }
    const float t = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)n * kSsimBlocks + blockIdx.x] = t;
}

__global__ __launch_bounds__(128) void ssim_finish_kernel(const float* __restrict__ partial, float inv_count,
                                                          float* __restrict__ out) {
    __shared__ double sh[128];
    sh[threadIdx.x] = (threadIdx.x < kSsimBlocks) ? (double)partial[(size_t)blockIdx.x * kSsimBlocks + threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(sh[0] * (double)inv_count);
}

size_t ssim_scratch_bytes(int N) { return (size_t)(N > 0 ? N : 0) * kSsimBlocks * sizeof(float); }

hipError_t launch_ssim(const float* a, const float* b, float* out, int N, int H, int W, int C, float max_val,
                       float* scratch, hipStream_t s) {
    const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(kSsimBlocks, N), dim3(256), 0, s, a, b, H, W, C, c1, c2, scratch);
    const double count = (double)(H - 10) * (W - 10) * C;
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(N), dim3(128), 0, s, scratch, (float)(1.0 / count), out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LR synthesis ("next" row N1): uint8 -> float, separable gaussian blur, bilinear resize
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void u8_to_float_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                          size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (float)in[i] / 255.0f;
}

// AXIS 0: along H, 1: along W.  Border: replicate (index clamp).
template <int AXIS>
__global__ __launch_bounds__(256) void gaussian_1d_kernel(const float* __restrict__ in, float* __restrict__ out, int N,
                                                          int H, int W, int C, float sigma, int radius) {
    __shared__ float wts[64];
    if (threadIdx.x <= (unsigned)radius && threadIdx.x < 64) {
        float sum = 0.f;
        for (int i = -radius; i <= radius; ++i) sum += expf(-0.5f * (float)(i * i) / (sigma * sigma));
        wts[threadIdx.x] = expf(-0.5f * (float)(threadIdx.x * threadIdx.x) / (sigma * sigma)) / sum;
    }
    __syncthreads();
    const size_t total = (size_t)N * H * W * C;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int c = (int)(o % C);
        size_t t = o / C;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const size_t n = t / H;
        const float* base = in + n * (size_t)H * W * C + c;
        float acc = 0.f;
        for (int i = -radius; i <= radius; ++i) {
            int hh = h, ww = w;
            if (AXIS == 0) { hh = h + i; hh = hh < 0 ? 0 : (hh >= H ? H - 1 : hh); }
            else { ww = w + i; ww = ww < 0 ? 0 : (ww >= W ? W - 1 : ww); }
            acc += wts[i < 0 ? -i : i] * base[((size_t)hh * W + ww) * C];
        }
        out[o] = acc;
    }
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int N, int H, int W, int C, int OH, int OW) {
    const size_t total = (size_t)N * OH * OW * C;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int c = (int)(o % C);
        size_t t = o / C;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const size_t n = t / OH;
        float fy = ((float)oh + 0.5f) * sy - 0.5f, fx = ((float)ow + 0.5f) * sx - 0.5f;
        fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
        fx = fminf(fmaxf(fx, 0.f), (float)(W - 1));
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
        const float wy = fy - (float)y0, wx = fx - (float)x0;
        const float* base = in + n * (size_t)H * W * C + c;
        const float v00 = base[((size_t)y0 * W + x0) * C], v01 = base[((size_t)y0 * W + x1) * C];
        const float v10 = base[((size_t)y1 * W + x0) * C], v11 = base[((size_t)y1 * W + x1) * C];
        out[o] = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
    }
}

hipError_t launch_u8_to_float(const uint8_t* in, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(u8_to_float_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

hipError_t launch_gaussian_blur(const float* in, float* out, float* tmp, int N, int H, int W, int C, float sigma,
                                hipStream_t s) {
    const size_t total = (size_t)N * H * W * C;
    if (sigma <= 0.f) return hipMemcpyAsync(out, in, total * sizeof(float), hipMemcpyDeviceToDevice, s);
    const int radius = (int)(4.0f * sigma + 0.5f);
    if (radius > 63) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gaussian_1d_kernel<0>, dim3(ew_grid(total, 1)), dim3(256), 0, s, in, tmp, N, H, W, C, sigma, radius);
    hipLaunchKernelGGL(gaussian_1d_kernel<1>, dim3(ew_grid(total, 1)), dim3(256), 0, s, tmp, out, N, H, W, C, sigma, radius);
    return hipGetLastError();
}

hipError_t launch_resize_bilinear(const float* in, float* out, int N, int H, int W, int C, int OH, int OW, hipStream_t s) {
    const size_t total = (size_t)N * OH * OW * C;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(ew_grid(total, 1)), dim3(256), 0, s, in, out, N, H, W, C, OH, OW);
    return hipGetLastError();
}

hipError_t launch_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act, hipStream_t s) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, s, dy, y, dpre, n, act);
    return hipGetLastError();
}
hipError_t launch_affine(const float* x, float* out, size_t n, float a, float b, hipStream_t s) {
    hipLaunchKernelGGL(affine_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, s, x, out, n, a, b);
    return hipGetLastError();
}
hipError_t launch_saturate_u8(const float* x, uint8_t* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(saturate_u8_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, s, x, out, n);
    return hipGetLastError();
}
hipError_t launch_psnr(const float* a, const float* b, float* out, int N, size_t per_image, float max_val,
                       hipStream_t s) {
    hipLaunchKernelGGL(psnr_kernel, dim3(N), dim3(256), 0, s, a, b, out, per_image, max_val);
    return hipGetLastError();
}
hipError_t launch_upsample_nearest(const float* in, float* out, int N, int H, int W, int C, int f, hipStream_t s) {
    const bool v4 = (C % 4) == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
    const size_t total = (size_t)N * H * f * W * f * (v4 ? C / 4 : C);
    if (v4)
        hipLaunchKernelGGL(upsample_nearest_kernel<4>, dim3(ew_grid(total, 1)), dim3(256), 0, s, in, out, N, H, W, C, f);
    else
        hipLaunchKernelGGL(upsample_nearest_kernel<1>, dim3(ew_grid(total, 1)), dim3(256), 0, s, in, out, N, H, W, C, f);
    return hipGetLastError();
}
hipError_t launch_upsample_nearest_bwd(const float* dout, float* din, int N, int H, int W, int C, int f, hipStream_t s) {
    const bool v4 = (C % 4) == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)din % 16) == 0;
    const size_t total = (size_t)N * H * W * (v4 ? C / 4 : C);
    if (v4)
        hipLaunchKernelGGL(upsample_nearest_bwd_kernel<4>, dim3(ew_grid(total, 1)), dim3(256), 0, s, dout, din, N, H, W, C, f);
    else
        hipLaunchKernelGGL(upsample_nearest_bwd_kernel<1>, dim3(ew_grid(total, 1)), dim3(256), 0, s, dout, din, N, H, W, C, f);
    return hipGetLastError();
}
hipError_t launch_add_relu_grad(const float* a, const float* b, const float* y, float* out, size_t n, hipStream_t s) {
    const bool v4 = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y | (uintptr_t)out) % 16) == 0;
    const size_t n4 = v4 ? n / 4 : 0;
    hipLaunchKernelGGL(add_relu_grad_kernel, dim3(ew_grid(n4 ? n4 : n, 1)), dim3(256), 0, s, a, b, y, out, n4, n);
    return hipGetLastError();
}

}  // namespace srx
