// enet_ops.hip -- the operators EnhanceNet-PAT's loss side needs beside the convolutions (SURVEY 8a row A14,
// 8f row N4; reference enet/enet/model_enet.py:118-261, enet/enet/model_vgg.py:28-36): 2x2 max-pooling, the
// stride-2 sample map, block <-> NHWC channel layouts, channel-mean normalisation, 16x16 patch extraction,
// log-loss, the VGG input map, and an exact-fp32 MFMA GEMM (dense layers, gram matrices).
// Parity-first kernels: straightforward, HBM-bound apart from the GEMM; all with extern "C" entry points below.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/srx.h"
#include "conv_kernels.hip.h"

namespace srx {
int set_error(int code, const char* fmt, ...);   // srx_api.hip: thread-local error text

namespace {

inline int ew_blocks(size_t n) {
    size_t nb = (n + 255) / 256;
    if (nb < 1) nb = 1;
    return (int)(nb < 4096 ? nb : 4096);
}
#define SRX_GRID_STRIDE(i, n) for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (size_t)gridDim.x * 256)

// ---------------------------------------------------------------------------------------------
// tf.nn.max_pool(ksize 2x2, strides 2x2, padding='SAME'), NHWC, C % 4 == 0.  OH = ceil(H/2); a window that sticks
// out of the image (odd H / W) covers the inside part only.  Backward (MaxPoolGrad): the gradient of a window goes
// to its FIRST maximum in scan order (dy, dx) -- a tie rule; after a ReLU the ties are zeros whose gradient the
// ReluGrad below discards anyway.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int N, int H,
                                                           int W, int C4, int OH, int OW) {
    const size_t total = (size_t)N * OH * OW * C4;
    SRX_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int n = (int)(p / OH);
        const f32x4* base = in + ((size_t)n * H * W) * C4 + c;
        f32x4 m = base[((size_t)(2 * oh) * W + 2 * ow) * C4];
        const bool r1 = 2 * oh + 1 < H, c1 = 2 * ow + 1 < W;
        if (c1) { const f32x4 v = base[((size_t)(2 * oh) * W + 2 * ow + 1) * C4]; for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]); }
        if (r1) { const f32x4 v = base[((size_t)(2 * oh + 1) * W + 2 * ow) * C4]; for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]); }
        if (r1 && c1) { const f32x4 v = base[((size_t)(2 * oh + 1) * W + 2 * ow + 1) * C4]; for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]); }
        out[i] = m;
    }
}

// mask_act != ACT_NONE: the gradient of the activation that PRODUCED x rides along (din *= act'(x), x being the
// post-activation tensor): the ReluGrad that follows every pooling gradient in VGG-19's backward pass costs no launch
// and no extra pass over the tensor -- the kernel has x in registers to find the arg-max anyway.
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dout,
                                                           f32x4* __restrict__ din, int N, int H, int W, int C4, int OH, int OW,
                                                           int mask_act) {
    const size_t total = (size_t)N * H * W * C4;
    SRX_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const int oh = h >> 1, ow = w >> 1;
        const int me = (h & 1) * 2 + (w & 1);
        const f32x4* base = x + ((size_t)n * H * W) * C4 + c;
        f32x4 v[4];
        bool in_img[4];
        for (int k = 0; k < 4; ++k) {
            const int hh = 2 * oh + (k >> 1), ww = 2 * ow + (k & 1);
            in_img[k] = hh < H && ww < W;
            v[k] = in_img[k] ? base[((size_t)hh * W + ww) * C4] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const f32x4 g = dout[(((size_t)n * OH + oh) * OW + ow) * C4 + c];
        f32x4 r;
        for (int e = 0; e < 4; ++e) {
            int arg = 0;
            float best = v[0][e];
            for (int k = 1; k < 4; ++k)
                if (in_img[k] && v[k][e] > best) { best = v[k][e]; arg = k; }
            r[e] = (arg == me) ? g[e] : 0.f;
            if (mask_act) r[e] *= act_grad_from_y(v[me][e], mask_act);
        }
        din[i] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// Stride-2 sample map.  A stride-2 SAME 3x3 convolution on an even-sized image (TF pads 0 before / 1 after:
// enet/enet/model_enet.py:136-146) equals the stride-1 SAME convolution sampled at the odd positions
// (2i+1, 2j+1); its gradients take the zero-stuffed upstream gradient through the stride-1 backward kernels.
//   fwd: out[n,i,j,:] = in[n,2i+oy,2j+ox,:]      bwd: din[n,h,w,:] = (h%2==oy && w%2==ox) ? dout[n,h/2,w/2,:] : 0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subsample2_fwd_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int N, int H,
                                                             int W, int C4, int oy, int ox) {
    const int OH = H / 2, OW = W / 2;
    const size_t total = (size_t)N * OH * OW * C4;
    SRX_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int n = (int)(p / OH);
        out[i] = in[(((size_t)n * H + 2 * oh + oy) * W + 2 * ow + ox) * C4 + c];
    }
}
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const f32x4* __restrict__ dout, f32x4* __restrict__ din, int N, int H,
                                                             int W, int C4, int oy, int ox) {
    const int OH = H / 2, OW = W / 2;
    const size_t total = (size_t)N * H * W * C4;
    SRX_GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const bool hit = (h & 1) == oy && (w & 1) == ox;
        din[i] = hit ? dout[(((size_t)n * OH + (h >> 1)) * OW + (w >> 1)) * C4 + c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---------------------------------------------------------------------------------------------
// Channel-blocked <-> NHWC.  Layers with more than 64 channels keep their activations as CB tensors of
// [P, 64] (P = N*H*W pixels) so that the 64-channel convolution kernels run on them per block pair;
// the loss operators (and checkpoints) want plain [P, CB*64].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deblock_kernel(const f32x4* __restrict__ blk, f32x4* __restrict__ plain, size_t P, int CB) {
    const size_t total = P * CB * 16;
    SRX_GRID_STRIDE(i, total) {
        const int c4 = (int)(i % 16);
        size_t r = i / 16;
        const int b = (int)(r % CB);
        const size_t p = r / CB;
        plain[i] = blk[((size_t)b * P + p) * 16 + c4];
    }
}
__global__ __launch_bounds__(256) void block_kernel(const f32x4* __restrict__ plain, f32x4* __restrict__ blk, size_t P, int CB) {
    const size_t total = P * CB * 16;
    SRX_GRID_STRIDE(i, total) {
        const int c4 = (int)(i % 16);
        size_t r = i / 16;
        const int b = (int)(r % CB);
        const size_t p = r / CB;
        blk[((size_t)b * P + p) * 16 + c4] = plain[i];
    }
}

// ---------------------------------------------------------------------------------------------
// normalize(): y = x / (mean_c(x) + 1e-6)  (enet/enet/model_enet.py:34-41), x [P, C]; one wavefront per pixel.
// bwd: with m = mean + eps, s = sum_c dy_c x_c:  dx_c = dy_c / m - s / (C m^2)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void chan_norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t P, int C, float eps) {
    const int lane = threadIdx.x & 63;
    for (size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (size_t)gridDim.x * 4) {
        const float* xp = x + p * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xp[c];
        const float m = wave_sum(s) / (float)C + eps;
        for (int c = lane; c < C; c += 64) y[p * C + c] = xp[c] / m;
    }
}
__global__ __launch_bounds__(256) void chan_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dx, size_t P, int C, float eps) {
    const int lane = threadIdx.x & 63;
    for (size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (size_t)gridDim.x * 4) {
        const float* xp = x + p * C;
        const float* gp = dy + p * C;
        float s = 0.f, t = 0.f;
        for (int c = lane; c < C; c += 64) { s += xp[c]; t += gp[c] * xp[c]; }
        const float m = wave_sum(s) / (float)C + eps;
        const float k = wave_sum(t) / ((float)C * m * m);
        for (int c = lane; c < C; c += 64) dx[p * C + c] = gp[c] / m - k;
    }
}

// ---------------------------------------------------------------------------------------------
// tf.extract_image_patches(ksizes 16x16, strides 16x16, VALID) + reshape [-1, h*w/256, 256, c]
// (enet/enet/model_enet.py:237-250): x [N,H,W,C] -> patches [N, (H/16)*(W/16), 256, C], a permutation of pixels.
// bwd = the inverse permutation.  H, W multiples of 16.
// ---------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(256) void patches16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int N, int H,
                                                        int W, int C4) {
    const int PW = W / 16;
    const size_t total = (size_t)N * H * W * C4;
    SRX_GRID_STRIDE(i, total) {            // i indexes the PATCH layout
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int kx = (int)(r % 16); r /= 16;
        const int ky = (int)(r % 16); r /= 16;
        const int px = (int)(r % PW); r /= PW;
        const int py = (int)(r % (H / 16));
        const int n = (int)(r / (H / 16));
        const size_t img = (((size_t)n * H + py * 16 + ky) * W + px * 16 + kx) * C4 + c;
        if (INVERSE) dst[img] = src[i];
        else dst[i] = src[img];
    }
}

// ---------------------------------------------------------------------------------------------
// tf.losses.log_loss(labels, predictions, epsilon=1e-7, reduction=MEAN) (enet/enet/model_enet.py:165-182):
//   loss = mean( -y log(p + eps) - (1 - y) log(1 - p + eps) ),  dp = ( -y/(p+eps) + (1-y)/(1-p+eps) ) * gscale / n
// One block (n is a batch size).  *loss_out (+)= lscale * loss.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void log_loss_kernel(const float* __restrict__ p, float label, int n, float eps, float lscale,
                                                       float gscale, float* __restrict__ loss_out, int accumulate,
                                                       float* __restrict__ dp) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = p[i];
        s += -label * logf(v + eps) - (1.f - label) * logf(1.f - v + eps);
        if (dp) dp[i] = (-label / (v + eps) + (1.f - label) / (1.f - v + eps)) * gscale / (float)n;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss_out) *loss_out = (accumulate ? *loss_out : 0.f) + lscale * sh[0] / (float)n;
}

// ---------------------------------------------------------------------------------------------
// VGG input map (enet/enet/model_enet.py:288-289 + enet/enet/model_vgg.py:72-76):
//   out[..., c] = (in[..., 2-c] * 127.5 + 127.5) - mean_bgr[c]      bwd: din[..., c] = 127.5 * dout[..., 2-c]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vgg_pre_kernel(const float* __restrict__ in, float* __restrict__ out, size_t npix, int bwd) {
    const float mean[3] = {103.939f, 116.779f, 123.68f};
    SRX_GRID_STRIDE(i, npix * 3) {
#pragma clang fp contract(off)     // multiply, add and subtract are three TensorFlow ops: three roundings
        const int c = (int)(i % 3);
        const size_t j = i - c + (2 - c);
        if (bwd) out[i] = 127.5f * in[j];
        else { const float t = in[j] * 127.5f; out[i] = (t + 127.5f) - mean[c]; }
    }
}

// ---------------------------------------------------------------------------------------------
// tf.image.resize_bicubic(images, [OH, OW]) as TensorFlow 1.x computes it (align_corners=False, no half-pixel
// centres): SRCNN's in-graph degradation, srcnn/srcnn.py:89-93 (down by the factor, up again).
//   in = out * (IN / OUT);  lower = floor(in);  offset = lrint((in - lower) * 1024)
//   weights from the cubic convolution kernel with A = -0.75 evaluated at offset / 1024 (TF's 1024-entry table):
//     w1 = ((A+2) x - (A+3)) x^2 + 1,  w0 = ((A x' - 5A) x' + 8A) x' - 4A with x' = x + 1,  w2 / w3 the same at 1 - x
//   taps lower-1 .. lower+2, clamped to the image; rows are interpolated along x first, then along y.
// An integer down-scaling factor gives offset 0 everywhere: weights (0,1,0,0) = plain decimation, no anti-aliasing.
// Pinned by the reference's own outputs: assets/srcnn_00{0,1}.jpg show hd | sd | sr panels, and this restatement
// reproduces their sd panel from their hd panel to JPEG noise (41-42.5 dB; A = -0.5: 1.6 dB less; half-pixel centres:
// 20-24 dB) -- tests/test_oracle_pins.py::test_p6_*.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bicubic_tf_taps(int o, float scale, int limit, int (&idx)[4], float (&w)[4]) {
#pragma clang fp contract(off)
    const float A = -0.75f;
    const float in = (float)o * scale;
    const float fl = floorf(in);
    const int lower = (int)fl;
    const int offset = (int)lrintf((in - fl) * 1024.0f);
    const float x = (float)offset / 1024.0f, xr = (float)(1024 - offset) / 1024.0f;
    const float x1 = x + 1.0f, xr1 = xr + 1.0f;
    w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    w[0] = ((A * x1 - 5.0f * A) * x1 + 8.0f * A) * x1 - 4.0f * A;
    w[2] = ((A + 2.0f) * xr - (A + 3.0f)) * xr * xr + 1.0f;
    w[3] = ((A * xr1 - 5.0f * A) * xr1 + 8.0f * A) * xr1 - 4.0f * A;
    for (int k = 0; k < 4; ++k) {
        int i = lower - 1 + k;
        idx[k] = i < 0 ? 0 : (i > limit - 1 ? limit - 1 : i);
    }
}

__global__ __launch_bounds__(256) void resize_bicubic_tf_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H,
                                                                int W, int C, int OH, int OW, float sy, float sx) {
    const size_t total = (size_t)N * OH * OW * C;
    SRX_GRID_STRIDE(i, total) {
#pragma clang fp contract(off)
        const int c = (int)(i % C);
        size_t p = i / C;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int n = (int)(p / OH);
        int iy[4], ix[4];
        float wy[4], wx[4];
        bicubic_tf_taps(oy, sy, H, iy, wy);
        bicubic_tf_taps(ox, sx, W, ix, wx);
        const float* img = in + (size_t)n * H * W * C + c;
        float acc = 0.f;
        for (int r = 0; r < 4; ++r) {
            const float* row = img + (size_t)iy[r] * W * C;
            float v = 0.f;
            for (int k = 0; k < 4; ++k) v += wx[k] * row[(size_t)ix[k] * C];
            acc += wy[r] * v;
        }
        out[i] = acc;
    }
}

// out = alpha * a + beta * b  (gradients that reach one tensor from two consumers; loss-weight scaling)
__global__ __launch_bounds__(256) void add_scaled_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                         size_t n, float alpha, float beta) {
    SRX_GRID_STRIDE(i, n) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// out[j] (+)= sum_i a[i*ld + j]   (bias gradient of a dense layer), one thread per column
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, float* __restrict__ out, int rows, int cols, int ld) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    float s = 0.f;
    for (int i = 0; i < rows; ++i) s += a[(size_t)i * ld + j];
    out[j] = s;
}

// ---------------------------------------------------------------------------------------------
// Exact-fp32 GEMM on v_mfma_f32_16x16x4_f32 with general strides, batched:
//   C_b(m,n) = act( alpha * sum_k A_b(m,k) B_b(k,n) + bias(n) ) [+ C_b(m,n) if accumulate]
//   X_b(i,j) = X + b*batch_stride + i*s_i + j*s_j
// A workgroup owns a 64x64 tile of C (wave w: rows 16w..16w+15, four 16-column accumulators) and, with split-K,
// one K range; split-K partials go to `part` [S][batch? no: batch == 1][M][N] and a second kernel adds them in a fixed
// order (deterministic) and applies the epilogue.  Operands straight from global memory (L1/L2): these are the
// dense layers (M = batch), the 16x16-patch gram matrices (K = 256) and their gradients -- small next to the
// convolutions.  tf.layers.dense / tf.matmul: enet/enet/model_enet.py:148-160,252-255.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias; float* part;
    int M, N, K;
    long sam, sak, sbk, sbn, scm, scn, batchA, batchB, batchC;
    float alpha;
    int act, accumulate, splits, ksplit;   // ksplit: K range per split (multiple of 4)
};

__device__ __forceinline__ float gemm_act(float v, int act) {
    if (act == ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == ACT_TANH) return srx_tanhf(v);
    return v;
}

__global__ __launch_bounds__(256) void gemm_mfma_kernel(const GemmArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int tiles_n = (g.N + 63) / 64;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int split = blockIdx.y, batch = blockIdx.z;
    const float* A = g.A + (size_t)batch * g.batchA;
    const float* B = g.B + (size_t)batch * g.batchB;
    const int m = tm * 64 + wave * 16 + li;            // A row of this lane
    const bool m_ok = m < g.M;
    const float* Arow = A + (long)(m_ok ? m : 0) * g.sam;
    int ncol[4];
    bool n_ok[4];
    for (int j = 0; j < 4; ++j) { ncol[j] = tn * 64 + 16 * j + li; n_ok[j] = ncol[j] < g.N; if (!n_ok[j]) ncol[j] = 0; }
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int k0 = split * g.ksplit;
    int k1 = k0 + g.ksplit;
    if (k1 > g.K) k1 = g.K;
    for (int kb = k0; kb < k1; kb += 16) {
        float a[4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + 4 * u + kq;
            const bool k_ok = k < k1;
            const long ks = k_ok ? k : k0;
            a[u] = (m_ok && k_ok) ? Arow[ks * g.sak] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[u][j] = (n_ok[j] && k_ok) ? B[ks * g.sbk + (long)ncol[j] * g.sbn] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][j], acc[j], 0, 0, 0);
    }
    // lane holds C(rows 4*kq+r of the wave's 16, column li of each 16-column block)
    for (int j = 0; j < 4; ++j) {
        const int n = tn * 64 + 16 * j + li;
        if (n >= g.N) continue;
        for (int r = 0; r < 4; ++r) {
            const int mm = tm * 64 + wave * 16 + 4 * kq + r;
            if (mm >= g.M) continue;
            if (g.splits > 1) {
                g.part[((size_t)split * g.M + mm) * g.N + n] = acc[j][r];
            } else {
                float* c = g.C + (size_t)batch * g.batchC + (long)mm * g.scm + (long)n * g.scn;
                float v = g.alpha * acc[j][r] + (g.bias ? g.bias[n] : 0.f);
                v = gemm_act(v, g.act);
                *c = g.accumulate ? *c + v : v;
            }
        }
    }
}

// The same GEMM with a 64x64 tile PER WAVE (16 accumulators): 8 operand loads per 16 MFMAs instead of 20 -- the gram
// matrices (M = N = C >= 64, K = 256, thousands of batches) and the dense layers' filter gradients.  A workgroup's four
// waves take four consecutive (batch, tile) units.
__global__ __launch_bounds__(256) void gemm_mfma_w64_kernel(const GemmArgs g, int tiles_m, int tiles_n, long units) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const long unit = (long)blockIdx.x * 4 + wave;
    if (unit >= units) return;
    const int tn = (int)(unit % tiles_n);
    const long r1 = unit / tiles_n;
    const int tm = (int)(r1 % tiles_m);
    const int batch = (int)(r1 / tiles_m);
    const float* A = g.A + (size_t)batch * g.batchA;
    const float* B = g.B + (size_t)batch * g.batchB;
    long aoff[4], boff[4];
    bool m_ok[4], n_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = tm * 64 + 16 * j + li, n = tn * 64 + 16 * j + li;
        m_ok[j] = m < g.M; n_ok[j] = n < g.N;
        aoff[j] = (long)(m_ok[j] ? m : 0) * g.sam;
        boff[j] = (long)(n_ok[j] ? n : 0) * g.sbn;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < g.K; kb += 8) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = kb + 4 * u + kq;
            const bool k_ok = k < g.K;
            const long ks = k_ok ? k : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[u][j] = (m_ok[j] && k_ok) ? A[aoff[j] + ks * g.sak] : 0.f;
                b[u][j] = (n_ok[j] && k_ok) ? B[boff[j] + ks * g.sbk] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * 64 + 16 * j + li;
            if (n >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = tm * 64 + 16 * i + 4 * kq + r;
                if (mm >= g.M) continue;
                float* c = g.C + (size_t)batch * g.batchC + (long)mm * g.scm + (long)n * g.scn;
                float v = g.alpha * acc[i][j][r] + (g.bias ? g.bias[n] : 0.f);
                v = gemm_act(v, g.act);
                *c = g.accumulate ? *c + v : v;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Skinny GEMMs: M <= 16 rows against a large row-major matrix, one pass over the matrix at HBM speed.  The
// discriminator's first dense layer at BASELINE config 5's tile size has 16 x 16 x 512 = 131,072 inputs x 1,024 units =
// 537 MB of weights for a batch of 4-8 rows; the 64 x 64-tile kernel above reads them as 4-byte pieces of 64 bytes per
// row (forward, 16 splits = 256 workgroups) or with a 4-KB stride between lanes (data gradient): 311-380 us = 1.4-1.7 TB/s.
//  NN  C[m,n] = sum_k A[m,k] W[k,n]: a thread owns four columns (one 16-byte load per row of W), a workgroup 1,024
//      columns and one K range whose slice of A sits in LDS (broadcast reads); split-K partials + the finish kernel above.
//  NT  C[m,n] = sum_k A[m,k] W[n,k]: a wave owns 16 / MT rows of W at a time (16-byte loads along the row), A [MT][K] in
//      LDS, per-lane partial sums for its 16 (row, m) results, then a transposing butterfly: in step o = 32, 16, 8, 4 a
//      lane hands half of its values to lane ^ o and keeps the other half (8 + 4 + 2 + 1 exchanges instead of 16 x 6),
//      two plain steps finish the sum over the remaining four lanes.
// tf.layers.dense forward / tf.gradients: enet/enet/model_enet.py:148-154, 331-346.
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_nn_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float a_lds[];      // [MT][ksplit]
    const int tid = threadIdx.x;
    const int n = blockIdx.x * 1024 + tid * 4;
    const int split = blockIdx.y;
    const int k0 = split * g.ksplit;
    const int k1 = (k0 + g.ksplit < g.K) ? k0 + g.ksplit : g.K;
    const int kc = k1 - k0;                                              // (a multiple of 4: K and ksplit are)
    for (int i = tid; i < MT * g.ksplit; i += 256) {
        const int m = i / g.ksplit, k = i - m * g.ksplit;
        a_lds[i] = (m < g.M && k < kc) ? g.A[(long)m * g.sam + (long)(k0 + k) * g.sak] : 0.f;
    }
    __syncthreads();
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < g.N) {
        const float* bp = g.B + (long)k0 * g.sbk + n;
#pragma unroll 2
        for (int k = 0; k < kc; k += 4) {
            f32x4 b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bp + (long)(k + u) * g.sbk));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(a_lds + m * g.ksplit + k);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[m] += a[u] * b[u];
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (m < g.M) *reinterpret_cast<f32x4*>(g.part + ((size_t)split * g.M + m) * g.N + n) = acc[m];
    }
}

template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_nt_kernel(const GemmArgs g) {
    constexpr int R = 16 / MT;                                         // rows of W per wave and trip
    extern __shared__ __attribute__((aligned(16))) float a_lds[];      // [MT][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < MT * g.K; i += 256) {
        const int m = i / g.K, k = i - m * g.K;
        a_lds[i] = (m < g.M) ? g.A[(long)m * g.sam + (long)k * g.sak] : 0.f;
    }
    __syncthreads();
    const long trips = ((long)g.N + R - 1) / R;
    for (long t = (long)blockIdx.x * 4 + wave; t < trips; t += (long)gridDim.x * 4) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
        const float* wrow[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long n = t * R + r;
            wrow[r] = g.B + (n < g.N ? n : (long)g.N - 1) * g.sbn;        // (a surplus row re-reads the last one; never written)
        }
        for (int k = lane * 4; k < g.K; k += 256) {
            f32x4 w[R];
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wrow[r] + k));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(a_lds + m * g.K + k);
#pragma unroll
                for (int r = 0; r < R; ++r) v[r * MT + m] += (a[0] * w[r][0] + a[1] * w[r][1]) + (a[2] * w[r][2] + a[3] * w[r][3]);
            }
        }
        // transposing butterfly: afterwards v[0] of lane l is value (l >> 2) summed over the lanes that differ from l in bits 2..5
#pragma unroll
        for (int o = 32, c = 8; c >= 1; o >>= 1, c >>= 1) {
            const bool hi = lane & o;
#pragma unroll
            for (int i = 0; i < c; ++i) {
                const float send = hi ? v[i] : v[i + c];
                const float keep = hi ? v[i + c] : v[i];
                v[i] = keep + __shfl_xor(send, o);
            }
        }
        float sum = v[0];
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 1);
        // bits 5..2 of the lane chose the upper (1) or lower half at o = 32, 16, 8, 4: value index = those bits, high to low
        const int idx = lane >> 2, r = idx / MT, m = idx - r * MT;
        const long n = t * R + r;
        if ((lane & 3) == 0 && n < g.N && m < g.M) {
            float* c = g.C + (long)m * g.scm + n * g.scn;
            const float out = g.alpha * sum;
            *c = g.accumulate ? *c + out : out;
        }
    }
}

// Adds the split-K partials in a fixed order and applies the epilogue.  A workgroup owns 64 consecutive elements of C; its
// four waves each add a quarter of the splits (eight independent running sums: 512 splits of the skinny forward are
// 16 rounds of loads per thread, not 512), wave 0 adds the four quarter sums.
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const GemmArgs g) {
    __shared__ float quarter[3][64];
    const size_t total = (size_t)g.M * g.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const int per = (g.splits + 3) / 4, s0 = wave * per, s1 = (s0 + per < g.splits) ? s0 + per : g.splits;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    if (i < total) {
        int sp = s0;
        for (; sp + 8 <= s1; sp += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += g.part[(size_t)(sp + u) * total + i];
        for (; sp < s1; ++sp) acc[0] += g.part[(size_t)sp * total + i];
    }
    const float mine = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (wave) quarter[wave - 1][lane] = mine;
    __syncthreads();
    if (wave == 0 && i < total) {
        const float s = (mine + quarter[0][lane]) + (quarter[1][lane] + quarter[2][lane]);
        const int mm = (int)(i / g.N), n = (int)(i % g.N);
        float* c = g.C + (long)mm * g.scm + (long)n * g.scn;
        float v = g.alpha * s + (g.bias ? g.bias[n] : 0.f);
        v = gemm_act(v, g.act);
        *c = g.accumulate ? *c + v : v;
    }
}

}  // namespace
}  // namespace srx

using namespace srx;

#define SRX_LAUNCHED(what)                                                                                 \
    do {                                                                                                   \
        hipError_t e_ = hipGetLastError();                                                                 \
        if (e_ != hipSuccess) return set_error(SRX_ERR_LAUNCH, what ": %s", hipGetErrorString(e_));         \
        return SRX_OK;                                                                                     \
    } while (0)

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" {

int srx_maxpool2x2(const float* in, float* out, int N, int H, int W, int C, srx_stream_t stream) {
    if (!in || !out) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return set_error(SRX_ERR_BAD_ARG, "maxpool: bad dims (C must be a multiple of 4)");
    if (!al16(in) || !al16(out)) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_blocks((size_t)N * OH * OW * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)in, (f32x4*)out, N, H, W, C / 4, OH, OW);
    SRX_LAUNCHED("maxpool");
}

int srx_maxpool2x2_bwd_masked(const float* x, const float* dout, float* din, int N, int H, int W, int C, int mask_act, srx_stream_t stream) {
    if (mask_act < SRX_ACT_NONE || mask_act > SRX_ACT_SIGMOID) return set_error(SRX_ERR_BAD_ARG, "bad mask_act");
    if (!x || !dout || !din) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return set_error(SRX_ERR_BAD_ARG, "maxpool: bad dims (C must be a multiple of 4)");
    if (!al16(x) || !al16(dout) || !al16(din)) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_blocks((size_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)x, (const f32x4*)dout, (f32x4*)din, N, H, W, C / 4, (H + 1) / 2, (W + 1) / 2, mask_act);
    SRX_LAUNCHED("maxpool_bwd");
}

int srx_maxpool2x2_bwd(const float* x, const float* dout, float* din, int N, int H, int W, int C, srx_stream_t stream) {
    return srx_maxpool2x2_bwd_masked(x, dout, din, N, H, W, C, SRX_ACT_NONE, stream);
}

int srx_subsample2(const float* in, float* out, int N, int H, int W, int C, int oy, int ox, srx_stream_t stream) {
    if (!in || !out) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (H & 1) || (W & 1) || (oy & ~1) || (ox & ~1))
        return set_error(SRX_ERR_BAD_ARG, "subsample2: H, W must be even, C a multiple of 4, offsets 0 or 1");
    if (!al16(in) || !al16(out)) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    hipLaunchKernelGGL(subsample2_fwd_kernel, dim3(ew_blocks((size_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, (const f32x4*)in, (f32x4*)out, N, H, W, C / 4, oy, ox);
    SRX_LAUNCHED("subsample2");
}

int srx_subsample2_bwd(const float* dout, float* din, int N, int H, int W, int C, int oy, int ox, srx_stream_t stream) {
    if (!dout || !din) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (H & 1) || (W & 1) || (oy & ~1) || (ox & ~1))
        return set_error(SRX_ERR_BAD_ARG, "subsample2: H, W must be even, C a multiple of 4, offsets 0 or 1");
    if (!al16(dout) || !al16(din)) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(ew_blocks((size_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)dout, (f32x4*)din, N, H, W, C / 4, oy, ox);
    SRX_LAUNCHED("subsample2_bwd");
}

int srx_channel_blocks_to_nhwc(const float* blocked, float* plain, size_t pixels, int blocks, srx_stream_t stream) {
    if (!blocked || !plain) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (pixels == 0 || blocks <= 0) return set_error(SRX_ERR_BAD_ARG, "bad dims");
    if (!al16(blocked) || !al16(plain)) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    hipLaunchKernelGGL(deblock_kernel, dim3(ew_blocks(pixels * blocks * 16)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)blocked, (f32x4*)plain, pixels, blocks);
    SRX_LAUNCHED("channel_blocks_to_nhwc");
}

int srx_nhwc_to_channel_blocks(const float* plain, float* blocked, size_t pixels, int blocks, srx_stream_t stream) {
    if (!blocked || !plain) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (pixels == 0 || blocks <= 0) return set_error(SRX_ERR_BAD_ARG, "bad dims");
    if (!al16(blocked) || !al16(plain)) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    hipLaunchKernelGGL(block_kernel, dim3(ew_blocks(pixels * blocks * 16)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)plain, (f32x4*)blocked, pixels, blocks);
    SRX_LAUNCHED("nhwc_to_channel_blocks");
}

int srx_channel_normalize(const float* x, float* y, size_t pixels, int C, float eps, srx_stream_t stream) {
    if (!x || !y) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (pixels == 0 || C <= 0) return set_error(SRX_ERR_BAD_ARG, "bad dims");
    const size_t nb = (pixels + 3) / 4;
    hipLaunchKernelGGL(chan_norm_fwd_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, x, y, pixels, C, eps);
    SRX_LAUNCHED("channel_normalize");
}

int srx_channel_normalize_bwd(const float* x, const float* dy, float* dx, size_t pixels, int C, float eps, srx_stream_t stream) {
    if (!x || !dy || !dx) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (pixels == 0 || C <= 0) return set_error(SRX_ERR_BAD_ARG, "bad dims");
    const size_t nb = (pixels + 3) / 4;
    hipLaunchKernelGGL(chan_norm_bwd_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, pixels, C, eps);
    SRX_LAUNCHED("channel_normalize_bwd");
}

int srx_extract_patches16(const float* x, float* patches, int N, int H, int W, int C, int inverse, srx_stream_t stream) {
    if (!x || !patches) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (H & 15) || (W & 15))
        return set_error(SRX_ERR_BAD_ARG, "extract_patches16: H, W must be multiples of 16, C of 4");
    if (!al16(x) || !al16(patches)) return set_error(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    const size_t total = (size_t)N * H * W * (C / 4);
    if (inverse)   // x is the PATCH layout, `patches` the image
        hipLaunchKernelGGL(patches16_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)patches, N, H, W, C / 4);
    else
        hipLaunchKernelGGL(patches16_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)patches, N, H, W, C / 4);
    SRX_LAUNCHED("extract_patches16");
}

int srx_log_loss(const float* p, float label, int n, float eps, float loss_scale, float grad_scale, float* loss_out,
                 int accumulate, float* dp, srx_stream_t stream) {
    if (!p) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (n <= 0) return set_error(SRX_ERR_BAD_ARG, "bad count");
    hipLaunchKernelGGL(log_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, label, n, eps, loss_scale, grad_scale, loss_out,
                       accumulate, dp);
    SRX_LAUNCHED("log_loss");
}

int srx_vgg_preprocess(const float* in, float* out, size_t pixels, int backward, srx_stream_t stream) {
    if (!in || !out) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (in == out) return set_error(SRX_ERR_BAD_ARG, "vgg_preprocess cannot run in place (it reverses the channels)");
    if (pixels == 0) return SRX_OK;
    hipLaunchKernelGGL(vgg_pre_kernel, dim3(ew_blocks(pixels * 3)), dim3(256), 0, (hipStream_t)stream, in, out, pixels, backward);
    SRX_LAUNCHED("vgg_preprocess");
}

int srx_resize_bicubic_tf(const float* in, float* out, int N, int H, int W, int C, int OH, int OW, srx_stream_t stream) {
    if (!in || !out) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return set_error(SRX_ERR_BAD_ARG, "bad resize dims");
    if (in == out) return set_error(SRX_ERR_BAD_ARG, "resize cannot run in place");
    hipLaunchKernelGGL(resize_bicubic_tf_kernel, dim3(ew_blocks((size_t)N * OH * OW * C)), dim3(256), 0, (hipStream_t)stream, in, out, N, H,
                       W, C, OH, OW, (float)H / (float)OH, (float)W / (float)OW);
    SRX_LAUNCHED("resize_bicubic_tf");
}

int srx_add_scaled(const float* a, const float* b, float* out, size_t n, float alpha, float beta, srx_stream_t stream) {
    if (!a || !out) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (n == 0) return SRX_OK;
    hipLaunchKernelGGL(add_scaled_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, alpha, beta);
    SRX_LAUNCHED("add_scaled");
}

int srx_column_sums(const float* a, float* out, int rows, int cols, int ld, srx_stream_t stream) {
    if (!a || !out) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (rows <= 0 || cols <= 0 || ld < cols) return set_error(SRX_ERR_BAD_ARG, "bad dims");
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, out, rows, cols, ld);
    SRX_LAUNCHED("column_sums");
}

// split-K of the 64 x 64-tile kernel: 0 = none
static long gemm_tile_splits(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (tiles >= 128 || K < 512) return 0;
    long splits = (256 + tiles - 1) / tiles;
    if (splits > K / 64) splits = K / 64;
    return splits < 2 ? 0 : splits;
}
// split-K of gemm_skinny_nn_kernel (M <= 16, a long K, whole float4s): K ranges of 256 or more, about two workgroups per CU
static long gemm_skinny_nn_splits(int M, int N, int K) {
    if (M > 16 || K < 4096 || (K & 3) || (N & 3) || N < 256) return 0;
    const long nx = (N + 1023) / 1024;
    long splits = 512 / nx;
    if (splits > K / 256) splits = K / 256;
    return splits < 2 ? 0 : splits;
}

size_t srx_gemm_workspace_bytes(int M, int N, int K, int batch) {
    if (batch != 1 || M <= 0 || N <= 0 || K <= 0) return 0;
    const long a = gemm_tile_splits(M, N, K), b = gemm_skinny_nn_splits(M, N, K);
    return (size_t)(a > b ? a : b) * M * N * sizeof(float);
}

int srx_gemm(const srx_gemm_desc* d, const float* A, const float* B, const float* bias, float* C, void* ws, size_t ws_bytes,
             srx_stream_t stream) {
    if (!d || !A || !B || !C) return set_error(SRX_ERR_BAD_ARG, "null pointer");
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return set_error(SRX_ERR_BAD_ARG, "gemm: non-positive dimension");
    if (d->act < SRX_ACT_NONE || d->act > SRX_ACT_SIGMOID) return set_error(SRX_ERR_BAD_ARG, "bad activation");
    if (d->batch > 65535) return set_error(SRX_ERR_UNSUPPORTED, "gemm: more than 65535 batches per call");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.part = nullptr;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.sam = d->a_row_stride; g.sak = d->a_col_stride; g.sbk = d->b_row_stride; g.sbn = d->b_col_stride;
    g.scm = d->c_row_stride; g.scn = d->c_col_stride;
    g.batchA = d->a_batch_stride; g.batchB = d->b_batch_stride; g.batchC = d->c_batch_stride;
    g.alpha = d->alpha; g.act = d->act; g.accumulate = d->accumulate;
    g.splits = 1; g.ksplit = (d->K + 3) / 4 * 4;
    const size_t need = srx_gemm_workspace_bytes(d->M, d->N, d->K, d->batch);
    const bool have_ws = need && ws && ws_bytes >= need && al16(ws);
    const bool rows_a = d->a_col_stride == 1 || d->M == 1;
    // skinny routes (M <= 16, one matrix): forward of a dense layer ...
    const long nn_splits = d->batch == 1 ? gemm_skinny_nn_splits(d->M, d->N, d->K) : 0;
    // (its K slice of the M <= 16 rows of A lives in dynamic LDS: only while that stays below the 64-KiB launch limit --
    // N = K = 65536 with M <= 4 would ask for 128 KiB; such shapes fall through to the 64 x 64-tile kernel)
    const int nn_mt = d->M <= 4 ? 4 : (d->M <= 8 ? 8 : 16);
    const size_t nn_lds = nn_splits ? (size_t)nn_mt * (size_t)(((d->K + nn_splits - 1) / nn_splits + 15) / 16 * 16) * sizeof(float) : 0;
    if (nn_splits && nn_lds <= 48 * 1024 && have_ws && d->b_col_stride == 1 && (d->b_row_stride & 3) == 0 && al16(B)) {
        g.splits = (int)nn_splits;
        g.ksplit = (int)(((d->K + nn_splits - 1) / nn_splits + 15) / 16 * 16);
        g.part = (float*)ws;
        const int mt = nn_mt;
        const dim3 grid((unsigned)((d->N + 1023) / 1024), (unsigned)g.splits);
        const size_t lds = (size_t)mt * g.ksplit * sizeof(float);
        if (mt == 4) hipLaunchKernelGGL(gemm_skinny_nn_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, g);
        else if (mt == 8) hipLaunchKernelGGL(gemm_skinny_nn_kernel<8>, grid, dim3(256), lds, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(gemm_skinny_nn_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, g);
        hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)(((size_t)d->M * d->N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, g);
        SRX_LAUNCHED("gemm");
    }
    // ... and its data gradient: the rows of B are the contiguous ones
    if (d->batch == 1 && d->M <= 16 && d->N >= 2048 && d->b_row_stride == 1 && (d->b_col_stride & 3) == 0 && al16(B) && rows_a &&
        (d->K & 255) == 0 && !bias && d->act == SRX_ACT_NONE) {
        const int mt = d->M <= 4 ? 4 : (d->M <= 8 ? 8 : 16);
        const size_t lds = (size_t)mt * d->K * sizeof(float);
        if (lds <= 48 * 1024) {
            const long trips = ((long)d->N + 16 / mt - 1) / (16 / mt);
            const unsigned grid = (unsigned)(trips < 4096 ? (trips + 3) / 4 : 1024);
            if (mt == 4) hipLaunchKernelGGL(gemm_skinny_nt_kernel<4>, dim3(grid), dim3(256), lds, (hipStream_t)stream, g);
            else if (mt == 8) hipLaunchKernelGGL(gemm_skinny_nt_kernel<8>, dim3(grid), dim3(256), lds, (hipStream_t)stream, g);
            else hipLaunchKernelGGL(gemm_skinny_nt_kernel<16>, dim3(grid), dim3(256), lds, (hipStream_t)stream, g);
            SRX_LAUNCHED("gemm");
        }
    }
    const long tile_splits = d->batch == 1 ? gemm_tile_splits(d->M, d->N, d->K) : 0;
    if (tile_splits && have_ws) {
        g.splits = (int)tile_splits;
        g.ksplit = ((d->K + g.splits - 1) / g.splits + 15) / 16 * 16;
        g.part = (float*)ws;
    }
    const long tiles = (long)((d->M + 63) / 64) * ((d->N + 63) / 64);
    if (tiles > 0x7fffffffL) return set_error(SRX_ERR_UNSUPPORTED, "gemm: too many tiles");
    // enough 64x64 units to give every SIMD one: a tile per WAVE (fewer operand loads per MFMA)
    const long units = tiles * d->batch;
    if (g.splits == 1 && d->M >= 48 && d->N >= 48 && units >= 1024) {
        const long blocks = (units + 3) / 4;
        if (blocks > 0x7fffffffL) return set_error(SRX_ERR_UNSUPPORTED, "gemm: too many tiles");
        hipLaunchKernelGGL(gemm_mfma_w64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, (d->M + 63) / 64,
                           (d->N + 63) / 64, units);
        SRX_LAUNCHED("gemm");
    }
    hipLaunchKernelGGL(gemm_mfma_kernel, dim3((unsigned)tiles, (unsigned)g.splits, (unsigned)d->batch), dim3(256), 0, (hipStream_t)stream, g);
    if (g.splits > 1)
        hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)(((size_t)d->M * d->N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, g);
    SRX_LAUNCHED("gemm");
}

}  // extern "C"
