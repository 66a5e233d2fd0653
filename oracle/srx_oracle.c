/*
 * srx_oracle.c -- CPU restatement of the super-resolution conv hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ml_super_resolution_amd/)
 * may link, load or call this file.  Allowed callers: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, always as the checker / the CPU figure beside the
 * GPU one, never as the thing shipped.
 *
 * PARITY STATUS: "parity unpinned" for the convolution / backward / optimizer
 * arithmetic.  The reference (imironhead/ml_super_resolution) delegates all of it
 * to TensorFlow 1.8 (pin: vdsr/makefile:11-14 --runtime-version=1.8), which is not
 * vendored in the reference tree, not installed here and not installable offline.
 * The reference has no tests and no golden vectors for these ops.
 * What IS pinned by reference data: the sub-pixel index map (three reference
 * spellings, see oracle.py), the post-ReLU taps and the residual-add + truncating
 * uint8 encode (tests/test_oracle_pins.py).  The arithmetic below restates the
 * published TF-1.8 op semantics at the reference's call sites:
 *
 *   Conv2D       cross-correlation, NHWC activations, HWIO filters, stride 1,
 *                explicit (pad_t, pad_l); TF "SAME" => pad_before = (k-1)/2,
 *                "VALID" => 0.        vdsr/vdsr/model_vdsr.py:62-70,85-93
 *                                     espcn/espcn/model_espcn.py:30-62,117-134
 *                                     srcnn/srcnn.py:100-130
 *   BiasAdd      + b[co]              same call sites (use_bias=True default)
 *   activations  relu / tanh / leaky_relu(0.2) / sigmoid
 *   ReluGrad     mask on y > 0; TanhGrad dy*(1-y^2)   (TF autodiff of the above)
 *   mean_squared_error(reduction=MEAN) = sum((a-b)^2)/numel
 *                                     vdsr/vdsr/model_vdsr.py:120-123
 *   l2_regularizer(s)(w) = s * sum(w^2)/2             vdsr/vdsr/model_vdsr.py:34
 *   AdamOptimizer (epsilon-hat form)  vdsr/vdsr/model_vdsr.py:145-148
 *   MomentumOptimizer + clip_by_value vdsr/vdsr/model_vdsr.py:158-184
 *   depth_to_space / space_to_depth   espcn/espcn/experiment_test.py:171-177,
 *                                     espcn/espcn/dataset.py:140-156
 *
 * Accumulation is fp32 in a fixed (kh, kw, ci) order with the co loop innermost
 * (vectorisable), i.e. what an Eigen/TF CPU build does up to summation order.
 * A float64 NumPy restatement of the same formulas lives in oracle.py and is the
 * tighter checker for small cases.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum { SRX_REF_ACT_NONE = 0, SRX_REF_ACT_RELU = 1, SRX_REF_ACT_TANH = 2,
       SRX_REF_ACT_LRELU = 3, SRX_REF_ACT_SIGMOID = 4 };

int srx_ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* the timing leg of bench.py sizes the thread pool to the CPUs the process is really granted */
int srx_ref_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

static inline float act_apply(float v, int act) {
    switch (act) {
    case SRX_REF_ACT_RELU:    return v > 0.0f ? v : 0.0f;
    case SRX_REF_ACT_TANH:    return tanhf(v);
    case SRX_REF_ACT_LRELU:   return v > 0.0f ? v : 0.2f * v;   /* enet/enet/model_enet.py:130-146 */
    case SRX_REF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default:                  return v;
    }
}

/* derivative of the activation expressed through its OUTPUT y (what TF's
 * ReluGrad / TanhGrad / SigmoidGrad consume). */
static inline float act_grad_from_y(float y, int act) {
    switch (act) {
    case SRX_REF_ACT_RELU:    return y > 0.0f ? 1.0f : 0.0f;
    case SRX_REF_ACT_TANH:    return 1.0f - y * y;
    case SRX_REF_ACT_LRELU:   return y > 0.0f ? 1.0f : 0.2f;
    case SRX_REF_ACT_SIGMOID: return y * (1.0f - y);
    default:                  return 1.0f;
    }
}

/* y[n,oh,ow,co] = act(b[co] + sum_{kh,kw,ci} x[n,oh+kh-pad_t,ow+kw-pad_l,ci] * w[kh,kw,ci,co])
 *                 (+ skip[n,oh,ow,co]) (then relu if post_relu)
 * vdsr/vdsr/model_vdsr.py:62-76 (conv+bias+relu), :85-104 (conv+bias, + sd_images). */
void srx_ref_conv2d_fwd(const float* x, const float* w, const float* b, const float* skip,
                        float* y, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                        int pad_t, int pad_l, int OH, int OW, int act, int post_relu) {
    const long rows = (long)N * OH;
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * (size_t)Cout);
#pragma omp for schedule(static)
        for (long r = 0; r < rows; ++r) {
            const int n = (int)(r / OH), oh = (int)(r % OH);
            for (int ow = 0; ow < OW; ++ow) {
                for (int co = 0; co < Cout; ++co) acc[co] = b ? b[co] : 0.0f;
                for (int kh = 0; kh < KH; ++kh) {
                    const int ih = oh + kh - pad_t;
                    if (ih < 0 || ih >= H) continue;
                    for (int kw = 0; kw < KW; ++kw) {
                        const int iw = ow + kw - pad_l;
                        if (iw < 0 || iw >= W) continue;
                        const float* xp = x + (((size_t)n * H + ih) * W + iw) * Cin;
                        const float* wp = w + ((size_t)kh * KW + kw) * Cin * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float xv = xp[ci];
                            const float* wr = wp + (size_t)ci * Cout;
                            for (int co = 0; co < Cout; ++co) acc[co] += xv * wr[co];
                        }
                    }
                }
                const size_t o = (((size_t)n * OH + oh) * OW + ow) * Cout;
                for (int co = 0; co < Cout; ++co) {
                    float v = act_apply(acc[co], act);
                    if (skip) v += skip[o + co];
                    if (post_relu) v = v > 0.0f ? v : 0.0f;
                    y[o + co] = v;
                }
            }
        }
        free(acc);
    }
}

/* dpre = dy * act'(y)   (TF ReluGrad on the post-activation tensor, etc.) */
void srx_ref_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) dpre[i] = dy[i] * act_grad_from_y(y[i], act);
}

/* Conv2DBackpropInput:
 * dx[n,h,w,ci] = sum_{kh,kw,co} dpre[n,h+pad_t-kh,w+pad_l-kw,co] * w[kh,kw,ci,co] */
void srx_ref_conv2d_bwd_data(const float* dpre, const float* w, float* dx, int N, int H, int W,
                             int Cin, int Cout, int KH, int KW, int pad_t, int pad_l,
                             int OH, int OW) {
    const long rows = (long)N * H;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const int n = (int)(r / H), h = (int)(r % H);
        for (int iw = 0; iw < W; ++iw) {
            float* dxp = dx + (((size_t)n * H + h) * W + iw) * Cin;
            for (int ci = 0; ci < Cin; ++ci) dxp[ci] = 0.0f;
            for (int kh = 0; kh < KH; ++kh) {
                const int oh = h + pad_t - kh;
                if (oh < 0 || oh >= OH) continue;
                for (int kw = 0; kw < KW; ++kw) {
                    const int ow = iw + pad_l - kw;
                    if (ow < 0 || ow >= OW) continue;
                    const float* dp = dpre + (((size_t)n * OH + oh) * OW + ow) * Cout;
                    const float* wp = w + ((size_t)kh * KW + kw) * Cin * Cout;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float* wr = wp + (size_t)ci * Cout;
                        float s = 0.0f;
                        for (int co = 0; co < Cout; ++co) s += dp[co] * wr[co];
                        dxp[ci] += s;
                    }
                }
            }
        }
    }
}

/* Conv2DBackpropFilter + BiasAddGrad:
 * dw[kh,kw,ci,co] = sum_{n,oh,ow} x[n,oh+kh-pad_t,ow+kw-pad_l,ci] * dpre[n,oh,ow,co]
 * db[co]          = sum_{n,oh,ow} dpre[n,oh,ow,co]
 * Accumulated in double per thread, reduced in thread order (deterministic for a
 * fixed thread count). */
void srx_ref_conv2d_bwd_filter(const float* x, const float* dpre, float* dw, float* db, int N,
                               int H, int W, int Cin, int Cout, int KH, int KW, int pad_t,
                               int pad_l, int OH, int OW) {
    const size_t wn = (size_t)KH * KW * Cin * Cout;
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    double* part = (double*)calloc((wn + (size_t)Cout) * (size_t)nt, sizeof(double));
    const long rows = (long)N * OH;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* pw = part + (wn + (size_t)Cout) * (size_t)tid;
        double* pb = pw + wn;
#pragma omp for schedule(static)
        for (long r = 0; r < rows; ++r) {
            const int n = (int)(r / OH), oh = (int)(r % OH);
            for (int ow = 0; ow < OW; ++ow) {
                const float* dp = dpre + (((size_t)n * OH + oh) * OW + ow) * Cout;
                for (int co = 0; co < Cout; ++co) pb[co] += dp[co];
                for (int kh = 0; kh < KH; ++kh) {
                    const int ih = oh + kh - pad_t;
                    if (ih < 0 || ih >= H) continue;
                    for (int kw = 0; kw < KW; ++kw) {
                        const int iw = ow + kw - pad_l;
                        if (iw < 0 || iw >= W) continue;
                        const float* xp = x + (((size_t)n * H + ih) * W + iw) * Cin;
                        double* wq = pw + ((size_t)kh * KW + kw) * Cin * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const double xv = xp[ci];
                            double* wr = wq + (size_t)ci * Cout;
                            for (int co = 0; co < Cout; ++co) wr[co] += xv * dp[co];
                        }
                    }
                }
            }
        }
    }
    for (size_t i = 0; i < wn; ++i) {
        double s = 0.0;
        for (int t = 0; t < nt; ++t) s += part[(wn + (size_t)Cout) * (size_t)t + i];
        dw[i] = (float)s;
    }
    if (db) {
        for (int co = 0; co < Cout; ++co) {
            double s = 0.0;
            for (int t = 0; t < nt; ++t) s += part[(wn + (size_t)Cout) * (size_t)t + wn + co];
            db[co] = (float)s;
        }
    }
    free(part);
}

/* Sub-pixel maps (pure index permutation, bit-exact).
 * d2s: out[n, h*r+dy, w*r+dx, c] = in[n, h, w, (dy*r+dx)*C + c]
 * espcn/espcn/experiment_test.py:171-177 (numpy), espcn/espcn/experiment_train.py:47-56 */
void srx_ref_depth_to_space(const float* in, float* out, int N, int H, int W, int C, int r) {
    const int Cin = C * r * r;
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                for (int dy = 0; dy < r; ++dy)
                    for (int dx = 0; dx < r; ++dx)
                        for (int c = 0; c < C; ++c)
                            out[((((size_t)n * H * r) + (size_t)h * r + dy) * W * r + (size_t)w * r + dx) * C + c] =
                                in[(((size_t)n * H + h) * W + w) * Cin + (dy * r + dx) * C + c];
}

/* s2d (inverse): out[n,h,w,(dy*r+dx)*C+c] = in[n,h*r+dy,w*r+dx,c]
 * espcn/espcn/dataset.py:140-156, espcn/espcn/experiment_test.py:91-96 */
void srx_ref_space_to_depth(const float* in, float* out, int N, int H, int W, int C, int r) {
    /* H, W are the LOW-resolution (output) spatial dims */
    const int Cout = C * r * r;
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                for (int dy = 0; dy < r; ++dy)
                    for (int dx = 0; dx < r; ++dx)
                        for (int c = 0; c < C; ++c)
                            out[(((size_t)n * H + h) * W + w) * Cout + (dy * r + dx) * C + c] =
                                in[((((size_t)n * H * r) + (size_t)h * r + dy) * W * r + (size_t)w * r + dx) * C + c];
}

/* loss = sum((pred-target)^2)/numel ; dpred = 2*(pred-target)/numel_global
 * vdsr/vdsr/model_vdsr.py:120-123.  numel_global lets a data-parallel shard scale
 * by the global batch.  Returns the local sum of squares / numel_global. */
double srx_ref_mse_fwd_bwd(const float* pred, const float* target, float* dpred, size_t numel,
                           double numel_global) {
    double s = 0.0;
    for (size_t i = 0; i < numel; ++i) {
        const double d = (double)pred[i] - (double)target[i];
        s += d * d;
        if (dpred) dpred[i] = (float)(2.0 * d / numel_global);
    }
    return s / numel_global;
}

/* tf.nn.l2_loss(w) = sum(w^2)/2 ; regulariser = scale * l2_loss   model_vdsr.py:34 */
double srx_ref_l2_loss(const float* w, size_t n) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += (double)w[i] * (double)w[i];
    return 0.5 * s;
}

/* TF-1.x AdamOptimizer._apply_dense ("epsilon hat"):
 *   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; w -= lr_t * m / (sqrt(v) + eps)
 * t is the 1-based step count.  vdsr/vdsr/model_vdsr.py:145-148,
 * espcn/espcn/model_espcn.py:87-89, srcnn/srcnn.py:155-157 */
void srx_ref_adam_tf(float* w, const float* g, float* m, float* v, size_t n, float lr, float b1,
                     float b2, float eps, long t) {
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) /
                               (1.0 - pow((double)b1, (double)t)));
    for (size_t i = 0; i < n; ++i) {
        const float gi = g[i];
        m[i] = b1 * m[i] + (1.0f - b1) * gi;
        v[i] = b2 * v[i] + (1.0f - b2) * gi * gi;
        w[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
    }
}

/* MomentumOptimizer(lr, 0.9) on gradients clipped element-wise to +-cap
 * (cap = 0.01 / lr): acc = mom*acc + g ; w -= lr*acc   model_vdsr.py:158-184 */
void srx_ref_momentum_clip(float* w, const float* g, float* acc, size_t n, float lr, float mom,
                           float cap) {
    for (size_t i = 0; i < n; ++i) {
        float gi = g[i];
        gi = gi < -cap ? -cap : (gi > cap ? cap : gi);
        acc[i] = mom * acc[i] + gi;
        w[i] -= lr * acc[i];
    }
}

/* tf.image.psnr(a, b, max_val) per image over H,W,C:
 *   20*log10(max_val) - 10*log10(mean((a-b)^2))   vdsr/vdsr/experiment_train.py:80-82 */
void srx_ref_psnr(const float* a, const float* b, float* out, int N, size_t per_image,
                  float max_val) {
    for (int n = 0; n < N; ++n) {
        double s = 0.0;
        for (size_t i = 0; i < per_image; ++i) {
            const double d = (double)a[n * per_image + i] - (double)b[n * per_image + i];
            s += d * d;
        }
        out[n] = (float)(20.0 * log10((double)max_val) - 10.0 * log10(s / (double)per_image));
    }
}

/* tf.saturate_cast(x*127.5+127.5, uint8): clamp to [0,255] then truncate toward zero
 * vdsr/vdsr/experiment_resolve.py:65-69, espcn/espcn/experiment_train.py:58 */
void srx_ref_saturate_u8(const float* x, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        float v = x[i] * 127.5f + 127.5f;
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        out[i] = (uint8_t)v;
    }
}
