/*
 * srx.h -- C ABI of libsrx.so, the MI355X (gfx950) super-resolution conv engine.
 *
 * The reference (imironhead/ml_super_resolution) has no FFI / plugin boundary: its
 * hot path is reached through TensorFlow-1.8's Python op API.  Each entry point below
 * names the reference call site(s) whose TF op it replaces (paths relative to
 * /root/reference).  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (e.g. a PyTorch-ROCm
 *     tensor's data_ptr()); fp32; activations NHWC, filters HWIO [KH,KW,Cin,Cout];
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); the
 *     library never synchronises the device and allocates nothing per call;
 *   - returns SRX_OK (0) or a negative srx_status; srx_last_error() gives the text
 *     (thread-local); bad shapes / alignment are errors, never undefined behaviour;
 *   - base pointers of activations / filters must be 16-byte aligned.
 */
#ifndef SRX_H_
#define SRX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* srx_stream_t; /* hipStream_t */

typedef enum {
    SRX_OK = 0,
    SRX_ERR_BAD_ARG = -1,     /* null pointer, non-positive dim, unknown enum */
    SRX_ERR_UNSUPPORTED = -2, /* shape outside the compiled kernel set (e.g. stride 3, a stride-2 data gradient) */
    SRX_ERR_WORKSPACE = -3,   /* workspace missing or too small */
    SRX_ERR_LAUNCH = -4,      /* HIP runtime reported an error at launch */
    SRX_ERR_ALIGN = -5        /* pointer not 16-byte aligned */
} srx_status;

typedef enum { SRX_PAD_SAME = 0, SRX_PAD_VALID = 1 } srx_pad_mode;

typedef enum {
    SRX_ACT_NONE = 0,
    SRX_ACT_RELU = 1,   /* vdsr/vdsr/model_vdsr.py:68, srcnn/srcnn.py:106,117 */
    SRX_ACT_TANH = 2,   /* espcn/espcn/model_espcn.py:36,46,120,126; srcnn/srcnn.py:128 */
    SRX_ACT_LRELU = 3,  /* leaky_relu(0.2): enet/enet/model_enet.py:130-146 */
    SRX_ACT_SIGMOID = 4 /* enet/enet/model_enet.py:160 */
} srx_act;

typedef enum { SRX_OP_FWD = 0, SRX_OP_BWD_DATA = 1, SRX_OP_BWD_FILTER = 2 } srx_conv_op;

/* One convolution layer.  N,H,W,Cin describe the layer INPUT x; the output is [N,OH,OW,Cout] with OH,OW from
 * pad_mode and stride, TensorFlow's geometry: SAME -> OH = ceil(H / stride), pad_total = max((OH-1) stride + KH - H, 0),
 * pad_before = pad_total / 2 (the odd pixel goes AFTER); VALID -> OH = (H - KH) / stride + 1. */
typedef struct {
    int32_t N, H, W, Cin, Cout, KH, KW;
    int32_t stride;        /* 1, or 2: tf.layers.conv2d(strides=2, padding='same') of ENet's discriminator
                            * (enet/enet/model_enet.py:136-146; 0 before / 1 after on an even-sized image).  Stride 2 is
                            * implemented by srx_conv2d_fwd and srx_conv2d_bwd_filter (+ _partials / _reduce / the
                            * workspace query; there dpre is the gradient at the half-resolution output), at a quarter of
                            * the stride-1 layer's MFMA work.  srx_conv2d_bwd_data(_acc) refuses it: the data gradient of
                            * a stride-2 layer is the stride-1 data gradient of the zero-stuffed upstream gradient
                            * (srx_subsample2_bwd below), and srx_conv3x3_blocked (> 64 channels) is stride 1: the
                            * stride-1 layer sampled at the odd positions (srx_subsample2). */
    int32_t pad_mode;      /* srx_pad_mode */
    int32_t act;           /* srx_act fused after bias */
    int32_t post_add_relu; /* activation applied AFTER the skip add: 0 none, SRX_ACT_RELU (1) relu(conv + skip) as in
                            * enet/enet/model_enet.py:8-31, SRX_ACT_LRELU (3) leaky ReLU -- the closing launch of a
                            * layer whose input channels are accumulated over several launches (see
                            * srx_conv2d_bwd_data_acc) */
    int32_t precision;     /* 0 = exact fp32 (v_mfma_f32_16x16x4_f32); the only mode */
    int32_t subpixel_r;    /* srx_conv2d_fwd only.  0 / 1: y is [N,OH,OW,Cout].  r > 1: the epilogue stores through the
                            * sub-pixel (depth-to-space) index map, y is [N,OH*r,OW*r,Cout/(r*r)]:
                            *   y[n, h*r+dy, w*r+dx, c] = conv[n, h, w, (dy*r+dx)*C + c]
                            * bit-identical to srx_conv2d_fwd followed by srx_depth_to_space, without the intermediate
                            * tensor (ESPCN's f3 layer + espcn/espcn/experiment_test.py:171-177 in one launch).
                            * Ignored (must be 0 or 1) by the backward entry points. */
} srx_conv_desc;

const char* srx_version(void);
const char* srx_last_error(void);

/* Selects the forward / dgrad kernel family for layers with >= 16 input channels:
 *   1 (default)  3x3 body layers (16..64 exact-fit input channels, none / ReLU / ReluGrad / residual
 *                epilogues; full-width tiles, or column strips for 64 -> 64 layers of images too wide
 *                for them) run on one workgroup per CU with a double-buffered LDS tile, everything but
 *                the MFMAs done by scalar and memory instructions; all other shapes use the kernels of
 *                path 0;
 *   0            two persistent workgroups per CU, one LDS tile each (conv_mfma_kernel), for every shape.
 * Same results bit for bit on both paths, with ONE exception: path 1 runs SRCNN's 5x5 32 -> 3 layer (from
 * SRX_KWROWS_MIN_PIXELS = 4096 output pixels) on conv_kwrows_kernel, which adds the kw partial sums of an output in
 * another order: equal to path 0 to rounding (<= 2e-6 of the output scale).  (Path 1's kernels for the RGB-input 9x9 / 5x5
 * layers, conv_pack3.hip, keep the order of the products and are bit-identical to path 0.)
 * A tuning / A-B switch (also: environment SRX_PIPE).  Returns the old value. */
int srx_set_conv_path(int pipelined);

/* Selects the filter-gradient (Conv2DBackpropFilter) kernel family:
 *   2 (default)  3x3 64 -> 64 layers on one workgroup per CU with a double-buffered LDS tile (full-width tiles:
 *                wgrad_pipe_kernel; images too wide for them: 32-column strips, wgrad_rows_strip_kernel); other
 *                shapes as path 1;
 *   1            the linear-walk kernels with two workgroups per CU (wgrad_lin_kernel / wgrad_lin_strip_kernel)
 *                wherever they apply; other shapes as path 0;
 *   0            the per-lane cursor kernel (wgrad_mfma_kernel) for every shape;
 *   < 0          back to the environment's default (SRX_WGRAD_LIN / SRX_WGRAD_PIPE / SRX_WGRAD_PIPE_STRIP).
 * The paths differ in how a layer's pixels are dealt out to workgroups, i.e. in the order of the fp32 additions:
 * results agree to rounding.  A tuning / A-B switch.  Returns the old value. */
int srx_set_wgrad_path(int path);

/* Bytes of caller-owned workspace an op uses.  BWD_FILTER: required (per-workgroup partials).
 * FWD / BWD_DATA: optional 256 bytes holding the tile counter of dynamic scheduling; with ws == NULL
 * those ops fall back to a static work split (same results, a few per cent slower at large sizes). */
size_t srx_conv2d_workspace_bytes(const srx_conv_desc* d, int op);

/* y = act(bias + x (*) w) [+ skip] [relu]
 * Replaces tf.layers.conv2d / tf.contrib.layers.convolution2d / tf.nn.conv2d +
 * tf.nn.bias_add + tf.nn.{relu,tanh} and the residual add:
 *   vdsr/vdsr/model_vdsr.py:62-76 (conv+bias+relu), :85-104 (conv+bias, sd + res)
 *   espcn/espcn/model_espcn.py:30-62, :117-134
 *   srcnn/srcnn.py:100-130
 *   enet/enet/model_enet.py:13-29, :63-113
 * bias, skip nullable.  skip has the output's shape. */
int srx_conv2d_fwd(const srx_conv_desc* d, const float* x, const float* w, const float* bias,
                   const float* skip, float* y, void* ws, size_t ws_bytes, srx_stream_t stream);

/* Conv2DBackpropInput with the UPSTREAM activation gradient fused:
 *   dx[n,h,w,ci] = sum_{kh,kw,co} dpre[n,h+pt-kh,w+pl-kw,co] * w[kh,kw,ci,co]
 *   dx_out = dx * act'(x_in)      (x_in = this layer's input = the previous layer's
 *                                  post-activation output; ReluGrad masks on x_in > 0)
 * dpre is the gradient w.r.t. this layer's PRE-activation output.  x_in nullable
 * (then in_act is ignored and plain dx is written).
 * Replaces the Conv2DBackpropInput + ReluGrad/TanhGrad pairs TF autodiff emits for
 * optimizer.minimize: vdsr/vdsr/model_vdsr.py:146-148,174;
 * espcn/espcn/model_espcn.py:87-89; srcnn/srcnn.py:155-157. */
int srx_conv2d_bwd_data(const srx_conv_desc* d, const float* dpre, const float* w,
                        const float* x_in, int in_act, float* dx_out, void* ws, size_t ws_bytes,
                        srx_stream_t stream);

/* The same without the activation mask but with an ACCUMULATE operand: dx_out = dx + dx_acc (dx_out may alias
 * dx_acc).  Layers wider than the kernels' 64 channels (ENet's discriminator, VGG-19: enet/enet/model_enet.py:
 * 118-162, enet/enet/model_vgg.py:65-99) keep their tensors as blocks of 64 channels; the data gradient of an
 * input block is the sum over the output blocks of one launch each:
 *   dx[ib] = sum_ob bwd_data(dpre[ob], w[ib][ob]);   then srx_act_bwd applies the activation mask once.
 * The forward counterpart needs no extra entry point: y[ob] = act(sum_ib conv(x[ib], w[ib][ob]) + bias) is
 * srx_conv2d_fwd with skip = the running sum, act = NONE and, in the last launch, post_add_relu = the layer's
 * activation.  The filter gradients of the block pairs are independent srx_conv2d_bwd_filter calls. */
int srx_conv2d_bwd_data_acc(const srx_conv_desc* d, const float* dpre, const float* w, const float* dx_acc,
                            float* dx_out, void* ws, size_t ws_bytes, srx_stream_t stream);

/* Conv2DBackpropFilter + BiasAddGrad (+ the L2 regulariser's gradient):
 *   dw[kh,kw,ci,co] = sum_{n,oh,ow} x[n,oh+kh-pt,ow+kw-pl,ci] * dpre[n,oh,ow,co]
 *                     + wd_scale * w[kh,kw,ci,co]          (if w_for_decay != NULL)
 *   dbias[co]       = sum_{n,oh,ow} dpre[n,oh,ow,co]       (if dbias != NULL)
 * Deterministic: fixed work partition, partials reduced in a fixed order.
 * Same reference call sites as srx_conv2d_bwd_data; the regulariser is
 * tf.contrib.layers.l2_regularizer(1e-4): vdsr/vdsr/model_vdsr.py:34,70,93,125. */
int srx_conv2d_bwd_filter(const srx_conv_desc* d, const float* x, const float* dpre, float* dw,
                          float* dbias, const float* w_for_decay, float wd_scale, void* ws,
                          size_t ws_bytes, srx_stream_t stream);

/* The same in two calls, for callers that want the (HBM-bound) reduction of the per-workgroup partials to run
 * on another stream than the (MFMA-bound) gradient kernel -- e.g. under the next layer's dgrad:
 *   srx_conv2d_bwd_filter_partials   fills `ws` with *n_partials partial filters;
 *   srx_conv2d_bwd_filter_reduce     sums them in a fixed order into dw / dbias (+ the regulariser term).
 * The caller orders the two (event / stream wait) and keeps `ws` untouched in between.
 * srx_conv2d_bwd_filter(...) == partials + reduce on one stream, bit for bit. */
int srx_conv2d_bwd_filter_partials(const srx_conv_desc* d, const float* x, const float* dpre, void* ws,
                                   size_t ws_bytes, int* n_partials, srx_stream_t stream);
int srx_conv2d_bwd_filter_reduce(const srx_conv_desc* d, const void* ws, int n_partials, float* dw,
                                 float* dbias, const float* w_for_decay, float wd_scale,
                                 srx_stream_t stream);

/* dpre = dy * act'(y)  (ReluGrad / TanhGrad on the post-activation tensor). */
int srx_act_bwd(const float* dy, const float* y, float* dpre, size_t numel, int act,
                srx_stream_t stream);

/* Sub-pixel (depth-to-space) index map, bit-exact permutation:
 *   out[n, h*r+dy, w*r+dx, c] = in[n, h, w, (dy*r+dx)*C + c]
 * in [N,H,W,C*r*r] -> out [N,H*r,W*r,C].  Replaces the host NumPy split/reshape/
 * concatenate of espcn/espcn/experiment_test.py:171-177 and the TF split/reshape/concat
 * of espcn/espcn/experiment_train.py:47-56. */
int srx_depth_to_space(const float* in, float* out, int N, int H, int W, int C, int r,
                       srx_stream_t stream);

/* Measurement aid, not an operator of the path: a plain streaming copy (nontemporal 16-byte loads and stores, the
 * launch shape of the sub-pixel kernels).  bench.py times it on the sub-pixel map's own buffers: what a kernel that
 * only moves these bytes reaches at this transfer size is the ceiling the map is compared with
 * (`subpixel.copy_ceiling_gbps`).  in / out 16-byte aligned, bytes a multiple of 16.  No reference counterpart. */
int srx_stream_copy(const void* in, void* out, size_t bytes, srx_stream_t stream);

/* ESPCN inference in ONE launch (espcn/espcn/model_espcn.py:117-134 + espcn/espcn/experiment_test.py:171-177):
 *   t1 = tanh(conv5x5(x; 3->64) + b1), t2 = tanh(conv3x3(t1; 64->32) + b2), y = conv3x3(t2; 32->3 r^2) + b3 (all SAME),
 *   hr[n, h r + dy, w r + dx, c] = y[n, h, w, (dy r + dx) 3 + c]
 * x [N,H,W,3], filters HWIO ([5,5,3,64], [3,3,64,32], [3,3,32,3 r^2]), hr [N,H r,W r,3], r in 2..4.  A workgroup chains
 * the three layers through LDS on a tile of <= 16x16 LR pixels (halo recomputed per tile): meant for latency-bound sizes
 * such as BASELINE configs[1] (batch 32 of 17x17 patches) and images of up to ~130 k pixels; beyond, the per-layer launches
 * do less work.
 * Bit-identical to the three srx_conv2d_fwd launches (the last with subpixel_r). */
int srx_espcn_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* w3, const float* b3, float* hr, int N, int H, int W, int r,
                      srx_stream_t stream);
/* The same launch as the FORWARD PASS OF A TRAIN STEP (espcn/espcn/model_espcn.py:117-134 under the trainer of
 * model_espcn.py:137-160): besides chaining the layers through LDS it writes what backward needs -- t1 [N,H,W,64] and
 * t2 [N,H,W,32] (post-tanh; every pixel by the one tile that owns it) -- and y [N,H,W,3 r^2] in sub-pixel space (the loss of
 * the reference is taken there), instead of the shuffled HR image.  Bit-identical to three srx_conv2d_fwd launches.
 * t1 / t2 16-byte aligned. */
int srx_espcn_forward_keep(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* w3, const float* b3, float* t1, float* t2, float* y, int N, int H, int W, int r,
                           srx_stream_t stream);

/* SRCNN inference in ONE launch (srcnn/srcnn.py:100-130, tf.contrib.layers.convolution2d x 3, padding 'VALID'):
 *   t1 = relu(conv9x9(x; 3->64) + b1), t2 = relu(conv1x1(t1; 64->32) + b2), y = tanh(conv5x5(t2; 32->3) + b3)
 * x [N,H,W,3] (H, W >= 13), filters HWIO ([9,9,3,64], [1,1,64,32], [5,5,32,3]), y [N,H-12,W-12,3].  A workgroup chains the
 * three layers through LDS for a tile of <= 15x15 output pixels; bit-identical to three srx_conv2d_fwd launches.  For
 * latency-bound problems (BASELINE configs[0]: one 243x243 image = 256 tiles, one per CU); halo pixels of the hidden
 * layers are recomputed per tile (x1.6), so large batches stay on the per-layer launches.  b1 / b2 16-byte aligned. */
int srx_srcnn_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* w3, const float* b3, float* y, int N, int H, int W, srx_stream_t stream);

/* Inverse map (HR image -> sub-pixel label layout): in [N,H*r,W*r,C] -> out [N,H,W,C*r*r].
 * Replaces espcn/espcn/dataset.py:140-156 and espcn/espcn/experiment_test.py:91-96. */
int srx_space_to_depth(const float* in, float* out, int N, int H, int W, int C, int r,
                       srx_stream_t stream);

/* tf.losses.mean_squared_error(reduction=MEAN) forward + gradient:
 *   *loss_out (+)= sum((pred-target)^2) * inv_numel     (device scalar; accumulate != 0 adds)
 *   dpred      = 2 * (pred-target) * inv_numel          (dpred nullable)
 * vdsr/vdsr/model_vdsr.py:120-123, espcn/espcn/model_espcn.py:76-77.
 * scratch: >= srx_reduce_scratch_bytes() bytes of device memory. */
int srx_mse_fwd_bwd(const float* pred, const float* target, size_t numel, float inv_numel,
                    float* loss_out, int accumulate, float* dpred, void* scratch,
                    srx_stream_t stream);

/* *loss_out (+)= scale * sum(mask * w^2)/2  -- sum over kernels of
 * tf.contrib.layers.l2_regularizer(scale)(w), vdsr/vdsr/model_vdsr.py:34,125.
 * mask (nullable) has w's shape: 1 for regularised elements (kernels), 0 otherwise (biases), so
 * one launch covers a model's whole flat parameter buffer. */
int srx_l2_loss(const float* w, const float* mask, size_t numel, float scale, float* loss_out,
                int accumulate, void* scratch, srx_stream_t stream);

size_t srx_reduce_scratch_bytes(void);

/* TF-1.x AdamOptimizer._apply_dense over a flat parameter buffer ("epsilon hat"):
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m=b1*m+(1-b1)*g; v=b2*v+(1-b2)*g*g;
 *   w -= lr_t*m/(sqrt(v)+eps);  g is multiplied by grad_scale first (1/world for DP sums).
 * t = 1-based step.  vdsr/vdsr/model_vdsr.py:145-148; espcn/espcn/model_espcn.py:87-89;
 * srcnn/srcnn.py:155-157 (b1 .5, b2 .9). */
int srx_adam_tf_step(float* w, const float* g, float* m, float* v, size_t numel, float lr,
                     float beta1, float beta2, float eps, int64_t t, float grad_scale,
                     srx_stream_t stream);

/* The same update with the step count and the learning rate in DEVICE memory, so that a whole train step captured as a
 * HIP graph replays without a per-step kernel argument (ESPCN / SRCNN steps are a dozen launches of ~10 us:
 * espcn/makefile:30-36 trains 1.6 M of them).  `state` is 32 bytes of 16-byte-aligned device memory owned by the caller:
 *   { int64 t;      steps applied so far (0 before the first); the kernel uses t + 1 and advances it
 *     float lr;     the fed learning rate (the caller rewrites it when its schedule changes it)
 *     float lr_t;   out: the bias-corrected rate the last step used
 *     uint32 done, pad;   zero-initialised scratch }
 * lr_t = lr*sqrt(1-b2^(t+1))/(1-b1^(t+1)) is evaluated on the device in double precision, the expression of
 * srx_adam_tf_step.  Steps on one state block must be ordered on one stream. */
int srx_adam_tf_step_dev(float* w, const float* g, float* m, float* v, size_t numel, void* state,
                         float beta1, float beta2, float eps, float grad_scale, srx_stream_t stream);

/* tf.train.MomentumOptimizer(lr, mom) on gradients clipped element-wise to +-cap:
 *   g=clip(g*grad_scale,-cap,cap); acc=mom*acc+g; w-=lr*acc.
 * vdsr/vdsr/model_vdsr.py:158-184 (cap = 0.01/lr). */
int srx_momentum_clip_step(float* w, const float* g, float* acc, size_t numel, float lr,
                           float momentum, float cap, float grad_scale, srx_stream_t stream);

/* SRCNN's loss (srcnn/srcnn.py:142-144): rows = reshape(pred - target, [-1, row_len]);
 *   *loss_out = mean_rows( ||row||_2 )        dpred = d / (||row||_2 * rows)   (dpred nullable)
 * (the reference reshapes to [-1, bb*bb], so a row mixes channels: 3 rows per RGB image).
 * row_norms: caller-owned device scratch of `rows` floats. */
int srx_rownorm_loss_fwd_bwd(const float* pred, const float* target, size_t rows, size_t row_len,
                             float* loss_out, float* dpred, float* row_norms, srx_stream_t stream);

/* tf.image.psnr(a,b,max_val) per image: out[n] = 20log10(max) - 10log10(mean((a-b)^2)).
 * vdsr/vdsr/experiment_train.py:80-82, vdsr/vdsr/experiment_evaluate.py:57-60. */
int srx_psnr(const float* a, const float* b, float* out, int N, size_t per_image, float max_val,
             srx_stream_t stream);

/* tf.image.ssim(a, b, max_val) per image (TF-1.8 defaults: 11x11 gaussian window sigma 1.5, k1 .01,
 * k2 .03, VALID windows, mean over positions then channels):
 *   out[n] = mean_c mean_{oh,ow} [ (2 mu_a mu_b + c1)/(mu_a^2 + mu_b^2 + c1) *
 *                                  (2 cov_ab + c2)/(var_a + var_b + c2) ],  c_i = (k_i max_val)^2
 * a, b: [N,H,W,C] with H, W >= 11.  scratch: >= srx_ssim_scratch_bytes(N) bytes.
 * vdsr/vdsr/experiment_evaluate.py:57-60, vdsr/vdsr/experiment_resolve.py, espcn/espcn/experiment_test.py:55. */
int srx_ssim(const float* a, const float* b, float* out, int N, int H, int W, int C, float max_val,
             void* scratch, srx_stream_t stream);
size_t srx_ssim_scratch_bytes(int N);

/* tf.saturate_cast(x*127.5+127.5, uint8): clamp to [0,255], truncate.
 * vdsr/vdsr/experiment_resolve.py:65-69, espcn/espcn/experiment_train.py:58. */
int srx_saturate_u8(const float* x, uint8_t* out, size_t numel, srx_stream_t stream);

/* out = a*x + b (elementwise); used for the [0,255] <-> [-1,1] maps at the model edge
 * (espcn/espcn/experiment_test.py:160,179). */
int srx_affine(const float* x, float* out, size_t numel, float a, float b, srx_stream_t stream);

/* ---- "next" row N1: on-device low-resolution synthesis (vdsr/vdsr/dataset.py:13-38,95-115) ---- */

/* skimage.util.img_as_float32 of uint8 data: out = in / 255. */
int srx_u8_to_unit_float(const uint8_t* in, float* out, size_t numel, srx_stream_t stream);

/* skimage.filters.gaussian(image, sigma, mode='nearest') on [N,H,W,C] (channels are not blurred):
 * separable 1-D kernels of radius int(4*sigma + 0.5), weights exp(-x^2/(2 sigma^2)) normalised, borders
 * replicated (scipy.ndimage.gaussian_filter, truncate 4.0).  tmp: N*H*W*C floats.  sigma <= 0 copies. */
int srx_gaussian_blur(const float* in, float* out, float* tmp, int N, int H, int W, int C, float sigma,
                      srx_stream_t stream);

/* skimage.transform.resize(image, [OH, OW], mode='edge', anti_aliasing=False) with order 1: bilinear
 * sampling at in = (out + 0.5) * (H / OH) - 0.5, coordinates clamped to the image.  [N,H,W,C] -> [N,OH,OW,C]. */
int srx_resize_bilinear(const float* in, float* out, int N, int H, int W, int C, int OH, int OW,
                        srx_stream_t stream);

/* tf.image.resize_bicubic(images, [OH, OW]) with TensorFlow 1.x semantics (align_corners=False, no half-pixel centres:
 * in = out * IN / OUT; cubic kernel A = -0.75 evaluated on TF's 1024-step grid; taps clamped to the image): SRCNN's
 * in-graph degradation, srcnn/srcnn.py:89-93.  [N,H,W,C] -> [N,OH,OW,C].  An integer down-scaling factor is plain
 * decimation (weights 0,1,0,0).  Pinned against the reference's assets/srcnn_00{0,1}.jpg panels. */
int srx_resize_bicubic_tf(const float* in, float* out, int N, int H, int W, int C, int OH, int OW,
                          srx_stream_t stream);

/* tf.image.resize_nearest_neighbor by an integer factor (pixel replication):
 * in [N,H,W,C] -> out [N,H*f,W*f,C].  enet/enet/model_enet.py:78-80. */
int srx_upsample_nearest(const float* in, float* out, int N, int H, int W, int C, int f,
                         srx_stream_t stream);

/* Gradient of srx_upsample_nearest (what TF's ResizeNearestNeighborGrad computes for an integer factor):
 * din[n,h,w,c] = sum of the f x f block of dout.  dout [N,H*f,W*f,C] -> din [N,H,W,C].
 * Backward of enet/enet/model_enet.py:78-80 inside the generator's training graph (:331-337). */
int srx_upsample_nearest_bwd(const float* dout, float* din, int N, int H, int W, int C, int f,
                             srx_stream_t stream);

/* Test aid: fills the LDS of every CU with quiet NaNs (one launch).  A kernel that reads LDS it has not written must not
 * depend on what it finds there; the Python wrappers call this before every entry point when SRX_POISON_LDS=1, which turns such a
 * dependence into NaNs in the parity tests (round 4: conv_pack3_kernel read one float behind its tile).  No reference counterpart. */
int srx_debug_poison_lds(srx_stream_t stream);

/* Gradient through t = relu(x + f(x)) of a residual block (enet/enet/model_enet.py:8-31) where the
 * gradients via the skip path and via the conv path arrive separately:
 * out = (y > 0) ? a + b : 0, y = the block input as saved (post-ReLU).  out may alias a or b. */
int srx_add_relu_grad(const float* a, const float* b, const float* y, float* out, size_t numel,
                      srx_stream_t stream);

/* ---- row A14 / "next" row N4: EnhanceNet-PAT's loss side (discriminator, VGG-19 features, perceptual / texture /
 * adversarial losses: enet/enet/model_enet.py:118-261, enet/enet/model_vgg.py:11-99) -------------------------------- */

/* One launch for a whole 3x3 SAME stride-1 layer wider than 64 channels, on channel-blocked tensors (see
 * srx_conv2d_bwd_data_acc): x [staged_blocks][N,H,W,64] -> y [produced_blocks][N,H,W,64];
 * w = the FORWARD layer's filters as [CIB][COB][3][3][64][64] (block [ib][ob] = HWIO[:, :, 64 ib:64 ib+64, 64 ob:64 ob+64]).
 *   transpose_filters == 0 (forward):  y[ob] = act( sum_ib conv(x[ib], w[ib][ob]) + bias[64 ob ..] ),
 *                                      staged_blocks = CIB, produced_blocks = COB;
 *   transpose_filters != 0 (dgrad):    y[ib] = sum_ob conv(x[ob], flipped / transposed w[ib][ob]),
 *                                      staged_blocks = COB, produced_blocks = CIB, bias NULL, act NONE.
 * mask (nullable, produced-shaped, blocked): y *= act'(mask) with mask_act -- the activation gradient of the layer
 * below fused into a data-gradient launch (ReluGrad / leaky-ReLU gradient on its saved post-activation output).
 * Same arithmetic as the block-pair launches (exact fp32 MFMA); the sum over the staged blocks stays in registers.
 * VGG-19 blocks 2-5 and the discriminator's 128..512-channel layers: enet/enet/model_vgg.py:65-99,
 * enet/enet/model_enet.py:118-146.  act: SRX_ACT_NONE / RELU / LRELU.  Any image width (rows wider than 64 pixels are
 * cut into column strips). */
int srx_conv3x3_blocked(const float* x, const float* w, const float* bias, const float* mask, int mask_act, float* y,
                        int N, int H, int W, int staged_blocks, int produced_blocks, int act,
                        int transpose_filters, srx_stream_t stream);

/* The filter gradient of such a layer (the discriminator's 128..512-channel layers; tf.gradients of
 * tf.layers.conv2d, enet/enet/model_enet.py:118-146, 331-346): x [staged_blocks][N,H,W,64] (the layer's input),
 * dpre [produced_blocks][N,H,W,64] -> dw [CIB][COB][3][3][64][64] (the layout of `w` above), dbias [64 COB] (nullable).
 * Every (input block, output block) pair is a 64 -> 64 problem: ONE launch over all pairs and one reduction of the
 * per-workgroup partial filters (deterministic order) where the linear-walk kernel covers the shape (rows of up to
 * about 120 pixels), otherwise one srx_conv2d_bwd_filter per pair.  Workspace: ..._workspace_bytes, 16-byte aligned. */
size_t srx_conv3x3_blocked_bwd_filter_workspace_bytes(int N, int H, int W, int staged_blocks, int produced_blocks);
int srx_conv3x3_blocked_bwd_filter(const float* x, const float* dpre, float* dw, float* dbias, int N, int H, int W,
                                   int staged_blocks, int produced_blocks, void* ws, size_t ws_bytes, srx_stream_t stream);

/* The texture-matching statistics of texture_matching_loss (enet/enet/model_enet.py:225-259) in one pass:
 *   gram[n*P + p] = patches_p^T patches_p,   patches = extract_image_patches(16x16, stride 16) of normalize(x)
 *   (normalize: x / (mean over channels + eps), :34-41) -- x [N,H,W,C] -> gram [N*(H/16)*(W/16), C, C].
 * The same numbers as srx_channel_normalize -> srx_extract_patches16 -> srx_gemm(trans_a) without the two feature-sized
 * intermediates; exact-fp32 MFMA.  C = 64, 128 or 256 (VGG-19 block1/2/3_conv1), H and W multiples of 16.
 * _bwd: dx = d loss / d x given dgram = d loss / d gram, SYMMETRIC (the difference of two gram matrices is):
 *   dn = alpha * n dgram (alpha = 2), pushed through the patch permutation and the gradient of normalize. */
int srx_texture_gram(const float* x, float* gram, int N, int H, int W, int C, float eps, srx_stream_t stream);
int srx_texture_gram_bwd(const float* x, const float* dgram, float* dx, int N, int H, int W, int C, float eps, float alpha,
                         srx_stream_t stream);

/* The reference's uint8 image resizes, byte for byte: scipy.misc.imresize (enet/enet/datasets.py:110-111: 25 % bilinear
 * down, 400 % bicubic up; enet/enet/experiment_resolve.py:78-79: 400 % bicubic) is Pillow's Image.resize on the uint8
 * image -- two passes (horizontal, then vertical) of  out = clip8((2^21 + sum_k kk[k] * in[xmin + k]) >> 22)  with
 * integer coefficients (libImaging/Resample.c).  Pinned by the reference's own output, assets/enet_eagle_bq.png.
 *   srx_pil_resample_ksize / _coeffs: the coefficient tables of one axis, computed on the HOST in double precision as
 *     Pillow does: bounds [out_size][2] = (first input index, count), kk [out_size][ksize].
 *   srx_resample_u8: one pass on the device over [outer][in_size][inner] -> [outer][out_size][inner] (horizontal pass of
 *     an [N,H,W,C] image: outer = N*H, inner = C; vertical: outer = N, inner = W*C); bounds / kk are DEVICE copies.
 *   srx_u8_to_pm1: astype(float32) / 127.5 - 1.0 (two roundings). */
typedef enum { SRX_RESAMPLE_BILINEAR = 0, SRX_RESAMPLE_BICUBIC = 1 } srx_resample_filter;
int srx_pil_resample_ksize(int in_size, int out_size, int filter);
int srx_pil_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk);
int srx_resample_u8(const uint8_t* in, uint8_t* out, long outer, int in_size, int out_size, long inner,
                    const int32_t* bounds, const int32_t* kk, int ksize, srx_stream_t stream);
int srx_u8_to_pm1(const uint8_t* in, float* out, size_t n, srx_stream_t stream);

/* tf.nn.max_pool(ksize 2x2, strides 2x2, padding='SAME') (enet/enet/model_vgg.py:28-36): [N,H,W,C] ->
 * [N,ceil(H/2),ceil(W/2),C], C % 4 == 0.  _bwd = MaxPoolGrad given the forward INPUT x: the gradient of a window goes
 * to its first maximum in scan order. */
int srx_maxpool2x2(const float* in, float* out, int N, int H, int W, int C, srx_stream_t stream);
int srx_maxpool2x2_bwd(const float* x, const float* dout, float* din, int N, int H, int W, int C,
                       srx_stream_t stream);
/* The same with the gradient of the activation that produced x fused in: din *= act'(x) (x is the post-activation
 * tensor; mask_act = SRX_ACT_RELU for VGG-19, whose every pooling layer follows a ReLU convolution:
 * enet/enet/model_vgg.py:11-36).  Same bits as srx_maxpool2x2_bwd followed by srx_act_bwd, one pass instead of two. */
int srx_maxpool2x2_bwd_masked(const float* x, const float* dout, float* din, int N, int H, int W, int C, int mask_act,
                              srx_stream_t stream);

/* out[n,i,j,:] = in[n,2i+oy,2j+ox,:] ([N,H,W,C] -> [N,H/2,W/2,C]; H, W even, C % 4 == 0), and its gradient (zero
 * stuffing): din[n,h,w,:] = (h%2==oy && w%2==ox) ? dout[n,h/2,w/2,:] : 0.
 * tf.layers.conv2d(kernel_size=3, strides=2, padding='same') on an even-sized image pads 0 before / 1 after
 * (enet/enet/model_enet.py:136-146), i.e. it IS the stride-1 SAME convolution sampled at (oy, ox) = (1, 1); its
 * gradients are the stride-1 gradients of the zero-stuffed upstream gradient. */
int srx_subsample2(const float* in, float* out, int N, int H, int W, int C, int oy, int ox, srx_stream_t stream);
int srx_subsample2_bwd(const float* dout, float* din, int N, int H, int W, int C, int oy, int ox,
                       srx_stream_t stream);

/* Channel-blocked [blocks][pixels][64] <-> NHWC [pixels][blocks*64] (see srx_conv2d_bwd_data_acc). */
int srx_channel_blocks_to_nhwc(const float* blocked, float* plain, size_t pixels, int blocks, srx_stream_t stream);
int srx_nhwc_to_channel_blocks(const float* plain, float* blocked, size_t pixels, int blocks, srx_stream_t stream);

/* normalize() of enet/enet/model_enet.py:34-41: y = x / (mean over channels + eps) per pixel, x [pixels, C]; and its
 * gradient dx = dy / m - sum_c(dy x) / (C m^2), m = mean + eps. */
int srx_channel_normalize(const float* x, float* y, size_t pixels, int C, float eps, srx_stream_t stream);
int srx_channel_normalize_bwd(const float* x, const float* dy, float* dx, size_t pixels, int C, float eps,
                              srx_stream_t stream);

/* tf.extract_image_patches(ksizes 16x16, strides 16x16, 'VALID') + reshape to [N, (H/16)*(W/16), 256, C]
 * (enet/enet/model_enet.py:237-250): x [N,H,W,C] -> patches; inverse != 0: patches (first argument) -> image
 * (second argument), which is also the gradient of the forward map.  H, W multiples of 16, C of 4. */
int srx_extract_patches16(const float* x, float* patches, int N, int H, int W, int C, int inverse,
                          srx_stream_t stream);

/* tf.losses.log_loss(labels = label everywhere, predictions = p, epsilon = eps, reduction=MEAN)
 * (enet/enet/model_enet.py:165-182):
 *   *loss_out (+)= loss_scale * mean_i( -label log(p_i + eps) - (1 - label) log(1 - p_i + eps) )
 *   dp_i = grad_scale / n * ( -label / (p_i + eps) + (1 - label) / (1 - p_i + eps) )       (dp nullable) */
int srx_log_loss(const float* p, float label, int n, float eps, float loss_scale, float grad_scale,
                 float* loss_out, int accumulate, float* dp, srx_stream_t stream);

/* VGG-19's input map (enet/enet/model_enet.py:288-289, enet/enet/model_vgg.py:72-76), [pixels, 3]:
 *   out[..., c] = (in[..., 2-c] * 127.5 + 127.5) - (103.939, 116.779, 123.68)[c]
 * backward != 0: the gradient map din[..., c] = 127.5 * dout[..., 2-c] (in = dout, out = din). */
int srx_vgg_preprocess(const float* in, float* out, size_t pixels, int backward, srx_stream_t stream);

/* out = alpha * a + beta * b (b nullable: out = alpha * a); out may alias a or b.  Sums of gradients that reach one
 * tensor from two consumers (sr_images feeds VGG-19 and the discriminator: enet/enet/model_enet.py:291-298) and the
 * loss weights of :204,:314-317. */
int srx_add_scaled(const float* a, const float* b, float* out, size_t n, float alpha, float beta,
                   srx_stream_t stream);

/* out[j] = sum_i a[i*ld + j]: bias gradient of tf.layers.dense (enet/enet/model_enet.py:148-160). */
int srx_column_sums(const float* a, float* out, int rows, int cols, int ld, srx_stream_t stream);

/* Exact-fp32 batched GEMM with general strides (v_mfma_f32_16x16x4_f32):
 *   C_b(m,n) = act( alpha * sum_k A_b(m,k) B_b(k,n) + bias(n) ) [+ C_b(m,n)]
 *   X_b(i,j) = X[b * x_batch_stride + i * x_row_stride + j * x_col_stride]        (strides in elements)
 * tf.layers.dense forward / backward (enet/enet/model_enet.py:148-160) and the gram matrices tf.matmul(x, x,
 * transpose_a=True) over 16x16 patches and their gradient (:252-255).  bias nullable.  ws: optional workspace of
 * srx_gemm_workspace_bytes() for a deterministic split of K when M*N is small (dense layers with M = batch). */
typedef struct {
    int32_t M, N, K, batch;
    int64_t a_row_stride, a_col_stride, a_batch_stride;
    int64_t b_row_stride, b_col_stride, b_batch_stride;
    int64_t c_row_stride, c_col_stride, c_batch_stride;
    float alpha;
    int32_t act;        /* srx_act */
    int32_t accumulate; /* != 0: add to C */
} srx_gemm_desc;
size_t srx_gemm_workspace_bytes(int M, int N, int K, int batch);
int srx_gemm(const srx_gemm_desc* d, const float* A, const float* B, const float* bias, float* C, void* ws,
             size_t ws_bytes, srx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H_ */
