"""
oracle.py -- NumPy (float64) restatement of the reference's super-resolution hot
path, plus a ctypes binding of the C restatement (srx_oracle.c).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module, and only as the checker / the CPU figure.
The product package (ml_super_resolution_amd) never imports it.

PARITY STATUS: "parity unpinned" for backward / optimizer arithmetic: the reference
(imironhead/ml_super_resolution) delegates the arithmetic to TensorFlow 1.8, which is
absent from /root/reference and not installable here, and the reference holds no
tests or golden vectors.  The FORWARD convolution is pinned by the
reference's own feature maps since round 3 (P7).  Pinned by reference data / code:
  P1  the sub-pixel index map: three independent reference spellings restated
      verbatim-in-behaviour below (`*_ref_spelling_*`) must agree with the closed
      form used everywhere else;
  P2  conv.N taps are post-ReLU (assets vdsr-fig2-conv.N == vdsr-fig2-relu.N);
  P3  residual add + truncating uint8 encode (assets vdsr-fig2-{sd,conv.20,sr});
  P4  SRCNN VALID geometry 243 -> 231;
  P5  scipy.misc.imresize == Pillow's integer resample: assets/enet_eagle_bq.png, 0 differing bytes;
  P6  tf.image.resize_bicubic (TF 1.x) reproduces the reference's SRCNN hd | sd panels to JPEG noise;
  P7  conv2d_fwd = 3x3 support + bias + ReLU clamp + zero SAME padding (+ residual add): weights fitted
      from assets/vdsr-fig2-conv.N.png off four held-out corners predict those corners, image border
      included, to the encoding's quantisation noise; edge / reflect padding miss the border 2-20x
      (tests/golden/make_pin_p7.py, tests/test_oracle_pins.py::test_p7_*).
All citations are relative to /root/reference.
"""
import ctypes
import os

import numpy as np

ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3, 4
_ACT_BY_NAME = {None: ACT_NONE, 'none': ACT_NONE, 'relu': ACT_RELU, 'tanh': ACT_TANH,
                'lrelu': ACT_LRELU, 'sigmoid': ACT_SIGMOID}


# ----------------------------------------------------------------------------
# geometry helpers (TF-1.8 padding semantics, stride 1)
# ----------------------------------------------------------------------------
def same_pad(k):
    """TF 'SAME', stride 1: pad_total = k-1, pad_before = pad_total // 2."""
    return (k - 1) // 2


def conv_geometry(H, W, KH, KW, padding):
    """-> (pad_t, pad_l, OH, OW) for stride-1 SAME / VALID."""
    if padding.upper() == 'SAME':
        return same_pad(KH), same_pad(KW), H, W
    if padding.upper() == 'VALID':
        return 0, 0, H - KH + 1, W - KW + 1
    raise ValueError(padding)


def srcnn_sanity_check(crop_image_size=256, fsub=33, f1=9, f2=1, f3=5):
    """srcnn/srcnn.py:28-40 (Python-2 integer division)."""
    smaller = fsub - f1 - f2 - f3 + 3
    boundary = (fsub - smaller) // 2
    crop = (crop_image_size - boundary * 2) // smaller
    return boundary, crop * smaller + boundary * 2   # (crop_image_side, crop_image_size)


# ----------------------------------------------------------------------------
# activations
# ----------------------------------------------------------------------------
def act_apply(v, act):
    act = _ACT_BY_NAME.get(act, act)
    if act == ACT_RELU:
        return np.maximum(v, 0.0)
    if act == ACT_TANH:
        return np.tanh(v)
    if act == ACT_LRELU:
        return np.where(v > 0, v, 0.2 * v)
    if act == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-v))
    return v


def act_grad_from_y(y, act):
    act = _ACT_BY_NAME.get(act, act)
    if act == ACT_RELU:
        return (y > 0).astype(y.dtype)
    if act == ACT_TANH:
        return 1.0 - y * y
    if act == ACT_LRELU:
        return np.where(y > 0, 1.0, 0.2).astype(y.dtype)
    if act == ACT_SIGMOID:
        return y * (1.0 - y)
    return np.ones_like(y)


# ----------------------------------------------------------------------------
# convolution (float64 NumPy)
# ----------------------------------------------------------------------------
def _pad_input(x, KH, KW, pad_t, pad_l, OH, OW):
    N, H, W, C = x.shape
    pad_b = max(OH + KH - 1 - pad_t - H, 0)
    pad_r = max(OW + KW - 1 - pad_l - W, 0)
    return np.pad(x, ((0, 0), (pad_t, pad_b), (pad_l, pad_r), (0, 0)))


def conv2d_fwd(x, w, b=None, padding='SAME', act=None, skip=None, post_relu=False,
               dtype=np.float64):
    """y = act(b + x (*) w) [+ skip] [relu].  NHWC x, HWIO w, cross-correlation.
    vdsr/vdsr/model_vdsr.py:62-76,85-104; espcn/espcn/model_espcn.py:117-134."""
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = w.shape
    pad_t, pad_l, OH, OW = conv_geometry(H, W, KH, KW, padding)
    xp = _pad_input(x, KH, KW, pad_t, pad_l, OH, OW)
    y = np.zeros((N, OH, OW, Cout), dtype)
    for kh in range(KH):
        for kw in range(KW):
            y += xp[:, kh:kh + OH, kw:kw + OW, :] @ w[kh, kw]
    if b is not None:
        y = y + np.asarray(b, dtype)
    y = act_apply(y, act)
    if skip is not None:
        y = y + np.asarray(skip, dtype)
    if post_relu:
        y = np.maximum(y, 0.0)
    return y


def conv2d_bwd_data(dpre, w, in_hw, padding='SAME', dtype=np.float64):
    """dx[n,h,w,ci] = sum dpre[n,h+pad_t-kh,w+pad_l-kw,co] w[kh,kw,ci,co]."""
    dpre = np.asarray(dpre, dtype)
    w = np.asarray(w, dtype)
    H, W = in_hw
    KH, KW, Cin, Cout = w.shape
    pad_t, pad_l, OH, OW = conv_geometry(H, W, KH, KW, padding)
    N = dpre.shape[0]
    dxp = np.zeros((N, OH + KH - 1, OW + KW - 1, Cin), dtype)
    for kh in range(KH):
        for kw in range(KW):
            dxp[:, kh:kh + OH, kw:kw + OW, :] += dpre @ w[kh, kw].T
    return dxp[:, pad_t:pad_t + H, pad_l:pad_l + W, :]


def conv2d_bwd_filter(x, dpre, ksize, padding='SAME', dtype=np.float64):
    """dw[kh,kw,ci,co] = sum x[n,oh+kh-pad_t,ow+kw-pad_l,ci] dpre[n,oh,ow,co]; db = sum dpre."""
    x = np.asarray(x, dtype)
    dpre = np.asarray(dpre, dtype)
    KH, KW = ksize
    N, H, W, Cin = x.shape
    Cout = dpre.shape[-1]
    pad_t, pad_l, OH, OW = conv_geometry(H, W, KH, KW, padding)
    xp = _pad_input(x, KH, KW, pad_t, pad_l, OH, OW)
    dw = np.zeros((KH, KW, Cin, Cout), dtype)
    d2 = dpre.reshape(-1, Cout)
    for kh in range(KH):
        for kw in range(KW):
            dw[kh, kw] = xp[:, kh:kh + OH, kw:kw + OW, :].reshape(-1, Cin).T @ d2
    return dw, d2.sum(axis=0)


# ----------------------------------------------------------------------------
# sub-pixel index map (pin P1)
# ----------------------------------------------------------------------------
def depth_to_space(x, r):
    """Closed form: out[n,h*r+dy,w*r+dx,c] = in[n,h,w,(dy*r+dx)*C+c]."""
    N, H, W, D = x.shape
    C = D // (r * r)
    return x.reshape(N, H, W, r, r, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, H * r, W * r, C)


def space_to_depth(x, r):
    """Inverse of depth_to_space."""
    N, HR, WR, C = x.shape
    H, W = HR // r, WR // r
    return x.reshape(N, H, r, W, r, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, H, W, r * r * C)


def d2s_ref_spelling_test(sr_result, sf):
    """Behavioural restatement of espcn/espcn/experiment_test.py:171-177 for one
    image [lrh, lrw, 3*sf*sf]: split along width, reshape, concatenate."""
    lrh, lrw, _ = sr_result.shape
    patches = np.split(sr_result, lrw, axis=1)
    patches = [np.reshape(p, [lrh * sf, sf, 3]) for p in patches]
    return np.concatenate(patches, axis=1)


def d2s_ref_spelling_train(batch, lr_patch_size, sf):
    """Behavioural restatement of espcn/espcn/experiment_train.py:49-56 for a batch
    [B, P, P, 3*sf*sf] -> [1, B*P*sf, P*sf, 3] (one tall strip, as the summary does)."""
    B = batch.shape[0]
    shape = [1, B * lr_patch_size * sf, sf, 3]
    subs = np.split(batch, lr_patch_size, axis=2)
    subs = [np.reshape(s, shape) for s in subs]
    return np.concatenate(subs, axis=2)


def s2d_ref_spelling_dataset(hr_patch, lr_patch_size):
    """Behavioural restatement of espcn/espcn/dataset.py:140-156 (and
    experiment_test.py:91-96) for one HR patch [P*r, P*r, C]."""
    subs = np.split(hr_patch, lr_patch_size, axis=1)
    subs = [s.reshape([lr_patch_size, 1, -1]) for s in subs]
    return np.concatenate(subs, axis=1)


# ----------------------------------------------------------------------------
# losses / optimizers / metrics
# ----------------------------------------------------------------------------
def mse_fwd_bwd(pred, target, numel_global=None):
    """tf.losses.mean_squared_error(reduction=MEAN); vdsr/vdsr/model_vdsr.py:120-123."""
    pred = np.asarray(pred, np.float64)
    target = np.asarray(target, np.float64)
    n = pred.size if numel_global is None else numel_global
    d = pred - target
    return float((d * d).sum() / n), 2.0 * d / n


def l2_loss(w):
    """tf.nn.l2_loss = sum(w^2)/2; model_vdsr.py:34 multiplies by 1e-4."""
    w = np.asarray(w, np.float64)
    return 0.5 * float((w * w).sum())


def adam_tf(w, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-8):
    """TF-1.x Adam (epsilon-hat).  t = 1-based step.  Returns (w, m, v)."""
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    w = w - lr_t * m / (np.sqrt(v) + eps)
    return w, m, v


def momentum_clip(w, g, acc, lr, mom=0.9, gradient_cap=0.01):
    """vdsr/vdsr/model_vdsr.py:158-184: clip to +-(cap/lr) then TF Momentum."""
    cap = gradient_cap / lr
    g = np.clip(g, -cap, cap)
    acc = mom * acc + g
    return w - lr * acc, acc


def psnr(a, b, max_val):
    """tf.image.psnr per image; vdsr/vdsr/experiment_train.py:80-82."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(axis=1)
    return 20.0 * np.log10(max_val) - 10.0 * np.log10(mse)


def ssim(a, b, max_val, filter_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim (TF 1.8 image_ops_impl._ssim_per_channel): gaussian window from _fspecial_gauss,
    VALID depthwise filtering, luminance * contrast-structure, mean over positions then channels.
    vdsr/vdsr/experiment_evaluate.py:57-60; espcn/espcn/experiment_test.py:55.  Parity unpinned (TF absent)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    coords = np.arange(filter_size) - (filter_size - 1) / 2.0
    g = np.exp(-coords ** 2 / (2.0 * sigma ** 2))
    g = g / g.sum()
    win = np.outer(g, g)

    def reducer(x):
        N, H, W, C = x.shape
        OH, OW = H - filter_size + 1, W - filter_size + 1
        out = np.zeros((N, OH, OW, C))
        for i in range(filter_size):
            for j in range(filter_size):
                out += win[i, j] * x[:, i:i + OH, j:j + OW, :]
        return out

    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mean0, mean1 = reducer(a), reducer(b)
    num0, den0 = mean0 * mean1 * 2.0, mean0 ** 2 + mean1 ** 2
    lum = (num0 + c1) / (den0 + c1)
    num1, den1 = reducer(a * b) * 2.0, reducer(a * a + b * b)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(axis=(1, 2)).mean(axis=-1)


def gaussian_blur(x, sigma):
    """skimage.filters.gaussian(image, sigma, mode='nearest') per image of [N,H,W,C]: scipy.ndimage's
    gaussian_filter on the two spatial axes, truncate 4.0, replicated borders.
    vdsr/vdsr/dataset.py:28; espcn/espcn/dataset.py:101.  Parity unpinned (skimage absent)."""
    from scipy.ndimage import gaussian_filter
    x = np.asarray(x, np.float64)
    if sigma <= 0:
        return x.copy()
    return gaussian_filter(x, sigma=(0, sigma, sigma, 0), mode='nearest', truncate=4.0)


def resize_bilinear(x, oh, ow):
    """skimage.transform.resize(image, [oh, ow], mode='edge', anti_aliasing=False) (order 1) per image of
    [N,H,W,C]: sample at (out + 0.5) * (in/out) - 0.5 with edge clamping.  vdsr/vdsr/dataset.py:31-35."""
    from scipy.ndimage import map_coordinates
    x = np.asarray(x, np.float64)
    N, H, W, C = x.shape
    ys = np.clip((np.arange(oh) + 0.5) * H / oh - 0.5, 0, H - 1)
    xs = np.clip((np.arange(ow) + 0.5) * W / ow - 0.5, 0, W - 1)
    yy, xx = np.meshgrid(ys, xs, indexing='ij')
    out = np.empty((N, oh, ow, C))
    for n in range(N):
        for c in range(C):
            out[n, :, :, c] = map_coordinates(x[n, :, :, c], [yy, xx], order=1, mode='nearest')
    return out


def resize_bicubic_tf(x, oh, ow, A=-0.75, half_pixel=False):
    """tf.image.resize_bicubic(x, [oh, ow]) as TensorFlow 1.x computes it (align_corners=False; resize_bicubic_op.cc as
    remembered, and PINNED by the reference's assets/srcnn_00{0,1}.jpg panels -- pin P6): in = out * IN / OUT (no
    half-pixel centres), offset = rint(frac * 1024), cubic-convolution weights with A = -0.75 at offset / 1024, taps
    lower-1..lower+2 clamped to the image.  srcnn/srcnn.py:89-93.  `A` / `half_pixel` exist only so that the pin test
    can show that the alternatives do NOT reproduce the reference's panels."""
    x = np.asarray(x, np.float64)
    K = 1024

    def taps(out_size, in_size):
        scale = in_size / out_size
        idx = np.zeros((out_size, 4), np.int64)
        w = np.zeros((out_size, 4))
        for o in range(out_size):
            loc = (o + 0.5) * scale - 0.5 if half_pixel else o * scale
            fl = int(np.floor(loc))
            off = int(np.rint((loc - fl) * K))
            xs = np.array([off / K + 1.0, off / K, (K - off) / K, (K - off) / K + 1.0])
            w[o, 1] = ((A + 2) * xs[1] - (A + 3)) * xs[1] ** 2 + 1
            w[o, 2] = ((A + 2) * xs[2] - (A + 3)) * xs[2] ** 2 + 1
            w[o, 0] = ((A * xs[0] - 5 * A) * xs[0] + 8 * A) * xs[0] - 4 * A
            w[o, 3] = ((A * xs[3] - 5 * A) * xs[3] + 8 * A) * xs[3] - 4 * A
            idx[o] = np.clip([fl - 1, fl, fl + 1, fl + 2], 0, in_size - 1)
        return idx, w

    N, H, W, C = x.shape
    iy, wy = taps(oh, H)
    ix, wx = taps(ow, W)
    rows = sum(x[:, :, ix[:, k], :] * wx[:, k][None, None, :, None] for k in range(4))      # along x first
    return sum(rows[:, iy[:, k]] * wy[:, k][None, :, None, None] for k in range(4))


def hd_to_sd(hd01, scaling_factor):
    """vdsr/vdsr/dataset.py:13-38 on a batch of float images in [0,1]."""
    N, H, W, C = hd01.shape
    sd_h, sd_w = int(H / scaling_factor), int(W / scaling_factor)
    bl = gaussian_blur(hd01, max(0.0, 0.5 * (scaling_factor - 1.0)))
    return resize_bilinear(resize_bilinear(bl, sd_h, sd_w), H, W)


def saturate_u8(x):
    """tf.saturate_cast(x*127.5+127.5, uint8): clamp then truncate.
    vdsr/vdsr/experiment_resolve.py:65-69."""
    v = np.asarray(x, np.float32) * np.float32(127.5) + np.float32(127.5)
    return np.clip(v, 0.0, 255.0).astype(np.uint8)


def lr_schedule(lr0, factor, step, decay_steps):
    """vdsr/vdsr/experiment_train.py:130; espcn/espcn/experiment_train.py:101-107."""
    return lr0 * (factor ** (step // decay_steps))


# ----------------------------------------------------------------------------
# whole networks
# ----------------------------------------------------------------------------
def vdsr_param_shapes(num_layers=20, channels=3, width=64):
    shapes = []
    cin = channels
    for i in range(num_layers - 1):
        shapes.append(((3, 3, cin, width), (width,)))
        cin = width
    shapes.append(((3, 3, cin, channels), (channels,)))
    return shapes


def xavier_uniform(rng, shape):
    """tf.contrib.layers.xavier_initializer(): U(+-sqrt(6/(fan_in+fan_out))),
    fan = kh*kw*C.  (TF's RNG stream itself is not reproducible: tests inject weights.)"""
    kh, kw, cin, cout = shape
    lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal_initializer: redraw beyond 2 sigma."""
    out = rng.normal(0.0, stddev, size=shape)
    bad = np.abs(out) > 2 * stddev
    while bad.any():
        out[bad] = rng.normal(0.0, stddev, size=int(bad.sum()))
        bad = np.abs(out) > 2 * stddev
    return out.astype(np.float32)


def vdsr_forward(sd, params, dtype=np.float64):
    """vdsr/vdsr/model_vdsr.py:47-106.  params = [(kernel, bias)] * num_layers.
    Returns dict with conv.i / relu.i (same post-ReLU tensor), conv.N, sr_images."""
    out = {'sd_images': np.asarray(sd, dtype)}
    t = out['sd_images']
    n = len(params)
    for i, (k, b) in enumerate(params[:-1]):
        t = conv2d_fwd(t, k, b, 'SAME', 'relu', dtype=dtype)
        out['conv.%d' % (i + 1)] = t
        out['relu.%d' % (i + 1)] = t          # second relu is idempotent (:74)
    k, b = params[-1]
    res = conv2d_fwd(t, k, b, 'SAME', None, dtype=dtype)
    out['conv.%d' % n] = res
    out['sr_images'] = out['sd_images'] + res
    return out


def vdsr_loss_and_grads(sd, hd, params, weight_decay=1e-4, numel_global=None, dtype=np.float64):
    """Forward + TF-autodiff-equivalent backward of model_vdsr.py:120-125.
    Returns (loss, [(dk, db)], fwd dict).  With numel_global set (data-parallel
    shard) the regulariser gradient is NOT added (caller adds it once after the
    reduce) -- see grads_add_l2."""
    fwd = vdsr_forward(sd, params, dtype)
    mse, d_sr = mse_fwd_bwd(fwd['sr_images'], hd, numel_global)
    reg = sum(weight_decay * l2_loss(k) for k, _ in params)
    n = len(params)
    grads = [None] * n
    acts = [fwd['sd_images']] + [fwd['conv.%d' % (i + 1)] for i in range(n - 1)]
    dpre = d_sr                                        # last layer has no activation
    for i in range(n - 1, -1, -1):
        k, _ = params[i]
        dk, db = conv2d_bwd_filter(acts[i], dpre, k.shape[:2], 'SAME', dtype)
        grads[i] = (dk, db)
        if i > 0:
            dx = conv2d_bwd_data(dpre, k, acts[i].shape[1:3], 'SAME', dtype)
            dpre = dx * act_grad_from_y(acts[i], 'relu')
    if numel_global is None:
        grads = [(dk + weight_decay * np.asarray(k, dtype), db) for (dk, db), (k, _) in zip(grads, params)]
    return mse + reg, grads, fwd


def enet_generator_forward(sd, bq, params, dtype=np.float64, keep=False):
    """enet/enet/model_enet.py:44-115 (+ residual_block :8-31).  params = [(kernel, bias)] * 25 in creation order:
    3x3 3->64 ReLU; 10 x [3x3 ReLU, 1x1, + block input, ReLU]; 2 x [nearest x2, 3x3 ReLU]; 3x3 ReLU; 3x3 -> 3; + bq.
    tf.image.resize_nearest_neighbor to an integer multiple (:78-80) replicates pixels.
    keep=True also returns the input tensor of every conv (all of them post-ReLU except the fed image)."""
    ins = [np.asarray(sd, dtype)]
    t = conv2d_fwd(ins[0], *params[0], 'SAME', 'relu', dtype=dtype)
    i = 1
    for _ in range(10):
        ins.append(t)
        x = conv2d_fwd(t, *params[i], 'SAME', 'relu', dtype=dtype)
        ins.append(x)
        t = conv2d_fwd(x, *params[i + 1], 'SAME', None, skip=t, post_relu=True, dtype=dtype)
        i += 2
    for _ in range(2):
        t = np.repeat(np.repeat(t, 2, axis=1), 2, axis=2)
        ins.append(t)
        t = conv2d_fwd(t, *params[i], 'SAME', 'relu', dtype=dtype)
        i += 1
    ins.append(t)
    t = conv2d_fwd(t, *params[i], 'SAME', 'relu', dtype=dtype)
    ins.append(t)
    sr = conv2d_fwd(t, *params[i + 1], 'SAME', None, skip=np.asarray(bq, dtype), dtype=dtype)
    return (sr, ins) if keep else sr


def enet_generator_backward(ins, d_sr, params, dtype=np.float64):
    """Gradients of the 25 (kernel, bias) pairs for a given d(loss)/d(sr_images): what TF's autodiff does for the
    generator inside `g_trainer.minimize(g_losses, var_list=g_vars)` (enet/enet/model_enet.py:331-337):
    ReluGrad = dy * [y > 0] on the post-ReLU tensor, ResizeNearestNeighborGrad = sum over each 2x2 block, the
    residual sum passes its gradient to both branches.  `ins` from enet_generator_forward(..., keep=True)."""
    grads = [None] * 25

    def relu_mask(g, y):
        return g * (y > 0)

    def block_sum(g):
        n, h, w, c = g.shape
        return g.reshape(n, h // 2, 2, w // 2, 2, c).sum(axis=(2, 4))

    g = np.asarray(d_sr, dtype)
    for i in (24, 23):
        k = params[i][0]
        grads[i] = conv2d_bwd_filter(ins[i], g, k.shape[:2], 'SAME', dtype=dtype)
        g = relu_mask(conv2d_bwd_data(g, k, ins[i].shape[1:3], 'SAME', dtype=dtype), ins[i])
    for i in (22, 21):
        k = params[i][0]
        grads[i] = conv2d_bwd_filter(ins[i], g, k.shape[:2], 'SAME', dtype=dtype)
        g = block_sum(relu_mask(conv2d_bwd_data(g, k, ins[i].shape[1:3], 'SAME', dtype=dtype), ins[i]))
    for b in range(9, -1, -1):
        i3, i1 = 1 + 2 * b, 2 + 2 * b
        grads[i1] = conv2d_bwd_filter(ins[i1], g, (1, 1), 'SAME', dtype=dtype)
        d3 = relu_mask(conv2d_bwd_data(g, params[i1][0], ins[i1].shape[1:3], 'SAME', dtype=dtype), ins[i1])
        grads[i3] = conv2d_bwd_filter(ins[i3], d3, (3, 3), 'SAME', dtype=dtype)
        g = relu_mask(g + conv2d_bwd_data(d3, params[i3][0], ins[i3].shape[1:3], 'SAME', dtype=dtype), ins[i3])
    grads[0] = conv2d_bwd_filter(ins[0], g, (3, 3), 'SAME', dtype=dtype)
    return grads


def espcn_forward(lr, params, dtype=np.float64):
    """espcn/espcn/model_espcn.py:30-62 / :117-134: tanh, tanh, linear; all SAME."""
    (k1, b1), (k2, b2), (k3, b3) = params
    t = conv2d_fwd(lr, k1, b1, 'SAME', 'tanh', dtype=dtype)
    t = conv2d_fwd(t, k2, b2, 'SAME', 'tanh', dtype=dtype)
    return conv2d_fwd(t, k3, b3, 'SAME', None, dtype=dtype)


def espcn_scaling_factor(f3_bias):
    """espcn/espcn/model_espcn.py:108."""
    return int((np.asarray(f3_bias).size // 3) ** 0.5)


def srcnn_forward(lo, params, dtype=np.float64):
    """srcnn/srcnn.py:100-130: relu, relu, tanh; all VALID."""
    (k1, b1), (k2, b2), (k3, b3) = params
    t = conv2d_fwd(lo, k1, b1, 'VALID', 'relu', dtype=dtype)
    t = conv2d_fwd(t, k2, b2, 'VALID', 'relu', dtype=dtype)
    return conv2d_fwd(t, k3, b3, 'VALID', 'tanh', dtype=dtype)


def srcnn_loss(sr, hi):
    """srcnn/srcnn.py:142-144: mean over rows of the L2 norm of reshape(diff,[-1,bb^2])."""
    sr = np.asarray(sr, np.float64)
    bb = sr.shape[1]
    d = (sr - np.asarray(hi, np.float64)).reshape(-1, bb * bb)
    return float(np.sqrt((d * d).sum(axis=1)).mean())


def srcnn_loss_and_grad(sr, hi):
    """Loss of srcnn.py:142-144 and d loss / d sr."""
    sr = np.asarray(sr, np.float64)
    bb = sr.shape[1]
    d = (sr - np.asarray(hi, np.float64)).reshape(-1, bb * bb)
    nr = np.sqrt((d * d).sum(axis=1, keepdims=True))
    return float(nr.mean()), (d / nr / d.shape[0]).reshape(sr.shape)


# ----------------------------------------------------------------------------
# C restatement (srx_oracle.c) via ctypes -- used for larger cases and as the
# cpu_baseline "port"
# ----------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_clib = None


def clib():
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, 'libsrx_oracle.so')
        if not os.path.exists(path):
            raise RuntimeError('oracle C library not built: run `make -C oracle` '
                               '(or __graft_entry__.build())')
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        i = ctypes.c_int
        L.srx_ref_conv2d_fwd.argtypes = [fp, fp, fp, fp, fp] + [i] * 13
        L.srx_ref_conv2d_fwd.restype = None
        L.srx_ref_act_bwd.argtypes = [fp, fp, fp, ctypes.c_size_t, i]
        L.srx_ref_conv2d_bwd_data.argtypes = [fp, fp, fp] + [i] * 11
        L.srx_ref_conv2d_bwd_filter.argtypes = [fp, fp, fp, fp] + [i] * 11
        L.srx_ref_depth_to_space.argtypes = [fp, fp] + [i] * 5
        L.srx_ref_space_to_depth.argtypes = [fp, fp] + [i] * 5
        L.srx_ref_mse_fwd_bwd.argtypes = [fp, fp, fp, ctypes.c_size_t, ctypes.c_double]
        L.srx_ref_mse_fwd_bwd.restype = ctypes.c_double
        L.srx_ref_l2_loss.argtypes = [fp, ctypes.c_size_t]
        L.srx_ref_l2_loss.restype = ctypes.c_double
        L.srx_ref_adam_tf.argtypes = [fp, fp, fp, fp, ctypes.c_size_t] + [ctypes.c_float] * 4 + [ctypes.c_long]
        L.srx_ref_momentum_clip.argtypes = [fp, fp, fp, ctypes.c_size_t] + [ctypes.c_float] * 3
        L.srx_ref_psnr.argtypes = [fp, fp, fp, i, ctypes.c_size_t, ctypes.c_float]
        L.srx_ref_saturate_u8.argtypes = [fp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t]
        L.srx_ref_num_threads.restype = i
        L.srx_ref_set_num_threads.argtypes = [i]
        L.srx_ref_set_num_threads.restype = i
        _clib = L
    return _clib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def c_conv2d_fwd(x, w, b=None, padding='SAME', act=None, skip=None, post_relu=False):
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    skip = None if skip is None else _f32(skip)
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = w.shape
    pad_t, pad_l, OH, OW = conv_geometry(H, W, KH, KW, padding)
    y = np.empty((N, OH, OW, Cout), np.float32)
    clib().srx_ref_conv2d_fwd(_p(x), _p(w), _p(b), _p(skip), _p(y), N, H, W, Cin, Cout, KH, KW,
                              pad_t, pad_l, OH, OW, _ACT_BY_NAME.get(act, act), int(post_relu))
    return y


def c_act_bwd(dy, y, act):
    dy, y = _f32(dy), _f32(y)
    out = np.empty_like(dy)
    clib().srx_ref_act_bwd(_p(dy), _p(y), _p(out), dy.size, _ACT_BY_NAME.get(act, act))
    return out


def c_conv2d_bwd_data(dpre, w, in_hw, padding='SAME'):
    dpre, w = _f32(dpre), _f32(w)
    H, W = in_hw
    KH, KW, Cin, Cout = w.shape
    pad_t, pad_l, OH, OW = conv_geometry(H, W, KH, KW, padding)
    N = dpre.shape[0]
    dx = np.empty((N, H, W, Cin), np.float32)
    clib().srx_ref_conv2d_bwd_data(_p(dpre), _p(w), _p(dx), N, H, W, Cin, Cout, KH, KW,
                                   pad_t, pad_l, OH, OW)
    return dx


def c_conv2d_bwd_filter(x, dpre, ksize, padding='SAME'):
    x, dpre = _f32(x), _f32(dpre)
    KH, KW = ksize
    N, H, W, Cin = x.shape
    Cout = dpre.shape[-1]
    pad_t, pad_l, OH, OW = conv_geometry(H, W, KH, KW, padding)
    dw = np.empty((KH, KW, Cin, Cout), np.float32)
    db = np.empty((Cout,), np.float32)
    clib().srx_ref_conv2d_bwd_filter(_p(x), _p(dpre), _p(dw), _p(db), N, H, W, Cin, Cout, KH, KW,
                                     pad_t, pad_l, OH, OW)
    return dw, db


def c_depth_to_space(x, r):
    x = _f32(x)
    N, H, W, D = x.shape
    C = D // (r * r)
    out = np.empty((N, H * r, W * r, C), np.float32)
    clib().srx_ref_depth_to_space(_p(x), _p(out), N, H, W, C, r)
    return out


def c_space_to_depth(x, r):
    x = _f32(x)
    N, HR, WR, C = x.shape
    H, W = HR // r, WR // r
    out = np.empty((N, H, W, C * r * r), np.float32)
    clib().srx_ref_space_to_depth(_p(x), _p(out), N, H, W, C, r)
    return out


def c_adam_tf(w, g, m, v, lr, t, b1=0.9, b2=0.999, eps=1e-8):
    w, g, m, v = _f32(w).copy(), _f32(g), _f32(m).copy(), _f32(v).copy()
    clib().srx_ref_adam_tf(_p(w), _p(g), _p(m), _p(v), w.size, lr, b1, b2, eps, int(t))
    return w, m, v


def c_vdsr_forward(sd, params):
    """C-oracle VDSR forward (fp32), used for the cpu_baseline timing."""
    t = _f32(sd)
    for k, b in params[:-1]:
        t = c_conv2d_fwd(t, k, b, 'SAME', 'relu')
    k, b = params[-1]
    return c_conv2d_fwd(t, k, b, 'SAME', None, skip=_f32(sd))


def c_vdsr_train_step_grads(sd, hd, params, weight_decay=1e-4):
    """C-oracle forward + backward (fp32); returns (loss, grads)."""
    acts = [_f32(sd)]
    for k, b in params[:-1]:
        acts.append(c_conv2d_fwd(acts[-1], k, b, 'SAME', 'relu'))
    k, b = params[-1]
    sr = c_conv2d_fwd(acts[-1], k, b, 'SAME', None, skip=acts[0])
    hd = _f32(hd)
    dpre = np.empty_like(sr)
    mse = clib().srx_ref_mse_fwd_bwd(_p(sr), _p(hd), _p(dpre), sr.size, float(sr.size))
    reg = sum(weight_decay * clib().srx_ref_l2_loss(_p(_f32(k)), k.size) for k, _ in params)
    grads = [None] * len(params)
    for i in range(len(params) - 1, -1, -1):
        k, _ = params[i]
        dk, db = c_conv2d_bwd_filter(acts[i], dpre, k.shape[:2], 'SAME')
        grads[i] = (dk + np.float32(weight_decay) * _f32(k), db)
        if i > 0:
            dx = c_conv2d_bwd_data(dpre, k, acts[i].shape[1:3], 'SAME')
            dpre = c_act_bwd(dx, acts[i], 'relu')
    return mse + reg, grads


# ---------------------------------------------------------------------------------------------------------------------
# scipy.misc.imresize on uint8 images = Pillow's Image.resize (enet/enet/datasets.py:110-111, enet/enet/
# experiment_resolve.py:78-79).  Pillow is a dependency of the reference (through scipy.misc; version not pinned by
# the reference, the algorithm below is libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
# ImagingResampleHorizontal_8bpc / Vertical_8bpc).  Pinned twice: against Pillow itself where it is installed
# (tests/test_oracle_pins.py) and against the reference's own output assets/enet_eagle_bq.png (P5).
# ---------------------------------------------------------------------------------------------------------------------
def _pil_filter(name):
    def bilinear(x):
        x = abs(x)
        return 1.0 - x if x < 1.0 else 0.0

    def bicubic(x, a=-0.5):
        x = abs(x)
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0
    return {'bilinear': (bilinear, 1.0), 'bicubic': (bicubic, 2.0)}[name]


def pil_resample_coeffs(in_size, out_size, filt):
    """(bounds int64 [out, 2] = (first input index, count), kk int64 [out, ksize]): Resample.c precompute_coeffs +
    normalize_coeffs_8bpc (22 fractional bits)."""
    f, sup = _pil_filter(filt)
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = sup * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([f((x + xmin - center + 0.5) * ss) for x in range(xmax)], np.float64)
        ww = 0.0
        for v in w:                     # (the C loop adds them one by one)
            ww += v
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << 22)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pil_resample_axis(img, out_size, axis, filt):
    x = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, kk = pil_resample_coeffs(x.shape[0], out_size, filt)
    out = np.empty((out_size,) + x.shape[1:], np.int64)
    for o in range(out_size):
        xmin, n = bounds[o]
        out[o] = np.clip(((1 << 21) + np.tensordot(kk[o, :n], x[xmin:xmin + n], axes=(0, 0))) >> 22, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def pil_resize_u8(img, out_h, out_w, filt='bicubic'):
    """PIL.Image.fromarray(img).resize((out_w, out_h), filt) for uint8 [H,W,C] or [N,H,W,C]: horizontal pass, then
    vertical, the intermediate in uint8."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    h_axis = img.ndim - 3
    t = img
    if out_w != img.shape[h_axis + 1]:
        t = _pil_resample_axis(t, out_w, h_axis + 1, filt)
    if out_h != img.shape[h_axis]:
        t = _pil_resample_axis(t, out_h, h_axis, filt)
    return t


def u8_to_pm1(x):
    """astype(float32) / 127.5 - 1.0 (enet/enet/datasets.py:113-115)."""
    return np.asarray(x).astype(np.float32) / np.float32(127.5) - np.float32(1.0)
