"""
oracle_enet.py -- CPU restatement (NumPy float64) of EnhanceNet-PAT's loss side: TEST INFRASTRUCTURE, never imported
by the product (ml_super_resolution_amd/).  Each function cites the reference lines it restates
(paths relative to /root/reference).

PARITY UNPINNED: the arithmetic lives in TensorFlow 1.8 (absent here), the reference holds no tests or vectors for
these functions, and the VGG-19 weights it downloads are not available offline.  tests/test_oracle_enet_pat.py
cross-checks every function below against an independent implementation (torch CPU float64 + autograd).

  discriminator        enet/enet/model_enet.py:118-162
  log losses           enet/enet/model_enet.py:165-182
  normalize            enet/enet/model_enet.py:34-41
  perceptual loss      enet/enet/model_enet.py:185-206
  texture loss         enet/enet/model_enet.py:209-261
  build_enet wiring    enet/enet/model_enet.py:264-350  (loss weights, which variables each trainer updates)
  VGG-19               enet/enet/model_vgg.py:11-36,65-99
"""
import numpy as np

from . import oracle as O

VGG_LAYERS = ['block1_conv1', 'block1_conv2', 'block1_pool',
              'block2_conv1', 'block2_conv2', 'block2_pool',
              'block3_conv1', 'block3_conv2', 'block3_conv3', 'block3_conv4', 'block3_pool',
              'block4_conv1', 'block4_conv2', 'block4_conv3', 'block4_conv4', 'block4_pool',
              'block5_conv1', 'block5_conv2', 'block5_conv3', 'block5_conv4', 'block5_pool']   # model_vgg.py:79-88
VGG_MEAN_BGR = np.array([103.939, 116.779, 123.68])                                            # model_vgg.py:76
TEXTURE_LAYERS = [('block1_conv1', 3e-7), ('block2_conv1', 1e-6), ('block3_conv1', 1e-6)]      # model_enet.py:214-218
LOG_EPS = 1e-7                                                                                 # tf.losses.log_loss default


def vgg19_channels(width=64):
    """{layer: (cin, cout)} of the 16 convolutions; width = block1's channel count (64 in VGG-19)."""
    out, cin = {}, 3
    for name in VGG_LAYERS:
        if name.endswith('pool'):
            continue
        block = int(name[5])
        cout = width * min(2 ** (block - 1), 8)
        out[name] = (cin, cout)
        cin = cout
    return out


# ---- strided SAME convolution (TF: out = ceil(in / s), pad_total = max((out-1)*s + k - in, 0), pad_before = total // 2)
def _same_geometry(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def conv2d_same_fwd(x, w, b, stride=1):
    """tf.layers.conv2d(kernel_size=k, strides=s, padding='same') without the activation (model_enet.py:126-146)."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
    N, H, W, _ = x.shape
    KH, KW, _, Cout = w.shape
    OH, pt, pb = _same_geometry(H, KH, stride)
    OW, pl, pr = _same_geometry(W, KW, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((N, OH, OW, Cout))
    for kh in range(KH):
        for kw in range(KW):
            y += xp[:, kh:kh + (OH - 1) * stride + 1:stride, kw:kw + (OW - 1) * stride + 1:stride, :] @ w[kh, kw]
    return y + np.asarray(b, np.float64)


def conv2d_same_bwd(x, w, dpre, stride=1, want_dx=True):
    """Gradients of conv2d_same_fwd: (dx, dw, db)."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64); dpre = np.asarray(dpre, np.float64)
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = w.shape
    OH, pt, pb = _same_geometry(H, KH, stride)
    OW, pl, pr = _same_geometry(W, KW, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    d2 = dpre.reshape(-1, Cout)
    for kh in range(KH):
        for kw in range(KW):
            sl = (slice(None), slice(kh, kh + (OH - 1) * stride + 1, stride), slice(kw, kw + (OW - 1) * stride + 1, stride))
            dw[kh, kw] = xp[sl].reshape(-1, Cin).T @ d2
            if want_dx:
                dxp[sl] += dpre @ w[kh, kw].T
    dx = dxp[:, pt:pt + H, pl:pl + W, :] if want_dx else None
    return dx, dw, d2.sum(axis=0)


# ---- max pooling 2x2 / 2 SAME (model_vgg.py:28-36) -----------------------------------------------------------------
def maxpool2x2_fwd(x):
    x = np.asarray(x, np.float64)
    N, H, W, C = x.shape
    OH, OW = -(-H // 2), -(-W // 2)
    xp = np.full((N, 2 * OH, 2 * OW, C), -np.inf)
    xp[:, :H, :W] = x
    return xp.reshape(N, OH, 2, OW, 2, C).max(axis=(2, 4))


def maxpool2x2_bwd(x, dout):
    """MaxPoolGrad: the window's gradient goes to its first maximum in scan order (dy, dx)."""
    x = np.asarray(x, np.float64)
    N, H, W, C = x.shape
    OH, OW = -(-H // 2), -(-W // 2)
    xp = np.full((N, 2 * OH, 2 * OW, C), -np.inf)
    xp[:, :H, :W] = x
    win = xp.reshape(N, OH, 2, OW, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(N, OH, OW, C, 4)
    arg = win.argmax(axis=-1)                                     # first maximum
    onehot = (np.arange(4) == arg[..., None]).astype(np.float64) * np.asarray(dout, np.float64)[..., None]
    dxp = onehot.reshape(N, OH, OW, C, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(N, 2 * OH, 2 * OW, C)
    return dxp[:, :H, :W]


# ---- VGG-19 (model_vgg.py:11-25, 65-99) ---------------------------------------------------------------------------
def vgg_preprocess(x_pm1):
    """model_enet.py:288-289 (x * 127.5 + 127.5) then model_vgg.py:72-76 (RGB -> BGR, subtract the mean colour)."""
    v = np.asarray(x_pm1, np.float64) * 127.5 + 127.5
    return v[..., ::-1] - VGG_MEAN_BGR


def vgg19_forward(x_pm1, weights):
    """weights: {layer_name: (kernel HWIO, bias)}.  Returns {layer_name: tensor} for all 21 layers (+ 'input')."""
    feats = {'input': vgg_preprocess(x_pm1)}
    t = feats['input']
    for name in VGG_LAYERS:
        if name.endswith('pool'):
            t = maxpool2x2_fwd(t)
        else:
            k, b = weights[name]
            t = np.maximum(conv2d_same_fwd(t, k, b), 0.0)         # conv + bias_add + relu (model_vgg.py:21-23)
        feats[name] = t
    return feats


def vgg19_backward(feats, weights, dtaps):
    """d(loss)/d(x_pm1) given d(loss)/d(feature) for some layers (dtaps: {layer_name: array}); the weights are
    constants (model_vgg.py:55-56: tf.constant)."""
    g = None
    for idx in range(len(VGG_LAYERS) - 1, -1, -1):
        name = VGG_LAYERS[idx]
        if name in dtaps:
            g = dtaps[name] if g is None else g + dtaps[name]
        if g is None:
            continue
        below = feats[VGG_LAYERS[idx - 1]] if idx > 0 else feats['input']
        if name.endswith('pool'):
            g = maxpool2x2_bwd(below, g)
        else:
            dpre = g * (feats[name] > 0)
            g, _, _ = conv2d_same_bwd(below, weights[name][0], dpre)
    return 127.5 * g[..., ::-1]                                    # through the reverse and the scale of vgg_preprocess


# ---- discriminator (model_enet.py:118-162) ---------------------------------------------------------------------------
def lrelu(v):
    return np.where(v > 0, v, 0.2 * v)


def discriminator_forward(x, convs, dense, keep=False):
    """convs: 10 x (kernel, bias), 3x3: (stride 1, stride 2) x 5 with 2**(i+5) filters, leaky ReLU (0.2);
    dense: [(W [F,1024], b), (W [1024,1], b)], leaky ReLU then sigmoid.  x [N,S,S,3] with S/32 * S/32 * C5 == F."""
    acts = [np.asarray(x, np.float64)]
    t = acts[0]
    for i, (k, b) in enumerate(convs):
        t = lrelu(conv2d_same_fwd(t, k, b, stride=1 + (i & 1)))
        acts.append(t)
    flat = t.reshape(t.shape[0], -1)                               # tf.layers.flatten (NHWC order)
    h = lrelu(flat @ np.asarray(dense[0][0], np.float64) + dense[0][1])
    z = h @ np.asarray(dense[1][0], np.float64) + dense[1][1]
    p = 1.0 / (1.0 + np.exp(-z))
    return (p, (acts, flat, h)) if keep else p


def discriminator_backward(saved, p, dp, convs, dense, want_dx=True):
    """(dx, conv grads [(dk, db)], dense grads [(dW, db)]) for d(loss)/d(p) = dp."""
    acts, flat, h = saved
    dz = dp * p * (1.0 - p)
    W2 = np.asarray(dense[1][0], np.float64); W1 = np.asarray(dense[0][0], np.float64)
    g_dense2 = (h.T @ dz, dz.sum(axis=0))
    dh = (dz @ W2.T) * np.where(h > 0, 1.0, 0.2)
    g_dense1 = (flat.T @ dh, dh.sum(axis=0))
    g = (dh @ W1.T).reshape(acts[-1].shape)
    cgrads = [None] * len(convs)
    for i in range(len(convs) - 1, -1, -1):
        dpre = g * np.where(acts[i + 1] > 0, 1.0, 0.2)
        g, dk, db = conv2d_same_bwd(acts[i], convs[i][0], dpre, stride=1 + (i & 1), want_dx=(want_dx or i > 0))
        cgrads[i] = (dk, db)
    return g, cgrads, [g_dense1, g_dense2]


# ---- losses -----------------------------------------------------------------------------------------------------------
def log_loss(label, p):
    """tf.losses.log_loss(labels, predictions, epsilon=1e-7, reduction=MEAN) and its gradient (model_enet.py:165-182)."""
    p = np.asarray(p, np.float64)
    loss = np.mean(-label * np.log(p + LOG_EPS) - (1.0 - label) * np.log(1.0 - p + LOG_EPS))
    dp = (-label / (p + LOG_EPS) + (1.0 - label) / (1.0 - p + LOG_EPS)) / p.size
    return loss, dp


def normalize(t):
    """model_enet.py:34-41."""
    t = np.asarray(t, np.float64)
    return t / (t.mean(axis=-1, keepdims=True) + 1e-6)


def normalize_bwd(t, dy):
    t = np.asarray(t, np.float64)
    C = t.shape[-1]
    m = t.mean(axis=-1, keepdims=True) + 1e-6
    return dy / m - (dy * t).sum(axis=-1, keepdims=True) / (C * m * m)


def mse(a, b):
    d = a - b
    return np.mean(d * d), 2.0 * d / d.size


def perceptual_loss(sr_feats, hd_feats):
    """model_enet.py:185-206.  Returns (loss, {layer: d loss / d sr feature})."""
    loss, dt = 0.0, {}
    for name, wgt in (('block2_pool', 0.2), ('block5_pool', 0.02)):
        l, d = mse(normalize(sr_feats[name]), normalize(hd_feats[name]))
        loss += wgt * l
        dt[name] = normalize_bwd(sr_feats[name], wgt * d)
    return loss, dt


def patches16(t):
    """tf.extract_image_patches(16x16 / 16, VALID) + reshape [-1, h*w//256, 256, c] (model_enet.py:237-250)."""
    N, H, W, C = t.shape
    return t.reshape(N, H // 16, 16, W // 16, 16, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, (H // 16) * (W // 16), 256, C)


def patches16_bwd(dp, shape):
    N, H, W, C = shape
    return dp.reshape(N, H // 16, W // 16, 16, 16, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, H, W, C)


def texture_loss(sr_feats, hd_feats):
    """model_enet.py:209-261: gram matrices x^T x of every 16x16 patch of the normalised features, MSE, weighted sum."""
    loss, dt = 0.0, {}
    for name, wgt in TEXTURE_LAYERS:
        s, h = sr_feats[name], hd_feats[name]
        sp, hp = patches16(normalize(s)), patches16(normalize(h))
        gs = np.einsum('npki,npkj->npij', sp, sp)
        gh = np.einsum('npki,npkj->npij', hp, hp)
        l, dg = mse(gs, gh)
        loss += wgt * l
        dsp = np.einsum('npki,npij->npkj', sp, wgt * (dg + dg.transpose(0, 1, 3, 2)))
        dt[name] = normalize_bwd(s, patches16_bwd(dsp, s.shape))
    return loss, dt


def enet_losses_and_sr_gradient(sr, hd, vgg_weights, d_convs, d_dense, pat_model='pat', at_sr_feats=None, at_fake=None):
    """The generator's objective and its gradient with respect to sr_images (build_enet, model_enet.py:286-326):
    g_losses = p_loss [+ g_loss * (2.0 if 't' in model else 1.0)] [+ t_loss].  Returns (dict of losses, d_sr).
    at_sr_feats / at_fake: differentiate AT these activations (VGG-19 features of sr; (fake, saved) of the
    discriminator) instead of the ones recomputed here.  The gradient is discontinuous where a ReLU input crosses zero
    or two pooling candidates tie: a float32 forward pass lands on the other side of such a point now and then, and a
    comparison must hold both gradients to the same side."""
    sr_f = at_sr_feats if at_sr_feats is not None else vgg19_forward(sr, vgg_weights)
    hd_f = vgg19_forward(hd, vgg_weights)
    losses = {}
    p_loss, dt = perceptual_loss(sr_f, hd_f)
    losses['p_loss'] = p_loss
    total = p_loss
    d_sr = 0.0
    if 'a' in pat_model:
        fake, saved = at_fake if at_fake is not None else discriminator_forward(sr, d_convs, d_dense, keep=True)
        real = discriminator_forward(hd, d_convs, d_dense)
        lf, _ = log_loss(0.0, fake)
        lr_, _ = log_loss(1.0, real)
        losses['a_loss'] = lf + lr_                                  # model_enet.py:172-182
        g_loss, dp = log_loss(1.0, fake)                             # model_enet.py:165-169
        losses['g_loss'] = g_loss
        gw = 2.0 if 't' in pat_model else 1.0                        # model_enet.py:310-313
        total = total + gw * g_loss
        dx, _, _ = discriminator_backward(saved, fake, gw * dp, d_convs, d_dense)
        d_sr = d_sr + dx
    if 't' in pat_model:
        t_loss, dtt = texture_loss(sr_f, hd_f)
        losses['t_loss'] = t_loss
        total = total + t_loss
        for k, v in dtt.items():
            dt[k] = dt[k] + v if k in dt else v
    losses['g_loss_all'] = total
    d_sr = d_sr + vgg19_backward(sr_f, vgg_weights, dt)
    return losses, d_sr


def discriminator_loss_and_grads(sr, hd, d_convs, d_dense, at_fake=None, at_real=None):
    """a_loss = log_loss(0, D(sr)) + log_loss(1, D(hd)) and its gradient for the d_ variables (d_trainer,
    model_enet.py:339-343; sr is the generator's output, a constant for this trainer's var_list).
    at_fake / at_real: (p, saved) to differentiate at (see enet_losses_and_sr_gradient)."""
    fake, sf = at_fake if at_fake is not None else discriminator_forward(sr, d_convs, d_dense, keep=True)
    real, sr_ = at_real if at_real is not None else discriminator_forward(hd, d_convs, d_dense, keep=True)
    lf, dpf = log_loss(0.0, fake)
    lr_, dpr = log_loss(1.0, real)
    _, cf, df = discriminator_backward(sf, fake, dpf, d_convs, d_dense, want_dx=False)
    _, cr, dr = discriminator_backward(sr_, real, dpr, d_convs, d_dense, want_dx=False)
    convs = [(a[0] + b[0], a[1] + b[1]) for a, b in zip(cf, cr)]
    dense = [(a[0] + b[0], a[1] + b[1]) for a, b in zip(df, dr)]
    return lf + lr_, convs, dense
