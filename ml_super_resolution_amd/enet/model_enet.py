"""
model_enet.py -- mirror of enet/enet/model_enet.py (reference) on the MI355X engine: the EnhanceNet generator, the
discriminator, the perceptual / texture / adversarial losses on VGG-19 features and the two trainers of `build_enet`.

Generator (model_enet.py:44-115), variables under `g_`:
  3x3 conv 3->64 ReLU                                   (:63-70)
  10 x residual_block: 3x3 ReLU -> 1x1 -> relu(x + .)   (:8-31, :74-75)
  2 x [nearest-neighbour upsample x2 -> 3x3 conv ReLU]  (:78-89)
  3x3 conv ReLU, 3x3 conv -> 3, + bq_images             (:92-113)

Discriminator (:118-162, row A14), variables under `d_`: 5 x [3x3 s1, 3x3 s2] with 32..512 filters, leaky ReLU 0.2,
flatten, dense 1024 leaky ReLU, dense 1 sigmoid.  `build_enet` calls build_discriminator twice (real, fake) inside
`tf.variable_scope('d_', reuse=tf.AUTO_REUSE)`; the scope is closed in between, which resets TensorFlow's counters
for the unnamed tf.layers scopes, so both calls resolve to d_/conv2d .. d_/conv2d_9, d_/dense, d_/dense_1: ONE set
of weights (as the GAN needs).

Losses (:165-261, row N4): log-loss GAN terms, perceptual MSE on normalised block2_pool / block5_pool (weights 0.2 /
0.02), texture MSE between gram matrices of 16x16 patches of normalised block1/2/3_conv1 (3e-7 / 1e-6 / 1e-6).
`build_enet` (:264-350): g_losses = p_loss [+ g_loss * (2 if 't' else 1)] [+ t_loss]; g_trainer = Adam(1e-4) on the
`g_` variables (increments global_step), d_trainer = Adam(1e-4) on the `d_` variables minimising a_loss.

`build_enet(sd_images, bq_images, hd_images, pat_model, vgg19_path)` keeps the reference's keys:
  {sd_images, bq_images, sr_images} (+ hd_images, a_loss, g_loss, t_loss, p_loss, g_loss_all, g_trainer, d_trainer, step).
"""
import os

import numpy as np
import torch

from .. import graph, ops
from ..blocked import BlockedConv, DenseLayer, ParamPool, to_blocks, to_nhwc
from ..engine import truncated_normal_
from . import model_vgg


def generator_layers():
    """[(kernel size, cin, cout)] in creation order = tf.layers naming order."""
    layers = [(3, 3, 64)]
    for _ in range(10):
        layers += [(3, 64, 64), (1, 64, 64)]
    layers += [(3, 64, 64), (3, 64, 64), (3, 64, 64), (3, 64, 3)]
    return layers


class EnetGenerator(object):
    def __init__(self, device='cuda', seed=None):
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        # all variables in ONE flat buffer (16-byte aligned slices), gradients in a second one of the same layout:
        # the Adam update of the 50 tensors is a single launch
        shapes, off = [], 0
        for k, cin, cout in generator_layers():
            for shape in ((k, k, cin, cout), (cout,)):
                n = 1
                for d in shape:
                    n *= d
                shapes.append((off, n, shape))
                off += (n + 3) // 4 * 4
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        view = lambda buf, j: buf[shapes[j][0]:shapes[j][0] + shapes[j][1]].view(shapes[j][2])
        self.kernels = [view(self.params, 2 * i) for i in range(len(shapes) // 2)]
        self.biases = [view(self.params, 2 * i + 1) for i in range(len(shapes) // 2)]
        self._gk = [view(self.grads, 2 * i) for i in range(len(shapes) // 2)]
        self._gb = [view(self.grads, 2 * i + 1) for i in range(len(shapes) // 2)]
        for w in self.kernels:
            truncated_normal_(w, 0.02, gen)                               # model_enet.py:10,46 (biases: zeros)
        self.placeholders = {}
        self._saved = None

    def variables(self):
        out = {}
        for i, (w, b) in enumerate(zip(self.kernels, self.biases)):
            scope = 'g_/conv2d' if i == 0 else 'g_/conv2d_%d' % i
            out[scope + '/kernel'] = w
            out[scope + '/bias'] = b
        return out

    def set_params(self, pairs):
        for i, (k, b) in enumerate(pairs):
            self.kernels[i].copy_(torch.as_tensor(k, dtype=torch.float32).to(self.device))
            self.biases[i].copy_(torch.as_tensor(b, dtype=torch.float32).to(self.device))

    def forward(self, sd_images, bq_images, keep=False):
        """sd_images [N,h,w,3], bq_images [N,4h,4w,3] (bicubic-upscaled) -> sr_images [N,4h,4w,3].
        keep=True saves what `backward` needs (the input of every conv; all of them post-ReLU tensors)."""
        K, B = self.kernels, self.biases
        ins = [sd_images]                        # ins[i] = input of conv i
        t = ops.conv2d_fwd(sd_images, K[0], B[0], 'same', 'relu')
        i = 1
        for _ in range(10):
            ins.append(t)
            x = ops.conv2d_fwd(t, K[i], B[i], 'same', 'relu')
            ins.append(x)
            # 1x1 conv, + block input, ReLU -- one launch
            t = ops.conv2d_fwd(x, K[i + 1], B[i + 1], 'same', None, skip=t, post_add_relu=True)
            i += 2
        for _ in range(2):
            t = ops.upsample_nearest(t, 2)       # resize_nearest_neighbor to 2h then 4h = two doublings
            ins.append(t)
            t = ops.conv2d_fwd(t, K[i], B[i], 'same', 'relu')
            i += 1
        ins.append(t)
        t = ops.conv2d_fwd(t, K[i], B[i], 'same', 'relu')
        ins.append(t)
        self._saved = ins if keep else None
        return ops.conv2d_fwd(t, K[i + 1], B[i + 1], 'same', None, skip=bq_images)

    def backward(self, d_sr):
        """Gradients of every generator variable given d(loss)/d(sr_images) -- the part of
        `g_trainer = AdamOptimizer(1e-4).minimize(g_losses, var_list=g_vars)` (model_enet.py:331-337) that runs
        through the generator, whatever the loss on sr_images is.  Returns [(dkernel, dbias)] in layer order.

        Every conv input is a post-ReLU tensor, so each data gradient is produced with the ReluGrad of the layer
        below already applied (fused in the dgrad kernel's epilogue):
          last conv / 4x convs:  plain chain, as in VDSR;
          upsampling:            the mask is constant over a 2x2 block, so it is applied at the high resolution
                                 (on the saved upsampled tensor) and the 2x2 block sum follows;
          residual block t' = relu(t + conv1x1(relu(conv3x3(t)))): the incoming gradient g (already masked by
                                 t' > 0) goes to the 1x1 conv and, unchanged, to the skip path; the block input's
                                 gradient is (t > 0) ? g + dgrad3x3 : 0."""
        if self._saved is None:
            raise RuntimeError('backward() needs a forward(..., keep=True) first')
        K, ins = self.kernels, self._saved
        grads = [None] * len(K)

        def wgrad(i, dpre):
            grads[i] = ops.conv2d_bwd_filter(ins[i], dpre, K[i].shape, 'same', dw=self._gk[i], dbias=self._gb[i])

        i = len(K) - 1                           # 24: 64 -> 3, no activation
        g = d_sr.contiguous()
        wgrad(i, g)
        g = ops.conv2d_bwd_data(g, K[i], ins[i].shape, 'same', x_in=ins[i], in_act='relu')
        i -= 1                                   # 23: 3x3 ReLU at 4x
        wgrad(i, g)
        g = ops.conv2d_bwd_data(g, K[i], ins[i].shape, 'same', x_in=ins[i], in_act='relu')
        for _ in range(2):                       # 22, 21: upsample -> 3x3 ReLU
            i -= 1
            wgrad(i, g)
            g = ops.conv2d_bwd_data(g, K[i], ins[i].shape, 'same', x_in=ins[i], in_act='relu')
            g = ops.upsample_nearest_bwd(g, 2)
        for _ in range(10):                      # residual blocks, last to first
            i -= 2                               # i = the block's 3x3 conv, i + 1 its 1x1 conv
            wgrad(i + 1, g)
            d3 = ops.conv2d_bwd_data(g, K[i + 1], ins[i + 1].shape, 'same', x_in=ins[i + 1], in_act='relu')
            wgrad(i, d3)
            via_conv = ops.conv2d_bwd_data(d3, K[i], ins[i].shape, 'same')
            g = ops.add_relu_grad(g, via_conv, ins[i], out=via_conv)
        wgrad(0, g)                              # first conv: its input is the fed image, no data gradient
        return grads

    def adam_step(self, grads, state, lr=1e-4):
        """tf.train.AdamOptimizer(learning_rate=0.0001) on the g_ variables (model_enet.py:336-337); `state` is a
        dict this method fills on first use (slots m, v and the step count).  `grads` as returned by backward()
        (views of self.grads: nothing to copy) or any list of (dkernel, dbias)."""
        if not state:
            state['t'] = 0
            state['m'] = torch.zeros_like(self.params)
            state['v'] = torch.zeros_like(self.params)
        for i, (dw, db) in enumerate(grads):
            if dw.data_ptr() != self._gk[i].data_ptr():
                self._gk[i].copy_(dw)
            if db.data_ptr() != self._gb[i].data_ptr():
                self._gb[i].copy_(db)
        state['t'] += 1
        ops.adam_tf_step(self.params, self.grads, state['m'], state['v'], lr, state['t'])

    def run(self, keys, feed_dict):
        feeds = {name: feed_dict[ph] for name, ph in self.placeholders.items() if ph in feed_dict}
        sd = graph.to_device(feeds['sd_images'], self.device)
        bq = graph.to_device(feeds['bq_images'], self.device)
        sr = self.forward(sd, bq)
        vals = {'sr_images': sr, 'sd_images': sd, 'bq_images': bq}
        return {k: vals[k].detach().cpu().numpy() for k in keys}


# ---------------------------------------------------------------------------------------------------------------------
# discriminator (model_enet.py:118-162)
# ---------------------------------------------------------------------------------------------------------------------
def discriminator_layers(width=32):
    """[(cin, cout, stride)] of the ten 3x3 convolutions: filters = 2 ** (i + 5) for i in 0..4 with width 32."""
    layers, cin = [], 3
    for i in range(5):
        f = width * 2 ** i
        layers += [(cin, f, 1), (f, f, 2)]
        cin = f
    return layers


class Discriminator(object):
    def __init__(self, device='cuda', seed=None, width=32, image_size=128, dense_units=1024):
        """width / image_size / dense_units: the reference's 32 / 128 / 1024 (tests use smaller networks)."""
        self.device = torch.device(device)
        self.convs_spec = discriminator_layers(width)
        self.features = (image_size // 32) ** 2 * self.convs_spec[-1][1]
        shapes = []
        for cin, cout, _ in self.convs_spec:
            shapes += [BlockedConv.kernel_shape(cin, cout), (cout,)]
        shapes += [(self.features, dense_units), (dense_units,), (dense_units, 1), (1,)]
        self.pool = ParamPool(shapes, self.device)
        P = self.pool
        self.convs = [BlockedConv(cin, cout, stride, 'lrelu', P.view(2 * i), P.view(2 * i + 1), P.view(2 * i, P.grads), P.view(2 * i + 1, P.grads))
                      for i, (cin, cout, stride) in enumerate(self.convs_spec)]
        j = 2 * len(self.convs)
        self.dense = [DenseLayer(self.features, dense_units, 'lrelu', P.view(j), P.view(j + 1), P.view(j, P.grads), P.view(j + 1, P.grads)),
                      DenseLayer(dense_units, 1, 'sigmoid', P.view(j + 2), P.view(j + 3), P.view(j + 2, P.grads), P.view(j + 3, P.grads))]
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for c in self.convs:                           # truncated_normal(stddev=0.02) (:122); biases zero
            k = torch.empty((3, 3, c.cin, c.cout))
            truncated_normal_(k, 0.02, gen)
            c.set_kernel_hwio(k)
        for d in self.dense:
            truncated_normal_(d.w, 0.02, gen)
        self._saved = None

    def variables(self):
        """{tf variable name: array-like in TensorFlow's layout} (kernels HWIO)."""
        out = {}
        for i, c in enumerate(self.convs):
            scope = 'd_/conv2d' if i == 0 else 'd_/conv2d_%d' % i
            out[scope + '/kernel'], out[scope + '/bias'] = c.kernel_hwio(), c.b
        for i, d in enumerate(self.dense):
            scope = 'd_/dense' if i == 0 else 'd_/dense_%d' % i
            out[scope + '/kernel'], out[scope + '/bias'] = d.w, d.b
        return out

    def gradients(self):
        out = {}
        for i, c in enumerate(self.convs):
            scope = 'd_/conv2d' if i == 0 else 'd_/conv2d_%d' % i
            out[scope + '/kernel'], out[scope + '/bias'] = c.kernel_hwio(c.dw), c.db
        for i, d in enumerate(self.dense):
            scope = 'd_/dense' if i == 0 else 'd_/dense_%d' % i
            out[scope + '/kernel'], out[scope + '/bias'] = d.dw, d.db
        return out

    def set_params(self, convs, dense):
        """convs: 10 x (kernel HWIO, bias); dense: 2 x (W, b)."""
        for c, (k, b) in zip(self.convs, convs):
            c.set_kernel_hwio(k, b)
        for d, (w, b) in zip(self.dense, dense):
            d.w.copy_(torch.as_tensor(w, dtype=torch.float32).to(self.device))
            d.b.copy_(torch.as_tensor(b, dtype=torch.float32).to(self.device))

    def forward(self, images, keep=False):
        """images [N,S,S,3] in [-1,1] -> probabilities [N,1]."""
        t = to_blocks(images.contiguous())
        acts = [t]
        for c in self.convs:
            t = c.forward(t)
            acts.append(t)
        flat = to_nhwc(t).contiguous().view(t.shape[1], -1)               # tf.layers.flatten of NHWC (:148)
        if flat.shape[1] != self.features:
            raise ValueError('discriminator built for %d features, got %d (its dense layer fixes the image size, '
                             'model_enet.py:148-154)' % (self.features, flat.shape[1]))
        h = self.dense[0].forward(flat)
        p = self.dense[1].forward(h)
        self._saved = (acts, flat, h, p) if keep else None
        return p

    def backward(self, dp, want_dx=True, want_dw=True):
        """dp = d(loss)/d(p) [N,1] for the last forward(keep=True).  want_dw: fill the gradient buffer (d_trainer);
        want_dx: return d(loss)/d(images) (the generator's adversarial gradient)."""
        acts, flat, h, p = self._saved
        dh = self.dense[1].backward(h, p, dp, want_dx=True, want_dw=want_dw)
        dflat = self.dense[0].backward(flat, h, dh, want_dx=True, want_dw=want_dw)
        last = acts[-1]
        g = to_blocks(dflat.view(last.shape[1], last.shape[2], last.shape[3], -1))
        masked = False                 # g already carries the leaky-ReLU gradient of the layer it belongs to
        for i in range(len(self.convs) - 1, -1, -1):
            c, y = self.convs[i], acts[i + 1]
            if masked:
                dpre = g
            else:
                dpre = ops.act_bwd(g.contiguous(), y, 'lrelu', out=torch.empty_like(y))
            stuffed = c.full_res(dpre)          # stride-2 layers: the zero-stuffed gradient, once for both gradients
            if want_dw:
                c.wgrad(acts[i], dpre, stuffed=stuffed)
            if i == 0 and not want_dx:
                return None
            # the activation gradient of layer i - 1 (whose output acts[i] is this layer's input) rides in the launch
            g = c.dgrad(dpre, mask=acts[i] if i > 0 else None, mask_act='lrelu' if i > 0 else None, stuffed=stuffed)
            masked = i > 0
        return to_nhwc(g)


# ---------------------------------------------------------------------------------------------------------------------
# losses on VGG-19 features (model_enet.py:185-261)
# ---------------------------------------------------------------------------------------------------------------------
PERCEPTUAL_LAYERS = (('block2_pool', 0.2), ('block5_pool', 0.02))                               # :186-206
TEXTURE_LAYERS = (('block1_conv1', 3e-7), ('block2_conv1', 1e-6), ('block3_conv1', 1e-6))      # :214-218


def perceptual_loss(sr_feats, hd_feats, loss_out, want_grad=True):
    """loss_out = 0.2 * MSE(normalize(block2_pool)) + 0.02 * MSE(normalize(block5_pool)); returns
    {layer: d loss / d sr feature (NHWC)}."""
    dt = {}
    for j, (name, wgt) in enumerate(PERCEPTUAL_LAYERS):
        s, h = model_vgg.Vgg19.tap(sr_feats, name), model_vgg.Vgg19.tap(hd_feats, name)
        sn, hn = ops.channel_normalize(s), ops.channel_normalize(h)
        # loss (+)= wgt * mean((sn - hn)^2); dsn = wgt * 2 (sn - hn) / numel
        dsn = ops.mse_fwd_bwd(sn, hn, loss_out, inv_numel=wgt / sn.numel(), accumulate=j > 0, want_grad=want_grad)
        if want_grad:
            dt[name] = ops.channel_normalize_bwd(s, dsn)
    return dt


# SRX_TEXTURE_FUSED=0: normalise, patch extraction and gram GEMM as separate launches (A/B; also the route of any
# channel count the fused kernel has no instance for)
FUSED_TEXTURE = os.environ.get('SRX_TEXTURE_FUSED', '1') != '0'


def texture_matching_loss(sr_feats, hd_feats, loss_out, want_grad=True):
    """loss_out = sum_l w_l * MSE(gram(sr_l), gram(hd_l)), gram = x^T x over each 16x16 patch of the normalised
    features ([N, h*w/256, 256, C] -> [N, h*w/256, C, C]); returns {layer: d loss / d sr feature (NHWC)}."""
    dt = {}
    for j, (name, wgt) in enumerate(TEXTURE_LAYERS):
        s, h = model_vgg.Vgg19.tap(sr_feats, name), model_vgg.Vgg19.tap(hd_feats, name)
        n, hh, ww, c = s.shape
        if FUSED_TEXTURE and c in (64, 128, 256):
            # normalise + patches + gram in one pass over the feature tensor (srx_texture_gram), and its gradient
            s, h = s.contiguous(), h.contiguous()
            gs, gh = ops.texture_gram(s), ops.texture_gram(h)
            dg = ops.mse_fwd_bwd(gs, gh, loss_out, inv_numel=wgt / gs.numel(), accumulate=j > 0, want_grad=want_grad)
            if want_grad:
                dt[name] = ops.texture_gram_bwd(s, dg)
            continue
        sp = ops.extract_patches16(ops.channel_normalize(s)).view(-1, 256, c)
        hp = ops.extract_patches16(ops.channel_normalize(h)).view(-1, 256, c)
        gs = ops.gemm(sp, sp, trans_a=True)                                 # [N*P, C, C]
        gh = ops.gemm(hp, hp, trans_a=True)
        dg = ops.mse_fwd_bwd(gs, gh, loss_out, inv_numel=wgt / gs.numel(), accumulate=j > 0, want_grad=want_grad)
        if want_grad:
            # d(x^T x) -> dx = x (dG + dG^T); dG is symmetric here (a difference of gram matrices): dx = 2 x dG
            dsp = ops.gemm(sp, dg, alpha=2.0)
            dsn = ops.extract_patches16_bwd(dsp.view(n, -1, 256, c), (n, hh, ww, c))
            dt[name] = ops.channel_normalize_bwd(s, dsn)
    return dt


# ---------------------------------------------------------------------------------------------------------------------
# build_enet (model_enet.py:264-350)
# ---------------------------------------------------------------------------------------------------------------------
class EnetModel(object):
    """Generator + VGG-19 + discriminator + the two Adam trainers.  The eager API works on device tensors:
      g_step(sd, bq, hd) -> losses      one `session.run(g_trainer)` (step += 1)
      d_step(sd, bq, hd) -> a_loss      one `session.run(d_trainer)`."""

    def __init__(self, pat_model='pat', vgg_weights=None, device='cuda', seed=None, d_width=32, image_size=128, dense_units=1024):
        self.device = torch.device(device)
        self.pat_model = pat_model
        self.generator = EnetGenerator(device=device, seed=seed)
        self.vgg = model_vgg.Vgg19(vgg_weights, device=device)
        self.discriminator = Discriminator(device=device, seed=None if seed is None else seed + 1, width=d_width,
                                           image_size=image_size, dense_units=dense_units) if 'a' in pat_model else None
        self.g_state = {}
        self.global_step = 0
        self.losses = {k: torch.zeros(1, dtype=torch.float32, device=self.device)
                       for k in ('p_loss', 't_loss', 'g_loss', 'a_loss', 'g_loss_all')}
        self.placeholders = {}
        self.last_sr = None
        self.grad_hook_g = self.grad_hook_d = None       # data parallelism: called with the flat gradient buffers

    # ---- objective and its gradient w.r.t. sr_images ------------------------------------------------------------
    def generator_objective(self, sr, hd, want_grad=True, want_a_loss=False):
        """Fills self.losses (device scalars) and returns d(g_losses)/d(sr_images) (model_enet.py:286-326).
        want_a_loss: also evaluate a_loss = log_loss(0, D(sr)) + log_loss(1, D(hd)) (:301; not part of g_losses)."""
        L = self.losses
        hd_f = self.vgg.forward(hd, keep=False)
        sr_f = self.vgg.forward(sr, keep=want_grad)                        # (last: backward() uses what it saved)
        dt = perceptual_loss(sr_f, hd_f, L['p_loss'], want_grad)
        ops.add_scaled(L['p_loss'], out=L['g_loss_all'])
        d_sr = None
        if 'a' in self.pat_model:
            D = self.discriminator
            gw = 2.0 if 't' in self.pat_model else 1.0                     # :310-313
            if want_a_loss:
                ops.log_loss(D.forward(hd), 1.0, L['a_loss'], want_grad=False)
            fake = D.forward(sr, keep=want_grad)
            if want_a_loss:
                ops.log_loss(fake, 0.0, L['a_loss'], accumulate=True, want_grad=False)
            dp = ops.log_loss(fake, 1.0, L['g_loss'], grad_scale=gw, want_grad=want_grad)      # generator_loss (:165-169)
            ops.add_scaled(L['g_loss_all'], L['g_loss'], 1.0, gw, out=L['g_loss_all'])
            if want_grad:
                d_sr = D.backward(dp, want_dx=True, want_dw=False).contiguous()
        if 't' in self.pat_model:
            dtt = texture_matching_loss(sr_f, hd_f, L['t_loss'], want_grad)
            ops.add_scaled(L['g_loss_all'], L['t_loss'], out=L['g_loss_all'])
            for k, v in dtt.items():
                dt[k] = ops.add_scaled(dt[k], v, out=v) if k in dt else v
        if not want_grad:
            return None
        d_vgg = self.vgg.backward(dt)
        return d_vgg if d_sr is None else ops.add_scaled(d_vgg, d_sr, out=d_vgg)

    # ---- trainers ---------------------------------------------------------------------------------------------------
    def g_step(self, sd, bq, hd, lr=1e-4, want_a_loss=False):
        sr = self.last_sr = self.generator.forward(sd, bq, keep=True)
        d_sr = self.generator_objective(sr, hd, want_a_loss=want_a_loss)
        grads = self.generator.backward(d_sr)
        if self.grad_hook_g is not None:
            self.grad_hook_g(self.generator.grads)
        self.generator.adam_step(grads, self.g_state, lr)                 # AdamOptimizer(0.0001) on g_vars (:336-337)
        self.global_step += 1
        return self.losses

    def d_step(self, sd, bq, hd, lr=1e-4):
        """a_loss = log_loss(0, D(G(sd))) + log_loss(1, D(hd)); Adam(0.0001) on the d_ variables (:339-343).  One
        discriminator pass over the concatenated batch [fake; real]: each half's mean has its own 1/N."""
        D = self.discriminator
        sr = self.generator.forward(sd, bq)
        n = sr.shape[0]
        both = torch.cat([sr, hd], dim=0)
        p = D.forward(both, keep=True)
        dp = torch.empty_like(p)
        dp[:n] = ops.log_loss(p[:n].contiguous(), 0.0, self.losses['a_loss'])
        dp[n:] = ops.log_loss(p[n:].contiguous(), 1.0, self.losses['a_loss'], accumulate=True)
        D.backward(dp, want_dx=False, want_dw=True)
        if self.grad_hook_d is not None:
            self.grad_hook_d(D.pool.grads)
        D.pool.adam_step(lr)
        return self.losses['a_loss']

    # ---- checkpoints (tf.train.Saver format: experiment_train.py:99-110) ------------------------------------------
    def _named_buffers(self):
        """[(tf variable name, value view, Adam m view or None, Adam v view or None, to_tf, from_tf)]: kernels are
        exchanged in TensorFlow's HWIO layout."""
        G, D = self.generator, self.discriminator
        gm, gv = self.g_state.get('m'), self.g_state.get('v')
        out = []
        ident = (lambda t: t, lambda a, like: a)
        for i, (k, b) in enumerate(zip(G.kernels, G.biases)):
            scope = 'g_/conv2d' if i == 0 else 'g_/conv2d_%d' % i
            for name, t in ((scope + '/kernel', k), (scope + '/bias', b)):
                off = t.data_ptr() - G.params.data_ptr()
                sl = lambda buf, off=off, t=t: None if buf is None else buf[off // 4:off // 4 + t.numel()].view(t.shape)
                out.append((name, t, sl(gm), sl(gv)) + ident)
        if D is not None:
            P = D.pool
            for i, c in enumerate(D.convs):
                scope = 'd_/conv2d' if i == 0 else 'd_/conv2d_%d' % i
                to_tf = lambda w, c=c: w.permute(2, 3, 0, 4, 1, 5).reshape(3, 3, c.cin, c.cout)
                from_tf = lambda a, like, c=c: a.reshape(3, 3, c.cib, c.ci, c.cob, c.co).permute(2, 4, 0, 1, 3, 5)
                out.append((scope + '/kernel', P.view(2 * i), None if P.opt_m is None else P.view(2 * i, P.opt_m),
                            None if P.opt_v is None else P.view(2 * i, P.opt_v), to_tf, from_tf))
                out.append((scope + '/bias', P.view(2 * i + 1), None if P.opt_m is None else P.view(2 * i + 1, P.opt_m),
                            None if P.opt_v is None else P.view(2 * i + 1, P.opt_v)) + ident)
            j = 2 * len(D.convs)
            for i in range(2):
                scope = 'd_/dense' if i == 0 else 'd_/dense_%d' % i
                for kind, idx in (('kernel', j + 2 * i), ('bias', j + 2 * i + 1)):
                    out.append((scope + '/' + kind, P.view(idx), None if P.opt_m is None else P.view(idx, P.opt_m),
                                None if P.opt_v is None else P.view(idx, P.opt_v)) + ident)
        return out

    def tf_checkpoint_tensors(self):
        """{checkpoint key: ndarray} as `tf.train.Saver()` of the reference's training graph names them: the g_ / d_
        variables (kernels HWIO), `global_step`, and per optimizer the slots `<var>/Adam`, `<var>/Adam_1` with the
        power accumulators `beta1_power`, `beta2_power` (g_trainer, created first) and `beta1_power_1`,
        `beta2_power_1` (d_trainer): beta ** (steps + 1), as in engine.ConvStack.tf_checkpoint_tensors.  (Slot and
        accumulator names as TensorFlow 1.8 is remembered to create them; no TensorFlow-written file to check against.)"""
        from .. import tf_bundle
        out = {'global_step': np.asarray(self.global_step, dtype=np.int64)}
        for name, val, m, v, to_tf, _ in self._named_buffers():
            out[name] = to_tf(val).detach().cpu().numpy()
            if m is not None:
                out[name + '/Adam'] = to_tf(m).detach().cpu().numpy()
                out[name + '/Adam_1'] = to_tf(v).detach().cpu().numpy()
        if self.g_state:
            out['beta1_power'] = tf_bundle.tf_beta_power(0.9, self.g_state['t'])
            out['beta2_power'] = tf_bundle.tf_beta_power(0.999, self.g_state['t'])
        if self.discriminator is not None and self.discriminator.pool.opt_m is not None:
            out['beta1_power_1'] = tf_bundle.tf_beta_power(0.9, self.discriminator.pool.t)
            out['beta2_power_1'] = tf_bundle.tf_beta_power(0.999, self.discriminator.pool.t)
            # not a TensorFlow variable (a Saver restoring by name ignores it): the count the powers stand for
            out['srx/d_trainer_steps'] = np.asarray(self.discriminator.pool.t, dtype=np.int64)
        return out

    def save_tf_checkpoint(self, prefix):
        from .. import tf_bundle
        tf_bundle.save_checkpoint(prefix, self.tf_checkpoint_tensors())
        tf_bundle.update_checkpoint_state(prefix)

    @staticmethod
    def _d_steps_from_checkpoint(values, global_step):
        """Adam step count of d_trainer.  In order: (1) the explicit count this package writes beside the TensorFlow
        variables; (2) the schedule of the training script -- d_trainer runs on every third step
        (enet/enet/experiment_train.py:112), so a checkpoint the reference wrote at `global_step` has seen exactly
        (global_step + 2) // 3 of them; (3) only if beta2_power_1 contradicts that schedule (a checkpoint trained under
        another one): the inverse of beta2_power_1 = prod of (t + 1) float32 multiplications by float32(0.999) =
        0.99900001287..., which stays a normal float32 for ~87,000 steps.  TensorFlow's repeated float32 rounding makes
        that inverse drift by about 1.3e-5 * t steps (off by one from t ~ 40,000 on), hence only the fallback.
        beta1_power_1 = 0.9 ** (t + 1) underflows after ~960 steps and is never used."""
        if 'srx/d_trainer_steps' in values:
            return int(values['srx/d_trainer_steps'])
        sched = (int(global_step) + 2) // 3
        b2p = float(values['beta2_power_1']) if 'beta2_power_1' in values else 0.0
        if np.isfinite(b2p) and 1.2e-38 < b2p < 1.0:
            t2 = max(int(round(np.log(b2p) / np.log(float(np.float32(0.999))))) - 1, 0)
            if abs(t2 - sched) > max(2, int(1e-4 * sched)):
                return t2
        return sched

    def load_tf_checkpoint(self, prefix):
        """Restores the g_ / d_ variables, global_step and (when present) both optimizers' state."""
        from .. import tf_bundle
        values = tf_bundle.load_checkpoint(prefix)
        self.global_step = int(values.get('global_step', 0))
        G, D = self.generator, self.discriminator
        have_g = 'g_/conv2d/kernel/Adam' in values
        have_d = D is not None and 'd_/conv2d/kernel/Adam' in values
        if have_g and not self.g_state:
            self.g_state.update({'t': 0, 'm': torch.zeros_like(G.params), 'v': torch.zeros_like(G.params)})
        if have_d and D.pool.opt_m is None:
            D.pool.opt_m, D.pool.opt_v = torch.zeros_like(D.pool.params), torch.zeros_like(D.pool.params)
        if have_g:
            # g_trainer is the only op that increments global_step (model_enet.py:336-337): its Adam step count IS
            # global_step.  (beta1_power = 0.9 ** (t + 1) cannot be inverted: it is 0.0 in float32 from t = 985 on, and
            # the training script saves at step % 1000 == 999.)
            self.g_state['t'] = self.global_step
        if have_d:
            D.pool.t = self._d_steps_from_checkpoint(values, self.global_step)
        for name, val, m, v, _, from_tf in self._named_buffers():
            if name not in values:
                raise KeyError('checkpoint %s lacks variable %s' % (prefix, name))
            put = lambda dst, arr: dst.copy_(from_tf(torch.as_tensor(np.asarray(arr), dtype=torch.float32), dst).to(dst.device))
            put(val, values[name])
            if m is not None and name + '/Adam' in values:
                put(m, values[name + '/Adam'])
                put(v, values[name + '/Adam_1'])

    # ---- Session.run backend ---------------------------------------------------------------------------------------
    def run(self, keys, feed_dict):
        feeds = {name: feed_dict[ph] for name, ph in self.placeholders.items() if ph in feed_dict}
        out = {}
        need = [k for k in keys if k != 'step']
        if need:
            for k in ('sd_images', 'bq_images'):
                if k not in feeds:
                    raise ValueError('%s must be fed to fetch %s' % (k, ', '.join(need)))
            sd = graph.to_device(feeds['sd_images'], self.device)
            bq = graph.to_device(feeds['bq_images'], self.device)
            hd = graph.to_device(feeds['hd_images'], self.device) if 'hd_images' in feeds else None
            loss_keys = [k for k in need if k in self.losses]
            if ('g_trainer' in keys or 'd_trainer' in keys or loss_keys) and hd is None:
                raise ValueError('hd_images must be fed to fetch losses / trainers')
            sr = None
            want_a = 'a_loss' in keys and 'a' in self.pat_model
            if 'd_trainer' in keys:
                self.d_step(sd, bq, hd)                 # (leaves a_loss of its batch in self.losses)
                want_a = False
            if 'g_trainer' in keys:
                self.g_step(sd, bq, hd, want_a_loss=want_a)
                sr = self.last_sr
            elif [k for k in loss_keys if k != 'a_loss' or want_a]:
                sr = self.generator.forward(sd, bq)
                self.generator_objective(sr, hd, want_grad=False, want_a_loss=want_a)
            vals = {'sd_images': sd, 'bq_images': bq, 'hd_images': hd}
            for k in need:
                if k in ('g_trainer', 'd_trainer'):
                    out[k] = None
                elif k in self.losses:
                    out[k] = float(self.losses[k].item())
                elif k == 'sr_images':
                    sr = sr if sr is not None else self.generator.forward(sd, bq)
                    out[k] = sr.detach().cpu().numpy()
                else:
                    out[k] = vals[k].detach().cpu().numpy()
        if 'step' in keys:
            out['step'] = self.global_step
        return out


def build_enet(sd_images, bq_images, hd_images=None, pat_model='pat', vgg19_path=None, device='cuda', seed=None,
               vgg_weights=None):
    """enet/enet/model_enet.py:264-350.  hd_images None: the inference graph {sd_images, bq_images, sr_images}.
    Otherwise the training graph; `vgg19_path` is the .npz of VGG-19 weights the reference loads (:283), or pass
    `vgg_weights` ({layer: {layer_W_1, layer_b_1}}) directly."""
    if hd_images is None:
        g = EnetGenerator(device=device, seed=seed)
        g.placeholders['sd_images'] = sd_images
        g.placeholders['bq_images'] = bq_images
        return {'sd_images': sd_images, 'bq_images': bq_images,
                'sr_images': graph.Tensor('sr_images', owner=g, key='sr_images'), '_model': g}
    if pat_model not in ('p', 'pa', 'pat'):
        pat_model = 'pat'                                # experiment_train.py:88-89
    if vgg_weights is None:
        vgg_weights = model_vgg.load_vgg_weights(vgg19_path)
        if not vgg_weights:
            raise ValueError('VGG-19 weights not found at %r (the reference downloads them: model_vgg.py:4-5)' % (vgg19_path,))
    m = EnetModel(pat_model, vgg_weights, device=device, seed=seed)
    m.placeholders.update({'sd_images': sd_images, 'bq_images': bq_images, 'hd_images': hd_images})
    T = lambda key: graph.Tensor(key, owner=m, key=key)
    model = {'sd_images': sd_images, 'bq_images': bq_images, 'sr_images': T('sr_images'), 'hd_images': hd_images, '_model': m}
    if 'a' in pat_model:
        model.update({'a_loss': T('a_loss'), 'g_loss': T('g_loss'), 'd_trainer': T('d_trainer')})
    if 't' in pat_model:
        model['t_loss'] = T('t_loss')
    model.update({'step': T('step'), 'p_loss': T('p_loss'), 'g_loss_all': T('g_loss_all'), 'g_trainer': T('g_trainer')})
    return model
