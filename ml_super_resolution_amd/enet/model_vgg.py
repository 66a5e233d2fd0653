"""
model_vgg.py -- mirror of enet/enet/model_vgg.py (reference): VGG-19's convolutional part as a FIXED feature extractor
(constant weights, model_vgg.py:39-62), on the MI355X engine.

  load_vgg_weights(weights_path) -> {layer_name: {layer_name_W_1: array, layer_name_b_1: array}}   (:39-62)
  build_vgg19_model / Vgg19.forward -> {layer_name: tensor} for the 21 layers                     (:65-99)

Input: images in [-1, 1], RGB; the reference scales them to 0..255 before the network (model_enet.py:288-289) and the
network reverses the channels and subtracts the mean colour (:72-76) -- one kernel here (srx_vgg_preprocess).
Every conv is 3x3 SAME + bias + ReLU (:11-25), every pool 2x2 / 2 SAME max-pooling (:28-36).  Layers wider than 64
channels run on the 64-channel kernels through blocked.BlockedConv.

The reference downloads the Keras weights (`vgg19_weights_tf_dim_ordering_tf_kernels_notop` as .npz, :4-5); they are
not available offline: tests and benchmarks use random VGG-shaped weights (timing only, parity unpinned).
`Vgg19.backward` propagates gradients given on any of the layers back to the input image -- what TensorFlow's autodiff
does for the perceptual / texture losses of model_enet.py:185-261.
"""
import numpy as np
import torch

from .. import ops
from ..blocked import BlockedConv, to_blocks, to_nhwc

LAYER_NAMES = [
    'block1_conv1', 'block1_conv2', 'block1_pool',
    'block2_conv1', 'block2_conv2', 'block2_pool',
    'block3_conv1', 'block3_conv2', 'block3_conv3', 'block3_conv4', 'block3_pool',
    'block4_conv1', 'block4_conv2', 'block4_conv3', 'block4_conv4', 'block4_pool',
    'block5_conv1', 'block5_conv2', 'block5_conv3', 'block5_conv4', 'block5_pool']      # model_vgg.py:79-88


def conv_channels(width=64):
    """{conv layer: (cin, cout)}; width = channels of block 1 (64 for VGG-19; tests use narrower ones)."""
    out, cin = {}, 3
    for name in LAYER_NAMES:
        if name.endswith('pool'):
            continue
        cout = width * min(2 ** (int(name[5]) - 1), 8)
        out[name] = (cin, cout)
        cin = cout
    return out


def load_vgg_weights(weights_path):
    """{scope: {const_name: ndarray}} from the .npz the reference reads (model_vgg.py:39-62): array names are
    `<layer>_W_1:0` / `<layer>_b_1:0`; scope = the first 12 characters, constant name = the name without ':0'."""
    weights = {}
    try:
        data = np.load(weights_path, encoding='bytes')
    except (OSError, TypeError):
        return weights                       # the reference returns {} when the file does not exist (:45-46)
    for name in data.files:
        scope_name, const_name = name[:12], name[:-2]
        weights.setdefault(scope_name, {})[const_name] = data[name]
    return weights


def random_vgg_weights(seed=0, width=64):
    """VGG-shaped weights with He-scaled random kernels (the real ones are not available offline)."""
    rng = np.random.default_rng(seed)
    weights = {}
    for name, (cin, cout) in conv_channels(width).items():
        k = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (3, 3, cin, cout)).astype(np.float32)
        if name == 'block1_conv1':
            k /= 60.0                        # its input has magnitude ~100 (0..255 minus the mean colour)
        weights[name] = {name + '_W_1': k, name + '_b_1': rng.normal(0, 0.05, cout).astype(np.float32)}
    return weights


class Vgg19(object):
    def __init__(self, weights, device='cuda'):
        self.device = torch.device(device)
        self.layers = {}
        for name in LAYER_NAMES:
            if name.endswith('pool'):
                continue
            k = np.asarray(weights[name][name + '_W_1'], np.float32)          # model_vgg.py:15-19
            b = np.asarray(weights[name][name + '_b_1'], np.float32)
            cin, cout = k.shape[2], k.shape[3]
            layer = BlockedConv(cin, cout, 1, 'relu',
                                kernel=torch.empty(BlockedConv.kernel_shape(cin, cout), dtype=torch.float32, device=self.device),
                                bias=torch.empty((cout,), dtype=torch.float32, device=self.device))
            layer.set_kernel_hwio(k, b)
            self.layers[name] = layer

    def forward(self, images_pm1, keep=False):
        """images in [-1, 1] RGB [N,H,W,3] -> {layer_name: blocked tensor [CB,N,h,w,64]}; keep=True retains what
        backward() needs."""
        t = to_blocks(ops.vgg_preprocess(images_pm1.contiguous()))
        feats = {'input': t}
        for name in LAYER_NAMES:
            if name.endswith('pool'):
                cb, n, h, w, c = t.shape
                # (the blocks of a channel-blocked tensor are contiguous: one launch with the blocks as extra images)
                p = torch.empty((cb, n, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float32, device=t.device)
                ops.maxpool2x2(t.view(cb * n, h, w, c), out=p.view(cb * n, (h + 1) // 2, (w + 1) // 2, c))
                t = p
            else:
                t = self.layers[name].forward(t)
            feats[name] = t
        self._saved = feats if keep else None
        return feats

    @staticmethod
    def tap(feats, name):
        """A layer's output as a plain NHWC tensor (what the loss operators take)."""
        return to_nhwc(feats[name])

    def backward(self, dtaps):
        """dtaps: {layer_name: NHWC gradient w.r.t. that layer's output}.  Returns d(loss)/d(images) [N,H,W,3] for the
        images of the last forward(keep=True)."""
        feats = self._saved
        if feats is None:
            raise RuntimeError('backward() needs a forward(..., keep=True) first')
        def relu_grad(t, y):                                               # ReluGrad on the post-ReLU tensor
            return ops.act_bwd(t.contiguous(), y, 'relu', out=torch.empty_like(y))

        g, masked = None, False        # masked: g is already multiplied by ReluGrad of the layer it belongs to
        for idx in range(len(LAYER_NAMES) - 1, -1, -1):
            name = LAYER_NAMES[idx]
            if name in dtaps:
                d = to_blocks(dtaps[name])
                if g is not None and masked:
                    # g already carries this layer's ReluGrad and the tap does not: (y > 0) ? g + d : 0 in one pass
                    g = ops.add_relu_grad(g.contiguous(), d.contiguous(), feats[name], out=g)
                else:
                    if masked:
                        d = relu_grad(d, feats[name])
                    g = d if g is None else ops_add_(g, d)
            if g is None:
                continue
            below_name = LAYER_NAMES[idx - 1] if idx > 0 else 'input'
            below = feats[below_name]
            if name.endswith('pool'):
                dx = torch.empty_like(below)
                cb, n, h, w, c = below.shape
                g = g.contiguous()
                # (every pooling layer of VGG-19 follows a ReLU convolution: its ReluGrad rides in the pooling gradient)
                fuse = 'conv' in below_name
                ops.maxpool2x2_bwd(below.view(cb * n, h, w, c), g.view(cb * n, g.shape[2], g.shape[3], c), out=dx.view(cb * n, h, w, c),
                                   mask_act='relu' if fuse else None)
                g, masked = dx, fuse
            else:
                dpre = g if masked else relu_grad(g, feats[name])
                # the ReluGrad of the conv layer below rides in this layer's data-gradient launch
                fuse = 'conv' in below_name
                g = self.layers[name].dgrad(dpre, mask=below if fuse else None, mask_act='relu' if fuse else None)
                masked = fuse
        return ops.vgg_preprocess(to_nhwc(g).contiguous(), backward=True)


def ops_add_(a, b):
    """a += b on the device (gradients arriving at one tensor from two consumers)."""
    return ops.add_scaled(a, b, out=a)


def build_vgg19_model(source_images, weights, device='cuda'):
    """The reference takes images already scaled to 0..255 (model_vgg.py:65-70); here `source_images` are the
    [-1, 1] images and the scaling is part of the fused input map.  Returns the feature dict of Vgg19.forward."""
    return Vgg19(weights, device=device).forward(source_images)
