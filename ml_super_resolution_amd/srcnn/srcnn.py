"""
srcnn.py -- mirror of srcnn/srcnn.py (reference) on the MI355X engine.

Keeps the reference's flag names/defaults (srcnn.py:8-23, dashes become underscores as tf.app.flags
does), `sanity_check()` (:28-43) and `build_srcnn()` with its result keys
{step, loss, trainer, hd_images, sd_images, sr_images} (:159-166).

The network (srcnn.py:100-130) is three VALID convolutions: 9x9 -> 64 ReLU, 1x1 -> 32 ReLU,
5x5 -> 3 tanh; the loss (:142-144) is the mean over rows of ||reshape(sr - hd, [-1, bb*bb])||_2;
the optimizer Adam(1e-3, beta1 .5, beta2 .9) (:155-157).
The reference builds its inputs in-graph (JPEG queue, random crop, bicubic down/up, :46-93); that
input pipeline is outside the hot path: here `hd_images` / `sd_images` are fed.
"""
import argparse

import torch

from .. import graph
from ..engine import ConvStack, LayerSpec, truncated_normal_


def _flags():
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--ckpt-dir-path', default='./ckpts/')
    ap.add_argument('--logs-dir-path', default='./logs/')
    ap.add_argument('--training-images-path', default=None)
    ap.add_argument('--sr-source-path', default=None)
    ap.add_argument('--sr-target-path', default=None)
    ap.add_argument('--train', action='store_true')
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--upscaling-factor', type=int, default=3)
    ap.add_argument('--crop-image-size', type=int, default=256)
    ap.add_argument('--crop-image-side', type=int, default=6)
    ap.add_argument('--srcnn-fsub', type=int, default=33)
    ap.add_argument('--srcnn-f1', type=int, default=9)
    ap.add_argument('--srcnn-f2', type=int, default=1)
    ap.add_argument('--srcnn-f3', type=int, default=5)
    ap.add_argument('--srcnn-n1', type=int, default=64)
    ap.add_argument('--srcnn-n2', type=int, default=32)
    return ap


FLAGS = _flags().parse_args([])


def sanity_check(flags=None):
    """srcnn.py:28-43 (the reference's `/` is Python-2 integer division)."""
    f = flags or FLAGS
    smaller_output_size = f.srcnn_fsub - f.srcnn_f1 - f.srcnn_f2 - f.srcnn_f3 + 3
    boundary = (f.srcnn_fsub - smaller_output_size) // 2
    crop_size = (f.crop_image_size - boundary * 2) // smaller_output_size
    f.crop_image_side = boundary
    f.crop_image_size = crop_size * smaller_output_size + boundary * 2
    if not f.train:
        f.batch_size = 1
    return f


def layer_specs(flags=None):
    f = flags or FLAGS
    return [LayerSpec(f.srcnn_f1, 3, f.srcnn_n1, 'valid', 'relu', 'patch_extraction'),
            LayerSpec(f.srcnn_f2, f.srcnn_n1, f.srcnn_n2, 'valid', 'relu', 'non_linear_mapping'),
            LayerSpec(f.srcnn_f3, f.srcnn_n2, 3, 'valid', 'tanh', 'reconstruction')]


class SrcnnModel(object):
    def __init__(self, flags=None, device='cuda', seed=None):
        self.flags = flags or FLAGS
        self.stack = ConvStack(layer_specs(self.flags), device=device, residual=False, weight_decay=0.0)
        self.stack.loss_kind = 'rownorm'
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for i in range(3):
            truncated_normal_(self.stack.kernel(i), 0.001, gen)    # srcnn.py:84; biases zero
        self.placeholders = {}

    def crop_side(self):
        f = self.flags
        return (f.srcnn_f1 - 1 + f.srcnn_f2 - 1 + f.srcnn_f3 - 1) // 2

    def forward(self, sd_images, keep=False):
        """[N,S,S,3] bicubic-interpolated input -> [N,S-12,S-12,3] (VALID 9-1-5)."""
        return self.stack.forward(sd_images, keep=keep)

    def train_step(self, sd_images, hd_images_cropped):
        """hd_images_cropped: ground truth cropped to the VALID region (srcnn.py:132-136)."""
        self.stack.forward(sd_images, keep=True)
        loss = self.stack.loss_and_backward(hd_images_cropped)
        self.stack.adam_step(0.001, beta1=0.5, beta2=0.9)          # srcnn.py:155-157
        return loss

    def run(self, keys, feed_dict):
        from .. import ops
        dev = self.stack.device
        feeds = {name: feed_dict[ph] for name, ph in self.placeholders.items() if ph in feed_dict}
        if 'sd_images' not in feeds:
            raise ValueError('sd_images must be fed (the bicubic down/up-sampled input, full size)')
        sd = graph.to_device(feeds['sd_images'], dev)
        side = self.crop_side()
        hd = None
        if 'hd_images' in feeds:
            hd_full = graph.to_device(feeds['hd_images'], dev)
            hd = hd_full[:, side:hd_full.shape[1] - side, side:hd_full.shape[2] - side].contiguous()
        loss = None
        if 'trainer' in keys:
            loss = self.train_step(sd, hd)
            sr = self.stack.acts[-1]
        else:
            sr = self.stack.forward(sd, keep=True)
            if 'loss' in keys:
                loss = self.stack.loss
                ops.rownorm_loss_fwd_bwd(sr, hd, sr.shape[1] * sr.shape[2], loss, want_grad=False)
        out = {}
        for k in keys:
            if k == 'trainer':
                out[k] = None
            elif k == 'loss':
                out[k] = float(loss.item())
            elif k == 'step':
                out[k] = self.stack.global_step
            elif k == 'sr_images':
                out[k] = sr.detach().cpu().numpy()
            elif k == 'hd_images':      # cropped to the valid region, as the reference returns them
                out[k] = hd.detach().cpu().numpy()
            elif k == 'sd_images':
                out[k] = sd[:, side:sd.shape[1] - side, side:sd.shape[2] - side].detach().cpu().numpy()
            else:
                raise KeyError(k)
        return out


def build_srcnn(hd_images=None, sd_images=None, flags=None, device='cuda', seed=None):
    """-> {'step','loss','trainer','hd_images','sd_images','sr_images'} (srcnn.py:159-166).
    hd_images / sd_images: graph.placeholders for the full-size ground truth and the bicubic
    down/up-sampled input (created here when omitted)."""
    m = SrcnnModel(flags, device=device, seed=seed)
    hd_images = hd_images or graph.placeholder([None, None, None, 3], name='hd_images')
    sd_images = sd_images or graph.placeholder([None, None, None, 3], name='sd_images')
    m.placeholders['hd_images'] = hd_images
    m.placeholders['sd_images'] = sd_images
    return {
        'step': graph.Tensor('global_step', owner=m, key='step'),
        'loss': graph.Tensor('loss', owner=m, key='loss'),
        'trainer': graph.Tensor('trainer', owner=m, key='trainer'),
        'hd_images': graph.Tensor('hd_images', owner=m, key='hd_images'),
        'sd_images': graph.Tensor('sd_images', owner=m, key='sd_images'),
        'sr_images': graph.Tensor('sr_images', owner=m, key='sr_images'),
        '_model': m, '_feed_hd_images': hd_images, '_feed_sd_images': sd_images,
    }
