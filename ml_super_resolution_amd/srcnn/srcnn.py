"""
srcnn.py -- mirror of srcnn/srcnn.py (reference) on the MI355X engine.

Keeps the reference's flag names/defaults (srcnn.py:8-23, dashes become underscores as tf.app.flags
does), `sanity_check()` (:28-43) and `build_srcnn()` with its result keys
{step, loss, trainer, hd_images, sd_images, sr_images} (:159-166).

The network (srcnn.py:100-130) is three VALID convolutions: 9x9 -> 64 ReLU, 1x1 -> 32 ReLU,
5x5 -> 3 tanh; the loss (:142-144) is the mean over rows of ||reshape(sr - hd, [-1, bb*bb])||_2;
the optimizer Adam(1e-3, beta1 .5, beta2 .9) (:155-157).
The reference builds its inputs in-graph (JPEG queue, random crop + flip, :46-82; bicubic down / up, :89-93).  Here
`hd_images` is fed (or drawn by `dataset_reader`, the host-side stand-in of the JPEG queue) and `sd_images`, when it
is not fed too, is computed ON THE GPU the way the reference's graph does: tf.image.resize_bicubic down by the
factor and up again (`srx_resize_bicubic_tf`; pinned against the reference's own panels).
`train()`, `super_resolution()` and `main()` mirror :208-298: checkpoints every 5000 steps under
`<ckpt-dir-path>/model.ckpt-<step>` in tf.train.Saver format with the graph's variable names
(`patch_extraction/weights`, `.../biases`, ...), resume from the latest, the hd | sd | sr panel as JPEG.
"""
import argparse
import glob
import os

import numpy as np
import torch

from .. import graph, ops
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
    ap.add_argument('--save-every', type=int, default=5000)      # the reference's constant (step % 5000 == 0, :253)
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
        self.stack.kernel_name, self.stack.bias_name = 'weights', 'biases'      # tf.contrib.layers.convolution2d
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for i in range(3):
            truncated_normal_(self.stack.kernel(i), 0.001, gen)    # srcnn.py:84; biases zero
        self.placeholders = {}
        self.use_single_launch = os.environ.get('SRX_SRCNN_FUSED', '1') != '0'
        # ONE launch (srx_srcnn_forward) for batches of small patches only.  Measured (scripts/time_srcnn.py, round 4): since the
        # per-layer route has its own kernels for the 9x9 3->64 and 5x5 32->3 layers, three launches win on single images of every
        # size (1 x 243^2: 52 against 58 us; 1 x 170^2: 43 against 58 -- a 15 x 15 tile costs the one-launch kernel ~55 us however few
        # of them there are); the one launch wins on many small patches (64 x 33^2: 46 against 54 us; 128 x 33^2: level).
        self.single_launch_max_pixels = int(os.environ.get('SRX_SRCNN_FUSED_MAX_PIXELS', '57600'))
        self.single_launch_min_pixels = int(os.environ.get('SRX_SRCNN_FUSED_MIN_PIXELS', '8000'))
        self.single_launch_max_patch = int(os.environ.get('SRX_SRCNN_FUSED_MAX_PATCH', '1024'))     # output pixels per image

    def crop_side(self):
        f = self.flags
        return (f.srcnn_f1 - 1 + f.srcnn_f2 - 1 + f.srcnn_f3 - 1) // 2

    def degrade(self, hd_images):
        """lo_images of srcnn.py:89-93: resize_bicubic to crop_image_size / upscaling_factor (Python-2 integer division)
        and back, on the device."""
        n, h, w, _ = hd_images.shape
        f = self.flags.upscaling_factor
        lo = ops.resize_bicubic_tf(hd_images.contiguous(), h // f, w // f)
        return ops.resize_bicubic_tf(lo, h, w)

    def forward(self, sd_images, keep=False, single_launch=None):
        """[N,S,S,3] bicubic-interpolated input -> [N,S-12,S-12,3] (VALID 9-1-5).
        Three launches (BASELINE configs[0]'s one 243 x 243 image: conv_pack3_kernel, the 1x1 layer, conv_kwrows_kernel).
        Inference (keep=False) on batches of small patches -- at most `single_launch_max_patch` output pixels per image,
        `single_launch_min_pixels` .. `single_launch_max_pixels` in all -- runs as ONE launch that chains the three layers
        through LDS (srx_srcnn_forward; bit-identical to the three launches on conv path 0, equal to rounding on the default
        path); training keeps the per-layer launches, whose activations backward needs."""
        f = self.flags
        if single_launch is None:
            n, h, w, _ = sd_images.shape
            single_launch = (self.use_single_launch and not keep and sd_images.is_cuda and (f.srcnn_f1, f.srcnn_f2, f.srcnn_f3) == (9, 1, 5)
                             and (f.srcnn_n1, f.srcnn_n2) == (64, 32) and h >= 13 and w >= 13
                             and (h - 12) * (w - 12) <= self.single_launch_max_patch
                             and self.single_launch_min_pixels <= n * (h - 12) * (w - 12) <= self.single_launch_max_pixels)
        if single_launch:
            params = [(self.stack.kernel(i), self.stack.bias(i)) for i in range(3)]
            return ops.srcnn_forward(sd_images.contiguous(), params)
        return self.stack.forward(sd_images, keep=keep)

    def train_step(self, sd_images, hd_images_cropped):
        """hd_images_cropped: ground truth cropped to the VALID region (srcnn.py:132-136)."""
        # srcnn.py:155-157: AdamOptimizer(0.001, beta1=0.5, beta2=0.9); one replayed HIP graph per batch shape
        return self.stack.train_step_replay(sd_images, hd_images_cropped, 0.001, beta1=0.5, beta2=0.9)

    def run(self, keys, feed_dict):
        from .. import ops
        dev = self.stack.device
        feeds = {name: feed_dict[ph] for name, ph in self.placeholders.items() if ph in feed_dict}
        if 'sd_images' not in feeds and 'hd_images' not in feeds:
            raise ValueError('hd_images must be fed (sd_images is then derived on the GPU as in the reference), or sd_images '
                             '(the bicubic down / up-sampled input, full size)')
        side = self.crop_side()
        hd = hd_full = None
        if 'hd_images' in feeds:
            hd_full = graph.to_device(feeds['hd_images'], dev)
            hd = hd_full[:, side:hd_full.shape[1] - side, side:hd_full.shape[2] - side].contiguous()
        sd = graph.to_device(feeds['sd_images'], dev) if 'sd_images' in feeds else self.degrade(hd_full)
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


# ---------------------------------------------------------------------------------------------------------------------
# the script: dataset reader, train(), super_resolution(), main()   (srcnn/srcnn.py:46-82, 169-298)
# ---------------------------------------------------------------------------------------------------------------------
def dataset_reader(flags, seed=None):
    """Host-side stand-in of build_dataset_reader (:46-82): JPEGs of --training-images-path (training) or the one
    --sr-source-path image, random crop to crop_image_size, random left-right flip, / 127.5 - 1; yields [B,S,S,3]."""
    from PIL import Image
    f = flags
    paths = sorted(glob.glob(os.path.join(f.training_images_path, '*.jpg'))) if f.train else [f.sr_source_path]
    if not paths:
        raise SystemExit('no *.jpg under %r' % (f.training_images_path,))
    rng = np.random.default_rng(seed)
    images = [np.asarray(Image.open(p).convert('RGB')) for p in paths]
    s, k = f.crop_image_size, 0
    while True:
        batch = []
        for _ in range(f.batch_size):
            im = images[k % len(images)]
            k += 1
            if im.shape[0] < s or im.shape[1] < s:
                raise SystemExit('image smaller than the %d-pixel crop' % s)
            y, x = int(rng.integers(0, im.shape[0] - s + 1)), int(rng.integers(0, im.shape[1] - s + 1))
            crop = im[y:y + s, x:x + s]
            if rng.random() < 0.5:
                crop = crop[:, ::-1]
            batch.append(crop.astype(np.float32) / np.float32(127.5) - np.float32(1.0))
        yield np.stack(batch)


def latest_checkpoint(ckpt_dir):
    from .. import tf_bundle
    return tf_bundle.latest_checkpoint(ckpt_dir) if ckpt_dir and os.path.isdir(ckpt_dir) else None


def build_sr_result(model, hd_full, sd_full, sr):
    """The panel hd | sd | sr (:169-184): three [B*width, width, 3] strips side by side, width = crop_image_size - 2 * side."""
    side = model.crop_side()
    crop = lambda t: t[:, side:t.shape[1] - side, side:t.shape[2] - side]
    strips = [t.reshape(1, -1, t.shape[2], 3) for t in (crop(hd_full), crop(sd_full), sr)]
    return torch.cat(strips, dim=2)


def train(flags, device='cuda', max_steps=None, seed=None, log=None):
    """srcnn.py:208-260: restore the latest checkpoint if there is one, then loop: one Adam(1e-3, .5, .9) step per batch,
    the loss every 100 steps, a checkpoint whenever step % 5000 == 0.  `max_steps` (not in the reference, whose loop
    never ends) stops after that many steps; `log`: optional callable receiving (step, loss)."""
    m = SrcnnModel(flags, device=device, seed=seed)
    source = latest_checkpoint(flags.ckpt_dir_path)
    if source is not None:
        m.stack.load_tf_checkpoint(source)
    batches = dataset_reader(flags, seed)
    side = m.crop_side()
    done = 0
    while max_steps is None or done < max_steps:
        hd_full = torch.from_numpy(next(batches)).to(m.stack.device)
        sd_full = m.degrade(hd_full)
        hd = hd_full[:, side:hd_full.shape[1] - side, side:hd_full.shape[2] - side].contiguous()
        loss = m.train_step(sd_full, hd)
        step = m.stack.global_step
        done += 1
        if log is not None:
            log(step, loss.item())
        if step % 100 == 0:
            print('loss[{}]: {}'.format(step, loss.item()), flush=True)
        if step % flags.save_every == 0:
            os.makedirs(flags.ckpt_dir_path, exist_ok=True)
            m.stack.save_tf_checkpoint(os.path.join(flags.ckpt_dir_path, 'model.ckpt-%d' % step), adam_betas=(0.5, 0.9))
    return m


def super_resolution(flags, device='cuda', seed=None):
    """srcnn.py:263-287: restore the latest checkpoint, run one (randomly cropped / flipped) crop of --sr-source-path
    through the net and write the hd | sd | sr panel to --sr-target-path as JPEG, pixels
    saturate_cast((x + 1) * 127.5, uint8)."""
    from PIL import Image
    m = SrcnnModel(flags, device=device, seed=seed)
    source = latest_checkpoint(flags.ckpt_dir_path)
    if source is None:
        raise SystemExit('no checkpoint under %r' % (flags.ckpt_dir_path,))
    m.stack.load_tf_checkpoint(source, with_optimizer=False)
    hd_full = torch.from_numpy(next(dataset_reader(flags, seed))).to(m.stack.device)
    sd_full = m.degrade(hd_full)
    sr = m.forward(sd_full)
    panel = build_sr_result(m, hd_full, sd_full, sr)[0].cpu().numpy()
    px = np.clip((panel + np.float32(1.0)) * np.float32(127.5), 0, 255).astype(np.uint8)      # clamp, truncate
    Image.fromarray(px).save(flags.sr_target_path, format='JPEG')
    return px


def main(argv=None):
    flags = sanity_check(_flags().parse_args(argv))
    if flags.train:
        train(flags)
    else:
        super_resolution(flags)


if __name__ == '__main__':
    main()
