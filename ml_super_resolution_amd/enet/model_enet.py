"""
model_enet.py -- the EnhanceNet GENERATOR of enet/enet/model_enet.py (reference) on the MI355X
engine: forward only (the discriminator, VGG-19 perceptual / texture losses and
the GAN training loop are rows A14 / N4 and out of scope here).

  3x3 conv 3->64 ReLU                                   (model_enet.py:63-70)
  10 x residual_block: 3x3 ReLU -> 1x1 -> relu(x + .)   (:8-31, :74-75)
  2 x [nearest-neighbour upsample x2 -> 3x3 conv ReLU]  (:78-89)
  3x3 conv ReLU, 3x3 conv -> 3, + bq_images             (:92-113)

`build_enet(sd_images, bq_images, hd_images=None, ...)` keeps the reference's inference keys
{'sd_images', 'bq_images', 'sr_images'} (:264-350 with hd_images None).  Variables live under the
`g_` scope with tf.layers' default names (g_/conv2d, g_/conv2d_1, ...), as `build_enet` selects
generator parameters by that prefix (:331-332).
"""
import torch

from .. import graph, ops
from ..engine import truncated_normal_


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
        self.kernels, self.biases = [], []
        for k, cin, cout in generator_layers():
            w = torch.empty((k, k, cin, cout), dtype=torch.float32, device=self.device)
            truncated_normal_(w, 0.02, gen)                               # model_enet.py:10,46
            self.kernels.append(w)
            self.biases.append(torch.zeros(cout, dtype=torch.float32, device=self.device))
        self.placeholders = {}

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

    def forward(self, sd_images, bq_images):
        """sd_images [N,h,w,3], bq_images [N,4h,4w,3] (bicubic-upscaled) -> sr_images [N,4h,4w,3]."""
        K, B = self.kernels, self.biases
        t = ops.conv2d_fwd(sd_images, K[0], B[0], 'same', 'relu')
        i = 1
        for _ in range(10):
            x = ops.conv2d_fwd(t, K[i], B[i], 'same', 'relu')
            # 1x1 conv, + block input, ReLU -- one launch
            t = ops.conv2d_fwd(x, K[i + 1], B[i + 1], 'same', None, skip=t, post_add_relu=True)
            i += 2
        for _ in range(2):
            t = ops.upsample_nearest(t, 2)       # resize_nearest_neighbor to 2h then 4h = two doublings
            t = ops.conv2d_fwd(t, K[i], B[i], 'same', 'relu')
            i += 1
        t = ops.conv2d_fwd(t, K[i], B[i], 'same', 'relu')
        return ops.conv2d_fwd(t, K[i + 1], B[i + 1], 'same', None, skip=bq_images)

    def run(self, keys, feed_dict):
        feeds = {name: feed_dict[ph] for name, ph in self.placeholders.items() if ph in feed_dict}
        sd = graph.to_device(feeds['sd_images'], self.device)
        bq = graph.to_device(feeds['bq_images'], self.device)
        sr = self.forward(sd, bq)
        vals = {'sr_images': sr, 'sd_images': sd, 'bq_images': bq}
        return {k: vals[k].detach().cpu().numpy() for k in keys}


def build_enet(sd_images, bq_images, hd_images=None, pat_model=True, vgg19_path=None, device='cuda', seed=None):
    """Inference graph of enet/enet/model_enet.py:264-350 (hd_images must be None: training the
    generator needs the discriminator / VGG-19 losses, which are out of scope)."""
    if hd_images is not None:
        raise NotImplementedError('EnhanceNet training (discriminator, VGG-19 perceptual / texture losses) is '
                                  'outside the ported hot path; only the generator forward is provided')
    g = EnetGenerator(device=device, seed=seed)
    g.placeholders['sd_images'] = sd_images
    g.placeholders['bq_images'] = bq_images
    return {'sd_images': sd_images, 'bq_images': bq_images,
            'sr_images': graph.Tensor('sr_images', owner=g, key='sr_images'), '_model': g}
