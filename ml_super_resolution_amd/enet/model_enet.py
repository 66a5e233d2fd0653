"""
model_enet.py -- the EnhanceNet GENERATOR of enet/enet/model_enet.py (reference) on the MI355X
engine: forward, and the backward + Adam update of the generator variables for a given gradient on
sr_images (the discriminator and the VGG-19 perceptual / texture losses that
produce that gradient in the reference are rows A14 / N4 and not built).

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
