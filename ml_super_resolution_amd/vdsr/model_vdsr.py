"""
model_vdsr.py -- mirror of vdsr/vdsr/model_vdsr.py (reference) on the MI355X engine.

`build_model(sd_images, hd_images=None, num_layers=20, use_adam=False)` keeps the reference's
signature, defaults and result-dict keys (model_vdsr.py:6,72,76,101,108-109,186-190):
  conv.1..conv.N, relu.1..relu.N-1, sd_images, sr_images
  (+ step, loss, trainer, hd_images, learning_rate when hd_images is given).
The values are `graph.Tensor` handles to be fetched with `graph.Session().run(...)`, exactly as
the reference fetches `tf.Tensor`s; `model['_model']` is the eager object for callers that keep
their tensors on the GPU (bench.py, the data-parallel trainer).
"""
import os

import torch

from .. import graph
from ..engine import ConvStack, LayerSpec, xavier_uniform_


def layer_specs(num_layers=20, channels=3, width=64):
    """(N-1) x [3x3 conv -> 64, bias, ReLU], then 3x3 conv -> 3 (model_vdsr.py:47-93).
    Variable scopes follow tf.layers' defaults: conv2d, conv2d_1, ..."""
    specs = []
    cin = channels
    for i in range(num_layers - 1):
        specs.append(LayerSpec(3, cin, width, 'same', 'relu', 'conv2d' if i == 0 else 'conv2d_%d' % i))
        cin = width
    specs.append(LayerSpec(3, cin, channels, 'same', None, 'conv2d_%d' % (num_layers - 1)))
    return specs


class VdsrModel(object):
    def __init__(self, num_layers=20, use_adam=False, device='cuda', seed=None):
        self.num_layers = num_layers
        self.use_adam = use_adam
        # l2_regularizer(0.0001) on every kernel (model_vdsr.py:34,70,93); residual sr = sd + conv.N (:104)
        self.stack = ConvStack(layer_specs(num_layers), device=device, residual=True, weight_decay=1e-4)
        self.learning_rate = 0.1          # tf.get_variable('learning_rate', init 0.1) (model_vdsr.py:136-141)
        self.step_graph = os.environ.get('SRX_VDSR_STEP_GRAPH', '0') == '1'
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for i in range(num_layers):
            xavier_uniform_(self.stack.kernel(i), gen)      # xavier_initializer (:27); biases zero
        self.placeholders = {}

    # ---- eager API (device tensors in, device tensors out) -----------------------------------
    def forward(self, sd_images, keep=False):
        """sr_images = sd_images + conv stack (model_vdsr.py:47-106)."""
        return self.stack.forward(sd_images, keep=keep)

    def train_step(self, sd_images, hd_images, learning_rate=None):
        """One `session.run(trainer)`: forward, loss (MSE + L2), backward, optimizer, step += 1.
        Returns the device scalar holding the loss of THIS forward (pre-update weights)."""
        lr = self.learning_rate if learning_rate is None else learning_rate
        if self.step_graph:
            # SRX_VDSR_STEP_GRAPH=1: the whole step as one replayed HIP graph per batch shape (engine.ConvStack.train_step_replay).
            # Off by default: measured on MI355X the step is bound by the GPU side of its ~100 dependent launches at every
            # batch size (batch 16 / 64 / 256: 1.94 / 4.56 / 14.09 ms replayed against 1.91 / 4.53 / 13.97 ms eager), so the
            # replay saves nothing and costs two input copies.
            if self.use_adam:
                return self.stack.train_step_replay(sd_images, hd_images, lr)                          # model_vdsr.py:145-148
            return self.stack.train_step_replay(sd_images, hd_images, lr, momentum=0.9, gradient_cap=0.01)   # model_vdsr.py:158-184
        self.stack.forward(sd_images, keep=True)
        loss = self.stack.loss_and_backward(hd_images)
        if self.use_adam:
            self.stack.adam_step(lr)                                   # model_vdsr.py:145-148
        else:
            self.stack.momentum_clip_step(lr, 0.9, gradient_cap=0.01)  # model_vdsr.py:158-184
        return loss

    def taps(self):
        """conv.i / relu.i are the same post-ReLU tensor (tf.layers.conv2d already applied the
        ReLU; the second tf.nn.relu is idempotent: model_vdsr.py:62-76)."""
        acts = self.stack.acts
        out = {}
        for i in range(self.num_layers - 1):
            out['conv.%d' % (i + 1)] = acts[i + 1]
            out['relu.%d' % (i + 1)] = acts[i + 1]
        return out

    # ---- Session.run backend -----------------------------------------------------------------
    def run(self, keys, feed_dict):
        from .. import ops
        dev = self.stack.device
        feeds = {}
        for name, ph in self.placeholders.items():
            if ph in feed_dict:
                feeds[name] = feed_dict[ph]
        lr = feeds.get('learning_rate', self.learning_rate)
        # `step` and `learning_rate` are variables: the reference reads them with no feed at all
        # (step = session.run(model['step']), vdsr/vdsr/experiment_train.py:126).  Only fetches that need a
        # forward pass need the image feeds.
        needs_forward = [k for k in keys if k not in ('step', 'learning_rate')]
        sd = hd = sr = loss = None
        if needs_forward:
            if 'sd_images' not in feeds:
                raise ValueError('sd_images must be fed to fetch %s' % ', '.join(needs_forward))
            sd = graph.to_device(feeds['sd_images'], dev)
            hd = graph.to_device(feeds['hd_images'], dev) if 'hd_images' in feeds else None
            want_train = 'trainer' in keys
            want_loss = 'loss' in keys
            if (want_train or want_loss or 'hd_images' in keys) and hd is None:
                raise ValueError('hd_images must be fed to fetch loss / trainer / hd_images')
            if want_train:
                loss = self.train_step(sd, hd, float(lr))
                sr = self.stack.acts[-1]
            else:
                sr = self.stack.forward(sd, keep=True)
                if want_loss:
                    loss = self.stack.loss
                    ops.mse_fwd_bwd(sr, hd, loss, accumulate=False, want_grad=False)
                    self.stack.add_regulariser_loss()
        out = {}
        taps = None
        for k in keys:
            if k == 'trainer':
                out[k] = None
            elif k == 'loss':
                out[k] = float(loss.item())
            elif k == 'step':
                out[k] = self.stack.global_step
            elif k == 'learning_rate':
                out[k] = float(lr)
            elif k == 'sr_images':
                out[k] = sr.detach().cpu().numpy()
            elif k == 'sd_images':
                out[k] = sd.detach().cpu().numpy()
            elif k == 'hd_images':
                out[k] = hd.detach().cpu().numpy()
            elif k == 'conv.%d' % self.num_layers:
                out[k] = (sr - sd).detach().cpu().numpy()      # residual = conv.N (model_vdsr.py:101-104)
            else:
                taps = taps or self.taps()
                out[k] = taps[k].detach().cpu().numpy()
        return out


def build_model(sd_images, hd_images=None, num_layers=20, use_adam=False, device='cuda', seed=None):
    """
    sd_images: lo resolution images to be super resolved (a graph.placeholder)
    hd_images: hi resolution images as ground truth (a graph.placeholder) or None
    num_layers: num of conv_relu layers
    """
    m = VdsrModel(num_layers=num_layers, use_adam=use_adam, device=device, seed=seed)
    model = {}
    for i in range(num_layers - 1):
        for kind in ('conv', 'relu'):
            name = '%s.%d' % (kind, i + 1)
            model[name] = graph.Tensor(name, owner=m, key=name)
    name = 'conv.%d' % num_layers
    model[name] = graph.Tensor(name, owner=m, key=name)
    m.placeholders['sd_images'] = sd_images
    model['sd_images'] = sd_images
    model['sr_images'] = graph.Tensor('sr_images', owner=m, key='sr_images')
    model['_model'] = m
    if hd_images is None:
        return model
    m.placeholders['hd_images'] = hd_images
    lr = graph.Tensor('learning_rate', owner=m, key='learning_rate')
    m.placeholders['learning_rate'] = lr        # the reference FEEDS this variable (experiment_train.py:137)
    model['step'] = graph.Tensor('global_step', owner=m, key='step')
    model['loss'] = graph.Tensor('loss', owner=m, key='loss')
    model['trainer'] = graph.Tensor('trainer', owner=m, key='trainer')
    model['hd_images'] = hd_images
    model['learning_rate'] = lr
    return model
