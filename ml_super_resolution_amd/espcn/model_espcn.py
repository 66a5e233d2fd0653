"""
model_espcn.py -- mirror of espcn/espcn/model_espcn.py (reference) on the MI355X engine.

  build_model(lr_source, scaling_factor=3, hr_target=None)
      -> {'lr_source', 'sr_result'} (+ 'hr_target', 'step', 'loss', 'optimizer', 'learning_rate')
  build_test_model(meta_path, ckpt_path) -> {'lr_sources', 'sr_results', 'scaling_factor'}
  extract_weights(meta_path, ckpt_path)  -> {'f1/kernel:0': ndarray, ...}
(reference: model_espcn.py:6,17-19,64,71,91-94,143-147,150-166).

Network (model_espcn.py:30-62 / :117-134): 5x5 conv -> 64 tanh, 3x3 -> 32 tanh, 3x3 -> 3*r*r linear,
all SAME.  `sr_result` stays in sub-pixel space [N,H,W,3*r*r], exactly like the reference; the
depth-to-space map (experiment_test.py:171-177) is either the separate op ops.depth_to_space or -- on the
inference path `super_resolve` -- the STORE MODE of the f3 layer (srx_conv_desc.subpixel_r): three launches,
no intermediate sub-pixel tensor, replayed as one HIP graph per input shape.
Checkpoints: the reference restores a TF bundle + .meta graph; here a checkpoint is an .npz of the
same variable names (`f1/kernel:0` ...), `meta_path` is accepted and ignored.
"""
import os

import numpy as np
import torch

from .. import graph, ops
from ..engine import ConvStack, LayerSpec, truncated_normal_


def layer_specs(scaling_factor=3):
    r2 = scaling_factor * scaling_factor
    return [LayerSpec(5, 3, 64, 'same', 'tanh', 'f1'),
            LayerSpec(3, 64, 32, 'same', 'tanh', 'f2'),
            LayerSpec(3, 32, 3 * r2, 'same', None, 'f3')]


class EspcnModel(object):
    def __init__(self, scaling_factor=3, device='cuda', seed=None):
        self.scaling_factor = scaling_factor
        self.stack = ConvStack(layer_specs(scaling_factor), device=device, residual=False, weight_decay=0.0)
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for i in range(3):
            truncated_normal_(self.stack.kernel(i), 0.02, gen)      # model_espcn.py:21; biases zero
        self.placeholders = {}
        # inference path: HIP-graph replay of the three launches (SRX_ESPCN_GRAPH=0: eager launches)
        self.use_graph = os.environ.get('SRX_ESPCN_GRAPH', '1') != '0'
        self._graphs = {}
        # one launch for the whole net while the problem is latency-bound (SRX_ESPCN_FUSED=0: never)
        self.use_single_launch = os.environ.get('SRX_ESPCN_FUSED', '1') != '0'
        # (measured, round 4 -- tiles of up to 16 x 16, LDS reads a block ahead of the MFMAs -- one launch against the three-launch
        # graph: 28 / 38 / 62 / 89 / 130 us against 67 / 78 / 79 / 102 / 137 us at 16 k / 29 k / 66 k / 90 k / 131 k LR pixels;
        # behind at 230 k: 213 against 181 us.  Round 3, 9 x 9 tiles: level at 46-58 k pixels, behind from 65 k on.)
        self.single_launch_max_pixels = int(os.environ.get('SRX_ESPCN_FUSED_MAX_PIXELS', '131072'))
        self.inference_path = ('<= %d LR pixels: ONE launch, the three layers chained through LDS per <= 16x16 tile with the '
                               'sub-pixel store (srx_espcn_forward); larger: f1, f2, f3 with the sub-pixel store fused into '
                               "f3's epilogue, 3 launches%s" % (self.single_launch_max_pixels,
                                                               ' replayed as one HIP graph' if self.use_graph else ', eager'))

        # the forward pass of a TRAIN step in one launch too (srx_espcn_forward_keep: writes t1, t2 and y for backward);
        # SRX_ESPCN_FUSED_TRAIN=0: three launches (A/B).  Measured (scripts/time_espcn_train.py, whole train step, one launch
        # against three): batch 16 / 32 / 64 of 17 x 17: 119 / 128 / 154 us against 149 / 144 / 169; batch 128 (37 k pixels): level.
        self.use_single_launch_train = os.environ.get('SRX_ESPCN_FUSED_TRAIN', '1') != '0'
        self.single_launch_train_max_pixels = int(os.environ.get('SRX_ESPCN_FUSED_TRAIN_MAX_PIXELS', '30000'))
        self.stack.forward_keep_hook = self._forward_keep_one_launch
        # train steps are replayed as a HIP graph for small batches only: measured (bench.py espcn_train_us, scripts/time_espcn_train.py)
        # batch 16 / 32 of 17 x 17: replay 119 / 128 us against 155 / 164 us of eager launches; batch 64: 155 against 148 -- the
        # launches of the larger problem keep the queue fed by themselves
        self.stack.step_graph_max_pixels = int(os.environ.get('SRX_ESPCN_STEP_GRAPH_MAX_PIXELS', '12000'))

    def _forward_keep_one_launch(self, x, outs):
        n, h, w, _ = x.shape
        if not (self.use_single_launch and self.use_single_launch_train and x.is_cuda and n * h * w <= self.single_launch_train_max_pixels
                and 2 <= self.scaling_factor <= 4):
            return False
        st = self.stack
        ops.espcn_forward_keep(x.contiguous(), [(st.kernel(i), st.bias(i)) for i in range(3)], self.scaling_factor, outs[0], outs[1], outs[2])
        return True

    # ---- eager API -----------------------------------------------------------------------------
    def forward(self, lr_source, keep=False):
        """sr_result in sub-pixel space [N,H,W,3*r*r]."""
        return self.stack.forward(lr_source, keep=keep)

    def super_resolve_two_step(self, lr_source):
        """forward + the standalone depth-to-space pass (4 launches): [N,H,W,3] -> [N,H*r,W*r,3]."""
        return ops.depth_to_space(self.forward(lr_source), self.scaling_factor)

    def _super_resolve_launches(self, lr_source, out=None, mids=None):
        """mids: the two intermediate tensors to use (a captured graph owns its own: the shared ones below are
        replaced when another image size comes along, and a replay must never write through a dangling pointer)."""
        st, r = self.stack, self.scaling_factor
        n, h, w, _ = lr_source.shape
        t1, t2 = mids if mids is not None else (st._buf(('sr', 0), (n, h, w, 64)), st._buf(('sr', 1), (n, h, w, 32)))
        t = ops.conv2d_fwd(lr_source, st.kernel(0), st.bias(0), 'same', 'tanh', out=t1)
        t = ops.conv2d_fwd(t, st.kernel(1), st.bias(1), 'same', 'tanh', out=t2)
        return ops.conv2d_fwd(t, st.kernel(2), st.bias(2), 'same', None, subpixel_r=r,
                              out=out if out is not None else st._buf(('sr', 2), (n, h * r, w * r, 3)))

    def super_resolve(self, lr_source, use_graph=None, single_launch=None):
        """The inference path (espcn/espcn/experiment_test.py:156-181: run the net, then the sub-pixel shuffle):
        [N,H,W,3] -> [N,H*r,W*r,3] in three launches -- the f3 layer stores straight through the depth-to-space map
        (bit-identical to super_resolve_two_step) -- replayed as ONE HIP graph per input shape: the problem is
        launch-latency-bound (0.57 GFLOP at BASELINE configs[1]).  Small problems (<= single_launch_max_pixels LR
        pixels) take ONE launch instead: srx_espcn_forward chains the three layers through LDS per <= 16x16 tile (same
        bits).  The returned tensor is a buffer owned by the model and overwritten by the next call."""
        if single_launch is None:
            single_launch = self.use_single_launch and lr_source.shape[0] * lr_source.shape[1] * lr_source.shape[2] <= self.single_launch_max_pixels
        if single_launch and lr_source.is_cuda:
            st, r = self.stack, self.scaling_factor
            n, h, w, _ = lr_source.shape
            return ops.espcn_forward(lr_source.contiguous(), [(st.kernel(i), st.bias(i)) for i in range(3)], r,
                                     out=st._buf(('sr1',), (n, h * r, w * r, 3)))
        if not (self.use_graph if use_graph is None else use_graph) or not lr_source.is_cuda:
            return self._super_resolve_launches(lr_source)
        key = tuple(lr_source.shape)
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = self._capture(lr_source)
        static_in, graph_obj, out = g[:3]
        if lr_source.data_ptr() != static_in.data_ptr():
            static_in.copy_(lr_source)
        graph_obj.replay()
        return out

    def _capture(self, lr_source):
        """Warm the kernels up on a side stream (function attributes and buffers must exist before capture),
        then record the three launches.  The graph reads the weights through their pointers: training steps or
        load_variables() that update them in place are seen by later replays."""
        if len(self._graphs) >= 8:                      # a directory of images of many sizes: keep the cache small
            self._graphs.pop(next(iter(self._graphs)))
        static_in = lr_source.clone()
        n, h, w, _ = lr_source.shape
        r = self.scaling_factor
        out = torch.empty((n, h * r, w * r, 3), dtype=torch.float32, device=lr_source.device)
        mids = (torch.empty((n, h, w, 64), dtype=torch.float32, device=lr_source.device),
                torch.empty((n, h, w, 32), dtype=torch.float32, device=lr_source.device))
        side = torch.cuda.Stream(device=lr_source.device)
        side.wait_stream(torch.cuda.current_stream(lr_source.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._super_resolve_launches(static_in, out=out, mids=mids)
        torch.cuda.current_stream(lr_source.device).wait_stream(side)
        graph_obj = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_obj):
            self._super_resolve_launches(static_in, out=out, mids=mids)
        return static_in, graph_obj, out, mids          # (the tuple keeps every buffer the graph points at alive)

    def train_step(self, lr_source, hr_target, learning_rate):
        """MSE in sub-pixel space (hr_target is the space-to-depth label, dataset.py:140-156) + Adam
        with the fed learning rate (model_espcn.py:76-89)."""
        # one replayed HIP graph per batch shape (engine.ConvStack.train_step_replay): a step is a dozen dependent launches
        return self.stack.train_step_replay(lr_source, hr_target, learning_rate)

    # ---- checkpoints ---------------------------------------------------------------------------
    def save(self, path):
        arrays = {k + ':0': v.detach().cpu().numpy() for k, v in self.stack.variables().items()}
        arrays['global_step'] = np.int64(self.stack.global_step)
        np.savez(path, **arrays)

    # ---- Session.run backend -------------------------------------------------------------------
    def run(self, keys, feed_dict):
        dev = self.stack.device
        feeds = {name: feed_dict[ph] for name, ph in self.placeholders.items() if ph in feed_dict}
        # `step` is a variable: the reference reads it with no feed (espcn/espcn/experiment_train.py:92)
        needs_forward = [k for k in keys if k != 'step']
        lr = hr = y = loss = None
        if needs_forward:
            if 'lr_source' not in feeds:
                raise ValueError('lr_source must be fed to fetch %s' % ', '.join(needs_forward))
            lr = graph.to_device(feeds['lr_source'], dev)
            hr = graph.to_device(feeds['hr_target'], dev) if 'hr_target' in feeds else None
            if 'optimizer' in keys:
                if hr is None or 'learning_rate' not in feeds:
                    raise ValueError('hr_target and learning_rate must be fed to run the optimizer')
                loss = self.train_step(lr, hr, float(feeds['learning_rate']))
                y = self.stack.acts[-1]
            else:
                y = self.stack.forward(lr, keep=True)
                if 'loss' in keys:
                    if hr is None:
                        raise ValueError('hr_target must be fed to fetch loss')
                    loss = self.stack.loss
                    ops.mse_fwd_bwd(y, hr, loss, accumulate=False, want_grad=False)
        out = {}
        for k in keys:
            if k == 'optimizer':
                out[k] = None
            elif k == 'loss':
                out[k] = float(loss.item())
            elif k == 'step':
                out[k] = self.stack.global_step
            elif k in ('sr_result', 'sr_results'):
                out[k] = y.detach().cpu().numpy()
            elif k in ('lr_source', 'lr_sources'):
                out[k] = lr.detach().cpu().numpy()
            elif k == 'hr_target':
                out[k] = hr.detach().cpu().numpy()
            else:
                raise KeyError(k)
        return out


def build_model(lr_source, scaling_factor=3, hr_target=None, device='cuda', seed=None):
    """
    lr_source:  source image batch (graph.placeholder) to be super resolved
    hr_target:  target batch as training labels in sub-pixel convolved shape; a partial (test)
                model is built if hr_target is None
    scaling_factor: factor of up scaling; the depth of the final layer is 3 * scaling_factor ** 2
    """
    m = EspcnModel(scaling_factor, device=device, seed=seed)
    m.placeholders['lr_source'] = lr_source
    model = {'lr_source': lr_source, '_model': m}
    model['sr_result'] = graph.Tensor('sr_result', owner=m, key='sr_result')
    if hr_target is None:
        return model
    m.placeholders['hr_target'] = hr_target
    lr = graph.placeholder([], name='learning_rate')         # tf.placeholder(shape=[]) (model_espcn.py:83)
    m.placeholders['learning_rate'] = lr
    model['hr_target'] = hr_target
    model['step'] = graph.Tensor('global_step', owner=m, key='step')
    model['loss'] = graph.Tensor('loss', owner=m, key='loss')
    model['optimizer'] = graph.Tensor('optimizer', owner=m, key='optimizer')
    model['learning_rate'] = lr
    return model


def extract_weights(meta_path, ckpt_path):
    """{variable name: ndarray} for every trainable variable (model_espcn.py:150-166).  `ckpt_path` is either a
    TensorFlow V2 checkpoint prefix (read by tf_bundle, no TensorFlow needed; `meta_path` is not used: the
    trainable variables are the `f{1,2,3}/{kernel,bias}` entries) or an .npz written by EspcnModel.save."""
    from .. import tf_bundle
    if tf_bundle.is_checkpoint_prefix(ckpt_path):
        values = tf_bundle.load_checkpoint(ckpt_path)
        return {k + ':0': v for k, v in values.items()
                if k.count('/') == 1 and k.split('/')[1] in ('kernel', 'bias')}
    z = np.load(ckpt_path)
    return {k: z[k] for k in z.files if k.endswith(':0')}


def build_test_model(meta_path, ckpt_path, device='cuda'):
    """Weights-as-constants inference model with dynamic image size (model_espcn.py:99-147).
    `ckpt_path` may also be an already-loaded {name: array} dict."""
    variables = ckpt_path if isinstance(ckpt_path, dict) else extract_weights(meta_path, ckpt_path)
    scaling_factor = int((np.asarray(variables['f3/bias:0']).size // 3) ** 0.5)      # model_espcn.py:108
    m = EspcnModel(scaling_factor, device=device)
    m.stack.load_variables(variables)
    lr_sources = graph.placeholder([None, None, None, 3], name='lr_sources')
    m.placeholders['lr_source'] = lr_sources
    return {'lr_sources': lr_sources,
            'sr_results': graph.Tensor('sr_results', owner=m, key='sr_results'),
            'scaling_factor': scaling_factor,
            '_model': m}
