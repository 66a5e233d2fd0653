"""
engine.py -- the layer-stack engine behind the model mirrors.

A `ConvStack` is what the reference's `build_model` functions describe: a chain of stride-1
convolutions with bias + activation (vdsr/vdsr/model_vdsr.py:47-106,
espcn/espcn/model_espcn.py:30-62, srcnn/srcnn.py:100-130), an MSE loss with optional L2
regulariser (model_vdsr.py:120-125), and a TF-semantics Adam / Momentum+clip update
(model_vdsr.py:145-184).  TensorFlow's autodiff is replaced by an explicit backward over the
C-ABI ops: per layer one fused dgrad (+ upstream activation gradient) and one wgrad
(+ bias grad + weight decay).

Memory layout (sized for 288 GB HBM: everything stays resident):
  * all kernels and biases live in ONE flat fp32 buffer (`params`), each tensor starting on a
    16-byte boundary; gradients, Adam m/v (or momentum) mirror that layout, so the optimizer is
    one launch and the data-parallel exchange is ONE all-reduce of `grads`;
  * every layer's post-activation output is kept for backward (VDSR-20 @ 256x41x41: 2.1 GB);
  * two ping-pong buffers hold the running pre-activation gradient.
"""
import math
import os

import torch

from . import ops


class LayerSpec(object):
    """One conv layer.  `scope` is the TF variable scope (`conv2d`, `conv2d_1`, `f1`, ...):
    variables are `<scope>/kernel` and `<scope>/bias`."""

    def __init__(self, ksize, cin, cout, padding='same', act=None, scope=None):
        self.kh, self.kw = (ksize, ksize) if isinstance(ksize, int) else ksize
        self.cin, self.cout = cin, cout
        self.padding = padding.lower()
        self.act = act
        self.scope = scope

    @property
    def kernel_shape(self):
        return (self.kh, self.kw, self.cin, self.cout)

    def out_hw(self, h, w):
        if self.padding == 'same':
            return h, w
        return h - self.kh + 1, w - self.kw + 1


def _align4(n):
    return (n + 3) // 4 * 4


class ConvStack(object):
    def __init__(self, specs, device='cuda', residual=False, weight_decay=0.0):
        """residual: output = input + stack(input) (VDSR, model_vdsr.py:104).
        weight_decay: scale of tf.contrib.layers.l2_regularizer on every kernel."""
        self.specs = list(specs)
        self.device = torch.device(device)
        self.residual = residual
        self.weight_decay = float(weight_decay)
        # ---- flat parameter layout
        self.slices = []          # per layer: (k_off, k_len, b_off, b_len)
        off = 0
        for s in self.specs:
            kn = s.kh * s.kw * s.cin * s.cout
            k_off = off
            off = _align4(off + kn)
            b_off = off
            off = _align4(off + s.cout)
            self.slices.append((k_off, kn, b_off, s.cout))
        self.flat_size = off
        self.num_params = sum(kn + bn for _, kn, _, bn in self.slices)
        z = lambda: torch.zeros(self.flat_size, dtype=torch.float32, device=self.device)
        self.params, self.grads = z(), z()
        self.opt_m, self.opt_v = None, None     # Adam slots / momentum accumulator (lazy)
        self.global_step = 0
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._acts = None
        self.step_graph_max_pixels = None  # optional: input pixels (N*H*W) above which train_step_replay issues eager launches
        self.forward_keep_hook = None      # optional: fn(x, [per-layer output buffers]) -> True when it ran the whole forward pass
        self._bufs = {}
        self._ws = None
        self.grad_hook = None      # called with the flat gradient after backward (DP all-reduce)
        self.loss_kind = 'mse'     # 'mse' (VDSR, ESPCN) or 'rownorm' (SRCNN)
        self._decay_mask = None
        self._side = None
        self.overlap_wgrad = os.environ.get('SRX_OVERLAP_WGRAD', '0') != '0'   # wgrads on a side stream (see loss_and_backward; measured 1 % slower: off)
        self.overlap_reduce = os.environ.get('SRX_OVERLAP_REDUCE', '0') != '0'  # partial-filter reductions on a side stream (see loss_and_backward)
        self._ws2 = None
        # variable leaf names: tf.layers.conv2d creates <scope>/kernel, <scope>/bias; tf.contrib.layers.convolution2d
        # (SRCNN, srcnn/srcnn.py:100-130) creates <scope>/weights, <scope>/biases
        self.kernel_name, self.bias_name = 'kernel', 'bias'
        # ---- Adam with its step count and learning rate in device memory, and whole train steps replayed as HIP graphs
        # (train_step_replay): ESPCN / SRCNN steps are a dozen dependent launches of ~10 us each, their recipes run 1.6 M
        # of them (espcn/makefile:30-36).  SRX_STEP_GRAPH=0: every step as eager launches.
        self.use_step_graph = os.environ.get('SRX_STEP_GRAPH', '1') != '0'
        self._adam_state = None     # ops.adam_state(): {int64 t; float lr; ...} on the device
        self._state_t, self._state_lr = None, None    # host mirror of what the device block holds
        self._step_graphs = {}

    # ---- parameter views -------------------------------------------------------------------
    def kernel(self, i, buf=None):
        k_off, kn, _, _ = self.slices[i]
        return (self.params if buf is None else buf)[k_off:k_off + kn].view(self.specs[i].kernel_shape)

    def bias(self, i, buf=None):
        _, _, b_off, bn = self.slices[i]
        return (self.params if buf is None else buf)[b_off:b_off + bn]

    def variables(self):
        """{tf variable name: tensor view}, e.g. 'conv2d_3/kernel'."""
        out = {}
        for i, s in enumerate(self.specs):
            scope = s.scope or ('layer%d' % i)
            out[scope + '/' + self.kernel_name] = self.kernel(i)
            out[scope + '/' + self.bias_name] = self.bias(i)
        return out

    def load_variables(self, values):
        """values: {name: array-like}; names as in variables() (a trailing ':0' is accepted)."""
        mine = self.variables()
        for name, val in values.items():
            key = name[:-2] if name.endswith(':0') else name
            if key not in mine:
                raise KeyError('unknown variable %r' % name)
            t = torch.as_tensor(val, dtype=torch.float32)
            if tuple(t.shape) != tuple(mine[key].shape):
                raise ValueError('%s: shape %s != %s' % (name, tuple(t.shape), tuple(mine[key].shape)))
            mine[key].copy_(t.to(self.device))

    def set_params(self, pairs):
        """pairs: [(kernel, bias)] per layer (numpy or torch)."""
        for i, (k, b) in enumerate(pairs):
            self.kernel(i).copy_(torch.as_tensor(k, dtype=torch.float32).to(self.device))
            self.bias(i).copy_(torch.as_tensor(b, dtype=torch.float32).to(self.device))

    # ---- buffers ------------------------------------------------------------------------------
    def _buf(self, key, shape):
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            if t is not None:
                self._step_graphs.clear()      # a captured step points at the buffer that is about to be released
            t = torch.empty(shape, dtype=torch.float32, device=self.device)
            self._bufs[key] = t
        return t

    def _shapes(self, x_shape):
        n, h, w, _ = x_shape
        shapes = []
        for s in self.specs:
            h, w = s.out_hw(h, w)
            shapes.append((n, h, w, s.cout))
        return shapes

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, keep=True):
        """Returns the stack output; with keep=True every layer's output is kept for backward()
        and exposed through `self.acts` (acts[0] is the input, acts[i+1] layer i's output)."""
        shapes = self._shapes(x.shape)
        if keep and self.forward_keep_hook is not None:
            # a model's one-launch forward pass (ESPCN: srx_espcn_forward_keep) writing the same activation buffers
            outs = [self._buf(('act', i), shapes[i]) for i in range(len(self.specs))]
            if self.forward_keep_hook(x, outs):
                self._acts = self.acts = [x] + outs
                return outs[-1]
        acts = [x]
        t = x
        last = len(self.specs) - 1
        for i, s in enumerate(self.specs):
            skip = x if (self.residual and i == last) else None
            if keep:
                out = self._buf(('act', i), shapes[i])
            else:
                # keyed by parity and channel count, not by the full shape: a new image size replaces the old
                # buffer instead of piling up one pair per size (a directory of images of many sizes)
                out = self._buf(('tmp', i & 1, s.cout), shapes[i])
            t = ops.conv2d_fwd(t, self.kernel(i), self.bias(i), s.padding, s.act, skip=skip, out=out)
            acts.append(t)
        self._acts = acts if keep else None
        self.acts = acts
        return t

    # ---- loss + backward ----------------------------------------------------------------------
    def loss_and_backward(self, target, numel_global=None):
        """MSE(output, target) [+ weight decay terms] into self.loss (device scalar) and the full
        gradient into self.grads.  With data parallelism pass nothing: every rank uses its LOCAL
        mean and the hook averages the gradients (identical to the global mean for equal shards;
        the regulariser gradient is the same on every rank, so averaging preserves it)."""
        if self._acts is None:
            raise RuntimeError('forward(keep=True) must run before loss_and_backward')
        acts = self._acts
        y = acts[-1]
        last = len(self.specs) - 1
        inv = 1.0 / (y.numel() if numel_global is None else numel_global)
        if self.loss_kind == 'rownorm':
            # srcnn/srcnn.py:142-144: mean over rows of ||reshape(diff, [-1, bb*bb])||_2
            row_len = y.shape[1] * y.shape[2]
            dy = ops.rownorm_loss_fwd_bwd(y, target, row_len, self.loss, dpred=self._buf(('dy', 0), y.shape),
                                          norms=self._buf(('rownorms',), (y.numel() // row_len,)))
        else:
            dy = self._buf(('dy', 0), y.shape)
            ops.mse_fwd_bwd(y, target, self.loss, inv_numel=inv, accumulate=False, dpred=dy)
        if self.weight_decay:
            self.add_regulariser_loss()
        if self.specs[last].act is not None:
            dpre = ops.act_bwd(dy, y, self.specs[last].act, out=self._buf(('dpre_last',), y.shape))
        else:
            dpre = dy
        # the number of partial filters grows with batch and image size: re-query on every call (host-only) and
        # grow the workspace when a larger feed arrives
        need = max(ops.bwd_filter_workspace_bytes(acts[i].shape, s.kernel_shape, s.padding)
                   for i, s in enumerate(self.specs))
        if self._ws is None or self._ws.numel() * 4 < need:
            self._step_graphs.clear()
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
            self._ws2 = None
        # dgrad of layer i and wgrad (+ partial reduce) of layer i are independent: both only read dpre_i.  With
        # SRX_OVERLAP_WGRAD=1 the wgrads run on a side stream, so that their prologues, the reduce kernels and the ragged
        # ends of the launches could overlap the dgrad chain.  Measured on MI355X: 14.59 ms per step against 14.45 ms
        # in one stream (two kernels of one persistent 160-KiB-LDS workgroup per CU do not interleave well), so it is
        # off by default.  The running pre-activation gradient rotates over three buffers either way.
        main = torch.cuda.current_stream(self.device) if self.device.type == 'cuda' else None
        two = self.overlap_wgrad and main is not None
        if two and self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        # SRX_OVERLAP_REDUCE=1: only the HBM-bound reduction of the per-workgroup partial filters (9 us per layer)
        # goes to the side stream, where it could run under the next MFMA-bound dgrad; two workspaces alternate.
        # Measured on MI355X (two A/B pairs): 14.16-14.19 ms per step against 14.04-14.11 ms in one stream -- the
        # reduction does not hide (the dgrad's one persistent workgroup per CU leaves it no registers to run beside)
        # and the cross-stream waits cost more than the 160 us at stake.  Off by default.
        red = self.overlap_reduce and main is not None and not two
        if red:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            if self._ws2 is None or self._ws2.numel() != self._ws.numel():
                self._ws2 = torch.empty_like(self._ws)
        ws_free = [None, None]   # event: the reduction reading workspace k has finished
        read_done = {}       # id of a rotating buffer -> event: the wgrad reading it has finished
        for i in range(last, -1, -1):
            s = self.specs[i]
            if red:
                k = i & 1
                ws = self._ws if k == 0 else self._ws2
                if ws_free[k] is not None:
                    main.wait_event(ws_free[k])
                n_part = ops.conv2d_bwd_filter_partials(acts[i], dpre, s.kernel_shape, s.padding, ws)
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(self._side):
                    self._side.wait_event(ready)
                    ops.conv2d_bwd_filter_reduce(acts[i].shape, s.kernel_shape, s.padding, ws, n_part,
                                                 self.kernel(i, self.grads), self.bias(i, self.grads),
                                                 w_for_decay=self.kernel(i) if self.weight_decay else None,
                                                 wd_scale=self.weight_decay)
                    ws_free[k] = torch.cuda.Event()
                    ws_free[k].record(self._side)
            elif two:
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(self._side):
                    self._side.wait_event(ready)
                    ops.conv2d_bwd_filter(acts[i], dpre, s.kernel_shape, s.padding,
                                          w_for_decay=self.kernel(i) if self.weight_decay else None,
                                          wd_scale=self.weight_decay, dw=self.kernel(i, self.grads),
                                          dbias=self.bias(i, self.grads), workspace=self._ws)
                    done = torch.cuda.Event()
                    done.record(self._side)
                read_done[dpre.data_ptr()] = done
            else:
                ops.conv2d_bwd_filter(acts[i], dpre, s.kernel_shape, s.padding,
                                      w_for_decay=self.kernel(i) if self.weight_decay else None,
                                      wd_scale=self.weight_decay, dw=self.kernel(i, self.grads),
                                      dbias=self.bias(i, self.grads), workspace=self._ws)
            if i > 0:
                prev_act = self.specs[i - 1].act
                out = self._buf(('dx', i % 3, acts[i].shape[3]), acts[i].shape)
                if two and out.data_ptr() in read_done:
                    main.wait_event(read_done.pop(out.data_ptr()))
                dpre = ops.conv2d_bwd_data(dpre, self.kernel(i), acts[i].shape, s.padding,
                                           x_in=acts[i] if prev_act is not None else None, in_act=prev_act, out=out)
        if two or red:
            main.wait_stream(self._side)
        if self.grad_hook is not None:
            self.grad_hook(self.grads)
        return self.loss

    def add_regulariser_loss(self):
        """self.loss += weight_decay * sum over kernels of sum(w^2)/2 -- one launch over the flat buffer."""
        if self._decay_mask is None:
            self._decay_mask = torch.zeros_like(self.params)
            for k_off, kn, _, _ in self.slices:
                self._decay_mask[k_off:k_off + kn] = 1.0
        ops.l2_loss(self.params, self.weight_decay, self.loss, accumulate=True, mask=self._decay_mask)

    # ---- optimizers ---------------------------------------------------------------------------
    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        """tf.train.AdamOptimizer(...).minimize(loss, global_step)."""
        if self.opt_m is None:
            self.opt_m = torch.zeros_like(self.params)
            self.opt_v = torch.zeros_like(self.params)
        self.global_step += 1
        ops.adam_tf_step(self.params, self.grads, self.opt_m, self.opt_v, lr, self.global_step, beta1, beta2, eps)

    # ---- the same with the step count and the learning rate in device memory -------------------
    def _sync_adam_state(self, lr):
        """The device block {t, lr} follows the host's `global_step` (a checkpoint load, dist.attach) and the fed rate;
        written only when one of them changed."""
        if self._adam_state is None:
            self._adam_state = ops.adam_state(self.device, t=self.global_step, lr=lr)
            self._state_t, self._state_lr = self.global_step, float(lr)
        if self._state_t != self.global_step:
            ops.adam_state_set(self._adam_state, t=self.global_step)
            self._state_t = self.global_step
        if self._state_lr != float(lr):
            ops.adam_state_set(self._adam_state, lr=lr)
            self._state_lr = float(lr)

    def adam_step_dev(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        """adam_step through srx_adam_tf_step_dev: no per-step kernel argument, so the launch can sit in a replayed graph."""
        if self.opt_m is None:
            self.opt_m = torch.zeros_like(self.params)
            self.opt_v = torch.zeros_like(self.params)
        self._sync_adam_state(lr)
        ops.adam_tf_step_dev(self.params, self.grads, self.opt_m, self.opt_v, self._adam_state, beta1, beta2, eps)
        self.global_step += 1
        self._state_t += 1

    def train_step_replay(self, x, target, lr, beta1=0.9, beta2=0.999, eps=1e-8, momentum=None, gradient_cap=0.01):
        """forward + loss_and_backward + optimizer as ONE replayed HIP graph per (input shape, target shape, optimizer
        constants).  momentum=None: TF-Adam with the step count and the learning rate in device memory
        (srx_adam_tf_step_dev) -- a changing rate needs no new graph; momentum=m: Momentum on clipped gradients
        (momentum_clip_step), whose rate is a kernel argument: the graph is keyed by it (VDSR's schedule changes it every
        few thousand steps).  The first two steps of a shape run as eager launches (they are real steps: buffers, workspaces
        and kernel attributes come into being), the third is captured -- capturing executes nothing -- and replayed, like
        every step after it.  Eager and replayed steps issue the same kernels with the same arguments: the weights are
        bit-identical either way (tests/test_gpu_models.py).  Falls back to eager launches when a gradient hook (data
        parallelism) is attached or SRX_STEP_GRAPH=0.  Returns the device scalar of the loss."""
        def optimizer_launch():
            if momentum is None:
                ops.adam_tf_step_dev(self.params, self.grads, self.opt_m, self.opt_v, self._adam_state, beta1, beta2, eps)
            else:
                ops.momentum_clip_step(self.params, self.grads, self.opt_m, lr, momentum, gradient_cap / lr)

        def eager(xx, tt):
            self.forward(xx, keep=True)
            self.loss_and_backward(tt)
            if momentum is None:
                self.adam_step_dev(lr, beta1, beta2, eps)
            else:
                self.momentum_clip_step(lr, momentum, gradient_cap)
            return self.loss
        if not self.use_step_graph or self.grad_hook is not None or self.device.type != 'cuda' or self.overlap_reduce:
            return eager(x, target)
        if self.step_graph_max_pixels is not None and x.shape[0] * x.shape[1] * x.shape[2] > self.step_graph_max_pixels:
            return eager(x, target)          # (measured per model: beyond this size the replay is no faster than the launches)
        opt_key = (float(beta1), float(beta2), float(eps)) if momentum is None else ('momentum', float(lr), float(momentum), float(gradient_cap))
        key = (tuple(x.shape), tuple(target.shape)) + opt_key
        ent = self._step_graphs.get(key)
        if ent is None:
            if len(self._step_graphs) >= 4:
                self._step_graphs.clear()
            ent = self._step_graphs[key] = {'warm': 0, 'graph': None}
        if ent['graph'] is None:
            if ent['warm'] < 2:
                ent['warm'] += 1
                loss = eager(x, target)
                if key not in self._step_graphs:         # (the first step of a new shape replaced buffers: start over, warm)
                    self._step_graphs[key] = ent
                return loss
            if self.opt_m is None:
                self.opt_m = torch.zeros_like(self.params)
                if momentum is None:
                    self.opt_v = torch.zeros_like(self.params)
            if momentum is None:
                self._sync_adam_state(lr)
            sx, st = x.clone(), target.clone()
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.forward(sx, keep=True)
                self.loss_and_backward(st)
                optimizer_launch()
            ent.update(graph=g, x=sx, t=st)
        if momentum is None:
            self._sync_adam_state(lr)
        if x.data_ptr() != ent['x'].data_ptr():
            ent['x'].copy_(x)
        if target.data_ptr() != ent['t'].data_ptr():
            ent['t'].copy_(target)
        ent['graph'].replay()
        self.global_step += 1
        if momentum is None:
            self._state_t += 1
        return self.loss

    def static_step_inputs(self, x_shape, target_shape):
        """The input / target tensors a captured train step of these shapes reads (None before it exists): a training loop
        that writes its batches straight into them (the host-to-device copy it makes anyway) and passes them to
        train_step_replay replays without the device-side copies."""
        for key, ent in self._step_graphs.items():
            if ent.get('graph') is not None and key[0] == tuple(x_shape) and key[1] == tuple(target_shape):
                return ent['x'], ent['t']
        return None

    def momentum_clip_step(self, lr, momentum=0.9, gradient_cap=0.01):
        """model_vdsr.py:158-184: clip every gradient element to +-gradient_cap/lr, then Momentum."""
        if self.opt_m is None:
            self.opt_m = torch.zeros_like(self.params)
        self.global_step += 1
        ops.momentum_clip_step(self.params, self.grads, self.opt_m, lr, momentum, gradient_cap / lr)

    # ---- checkpoint (own format, TF variable names) -----------------------------------------
    def state_dict(self):
        sd = {'global_step': self.global_step, 'params': self.params.detach().cpu()}
        if self.opt_m is not None:
            sd['opt_m'] = self.opt_m.detach().cpu()
        if self.opt_v is not None:
            sd['opt_v'] = self.opt_v.detach().cpu()
        return sd

    # ---- TensorFlow V2 checkpoints (tf.train.Saver format; tf_bundle.py) --------------------------
    def tf_checkpoint_tensors(self, adam_betas=(0.9, 0.999), extra=None):
        """{checkpoint key: ndarray} as tf.train.Saver would write this model: `<scope>/kernel`, `<scope>/bias`,
        `global_step` (int64) and, when an optimizer has run, its slots: Adam `<var>/Adam`, `<var>/Adam_1`,
        `beta1_power`, `beta2_power`; Momentum `<var>/Momentum`.  `extra`: further global variables of the
        reference graph ({name: array}; VDSR's non-trainable `learning_rate`, vdsr/vdsr/model_vdsr.py:136-141 --
        Saver().restore needs every global variable to be present).
        beta powers: TF-1.x AdamOptimizer creates them with the value beta and multiplies them by beta once per
        apply (_finish), so after N steps the checkpoint holds float32(beta) ** (N + 1) (tf_bundle.tf_beta_power).
        UNVERIFIED CONVENTIONS -- TensorFlow's source is not available here and no TF-written checkpoint exists to
        check against; a maintainer holding a real `model.ckpt-25600` should check, in
        this order: (1) the key set of its `.index` (tf_bundle.list_variables) against this function's: the non-slot
        accumulators are expected at top level as `beta1_power` / `beta2_power` (a second optimizer in the same graph:
        `beta1_power_1` / `beta2_power_1`), the slots as `<var>/Adam` (m) and `<var>/Adam_1` (v), Momentum's as
        `<var>/Momentum`; (2) that `beta1_power` equals 0.9 ** (global_step + 1) and not ** global_step; (3) that the
        layer scopes are `conv2d`, `conv2d_1`, ... in creation order (tf.layers' auto-numbering) for VDSR and
        `f1` / `f2` / `f3` for ESPCN; (4) dtype / shape of `global_step` (int64 scalar expected).  The container format
        itself (table blocks, CRCs, BundleEntryProto fields) is checked against published known-answer vectors only."""
        import numpy as np
        out = {k: v.detach().cpu().numpy() for k, v in self.variables().items()}
        out['global_step'] = np.asarray(self.global_step, dtype=np.int64)
        for k, v in (extra or {}).items():
            out[k] = np.asarray(v)
        if self.opt_m is not None:
            two = self.opt_v is not None
            for i, s in enumerate(self.specs):
                scope = s.scope or ('layer%d' % i)
                for kind, view in ((self.kernel_name, self.kernel), (self.bias_name, self.bias)):
                    out['%s/%s/%s' % (scope, kind, 'Adam' if two else 'Momentum')] = view(i, self.opt_m).detach().cpu().numpy()
                    if two:
                        out['%s/%s/Adam_1' % (scope, kind)] = view(i, self.opt_v).detach().cpu().numpy()
            if two:
                from . import tf_bundle
                out['beta1_power'] = tf_bundle.tf_beta_power(adam_betas[0], self.global_step)
                out['beta2_power'] = tf_bundle.tf_beta_power(adam_betas[1], self.global_step)
        return out

    def save_tf_checkpoint(self, prefix, adam_betas=(0.9, 0.999), extra=None, write_state=True):
        """Writes `<prefix>.index` + `<prefix>.data-00000-of-00001` and (write_state) the `checkpoint` state file
        (CheckpointState text proto) next to them, which is what tf.train.latest_checkpoint(dir) reads
        (vdsr/vdsr/experiment_train.py:108)."""
        from . import tf_bundle
        tf_bundle.save_checkpoint(prefix, self.tf_checkpoint_tensors(adam_betas, extra))
        if write_state:
            tf_bundle.update_checkpoint_state(prefix)

    def load_tf_checkpoint(self, prefix, with_optimizer=True):
        """Restores kernels / biases (and global_step, optimizer slots when present) from a checkpoint written by
        the reference's `tf.train.Saver` (vdsr/vdsr/experiment_train.py:100-121) or by save_tf_checkpoint."""
        from . import tf_bundle
        values = tf_bundle.load_checkpoint(prefix)
        mine = self.variables()
        missing = [k for k in mine if k not in values]
        if missing:
            raise KeyError('checkpoint %s lacks variables %s' % (prefix, missing[:4]))
        self.load_variables({k: values[k] for k in mine})
        if 'global_step' in values:
            self.global_step = int(values['global_step'])
        if not with_optimizer:
            return
        for slot, attr in (('Adam', 'opt_m'), ('Adam_1', 'opt_v'), ('Momentum', 'opt_m')):
            names = ['%s/%s' % (k, slot) for k in mine]
            if all(n in values for n in names):
                buf = torch.zeros_like(self.params)
                for i, s in enumerate(self.specs):
                    scope = s.scope or ('layer%d' % i)
                    self.kernel(i, buf).copy_(torch.as_tensor(values['%s/%s/%s' % (scope, self.kernel_name, slot)]).to(self.device))
                    self.bias(i, buf).copy_(torch.as_tensor(values['%s/%s/%s' % (scope, self.bias_name, slot)]).to(self.device))
                setattr(self, attr, buf)

    def load_checkpoint(self, path):
        """`path`: a TensorFlow V2 checkpoint prefix (as the reference's --ckpt_path, e.g. .../model.ckpt-25600)
        or a state dict saved with torch.save(stack.state_dict())."""
        from . import tf_bundle
        if tf_bundle.is_checkpoint_prefix(path):
            self.load_tf_checkpoint(path)
        else:
            self.load_state_dict(torch.load(path))

    def load_state_dict(self, sd):
        self.params.copy_(sd['params'].to(self.device))
        self.global_step = int(sd['global_step'])
        if 'opt_m' in sd:
            self.opt_m = sd['opt_m'].to(self.device).clone()
        if 'opt_v' in sd:
            self.opt_v = sd['opt_v'].to(self.device).clone()


# ---- initialisers (the reference's; TF's RNG stream itself cannot be reproduced) --------------
def xavier_uniform_(t, generator=None):
    """tf.contrib.layers.xavier_initializer(): U(+-sqrt(6/(fan_in+fan_out))), fan = kh*kw*C.
    vdsr/vdsr/model_vdsr.py:27."""
    kh, kw, cin, cout = t.shape
    lim = math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
    t.copy_((torch.rand(t.shape, generator=generator) * 2 - 1).mul_(lim).to(t.device))


def truncated_normal_(t, stddev, generator=None):
    """tf.truncated_normal_initializer(stddev): redraw beyond 2 sigma.
    espcn/espcn/model_espcn.py:21, srcnn/srcnn.py:84."""
    v = torch.randn(t.shape, generator=generator)
    bad = v.abs() > 2
    while bad.any():
        v[bad] = torch.randn(int(bad.sum()), generator=generator)
        bad = v.abs() > 2
    t.copy_(v.mul_(stddev).to(t.device))
