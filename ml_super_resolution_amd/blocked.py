"""
blocked.py -- 3x3 SAME convolution layers of ANY width on the engine's 64-channel kernels, and dense layers on its
GEMM: what EnhanceNet's discriminator (enet/enet/model_enet.py:118-162) and VGG-19 (enet/enet/model_vgg.py:65-99)
need beyond the 64-channel stacks of the other models.

Layout.  A tensor with C > 64 channels is kept as C/64 blocks: [CB, N, H, W, 64] (C <= 64: [1, N, H, W, C]), so that
every block is an ordinary NHWC tensor for the C-ABI kernels.  A layer's kernel is stored block-wise too,
[CIB, COB, 3, 3, ci, co] (each [ib][ob] slice a contiguous HWIO filter); `kernel_hwio()` / `set_kernel_hwio()`
convert to and from TensorFlow's [3, 3, Cin, Cout] for checkpoints and tests.

  forward   y[ob] = act( sum_ib conv(x[ib], w[ib][ob]) + bias[ob] ): rows of <= 64 pixels: ONE launch for the layer
            (srx_conv3x3_blocked: the sum over the input blocks stays in registers); wider images: one srx_conv2d_fwd
            per block pair, the running sum passed as the `skip` operand (bias in the first launch, the activation
            -- after the add -- in the last: post_add_relu = ReLU / leaky ReLU);
  dgrad     dx[ib] = sum_ob bwd_data(dpre[ob], w[ib][ob]): srx_conv3x3_blocked with transposed filters, or
            srx_conv2d_bwd_data followed by srx_conv2d_bwd_data_acc per further output block;
  wgrad     dw[ib][ob] = bwd_filter(x[ib], dpre[ob]): independent 64 -> 64 problems, ONE launch over all pairs + one
            reduction (srx_conv3x3_blocked_bwd_filter); dbias from the ib = 0 pairs;
  stride 2  (TF pads 0 before / 1 after on an even image, model_enet.py:136-146).  Layers of <= 64 channels run AT their
            stride in the forward pass and the filter gradient (srx_conv_desc.stride = 2: a quarter of the MFMA work).
            Wider layers, and every data gradient, use the identity "= the stride-1 layer sampled at the odd positions":
            srx_subsample2 after the forward, zero stuffing (srx_subsample2_bwd) before the gradients.
"""
import os

import torch

from . import ops

# layers wider than 64 channels on rows of <= 64 pixels: ONE launch per layer (srx_conv3x3_blocked) instead of one per
# block pair; SRX_WIDE=0 keeps the block-pair launches (A/B)
USE_WIDE = os.environ.get('SRX_WIDE', '1') != '0'
# stride-2 layers of <= 64 channels at their stride (forward and filter gradient: srx_conv_desc.stride = 2) instead of the
# stride-1 layer + sample map; SRX_TRUE_STRIDE2=0 for A/B
TRUE_STRIDE2 = os.environ.get('SRX_TRUE_STRIDE2', '1') != '0'


def n_blocks(c):
    return 1 if c <= 64 else c // 64


def block_width(c):
    if c > 64 and c % 64:
        raise ValueError('more than 64 channels must come in multiples of 64, got %d' % c)
    return c if c <= 64 else 64


def to_blocks(x_nhwc):
    return ops.nhwc_to_blocks(x_nhwc.contiguous())


def to_nhwc(x_blocked):
    return ops.blocks_to_nhwc(x_blocked)


class ParamPool(object):
    """All variables of a network in ONE flat fp32 buffer (16-byte aligned slices) with gradients in a second buffer of
    the same layout: the optimizer is one launch, the data-parallel exchange one all-reduce."""

    def __init__(self, shapes, device):
        self.device = torch.device(device)
        self.entries, off = [], 0
        for shape in shapes:
            n = 1
            for d in shape:
                n *= d
            self.entries.append((off, n, tuple(shape)))
            off += (n + 3) // 4 * 4
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.opt_m = self.opt_v = None
        self.t = 0

    def view(self, j, buf=None):
        off, n, shape = self.entries[j]
        return (self.params if buf is None else buf)[off:off + n].view(shape)

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        if self.opt_m is None:
            self.opt_m, self.opt_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.t += 1
        ops.adam_tf_step(self.params, self.grads, self.opt_m, self.opt_v, lr, self.t, beta1, beta2, eps)


class BlockedConv(object):
    def __init__(self, cin, cout, stride=1, act=None, kernel=None, bias=None, dkernel=None, dbias=None):
        """kernel / bias (and dkernel / dbias for trainable layers): views of shape [CIB, COB, 3, 3, ci, co] / [cout]."""
        self.cin, self.cout, self.stride, self.act = cin, cout, stride, act
        self.cib, self.cob = n_blocks(cin), n_blocks(cout)
        self.ci, self.co = block_width(cin), block_width(cout)
        self.w, self.b, self.dw, self.db = kernel, bias, dkernel, dbias
        self._scratch = {}

    @staticmethod
    def kernel_shape(cin, cout):
        return (n_blocks(cin), n_blocks(cout), 3, 3, block_width(cin), block_width(cout))

    def kernel_hwio(self, buf=None):
        w = self.w if buf is None else buf
        return w.permute(2, 3, 0, 4, 1, 5).reshape(3, 3, self.cin, self.cout)

    def set_kernel_hwio(self, k, bias=None):
        k = torch.as_tensor(k, dtype=torch.float32).reshape(3, 3, self.cib, self.ci, self.cob, self.co)
        self.w.copy_(k.permute(2, 4, 0, 1, 3, 5).to(self.w.device))
        if bias is not None:
            self.b.copy_(torch.as_tensor(bias, dtype=torch.float32).to(self.b.device))

    def _buf(self, key, shape, device):
        t = self._scratch.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._scratch[key] = torch.empty(shape, dtype=torch.float32, device=device)
        return t

    # ---- forward --------------------------------------------------------------------------------------------------
    def forward(self, x):
        """x [CIB, N, H, W, ci] -> y [COB, N, H/stride, W/stride, co] (a new tensor: the caller may keep it)."""
        cib, n, h, w, _ = x.shape
        assert cib == self.cib and x.shape[4] == self.ci
        if self._true_stride2(h, w):
            # one block in, one block out: the layer runs at its stride (srx_conv_desc.stride = 2), a quarter of the MFMA
            # work of the stride-1 layer + sample map
            y = torch.empty((1, n, h // 2, w // 2, self.co), dtype=torch.float32, device=x.device)
            ops.conv2d_fwd(x[0], self.w[0, 0], self.b, 'same', self.act, out=y[0], stride=2)
            return y
        y = torch.empty((self.cob, n, h, w, self.co), dtype=torch.float32, device=x.device)
        if self._wide_ok(w):
            ops.conv3x3_blocked(x, self.w, self.b, self.act, out=y)
            return self._subsampled(y) if self.stride == 2 else y
        tmp = self._buf('fwd', y.shape, x.device) if self.cib > 1 else None
        # the launches of one output block alternate between two buffers (the column-strip kernels do not take an
        # in-place skip operand); an odd / even count decides where the first one must go so that the last lands in y
        for ob in range(self.cob):
            bias = self.b[ob * self.co:(ob + 1) * self.co]
            if self.cib == 1:
                ops.conv2d_fwd(x[0], self.w[0, ob], bias, 'same', self.act, out=y[ob])
                continue
            bufs = (y, tmp) if self.cib % 2 == 1 else (tmp, y)
            for ib in range(self.cib):
                dst = bufs[ib % 2][ob]
                last = ib == self.cib - 1
                ops.conv2d_fwd(x[ib], self.w[ib, ob], bias if ib == 0 else None, 'same', None,
                               skip=bufs[(ib + 1) % 2][ob] if ib > 0 else None,
                               post_add_relu=_POST_ACT[self.act] if last else 0, out=dst)
        return self._subsampled(y) if self.stride == 2 else y

    def _true_stride2(self, h, w):
        return TRUE_STRIDE2 and self.stride == 2 and self.cib == 1 and self.cob == 1 and h % 2 == 0 and w % 2 == 0

    def _wide_ok(self, width):
        return (USE_WIDE and (self.cib > 1 or self.cob > 1) and self.ci == 64 and self.co == 64 and
                self.act in (None, 'relu', 'lrelu', 'leaky_relu'))

    def _subsampled(self, y):
        cob, n, h, w, co = y.shape
        ys = torch.empty((cob, n, h // 2, w // 2, co), dtype=torch.float32, device=y.device)
        # (the blocks are contiguous: one launch with them as extra images)
        ops.subsample2(y.view(cob * n, h, w, co), 1, 1, out=ys.view(cob * n, h // 2, w // 2, co))
        return ys

    # ---- backward -------------------------------------------------------------------------------------------------
    def _full_res(self, dpre):
        if self.stride == 1:
            return dpre
        cob, n, h, w, co = dpre.shape
        full = self._buf('stuff', (cob, n, 2 * h, 2 * w, co), dpre.device)
        ops.subsample2_bwd(dpre.contiguous().view(cob * n, h, w, co), 1, 1, out=full.view(cob * n, 2 * h, 2 * w, co))
        return full

    def full_res(self, dpre):
        """The zero-stuffed gradient that wgrad() and dgrad() of a stride-2 layer both start from: a caller that needs
        both computes it once and passes it as `stuffed` (one launch and one full-size write instead of two).  None where
        the layer does not need it."""
        if self.stride == 1:
            return None
        return self._full_res(dpre)

    def dgrad(self, dpre, mask=None, mask_act=None, stuffed=None):
        """dpre [COB, N, OH, OW, co] (gradient w.r.t. the layer's pre-activation output) -> dx [CIB, N, H, W, ci],
        the gradient w.r.t. the layer's input.  mask / mask_act: the layer's INPUT as saved (the post-activation
        output of the layer below) and that layer's activation -- the result is then already multiplied by the
        activation gradient (ReluGrad / leaky-ReLU gradient), fused into the launch where the kernel allows.
        stuffed: full_res(dpre) if the caller already has it."""
        dp = stuffed if stuffed is not None else self._full_res(dpre)
        _, n, h, w, _ = dp.shape
        dx = torch.empty((self.cib, n, h, w, self.ci), dtype=torch.float32, device=dp.device)
        if self._wide_ok(w):
            return ops.conv3x3_blocked(dp, self.w, None, None, transpose=True, out=dx, mask=mask, mask_act=mask_act)
        xs = (n, h, w, self.ci)
        if self.cob == 1 and mask is not None:
            for ib in range(self.cib):
                ops.conv2d_bwd_data(dp[0], self.w[ib, 0], xs, 'same', x_in=mask[ib], in_act=mask_act, out=dx[ib])
            return dx
        for ib in range(self.cib):
            ops.conv2d_bwd_data(dp[0], self.w[ib, 0], xs, 'same', out=dx[ib])
            for ob in range(1, self.cob):
                ops.conv2d_bwd_data_acc(dp[ob], self.w[ib, ob], xs, dx[ib], 'same', out=dx[ib])
            if mask is not None:
                ops.act_bwd(dx[ib], mask[ib], mask_act, out=dx[ib])
        return dx

    def wgrad(self, x, dpre, stuffed=None):
        """Fills self.dw / self.db from the layer input x [CIB, N, H, W, ci] and dpre (stuffed: full_res(dpre) if at hand)."""
        if self._true_stride2(x.shape[2], x.shape[3]):
            need = max(ops.bwd_filter_workspace_bytes(x[0].shape, self.w[0, 0].shape, 'same', stride=2), 16)
            ws = self._scratch.get('ws')
            if ws is None or ws.numel() * 4 < need:
                ws = self._scratch['ws'] = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
            ops.conv2d_bwd_filter(x[0], dpre[0].contiguous(), self.w[0, 0].shape, 'same', dw=self.dw[0, 0], dbias=self.db, workspace=ws,
                                  stride=2)
            return
        dp = stuffed if stuffed is not None else self._full_res(dpre)
        if self._wide_ok(x.shape[3]) and x.is_contiguous() and dp.is_contiguous() and self.dw.is_contiguous():
            _, n, h, w, _ = x.shape
            need = ops.conv3x3_blocked_bwd_filter_workspace_bytes(n, h, w, self.cib, self.cob)
            ws = self._scratch.get('ws')
            if ws is None or ws.numel() * 4 < need:
                ws = self._scratch['ws'] = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
            ops.conv3x3_blocked_bwd_filter(x, dp, self.dw, self.db, workspace=ws)
            return
        need = max(ops.bwd_filter_workspace_bytes(x[0].shape, self.w[0, 0].shape, 'same'), 16)
        ws = self._scratch.get('ws')
        if ws is None or ws.numel() * 4 < need:
            ws = self._scratch['ws'] = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        scratch_db = self._buf('db', (self.co,), x.device)
        for ib in range(self.cib):
            for ob in range(self.cob):
                db = self.db[ob * self.co:(ob + 1) * self.co] if ib == 0 else scratch_db
                ops.conv2d_bwd_filter(x[ib], dp[ob], self.w[ib, ob].shape, 'same', dw=self.dw[ib, ob], dbias=db, workspace=ws)


_POST_ACT = {None: 0, 'relu': 1, 'lrelu': 3, 'leaky_relu': 3}


class DenseLayer(object):
    """tf.layers.dense(units, activation) (enet/enet/model_enet.py:148-160) on srx_gemm."""

    def __init__(self, fin, fout, act, w, b, dw=None, db=None):
        self.fin, self.fout, self.act = fin, fout, act
        self.w, self.b, self.dw, self.db = w, b, dw, db

    def forward(self, x):
        return ops.gemm(x, self.w, bias=self.b, act=self.act)

    def backward(self, x, y, dy, want_dx=True, want_dw=True):
        """dy = gradient w.r.t. the activated output y.  Returns dx (or None)."""
        dz = ops.act_bwd(dy, y, self.act) if self.act is not None else dy
        if want_dw:
            ops.gemm(x, dz, trans_a=True, out=self.dw)
            ops.column_sums(dz, out=self.db)
        return ops.gemm(dz, self.w, trans_b=True) if want_dx else None
