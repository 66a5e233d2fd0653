"""Thin op wrappers over the C ABI.  Tensors are torch CUDA(ROCm) float32, NHWC activations,
HWIO filters.  Every call is asynchronous on torch's current stream."""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_BY_NAME, PAD_BY_NAME, ConvDesc, check
from ._lib import lib as _load_lib

# SRX_POISON_LDS=1 (test aid): every CU's LDS is filled with NaNs before each call into the library, so a kernel that depends on
# LDS bytes it has not written shows up in the parity tests (include/srx.h: srx_debug_poison_lds)
_POISON_LDS = os.environ.get('SRX_POISON_LDS', '0') == '1'


def lib():
    L = _load_lib()
    if _POISON_LDS and torch.cuda.is_available():
        L.srx_debug_poison_lds(_stream())
    return L



def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, name):
    if t is None:
        return
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError('%s must be a contiguous float32 tensor on the GPU' % name)
    if t.device.index != torch.cuda.current_device():
        # the launch goes to the CURRENT device's current stream: a tensor of another GPU would be read from the
        # wrong device (one process per GPU: call torch.cuda.set_device(local_rank) first)
        raise ValueError('%s lives on %s but the current device is cuda:%d' % (name, t.device, torch.cuda.current_device()))


def conv_desc(x_shape, w_shape, padding='same', act=None, post_add_relu=False, subpixel_r=0, stride=1):
    N, H, W, Cin = x_shape
    KH, KW, wci, Cout = w_shape
    if wci != Cin:
        raise ValueError('filter Cin %d != input channels %d' % (wci, Cin))
    return ConvDesc(N, H, W, Cin, Cout, KH, KW, int(stride), PAD_BY_NAME[padding.lower()],
                    ACT_BY_NAME[act] if not isinstance(act, int) else act, int(post_add_relu), 0, int(subpixel_r))


def out_shape(d):
    """TensorFlow's geometry: SAME ceil(in / stride); VALID (in - k) / stride + 1."""
    s = d.stride
    if d.pad_mode == _lib.PAD_SAME:
        return (d.N, (d.H + s - 1) // s, (d.W + s - 1) // s, d.Cout)
    return (d.N, (d.H - d.KH) // s + 1, (d.W - d.KW) // s + 1, d.Cout)


_scratch = {}
_sched_ws = {}


def sched_workspace(device):
    """256 bytes per (device, stream) lent to forward / dgrad launches for their dynamic tile counter."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _sched_ws.get(key)
    if t is None:
        t = torch.zeros(64, dtype=torch.int32, device=device)
        _sched_ws[key] = t
    return t


def reduce_scratch(device):
    key = (device.type, device.index)
    if key not in _scratch:
        _scratch[key] = torch.empty(lib().srx_reduce_scratch_bytes() // 4, dtype=torch.float32, device=device)
    return _scratch[key]


def conv2d_fwd(x, w, bias=None, padding='same', act=None, skip=None, post_add_relu=False, out=None, subpixel_r=0, stride=1):
    """act(bias + x (*) w) [+ skip] [relu] -- srx_conv2d_fwd.  subpixel_r > 1: the result is stored through the
    depth-to-space map, [N,OH*r,OW*r,Cout/r^2] (bit-identical to conv2d_fwd + depth_to_space, one launch).
    stride 1 or 2 (tf.layers.conv2d(strides=2, padding='same'): enet/enet/model_enet.py:136-146)."""
    for t, n in ((x, 'x'), (w, 'w'), (bias, 'bias'), (skip, 'skip')):
        _chk(t, n)
    d = conv_desc(x.shape, w.shape, padding, act, post_add_relu, subpixel_r, stride)
    shape = out_shape(d)
    if subpixel_r > 1:
        r = int(subpixel_r)
        if shape[3] % (r * r):
            raise ValueError('subpixel_r %d: %d output channels are not a multiple of r*r' % (r, shape[3]))
        shape = (shape[0], shape[1] * r, shape[2] * r, shape[3] // (r * r))
    if out is not None and tuple(out.shape) != tuple(shape):
        raise ValueError('out has shape %s, expected %s' % (tuple(out.shape), tuple(shape)))
    y = out if out is not None else torch.empty(shape, dtype=torch.float32, device=x.device)
    ws = sched_workspace(x.device)
    check(lib().srx_conv2d_fwd(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(skip), _ptr(y),
                               ctypes.c_void_p(ws.data_ptr()), 256, _stream()), 'srx_conv2d_fwd')
    return y


def conv2d_bwd_data(dpre, w, x_shape, padding='same', x_in=None, in_act=None, out=None):
    """dx * act'(x_in) -- srx_conv2d_bwd_data."""
    for t, n in ((dpre, 'dpre'), (w, 'w'), (x_in, 'x_in')):
        _chk(t, n)
    d = conv_desc(x_shape, w.shape, padding)
    dx = out if out is not None else torch.empty(tuple(x_shape), dtype=torch.float32, device=dpre.device)
    ws = sched_workspace(dpre.device)
    check(lib().srx_conv2d_bwd_data(ctypes.byref(d), _ptr(dpre), _ptr(w), _ptr(x_in), ACT_BY_NAME[in_act],
                                    _ptr(dx), ctypes.c_void_p(ws.data_ptr()), 256, _stream()), 'srx_conv2d_bwd_data')
    return dx


def bwd_filter_workspace_bytes(x_shape, w_shape, padding='same', stride=1):
    d = conv_desc(x_shape, w_shape, padding, stride=stride)
    return lib().srx_conv2d_workspace_bytes(ctypes.byref(d), _lib.OP_BWD_FILTER)


def conv2d_bwd_filter(x, dpre, w_shape, padding='same', w_for_decay=None, wd_scale=0.0, dw=None, dbias=None,
                      want_dbias=True, workspace=None, stride=1):
    """(dw, dbias) -- srx_conv2d_bwd_filter.  stride 2: dpre is the gradient at the layer's (half-resolution) output."""
    for t, n in ((x, 'x'), (dpre, 'dpre'), (w_for_decay, 'w_for_decay')):
        _chk(t, n)
    d = conv_desc(x.shape, w_shape, padding, stride=stride)
    if tuple(dpre.shape) != tuple(out_shape(d)):
        raise ValueError('dpre has shape %s, the layer output is %s' % (tuple(dpre.shape), tuple(out_shape(d))))
    if dw is None:
        dw = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    if dbias is None and want_dbias:
        dbias = torch.empty((w_shape[3],), dtype=torch.float32, device=x.device)
    need = lib().srx_conv2d_workspace_bytes(ctypes.byref(d), _lib.OP_BWD_FILTER)
    if workspace is None:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    check(lib().srx_conv2d_bwd_filter(ctypes.byref(d), _ptr(x), _ptr(dpre), _ptr(dw), _ptr(dbias),
                                      _ptr(w_for_decay), float(wd_scale), _ptr(workspace),
                                      workspace.numel() * 4, _stream()), 'srx_conv2d_bwd_filter')
    return dw, dbias


def conv2d_bwd_filter_partials(x, dpre, w_shape, padding, workspace):
    """First half of conv2d_bwd_filter: per-workgroup partial filters into `workspace`; returns their count."""
    _chk(x, 'x'); _chk(dpre, 'dpre'); _chk(workspace, 'workspace')
    d = conv_desc(x.shape, w_shape, padding)
    n = ctypes.c_int(0)
    check(lib().srx_conv2d_bwd_filter_partials(ctypes.byref(d), _ptr(x), _ptr(dpre), _ptr(workspace), workspace.numel() * 4,
                                               ctypes.byref(n), _stream()), 'srx_conv2d_bwd_filter_partials')
    return n.value


def conv2d_bwd_filter_reduce(x_shape, w_shape, padding, workspace, n_partials, dw, dbias=None, w_for_decay=None, wd_scale=0.0):
    """Second half: sums the partials into dw / dbias on the CURRENT stream (the caller orders it after the first half)."""
    _chk(workspace, 'workspace'); _chk(dw, 'dw')
    d = conv_desc(x_shape, w_shape, padding)
    check(lib().srx_conv2d_bwd_filter_reduce(ctypes.byref(d), _ptr(workspace), int(n_partials), _ptr(dw), _ptr(dbias),
                                             _ptr(w_for_decay), float(wd_scale), _stream()), 'srx_conv2d_bwd_filter_reduce')
    return dw, dbias


def act_bwd(dy, y, act, out=None):
    _chk(dy, 'dy'); _chk(y, 'y')
    out = out if out is not None else torch.empty_like(dy)
    check(lib().srx_act_bwd(_ptr(dy), _ptr(y), _ptr(out), dy.numel(), ACT_BY_NAME[act], _stream()), 'srx_act_bwd')
    return out


def depth_to_space(x, r, out=None):
    """[N,H,W,C*r*r] -> [N,H*r,W*r,C], TF channel order (dy,dx,c)."""
    _chk(x, 'x')
    N, H, W, D = x.shape
    if D % (r * r):
        raise ValueError('depth %d not divisible by r*r=%d' % (D, r * r))
    C = D // (r * r)
    out = out if out is not None else torch.empty((N, H * r, W * r, C), dtype=torch.float32, device=x.device)
    check(lib().srx_depth_to_space(_ptr(x), _ptr(out), N, H, W, C, r, _stream()), 'srx_depth_to_space')
    return out


def espcn_forward(x, params, r, out=None):
    """ESPCN inference in one launch -- srx_espcn_forward.  params = [(w1, b1), (w2, b2), (w3, b3)] (HWIO kernels);
    x [N,H,W,3] -> [N,H*r,W*r,3]."""
    _chk(x, 'x')
    for k, b in params:
        _chk(k, 'kernel'); _chk(b, 'bias')
    (w1, b1), (w2, b2), (w3, b3) = params
    N, H, W, C = x.shape
    if C != 3 or tuple(w1.shape) != (5, 5, 3, 64) or tuple(w2.shape) != (3, 3, 64, 32) or tuple(w3.shape) != (3, 3, 32, 3 * r * r):
        raise ValueError('espcn_forward: shapes do not describe ESPCN 5-3-3 with scaling factor %d' % r)
    out = out if out is not None else torch.empty((N, H * r, W * r, 3), dtype=torch.float32, device=x.device)
    check(lib().srx_espcn_forward(_ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(out),
                                  N, H, W, int(r), _stream()), 'srx_espcn_forward')
    return out


def espcn_forward_keep(x, params, r, t1, t2, y):
    """ESPCN's forward pass of a train step in one launch -- srx_espcn_forward_keep: writes the activations t1 [N,H,W,64],
    t2 [N,H,W,32] and the output y [N,H,W,3 r^2] (sub-pixel space) into the given tensors.  Returns y."""
    _chk(x, 'x')
    for k, b in params:
        _chk(k, 'kernel'); _chk(b, 'bias')
    (w1, b1), (w2, b2), (w3, b3) = params
    N, H, W, C = x.shape
    if C != 3 or tuple(w1.shape) != (5, 5, 3, 64) or tuple(w2.shape) != (3, 3, 64, 32) or tuple(w3.shape) != (3, 3, 32, 3 * r * r):
        raise ValueError('espcn_forward_keep: shapes do not describe ESPCN 5-3-3 with scaling factor %d' % r)
    for t, c, name in ((t1, 64, 't1'), (t2, 32, 't2'), (y, 3 * r * r, 'y')):
        _chk(t, name)
        if tuple(t.shape) != (N, H, W, c):
            raise ValueError('espcn_forward_keep: %s has shape %s, expected %s' % (name, tuple(t.shape), (N, H, W, c)))
    check(lib().srx_espcn_forward_keep(_ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(t1), _ptr(t2),
                                       _ptr(y), N, H, W, int(r), _stream()), 'srx_espcn_forward_keep')
    return y


def srcnn_forward(x, params, out=None):
    """SRCNN 9-1-5 VALID inference in one launch -- srx_srcnn_forward.  params = [(w1, b1), (w2, b2), (w3, b3)] (HWIO kernels);
    x [N,H,W,3] -> [N,H-12,W-12,3]."""
    _chk(x, 'x')
    for k, b in params:
        _chk(k, 'kernel'); _chk(b, 'bias')
    (w1, b1), (w2, b2), (w3, b3) = params
    N, H, W, C = x.shape
    if C != 3 or tuple(w1.shape) != (9, 9, 3, 64) or tuple(w2.shape) != (1, 1, 64, 32) or tuple(w3.shape) != (5, 5, 32, 3):
        raise ValueError('srcnn_forward: shapes do not describe SRCNN 9-1-5')
    if H < 13 or W < 13:
        raise ValueError('srcnn_forward: image %dx%d smaller than the 13-pixel receptive field' % (H, W))
    out = out if out is not None else torch.empty((N, H - 12, W - 12, 3), dtype=torch.float32, device=x.device)
    check(lib().srx_srcnn_forward(_ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(out),
                                  N, H, W, _stream()), 'srx_srcnn_forward')
    return out


def space_to_depth(x, r, out=None):
    """[N,H*r,W*r,C] -> [N,H,W,C*r*r]."""
    _chk(x, 'x')
    N, HR, WR, C = x.shape
    if HR % r or WR % r:
        raise ValueError('spatial dims %dx%d not divisible by r=%d' % (HR, WR, r))
    H, W = HR // r, WR // r
    out = out if out is not None else torch.empty((N, H, W, C * r * r), dtype=torch.float32, device=x.device)
    check(lib().srx_space_to_depth(_ptr(x), _ptr(out), N, H, W, C, r, _stream()), 'srx_space_to_depth')
    return out


def stream_copy(src, dst):
    """Plain streaming copy on the library's own copy kernel (a measurement aid: the ceiling of a byte-moving kernel)."""
    _chk(src, 'src'); _chk(dst, 'dst')
    if src.numel() != dst.numel():
        raise ValueError('stream_copy: sizes differ')
    check(lib().srx_stream_copy(_ptr(src), _ptr(dst), src.numel() * 4, _stream()), 'srx_stream_copy')
    return dst


def mse_fwd_bwd(pred, target, loss_out, inv_numel=None, accumulate=False, dpred=None, want_grad=True):
    """loss_out (+)= sum((pred-target)^2)*inv_numel; returns dpred = 2*(pred-target)*inv_numel."""
    _chk(pred, 'pred'); _chk(target, 'target')
    if inv_numel is None:
        inv_numel = 1.0 / pred.numel()
    if dpred is None and want_grad:
        dpred = torch.empty_like(pred)
    check(lib().srx_mse_fwd_bwd(_ptr(pred), _ptr(target), pred.numel(), float(inv_numel), _ptr(loss_out),
                                int(accumulate), _ptr(dpred), _ptr(reduce_scratch(pred.device)), _stream()),
          'srx_mse_fwd_bwd')
    return dpred


def l2_loss(w, scale, loss_out, accumulate=True, mask=None):
    """loss_out (+)= scale * sum(mask * w^2) / 2; mask selects the regularised elements of a flat buffer."""
    _chk(w, 'w'); _chk(mask, 'mask')
    check(lib().srx_l2_loss(_ptr(w), _ptr(mask), w.numel(), float(scale), _ptr(loss_out), int(accumulate),
                            _ptr(reduce_scratch(w.device)), _stream()), 'srx_l2_loss')


def adam_tf_step(w, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    for tns, n in ((w, 'w'), (g, 'g'), (m, 'm'), (v, 'v')):
        _chk(tns, n)
    check(lib().srx_adam_tf_step(_ptr(w), _ptr(g), _ptr(m), _ptr(v), w.numel(), float(lr), float(beta1),
                                 float(beta2), float(eps), int(t), float(grad_scale), _stream()), 'srx_adam_tf_step')


def adam_state(device, t=0, lr=0.0):
    """The 32-byte device block srx_adam_tf_step_dev works on: {int64 t; float lr; float lr_t; uint32 done, pad} (+ padding)."""
    st = torch.zeros(8, dtype=torch.int32, device=device)
    adam_state_set(st, t=t, lr=lr)
    return st


def adam_state_set(st, t=None, lr=None):
    if t is not None:
        st[0:2].copy_(torch.tensor([int(t)], dtype=torch.int64).view(torch.int32))
    if lr is not None:
        st[2:3].copy_(torch.tensor([float(lr)], dtype=torch.float32).view(torch.int32))


def adam_state_get(st):
    """(t, lr, lr_t) -- synchronises; for tests and checkpoints."""
    h = st.cpu()
    return int(h[0:2].view(torch.int64).item()), float(h[2:3].view(torch.float32).item()), float(h[3:4].view(torch.float32).item())


def adam_tf_step_dev(w, g, m, v, state, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    for tns, n in ((w, 'w'), (g, 'g'), (m, 'm'), (v, 'v')):
        _chk(tns, n)
    check(lib().srx_adam_tf_step_dev(_ptr(w), _ptr(g), _ptr(m), _ptr(v), w.numel(), _ptr(state), float(beta1), float(beta2),
                                     float(eps), float(grad_scale), _stream()), 'srx_adam_tf_step_dev')


def momentum_clip_step(w, g, acc, lr, momentum=0.9, cap=float('inf'), grad_scale=1.0):
    for tns, n in ((w, 'w'), (g, 'g'), (acc, 'acc')):
        _chk(tns, n)
    cap = min(float(cap), 3.0e38)
    check(lib().srx_momentum_clip_step(_ptr(w), _ptr(g), _ptr(acc), w.numel(), float(lr), float(momentum), cap,
                                       float(grad_scale), _stream()), 'srx_momentum_clip_step')


def rownorm_loss_fwd_bwd(pred, target, row_len, loss_out, want_grad=True, dpred=None, norms=None):
    """SRCNN loss: mean over rows of ||reshape(pred-target, [-1,row_len])||_2 (+ its gradient).  dpred / norms: caller-owned
    output / scratch tensors (a captured train step must not allocate)."""
    _chk(pred, 'pred'); _chk(target, 'target'); _chk(dpred, 'dpred'); _chk(norms, 'norms')
    rows = pred.numel() // row_len
    if rows * row_len != pred.numel():
        raise ValueError('numel %d not divisible by row_len %d' % (pred.numel(), row_len))
    if norms is None:
        norms = torch.empty(rows, dtype=torch.float32, device=pred.device)
    if dpred is None and want_grad:
        dpred = torch.empty_like(pred)
    check(lib().srx_rownorm_loss_fwd_bwd(_ptr(pred), _ptr(target), rows, row_len, _ptr(loss_out), _ptr(dpred),
                                         _ptr(norms), _stream()), 'srx_rownorm_loss_fwd_bwd')
    return dpred


def psnr(a, b, max_val):
    _chk(a, 'a'); _chk(b, 'b')
    N = a.shape[0]
    out = torch.empty((N,), dtype=torch.float32, device=a.device)
    check(lib().srx_psnr(_ptr(a), _ptr(b), _ptr(out), N, a.numel() // N, float(max_val), _stream()), 'srx_psnr')
    return out


def ssim(a, b, max_val):
    """tf.image.ssim(a, b, max_val) per image: [N,H,W,C] x2 -> [N]."""
    _chk(a, 'a'); _chk(b, 'b')
    N, H, W, C = a.shape
    out = torch.empty((N,), dtype=torch.float32, device=a.device)
    scratch = torch.empty(lib().srx_ssim_scratch_bytes(N) // 4, dtype=torch.float32, device=a.device)
    check(lib().srx_ssim(_ptr(a), _ptr(b), _ptr(out), N, H, W, C, float(max_val), _ptr(scratch), _stream()), 'srx_ssim')
    return out


def saturate_u8(x):
    _chk(x, 'x')
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib().srx_saturate_u8(_ptr(x), ctypes.c_void_p(out.data_ptr()), x.numel(), _stream()), 'srx_saturate_u8')
    return out


def affine(x, a, b, out=None):
    _chk(x, 'x')
    out = out if out is not None else torch.empty_like(x)
    check(lib().srx_affine(_ptr(x), _ptr(out), x.numel(), float(a), float(b), _stream()), 'srx_affine')
    return out


def u8_to_unit_float(x):
    """uint8 tensor -> float32 in [0,1] (skimage.util.img_as_float32)."""
    if not x.is_cuda or x.dtype != torch.uint8 or not x.is_contiguous():
        raise ValueError('x must be a contiguous uint8 tensor on the GPU')
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().srx_u8_to_unit_float(ctypes.c_void_p(x.data_ptr()), _ptr(out), x.numel(), _stream()), 'srx_u8_to_unit_float')
    return out


RESAMPLE_FILTERS = {'bilinear': 0, 'bicubic': 1}
_resample_tables = {}


def pil_resample_coeffs(in_size, out_size, filt):
    """Pillow's coefficient tables for one axis (srx_pil_resample_coeffs): (bounds int32 [out, 2], kk int32 [out, ksize])."""
    import numpy as np
    f = RESAMPLE_FILTERS[filt]
    ksize = lib().srx_pil_resample_ksize(in_size, out_size, f)
    if ksize < 0:
        raise ValueError('bad resample sizes %d -> %d' % (in_size, out_size))
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    check(lib().srx_pil_resample_coeffs(in_size, out_size, f, ctypes.c_void_p(bounds.ctypes.data), ctypes.c_void_p(kk.ctypes.data)),
          'srx_pil_resample_coeffs')
    return bounds, kk


def _device_tables(in_size, out_size, filt, device):
    key = (in_size, out_size, filt, str(device))
    t = _resample_tables.get(key)
    if t is None:
        b, k = pil_resample_coeffs(in_size, out_size, filt)
        t = _resample_tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1])
    return t


def resize_pil_u8(x, out_h, out_w, filt='bicubic'):
    """scipy.misc.imresize / PIL.Image.resize of uint8 images on the GPU, byte for byte: x [N,H,W,C] uint8 ->
    [N,out_h,out_w,C] uint8; two passes of srx_resample_u8 (horizontal, then vertical), the intermediate in uint8."""
    if not x.is_cuda or x.dtype != torch.uint8 or not x.is_contiguous() or x.dim() != 4:
        raise ValueError('x must be a contiguous uint8 [N,H,W,C] tensor on the GPU')
    N, H, W, C = x.shape
    t = x
    if out_w != W:
        b, k, ks = _device_tables(W, out_w, filt, x.device)
        t2 = torch.empty((N, H, out_w, C), dtype=torch.uint8, device=x.device)
        check(lib().srx_resample_u8(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t2.data_ptr()), N * H, W, out_w, C,
                                    ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(k.data_ptr()), ks, _stream()), 'srx_resample_u8')
        t = t2
    if out_h != H:
        b, k, ks = _device_tables(H, out_h, filt, x.device)
        t2 = torch.empty((N, out_h, t.shape[2], C), dtype=torch.uint8, device=x.device)
        check(lib().srx_resample_u8(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t2.data_ptr()), N, H, out_h, t.shape[2] * C,
                                    ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(k.data_ptr()), ks, _stream()), 'srx_resample_u8')
        t = t2
    return t if t is not x else x.clone()


def u8_to_pm1(x):
    """uint8 -> float32 in [-1, 1]: astype(float32) / 127.5 - 1.0 (two roundings), as the reference's data pipelines do."""
    if not x.is_cuda or x.dtype != torch.uint8 or not x.is_contiguous():
        raise ValueError('x must be a contiguous uint8 tensor on the GPU')
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().srx_u8_to_pm1(ctypes.c_void_p(x.data_ptr()), _ptr(out), x.numel(), _stream()), 'srx_u8_to_pm1')
    return out


def gaussian_blur(x, sigma):
    """skimage.filters.gaussian(x, sigma, mode='nearest') on [N,H,W,C]."""
    _chk(x, 'x')
    N, H, W, C = x.shape
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    check(lib().srx_gaussian_blur(_ptr(x), _ptr(out), _ptr(tmp), N, H, W, C, float(sigma), _stream()), 'srx_gaussian_blur')
    return out


def resize_bilinear(x, oh, ow):
    """skimage.transform.resize(x, [oh, ow], mode='edge', anti_aliasing=False), order 1."""
    _chk(x, 'x')
    N, H, W, C = x.shape
    out = torch.empty((N, oh, ow, C), dtype=torch.float32, device=x.device)
    check(lib().srx_resize_bilinear(_ptr(x), _ptr(out), N, H, W, C, int(oh), int(ow), _stream()), 'srx_resize_bilinear')
    return out


def resize_bicubic_tf(x, oh, ow):
    """tf.image.resize_bicubic(x, [oh, ow]) with TensorFlow 1.x semantics (srcnn/srcnn.py:89-93)."""
    _chk(x, 'x')
    N, H, W, C = x.shape
    out = torch.empty((N, int(oh), int(ow), C), dtype=torch.float32, device=x.device)
    check(lib().srx_resize_bicubic_tf(_ptr(x), _ptr(out), N, H, W, C, int(oh), int(ow), _stream()), 'srx_resize_bicubic_tf')
    return out


def upsample_nearest(x, f):
    _chk(x, 'x')
    N, H, W, C = x.shape
    out = torch.empty((N, H * f, W * f, C), dtype=torch.float32, device=x.device)
    check(lib().srx_upsample_nearest(_ptr(x), _ptr(out), N, H, W, C, f, _stream()), 'srx_upsample_nearest')
    return out


def upsample_nearest_bwd(dout, f, out=None):
    """Gradient of upsample_nearest: sums each f x f block.  dout [N,H*f,W*f,C] -> [N,H,W,C]."""
    _chk(dout, 'dout')
    N, HF, WF, C = dout.shape
    if HF % f or WF % f:
        raise ValueError('upsample_nearest_bwd: %dx%d is not a multiple of the factor %d' % (HF, WF, f))
    out = out if out is not None else torch.empty((N, HF // f, WF // f, C), dtype=torch.float32, device=dout.device)
    check(lib().srx_upsample_nearest_bwd(_ptr(dout), _ptr(out), N, HF // f, WF // f, C, f, _stream()), 'srx_upsample_nearest_bwd')
    return out


def add_relu_grad(a, b, y, out=None):
    """(y > 0) ? a + b : 0 -- the gradient at the input of a residual block's closing ReLU chain."""
    _chk(a, 'a'); _chk(b, 'b'); _chk(y, 'y')
    if a.shape != b.shape or a.shape != y.shape:
        raise ValueError('add_relu_grad: shapes differ')
    out = out if out is not None else torch.empty_like(a)
    check(lib().srx_add_relu_grad(_ptr(a), _ptr(b), _ptr(y), _ptr(out), a.numel(), _stream()), 'srx_add_relu_grad')
    return out


# ---- EnhanceNet-PAT's loss side ------------------------------------------------------------
def conv2d_bwd_data_acc(dpre, w, x_shape, dx_acc, padding='same', out=None):
    """dx_acc + Conv2DBackpropInput(dpre, w) -- srx_conv2d_bwd_data_acc (out may be dx_acc itself)."""
    for t, n in ((dpre, 'dpre'), (w, 'w'), (dx_acc, 'dx_acc')):
        _chk(t, n)
    d = conv_desc(x_shape, w.shape, padding)
    dx = out if out is not None else torch.empty(tuple(x_shape), dtype=torch.float32, device=dpre.device)
    ws = sched_workspace(dpre.device)
    check(lib().srx_conv2d_bwd_data_acc(ctypes.byref(d), _ptr(dpre), _ptr(w), _ptr(dx_acc), _ptr(dx),
                                        ctypes.c_void_p(ws.data_ptr()), 256, _stream()), 'srx_conv2d_bwd_data_acc')
    return dx


def conv3x3_blocked(x, w, bias=None, act=None, transpose=False, out=None, mask=None, mask_act=None):
    """A whole 3x3 SAME layer wider than 64 channels in one launch -- srx_conv3x3_blocked.
    x [SB, N, H, W, 64]; w [CIB, COB, 3, 3, 64, 64] (the forward layer's filters); forward: SB = CIB, result
    [COB, N, H, W, 64] = act(sum + bias); transpose=True (data gradient): SB = COB, result [CIB, N, H, W, 64]."""
    _chk(x, 'x'); _chk(w, 'w'); _chk(bias, 'bias'); _chk(mask, 'mask')
    sb, n, h, wd, c = x.shape
    cib, cob = w.shape[0], w.shape[1]
    if c != 64 or tuple(w.shape[2:]) != (3, 3, 64, 64) or sb != (cob if transpose else cib):
        raise ValueError('conv3x3_blocked: x %s does not fit filters %s' % (tuple(x.shape), tuple(w.shape)))
    pb = cib if transpose else cob
    out = out if out is not None else torch.empty((pb, n, h, wd, 64), dtype=torch.float32, device=x.device)
    if mask is not None and tuple(mask.shape) != tuple(out.shape):
        raise ValueError('conv3x3_blocked: mask %s does not have the result shape %s' % (tuple(mask.shape), tuple(out.shape)))
    check(lib().srx_conv3x3_blocked(_ptr(x), _ptr(w), _ptr(bias), _ptr(mask), ACT_BY_NAME[mask_act], _ptr(out), n, h, wd, sb, pb,
                                    ACT_BY_NAME[act], int(transpose), _stream()), 'srx_conv3x3_blocked')
    return out


def conv3x3_blocked_bwd_filter_workspace_bytes(n, h, w, cib, cob):
    return int(lib().srx_conv3x3_blocked_bwd_filter_workspace_bytes(n, h, w, cib, cob))


def conv3x3_blocked_bwd_filter(x, dpre, dw, dbias=None, workspace=None):
    """Filter gradient of a 3x3 SAME layer wider than 64 channels -- srx_conv3x3_blocked_bwd_filter: all block pairs in
    one launch + one reduction.  x [CIB, N, H, W, 64], dpre [COB, N, H, W, 64] -> dw [CIB, COB, 3, 3, 64, 64], dbias [64 COB]."""
    _chk(x, 'x'); _chk(dpre, 'dpre'); _chk(dw, 'dw'); _chk(dbias, 'dbias')
    cib, n, h, w, c = x.shape
    cob = dpre.shape[0]
    if c != 64 or tuple(dpre.shape[1:]) != (n, h, w, 64) or tuple(dw.shape) != (cib, cob, 3, 3, 64, 64) or \
            (dbias is not None and dbias.numel() != 64 * cob):
        raise ValueError('conv3x3_blocked_bwd_filter: x %s, dpre %s, dw %s do not fit' % (tuple(x.shape), tuple(dpre.shape), tuple(dw.shape)))
    need = conv3x3_blocked_bwd_filter_workspace_bytes(n, h, w, cib, cob)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((max(need, 16) + 3) // 4, dtype=torch.float32, device=x.device)
    check(lib().srx_conv3x3_blocked_bwd_filter(_ptr(x), _ptr(dpre), _ptr(dw), _ptr(dbias), n, h, w, cib, cob, _ptr(workspace),
                                               workspace.numel() * workspace.element_size(), _stream()), 'srx_conv3x3_blocked_bwd_filter')
    return dw


def texture_gram(x, eps=1e-6, out=None):
    """Gram matrices of the 16x16 patches of normalize(x) -- srx_texture_gram: [N,H,W,C] -> [N*(H/16)*(W/16), C, C]."""
    _chk(x, 'x')
    n, h, w, c = x.shape
    out = out if out is not None else torch.empty((n * (h // 16) * (w // 16), c, c), dtype=torch.float32, device=x.device)
    check(lib().srx_texture_gram(_ptr(x), _ptr(out), n, h, w, c, eps, _stream()), 'srx_texture_gram')
    return out


def texture_gram_bwd(x, dgram, eps=1e-6, alpha=2.0, out=None):
    """d loss / d x of texture_gram given the (symmetric) d loss / d gram -- srx_texture_gram_bwd."""
    _chk(x, 'x'); _chk(dgram, 'dgram')
    n, h, w, c = x.shape
    if tuple(dgram.shape) != (n * (h // 16) * (w // 16), c, c):
        raise ValueError('texture_gram_bwd: dgram %s does not fit x %s' % (tuple(dgram.shape), tuple(x.shape)))
    out = out if out is not None else torch.empty_like(x)
    check(lib().srx_texture_gram_bwd(_ptr(x), _ptr(dgram), _ptr(out), n, h, w, c, eps, alpha, _stream()), 'srx_texture_gram_bwd')
    return out


def maxpool2x2(x, out=None):
    """tf.nn.max_pool(ksize 2, strides 2, 'SAME'): [N,H,W,C] -> [N,ceil(H/2),ceil(W/2),C]."""
    _chk(x, 'x')
    N, H, W, C = x.shape
    out = out if out is not None else torch.empty((N, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
    check(lib().srx_maxpool2x2(_ptr(x), _ptr(out), N, H, W, C, _stream()), 'srx_maxpool2x2')
    return out


def maxpool2x2_bwd(x, dout, out=None, mask_act=None):
    """Gradient of 2x2 / 2 SAME max-pooling w.r.t. its input x; mask_act: also multiply by act'(x), the gradient of the
    activation that produced x (one pass instead of maxpool2x2_bwd + act_bwd, same bits)."""
    _chk(x, 'x'); _chk(dout, 'dout')
    N, H, W, C = x.shape
    out = out if out is not None else torch.empty_like(x)
    check(lib().srx_maxpool2x2_bwd_masked(_ptr(x), _ptr(dout), _ptr(out), N, H, W, C, ACT_BY_NAME[mask_act], _stream()),
          'srx_maxpool2x2_bwd_masked')
    return out


def subsample2(x, oy=1, ox=1, out=None):
    """x[:, oy::2, ox::2, :] -- with (1, 1): a stride-1 SAME 3x3 convolution's output -> the stride-2 layer's."""
    _chk(x, 'x')
    N, H, W, C = x.shape
    out = out if out is not None else torch.empty((N, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    check(lib().srx_subsample2(_ptr(x), _ptr(out), N, H, W, C, oy, ox, _stream()), 'srx_subsample2')
    return out


def subsample2_bwd(dout, oy=1, ox=1, out=None):
    """Zero stuffing: [N,h,w,C] -> [N,2h,2w,C] with dout at the (oy, ox) positions."""
    _chk(dout, 'dout')
    N, h, w, C = dout.shape
    out = out if out is not None else torch.empty((N, 2 * h, 2 * w, C), dtype=torch.float32, device=dout.device)
    check(lib().srx_subsample2_bwd(_ptr(dout), _ptr(out), N, 2 * h, 2 * w, C, oy, ox, _stream()), 'srx_subsample2_bwd')
    return out


def blocks_to_nhwc(blocked, out=None):
    """[CB, N, H, W, 64] -> [N, H, W, CB*64]."""
    _chk(blocked, 'blocked')
    CB, N, H, W, cb = blocked.shape
    if CB == 1:
        return blocked[0]
    if cb != 64:
        raise ValueError('channel blocks must hold 64 channels')
    out = out if out is not None else torch.empty((N, H, W, CB * 64), dtype=torch.float32, device=blocked.device)
    check(lib().srx_channel_blocks_to_nhwc(_ptr(blocked), _ptr(out), N * H * W, CB, _stream()), 'srx_channel_blocks_to_nhwc')
    return out


def nhwc_to_blocks(plain, out=None):
    """[N, H, W, C] -> [C/64, N, H, W, 64] (C <= 64: a view [1, N, H, W, C])."""
    _chk(plain, 'plain')
    N, H, W, C = plain.shape
    if C <= 64:
        return plain.view(1, N, H, W, C)
    if C % 64:
        raise ValueError('more than 64 channels must come in multiples of 64')
    out = out if out is not None else torch.empty((C // 64, N, H, W, 64), dtype=torch.float32, device=plain.device)
    check(lib().srx_nhwc_to_channel_blocks(_ptr(plain), _ptr(out), N * H * W, C // 64, _stream()), 'srx_nhwc_to_channel_blocks')
    return out


def channel_normalize(x, eps=1e-6, out=None):
    """enet normalize(): x / (mean over the last axis + eps)."""
    _chk(x, 'x')
    C = x.shape[-1]
    out = out if out is not None else torch.empty_like(x)
    check(lib().srx_channel_normalize(_ptr(x), _ptr(out), x.numel() // C, C, float(eps), _stream()), 'srx_channel_normalize')
    return out


def channel_normalize_bwd(x, dy, eps=1e-6, out=None):
    _chk(x, 'x'); _chk(dy, 'dy')
    C = x.shape[-1]
    out = out if out is not None else torch.empty_like(x)
    check(lib().srx_channel_normalize_bwd(_ptr(x), _ptr(dy), _ptr(out), x.numel() // C, C, float(eps), _stream()),
          'srx_channel_normalize_bwd')
    return out


def extract_patches16(x, out=None):
    """[N,H,W,C] -> [N, (H/16)*(W/16), 256, C] (tf.extract_image_patches 16x16 / 16, reshaped)."""
    _chk(x, 'x')
    N, H, W, C = x.shape
    out = out if out is not None else torch.empty((N, (H // 16) * (W // 16), 256, C), dtype=torch.float32, device=x.device)
    check(lib().srx_extract_patches16(_ptr(x), _ptr(out), N, H, W, C, 0, _stream()), 'srx_extract_patches16')
    return out


def extract_patches16_bwd(dpatches, image_shape, out=None):
    _chk(dpatches, 'dpatches')
    N, H, W, C = image_shape
    out = out if out is not None else torch.empty((N, H, W, C), dtype=torch.float32, device=dpatches.device)
    check(lib().srx_extract_patches16(_ptr(dpatches), _ptr(out), N, H, W, C, 1, _stream()), 'srx_extract_patches16')
    return out


def log_loss(p, label, loss_out, loss_scale=1.0, grad_scale=1.0, accumulate=False, want_grad=True, eps=1e-7):
    """loss_out (+)= loss_scale * tf.losses.log_loss(label, p); returns dp * grad_scale (or None)."""
    _chk(p, 'p')
    dp = torch.empty_like(p) if want_grad else None
    check(lib().srx_log_loss(_ptr(p), float(label), p.numel(), float(eps), float(loss_scale), float(grad_scale),
                             _ptr(loss_out), int(accumulate), _ptr(dp), _stream()), 'srx_log_loss')
    return dp


def vgg_preprocess(x, backward=False, out=None):
    """[N,H,W,3] in [-1,1] RGB -> BGR 0..255 minus the ImageNet means; backward=True maps the gradient back."""
    _chk(x, 'x')
    out = out if out is not None else torch.empty_like(x)
    check(lib().srx_vgg_preprocess(_ptr(x), _ptr(out), x.numel() // 3, int(backward), _stream()), 'srx_vgg_preprocess')
    return out


def add_scaled(a, b=None, alpha=1.0, beta=1.0, out=None):
    """alpha * a + beta * b (b None: alpha * a); out may be a or b."""
    _chk(a, 'a'); _chk(b, 'b')
    if b is not None and a.shape != b.shape:
        raise ValueError('add_scaled: shapes differ')
    out = out if out is not None else torch.empty_like(a)
    check(lib().srx_add_scaled(_ptr(a), _ptr(b), _ptr(out), a.numel(), float(alpha), float(beta), _stream()), 'srx_add_scaled')
    return out


def column_sums(a, out=None):
    _chk(a, 'a')
    rows, cols = a.shape
    out = out if out is not None else torch.empty((cols,), dtype=torch.float32, device=a.device)
    check(lib().srx_column_sums(_ptr(a), _ptr(out), rows, cols, cols, _stream()), 'srx_column_sums')
    return out


_gemm_ws = {}


def gemm(A, B, bias=None, act=None, alpha=1.0, trans_a=False, trans_b=False, out=None, accumulate=False):
    """C = act(alpha * op(A) @ op(B) + bias); A, B 2-D, or 3-D with a leading batch dimension (same batch count).
    Exact fp32 on the MFMA unit -- srx_gemm."""
    _chk(A, 'A'); _chk(B, 'B'); _chk(bias, 'bias')
    batched = A.dim() == 3
    if batched != (B.dim() == 3):
        raise ValueError('A and B must both be batched or both be matrices')
    a2, b2 = (A.shape[1:], B.shape[1:]) if batched else (A.shape, B.shape)
    M, K = (a2[1], a2[0]) if trans_a else (a2[0], a2[1])
    Kb, N = (b2[1], b2[0]) if trans_b else (b2[0], b2[1])
    if K != Kb:
        raise ValueError('inner dimensions differ: %d vs %d' % (K, Kb))
    batch = A.shape[0] if batched else 1
    shape = (batch, M, N) if batched else (M, N)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=A.device)
    elif tuple(out.shape) != shape:
        raise ValueError('out has shape %s, expected %s' % (tuple(out.shape), shape))
    _chk(out, 'out')
    d = _lib.GemmDesc(M, N, K, batch,
                      1 if trans_a else a2[1], a2[1] if trans_a else 1, a2[0] * a2[1] if batched else 0,
                      1 if trans_b else b2[1], b2[1] if trans_b else 1, b2[0] * b2[1] if batched else 0,
                      N, 1, M * N if batched else 0, float(alpha), ACT_BY_NAME[act], int(accumulate))
    need = lib().srx_gemm_workspace_bytes(M, N, K, batch)
    ws = None
    if need:
        key = (A.device.index, torch.cuda.current_stream(A.device).cuda_stream)
        ws = _gemm_ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            ws = _gemm_ws[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=A.device)
    check(lib().srx_gemm(ctypes.byref(d), _ptr(A), _ptr(B), _ptr(bias), _ptr(out), _ptr(ws), ws.numel() * 4 if ws is not None else 0,
                         _stream()), 'srx_gemm')
    return out
