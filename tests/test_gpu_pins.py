"""P7 on the GPU: the reference's own VDSR feature maps (assets/vdsr-fig2-*.png, committed as uint8 corner crops with the
weights fitted from the rest of the image -- tests/golden/make_pin_p7.py) reproduced by the HIP convolution through the
C ABI: srx_conv2d_fwd (both kernel families; conv_narrow_kernel for the 64 -> 3 output layer) + srx_saturate_u8."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.test_gpu_ops import conv_path, dev, ops  # noqa: F401  (fixtures)
from tests.test_oracle_pins import P7_DECISIVE, p7_decode, p7_errors, p7_fitted, p7_load, p7_masks, p7_predict

pytestmark = pytest.mark.gpu


def gpu_conv(ops):
    def conv(x, w, b, padding, act):
        y = ops.conv2d_fwd(dev(x), dev(w), None if b is None else dev(b), padding, act)
        return y.cpu().numpy()
    return conv


@pytest.mark.parametrize('n', P7_DECISIVE)
def test_p7_hip_conv_reproduces_the_reference_feature_maps(n, ops, conv_path):
    """vdsr/vdsr/model_vdsr.py:47-106: decode the reference's crop of layer n - 1, convolve on the GPU with zero SAME
    padding + bias + ReLU, encode with the truncating cast: the reference's bytes of layer n, image border included."""
    z = p7_load()
    cont = p7_predict(z, n, gpu_conv(ops))
    # same numbers as the float64 oracle on the same inputs (the north star's bound) ...
    ref_cont = p7_predict(z, n, O.conv2d_fwd)
    assert np.abs(cont - ref_cont).max() <= 1e-3 * np.abs(ref_cont - 127.5).max()
    # ... and therefore the same distance to the reference's maps: quantisation noise, border included
    inner, brd = p7_errors(z, n, cont)
    assert inner <= 0.5 and brd <= 0.65, (n, inner, brd)
    # bytes, through srx_saturate_u8 (x * 127.5 + 127.5, truncated): within one level of the reference's PNG
    x = p7_decode(z['sd'] if n == 1 else z['conv%d' % (n - 1)])
    y = ops.conv2d_fwd(dev(x), dev(z['w%d' % n]), dev(z['b%d' % n]), 'SAME', 'relu' if n < 20 else None)
    enc = ops.saturate_u8(y).cpu().numpy().astype(np.int64)
    valid, _ = p7_masks(int(z['corner']))
    ok = (np.abs(enc - z['conv%d' % n].astype(np.int64)) <= 1)[valid][:, p7_fitted(z, n)]
    assert ok.mean() >= (0.999 if n in (1, 20) else 0.95), (n, ok.mean())
    # edge padding is NOT what the reference did (same statement as the CPU test, on the device's numbers)
    _, brd_edge = p7_errors(z, n, p7_predict(z, n, gpu_conv(ops), 'edge'))
    assert brd_edge >= 1.75 * brd, (n, brd_edge, brd)


def test_p7_output_layer_residual_add_on_the_gpu(ops, conv_path):
    """model_vdsr.py:85-106: sr = sd + conv3x3(relu.19; 64 -> 3) as ONE launch (the skip operand of the narrow
    kernel), encoded: the reference's sr_image bytes (P3's chain, from the reference's own conv.19 crop)."""
    z = p7_load()
    x = dev(p7_decode(z['conv19']))
    sd = dev(p7_decode(z['sd']))
    sr = ops.conv2d_fwd(x, dev(z['w20']), dev(z['b20']), 'SAME', None, skip=sd)
    enc = ops.saturate_u8(sr).cpu().numpy().astype(np.int64)
    valid, _ = p7_masks(int(z['corner']))
    d = np.abs(enc - z['sr'].astype(np.int64))[valid]
    assert (d <= 1).mean() >= 0.99 and d.max() <= 2, ((d <= 1).mean(), d.max())


def test_p7_the_whole_sr_image_on_the_gpu(ops, conv_path):
    """The reference's ENTIRE sr_image.png from its conv.19.png through the C ABI: srx_conv2d_fwd (3x3 64 -> 3 with the
    residual operand: conv_narrow_kernel, or the MFMA kernel on the other path) + srx_saturate_u8, 256 x 256, all four
    image borders (every border pixel out of sample)."""
    from tests.test_oracle_pins import p7_full20
    z, f, ring = p7_full20()
    x, sd = dev(p7_decode(f['conv19'])[None]), dev(p7_decode(f['sd'])[None])
    sr = ops.conv2d_fwd(x, dev(z['w20']), dev(z['b20']), 'SAME', None, skip=sd)
    d = np.abs(ops.saturate_u8(sr)[0].cpu().numpy().astype(np.int64) - f['sr'].astype(np.int64))
    assert (d <= 1).mean() >= 0.999 and (d <= 1)[ring].mean() >= 0.995 and d.max() <= 2, ((d <= 1).mean(), (d <= 1)[ring].mean(), d.max())
    res = ops.conv2d_fwd(x, dev(z['w20']), dev(z['b20']), 'SAME', None)
    d20 = np.abs(ops.saturate_u8(res)[0].cpu().numpy().astype(np.int64) - f['conv20'].astype(np.int64))
    assert (d20 <= 1).mean() >= 0.999 and (d20 <= 1)[ring].mean() >= 0.995 and d20.max() <= 2


def test_p7_the_whole_conv1_image_on_the_gpu(ops, conv_path):
    """The reference's ENTIRE conv.1.png (64 maps of 256 x 256) from its sd_image.png through srx_conv2d_fwd (3x3 3 -> 64 +
    bias + ReLU, the RGB-input kernel) + srx_saturate_u8: every byte within one level, >= 97 % exact."""
    import os
    from tests.conftest import GOLDEN
    from tests.test_oracle_pins import p7_full20
    z, f, ring = p7_full20()
    conv1 = np.load(os.path.join(GOLDEN, 'pin_p7_layer1_full.npz'))['conv1'].astype(np.int64)
    y = ops.conv2d_fwd(dev(p7_decode(f['sd'])[None]), dev(z['w1']), dev(z['b1']), 'SAME', 'relu')
    d = np.abs(ops.saturate_u8(y)[0].cpu().numpy().astype(np.int64) - conv1)
    assert d.max() <= 1 and (d == 0).mean() >= 0.97, (d.max(), (d == 0).mean())
