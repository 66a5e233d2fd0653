"""The sub-pixel (depth-to-space) index map as the STORE MODE of a convolution (srx_conv_desc.subpixel_r) and the
ESPCN inference path built on it: bit-identical to conv -> srx_depth_to_space (espcn/espcn/model_espcn.py:117-134 +
espcn/espcn/experiment_test.py:171-177), <= 1e-3 against the oracle, at exactly BASELINE configs[1]'s shape
[32,17,17,3] r=3 and at the bandwidth shape [256,41,41,.]."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.test_gpu_ops import close, dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,cin,r,k', [((32, 17, 17), 32, 3, 3), ((256, 41, 41), 32, 3, 3), ((3, 9, 13), 32, 4, 3),
                                           ((2, 7, 5), 32, 2, 3), ((2, 11, 6), 64, 3, 3), ((1, 20, 33), 3, 3, 5),
                                           ((2, 6, 6), 32, 3, 1), ((1, 100, 203), 32, 3, 3), ((2, 37, 96), 32, 3, 3), ((1, 300, 260), 32, 3, 3),
                                           ((3, 5, 61), 32, 3, 3), ((1, 70, 130), 32, 2, 3), ((1, 64, 128), 32, 4, 3)],
                         ids=['c2_f3', 'bandwidth_shape', 'r4', 'r2', 'cin64', 'rgb_5x5', '1x1', 'wide_1x100x203', 'wide_2x37x96', 'wide_1x300x260',
                              'wide_short_3x5x61', 'wide_r2', 'wide_r4'])
def test_conv_with_subpixel_store_equals_conv_then_d2s(shape, cin, r, k):
    """(The wide shapes: images too wide for full-width tiles.  Since round 4 the 32 -> 27 layer of r = 3 runs them on the
    pipelined column-strip kernel with the sub-pixel map as a form of its deferred epilogue -- four 4-byte stores per lane,
    channels 27..31 out of range; widths that are / are not multiples of the 32-column strip, tiles cut short by the image
    and by the workgroup's range; r = 2 / 4 stay on the other kernels.)"""
    from ml_super_resolution_amd import ops
    n, h, w = shape
    g = torch.Generator(device='cuda').manual_seed(h * 100 + w)
    x = torch.rand((n, h, w, cin), device='cuda', generator=g) * 2 - 1
    wt = (torch.rand((k, k, cin, 3 * r * r), device='cuda', generator=g) * 2 - 1) * 0.1
    b = torch.rand((3 * r * r,), device='cuda', generator=g) - 0.5
    for act in (None, 'tanh'):
        two = ops.depth_to_space(ops.conv2d_fwd(x, wt, b, 'same', act), r)
        fused = ops.conv2d_fwd(x, wt, b, 'same', act, subpixel_r=r)
        assert fused.shape == (n, h * r, w * r, 3)
        assert torch.equal(fused, two)                     # a pure change of store addresses: bit for bit
    if n * h * w <= 40000:
        ref = O.depth_to_space(O.conv2d_fwd(x.cpu().numpy(), wt.cpu().numpy(), b.cpu().numpy(), 'SAME', None), r)
        close(ops.conv2d_fwd(x, wt, b, 'same', None, subpixel_r=r), ref)
    # the same bits on conv path 0 (conv_mfma_kernel for every shape)
    from ml_super_resolution_amd import _lib
    old = _lib.lib().srx_set_conv_path(0)
    try:
        y0 = ops.conv2d_fwd(x, wt, b, 'same', None, subpixel_r=r)
    finally:
        _lib.lib().srx_set_conv_path(old)
    assert torch.equal(ops.conv2d_fwd(x, wt, b, 'same', None, subpixel_r=r), y0)
    # VALID padding goes through the same store
    if k > 1 and h > k and w > k:
        two = ops.depth_to_space(ops.conv2d_fwd(x, wt, b, 'valid', None), r)
        assert torch.equal(ops.conv2d_fwd(x, wt, b, 'valid', None, subpixel_r=r), two)


def test_subpixel_store_argument_errors():
    from ml_super_resolution_amd import ops
    from ml_super_resolution_amd._lib import SrxError
    x = torch.zeros((1, 8, 8, 32), device='cuda')
    w = torch.zeros((3, 3, 32, 27), device='cuda')
    with pytest.raises(ValueError):
        ops.conv2d_fwd(x, w, None, 'same', None, subpixel_r=2)            # 27 is not a multiple of 4
    with pytest.raises(SrxError, match='skip'):
        ops.conv2d_fwd(x, w, None, 'same', None, skip=torch.zeros((1, 24, 24, 3), device='cuda'), subpixel_r=3)
    with pytest.raises(ValueError, match='out has shape'):
        ops.conv2d_fwd(x, w, None, 'same', None, out=torch.zeros((1, 8, 8, 27), device='cuda'), subpixel_r=3)


@pytest.mark.parametrize('n,h,w,r', [(32, 17, 17, 3), (256, 41, 41, 3), (1, 64, 48, 4)], ids=['config2', 'b256x41', 'image_r4'])
def test_espcn_inference_paths_agree(n, h, w, r):
    """EspcnModel.super_resolve -- three launches with the fused store, replayed as a HIP graph -- against the same
    launches issued eagerly and against forward + standalone depth-to-space (four launches): bit-identical; against
    the oracle: <= 1e-3 (elementwise bound of tests/test_gpu_ops.close)."""
    from ml_super_resolution_amd.espcn import model_espcn
    m = model_espcn.EspcnModel(r, device='cuda', seed=103)
    for i in range(3):
        m.stack.bias(i).uniform_(-0.1, 0.1)
    g = torch.Generator(device='cuda').manual_seed(102)
    x = torch.rand((n, h, w, 3), device='cuda', generator=g) * 2 - 1
    two = m.super_resolve_two_step(x).clone()
    eager = m.super_resolve(x, use_graph=False, single_launch=False).clone()
    graph1 = m.super_resolve(x, use_graph=True, single_launch=False).clone()
    assert two.shape == (n, h * r, w * r, 3)
    assert torch.equal(eager, two) and torch.equal(graph1, two)
    # replay with new data in another buffer, and after an in-place weight update: the graph follows both
    x2 = torch.rand((n, h, w, 3), device='cuda', generator=g) * 2 - 1
    assert torch.equal(m.super_resolve(x2, use_graph=True, single_launch=False), m.super_resolve_two_step(x2))
    m.stack.kernel(2).mul_(0.5)
    assert torch.equal(m.super_resolve(x2, use_graph=True, single_launch=False), m.super_resolve_two_step(x2))
    if n * h * w <= 10000:
        params = [(m.stack.kernel(i).cpu().numpy(), m.stack.bias(i).cpu().numpy()) for i in range(3)]
        ref = O.depth_to_space(O.espcn_forward(x2.cpu().numpy(), params), r)
        close(m.super_resolve(x2), ref)


@pytest.mark.parametrize('n,h,w,r', [(32, 17, 17, 3), (2, 17, 17, 2), (3, 9, 9, 4), (1, 40, 33, 3), (5, 1, 1, 3), (2, 8, 19, 4),
                                     (1, 10, 10, 3), (1, 236, 250, 3), (2, 160, 180, 2), (1, 200, 240, 4), (300, 13, 15, 3)],
                         ids=['config2', 'r2', 'r4_one_tile', 'image_ragged_tiles', 'one_pixel', 'r4_ragged', 'tiles_5x5',
                              'tiles_16x16_ragged', 'r2_tiles_16x15', 'r4_tiles_14x14', 'two_rounds_of_whole_patches'])
def test_espcn_single_launch_equals_three_launches(n, h, w, r):
    """srx_espcn_forward (the three layers chained through LDS per <= 16x16 tile, one launch) against the per-layer
    launches: the same products in the same order -> bit-identical; and <= 1e-3 (elementwise bound) against the
    oracle.  Shapes: exactly BASELINE configs[1]; every scaling factor; tiles that do not divide the image; images
    smaller than the halo; since round 4 images of up to 59 k pixels, which take the largest tiles (16 x 16: the input halo
    and t2 share their LDS bytes) -- just below the 60 k pixels from which the per-layer route's f1 groups its products
    differently (conv_pack3.hip) -- and more tiles than CUs."""
    from ml_super_resolution_amd.espcn import model_espcn
    m = model_espcn.EspcnModel(r, device='cuda', seed=200 + r)
    for i in range(3):
        m.stack.bias(i).uniform_(-0.1, 0.1)
    g = torch.Generator(device='cuda').manual_seed(n * 1000 + h * 10 + w)
    x = torch.rand((n, h, w, 3), device='cuda', generator=g) * 2 - 1
    one = m.super_resolve(x, single_launch=True).clone()
    three = m.super_resolve(x, use_graph=False, single_launch=False).clone()
    assert one.shape == (n, h * r, w * r, 3)
    assert torch.equal(one, three)
    params = [(m.stack.kernel(i).cpu().numpy(), m.stack.bias(i).cpu().numpy()) for i in range(3)]
    close(one, O.depth_to_space(O.espcn_forward(x.cpu().numpy(), params), r))
    # the default route picks the single launch for small problems and the per-layer launches for large ones
    assert torch.equal(m.super_resolve(x), one)


def test_espcn_graphs_of_several_shapes_own_their_buffers():
    """A replayed graph writes through the pointers it captured: the graphs of different input shapes must not share
    intermediate buffers (the model's shared ones are replaced when another image size comes along)."""
    from ml_super_resolution_amd.espcn import model_espcn
    m = model_espcn.EspcnModel(3, device='cuda', seed=5)
    g = torch.Generator(device='cuda').manual_seed(1)
    xs = [torch.rand(s, device='cuda', generator=g) * 2 - 1 for s in ((2, 30, 31, 3), (1, 64, 40, 3), (3, 20, 20, 3))]
    want = [m.super_resolve_two_step(x).clone() for x in xs]
    for _ in range(2):                       # capture each, then replay each after the others were captured
        for x, w in zip(xs, want):
            assert torch.equal(m.super_resolve(x, use_graph=True, single_launch=False), w)
    junk = [torch.rand((4, 50, 50, 64), device='cuda') for _ in range(4)]      # churn the allocator in between
    for x, w in zip(reversed(xs), reversed(want)):
        assert torch.equal(m.super_resolve(x, use_graph=True, single_launch=False), w)
    del junk


@pytest.mark.parametrize('n,h,w', [(1, 243, 243), (2, 33, 33), (1, 13, 13), (3, 20, 47), (1, 28, 27), (1, 14, 40), (5, 43, 44)],
                         ids=['config1', 'train_patch', 'one_pixel_out', 'ragged_tiles', 'tile_plus_one', 'short_wide', 'two_rounds'])
def test_srcnn_single_launch_equals_three_launches(n, h, w):
    """srx_srcnn_forward (9-1-5 VALID chained through LDS per <= 15x15 output tile, one launch; srcnn/srcnn.py:100-130)
    against the three per-layer launches: the same products in the same order -> bit-identical; and <= 1e-3 (elementwise
    bound) against the oracle.  Shapes: BASELINE configs[0] as the reference crops it (243 -> 231: 16 x 16 tiles, one per
    CU); the reference's 33 x 33 training patch; the smallest image; tiles that do not divide the output; more tiles than
    CUs."""
    from ml_super_resolution_amd import ops
    from ml_super_resolution_amd.srcnn import srcnn as srcnn_mod
    m = srcnn_mod.SrcnnModel(device='cuda', seed=300 + h)
    for i, s in enumerate((40.0, 80.0, 30.0)):             # O(1) activations instead of the reference's sigma 1e-3
        m.stack.kernel(i).mul_(s)
        m.stack.bias(i).uniform_(-0.1, 0.1)
    g = torch.Generator(device='cuda').manual_seed(n * 1000 + h * 10 + w)
    x = torch.rand((n, h, w, 3), device='cuda', generator=g) * 2 - 1
    one = m.forward(x, single_launch=True).clone()
    three = m.forward(x, single_launch=False).clone()
    assert one.shape == (n, h - 12, w - 12, 3)
    # bit-identical to the per-layer launches of conv path 0 (conv_mfma_kernel for every layer); the default path runs the
    # 9x9 and 5x5 layers on kernels that group the products differently (conv_pack3 / conv_kwrows, from 4096 output pixels)
    from ml_super_resolution_amd import _lib
    old = _lib.lib().srx_set_conv_path(0)
    try:
        three0 = m.forward(x, single_launch=False).clone()
    finally:
        _lib.lib().srx_set_conv_path(old)
    assert torch.equal(one, three0), float((one - three0).abs().max())
    assert float((one - three).abs().max()) <= 2e-6 * max(1.0, float(three.abs().max()))
    if n * (h - 12) * (w - 12) < 4096:
        assert torch.equal(one, three)
    params = [(m.stack.kernel(i).cpu().numpy(), m.stack.bias(i).cpu().numpy()) for i in range(3)]
    close(one, O.srcnn_forward(x.cpu().numpy(), params))
    assert float(one.abs().max()) > 1e-3                   # (not a test of zeros; the biases come from the global RNG)
    # the default route: one launch for batches of small patches, the three launches otherwise
    dflt = m.forward(x)
    assert torch.equal(dflt, one) or torch.equal(dflt, three)
    if (h - 12) * (w - 12) > 1024:
        assert torch.equal(dflt, three)
    with pytest.raises(ValueError):
        ops.srcnn_forward(x[:, :12], [(m.stack.kernel(i), m.stack.bias(i)) for i in range(3)])


def test_espcn_720p_routes_equal_conv_path_0():
    """ESPCN 3x on one 720 x 1280 LR frame -- the size the whole-image numbers of DESIGN.md are quoted at: f1 on conv_pack3_kernel,
    f2 on the two-chunk pipelined strips with the tanh epilogue (40 strips x 720 rows), f3 on the same strips with the sub-pixel
    map as epilogue.  Too large for the oracle in a test; the size-independent property: every route is bit-identical to conv
    path 0 (conv_mfma_kernel for all three layers, itself checked against the oracle on smaller shapes), and deterministic."""
    from ml_super_resolution_amd import _lib
    from ml_super_resolution_amd.espcn import model_espcn
    m = model_espcn.EspcnModel(3, device='cuda', seed=9)
    for i in range(3):
        m.stack.bias(i).copy_(torch.linspace(-0.1, 0.1, m.stack.bias(i).numel(), device='cuda'))
    g = torch.Generator(device='cuda').manual_seed(720)
    x = torch.rand((1, 720, 1280, 3), device='cuda', generator=g) * 2 - 1
    hr = m.super_resolve(x, use_graph=False, single_launch=False).clone()
    assert hr.shape == (1, 2160, 3840, 3) and torch.isfinite(hr).all()
    assert torch.equal(m.super_resolve(x, use_graph=True, single_launch=False), hr)
    old = _lib.lib().srx_set_conv_path(0)
    try:
        hr0 = m.super_resolve(x, use_graph=False, single_launch=False).clone()
    finally:
        _lib.lib().srx_set_conv_path(old)
    assert torch.equal(hr, hr0)
    assert float(hr.abs().max()) > 1e-3


def test_srcnn_720p_routes_against_conv_path_0():
    """SRCNN 9-1-5 on one 720 x 1280 frame (the size of DESIGN.md's whole-image numbers): conv_pack3_kernel<9,9> and conv_1x1_kernel
    are bit-identical to conv path 0 layer by layer; conv_kwrows_kernel agrees with it to rounding (another order of the kw
    partial sums); the whole net therefore to <= 2e-6 of its scale."""
    from ml_super_resolution_amd import _lib, ops
    g = torch.Generator(device='cuda').manual_seed(1280)
    rnd = lambda *s, sc=1.0: (torch.rand(s, device='cuda', generator=g) * 2 - 1) * sc
    w1, b1 = rnd(9, 9, 3, 64, sc=0.06), rnd(64, sc=0.1)
    w2, b2 = rnd(1, 1, 64, 32, sc=0.12), rnd(32, sc=0.1)
    w3, b3 = rnd(5, 5, 32, 3, sc=0.03), rnd(3, sc=0.1)
    x = rnd(1, 720, 1280, 3)

    def layers():
        t1 = ops.conv2d_fwd(x, w1, b1, 'valid', 'relu')
        t2 = ops.conv2d_fwd(t1, w2, b2, 'valid', 'relu')
        return t1, t2, ops.conv2d_fwd(t2, w3, b3, 'valid', 'tanh')
    t1, t2, y = layers()
    old = _lib.lib().srx_set_conv_path(0)
    try:
        t1_0, t2_0, y_0 = layers()
    finally:
        _lib.lib().srx_set_conv_path(old)
    assert y.shape == (1, 708, 1268, 3) and torch.isfinite(y).all()
    assert torch.equal(t1, t1_0) and torch.equal(t2, t2_0)
    assert float((y - y_0).abs().max()) <= 2e-6 * max(1.0, float(y_0.abs().max())) and not torch.equal(y, y_0)
