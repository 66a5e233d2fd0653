"""Filter gradient of the 3x3 64 -> 64 layers on images too wide for full-width tiles (the reference's own VDSR
recipe, vdsr/makefile:22-29: batch 64 of 128 x 128 patches): wgrad_rows_strip_kernel (one workgroup per CU, two LDS tile
buffers, 32-column strips, windows of two strip rows = 16 steps over real pixels only) against the oracle, beside the
two-workgroup strip kernel it replaces, and, at the recipe's size, through size-independent properties."""
import zlib

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.test_gpu_ops import close, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from ml_super_resolution_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def _with_wgrad_path(path, fn):
    from ml_super_resolution_amd import _lib
    old = _lib.lib().srx_set_wgrad_path(path)
    try:
        return fn()
    finally:
        _lib.lib().srx_set_wgrad_path(old)


# (N, H, W, padding): strips of 32 columns; widths whose last strip is 4..31 columns wide or exactly full, heights that
# are / are not multiples of the 6-row tile, one row, workgroup ranges that cut strips and images, VALID geometry (no pad
# column: the strip's first slot is a real pixel), last strips of 1 and 2 columns (the padded walk declined those)
STRIP_SHAPES = [
    (2, 33, 64, 'SAME'), (1, 9, 61, 'SAME'), (1, 12, 80, 'VALID'), (5, 4, 203, 'SAME'), (3, 21, 100, 'SAME'),
    (1, 6, 128, 'SAME'), (1, 1, 96, 'SAME'), (2, 7, 68, 'SAME'), (1, 40, 95, 'SAME'), (3, 18, 132, 'VALID'),
    (1, 50, 63, 'SAME'), (7, 5, 70, 'SAME'), (1, 300, 64, 'SAME'), (2, 13, 65, 'SAME'), (1, 128, 128, 'SAME'),
    (40, 12, 64, 'SAME'), (2, 9, 34, 'VALID'), (1, 5, 66, 'SAME'), (3, 2, 97, 'SAME'),
]


@pytest.mark.parametrize('shape', STRIP_SHAPES, ids=['%dx%dx%d_%s' % s for s in STRIP_SHAPES])
def test_strip_filter_gradient_vs_oracle_and_vs_two_workgroup_kernel(shape, ops):
    N, H, W, pad = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()))
    x = rng.uniform(-1, 1, (N, H, W, 64)).astype(np.float32)
    oh, ow = (H, W) if pad == 'SAME' else (H - 2, W - 2)
    dpre = rng.normal(0, 1, (N, oh, ow, 64)).astype(np.float32)
    w = rng.normal(0, 0.05, (3, 3, 64, 64)).astype(np.float32)
    dw_ref, db_ref = O.c_conv2d_bwd_filter(x, dpre, (3, 3), pad)
    xd, dd, wd = dev(x), dev(dpre), dev(w)
    got = {}
    for path in (2, 1):
        dw, db = _with_wgrad_path(path, lambda: ops.conv2d_bwd_filter(xd, dd, wd.shape, pad, w_for_decay=wd, wd_scale=1e-4))
        close(dw, dw_ref + 1e-4 * w)
        close(db, db_ref)
        dw2, db2 = _with_wgrad_path(path, lambda: ops.conv2d_bwd_filter(xd, dd, wd.shape, pad, w_for_decay=wd, wd_scale=1e-4))
        assert torch.equal(dw, dw2) and torch.equal(db, db2)          # deterministic
        got[path] = (dw, db)
    # (No bit-for-bit comparison between the two: the exact-rows kernel adds a strip row's 32 real positions in 8 steps of
    # 4, the padded walk of the two-workgroup kernel groups the 34 positions of a tile row -- 2 of them fake -- across row
    # ends; the MFMA adds the 4 positions of a step in one go, so the groupings round differently.  Both are compared with
    # the float64 oracle element by element, and each with itself for determinism.)
    ratio = (got[1][0].double() - got[2][0].double()).abs().max().item() / max(got[2][0].abs().max().item(), 1e-30)
    assert ratio < 2e-6


def test_strip_filter_gradient_nonfinite_free_and_zero_operands(ops):
    """dpre == 0 gives exactly zero gradients whatever x holds; the pad / fake positions never leak into the sums (x = 1,
    dpre = 1: every filter tap's gradient is the number of valid (pixel, tap) pairs, an integer below 2^24)."""
    N, H, W = 2, 20, 96
    xd = torch.ones((N, H, W, 64), device='cuda')
    dd = torch.ones((N, H, W, 64), device='cuda')
    dw, db = ops.conv2d_bwd_filter(xd, dd, (3, 3, 64, 64), 'same')
    counts = np.array([[(H - abs(kh - 1)) * (W - abs(kw - 1)) for kw in range(3)] for kh in range(3)], np.float64) * N
    np.testing.assert_array_equal(dw.cpu().numpy(), np.broadcast_to(counts[:, :, None, None], (3, 3, 64, 64)))
    np.testing.assert_array_equal(db.cpu().numpy(), np.full((64,), N * H * W, np.float32))
    dw0, db0 = ops.conv2d_bwd_filter(torch.randn((N, H, W, 64), device='cuda'), torch.zeros_like(dd), (3, 3, 64, 64), 'same')
    assert not dw0.any() and not db0.any()


def test_vdsr_recipe_size_filter_gradient_properties(ops):
    """The reference's recipe shape, batch 64 of 128 x 128 x 64 (vdsr/makefile:22-29): too large for the CPU oracle in a
    test, so (1) the gradient of the batch equals the sum of the gradients of its four quarters (other workgroup ranges,
    other partial sums: agreement to fp32 rounding of sums of ~1e6 terms), (2) a slice of 2 images against the oracle,
    (3) determinism."""
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.rand((64, 128, 128, 64), device='cuda', generator=g) * 2 - 1
    dpre = torch.randn((64, 128, 128, 64), device='cuda', generator=g)
    dw, db = ops.conv2d_bwd_filter(x, dpre, (3, 3, 64, 64), 'same')
    dw_b, db_b = ops.conv2d_bwd_filter(x, dpre, (3, 3, 64, 64), 'same')
    assert torch.equal(dw, dw_b) and torch.equal(db, db_b)
    acc_w = torch.zeros_like(dw, dtype=torch.float64)
    acc_b = torch.zeros_like(db, dtype=torch.float64)
    for q in range(4):
        dwq, dbq = ops.conv2d_bwd_filter(x[16 * q:16 * q + 16], dpre[16 * q:16 * q + 16], (3, 3, 64, 64), 'same')
        acc_w += dwq.double(); acc_b += dbq.double()
    scale = dw.abs().max().item()
    assert (dw.double() - acc_w).abs().max().item() <= 2e-5 * scale
    assert (db.double() - acc_b).abs().max().item() <= 2e-5 * db.abs().max().item() + 1e-2
    xs, ds = x[30:32].cpu().numpy(), dpre[30:32].cpu().numpy()
    dw_ref, db_ref = O.c_conv2d_bwd_filter(xs, ds, (3, 3), 'SAME')
    dws, dbs = ops.conv2d_bwd_filter(x[30:32], dpre[30:32], (3, 3, 64, 64), 'same')
    close(dws, dw_ref)
    close(dbs, db_ref)


# ---- 41-pixel rows (the VDSR patch of BASELINE's metric) on full-width tiles: wgrad_rows_full_kernel -----------------------
ROWS41_SHAPES = [(1, 41), (3, 41), (1, 1), (1, 2), (2, 3), (1, 4), (5, 5), (2, 40), (7, 13), (37, 41), (300, 6), (1, 200)]


@pytest.mark.parametrize('shape', ROWS41_SHAPES, ids=['%dx%dx41' % s for s in ROWS41_SHAPES])
def test_rows41_filter_gradient_vs_oracle_and_vs_padded_walk(shape, ops):
    """3x3 64 -> 64 SAME on 41-pixel rows: units of 3 rows, one 31-step window each (30 row steps of 4 real pixels + one
    column step taking the last column of the unit's rows), against the float64 oracle and beside the padded-position walk
    (wgrad_pipe_kernel, srx_set_wgrad_path(2) with SRX_WGRAD_ROWS_FULL=0 is not reachable at run time: path 1, the
    two-workgroup walk, is the comparison).  Heights of 1, 2 (short units only), 3, 4, 5 (one full + one short unit), 40, 41, 200;
    batches whose workgroup ranges cut images; many small images."""
    N, H = shape
    rng = np.random.default_rng(zlib.crc32(repr(('rows41',) + shape).encode()))
    x = rng.uniform(-1, 1, (N, H, 41, 64)).astype(np.float32)
    dpre = rng.normal(0, 1, (N, H, 41, 64)).astype(np.float32)
    dw_ref, db_ref = O.c_conv2d_bwd_filter(x, dpre, (3, 3), 'SAME')
    xd, dd = dev(x), dev(dpre)
    dw, db = ops.conv2d_bwd_filter(xd, dd, (3, 3, 64, 64), 'same')
    close(dw, dw_ref)
    close(db, db_ref)
    dw2, db2 = ops.conv2d_bwd_filter(xd, dd, (3, 3, 64, 64), 'same')
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dw1, db1 = _with_wgrad_path(1, lambda: ops.conv2d_bwd_filter(xd, dd, (3, 3, 64, 64), 'same'))
    close(dw1, dw_ref)
    assert (dw1.double() - dw.double()).abs().max().item() <= 2e-6 * max(dw.abs().max().item(), 1e-30)


def test_rows41_exact_integer_sums(ops):
    """x = 1, dpre = 1 on 41-pixel rows: every tap's gradient is the number of valid (pixel, tap) pairs -- the column step and
    the short last unit (41 = 13 x 3 + 2) must count every pixel exactly once."""
    N, H, W = 3, 41, 41
    xd = torch.ones((N, H, W, 64), device='cuda')
    dd = torch.ones((N, H, W, 64), device='cuda')
    dw, db = ops.conv2d_bwd_filter(xd, dd, (3, 3, 64, 64), 'same')
    counts = np.array([[(H - abs(kh - 1)) * (W - abs(kw - 1)) for kw in range(3)] for kh in range(3)], np.float64) * N
    np.testing.assert_array_equal(dw.cpu().numpy(), np.broadcast_to(counts[:, :, None, None], (3, 3, 64, 64)))
    np.testing.assert_array_equal(db.cpu().numpy(), np.full((64,), N * H * W, np.float32))
    # a position-dependent dpre: channel-wise sums of an integer ramp
    ramp = torch.arange(N * H * W, device='cuda', dtype=torch.float32).remainder(7).view(N, H, W, 1).expand(N, H, W, 64).contiguous()
    _, db2 = ops.conv2d_bwd_filter(xd, ramp, (3, 3, 64, 64), 'same')
    assert torch.equal(db2, ramp.sum(dim=(0, 1, 2)))
