#!/usr/bin/env python3
"""Random-shape parity sweep of round 3's entry points against the oracle (a development tool; the committed cases live
in tests/): stride-2 forward / filter gradient (srx_conv_desc.stride = 2), the sub-pixel maps on every kernel route
(bit-exact), srx_srcnn_forward against the three per-layer launches (bit-identical) and the oracle, the pooling gradient
with the fused activation gradient.
Usage: fuzz_round3.py [cases] [seed] [stride2|subpixel|srcnn|maxpool]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
from oracle import oracle as O
from oracle import oracle_enet as E

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(got, ref, tol=1e-3):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return False
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref)
    return bool(np.isfinite(got).all() and err.max() <= tol * scale and (err <= 1e-5 * scale + 1e-4 * np.abs(ref)).all())


S2_LAYERS = [(3, 32, 32), (3, 64, 64), (3, 64, 32), (3, 32, 64), (3, 3, 32), (3, 3, 64), (3, 64, 48), (1, 64, 64), (1, 64, 32), (3, 64, 16)]


def case_stride2(rng):
    k, cin, cout = S2_LAYERS[rng.integers(len(S2_LAYERS))]
    pad = 'SAME' if rng.random() < 0.75 else 'VALID'
    act = [None, 'relu', 'lrelu', 'tanh'][rng.integers(4)]
    n = int(rng.integers(1, 5))
    h = int(rng.integers(k if pad == 'VALID' else 1, 70))
    w = int(rng.integers(k if pad == 'VALID' else 1, 150 if rng.random() < 0.3 else 50))
    x = rng.uniform(-1, 1, (n, h, w, cin)).astype(np.float32)
    wt = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
    desc = 'stride2 k%d %d->%d %s %s %dx%dx%d' % (k, cin, cout, pad, act, n, h, w)
    y = ops.conv2d_fwd(dev(x), dev(wt), dev(b), pad, act, stride=2)
    pre = E.conv2d_same_fwd(x, wt, b, 2) if pad == 'SAME' else O.conv2d_fwd(x, wt, b, 'VALID')[:, ::2, ::2]
    ok = close(y, O.act_apply(pre, act))
    dpre = rng.normal(size=pre.shape).astype(np.float32)
    dw, db = ops.conv2d_bwd_filter(dev(x), dev(dpre), wt.shape, pad, stride=2)
    if pad == 'SAME':
        _, dw_ref, db_ref = E.conv2d_same_bwd(x, wt, dpre, 2, want_dx=False)
    else:
        stuffed = np.zeros((n, h - k + 1, w - k + 1, cout), np.float32)
        stuffed[:, ::2, ::2] = dpre
        dw_ref, db_ref = O.conv2d_bwd_filter(x, stuffed, (k, k), 'VALID')
    ok = ok and close(dw, dw_ref) and close(db, db_ref)
    return ok, desc


def case_subpixel(rng):
    r, c = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    kind = rng.random()
    if kind < 0.6:
        n, h, w = int(rng.integers(1, 6)), int(rng.integers(1, 60)), int(rng.integers(1, 120))
    elif kind < 0.8:
        n, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 6)), int(rng.integers(500, 5000))      # long rows
    else:
        n, h, w = int(rng.integers(50, 600)), int(rng.integers(20, 60)), int(rng.integers(20, 60))    # many chunks per workgroup
    bits = rng.integers(0, 1 << 32, size=(n, h, w, c * r * r), dtype=np.uint64).astype(np.uint32)
    t = torch.from_numpy(bits.view(np.int32)).cuda().view(torch.float32)
    d = ops.depth_to_space(t, r)
    step = 1 if n * h * w < 300000 else 41
    ok = np.array_equal(d[::step].view(torch.int32).cpu().numpy().view(np.uint32), O.depth_to_space(bits[::step], r))
    ok = ok and torch.equal(ops.space_to_depth(d, r).view(torch.int32), t.view(torch.int32))
    return ok, 'subpixel %dx%dx%dx%d r%d' % (n, h, w, c, r)


_srcnn = {}


def case_srcnn(rng):
    if not _srcnn:
        g = torch.Generator(device='cuda').manual_seed(5)
        rnd = lambda *s, sc=1.0: (torch.rand(s, device='cuda', generator=g) * 2 - 1) * sc
        _srcnn['p'] = [(rnd(9, 9, 3, 64, sc=0.06), rnd(64, sc=0.1)), (rnd(1, 1, 64, 32, sc=0.12), rnd(32, sc=0.1)), (rnd(5, 5, 32, 3, sc=0.03), rnd(3, sc=0.1))]
    p = _srcnn['p']
    n = int(rng.integers(1, 4))
    h, w = int(rng.integers(13, 90)), int(rng.integers(13, 90))
    x = rng.uniform(-1, 1, (n, h, w, 3)).astype(np.float32)
    xd = dev(x)
    one = ops.srcnn_forward(xd, p)

    def per_layer():
        t = ops.conv2d_fwd(xd, p[0][0], p[0][1], 'valid', 'relu')
        t = ops.conv2d_fwd(t, p[1][0], p[1][1], 'valid', 'relu')
        return ops.conv2d_fwd(t, p[2][0], p[2][1], 'valid', 'tanh')
    three = per_layer()
    # (since round 4 the default path runs the 5x5 32 -> 3 layer on conv_kwrows_kernel from 4,096 output pixels -- equal to
    # rounding; the one-launch kernel is bit-identical to conv path 0)
    from ml_super_resolution_amd import _lib
    old = _lib.lib().srx_set_conv_path(0)
    try:
        three0 = per_layer()
    finally:
        _lib.lib().srx_set_conv_path(old)
    ok = torch.equal(one, three0) and float((one - three).abs().max()) <= 2e-6 * max(1.0, float(three.abs().max()))
    ok = ok and close(one, O.srcnn_forward(x, [(k.cpu().numpy(), b.cpu().numpy()) for k, b in p]))
    return ok, 'srcnn %dx%dx%d' % (n, h, w)


def case_maxpool(rng):
    n, h, w, c = int(rng.integers(1, 4)), int(rng.integers(1, 40)), int(rng.integers(1, 40)), 4 * int(rng.integers(1, 20))
    act = ['relu', 'lrelu', 'tanh', None][rng.integers(4)]
    x = rng.normal(size=(n, h, w, c)).astype(np.float32)
    if act == 'relu':
        x = np.maximum(x, 0) * (rng.uniform(size=x.shape) > 0.4)
    dy = rng.normal(size=(n, (h + 1) // 2, (w + 1) // 2, c)).astype(np.float32)
    got = ops.maxpool2x2_bwd(dev(x), dev(dy), mask_act=act)
    two = ops.maxpool2x2_bwd(dev(x), dev(dy))
    if act is not None:
        two = ops.act_bwd(two, dev(x), act)
    ok = torch.equal(got, two) and np.array_equal(ops.maxpool2x2_bwd(dev(x), dev(dy)).cpu().numpy(), E.maxpool2x2_bwd(x, dy).astype(np.float32))
    return ok, 'maxpool %dx%dx%dx%d %s' % (n, h, w, c, act)


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    kinds = [case_stride2, case_subpixel, case_srcnn, case_maxpool]
    if len(sys.argv) > 3:
        kinds = [k for k in kinds if k.__name__ == 'case_' + sys.argv[3]]
    nbad = 0
    for it in range(cases):
        ok, desc = kinds[it % len(kinds)](rng)
        if not ok:
            nbad += 1
            print('BAD', desc, flush=True)
        if it % 50 == 49:
            print('... %d cases, %d bad' % (it + 1, nbad), flush=True)
    print('fuzz_round3: %d cases, %d bad' % (cases, nbad))
    sys.exit(1 if nbad else 0)
