#!/usr/bin/env python3
"""Random-shape parity sweep of round 4's kernels against the oracle (a development tool; the committed cases live in
tests/): the exact-rows strip filter gradient (wgrad_rows_strip_kernel: any width from 59 columns up, any last-strip
width, VALID and SAME, ranges that cut strips and images), the (kw, co)-rows 5x5 32 -> 3 kernel (conv_kwrows_kernel), the
sub-pixel map on shapes whose chunks end within 6 float4 of a multiple of 256 (the LDS slot guard of round 4), and the
device-state Adam against the host-argument one.
Late in the round: the RGB-input packed-K kernel (oracle + bit-identity with conv path 0), the two-chunk pipelined strips (tanh epilogue;
sub-pixel epilogue), the one-launch ESPCN kernel with tiles up to 16 x 16, the exact-rows filter gradient on 41-pixel rows.
Usage: fuzz_round4.py [cases] [seed] [wgrad_strip|kwrows|subpixel|adam|pack3|strip2|espcn_one_launch|wgrad_rows41|conv1x1]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
from oracle import oracle as O

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(got, ref, tol=1e-3):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return False
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref)
    return bool(np.isfinite(got).all() and err.max() <= tol * scale and (err <= 1e-5 * scale + 1e-4 * np.abs(ref)).all())


def case_wgrad_strip(rng):
    pad = 'SAME' if rng.random() < 0.7 else 'VALID'
    n = int(rng.integers(1, 6))
    h = int(rng.integers(3, 40)) if rng.random() < 0.8 else int(rng.integers(40, 200))
    w = int(rng.integers(61, 300))
    if n * h * w > 60000:
        n = max(1, 60000 // (h * w))
    x = rng.uniform(-1, 1, (n, h, w, 64)).astype(np.float32)
    oh, ow = (h, w) if pad == 'SAME' else (h - 2, w - 2)
    dpre = rng.normal(size=(n, oh, ow, 64)).astype(np.float32)
    dw, db = ops.conv2d_bwd_filter(dev(x), dev(dpre), (3, 3, 64, 64), pad)
    dw_ref, db_ref = O.c_conv2d_bwd_filter(x, dpre, (3, 3), pad)
    dw2, db2 = ops.conv2d_bwd_filter(dev(x), dev(dpre), (3, 3, 64, 64), pad)
    return close(dw, dw_ref) and close(db, db_ref) and torch.equal(dw, dw2) and torch.equal(db, db2), 'wgrad strip %s %dx%dx%d' % (pad, n, h, w)


def case_kwrows(rng):
    pad = 'VALID' if rng.random() < 0.6 else 'SAME'
    act = [None, 'relu', 'tanh', 'sigmoid'][rng.integers(4)]
    n = int(rng.integers(1, 4))
    h, w = int(rng.integers(60, 400)), int(rng.integers(60, 400))
    if rng.random() < 0.5:                      # (the route starts at 4,096 output pixels since late round 4: small shapes too)
        h, w = int(rng.integers(9, 120)), int(rng.integers(9, 120))
    while n * (h - 4) * (w - 4) < 4096:
        h += 17; w += 11
    x = rng.uniform(-1, 1, (n, h, w, 32)).astype(np.float32)
    wt = (rng.normal(size=(5, 5, 32, 3)) / np.sqrt(800)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, 3).astype(np.float32)
    y = ops.conv2d_fwd(dev(x), dev(wt), dev(b), pad, act)
    return close(y, O.c_conv2d_fwd(x, wt, b, pad, act)), 'kwrows %s %s %dx%dx%d' % (pad, act, n, h, w)


def case_subpixel(rng):
    # chunks of one block of B = W r rC floats with B / 4 % 256 in 250..255 or 0, more chunks than workgroups
    r, c = [(2, 1), (2, 3), (1, 4), (3, 1), (2, 2)][rng.integers(5)]
    rc = r * c
    target = int(rng.integers(2, 7)) * 256 - int(rng.integers(0, 7))          # float4s per chunk
    w = max(1, (4 * target) // (r * rc))
    nh = int(rng.integers(1100, 3000))
    n = int(rng.integers(1, 4))
    h = max(1, nh // n)
    bits = rng.integers(0, 1 << 32, size=(n, h, w, c * r * r), dtype=np.uint64).astype(np.uint32)
    t = torch.from_numpy(bits.view(np.int32)).cuda().view(torch.float32)
    d = ops.depth_to_space(t, r)
    ok = torch.equal(ops.space_to_depth(d, r).view(torch.int32), t.view(torch.int32))
    step = 53
    ok = ok and np.array_equal(d.view(n, h, r, w * r, c)[:, ::step].contiguous().view(n, -1, w * r, c).view(torch.int32).cpu().numpy().view(np.uint32),
                               O.depth_to_space(np.ascontiguousarray(bits[:, ::step]), r))
    return ok, 'subpixel %dx%dx%dx%d r%d (float4s per block %d)' % (n, h, w, c, r, w * r * rc // 4)


def case_adam(rng):
    n = int(rng.integers(1, 200000))
    b1, b2 = float(rng.choice([0.9, 0.5])), float(rng.choice([0.999, 0.9]))
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    wa = torch.randn(n + 3, device='cuda', generator=g)[:n].clone() if n % 4 else torch.randn(n, device='cuda', generator=g)
    wb = wa.clone()
    ma, va, mb, vb = (torch.zeros(n, device='cuda') for _ in range(4))
    t0 = int(rng.integers(0, 100000))
    lr = float(10 ** rng.uniform(-5, -2))
    st = ops.adam_state('cuda', t=t0, lr=lr)
    ok = True
    for i in range(4):
        grad = torch.randn(n, device='cuda', generator=g)
        ops.adam_tf_step(wa, grad, ma, va, lr, t0 + i + 1, b1, b2, 1e-8)
        ops.adam_tf_step_dev(wb, grad, mb, vb, st, b1, b2, 1e-8)
    ok = torch.equal(ma, mb) and torch.equal(va, vb) and (wa - wb).abs().max().item() <= 1e-7 * max(1.0, wa.abs().max().item())
    ok = ok and ops.adam_state_get(st)[0] == t0 + 4
    return ok, 'adam n=%d t0=%d betas %g %g' % (n, t0, b1, b2)


def _path0(fn):
    from ml_super_resolution_amd import _lib
    old = _lib.lib().srx_set_conv_path(0)
    try:
        return fn()
    finally:
        _lib.lib().srx_set_conv_path(old)


def case_pack3(rng):
    # RGB-input 9x9 (from 4,096 output pixels) / 5x5 (from 60,000): oracle parity and BIT-identity with conv path 0
    k = 9 if rng.random() < 0.6 else 5
    pad = 'VALID' if rng.random() < 0.5 else 'SAME'
    act = [None, 'relu', 'tanh'][rng.integers(3)]
    n = int(rng.integers(1, 5))
    h, w = int(rng.integers(k + 1, 200)), int(rng.integers(k + 1, 200))
    need = 4096 if k == 9 else 60000
    while n * (h - k + 1) * (w - k + 1) < need:
        h += 23; w += 29
    x = rng.uniform(-1, 1, (n, h, w, 3)).astype(np.float32)
    wt = (rng.normal(size=(k, k, 3, 64)) / np.sqrt(k * k * 3)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, 64).astype(np.float32)
    xd, wd, bd = dev(x), dev(wt), dev(b)
    y = ops.conv2d_fwd(xd, wd, bd, pad, act)
    y0 = _path0(lambda: ops.conv2d_fwd(xd, wd, bd, pad, act))
    ok_ref, ok_eq = close(y, O.c_conv2d_fwd(x, wt, b, pad, act)), torch.equal(y, y0)
    return ok_ref and ok_eq, 'pack3 k%d %s %s %dx%dx%d (oracle %s, path 0 equal %s, max diff %g)' % (k, pad, act, n, h, w, ok_ref, ok_eq,
                                                                                                 (y - y0).abs().max().item())


def case_strip2(rng):
    # 3x3 64 -> 32 (none / relu / tanh) and 3x3 32 -> 27 through the sub-pixel map on images too wide for full-width tiles
    d2s = rng.random() < 0.5
    n = int(rng.integers(1, 4))
    h = int(rng.integers(1, 60)) if rng.random() < 0.7 else int(rng.integers(60, 300))
    w = int(rng.integers(59, 400))
    if n * h * w > 90000:
        n = 1; h = max(1, 90000 // w)
    if d2s:
        x = rng.uniform(-1, 1, (n, h, w, 32)).astype(np.float32)
        wt = (rng.normal(size=(3, 3, 32, 27)) / np.sqrt(288)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, 27).astype(np.float32)
        xd, wd, bd = dev(x), dev(wt), dev(b)
        y = ops.conv2d_fwd(xd, wd, bd, 'same', None, subpixel_r=3)
        y0 = _path0(lambda: ops.conv2d_fwd(xd, wd, bd, 'same', None, subpixel_r=3))
        ref = O.depth_to_space(O.c_conv2d_fwd(x, wt, b, 'SAME', None), 3)
        return close(y, ref) and torch.equal(y, y0), 'strip d2s %dx%dx%d' % (n, h, w)
    act = [None, 'relu', 'tanh'][rng.integers(3)]
    x = rng.uniform(-1, 1, (n, h, w, 64)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, 64, 32)) / np.sqrt(576)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, 32).astype(np.float32)
    xd, wd, bd = dev(x), dev(wt), dev(b)
    y = ops.conv2d_fwd(xd, wd, bd, 'same', act)
    y0 = _path0(lambda: ops.conv2d_fwd(xd, wd, bd, 'same', act))
    return close(y, O.c_conv2d_fwd(x, wt, b, 'SAME', act)) and torch.equal(y, y0), 'strip2 %s %dx%dx%d' % (act, n, h, w)


def case_espcn_one_launch(rng):
    # srx_espcn_forward on shapes that take tiles of every size up to 16 x 16, against the per-layer launches of conv path 0
    r = int(rng.integers(2, 5))
    n = int(rng.integers(1, 4)) if rng.random() < 0.7 else int(rng.integers(4, 300))
    h, w = int(rng.integers(1, 260)), int(rng.integers(1, 260))
    if n * h * w > 70000:
        n = 1
    if n * h * w > 70000:
        h = max(1, 70000 // w)
    x = rng.uniform(-1, 1, (n, h, w, 3)).astype(np.float32)
    ws = [(rng.normal(size=(5, 5, 3, 64)) * 0.1).astype(np.float32), (rng.normal(size=(3, 3, 64, 32)) * 0.05).astype(np.float32),
          (rng.normal(size=(3, 3, 32, 3 * r * r)) * 0.06).astype(np.float32)]
    bs = [rng.uniform(-0.1, 0.1, c).astype(np.float32) for c in (64, 32, 3 * r * r)]
    xd = dev(x)
    params = [(dev(a), dev(b)) for a, b in zip(ws, bs)]
    one = ops.espcn_forward(xd, params, r)

    def three():
        t = ops.conv2d_fwd(xd, params[0][0], params[0][1], 'same', 'tanh')
        t = ops.conv2d_fwd(t, params[1][0], params[1][1], 'same', 'tanh')
        return ops.conv2d_fwd(t, params[2][0], params[2][1], 'same', None, subpixel_r=r)
    y0 = _path0(three)
    ok = torch.equal(one, y0)
    if n * h * w <= 20000:
        ok = ok and close(one, O.depth_to_space(O.espcn_forward(x, list(zip(ws, bs))), r))
    return ok, 'espcn one launch r%d %dx%dx%d' % (r, n, h, w)


def case_conv1x1(rng):
    # streaming 1x1 forward / data gradient (from 100,000 pixels): oracle parity and equality with conv path 0
    cin, cout = [(64, 32), (32, 64), (64, 64), (32, 32)][rng.integers(4)]
    act = [None, 'relu'][rng.integers(2)]
    n = int(rng.integers(1, 5))
    h, w = int(rng.integers(40, 400)), int(rng.integers(40, 400))
    while n * h * w < 100000:
        h += 31; w += 17
    if n * h * w > 400000:
        n = 1
    x = np.abs(rng.uniform(-1, 1, (n, h, w, cin))).astype(np.float32) * (rng.uniform(size=(n, h, w, cin)) > 0.3)
    x = x.astype(np.float32)
    wt = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
    dpre = rng.normal(size=(n, h, w, cout)).astype(np.float32)
    xd, wd, bd, dd = dev(x), dev(wt), dev(b), dev(dpre)
    run = lambda: (ops.conv2d_fwd(xd, wd, bd, 'same', act), ops.conv2d_bwd_data(dd, wd, xd.shape, 'same', x_in=xd, in_act='relu'))
    y, dx = run()
    y0, dx0 = _path0(run)
    ok = close(y, O.c_conv2d_fwd(x, wt, b, 'SAME', act)) and close(dx, O.c_conv2d_bwd_data(dpre, wt, (h, w), 'SAME') * (x > 0))
    return ok and torch.equal(y, y0) and torch.equal(dx, dx0), 'conv1x1 %d->%d %s %dx%dx%d' % (cin, cout, act, n, h, w)


def case_wgrad_rows41(rng):
    n = int(rng.integers(1, 40))
    h = 41 if rng.random() < 0.5 else int(rng.integers(1, 90))
    x = rng.uniform(-1, 1, (n, h, 41, 64)).astype(np.float32)
    dpre = rng.normal(size=(n, h, 41, 64)).astype(np.float32)
    dw, db = ops.conv2d_bwd_filter(dev(x), dev(dpre), (3, 3, 64, 64), 'SAME')
    dw_ref, db_ref = O.c_conv2d_bwd_filter(x, dpre, (3, 3), 'SAME')
    dw2, db2 = ops.conv2d_bwd_filter(dev(x), dev(dpre), (3, 3, 64, 64), 'SAME')
    return close(dw, dw_ref) and close(db, db_ref) and torch.equal(dw, dw2) and torch.equal(db, db2), 'wgrad rows41 %dx%d' % (n, h)


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    kinds = [case_wgrad_strip, case_kwrows, case_subpixel, case_adam, case_pack3, case_strip2, case_espcn_one_launch, case_wgrad_rows41, case_conv1x1]
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
    print('fuzz_round4: %d cases, %d bad' % (cases, nbad))
    sys.exit(1 if nbad else 0)
