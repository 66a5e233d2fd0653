#!/usr/bin/env python3
"""Random-shape parity sweep of round 4's kernels against the oracle (a development tool; the committed cases live in
tests/): the exact-rows strip filter gradient (wgrad_rows_strip_kernel: any width from 59 columns up, any last-strip
width, VALID and SAME, ranges that cut strips and images), the (kw, co)-rows 5x5 32 -> 3 kernel (conv_kwrows_kernel), the
sub-pixel map on shapes whose chunks end within 6 float4 of a multiple of 256 (the LDS slot guard of round 4), and the
device-state Adam against the host-argument one.
Usage: fuzz_round4.py [cases] [seed] [wgrad_strip|kwrows|subpixel|adam]"""
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
    while n * (h - 4) * (w - 4) < 60000:
        h += 37; w += 41
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


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    kinds = [case_wgrad_strip, case_kwrows, case_subpixel, case_adam]
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
