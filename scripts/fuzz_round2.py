#!/usr/bin/env python3
"""Random-shape parity sweep of the round-2 entry points against NumPy float64 / the oracle (a development tool; the
committed cases live in tests/): srx_conv3x3_blocked (forward, data gradient, fused mask), srx_espcn_forward,
srx_gemm, max-pooling, the stride-2 sample map, channel normalisation, patch extraction, srx_texture_gram(_bwd),
srx_conv3x3_blocked_bwd_filter, srx_resample_u8.
Usage: fuzz_round2.py [cases] [seed] [wide|espcn|gemm|small_ops|texture|blocked_wgrad|resample]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import blocked, ops
from oracle import oracle as O
from oracle import oracle_enet as E

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
npy = lambda t: t.detach().cpu().numpy().astype(np.float64)


def bad(got, ref, tol=1e-3):
    ref = np.asarray(ref, np.float64)
    return (not np.isfinite(got).all()) or got.shape != ref.shape or \
        np.abs(got - ref).max() > tol * max(np.abs(ref).max(), 1e-30)


def case_wide(rng):
    cin, cout = 64 * int(rng.integers(1, 6)), 64 * int(rng.integers(1, 6))
    if cin == 64 and cout == 64:
        cout = 128
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 40)), int(rng.integers(1, 150 if rng.random() < 0.3 else 70))
    if rng.random() < 0.25:        # many units per workgroup: the step pipeline crosses tiles, images and produced blocks
        n, h, w = int(rng.integers(20, 90)), int(rng.integers(1, 25)), int(rng.integers(1, 25))
        cin, cout = min(cin, 192), min(cout, 192)
        if cin == 64 and cout == 64:
            cin = 128
    act = [None, 'relu', 'lrelu'][rng.integers(3)]
    tag = 'wide N%d %dx%d %d->%d %s' % (n, h, w, cin, cout, act)
    k = rng.normal(0, 1 / np.sqrt(9 * cin), (3, 3, cin, cout)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    x = rng.normal(size=(n, h, w, cin)).astype(np.float32)
    layer = blocked.BlockedConv(cin, cout, 1, act, torch.empty(blocked.BlockedConv.kernel_shape(cin, cout), device='cuda'),
                                torch.empty(cout, device='cuda'))
    layer.set_kernel_hwio(k, b)
    y = layer.forward(blocked.to_blocks(dev(x)))
    if bad(npy(blocked.to_nhwc(y)), O.act_apply(E.conv2d_same_fwd(x, k, b, 1), act)):
        return tag + ': forward'
    dp = rng.normal(size=(n, h, w, cout)).astype(np.float32)
    dref, _, _ = E.conv2d_same_bwd(x, k, dp, 1)
    if bad(npy(blocked.to_nhwc(layer.dgrad(blocked.to_blocks(dev(dp))))), dref):
        return tag + ': dgrad'
    mact = ['relu', 'lrelu'][rng.integers(2)]
    xm = np.where(rng.uniform(size=x.shape) < 0.5, x, 0).astype(np.float32)
    got = layer.dgrad(blocked.to_blocks(dev(dp)), mask=blocked.to_blocks(dev(xm)), mask_act=mact)
    if bad(npy(blocked.to_nhwc(got)), dref * O.act_grad_from_y(xm.astype(np.float64), mact)):
        return tag + ': dgrad with fused %s mask' % mact
    return None


def case_espcn(rng):
    from ml_super_resolution_amd.espcn import model_espcn
    r = int(rng.integers(2, 5))
    n, h, w = int(rng.integers(1, 12)), int(rng.integers(1, 45)), int(rng.integers(1, 45))
    m = model_espcn.EspcnModel(r, device='cuda', seed=int(rng.integers(1 << 30)))
    for i in range(3):
        m.stack.bias(i).uniform_(-0.1, 0.1)
    x = torch.rand((n, h, w, 3), device='cuda') * 2 - 1
    one = m.super_resolve(x, single_launch=True).clone()
    three = m.super_resolve(x, use_graph=False, single_launch=False).clone()
    tag = 'espcn N%d %dx%d r%d' % (n, h, w, r)
    if not torch.equal(one, three):
        return tag + ': single launch differs from the per-layer launches (max %g)' % float((one - three).abs().max())
    params = [(m.stack.kernel(i).cpu().numpy(), m.stack.bias(i).cpu().numpy()) for i in range(3)]
    if bad(npy(one), O.depth_to_space(O.espcn_forward(x.cpu().numpy(), params), r)):
        return tag + ': against the oracle'
    return None


def case_gemm(rng):
    batch = int(rng.integers(1, 5)) if rng.random() < 0.4 else 0
    M, N, K = (int(rng.integers(1, 200)) for _ in range(3))
    if rng.random() < 0.2:
        M, N, K = int(rng.integers(1, 80)), int(rng.integers(1, 300)), int(rng.integers(512, 3000))      # split-K territory
    ta, tb = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    shp = lambda r, c: ((batch,) if batch else ()) + (r, c)
    A = rng.normal(size=shp(K, M) if ta else shp(M, K)).astype(np.float32)
    B = rng.normal(size=shp(N, K) if tb else shp(K, N)).astype(np.float32)
    opA = np.swapaxes(A, -1, -2) if ta else A
    opB = np.swapaxes(B, -1, -2) if tb else B
    ref = opA.astype(np.float64) @ opB.astype(np.float64)
    got = ops.gemm(dev(A), dev(B), trans_a=ta, trans_b=tb, alpha=0.5)
    if bad(npy(got), 0.5 * ref):
        return 'gemm batch %d M%d N%d K%d ta%d tb%d' % (batch, M, N, K, ta, tb)
    return None


def case_small_ops(rng):
    n, h, w, c = int(rng.integers(1, 4)), int(rng.integers(1, 30)), int(rng.integers(1, 30)), 4 * int(rng.integers(1, 40))
    x = rng.normal(size=(n, h, w, c)).astype(np.float32)
    if not np.array_equal(npy(ops.maxpool2x2(dev(x))), E.maxpool2x2_fwd(x).astype(np.float32).astype(np.float64)):
        return 'maxpool %s' % (x.shape,)
    dy = rng.normal(size=(n, (h + 1) // 2, (w + 1) // 2, c)).astype(np.float32)
    if not np.array_equal(npy(ops.maxpool2x2_bwd(dev(x), dev(dy))), E.maxpool2x2_bwd(x, dy).astype(np.float32).astype(np.float64)):
        return 'maxpool_bwd %s' % (x.shape,)
    xa = np.abs(x) + 0.1
    if bad(npy(ops.channel_normalize(dev(xa))), E.normalize(xa)) or \
            bad(npy(ops.channel_normalize_bwd(dev(xa), dev(x))), E.normalize_bwd(xa, x)):
        return 'normalize %s' % (x.shape,)
    he, we = 2 * ((h + 1) // 2), 2 * ((w + 1) // 2)
    xe = rng.normal(size=(n, he, we, c)).astype(np.float32)
    if not np.array_equal(npy(ops.subsample2(dev(xe), 1, 1)), xe[:, 1::2, 1::2].astype(np.float64)):
        return 'subsample %s' % (xe.shape,)
    hp, wp = 16 * int(rng.integers(1, 4)), 16 * int(rng.integers(1, 4))
    xp = rng.normal(size=(n, hp, wp, c)).astype(np.float32)
    pt = ops.extract_patches16(dev(xp))
    if not np.array_equal(npy(pt), E.patches16(xp).astype(np.float64)) or \
            not np.array_equal(npy(ops.extract_patches16_bwd(pt, xp.shape)), xp.astype(np.float64)):
        return 'patches16 %s' % (xp.shape,)
    return None


def case_texture(rng):
    """srx_texture_gram / _bwd against float64 (enet/enet/model_enet.py:34-41, 225-259)."""
    c = [64, 128, 256][rng.integers(3)]
    n, ph, pw = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    h, w = 16 * ph, 16 * pw
    tag = 'texture N%d %dx%d C%d' % (n, h, w, c)
    x = (np.abs(rng.normal(size=(n, h, w, c))) * (rng.random((n, h, w, c)) < 0.7) + 0.01).astype(np.float32)
    g = ops.texture_gram(dev(x))
    x64 = x.astype(np.float64)
    m = x64.mean(axis=-1, keepdims=True) + 1e-6
    pat = (x64 / m).reshape(n, ph, 16, pw, 16, c).transpose(0, 1, 3, 2, 4, 5).reshape(-1, 256, c)
    if bad(npy(g), np.einsum('pki,pkj->pij', pat, pat)):
        return tag + ': gram'
    dg = rng.normal(size=(n * ph * pw, c, c)).astype(np.float32)
    dg = (dg + dg.transpose(0, 2, 1)) * 0.5
    dx = ops.texture_gram_bwd(dev(x), dev(dg))
    dn = 2.0 * np.einsum('pkj,pjc->pkc', pat, dg.astype(np.float64))
    dn = dn.reshape(n, ph, pw, 16, 16, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h, w, c)
    if bad(npy(dx), dn / m - (dn * x64).sum(axis=-1, keepdims=True) / (c * m * m)):
        return tag + ': gradient'
    return None


def case_blocked_wgrad(rng):
    """srx_conv3x3_blocked_bwd_filter (all block pairs in one launch, or the per-pair fallback) against float64 sums."""
    cib, cob = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    n, h = int(rng.integers(1, 6)), int(rng.integers(1, 20))
    w = int(rng.integers(1, 140 if rng.random() < 0.2 else 40))
    tag = 'blocked wgrad %dx%d blocks N%d %dx%d' % (cib, cob, n, h, w)
    x = rng.normal(size=(cib, n, h, w, 64)).astype(np.float32)
    dp = rng.normal(size=(cob, n, h, w, 64)).astype(np.float32)
    dw = torch.full((cib, cob, 3, 3, 64, 64), float('nan'), device='cuda')
    db = torch.full((cob * 64,), float('nan'), device='cuda')
    ops.conv3x3_blocked_bwd_filter(dev(x), dev(dp), dw, db)
    if bad(npy(db), dp.astype(np.float64).sum(axis=(1, 2, 3)).reshape(-1)):
        return tag + ': bias gradient'
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1), (0, 0)))
    ib, ob = int(rng.integers(cib)), int(rng.integers(cob))
    ref = np.stack([np.stack([np.einsum('nhwi,nhwo->io', xp[ib][:, kh:kh + h, kw:kw + w, :], dp[ob].astype(np.float64))
                              for kw in range(3)]) for kh in range(3)])
    if bad(npy(dw[ib, ob]), ref):
        return tag + ': pair (%d, %d)' % (ib, ob)
    return None


def case_resample(rng):
    """ops.resize_pil_u8 (Pillow's integer resample on the GPU) against the oracle's restatement, byte for byte."""
    filt = ['bilinear', 'bicubic'][rng.integers(2)]
    n, c = int(rng.integers(1, 4)), [1, 3, 4][rng.integers(3)]
    h, w, oh, ow = (int(rng.integers(1, 90)) for _ in range(4))
    img = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
    if rng.random() < 0.2:
        img[:] = [0, 255][rng.integers(2)]            # saturated images: the clipping paths
    got = ops.resize_pil_u8(torch.from_numpy(img).cuda(), oh, ow, filt).cpu().numpy()
    if not np.array_equal(got, O.pil_resize_u8(img, oh, ow, filt)):
        return 'resample %s N%d %dx%dx%d -> %dx%d' % (filt, n, h, w, c, oh, ow)
    return None


if __name__ == '__main__':
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    kinds = [case_wide, case_espcn, case_gemm, case_small_ops, case_texture, case_blocked_wgrad, case_resample]
    if len(sys.argv) > 3:          # fuzz_round2.py cases seed wide|espcn|gemm|small_ops: one kind only
        kinds = [k for k in kinds if k.__name__ == 'case_' + sys.argv[3]]
    nbad = 0
    for it in range(cases):
        fn = kinds[it % len(kinds)]
        try:
            msg = fn(rng)
        except Exception as exc:
            msg = '%s: %r' % (fn.__name__, exc)
        if msg:
            nbad += 1
            print('FAIL', msg, flush=True)
    print('fuzz_round2: %d cases, %d bad' % (cases, nbad))
    sys.exit(1 if nbad else 0)
