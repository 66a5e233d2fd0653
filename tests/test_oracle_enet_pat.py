"""oracle/oracle_enet.py (EnhanceNet-PAT's loss side) against an independent implementation:
torch CPU float64 with autograd.  TensorFlow is not available, so this is a cross-check, not a pin (parity unpinned).
Same structure as the reference (VGG-19's 16 convolutions + 5 pools, the 10-conv + 2-dense discriminator), narrower
channels so that the whole check runs in seconds."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_enet as E

F64 = torch.float64       # (explicit everywhere: a process-wide default dtype would leak into the other test modules)


def _t(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    return t.requires_grad_(grad)


def _conv_same(x, w, b, stride=1):
    """NHWC x, HWIO w -> TF SAME with the given stride (asymmetric padding: extra pixel after)."""
    n, h, wd, _ = x.shape
    kh, kw = w.shape[:2]
    oh, ow = -(-h // stride), -(-wd // stride)
    ph, pw = max((oh - 1) * stride + kh - h, 0), max((ow - 1) * stride + kw - wd, 0)
    xp = F.pad(x.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return F.conv2d(xp, w.permute(3, 2, 0, 1), b, stride=stride).permute(0, 2, 3, 1)


def _vgg(x, weights):
    v = x * 127.5 + 127.5
    t = torch.flip(v, dims=[-1]) - torch.tensor(E.VGG_MEAN_BGR, dtype=F64)
    feats = {}
    for name in E.VGG_LAYERS:
        if name.endswith('pool'):
            t = F.max_pool2d(t.permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
        else:
            t = F.relu(_conv_same(t, *weights[name]))
        feats[name] = t
    return feats


def _disc(x, convs, dense):
    t = x
    for i, (k, b) in enumerate(convs):
        t = F.leaky_relu(_conv_same(t, k, b, 1 + (i & 1)), 0.2)
    h = F.leaky_relu(t.reshape(t.shape[0], -1) @ dense[0][0] + dense[0][1], 0.2)
    return torch.sigmoid(h @ dense[1][0] + dense[1][1])


def _log_loss(label, p):
    return torch.mean(-label * torch.log(p + 1e-7) - (1 - label) * torch.log(1 - p + 1e-7))


def _normalize(t):
    return t / (t.mean(dim=-1, keepdim=True) + 1e-6)


def _g_losses(sr, hd, vw, dc, dd, pat):
    sf, hf = _vgg(sr, vw), _vgg(hd, vw)
    p = 0.2 * F.mse_loss(_normalize(sf['block2_pool']), _normalize(hf['block2_pool'])) + \
        0.02 * F.mse_loss(_normalize(sf['block5_pool']), _normalize(hf['block5_pool']))
    out = {'p_loss': p}
    total = p
    if 'a' in pat:
        fake, real = _disc(sr, dc, dd), _disc(hd, dc, dd)
        out['a_loss'] = _log_loss(0.0, fake) + _log_loss(1.0, real)
        out['g_loss'] = _log_loss(1.0, fake)
        total = total + out['g_loss'] * (2.0 if 't' in pat else 1.0)
    if 't' in pat:
        t = 0.0
        for name, wgt in E.TEXTURE_LAYERS:
            grams = []
            for f in (sf[name], hf[name]):
                f = _normalize(f)
                n, h, w, c = f.shape
                pt = f.reshape(n, h // 16, 16, w // 16, 16, c).permute(0, 1, 3, 2, 4, 5).reshape(n, -1, 256, c)
                grams.append(pt.transpose(-1, -2) @ pt)
            t = t + wgt * F.mse_loss(grams[0], grams[1])
        out['t_loss'] = t
        total = total + t
    out['g_loss_all'] = total
    return out


def _make(rng, vgg_width=8, d_width=4, size=64):
    vw = {}
    for name, (cin, cout) in E.vgg19_channels(vgg_width).items():
        vw[name] = (rng.normal(0, np.sqrt(2.0 / (9 * cin)), (3, 3, cin, cout)), rng.normal(0, 0.05, cout))
    # scale the first layer for inputs of magnitude ~100
    vw['block1_conv1'] = (vw['block1_conv1'][0] / 60.0, vw['block1_conv1'][1])
    dc, cin = [], 3
    for i in range(5):
        f = d_width * 2 ** i
        for _ in range(2):
            dc.append((rng.normal(0, np.sqrt(1.5 / (9 * cin)), (3, 3, cin, f)), rng.normal(0, 0.05, f)))
            cin = f
    feat = (size // 32) ** 2 * cin
    dd = [(rng.normal(0, np.sqrt(1.0 / feat), (feat, 32)), rng.normal(0, 0.05, 32)),
          (rng.normal(0, np.sqrt(1.0 / 32), (32, 1)), rng.normal(0, 0.05, 1))]
    hd = rng.uniform(-1, 1, (2, size, size, 3))
    sr = np.clip(hd + rng.normal(0, 0.2, hd.shape), -1, 1)
    return vw, dc, dd, sr, hd


def test_strided_same_conv_and_pool_against_torch():
    rng = np.random.default_rng(0)
    for (h, w, s) in ((8, 8, 2), (7, 9, 2), (6, 5, 1), (10, 12, 2)):
        x, k, b = rng.normal(size=(2, h, w, 3)), rng.normal(size=(3, 3, 3, 5)), rng.normal(size=5)
        xt, kt, bt = _t(x, True), _t(k, True), _t(b, True)
        y = _conv_same(xt, kt, bt, s)
        np.testing.assert_allclose(E.conv2d_same_fwd(x, k, b, s), y.detach().numpy(), atol=1e-12)
        dy = rng.normal(size=y.shape)
        y.backward(_t(dy))
        dx, dk, db = E.conv2d_same_bwd(x, k, dy, s)
        np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-12)
        np.testing.assert_allclose(dk, kt.grad.numpy(), atol=1e-11)
        np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-11)
        # 2x2 / 2 SAME max-pooling, odd sizes included
        pt = _t(x, True)
        p = F.max_pool2d(pt.permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
        np.testing.assert_array_equal(E.maxpool2x2_fwd(x), p.detach().numpy())
        dp = rng.normal(size=p.shape)
        p.backward(_t(dp))
        np.testing.assert_allclose(E.maxpool2x2_bwd(x, dp), pt.grad.numpy(), atol=1e-14)   # (no ties in random data)
    # stride-2 SAME on an even image == the stride-1 SAME output at the odd positions (the engine's route)
    x, k, b = rng.normal(size=(1, 8, 6, 4)), rng.normal(size=(3, 3, 4, 2)), rng.normal(size=2)
    np.testing.assert_allclose(E.conv2d_same_fwd(x, k, b, 2), E.conv2d_same_fwd(x, k, b, 1)[:, 1::2, 1::2], atol=1e-13)


@pytest.mark.parametrize('pat', ['p', 'pa', 'pat'])
def test_generator_objective_and_sr_gradient_against_torch(pat):
    rng = np.random.default_rng(1)
    vw, dc, dd, sr, hd = _make(rng)
    losses, d_sr = E.enet_losses_and_sr_gradient(sr, hd, vw, dc, dd, pat)
    srt = _t(sr, True)
    ref = _g_losses(srt, _t(hd), {k: (_t(a), _t(b)) for k, (a, b) in vw.items()},
                    [(_t(a), _t(b)) for a, b in dc], [(_t(a), _t(b)) for a, b in dd], pat)
    assert set(losses) == set(ref)
    for k in ref:
        np.testing.assert_allclose(losses[k], ref[k].item(), rtol=1e-10)
    ref['g_loss_all'].backward()
    np.testing.assert_allclose(d_sr, srt.grad.numpy(), rtol=1e-7, atol=1e-9 * np.abs(srt.grad.numpy()).max())


def test_discriminator_gradients_against_torch():
    rng = np.random.default_rng(2)
    vw, dc, dd, sr, hd = _make(rng)
    a_loss, cg, dg = E.discriminator_loss_and_grads(sr, hd, dc, dd)
    dct = [(_t(a, True), _t(b, True)) for a, b in dc]
    ddt = [(_t(a, True), _t(b, True)) for a, b in dd]
    loss = _log_loss(0.0, _disc(_t(sr), dct, ddt)) + _log_loss(1.0, _disc(_t(hd), dct, ddt))
    loss.backward()
    np.testing.assert_allclose(a_loss, loss.item(), rtol=1e-12)
    for (gk, gb), (kt, bt) in zip(cg + dg, dct + ddt):
        np.testing.assert_allclose(gk, kt.grad.numpy(), rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(gb, bt.grad.numpy(), rtol=1e-7, atol=1e-12)


def test_vgg19_structure():
    ch = E.vgg19_channels(64)
    assert len(ch) == 16 and ch['block1_conv1'] == (3, 64) and ch['block3_conv1'] == (128, 256)
    assert ch['block4_conv1'] == (256, 512) and ch['block5_conv4'] == (512, 512)


def test_committed_fixture_is_what_the_oracle_computes():
    """tests/golden/enet_pat.npz freezes oracle/oracle_enet.py for the seeded case of make_golden.enet_pat_case()."""
    import os
    from tests.golden.make_golden import enet_pat_case
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'enet_pat.npz'))
    vgg, convs, dense, sr, hd = enet_pat_case()
    feats = E.vgg19_forward(sr, vgg)
    for name in ('block1_conv1', 'block2_pool', 'block3_conv1', 'block5_pool'):
        np.testing.assert_allclose(feats[name], z['vgg.' + name], rtol=0, atol=1e-5 * np.abs(feats[name]).max())
    losses, _ = E.enet_losses_and_sr_gradient(sr, hd, vgg, convs, dense, 'pat')
    for k, v in losses.items():
        np.testing.assert_allclose(v, float(z['loss.' + k]), rtol=1e-12)
