"""CPU tests of the oracle itself: float64 NumPy restatement vs the C restatement vs
an independent implementation (torch CPU conv2d), and vs the committed golden vectors.
None of these need a GPU."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.golden.make_golden import OP_CASES, vdsr_params, espcn_params, srcnn_params


def _torch_conv(x, w, b, pad, k):
    p = 0 if pad == 'VALID' else (k - 1) // 2
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    wt = torch.from_numpy(w).permute(3, 2, 0, 1).double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xt, wt, torch.from_numpy(b).double(), padding=p)
    return xt, wt, y


@pytest.mark.parametrize('case', OP_CASES, ids=[c[0] for c in OP_CASES])
def test_ops_numpy_c_torch_golden(case, golden_ops):
    name, k, cin, cout, pad, act, H, W = case
    g = {key.split('.', 1)[1]: golden_ops[key] for key in golden_ops.files if key.startswith(name + '.')}
    x, w, b, dy = g['x'], g['w'], g['b'], g['dy']
    # float64 numpy oracle reproduces the committed vectors
    y = O.conv2d_fwd(x, w, b, pad, act)
    np.testing.assert_allclose(y, g['y'], rtol=1e-6, atol=1e-6)
    # independent implementation (torch CPU, fp64)
    xt, wt, yt = _torch_conv(x, w, b, pad, k)
    np.testing.assert_allclose(O.act_apply(yt.detach().permute(0, 2, 3, 1).numpy(), act), y, rtol=1e-10, atol=1e-10)
    dpre = dy * O.act_grad_from_y(y, act)
    yt.backward(torch.from_numpy(dpre).permute(0, 3, 1, 2))
    dx = O.conv2d_bwd_data(dpre, w, (H, W), pad)
    dw, db = O.conv2d_bwd_filter(x, dpre, (k, k), pad)
    np.testing.assert_allclose(dx, xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dw, wt.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dx, g['dx'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dw, g['dw'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db, g['db'], rtol=1e-5, atol=1e-5)
    # C restatement (fp32) agrees with the fp64 one
    np.testing.assert_allclose(O.c_conv2d_fwd(x, w, b, pad, act), y, rtol=2e-5, atol=2e-5)
    dpre32 = O.c_act_bwd(dy, y.astype(np.float32), act)
    np.testing.assert_allclose(dpre32, dpre, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.c_conv2d_bwd_data(dpre, w, (H, W), pad), dx, rtol=1e-4, atol=1e-4)
    dwc, dbc = O.c_conv2d_bwd_filter(x, dpre, (k, k), pad)
    np.testing.assert_allclose(dwc, dw, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dbc, db, rtol=1e-4, atol=1e-4)


def test_skip_and_post_relu():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 6, 6, 8)).astype(np.float32)
    w = rng.normal(size=(3, 3, 8, 8)).astype(np.float32) * 0.2
    b = rng.normal(size=8).astype(np.float32)
    y = O.conv2d_fwd(x, w, b, 'SAME', None, skip=x, post_relu=True)
    ref = np.maximum(O.conv2d_fwd(x, w, b, 'SAME', None) + x, 0)
    np.testing.assert_allclose(y, ref)
    np.testing.assert_allclose(O.c_conv2d_fwd(x, w, b, 'SAME', None, skip=x, post_relu=True), ref, rtol=1e-5, atol=1e-5)


def test_same_valid_geometry():
    assert O.conv_geometry(41, 41, 3, 3, 'SAME') == (1, 1, 41, 41)
    assert O.conv_geometry(17, 17, 5, 5, 'SAME') == (2, 2, 17, 17)
    assert O.conv_geometry(243, 243, 9, 9, 'VALID') == (0, 0, 235, 235)
    # P4: SRCNN geometry (srcnn/srcnn.py:28-40): 256 -> crop 243, 9-1-5 VALID -> 231
    side, size = O.srcnn_sanity_check(256)
    assert (side, size) == (6, 243)
    assert size - 8 - 0 - 4 == 231


def test_adam_tf_epsilon_hat_differs_from_torch_adam():
    rng = np.random.default_rng(5)
    w = rng.normal(size=100); g = rng.normal(size=100) * 1e-6   # tiny grads expose the eps placement
    m = np.zeros(100); v = np.zeros(100)
    w1, m1, v1 = O.adam_tf(w, g, m, v, 1e-3, 1)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w1, w - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8))
    p = torch.tensor(w, requires_grad=True); p.grad = torch.tensor(g)
    torch.optim.Adam([p], lr=1e-3).step()
    assert np.abs(p.detach().numpy() - w1).max() > 1e-5      # epsilon-hat != torch.optim.Adam
    wc, mc, vc = O.c_adam_tf(w, g, m, v, 1e-3, 1)
    np.testing.assert_allclose(wc, w1, rtol=1e-5, atol=1e-7)
    # two steps, t=2 bias correction
    w2, m2, v2 = O.adam_tf(w1, g, m1, v1, 1e-3, 2)
    wc2, _, _ = O.c_adam_tf(wc, g, mc, vc, 1e-3, 2)
    np.testing.assert_allclose(wc2, w2, rtol=1e-5, atol=1e-7)


def test_momentum_clip():
    w = np.array([1.0, 1.0, 1.0]); g = np.array([10.0, -10.0, 0.01]); acc = np.zeros(3)
    w1, acc1 = O.momentum_clip(w, g, acc, lr=0.1)         # cap = 0.01/0.1 = 0.1
    np.testing.assert_allclose(acc1, [0.1, -0.1, 0.01])
    np.testing.assert_allclose(w1, [0.99, 1.01, 0.999])


def test_mse_l2_psnr_saturate():
    a = np.array([[0.5, -0.5], [1.0, 0.0]], np.float32).reshape(1, 2, 2, 1)
    b = np.zeros_like(a)
    loss, d = O.mse_fwd_bwd(a, b)
    assert abs(loss - 1.5 / 4) < 1e-12
    np.testing.assert_allclose(d, 2 * a / 4)
    assert abs(O.l2_loss(a) - 0.75) < 1e-12
    np.testing.assert_allclose(O.psnr(a, b, 2.0), 20 * np.log10(2.0) - 10 * np.log10(0.375))
    np.testing.assert_array_equal(O.saturate_u8(np.array([-2.0, -1.0, 0.0, 0.999, 1.0, 3.0])), [0, 0, 127, 254, 255, 255])
    assert O.lr_schedule(0.1, 0.1, 2559, 2560) == 0.1 and abs(O.lr_schedule(0.1, 0.1, 2560, 2560) - 0.01) < 1e-15


def test_vdsr_net_golden(golden_nets):
    g = golden_nets
    params = vdsr_params(106)
    loss, grads, fwd = O.vdsr_loss_and_grads(g['vdsr.sd'], g['vdsr.hd'], params)
    np.testing.assert_allclose(fwd['sr_images'], g['vdsr.sr'], rtol=1e-6, atol=1e-6)
    assert abs(loss - float(g['vdsr.loss'])) < 1e-9
    for i in (0, 1, 9, 18, 19):
        np.testing.assert_allclose(grads[i][0], g['vdsr.dk_%d' % i], rtol=1e-5, atol=1e-8)
    # conv.N taps are post-ReLU (pin P2 semantics): identical objects, all >= 0
    assert fwd['conv.5'] is fwd['relu.5'] and fwd['conv.5'].min() >= 0.0
    # C restatement of the whole step agrees at fp32 accuracy
    closs, cgrads = O.c_vdsr_train_step_grads(g['vdsr.sd'], g['vdsr.hd'], params)
    assert abs(closs - loss) < 1e-5 * max(1.0, abs(loss))
    for i in (0, 9, 19):
        scale = np.abs(grads[i][0]).max()
        assert np.abs(cgrads[i][0] - grads[i][0]).max() < 1e-3 * scale


def test_vdsr_gradient_is_the_gradient():
    """Finite-difference check of the restated backward (loss incl. L2 term)."""
    rng = np.random.default_rng(9)
    params = vdsr_params(11, num_layers=4)
    sd = rng.uniform(-1, 1, (1, 6, 5, 3)); hd = rng.uniform(-1, 1, (1, 6, 5, 3))
    loss, grads, _ = O.vdsr_loss_and_grads(sd, hd, params)
    for li, idx in ((0, (1, 2, 1, 7)), (2, (0, 1, 33, 12)), (3, (2, 2, 5, 1))):
        k = params[li][0].astype(np.float64).copy()
        eps = 1e-6
        k[idx] += eps
        p2 = list(params); p2[li] = (k, params[li][1])
        l2, _, _ = O.vdsr_loss_and_grads(sd, hd, p2)
        assert abs((l2 - loss) / eps - grads[li][0][idx]) < 1e-5


def test_espcn_srcnn_golden(golden_nets):
    g = golden_nets
    for r in (3, 4):
        y = O.espcn_forward(g['espcn%d.lr' % r], espcn_params(103 + r, r))
        np.testing.assert_allclose(y, g['espcn%d.y' % r], rtol=1e-5, atol=1e-6)
        assert O.espcn_scaling_factor(np.zeros(3 * r * r)) == r
        np.testing.assert_array_equal(O.depth_to_space(g['espcn%d.y' % r], r), g['espcn%d.d2s' % r])
    y = O.srcnn_forward(g['srcnn.lo'], srcnn_params(107))
    assert y.shape == (1, 21, 21, 3)
    np.testing.assert_allclose(y, g['srcnn.y'], rtol=1e-5, atol=1e-6)


def test_ssim_restatement_properties_and_cross_check():
    """tf.image.ssim restatement: identity -> 1, symmetry, decreases with noise, and agreement with an
    independent depthwise-conv formulation in torch (fp64)."""
    rng = np.random.default_rng(13)
    a = rng.uniform(-1, 1, (2, 23, 31, 3))
    b = np.clip(a + 0.2 * rng.normal(size=a.shape), -1, 1)
    c = np.clip(a + 0.6 * rng.normal(size=a.shape), -1, 1)
    np.testing.assert_allclose(O.ssim(a, a, 2.0), 1.0, atol=1e-12)
    np.testing.assert_allclose(O.ssim(a, b, 2.0), O.ssim(b, a, 2.0), rtol=1e-12)
    assert (O.ssim(a, b, 2.0) > O.ssim(a, c, 2.0)).all()
    # independent formulation: depthwise conv2d with the normalised outer-product gaussian
    x = torch.arange(11, dtype=torch.float64) - 5
    g = torch.exp(-x ** 2 / (2 * 1.5 ** 2)); g = g / g.sum()
    w = torch.outer(g, g)[None, None].repeat(3, 1, 1, 1)
    ta, tb = (torch.from_numpy(v).permute(0, 3, 1, 2) for v in (a, b))
    f = lambda z: torch.nn.functional.conv2d(z, w, groups=3)
    mu_a, mu_b = f(ta), f(tb)
    c1, c2 = (0.01 * 2.0) ** 2, (0.03 * 2.0) ** 2
    lum = (2 * mu_a * mu_b + c1) / (mu_a ** 2 + mu_b ** 2 + c1)
    cs = (2 * (f(ta * tb) - mu_a * mu_b) + c2) / (f(ta * ta) - mu_a ** 2 + f(tb * tb) - mu_b ** 2 + c2)
    ref = (lum * cs).mean(dim=(2, 3)).mean(dim=1).numpy()
    np.testing.assert_allclose(O.ssim(a, b, 2.0), ref, rtol=1e-10)
