"""EnhanceNet-PAT's loss side on the GPU against oracle/oracle_enet.py: the new C-ABI
operators one by one, layers wider than 64 channels and stride-2 layers on the 64-channel kernels, VGG-19, the
discriminator, and whole generator / discriminator training steps.  VGG-19 weights are random VGG-shaped tensors (the
real ones are not available offline): parity unpinned, as for every float path of this repository."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import oracle_enet as E
from tests.test_gpu_ops import close, dev

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def to_nhwc_np(blocked):
    from ml_super_resolution_amd.blocked import to_nhwc
    return _np(to_nhwc(blocked))


# ---------------------------------------------------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [(2, 8, 8, 8), (1, 7, 9, 4), (3, 5, 4, 64), (1, 1, 1, 4)])
def test_maxpool_fwd_bwd(shape):
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(sum(shape))
    x = rng.normal(size=shape).astype(np.float32)
    y = ops.maxpool2x2(dev(x))
    np.testing.assert_array_equal(_np(y), E.maxpool2x2_fwd(x).astype(np.float32))
    dy = rng.normal(size=y.shape).astype(np.float32)
    np.testing.assert_array_equal(_np(ops.maxpool2x2_bwd(dev(x), dev(dy))), E.maxpool2x2_bwd(x, dy).astype(np.float32))
    # ties (post-ReLU zeros): the first maximum in scan order takes the gradient
    xz = np.maximum(x, 0) * (rng.uniform(size=shape) > 0.5)
    np.testing.assert_array_equal(_np(ops.maxpool2x2_bwd(dev(xz), dev(dy))), E.maxpool2x2_bwd(xz, dy).astype(np.float32))
    # the activation gradient of the layer that produced x fused in: the same bits as the two launches
    for act in ('relu', 'lrelu', 'tanh'):
        xa = np.tanh(x) if act == 'tanh' else xz if act == 'relu' else x
        two = ops.act_bwd(ops.maxpool2x2_bwd(dev(xa), dev(dy)), dev(xa), act)
        assert torch.equal(ops.maxpool2x2_bwd(dev(xa), dev(dy), mask_act=act), two)


def test_subsample_blocks_patches_are_exact_permutations():
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 8, 12, 8)).astype(np.float32)
    np.testing.assert_array_equal(_np(ops.subsample2(dev(x), 1, 1)), x[:, 1::2, 1::2])
    np.testing.assert_array_equal(_np(ops.subsample2(dev(x), 0, 1)), x[:, 0::2, 1::2])
    d = rng.normal(size=(2, 4, 6, 8)).astype(np.float32)
    z = np.zeros_like(x); z[:, 1::2, 1::2] = d
    np.testing.assert_array_equal(_np(ops.subsample2_bwd(dev(d), 1, 1)), z)
    # channel blocks
    p = rng.normal(size=(2, 3, 5, 192)).astype(np.float32)
    blk = ops.nhwc_to_blocks(dev(p))
    assert blk.shape == (3, 2, 3, 5, 64)
    np.testing.assert_array_equal(_np(blk), p.reshape(2, 3, 5, 3, 64).transpose(3, 0, 1, 2, 4))
    np.testing.assert_array_equal(_np(ops.blocks_to_nhwc(blk)), p)
    assert ops.nhwc_to_blocks(dev(p[..., :32])).shape == (1, 2, 3, 5, 32)
    # 16x16 patches
    f = rng.normal(size=(2, 32, 48, 8)).astype(np.float32)
    pt = ops.extract_patches16(dev(f))
    np.testing.assert_array_equal(_np(pt), E.patches16(f))
    np.testing.assert_array_equal(_np(ops.extract_patches16_bwd(pt, f.shape)), f)


def test_normalize_logloss_preprocess_addscaled_colsum():
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(4)
    for C in (8, 64, 128, 512):
        x = np.abs(rng.normal(size=(3, 4, 5, C))).astype(np.float32)
        dy = rng.normal(size=x.shape).astype(np.float32)
        close(ops.channel_normalize(dev(x)), E.normalize(x))
        close(ops.channel_normalize_bwd(dev(x), dev(dy)), E.normalize_bwd(x, dy))
    p = rng.uniform(0.01, 0.99, (7, 1)).astype(np.float32)
    loss = torch.zeros(1, device='cuda')
    for label in (0.0, 1.0):
        dp = ops.log_loss(dev(p), label, loss)
        ref_l, ref_d = E.log_loss(label, p)
        assert abs(loss.item() - ref_l) < 1e-6 * abs(ref_l)
        close(dp, ref_d)
    ops.log_loss(dev(p), 0.0, loss, loss_scale=2.0, grad_scale=3.0, accumulate=True)
    assert abs(loss.item() - (E.log_loss(1.0, p)[0] + 2 * E.log_loss(0.0, p)[0])) < 1e-5
    img = rng.uniform(-1, 1, (2, 5, 6, 3)).astype(np.float32)
    close(ops.vgg_preprocess(dev(img)), E.vgg_preprocess(img))
    g = rng.normal(size=img.shape).astype(np.float32)
    close(ops.vgg_preprocess(dev(g), backward=True), 127.5 * g[..., ::-1])
    a, b = rng.normal(size=(1000,)).astype(np.float32), rng.normal(size=(1000,)).astype(np.float32)
    close(ops.add_scaled(dev(a), dev(b), 0.5, -2.0), 0.5 * a - 2.0 * b)
    m = rng.normal(size=(37, 130)).astype(np.float32)
    close(ops.column_sums(dev(m)), m.astype(np.float64).sum(axis=0))


@pytest.mark.parametrize('M,N,K', [(64, 1024, 8192), (5, 7, 3), (64, 64, 256), (130, 70, 33), (1, 1, 1024), (64, 8192, 1024),
                                   # skinny routes (round 3; the discriminator's dense layer at 512 x 512 inputs): M <= 16 against a long K
                                   # (gemm_skinny_nn_kernel, split-K) and, transposed, against many rows of B (gemm_skinny_nt_kernel)
                                   (4, 1024, 8192), (8, 4096, 1024), (3, 2052, 512), (16, 1024, 4096), (13, 2500, 768), (2, 260, 4100),
                                   (1, 2048, 4096)])
def test_gemm_against_float64(M, N, K):
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(M + N + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    close(ops.gemm(dev(A), dev(B)), ref)
    close(ops.gemm(dev(A.T.copy()), dev(B), trans_a=True), ref)
    close(ops.gemm(dev(A), dev(B.T.copy()), trans_b=True), ref)
    close(ops.gemm(dev(A), dev(B), bias=dev(bias), act='lrelu', alpha=0.5), E.lrelu(0.5 * ref + bias))
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    out = dev(c0)
    ops.gemm(dev(A), dev(B), out=out, accumulate=True)
    close(out, ref + c0)
    out = dev(c0)
    ops.gemm(dev(A), dev(B.T.copy()), trans_b=True, alpha=0.5, out=out, accumulate=True)
    close(out, 0.5 * ref + c0)
    # deterministic (split-K partials are added in a fixed order)
    assert torch.equal(ops.gemm(dev(A), dev(B)), ops.gemm(dev(A), dev(B)))


def test_gemm_batched_gram():
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(9)
    x = rng.normal(size=(6, 256, 128)).astype(np.float32)
    g = ops.gemm(dev(x), dev(x), trans_a=True)
    ref = np.einsum('bki,bkj->bij', x.astype(np.float64), x.astype(np.float64))
    close(g, ref)
    assert torch.equal(g, g.transpose(1, 2))                      # bitwise symmetric (same products, same order)
    dg = rng.normal(size=(6, 128, 128)).astype(np.float32)
    close(ops.gemm(dev(x), dev(dg), alpha=2.0), 2 * np.einsum('bki,bij->bkj', x.astype(np.float64), dg.astype(np.float64)))


# ---------------------------------------------------------------------------------------------------------------------
# layers wider than 64 channels / stride 2 on the 64-channel kernels
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,stride,act,hw', [(3, 32, 1, 'lrelu', 16), (32, 32, 2, 'lrelu', 16), (64, 128, 1, 'relu', 12),
                                                    (128, 128, 2, 'lrelu', 8), (256, 128, 1, 'relu', 6), (128, 256, 1, 'lrelu', 4),
                                                    (192, 192, 2, 'relu', 64)])
def test_blocked_conv_fwd_dgrad_wgrad(cin, cout, stride, act, hw):
    from ml_super_resolution_amd import ops
    from ml_super_resolution_amd.blocked import BlockedConv, to_blocks, to_nhwc
    rng = np.random.default_rng(cin + cout)
    n = 2
    k = rng.normal(0, np.sqrt(1.0 / (9 * cin)), (3, 3, cin, cout)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    x = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    ks = BlockedConv.kernel_shape(cin, cout)
    layer = BlockedConv(cin, cout, stride, act, torch.empty(ks, device='cuda'), torch.empty(cout, device='cuda'),
                        torch.empty(ks, device='cuda'), torch.empty(cout, device='cuda'))
    layer.set_kernel_hwio(k, b)
    np.testing.assert_array_equal(_np(layer.kernel_hwio()), k)
    xb = to_blocks(dev(x))
    y = layer.forward(xb)
    pre = E.conv2d_same_fwd(x, k, b, stride)
    yref = O.act_apply(pre, act)
    close(to_nhwc(y), yref)
    dy = rng.normal(size=yref.shape).astype(np.float32)
    # (the activation mask from the DEVICE's y: an output within rounding of zero may fall on the other side in the
    # float64 oracle, which would compare two different masks, not two convolutions)
    dpre_ref = dy * O.act_grad_from_y(_np(to_nhwc(y)).astype(np.float64), act)
    dpre = to_blocks(ops.act_bwd(dev(dy), to_nhwc(y).contiguous(), act))
    dx_ref, dk_ref, db_ref = E.conv2d_same_bwd(x, k, dpre_ref, stride)
    close(to_nhwc(layer.dgrad(dpre)), dx_ref)
    layer.wgrad(xb, dpre)
    close(layer.kernel_hwio(layer.dw), dk_ref)
    close(layer.db, db_ref)


# ---------------------------------------------------------------------------------------------------------------------
# VGG-19 and the discriminator
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_vgg_weights(w):
    return {name: (w[name][name + '_W_1'], w[name][name + '_b_1']) for name in w}


@pytest.mark.parametrize('width,size,n', [(8, 64, 2), (64, 128, 1)], ids=['narrow', 'vgg19'])
def test_vgg19_features_and_input_gradient(width, size, n):
    from ml_super_resolution_amd.enet import model_vgg
    w = model_vgg.random_vgg_weights(5, width)
    net = model_vgg.Vgg19(w, device='cuda')
    rng = np.random.default_rng(6)
    img = rng.uniform(-1, 1, (n, size, size, 3)).astype(np.float32)
    feats = net.forward(dev(img), keep=True)
    ow = _oracle_vgg_weights(w)
    ref = E.vgg19_forward(img, ow)
    for name in ('block1_conv1', 'block2_pool', 'block3_conv1', 'block5_conv4', 'block5_pool'):
        close(net.tap(feats, name), ref[name])
    assert net.tap(feats, 'block5_pool').shape == (n, size // 32, size // 32, 8 * width)
    taps = {}
    for name in ('block1_conv1', 'block2_pool', 'block3_conv1', 'block5_pool'):
        taps[name] = (rng.normal(size=ref[name].shape) / ref[name].size).astype(np.float32)
    got = net.backward({k: dev(v) for k, v in taps.items()})
    # the backward pass against the oracle's ON THE DEVICE'S ACTIVATIONS: ReluGrad masks and pooling arg-maxima are
    # discontinuous, and an activation within rounding of zero (or two window values within rounding of each other)
    # may fall on the other side in float64 -- that would compare two different gradient paths
    dev_feats = {'input': _np(net.tap(feats, 'input')).astype(np.float64)}
    for name in model_vgg.LAYER_NAMES:
        dev_feats[name] = _np(net.tap(feats, name)).astype(np.float64)
    close(got, E.vgg19_backward(dev_feats, ow, taps))
    flips = sum(int(((dev_feats[k] > 0) != (ref[k] > 0)).sum()) for k in model_vgg.LAYER_NAMES if 'conv' in k)
    print('activations on the other side of zero than in float64:', flips)


def _disc_params(rng, width, image_size, units):
    convs, cin = [], 3
    for i in range(5):
        f = width * 2 ** i
        for _ in range(2):
            convs.append((rng.normal(0, np.sqrt(1.5 / (9 * cin)), (3, 3, cin, f)).astype(np.float32),
                          rng.normal(0, 0.05, f).astype(np.float32)))
            cin = f
    feat = (image_size // 32) ** 2 * cin
    dense = [(rng.normal(0, np.sqrt(1.0 / feat), (feat, units)).astype(np.float32), rng.normal(0, 0.05, units).astype(np.float32)),
             (rng.normal(0, np.sqrt(1.0 / units), (units, 1)).astype(np.float32), rng.normal(0, 0.05, 1).astype(np.float32))]
    return convs, dense


@pytest.mark.parametrize('width,size,units,n', [(32, 64, 32, 3), (32, 128, 1024, 2)], ids=['small_image', 'reference_size'])
def test_discriminator_forward_backward(width, size, units, n):
    from ml_super_resolution_amd.enet import model_enet
    rng = np.random.default_rng(7)
    convs, dense = _disc_params(rng, width, size, units)
    D = model_enet.Discriminator(device='cuda', width=width, image_size=size, dense_units=units)
    D.set_params(convs, dense)
    names = list(D.variables())
    assert names[:2] == ['d_/conv2d/kernel', 'd_/conv2d/bias'] and names[-4:] == ['d_/dense/kernel', 'd_/dense/bias', 'd_/dense_1/kernel', 'd_/dense_1/bias']
    assert tuple(D.variables()['d_/conv2d_9/kernel'].shape) == (3, 3, 16 * width, 16 * width)
    x = rng.uniform(-1, 1, (n, size, size, 3)).astype(np.float32)
    p = D.forward(dev(x), keep=True)
    pref, saved = E.discriminator_forward(x, convs, dense, keep=True)
    close(p, pref)
    for a_dev, a_ref in zip(D._saved[0], saved[0]):
        close(to_nhwc_np(a_dev), a_ref)
    dp = rng.normal(size=pref.shape).astype(np.float32)
    dx = D.backward(dev(dp), want_dx=True, want_dw=True)
    # backward on the device's activations (leaky-ReLU slopes switch at zero: see the VGG test)
    dev_saved = ([to_nhwc_np(a).astype(np.float64) for a in D._saved[0]], _np(D._saved[1]).astype(np.float64),
                 _np(D._saved[2]).astype(np.float64))
    dx_ref, cg, dg = E.discriminator_backward(dev_saved, _np(p).astype(np.float64), dp, convs, dense)
    close(dx, dx_ref)
    grads = D.gradients()
    for i, (gk, gb) in enumerate(cg):
        scope = 'd_/conv2d' if i == 0 else 'd_/conv2d_%d' % i
        close(grads[scope + '/kernel'], gk)
        close(grads[scope + '/bias'], gb)
    for i, (gw, gb) in enumerate(dg):
        scope = 'd_/dense' if i == 0 else 'd_/dense_%d' % i
        close(grads[scope + '/kernel'], gw)
        close(grads[scope + '/bias'], gb)


# ---------------------------------------------------------------------------------------------------------------------
# whole training steps
# ---------------------------------------------------------------------------------------------------------------------
def _device_vgg_feats(vgg):
    """The activations the last Vgg19.forward(keep=True) saved, as float64 NHWC arrays keyed like the oracle's."""
    from ml_super_resolution_amd.enet import model_vgg
    return {name: to_nhwc_np(vgg._saved[name]).astype(np.float64) for name in ['input'] + model_vgg.LAYER_NAMES}


def _device_disc_state(D):
    """(p, (acts, flat, h)) of the last Discriminator.forward(keep=True), float64, in the oracle's form."""
    acts, flat, h, p = D._saved
    return _np(p).astype(np.float64), ([to_nhwc_np(a).astype(np.float64) for a in acts], _np(flat).astype(np.float64),
                                       _np(h).astype(np.float64))


def _enet_setup(pat, vgg_width, d_width, size, units, n, seed=11):
    from ml_super_resolution_amd.enet import model_enet, model_vgg
    rng = np.random.default_rng(seed)
    w = model_vgg.random_vgg_weights(seed, vgg_width)
    m = model_enet.EnetModel(pat, w, device='cuda', seed=seed, d_width=d_width, image_size=size, dense_units=units)
    g_pairs = []
    for i, (k, cin, cout) in enumerate(model_enet.generator_layers()):
        g_pairs.append((rng.normal(0, 1.0 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32),
                        rng.uniform(-0.05, 0.05, cout).astype(np.float32)))
    m.generator.set_params(g_pairs)
    convs, dense = _disc_params(rng, d_width, size, units)
    if m.discriminator is not None:
        m.discriminator.set_params(convs, dense)
    hd = rng.uniform(-1, 1, (n, size, size, 3)).astype(np.float32)
    sd = hd.reshape(n, size // 4, 4, size // 4, 4, 3).mean(axis=(2, 4)).astype(np.float32)
    bq = np.repeat(np.repeat(sd, 4, axis=1), 4, axis=2)
    return m, w, g_pairs, convs, dense, sd, bq, hd


@pytest.mark.parametrize('pat,vgg_width,d_width,size,units,n', [('pat', 8, 32, 64, 32, 2), ('pa', 8, 32, 64, 32, 2), ('p', 8, 32, 64, 32, 1),
                                                                ('pat', 64, 32, 128, 1024, 1)],
                         ids=['pat_narrow', 'pa_narrow', 'p_narrow', 'pat_reference_size'])
def test_generator_step_against_oracle(pat, vgg_width, d_width, size, units, n):
    """g_trainer (model_enet.py:336-337): losses, d(g_losses)/d(sr), the generator gradients and one Adam(1e-4) step."""
    m, w, g_pairs, convs, dense, sd, bq, hd = _enet_setup(pat, vgg_width, d_width, size, units, n)
    sr_ref, ins = O.enet_generator_forward(sd, bq, g_pairs, keep=True)
    losses_ref, dsr_ref = E.enet_losses_and_sr_gradient(sr_ref, hd, _oracle_vgg_weights(w), convs, dense, pat)
    # the objective alone (the eager API)
    sr = m.generator.forward(dev(sd), dev(bq), keep=True)
    close(sr, sr_ref)
    d_sr = m.generator_objective(sr, dev(hd), want_a_loss=True)
    for k, v in losses_ref.items():
        assert abs(m.losses[k].item() - v) <= 2e-4 * abs(v) + 1e-9, (k, m.losses[k].item(), v)
    # the gradient: the oracle differentiates at the DEVICE's activations (ReLU masks / pooling arg-maxima are
    # discontinuous; see test_vgg19_features_and_input_gradient)
    dev_feats = _device_vgg_feats(m.vgg)
    at_fake = _device_disc_state(m.discriminator) if m.discriminator is not None else None
    _, dsr_ref = E.enet_losses_and_sr_gradient(_np(sr).astype(np.float64), hd, _oracle_vgg_weights(w), convs, dense, pat,
                                               at_sr_feats=dev_feats, at_fake=at_fake)
    close(d_sr, dsr_ref)
    # the whole trainer run
    before = m.generator.params.clone()
    m.g_step(dev(sd), dev(bq), dev(hd))
    assert m.global_step == 1
    ins = [_np(t).astype(np.float64) for t in m.generator._saved]
    grads_ref = O.enet_generator_backward(ins, _np(d_sr).astype(np.float64), g_pairs)
    for i in (0, 5, 12, 21, 24):
        close(m.generator._gk[i], grads_ref[i][0])
        k0, gk = g_pairs[i][0].astype(np.float64), grads_ref[i][0]
        wk, _, _ = O.adam_tf(k0, gk, np.zeros_like(gk), np.zeros_like(gk), 1e-4, 1)
        np.testing.assert_allclose(_np(m.generator.kernels[i]), wk, rtol=0, atol=2e-6)
    assert not torch.equal(before, m.generator.params)
    if m.discriminator is not None:                   # g_trainer's var_list holds the g_ variables only
        for c, (k, _) in zip(m.discriminator.convs, convs):
            np.testing.assert_array_equal(_np(c.kernel_hwio()), k)


def test_discriminator_step_against_oracle():
    """d_trainer (model_enet.py:339-343): a_loss, gradients of the d_ variables, one Adam(1e-4) step; the generator is
    untouched and the global step does not move."""
    m, w, g_pairs, convs, dense, sd, bq, hd = _enet_setup('pat', 8, 32, 64, 32, 2)
    sr_ref = O.enet_generator_forward(sd, bq, g_pairs)
    a_ref, cg, dg = E.discriminator_loss_and_grads(sr_ref, hd, convs, dense)
    g_before = m.generator.params.clone()
    a_loss = m.d_step(dev(sd), dev(bq), dev(hd))
    assert abs(a_loss.item() - a_ref) <= 1e-5 * abs(a_ref)
    assert m.global_step == 0 and torch.equal(g_before, m.generator.params)
    # gradients at the device's activations (the step ran D on the concatenated batch [fake; real])
    p_all, (acts, flat, h) = _device_disc_state(m.discriminator)
    n = sd.shape[0]
    halves = [(p_all[sl], ([a[sl] for a in acts], flat[sl], h[sl])) for sl in (slice(0, n), slice(n, 2 * n))]
    a_ref, cg, dg = E.discriminator_loss_and_grads(sr_ref, hd, convs, dense, at_fake=halves[0], at_real=halves[1])
    grads = m.discriminator.gradients()
    for i, (gk, gb) in enumerate(cg):
        scope = 'd_/conv2d' if i == 0 else 'd_/conv2d_%d' % i
        close(grads[scope + '/kernel'], gk)
        close(grads[scope + '/bias'], gb)
    close(grads['d_/dense/kernel'], dg[0][0]); close(grads['d_/dense_1/bias'], dg[1][1])
    for i in (0, 3, 9):
        k0, gk = convs[i][0].astype(np.float64), cg[i][0]
        wk, _, _ = O.adam_tf(k0, gk, np.zeros_like(gk), np.zeros_like(gk), 1e-4, 1)
        np.testing.assert_allclose(_np(m.discriminator.convs[i].kernel_hwio()), wk, rtol=0, atol=2e-6)


def test_build_enet_keys_and_train_script_schedule():
    """build_enet's keys (model_enet.py:270-350) and experiment_train's loop (:105-160): a discriminator run on every
    3rd step, a generator run on every step, each on a batch of its own."""
    from ml_super_resolution_amd import graph
    from ml_super_resolution_amd.enet import experiment_train, model_enet, model_vgg
    ph = [graph.placeholder(name=n) for n in ('sd_images', 'bq_images', 'hd_images')]
    w = model_vgg.random_vgg_weights(1, 8)
    full = model_enet.build_enet(ph[0], ph[1], ph[2], 'pat', None, vgg_weights=w)
    assert {k for k in full if not k.startswith('_')} == {'sd_images', 'bq_images', 'sr_images', 'hd_images', 'a_loss', 'g_loss',
                                                        't_loss', 'p_loss', 'g_loss_all', 'g_trainer', 'd_trainer', 'step'}
    p_only = model_enet.build_enet(ph[0], ph[1], ph[2], 'p', None, vgg_weights=w)
    assert {k for k in p_only if not k.startswith('_')} == {'sd_images', 'bq_images', 'sr_images', 'hd_images', 'p_loss',
                                                          'g_loss_all', 'g_trainer', 'step'}
    with graph.Session() as session:
        assert session.run(full['step']) == 0
        rng = np.random.default_rng(0)
        hd = rng.uniform(-1, 1, (1, 128, 128, 3)).astype(np.float32)
        sd = hd.reshape(1, 32, 4, 32, 4, 3).mean(axis=(2, 4)).astype(np.float32)
        bq = np.repeat(np.repeat(sd, 4, axis=1), 4, axis=2)
        feeds = {full['sd_images']: sd, full['bq_images']: bq, full['hd_images']: hd}
        fetched = session.run({'step': full['step'], 'trainer': full['d_trainer']}, feed_dict=feeds)
        assert fetched['step'] == 0
        fetched = session.run({'step': full['step'], 'trainer': full['g_trainer'], 'p': full['p_loss'], 'g': full['g_loss_all']}, feed_dict=feeds)
        assert fetched['step'] == 1 and np.isfinite(fetched['g']) and fetched['g'] >= fetched['p']
        assert session.run(full['sr_images'], feed_dict=feeds).shape == (1, 128, 128, 3)
    log = []
    m = experiment_train.main(['--model', 'pat', '--batch_size', '2', '--stop_training_at_k_step', '5', '--allow_random_vgg', 'true'],
                              log=log.append)
    assert [(r['step'], r['trainer']) for r in log] == [(0, 'd'), (0, 'g'), (1, 'g'), (2, 'g'), (3, 'd'), (3, 'g'), (4, 'g')]
    assert m.global_step == 5 and all(np.isfinite(r.get('g_loss_all', 0.0)) for r in log)
    with pytest.raises(SystemExit):
        experiment_train.main(['--model', 'pat', '--batch_size', '2', '--stop_training_at_k_step', '1'])


@pytest.mark.parametrize('n,hw,cin,cout', [(2, 16, 8, 4), (2, 16, 4, 4), (2, 16, 3, 8), (2, 64, 3, 4)])
def test_generic_kernel_without_aux_operand(n, hw, cin, cout):
    """Regression (round 2): shapes outside the tuned instance set run conv_mfma_generic_kernel, which is always built
    with the aux-operand epilogue; with 16-byte output vectors and NO skip / mask operand its prefetch read through a
    null pointer (a GPU memory fault, found with a 4-channel discriminator).  Data gradients towards <= 4 staged
    channels, and a 4x4 filter forward."""
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(cin * 10 + cout)
    k = rng.normal(0, 0.3, (3, 3, cin, cout)).astype(np.float32)
    dp = rng.normal(size=(n, hw, hw, cout)).astype(np.float32)
    ref, _, _ = E.conv2d_same_bwd(np.zeros((n, hw, hw, cin)), k, dp, 1)
    close(ops.conv2d_bwd_data(dev(dp), dev(k), (n, hw, hw, cin), 'same'), ref)
    x = rng.normal(size=(n, 12, 12, 32)).astype(np.float32)
    w = rng.normal(0, 0.1, (4, 4, 32, 8)).astype(np.float32)
    close(ops.conv2d_fwd(dev(x), dev(w), None, 'valid', None), O.conv2d_fwd(x, w, None, 'VALID'))


@pytest.mark.parametrize('cin,cout,hw,n,act', [(128, 128, 8, 3, 'relu'), (512, 512, 4, 2, 'lrelu'), (256, 128, 33, 1, None),
                                               (128, 256, 64, 1, 'relu'), (192, 64, 7, 2, 'relu'), (64, 192, 16, 2, 'lrelu'),
                                               (128, 128, 130, 1, 'relu'), (128, 64, 65, 1, 'lrelu'),
                                               # several units per workgroup: the step pipeline crosses tiles, images, blocks
                                               (128, 192, 20, 60, 'relu'), (192, 128, 8, 150, 'lrelu')])
def test_one_launch_wide_layer_equals_block_pair_launches(cin, cout, hw, n, act):
    """srx_conv3x3_blocked (one launch, the sum over the input blocks in registers) against the block-pair launches of
    the 64-channel kernels (the running sum through memory) and against the oracle: forward and data gradient."""
    from ml_super_resolution_amd import blocked, ops
    rng = np.random.default_rng(cin + cout + hw)
    k = rng.normal(0, np.sqrt(1.0 / (9 * cin)), (3, 3, cin, cout)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    x = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    ks = blocked.BlockedConv.kernel_shape(cin, cout)
    layer = blocked.BlockedConv(cin, cout, 1, act, torch.empty(ks, device='cuda'), torch.empty(cout, device='cuda'))
    layer.set_kernel_hwio(k, b)
    xb = blocked.to_blocks(dev(x))
    dpre = rng.normal(size=(n, hw, hw, cout)).astype(np.float32)
    dpb = blocked.to_blocks(dev(dpre))
    old = blocked.USE_WIDE
    try:
        blocked.USE_WIDE = True
        y1, d1 = layer.forward(xb), layer.dgrad(dpb)
        blocked.USE_WIDE = False
        y0, d0 = layer.forward(xb), layer.dgrad(dpb)
    finally:
        blocked.USE_WIDE = old
    yref = O.act_apply(E.conv2d_same_fwd(x, k, b, 1), act)
    dref, _, _ = E.conv2d_same_bwd(x, k, dpre, 1)
    close(blocked.to_nhwc(y1), yref); close(blocked.to_nhwc(y0), yref)
    close(blocked.to_nhwc(d1), dref); close(blocked.to_nhwc(d0), dref)
    # same products, but the partial sums of the block-pair route are rounded to fp32 between launches
    assert float((y1 - y0).abs().max()) <= 1e-5 * float(y0.abs().max())
    # the activation gradient of the layer below fused into the data-gradient launch (mask = the layer's saved input)
    for mact in ('relu', 'lrelu'):
        xm = blocked.to_blocks(dev(np.where(rng.uniform(size=x.shape) < 0.5, x, 0).astype(np.float32)))
        want = blocked.to_nhwc(d1) * torch.from_numpy(O.act_grad_from_y(_np(blocked.to_nhwc(xm)).astype(np.float64), mact)).float().cuda()
        for wide in (True, False):
            blocked.USE_WIDE = wide
            try:
                got = layer.dgrad(dpb, mask=xm, mask_act=mact)
            finally:
                blocked.USE_WIDE = old
            close(blocked.to_nhwc(got), _np(want).astype(np.float64))


def test_enet_train_script_checkpoint_and_resume(tmp_path):
    """experiment_train's checkpointing (enet/enet/experiment_train.py:96-117): a tf.train.Saver-format checkpoint
    whenever step % save_every == save_every - 1, restored at the next start -- g_ / d_ variables (kernels HWIO),
    global_step and both optimizers' slots: resuming equals never having stopped, bit for bit."""
    from ml_super_resolution_amd import tf_bundle
    from ml_super_resolution_amd.enet import experiment_train, model_enet, model_vgg
    ckpt = str(tmp_path / 'ckpt')
    common = ['--model', 'pat', '--batch_size', '2', '--allow_random_vgg', 'true', '--ckpt_path', ckpt, '--save_every', '2']
    torch.manual_seed(4321)
    experiment_train.main(common + ['--stop_training_at_k_step', '4'])
    assert sorted(n for n in __import__('os').listdir(ckpt) if n.endswith('.index')) == ['model.ckpt-1.index', 'model.ckpt-3.index']
    assert tf_bundle.latest_checkpoint(ckpt).endswith('model.ckpt-3')
    saved = tf_bundle.load_checkpoint(tf_bundle.latest_checkpoint(ckpt))
    assert int(saved['global_step']) == 3 and saved['d_/conv2d_9/kernel'].shape == (3, 3, 512, 512)
    assert saved['d_/dense/kernel'].shape == (8192, 1024) and saved['g_/conv2d_24/kernel'].shape == (3, 3, 64, 3)
    for key in ('g_/conv2d/kernel/Adam', 'd_/dense_1/bias/Adam_1', 'beta1_power', 'beta2_power_1'):
        assert key in saved, key
    np.testing.assert_allclose(saved['beta1_power'], 0.9 ** 4, rtol=1e-6)       # 3 generator steps
    np.testing.assert_allclose(saved['beta1_power_1'], 0.9 ** 2, rtol=1e-6)     # 1 discriminator step
    log = []
    resumed = experiment_train.main(common + ['--stop_training_at_k_step', '5'], log=log.append)
    assert [(r['step'], r['trainer']) for r in log] == [(3, 'd'), (3, 'g'), (4, 'g')]
    # the same steps on one model object, never saved / loaded
    torch.manual_seed(4321)
    ref = model_enet.EnetModel('pat', model_vgg.random_vgg_weights(0), device='cuda')
    dev_ = torch.device('cuda', torch.cuda.current_device())
    for first, last in ((0, 3), (3, 5)):
        batches = experiment_train.synthetic_batches(2, dev_, seed=0)          # every start restarts its data
        for step in range(first, last):
            if step % 3 == 0:
                ref.d_step(*next(batches))
            ref.g_step(*next(batches))
    assert ref.global_step == resumed.global_step == 5
    assert torch.equal(ref.generator.params, resumed.generator.params)
    assert torch.equal(ref.discriminator.pool.params, resumed.discriminator.pool.params)
    assert torch.equal(ref.discriminator.pool.opt_v, resumed.discriminator.pool.opt_v)
    assert torch.equal(ref.g_state['m'], resumed.g_state['m'])


def test_enet_pat_against_committed_golden_vectors():
    """The committed fixture tests/golden/enet_pat.npz (oracle outputs, generated by tests/golden/make_golden.py):
    VGG-19 features, discriminator outputs and the five losses of build_enet for seeded inputs and weights."""
    import os
    from tests.golden.make_golden import enet_pat_case
    from ml_super_resolution_amd.enet import model_enet
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'enet_pat.npz'))
    vgg, convs, dense, sr, hd = enet_pat_case()
    np.testing.assert_array_equal(sr, z['sr'])
    w = {name: {name + '_W_1': k, name + '_b_1': b} for name, (k, b) in vgg.items()}
    m = model_enet.EnetModel('pat', w, device='cuda', seed=0, d_width=32, image_size=64, dense_units=32)
    m.discriminator.set_params(convs, dense)
    feats = m.vgg.forward(dev(sr))
    for name in ('block1_conv1', 'block2_pool', 'block3_conv1', 'block5_pool'):
        close(m.vgg.tap(feats, name), z['vgg.' + name])
    close(m.discriminator.forward(dev(sr)), z['d.fake'])
    close(m.discriminator.forward(dev(hd)), z['d.real'])
    m.generator_objective(dev(sr), dev(hd), want_grad=False, want_a_loss=True)
    for k in ('p_loss', 't_loss', 'g_loss', 'a_loss', 'g_loss_all'):
        ref = float(z['loss.' + k])
        assert abs(m.losses[k].item() - ref) <= 2e-4 * abs(ref), (k, m.losses[k].item(), ref)


@pytest.mark.parametrize('k,cin,cout,h,w', [(2, 16, 16, 14, 9), (4, 32, 8, 11, 13), (2, 3, 64, 6, 6), (4, 64, 64, 9, 20)])
def test_even_filter_sizes_with_same_padding(k, cin, cout, h, w):
    """Regression (round 2, found by scripts/fuzz_conv.py): TF SAME pads (K-1)/2 before and the rest after, so an even
    filter pads one more column on the right; the tile rows were one slot short for that and read the next row's first
    pixel instead of zero.  (No layer of the reference has an even filter; the C ABI accepts any KH x KW.)"""
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(k * 100 + cin)
    x = rng.normal(size=(2, h, w, cin)).astype(np.float32)
    wt = rng.normal(0, 1.0 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    close(ops.conv2d_fwd(dev(x), dev(wt), dev(b), 'same', 'relu'), O.conv2d_fwd(x, wt, b, 'SAME', 'relu'))
    dpre = rng.normal(size=(2, h, w, cout)).astype(np.float32)
    close(ops.conv2d_bwd_data(dev(dpre), dev(wt), x.shape, 'same'), O.conv2d_bwd_data(dpre, wt, (h, w), 'SAME'))


@pytest.mark.parametrize('cib,cob,n,h,w', [(2, 4, 8, 16, 16), (4, 2, 16, 8, 8), (8, 8, 6, 4, 4), (1, 3, 3, 32, 32), (2, 2, 2, 9, 37),
                                           (2, 1, 1, 6, 130), (1, 1, 2, 12, 12),
                                           # 32-wide rows with enough of them for the planner's tallest tile (7 + 2 rows of 33 slots
                                           # = the whole 80-KiB budget): the pairs launch takes one row less (round 3)
                                           (2, 4, 8, 32, 32), (2, 2, 40, 32, 32),
                                           # rows cut into column strips: all pairs in one launch of the strip body (round 3)
                                           (2, 2, 4, 64, 64), (1, 2, 2, 20, 128), (2, 2, 1, 16, 100), (2, 3, 2, 7, 61)])
def test_blocked_filter_gradient_all_pairs_in_one_launch(cib, cob, n, h, w):
    """srx_conv3x3_blocked_bwd_filter (one launch over the block pairs + one reduction -- full-width tiles or column strips;
    per-pair launches where the linear-walk kernel does not cover the shape, and for a single pair) against one srx_conv2d_bwd_filter
    per pair and against the float64 sums."""
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(cib * 100 + cob * 10 + h)
    x = rng.normal(size=(cib, n, h, w, 64)).astype(np.float32)
    dp = rng.normal(size=(cob, n, h, w, 64)).astype(np.float32)
    xd, dd = dev(x), dev(dp)
    dw = torch.full((cib, cob, 3, 3, 64, 64), float('nan'), device='cuda')
    db = torch.full((cob * 64,), float('nan'), device='cuda')
    ops.conv3x3_blocked_bwd_filter(xd, dd, dw, db)
    ref_db = dp.astype(np.float64).sum(axis=(1, 2, 3)).reshape(-1)
    close(db, ref_db)
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1), (0, 0)))
    for ib in range(cib):
        for ob in range(cob):
            one, _ = ops.conv2d_bwd_filter(xd[ib], dd[ob], (3, 3, 64, 64), 'same')
            assert float((dw[ib, ob] - one).abs().max()) <= 1e-5 * float(one.abs().max())
    # float64 sums for one pair and one tap row (the whole tensor would take a minute in NumPy)
    ib, ob = cib - 1, cob - 1
    for kh in (0, 2):
        for kw in (1,):
            ref = np.einsum('nhwi,nhwo->io', xp[ib][:, kh:kh + h, kw:kw + w, :], dp[ob].astype(np.float64))
            close(dw[ib, ob, kh, kw], ref)
    # without a bias gradient
    dw2 = torch.empty_like(dw)
    ops.conv3x3_blocked_bwd_filter(xd, dd, dw2, None)
    assert torch.equal(dw2, dw)


@pytest.mark.parametrize('c,n,h,w', [(64, 2, 32, 48), (128, 3, 16, 32), (256, 2, 32, 16), (64, 1, 16, 16)])
def test_texture_gram_one_pass_against_the_three_launches_and_float64(c, n, h, w):
    """srx_texture_gram / _bwd (normalise + 16x16 patches + gram matrix in one pass, enet/enet/model_enet.py:34-41,
    225-259) against srx_channel_normalize -> srx_extract_patches16 -> srx_gemm and against float64 NumPy."""
    from ml_super_resolution_amd import ops
    rng = np.random.default_rng(c + h)
    x = np.abs(rng.normal(size=(n, h, w, c))).astype(np.float32) + 0.05       # (post-ReLU features: non-negative)
    xd = dev(x)
    g = ops.texture_gram(xd)
    sp = ops.extract_patches16(ops.channel_normalize(xd)).view(-1, 256, c)
    g3 = ops.gemm(sp, sp, trans_a=True)
    assert float((g - g3).abs().max()) <= 2e-5 * float(g3.abs().max())
    x64 = x.astype(np.float64)
    nrm = x64 / (x64.mean(axis=-1, keepdims=True) + 1e-6)
    pat = nrm.reshape(n, h // 16, 16, w // 16, 16, c).transpose(0, 1, 3, 2, 4, 5).reshape(-1, 256, c)
    gref = np.einsum('pki,pkj->pij', pat, pat)
    close(g, gref)
    # gradient for a symmetric d loss / d gram
    dg = rng.normal(size=gref.shape).astype(np.float32)
    dg = (dg + dg.transpose(0, 2, 1)) * 0.5
    dgd = dev(dg)
    dx = ops.texture_gram_bwd(xd, dgd)
    dsp = ops.gemm(sp, dgd, alpha=2.0)
    dx3 = ops.channel_normalize_bwd(xd, ops.extract_patches16_bwd(dsp.view(n, -1, 256, c), (n, h, w, c)))
    assert float((dx - dx3).abs().max()) <= 2e-5 * float(dx3.abs().max())
    dn = 2.0 * np.einsum('pkj,pjc->pkc', pat, dg.astype(np.float64))
    dn = dn.reshape(n, h // 16, w // 16, 16, 16, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h, w, c)
    m = x64.mean(axis=-1, keepdims=True) + 1e-6
    dref = dn / m - (dn * x64).sum(axis=-1, keepdims=True) / (c * m * m)
    close(dx, dref)


def test_enet_train_script_reads_a_directory_of_images(tmp_path):
    """`--train_dir_path <directory>`: the reference's own data path (enet/enet/datasets.py) feeding the trainers --
    three generator steps and the discriminator step of the schedule on crops of real files."""
    from PIL import Image
    from ml_super_resolution_amd.enet import experiment_train
    rng = np.random.default_rng(3)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(str(tmp_path / ('img%d.png' % i)))
    log = []
    m = experiment_train.main(['--model', 'pat', '--batch_size', '2', '--stop_training_at_k_step', '3', '--allow_random_vgg', 'true',
                               '--train_dir_path', str(tmp_path)], log=log.append)
    assert m.global_step == 3
    assert all(np.isfinite(v) for rec in log for v in rec.values() if isinstance(v, float))
