"""VDSR end to end on the GPU through the reference-shaped API (build_model + Session.run)
against the oracle's whole-net golden vectors: forward, loss, every gradient, one TF-Adam step."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.golden.make_golden import vdsr_params
from tests.test_gpu_ops import close, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def built():
    from ml_super_resolution_amd import graph
    from ml_super_resolution_amd.vdsr import model_vdsr
    sd = graph.placeholder([None, None, None, 3], name='sd_images')
    hd = graph.placeholder([None, None, None, 3], name='hd_images')
    model = model_vdsr.build_model(sd, hd, num_layers=20, use_adam=True)
    model['_model'].stack.set_params(vdsr_params(106))
    return graph, model


def test_result_dict_keys_match_reference(built):
    _, model = built
    keys = set(k for k in model if not k.startswith('_'))
    want = {'conv.%d' % i for i in range(1, 21)} | {'relu.%d' % i for i in range(1, 20)}
    want |= {'sd_images', 'sr_images', 'step', 'loss', 'trainer', 'hd_images', 'learning_rate'}
    assert keys == want
    names = model['_model'].stack.variables().keys()
    assert 'conv2d/kernel' in names and 'conv2d_19/bias' in names and len(names) == 40


def test_forward_loss_grads_adam_vs_golden(built, golden_nets):
    graph, model = built
    g = golden_nets
    m = model['_model']
    params = vdsr_params(106)
    m.stack.set_params(params)
    m.stack.global_step = 0
    m.stack.opt_m = m.stack.opt_v = None
    with graph.Session() as session:
        fetched = session.run({'sr': model['sr_images'], 'c1': model['conv.1'], 'r1': model['relu.1'],
                               'c10': model['conv.10'], 'c19': model['conv.19'], 'c20': model['conv.20'],
                               'loss': model['loss']},
                              feed_dict={model['sd_images']: g['vdsr.sd'], model['hd_images']: g['vdsr.hd']})
    close(fetched['sr'], g['vdsr.sr'])
    close(fetched['c1'][:, :8, :8], g['vdsr.conv_1'])
    close(fetched['c10'][:, :8, :8], g['vdsr.conv_10'])
    close(fetched['c19'][:, :8, :8], g['vdsr.conv_19'])
    np.testing.assert_array_equal(fetched['c1'], fetched['r1'])        # pin P2: conv.N tap is post-ReLU
    assert fetched['c1'].min() >= 0
    close(fetched['c20'], g['vdsr.sr'].astype(np.float64) - g['vdsr.sd'], 2e-3)
    assert abs(fetched['loss'] - float(g['vdsr.loss'])) <= 1e-4 * float(g['vdsr.loss'])

    # one training step at lr 5e-5 (vdsr/makefile:27), fetched exactly as experiment_train.py:134-151 does
    with graph.Session() as session:
        out = session.run({'step': model['step'], 'loss': model['loss'], 'trainer': model['trainer']},
                          feed_dict={model['sd_images']: g['vdsr.sd'], model['hd_images']: g['vdsr.hd'],
                                     model['learning_rate']: 5e-5})
    assert out['step'] == 1
    assert abs(out['loss'] - float(g['vdsr.loss'])) <= 1e-4 * float(g['vdsr.loss'])
    st = m.stack
    for i in (0, 1, 9, 18, 19):
        close(st.kernel(i, st.grads), g['vdsr.dk_%d' % i])
    db = np.concatenate([st.bias(i, st.grads).cpu().numpy().ravel() for i in range(20)])
    close(db, g['vdsr.db_all'])
    sums = np.array([st.kernel(i, st.grads).double().sum().item() for i in range(20)])
    asums = np.array([st.kernel(i, st.grads).double().abs().sum().item() for i in range(20)])
    assert np.abs(sums - g['vdsr.dk_sums']).max() <= 1e-3 * g['vdsr.dk_abs_sums'].max()
    assert np.abs(asums / g['vdsr.dk_abs_sums'] - 1).max() <= 1e-3
    # Adam (TF epsilon-hat) moved every weight by ~lr in the right direction
    close(st.kernel(0), g['vdsr.adam_k0'], 1e-5)
    new_sums = np.array([[st.kernel(i).double().sum().item(), st.bias(i).double().sum().item()] for i in range(20)])
    assert np.abs(new_sums - g['vdsr.adam_sums']).max() <= 2e-3


def test_momentum_clip_path(built, golden_nets):
    """use_adam=False: Momentum(0.9) on gradients clipped to +-0.01/lr (model_vdsr.py:158-184)."""
    from ml_super_resolution_amd.vdsr import model_vdsr
    g = golden_nets
    m = model_vdsr.VdsrModel(num_layers=20, use_adam=False)
    params = vdsr_params(106)
    m.stack.set_params(params)
    lr = 0.1
    m.train_step(dev(g['vdsr.sd']), dev(g['vdsr.hd']), lr)
    _, grads, _ = O.vdsr_loss_and_grads(g['vdsr.sd'], g['vdsr.hd'], params)
    for i in (0, 9, 19):
        wref, _ = O.momentum_clip(params[i][0].astype(np.float64), grads[i][0], 0.0, lr)
        close(m.stack.kernel(i), wref, 1e-5)
    assert m.stack.global_step == 1


def test_kernel_families_agree_bit_for_bit():
    """The two forward/dgrad kernel families (srx_set_conv_path) compute every output element with the
    same fp32 operation order: identical bits, so switching is purely a tuning decision."""
    from ml_super_resolution_amd import _lib
    from ml_super_resolution_amd.vdsr import model_vdsr
    m = model_vdsr.VdsrModel(num_layers=20, use_adam=True, seed=9)
    gen = torch.Generator(device='cuda').manual_seed(2)
    hd = torch.rand((16, 41, 41, 3), device='cuda', generator=gen) * 2 - 1
    sd = (hd + 0.1 * torch.randn((16, 41, 41, 3), device='cuda', generator=gen)).clamp(-1, 1)
    outs = []
    for path in (0, 1):
        old = _lib.lib().srx_set_conv_path(path)
        try:
            m.stack.forward(sd, keep=True)
            m.stack.loss_and_backward(hd)
            outs.append((m.stack.acts[-1].clone(), m.stack.grads.clone()))
        finally:
            _lib.lib().srx_set_conv_path(old)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    # the two optional stream arrangements of the backward pass (wgrads / partial reductions on a side stream) only
    # reorder launches: same gradient bits
    for attr in ('overlap_wgrad', 'overlap_reduce'):
        setattr(m.stack, attr, True)
        try:
            m.stack.forward(sd, keep=True)
            m.stack.loss_and_backward(hd)
            torch.cuda.synchronize()
            assert torch.equal(m.stack.grads, outs[1][1]), attr
        finally:
            setattr(m.stack, attr, False)


def test_full_size_batch_properties():
    """BASELINE config 3 shape (256x41x41): size-independent properties -- batch elements are
    independent (a shard of the batch gives the same rows), determinism, gradient additivity."""
    from ml_super_resolution_amd.vdsr import model_vdsr
    m = model_vdsr.VdsrModel(num_layers=20, use_adam=True, seed=3)
    gen = torch.Generator(device='cuda').manual_seed(5)
    hd = torch.rand((256, 41, 41, 3), device='cuda', generator=gen) * 2 - 1
    sd = (hd + 0.1 * torch.randn((256, 41, 41, 3), device='cuda', generator=gen)).clamp(-1, 1)
    sr = m.forward(sd).clone()
    sr_part = m.forward(sd[100:116].contiguous()).clone()
    assert torch.equal(sr[100:116], sr_part)                   # same bits whatever the tiling / batch
    sr2 = m.forward(sd).clone()
    assert torch.equal(sr, sr2)
    # oracle check on a strided sample of the batch
    params = [(m.stack.kernel(i).cpu().numpy(), m.stack.bias(i).cpu().numpy()) for i in range(20)]
    ref = O.c_vdsr_forward(sd[::64].cpu().numpy(), params)
    close(sr[::64], ref)
    # gradient of the mean loss over the batch = mean of the two half-batch gradients
    m.stack.forward(sd, keep=True); m.stack.loss_and_backward(hd); g_full = m.stack.grads.clone()
    m.stack.forward(sd[:128].contiguous(), keep=True); m.stack.loss_and_backward(hd[:128].contiguous()); g_a = m.stack.grads.clone()
    m.stack.forward(sd[128:].contiguous(), keep=True); m.stack.loss_and_backward(hd[128:].contiguous()); g_b = m.stack.grads.clone()
    close(0.5 * (g_a + g_b), g_full.cpu().numpy(), 1e-3)
    m.stack.forward(sd, keep=True); m.stack.loss_and_backward(hd)
    assert torch.equal(m.stack.grads, g_full)                  # deterministic wgrad


def test_short_training_run_learns():
    """Functional end-to-end check: a small VDSR trained for a few dozen Adam steps on degraded smooth
    images must lower its loss and beat the degraded input's PSNR (forward, all backward kernels, loss,
    optimizer and the on-device degradation working together)."""
    from ml_super_resolution_amd import ops
    from ml_super_resolution_amd.vdsr import dataset, model_vdsr
    torch.manual_seed(0)
    # smooth synthetic "images": low-frequency random fields in [0,1]
    base = torch.rand((32, 6, 6, 3), device='cuda')
    hd01 = ops.resize_bilinear(base.contiguous(), 41, 41).clamp(0, 1).contiguous()
    sd01 = dataset.degrade_on_device(hd01, 3.0)
    hd, sd = ops.affine(hd01, 2.0, -1.0), ops.affine(sd01, 2.0, -1.0)
    m = model_vdsr.VdsrModel(num_layers=8, use_adam=True, seed=4)
    losses = []
    for step in range(80):
        losses.append(m.train_step(sd, hd, 1e-3).item())
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    sr = m.forward(sd)
    p_sd = ops.psnr(hd, sd, 2.0).mean().item()
    p_sr = ops.psnr(hd, sr, 2.0).mean().item()
    assert p_sr > p_sd + 0.5, (p_sd, p_sr)
    assert m.stack.global_step == 80


@pytest.mark.parametrize('ground_truth', [True, False])
def test_experiment_resolve_script(ground_truth, tmp_path):
    """vdsr/vdsr/experiment_resolve.py: TF-format checkpoint in, PNG of truncating saturate_cast bytes out; ground-truth
    mode degrades the given image first (:25-26), the other mode up-scales it bilinearly (:28-35)."""
    from PIL import Image
    from ml_super_resolution_amd.vdsr import dataset, experiment_resolve, model_vdsr
    layers = 5
    params = vdsr_params(77, layers)
    m = model_vdsr.VdsrModel(layers, device='cuda')
    m.stack.set_params(params)
    prefix = str(tmp_path / 'model.ckpt-10')
    m.stack.save_tf_checkpoint(prefix)
    img = np.random.default_rng(5).integers(0, 256, (23, 31, 3), dtype=np.uint8)
    src, out = str(tmp_path / 'in.png'), str(tmp_path / 'out.png')
    Image.fromarray(img).save(src)
    experiment_resolve.main(['--ckpt_path', prefix, '--hd_image_path', src, '--sr_image_path', out, '--num_layers', str(layers),
                             '--scaling_factor', '2', '--ground_truth_mode', 'true' if ground_truth else 'false'])
    hd = img.astype(np.float32) / np.float32(255.0)
    if ground_truth:
        sd = dataset.hd_image_to_sd_image(hd, 2.0)
    else:
        sd = O.resize_bilinear(hd[None], 46, 62)[0].astype(np.float32)
    sr_ref = O.vdsr_forward((sd * 2.0 - 1.0)[None].astype(np.float32), params)['sr_images']
    want = O.saturate_u8(sr_ref.astype(np.float32))[0].astype(np.int32)
    got = np.asarray(Image.open(out)).astype(np.int32)
    assert got.shape == want.shape == ((23, 31, 3) if ground_truth else (46, 62, 3))
    # fp32 device arithmetic vs the float64 oracle: a value next to an integer may truncate to the neighbouring byte
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01
