"""Host-side logic that needs no GPU: variable fetches without feeds, checkpoint naming / state file / beta powers,
flag parsing.  (The C ABI's threading promise is checked in tests/test_tsan_host.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_and_learning_rate_fetch_need_no_feed():
    """`step = session.run(model['step'])` is the first line of the reference's loops
    (vdsr/vdsr/experiment_train.py:126, espcn/espcn/experiment_train.py:92): no placeholder is fed."""
    from ml_super_resolution_amd import graph
    from ml_super_resolution_amd.espcn import model_espcn
    from ml_super_resolution_amd.vdsr import model_vdsr
    sd, hd = graph.placeholder(name='sd'), graph.placeholder(name='hd')
    model = model_vdsr.build_model(sd, hd, num_layers=3, use_adam=True, device='cpu', seed=0)
    with graph.Session() as session:
        assert session.run(model['step']) == 0
        assert session.run(model['learning_rate']) == pytest.approx(0.1)        # model_vdsr.py:136-141
        assert session.run({'s': model['step']}, feed_dict={model['learning_rate']: 0.01}) == {'s': 0}
        with pytest.raises(ValueError, match='sd_images must be fed'):
            session.run(model['sr_images'])
        with pytest.raises(ValueError, match='sd_images must be fed'):
            session.run([model['step'], model['loss']])
    lr_src, hr_t = graph.placeholder(name='lr'), graph.placeholder(name='hr')
    em = model_espcn.build_model(lr_src, 3, hr_t, device='cpu', seed=0)
    with graph.Session() as session:
        assert session.run(em['step']) == 0
        with pytest.raises(ValueError, match='lr_source must be fed'):
            session.run(em['sr_result'])


def test_tf_checkpoint_names_state_file_and_beta_powers(tmp_path):
    from ml_super_resolution_amd import tf_bundle
    from ml_super_resolution_amd.engine import ConvStack
    from ml_super_resolution_amd.vdsr import model_vdsr
    stack = ConvStack(model_vdsr.layer_specs(3), device='cpu', residual=True, weight_decay=1e-4)
    g = torch.Generator().manual_seed(0)
    stack.params.copy_(torch.randn(stack.flat_size, generator=g))
    stack.opt_m = torch.randn(stack.flat_size, generator=g)
    stack.opt_v = torch.rand(stack.flat_size, generator=g)
    stack.global_step = 7
    d = tmp_path / 'ckpt'
    d.mkdir()
    assert tf_bundle.latest_checkpoint(str(d)) is None
    stack.save_tf_checkpoint(str(d / 'model.ckpt-7'), extra={'learning_rate': np.float32(0.1)})
    vals = tf_bundle.load_checkpoint(str(d / 'model.ckpt-7'))
    # every global variable of the reference's graph: 3 x (kernel, bias) x (value, Adam, Adam_1) + beta powers +
    # global_step + learning_rate (vdsr/vdsr/model_vdsr.py:136-147)
    assert len(vals) == 3 * 2 * 3 + 4
    for key in ('conv2d/kernel', 'conv2d_1/bias', 'conv2d_2/kernel/Adam', 'conv2d_2/bias/Adam_1', 'learning_rate'):
        assert key in vals
    # TF's AdamOptimizer starts the powers at beta and multiplies once per step: beta ** (N + 1)
    np.testing.assert_allclose(vals['beta1_power'], 0.9 ** 8, rtol=1e-6)
    np.testing.assert_allclose(vals['beta2_power'], 0.999 ** 8, rtol=1e-6)
    assert vals['global_step'].dtype == np.int64 and int(vals['global_step']) == 7
    # the `checkpoint` state file (what tf.train.latest_checkpoint reads, experiment_train.py:108)
    text = open(str(d / 'checkpoint')).read()
    assert text == 'model_checkpoint_path: "model.ckpt-7"\nall_model_checkpoint_paths: "model.ckpt-7"\n'
    stack.global_step = 9
    stack.save_tf_checkpoint(str(d / 'model.ckpt-9'))
    text = open(str(d / 'checkpoint')).read().splitlines()
    assert text[0] == 'model_checkpoint_path: "model.ckpt-9"'
    assert text[1:] == ['all_model_checkpoint_paths: "model.ckpt-7"', 'all_model_checkpoint_paths: "model.ckpt-9"']
    assert tf_bundle.latest_checkpoint(str(d)) == str(d / 'model.ckpt-9')
    # restore: weights, slots, step
    other = ConvStack(model_vdsr.layer_specs(3), device='cpu', residual=True, weight_decay=1e-4)
    other.load_tf_checkpoint(str(d / 'model.ckpt-7'))
    assert other.global_step == 7
    for i in range(3):          # (the flat buffers' alignment padding is not part of any variable)
        for buf_a, buf_b in ((None, None), (other.opt_m, stack.opt_m), (other.opt_v, stack.opt_v)):
            assert torch.equal(other.kernel(i, buf_a), stack.kernel(i, buf_b))
            assert torch.equal(other.bias(i, buf_a), stack.bias(i, buf_b))


def test_use_adam_flag_forms():
    """tf.app.flags booleans: bare `--use_adam` (vdsr/makefile:26), `--use_adam=false`, `--use_adam false`."""
    from ml_super_resolution_amd.vdsr.experiment_train import parse_flags
    assert parse_flags([]).use_adam is True
    assert parse_flags(['--use_adam']).use_adam is True
    assert parse_flags(['--use_adam', '--batch_size', '8']).batch_size == 8
    assert parse_flags(['--use_adam=false']).use_adam is False
    assert parse_flags(['--use_adam', 'False']).use_adam is False


def test_forward_buffers_do_not_pile_up_per_image_size():
    """keep=False temporaries are keyed by (parity, channels): a new image size replaces the old buffers."""
    from ml_super_resolution_amd.engine import ConvStack
    from ml_super_resolution_amd.vdsr import model_vdsr
    stack = ConvStack(model_vdsr.layer_specs(6), device='cpu', residual=True)
    shapes = stack._shapes((1, 30, 20, 3))
    for i, s in enumerate(stack.specs):
        stack._buf(('tmp', i & 1, s.cout), shapes[i])
    n = len(stack._bufs)
    for hw in ((31, 21), (50, 60), (8, 8)):
        shapes = stack._shapes((1,) + hw + (3,))
        for i, s in enumerate(stack.specs):
            stack._buf(('tmp', i & 1, s.cout), shapes[i])
    assert len(stack._bufs) == n == 3


def test_blocked_kernel_layout_and_enet_variable_names():
    """Channel-blocked filters <-> TensorFlow's HWIO, and the variable names / checkpoint keys of the EnhanceNet
    training graph (enet/enet/model_enet.py:118-162, 331-343) -- host logic, no kernels."""
    from ml_super_resolution_amd.blocked import BlockedConv, ParamPool
    from ml_super_resolution_amd.enet import model_enet
    rng = np.random.default_rng(0)
    for cin, cout in ((3, 32), (64, 128), (256, 192), (32, 64)):
        shape = BlockedConv.kernel_shape(cin, cout)
        layer = BlockedConv(cin, cout, 1, 'relu', torch.empty(shape), torch.empty(cout))
        k = rng.normal(size=(3, 3, cin, cout)).astype(np.float32)
        layer.set_kernel_hwio(k, np.arange(cout, dtype=np.float32))
        np.testing.assert_array_equal(layer.kernel_hwio().numpy(), k)
        # block [ib][ob] is the HWIO sub-filter of input channels 64 ib.. and output channels 64 ob..
        ib, ob = shape[0] - 1, shape[1] - 1
        np.testing.assert_array_equal(layer.w[ib, ob].numpy(), k[:, :, 64 * ib:64 * ib + shape[4], 64 * ob:64 * ob + shape[5]])
    pool = ParamPool([(3, 3, 3, 32), (32,), (5,)], 'cpu')
    assert pool.params.numel() == 864 + 32 + 8 and pool.view(2).shape == (5,)
    assert pool.view(1).data_ptr() % 16 == 0 and pool.view(2).data_ptr() % 16 == 0
    D = model_enet.Discriminator(device='cpu', seed=0, width=32, image_size=128, dense_units=1024)
    v = D.variables()
    assert list(v)[:4] == ['d_/conv2d/kernel', 'd_/conv2d/bias', 'd_/conv2d_1/kernel', 'd_/conv2d_1/bias']
    assert tuple(v['d_/conv2d/kernel'].shape) == (3, 3, 3, 32) and tuple(v['d_/conv2d_9/kernel'].shape) == (3, 3, 512, 512)
    assert tuple(v['d_/dense/kernel'].shape) == (8192, 1024) and tuple(v['d_/dense_1/kernel'].shape) == (1024, 1)
    assert [s for _, _, s in model_enet.discriminator_layers()] == [1, 2] * 5
    assert [c for _, c, _ in model_enet.discriminator_layers()] == [32, 32, 64, 64, 128, 128, 256, 256, 512, 512]
    # truncated_normal(0.02): nothing beyond two sigma
    assert float(v['d_/conv2d_9/kernel'].abs().max()) <= 0.04 + 1e-6


def _small_enet(seed=0):
    from ml_super_resolution_amd.enet import model_enet, model_vgg
    return model_enet.EnetModel('pat', model_vgg.random_vgg_weights(0, width=4), device='cpu', seed=seed, d_width=4,
                                image_size=32, dense_units=8)


@pytest.mark.parametrize('g_steps', [999, 5000, 100000])
def test_enet_resume_step_counts_survive_beta_power_underflow(tmp_path, g_steps):
    """enet/enet/experiment_train.py:99-160 saves at step % 1000 == 999; float32 0.9 ** 1000 is 0.0 (TensorFlow would
    store the same), so the Adam step counts must not be recovered by inverting beta1_power."""
    from ml_super_resolution_amd import tf_bundle
    m = _small_enet()
    G, P = m.generator, m.discriminator.pool
    gen = torch.Generator().manual_seed(1)
    m.global_step = g_steps
    m.g_state.update({'t': g_steps, 'm': torch.randn(G.params.shape, generator=gen), 'v': torch.rand(G.params.shape, generator=gen)})
    P.t = (g_steps + 2) // 3 + 5                  # (not what the schedule would give: must come back as stored)
    P.opt_m, P.opt_v = torch.randn(P.params.shape, generator=gen), torch.rand(P.params.shape, generator=gen)
    tensors = m.tf_checkpoint_tensors()
    if g_steps >= 999:
        assert float(tensors['beta1_power']) == 0.0       # the value that used to raise OverflowError on resume
    prefix = str(tmp_path / ('model.ckpt-%d' % g_steps))
    m.save_tf_checkpoint(prefix)
    other = _small_enet(seed=5)
    other.load_tf_checkpoint(prefix)
    assert other.global_step == g_steps and other.g_state['t'] == g_steps and other.discriminator.pool.t == P.t
    for (name, val, am, av, _, _), (_, val2, bm, bv, _, _) in zip(m._named_buffers(), other._named_buffers()):
        # (views of the variables: the flat buffers' alignment padding belongs to no variable)
        assert torch.equal(val, val2) and torch.equal(am, bm) and torch.equal(av, bv), name
    # a TensorFlow-written file has no explicit count: beta2_power_1 while it is a normal float, the schedule after that
    del tensors['srx/d_trainer_steps']
    prefix2 = str(tmp_path / 'foreign.ckpt')
    tf_bundle.save_checkpoint(prefix2, tensors)
    third = _small_enet(seed=6)
    third.load_tf_checkpoint(prefix2)
    assert third.g_state['t'] == g_steps
    # (P.t is 5 off the schedule: beta2_power_1 contradicts it and wins while it is a normal float32)
    if 0.999 ** (P.t + 1) > 1.2e-38:
        assert third.discriminator.pool.t == P.t
    else:
        assert third.discriminator.pool.t == (g_steps + 2) // 3


def test_enet_d_steps_guards():
    from ml_super_resolution_amd.enet.model_enet import EnetModel
    f = EnetModel._d_steps_from_checkpoint

    def tf_accumulator(t):
        # what TensorFlow holds after t applies: created as float32(0.999), multiplied by float32(0.999) t times, in float32
        b, p = np.float32(0.999), np.float32(0.999)
        for _ in range(t):
            p = np.float32(p * b)
        return p
    for t in (0, 1, 333, 1667, 20000, 80000):
        # a schedule that does not fit (it would give t + 10): the accumulator decides, and float32's base is the one inverted
        assert f({'beta2_power_1': tf_accumulator(t)}, 3 * (t + 10) - 2) == t
    # a checkpoint the reference wrote: d_trainer ran on steps 0, 3, 6, ... -> the schedule's count, exactly, whatever the
    # accumulator's float32 drift (round-3 advisor: log(0.999) instead of log(float32(0.999)) is off by one from ~40k steps)
    for g in (999, 29999, 119999, 299999):
        t = (g + 2) // 3
        assert f({'beta2_power_1': np.float32(0.999 ** (t + 1))}, g) == t
        assert f({'beta2_power_1': np.float32(float(np.float32(0.999)) ** (t + 1))}, g) == t
    for bad in (np.float32(0.0), np.float32(1e-42), np.float32('nan'), np.float32(1.0)):
        assert f({'beta2_power_1': bad, 'beta1_power_1': np.float32(0.0)}, 2999) == 1000
    assert f({}, 10) == 4
    assert f({'srx/d_trainer_steps': np.int64(77), 'beta2_power_1': np.float32(0.5)}, 10) == 77


def test_load_vgg_weights_reads_the_keras_named_npz(tmp_path):
    """enet/enet/model_vgg.py:39-62: the .npz holds `<layer>_W_1:0` / `<layer>_b_1:0`; scope = first 12 characters,
    constant name = the name without ':0'.  The path a user with the real file takes: npz -> load_vgg_weights -> Vgg19."""
    from ml_super_resolution_amd.enet import model_vgg
    w = model_vgg.random_vgg_weights(3, width=8)
    arrays = {}
    for layer, d in w.items():
        for const, a in d.items():
            arrays[const + ':0'] = a
    path = str(tmp_path / 'vgg19_weights_tf_dim_ordering_tf_kernels_notop.npz')
    np.savez(path, **arrays)
    got = model_vgg.load_vgg_weights(path)
    assert sorted(got) == sorted(n for n in model_vgg.LAYER_NAMES if 'conv' in n) and len(got) == 16
    assert sorted(got['block3_conv4']) == ['block3_conv4_W_1', 'block3_conv4_b_1']
    net = model_vgg.Vgg19(got, device='cpu')
    for layer in w:
        np.testing.assert_array_equal(net.layers[layer].kernel_hwio().numpy(), w[layer][layer + '_W_1'])
        np.testing.assert_array_equal(net.layers[layer].b.numpy(), w[layer][layer + '_b_1'])
    assert model_vgg.load_vgg_weights(str(tmp_path / 'missing.npz')) == {}          # :45-46
    from ml_super_resolution_amd import graph
    from ml_super_resolution_amd.enet import model_enet
    with pytest.raises(ValueError, match='VGG-19 weights not found'):
        model_enet.build_enet(graph.placeholder(name='sd'), graph.placeholder(name='bq'), graph.placeholder(name='hd'),
                              'pat', str(tmp_path / 'missing.npz'), device='cpu')
