"""Data-parallel host logic on CPU: two gloo ranks, each with half of the batch; the averaged flat
gradient must equal the single-process gradient of the whole batch, and replicas must
start identical.  The per-shard gradients come from the oracle here -- the GPU kernels are covered
by the -m gpu tests; this test covers dist.py (shard / broadcast / all-reduce(mean) / hook)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _FakeStack(object):
    """The slice of ConvStack that dist.attach() touches."""

    def __init__(self, params):
        self.params = params
        self.grad_hook = None


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ml_super_resolution_amd import dist as srx_dist
    from oracle import oracle as O
    from tests.golden.make_golden import vdsr_params
    backend = srx_dist.init_process_group(rank, world, backend='gloo')
    assert backend == 'gloo'
    params = vdsr_params(300, num_layers=4)
    flat0 = np.concatenate([np.concatenate([k.ravel(), b.ravel()]) for k, b in params])
    # rank 1 starts from garbage: attach() must broadcast rank 0's parameters
    flat = torch.from_numpy(flat0.copy() if rank == 0 else np.zeros_like(flat0))
    stack = _FakeStack(flat)
    srx_dist.attach(stack, world)
    assert np.array_equal(stack.params.numpy(), flat0)

    rng = np.random.default_rng(7)
    hd = rng.uniform(-1, 1, (4, 9, 9, 3)).astype(np.float32)
    sd = rng.uniform(-1, 1, (4, 9, 9, 3)).astype(np.float32)
    my_sd = srx_dist.shard(torch.from_numpy(sd), rank, world).numpy()
    my_hd = srx_dist.shard(torch.from_numpy(hd), rank, world).numpy()
    assert my_sd.shape[0] == 2
    _, grads, _ = O.vdsr_loss_and_grads(my_sd, my_hd, params)         # local mean loss + L2 term
    g = torch.from_numpy(np.concatenate([np.concatenate([dk.ravel(), db.ravel()]) for dk, db in grads]))
    stack.grad_hook(g)                                                   # the DP exchange: ONE all-reduce
    np.save(os.path.join(out_dir, 'g%d.npy' % rank), g.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_equals_single_process(tmp_path):
    from oracle import oracle as O
    from tests.golden.make_golden import vdsr_params
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = np.load(tmp_path / 'g0.npy')
    g1 = np.load(tmp_path / 'g1.npy')
    np.testing.assert_array_equal(g0, g1)                                # every rank holds the same averaged gradient
    params = vdsr_params(300, num_layers=4)
    rng = np.random.default_rng(7)
    hd = rng.uniform(-1, 1, (4, 9, 9, 3)).astype(np.float32)
    sd = rng.uniform(-1, 1, (4, 9, 9, 3)).astype(np.float32)
    _, grads, _ = O.vdsr_loss_and_grads(sd, hd, params)                  # single process, whole batch
    ref = np.concatenate([np.concatenate([dk.ravel(), db.ravel()]) for dk, db in grads])
    np.testing.assert_allclose(g0, ref, rtol=1e-9, atol=1e-12)


def test_shard_rejects_ragged_batches():
    import pytest
    from ml_super_resolution_amd import dist as srx_dist
    with pytest.raises(ValueError):
        srx_dist.shard(torch.zeros(5, 2), 0, 2)


def _flat_worker(rank, world, port, out_dir):
    """dist.attach_flat on the real EnetModel object (host tensors, no kernels): rank 0 resumed from a checkpoint
    (step 6, slots), rank 1 did not (step 5, other weights, no slots) -- or the other way round for `case` 1."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ml_super_resolution_amd import dist as srx_dist
    from ml_super_resolution_amd.enet import model_enet, model_vgg
    srx_dist.init_process_group(rank, world, backend='gloo')
    for case in (0, 1):
        m = model_enet.EnetModel('pat', model_vgg.random_vgg_weights(0, width=4), device='cpu', seed=10 * rank + case,
                                 d_width=4, image_size=32, dense_units=8)
        G, P = m.generator, m.discriminator.pool
        has_state = (rank == 0) == (case == 0)
        if has_state:
            gen = torch.Generator().manual_seed(5 + rank)
            m.global_step = 6 + rank
            m.g_state.update({'t': 6 + rank, 'm': torch.randn(G.params.shape, generator=gen), 'v': torch.rand(G.params.shape, generator=gen)})
            P.t = 2 + rank
            P.opt_m, P.opt_v = torch.randn(P.params.shape, generator=gen), torch.rand(P.params.shape, generator=gen)
        else:
            m.global_step = 5
        srx_dist.attach_flat(m, world)
        out = {'g': G.params.numpy(), 'd': P.params.numpy(), 'step': np.int64(m.global_step), 'has_g': np.int64(bool(m.g_state)),
               'has_d': np.int64(P.opt_m is not None), 'd_t': np.int64(P.t)}
        if m.g_state:
            out.update(g_t=np.int64(m.g_state['t']), g_m=m.g_state['m'].numpy(), g_v=m.g_state['v'].numpy(), d_m=P.opt_m.numpy(), d_v=P.opt_v.numpy())
        # the hooks average over the ranks
        g = torch.full((8,), float(rank + 1))
        m.grad_hook_g(g)
        m.grad_hook_d(g)
        out['hooked'] = g.numpy()
        np.savez(os.path.join(out_dir, 'flat%d_%d.npz' % (case, rank)), **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_attach_flat_levels_step_counts_and_adam_slots(tmp_path):
    """ADVICE r2 (dist.py:115): EnhanceNet's `global_step` decides which steps run d_trainer
    (enet/enet/experiment_train.py:112); ranks that disagree issue different collectives.  Everything comes from rank 0 --
    also when rank 0 is the one WITHOUT optimizer state (the other rank's slots are dropped)."""
    port = _free_port()
    mp.spawn(_flat_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / 'flat0_0.npz'), np.load(tmp_path / 'flat0_1.npz')
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert (int(a['step']), int(a['g_t']), int(a['d_t']), int(a['has_g']), int(a['has_d'])) == (6, 6, 2, 1, 1)
    np.testing.assert_array_equal(a['hooked'], np.full(8, 1.5, np.float32))          # mean of 1 and 2, reduced twice: still 1.5
    a, b = np.load(tmp_path / 'flat1_0.npz'), np.load(tmp_path / 'flat1_1.npz')
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert (int(b['step']), int(b['has_g']), int(b['has_d']), int(b['d_t'])) == (5, 0, 0, 0)
