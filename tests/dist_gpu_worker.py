"""One rank of the data-parallel GPU rehearsal (started by tests/test_gpu_dist.py as a fresh child process).

Builds the REAL model (VdsrModel on libsrx.so), attaches dist.py's gradient exchange, runs `steps` train steps on
its shard of a seeded global batch and saves its parameters / optimizer slots / averaged gradient.  With
SRX_DIST_BACKEND=gloo several ranks share one GPU (RCCL needs a GPU per rank), which exercises everything of the
N > 1 path except the RCCL transport itself.

  python tests/dist_gpu_worker.py OUT_DIR USE_ADAM STEPS LAYERS GLOBAL_BATCH LR
  python tests/dist_gpu_worker.py OUT_DIR enet|enet_resumed GLOBAL_BATCH
SRX_TEST_GROUP_OF_ONE=1: a process group even at world size 1 (with SRX_DIST_BACKEND unset that is RCCL: the `nccl`
branches of dist.py on the one GPU of the box).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def global_batch(n, size=41, seed=7):
    rng = np.random.default_rng(seed)
    hd = rng.uniform(-1, 1, (n, size, size, 3)).astype(np.float32)
    sd = np.clip(hd + 0.1 * rng.normal(size=hd.shape), -1, 1).astype(np.float32)
    return sd, hd


def run(out_dir, use_adam, steps, layers, n_global, lr):
    from ml_super_resolution_amd import dist as srx_dist
    from ml_super_resolution_amd.vdsr import model_vdsr
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    grouped = world > 1 or os.environ.get('SRX_TEST_GROUP_OF_ONE') == '1'
    if grouped:
        backend = srx_dist.init_process_group(rank, world, local_rank)
        if os.environ.get('SRX_TEST_EXPECT_BACKEND'):
            assert backend == torch.distributed.get_backend() == os.environ['SRX_TEST_EXPECT_BACKEND'], backend
    # every rank but 0 starts from different weights and a different step count: attach() must fix both
    model = model_vdsr.VdsrModel(num_layers=layers, use_adam=use_adam, device=dev, seed=11 + 100 * rank)
    model.stack.global_step = 5 * rank
    if grouped:
        srx_dist.attach(model.stack, world, timed=True)
    sd, hd = global_batch(n_global)
    sd, hd = torch.from_numpy(sd).to(dev), torch.from_numpy(hd).to(dev)
    if world > 1:
        sd, hd = srx_dist.shard(sd, rank, world).contiguous(), srx_dist.shard(hd, rank, world).contiguous()
    losses = []
    first_grad = None
    for s in range(steps):
        loss = model.train_step(sd, hd, lr)
        losses.append(float(loss.item()))
        if s == 0:
            first_grad = model.stack.grads.detach().cpu().numpy().copy()
    torch.cuda.synchronize()
    out = {'params': model.stack.params.detach().cpu().numpy(), 'first_grad': first_grad,
           'opt_m': model.stack.opt_m.detach().cpu().numpy(), 'losses': np.asarray(losses),
           'global_step': np.int64(model.stack.global_step)}
    if model.stack.opt_v is not None:
        out['opt_v'] = model.stack.opt_v.detach().cpu().numpy()
    if grouped:
        out['allreduce_ms'] = np.float64(srx_dist.allreduce_ms(model.stack))
        out['n_hook_calls'] = np.int64(len(model.stack.allreduce_events))
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), **out)
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        assert not torch.distributed.is_initialized()


def run_enet(out_dir, n_global, resumed=False):
    """EnhanceNet-PAT (BASELINE config 5's data parallelism, SURVEY 8e: two flat buffers, `g_` and `d_`, one all-reduce
    per trainer run): one discriminator run and one generator run on this rank's shard."""
    from ml_super_resolution_amd import dist as srx_dist
    from ml_super_resolution_amd.enet import model_enet, model_vgg
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    grouped = world > 1 or os.environ.get('SRX_TEST_GROUP_OF_ONE') == '1'
    if grouped:
        backend = srx_dist.init_process_group(rank, world, local_rank)
        if os.environ.get('SRX_TEST_EXPECT_BACKEND'):
            assert backend == torch.distributed.get_backend() == os.environ['SRX_TEST_EXPECT_BACKEND'], backend
    # narrow VGG-shaped features, the reference's discriminator widths on 64x64 images; rank 1 starts from other weights
    m = model_enet.EnetModel('pat', model_vgg.random_vgg_weights(3, 8), device=dev, seed=50 + 100 * rank, d_width=32,
                             image_size=64, dense_units=32)
    if resumed:
        # rank 0 (and the single process) come out of a checkpoint: step 6, both optimizers' slots and step counts;
        # every other rank found no checkpoint and sits at another step with no slots -- attach_flat must level them
        G, P = m.generator, m.discriminator.pool
        if rank == 0:
            gen = torch.Generator().manual_seed(77)
            m.global_step = 6
            m.g_state.update({'t': 6, 'm': (1e-3 * torch.randn(G.params.shape, generator=gen)).to(dev),
                              'v': (1e-6 * torch.rand(G.params.shape, generator=gen)).to(dev)})
            P.t = 2
            P.opt_m = (1e-3 * torch.randn(P.params.shape, generator=gen)).to(dev)
            P.opt_v = (1e-6 * torch.rand(P.params.shape, generator=gen)).to(dev)
        else:
            m.global_step = 5
    if grouped:
        srx_dist.attach_flat(m, world, timed=True)
    rng = np.random.default_rng(9)
    hd = rng.uniform(-1, 1, (n_global, 64, 64, 3)).astype(np.float32)
    sd = hd.reshape(n_global, 16, 4, 16, 4, 3).mean(axis=(2, 4)).astype(np.float32)
    bq = np.repeat(np.repeat(sd, 4, axis=1), 4, axis=2)
    t = [torch.from_numpy(a).to(dev) for a in (sd, bq, hd)]
    if world > 1:
        t = [srx_dist.shard(a, rank, world).contiguous() for a in t]
    a_loss = float(m.d_step(*t).item())
    d_grad = m.discriminator.pool.grads.detach().cpu().numpy().copy()
    losses = {k: float(v.item()) for k, v in m.g_step(*t).items()}
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), g_params=m.generator.params.detach().cpu().numpy(),
             d_params=m.discriminator.pool.params.detach().cpu().numpy(), g_grad=m.generator.grads.detach().cpu().numpy(),
             d_grad=d_grad, a_loss=np.float64(a_loss), g_loss_all=np.float64(losses['g_loss_all']),
             global_step=np.int64(m.global_step), g_t=np.int64(m.g_state['t']), d_t=np.int64(m.discriminator.pool.t),
             g_m=m.g_state['m'].detach().cpu().numpy(), g_v=m.g_state['v'].detach().cpu().numpy(),
             d_m=m.discriminator.pool.opt_m.detach().cpu().numpy(), d_v=m.discriminator.pool.opt_v.detach().cpu().numpy(),
             n_hook_calls=np.int64(len(getattr(m, 'allreduce_events', []))),
             allreduce_ms=np.float64(srx_dist.allreduce_ms(m) or 0.0) if grouped else np.float64(0.0))
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        assert not torch.distributed.is_initialized()


if __name__ == '__main__':
    if sys.argv[2] in ('enet', 'enet_resumed'):
        run_enet(sys.argv[1], int(sys.argv[3]), resumed=sys.argv[2] == 'enet_resumed')
    else:
        run(sys.argv[1], sys.argv[2] == '1', int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]))
