"""
dist.py -- data parallelism for the training path (the reference is single-GPU).

One process per GPU (torchrun-style env), `torch.distributed` with backend "nccl" (= RCCL on
ROCm, over xGMI inside a node).  Patches are independent, so the batch is sharded across ranks
and the ONLY exchange is one all-reduce(AVG) of the flat gradient buffer per step (668,227
floats = 2.67 MB for VDSR-20): every rank computes the gradient of its LOCAL mean loss + the
(identical) regulariser gradient, and the average over equal shards equals the single-process
gradient of the concatenated batch.  The momentum path clips AFTER the reduce, as required.
On CPU (tests) the same code runs over gloo.
"""
import os

import torch
import torch.distributed as td


def init_process_group(rank=None, world_size=None, local_rank=None, backend=None):
    rank = int(os.environ.get('RANK', 0)) if rank is None else rank
    world_size = int(os.environ.get('WORLD_SIZE', 1)) if world_size is None else world_size
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        # SRX_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a 1-GPU box); RCCL needs one GPU per rank
        backend = os.environ.get('SRX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if not td.is_initialized():
        td.init_process_group(backend=backend, rank=rank, world_size=world_size)
    return backend


def allreduce_mean_(flat, world_size):
    """In-place average of a flat gradient buffer over all ranks: one collective."""
    if td.get_backend() == 'nccl':
        td.all_reduce(flat, op=td.ReduceOp.AVG)
    elif flat.is_cuda:                      # gloo rehearsal with device tensors: reduce through the host
        host = flat.detach().cpu()
        td.all_reduce(host, op=td.ReduceOp.SUM)
        flat.copy_(host.div_(world_size))
    else:                                   # gloo has no AVG
        td.all_reduce(flat, op=td.ReduceOp.SUM)
        flat.div_(world_size)
    return flat


def attach(stack, world_size):
    """Install the gradient all-reduce on a ConvStack and make the replicas start identical."""
    if td.get_backend() != 'nccl' and stack.params.is_cuda:
        host = stack.params.detach().cpu()
        td.broadcast(host, src=0)
        stack.params.copy_(host)
    else:
        td.broadcast(stack.params, src=0)
    stack.grad_hook = lambda g: allreduce_mean_(g, world_size)
    return stack


def shard(batch, rank, world_size):
    """Rank's contiguous shard of a global batch (equal shards required for AVG == global mean)."""
    n = batch.shape[0]
    if n % world_size:
        raise ValueError('global batch %d not divisible by world size %d' % (n, world_size))
    per = n // world_size
    return batch[rank * per:(rank + 1) * per]
