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


def broadcast_(t, src=0):
    """In-place broadcast of a tensor from rank `src` (through the host for device tensors on gloo)."""
    if td.get_backend() != 'nccl' and t.is_cuda:
        host = t.detach().cpu()
        td.broadcast(host, src=src)
        t.copy_(host)
    else:
        td.broadcast(t, src=src)
    return t


def attach(stack, world_size, timed=False):
    """Install the gradient all-reduce on a ConvStack and make the replicas start identical: parameters,
    `global_step` and the optimizer slots all come from rank 0 (a rank that resumed from a different
    checkpoint -- or none -- would otherwise run a different number of steps and hang the collective).
    timed=True: every hook call is bracketed by two events on the launch stream, kept in
    `stack.allreduce_events`, so the caller can report where a scaling loss goes (bench.py `allreduce_ms`)."""
    broadcast_(stack.params, 0)
    if hasattr(stack, 'global_step'):
        # [global_step, has opt_m, has opt_v] of rank 0
        head = torch.tensor([stack.global_step, int(stack.opt_m is not None), int(stack.opt_v is not None)],
                            dtype=torch.int64)
        if td.get_backend() == 'nccl':
            head = head.to(stack.params.device)
        td.broadcast(head, src=0)
        step, has_m, has_v = (int(v) for v in head.tolist())
        stack.global_step = step
        for flag, name in ((has_m, 'opt_m'), (has_v, 'opt_v')):
            if flag:
                if getattr(stack, name) is None:
                    setattr(stack, name, torch.zeros_like(stack.params))
                broadcast_(getattr(stack, name), 0)
            else:
                setattr(stack, name, None)
    stack.allreduce_events = []

    def hook(g):
        if timed and g.is_cuda:
            s = torch.cuda.current_stream(g.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            allreduce_mean_(g, world_size)
            e1.record(s)
            stack.allreduce_events.append((e0, e1))
        else:
            allreduce_mean_(g, world_size)
        return g

    stack.grad_hook = hook
    return stack


def allreduce_ms(stack, last=None):
    """Mean duration (ms) of the timed hook calls recorded by attach(timed=True); call after a synchronize."""
    ev = stack.allreduce_events[-last:] if last else stack.allreduce_events
    if not ev:
        return None
    return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


def shard(batch, rank, world_size):
    """Rank's contiguous shard of a global batch (equal shards required for AVG == global mean)."""
    n = batch.shape[0]
    if n % world_size:
        raise ValueError('global batch %d not divisible by world size %d' % (n, world_size))
    per = n // world_size
    return batch[rank * per:(rank + 1) * per]


def _timed_hook(owner, world_size, timed):
    """The gradient exchange as a hook; timed=True brackets every call with two events on the launch stream, kept in
    `owner.allreduce_events` (read by allreduce_ms)."""
    owner.allreduce_events = getattr(owner, 'allreduce_events', [])

    def hook(g):
        if timed and g.is_cuda:
            s = torch.cuda.current_stream(g.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            allreduce_mean_(g, world_size)
            e1.record(s)
            owner.allreduce_events.append((e0, e1))
        else:
            allreduce_mean_(g, world_size)
        return g
    return hook


def attach_flat(model, world_size, timed=False):
    """EnhanceNet (BASELINE config 5): two parameter groups, `g_` and `d_` (enet/enet/model_enet.py:331-343), each
    one flat buffer -> one all-reduce(AVG) per trainer run.  As attach() does for a ConvStack, EVERYTHING the schedule
    and the optimizers depend on comes from rank 0: both parameter buffers, `global_step` (it decides which steps run
    d_trainer -- experiment_train.py:112 -- and when to save / stop: ranks that disagree issue different collectives
    and hang), both Adam step counts and both pairs of Adam slots."""
    G = model.generator
    D = getattr(model, 'discriminator', None)
    P = D.pool if D is not None else None
    broadcast_(G.params, 0)
    if P is not None:
        broadcast_(P.params, 0)
    gs = model.g_state
    head = torch.tensor([int(model.global_step), int(bool(gs)), int(gs.get('t', 0)) if gs else 0,
                         int(P is not None and P.opt_m is not None), int(P.t) if P is not None else 0], dtype=torch.int64)
    if td.get_backend() == 'nccl':
        head = head.to(G.params.device)
    td.broadcast(head, src=0)
    step, has_g, g_t, has_d, d_t = (int(v) for v in head.tolist())
    model.global_step = step
    if has_g:
        if not gs:
            gs.update({'m': torch.zeros_like(G.params), 'v': torch.zeros_like(G.params)})
        gs['t'] = g_t
        broadcast_(gs['m'], 0)
        broadcast_(gs['v'], 0)
    else:
        gs.clear()
    if P is not None:
        P.t = d_t
        if has_d:
            if P.opt_m is None:
                P.opt_m, P.opt_v = torch.zeros_like(P.params), torch.zeros_like(P.params)
            broadcast_(P.opt_m, 0)
            broadcast_(P.opt_v, 0)
        else:
            P.opt_m = P.opt_v = None
    model.allreduce_events = []
    model.grad_hook_g = _timed_hook(model, world_size, timed)
    model.grad_hook_d = _timed_hook(model, world_size, timed)
    return model
