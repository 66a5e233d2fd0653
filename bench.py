#!/usr/bin/env python3
"""
bench.py -- BASELINE.json metric: "HR megapixels/sec/GPU (fwd) + patches/sec (train),
VDSR-20 4x @41x41" on configs[2]: VDSR-20 (3x3x64, residual), batch 256 x 41x41 per GPU,
synthetic patches resident in HBM, random-init (Xavier) weights, fp32.

A "step" is one full training step over one batch: forward (20 convs), MSE + L2 loss,
backward (19 dgrads + 20 wgrads), gradient all-reduce over RCCL when N > 1, TF-Adam update.
`value` = patches/s over all ranks.  The forward-only rate (HR megapixels/s) is reported in
`fwd_hr_mpix_per_s`.  Data parallel: one process per GPU, weak scaling (256 patches per GPU),
ONE all-reduce(AVG) of the flat 2.67 MB gradient per step.

  python bench.py [--gpus N --steps K --warmup W]        (N > 1 under torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH, SIZE = 256, 41
FWD_FLOP_PER_PX = 1334016.0            # 667,008 MAC/px
TRAIN_FLOP_PER_PX = 3998592.0          # fwd + wgrad (all layers) + dgrad (layers 2..20)
MID_LAYER_FLOP_PER_PX = 73728.0        # 3x3x64x64 MACs * 2
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def hip_event_time_ms(fn, iters, stream):
    """Average duration of fn() in ms measured with HIP events on `stream` (torch.cuda.Event
    records on torch's current stream, which is the stream the library launches on)."""
    with torch.cuda.stream(stream):
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record(stream)
        for _ in range(iters):
            fn()
        end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) / iters


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows all of
    the host's logical CPUs but grants a share of them; one thread per visible CPU would only oversubscribe it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(rank):
    """The oracle's C restatement ("port") timed on this host's cores on a bounded sample of the
    same workload: forward + backward of VDSR-20 on a few 41x41 patches."""
    if rank != 0:
        return None
    from oracle import oracle as O
    import subprocess
    so = os.path.join(ROOT, 'oracle', 'libsrx_oracle.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(106)
    params = [(O.xavier_uniform(rng, ks), np.zeros(bs, np.float32)) for ks, bs in O.vdsr_param_shapes(20)]
    # one thread per CPU the process is granted (torch has initialised OpenMP with one per visible CPU already)
    cores = O.clib().srx_ref_set_num_threads(usable_cpus())
    n = max(4, cores)                      # a few patches per core
    hd = np.random.default_rng(104).uniform(-1, 1, (n, SIZE, SIZE, 3)).astype(np.float32)
    sd = np.clip(hd + 0.1 * np.random.default_rng(105).normal(0, 1, hd.shape), -1, 1).astype(np.float32)
    O.c_vdsr_train_step_grads(sd[:1], hd[:1], params)        # warm the library
    t0 = time.perf_counter()
    reps = 0
    while True:
        O.c_vdsr_train_step_grads(sd, hd, params)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 8:
            break
    return {'value': round(n * reps / dt, 3), 'unit': 'patches/s', 'cores': int(cores), 'kind': 'port',
            'sample': '%d x VDSR-20 fwd+bwd on %d patches of 41x41 (oracle/srx_oracle.c, OpenMP, fp32)' % (reps, n)}


def cpu_library_baseline(rank):
    """Second CPU comparator: the same VDSR-20 forward + backward through torch's CPU convolutions
    (oneDNN, all host cores) -- the optimised-library stand-in closest to what the reference's TensorFlow-Eigen CPU
    path would do.  Neither the reference nor a port of it: reported beside `cpu_baseline`, never instead of it."""
    if rank != 0:
        return None
    try:
        import torch.nn.functional as F
        threads = usable_cpus()
        torch.set_num_threads(threads)
        g = torch.Generator().manual_seed(106)
        ws = [torch.randn((64 if i < 19 else 3, 3 if i == 0 else 64, 3, 3), generator=g) * 0.05 for i in range(20)]
        bs = [torch.zeros(w.shape[0]) for w in ws]
        for t in ws + bs:
            t.requires_grad_(True)
        n = max(32, min(256, threads))
        hd = torch.rand((n, 3, SIZE, SIZE), generator=g) * 2 - 1
        sd = (hd + 0.1 * torch.randn(hd.shape, generator=g)).clamp(-1, 1)

        def step():
            t = sd
            for i in range(19):
                t = F.relu(F.conv2d(t, ws[i], bs[i], padding=1))
            sr = sd + F.conv2d(t, ws[19], bs[19], padding=1)
            loss = F.mse_loss(sr, hd)
            loss.backward()
            for t_ in ws + bs:
                t_.grad = None

        step()
        t0 = time.perf_counter()
        reps = 0
        while True:
            step()
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 8.0 or reps >= 6:
                break
        return {'value': round(n * reps / dt, 2), 'unit': 'patches/s', 'cores': int(threads), 'kind': 'library stand-in',
                'sample': '%d x VDSR-20 fwd+bwd on %d patches of 41x41 (torch %s CPU conv2d + autograd, fp32)' % (reps, n, torch.__version__)}
    except Exception as exc:     # a comparator, not part of the measurement
        return {'value': None, 'error': repr(exc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                         '--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...' % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev            # (rehearsals may put several gloo ranks on one GPU)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)

    from ml_super_resolution_amd.vdsr import model_vdsr
    from ml_super_resolution_amd import dist as srx_dist

    if world > 1:
        srx_dist.init_process_group(rank, world, local_rank)

    model = model_vdsr.VdsrModel(num_layers=20, use_adam=True, device=dev, seed=106)
    if world > 1:
        srx_dist.attach(model.stack, world)

    # synthetic patches: rank r uses seeds 104+10r / 105+10r
    g = torch.Generator(device=dev).manual_seed(104 + 10 * rank)
    hd = torch.rand((BATCH, SIZE, SIZE, 3), device=dev, generator=g) * 2 - 1
    g2 = torch.Generator(device=dev).manual_seed(105 + 10 * rank)
    sd = (hd + 0.1 * torch.randn((BATCH, SIZE, SIZE, 3), device=dev, generator=g2)).clamp(-1, 1)
    lr = 5e-5                                                        # vdsr/makefile:27

    def step():
        model.train_step(sd, hd, lr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        nccl = torch.distributed.get_backend() == 'nccl'
        t = torch.tensor([dt], dtype=torch.float64, device=dev if nccl else 'cpu')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    patches_per_s = world * BATCH * args.steps / dt

    # ---- forward-only rate and the dominant kernel's roofline (rank 0, HIP events on the launch stream)
    stream = torch.cuda.current_stream()
    px = BATCH * SIZE * SIZE
    fwd_ms = hip_event_time_ms(lambda: model.forward(sd), 10, stream)
    from ml_super_resolution_amd import ops
    x64 = torch.rand((BATCH, SIZE, SIZE, 64), device=dev) * 2 - 1
    y64 = torch.empty_like(x64)
    k, b = model.stack.kernel(5), model.stack.bias(5)
    ops.conv2d_fwd(x64, k, b, 'same', 'relu', out=y64)
    mid_ms = hip_event_time_ms(lambda: ops.conv2d_fwd(x64, k, b, 'same', 'relu', out=y64), 20, stream)
    achieved_tf = MID_LAYER_FLOP_PER_PX * px / (mid_ms * 1e-3) / 1e12
    train_tf = TRAIN_FLOP_PER_PX * px / (ms_per_step * 1e-3) / 1e12

    if rank == 0:
        line = {
            'metric': 'VDSR-20 training patches/sec (41x41, batch 256 per GPU); fwd HR megapixels/sec alongside',
            'value': round(patches_per_s, 1), 'unit': 'patches/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[2]: VDSR 20-layer 3x3x64 residual, RGB, batch 256x41x41 '
                                   'per GPU, fwd+bwd+TF-Adam, random-init weights',
                       'global_batch': world * BATCH, 'patch': SIZE, 'parallelism': 'dp%d' % world},
            'fwd_hr_mpix_per_s': round(px / (fwd_ms * 1e-3) / 1e6, 2),
            'fwd_ms': round(fwd_ms, 3),
            'train_step_tflops': round(train_tf, 2),
            'train_step_frac_of_fp32_mfma_peak': round(train_tf / PEAK_FP32_MFMA_TFLOPS, 4),
            'roofline': {'bound': 'mfma', 'kernel': 'conv_pipe_kernel<3,3,64,4,fwd> (3x3 64->64 fwd+bias+ReLU)',
                         'achieved': round(achieved_tf, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                         # HBM bytes per launch from the PMC passes committed under profiles/ (2*FETCH_SIZE +
                         # WRITE_SIZE, gfx950 correction); algorithmic bytes are 220.3e6 (input + output once)
                         'traffic': 2.225e8, 'traffic_source': 'profiles/r01_prof_conv_hbm_counters.csv',
                         'launch_ms': round(mid_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(rank)
            line['cpu_library_baseline'] = cpu_library_baseline(rank)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
