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

  python bench.py [--gpus N --steps K --warmup W]

N > 1: either started by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or started bare -- then this process makes no
GPU call at all and starts its own N ranks as fresh child processes (never an exec), relays rank 0's JSON line
and exits with the children's worst return code.  With fewer GPUs than ranks (a rehearsal on a one-GPU box) the
ranks share the visible GPUs and exchange gradients over gloo; the line then says so in `config.rehearsal`.

Extra keys beside the primary metric (rank 0, after the timed region): the other north-star numbers --
`subpixel` (depth-to-space GB/s at [256,41,41,27], rotating buffers), `espcn_c2_us` (BASELINE configs[1]),
`srcnn_c1_us` (configs[0] shape), `dgrad` / `wgrad` fractions of the fp32-MFMA peak.
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
        # best of at least three repetitions (the host is shared: between two runs of this leg the AVERAGE moved 137 <-> 523
        # patches/s; the fastest repetition is the one the other tenants disturbed least), thread count pinned above
        t0 = time.perf_counter()
        times = []
        while True:
            t1 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t1)
            if len(times) >= 3 and (time.perf_counter() - t0 > 8.0 or len(times) >= 8):
                break
        best = min(times)
        return {'value': round(n / best, 2), 'unit': 'patches/s', 'cores': int(threads), 'kind': 'library stand-in',
                'statistic': 'best of %d repetitions (mean %.2f patches/s)' % (len(times), n * len(times) / sum(times)),
                'sample': '%d x VDSR-20 fwd+bwd on %d patches of 41x41 (torch %s CPU conv2d + autograd, fp32)' % (len(times), n, torch.__version__)}
    except Exception as exc:     # a comparator, not part of the measurement
        return {'value': None, 'error': repr(exc)}


def cpu_small_configs(rank):
    """BASELINE configs[0] ("SRCNN 9-1-5 on one 256x256 image, CPU reference path") and configs[1] (ESPCN 3x inference, batch
    32 of 17x17 LR patches) timed on this host's cores, as BASELINE.md section 4 promises: the oracle's C restatement
    (`port`, OpenMP) and torch's CPU convolutions (oneDNN: the optimised-library stand-in for the TensorFlow-Eigen path
    that cannot run here).  configs[0]'s network input is the 243x243 RGB crop the reference's VALID geometry implies
    (srcnn/srcnn.py:14-16,28-40: 256 -> 243 -> 231)."""
    if rank != 0:
        return {}
    out = {}
    try:
        from oracle import oracle as O
        import torch.nn.functional as F
        cores = int(O.clib().srx_ref_set_num_threads(usable_cpus()))
        torch.set_num_threads(usable_cpus())
        rng = np.random.default_rng(101)

        def best_ms(fn, budget_s=3.0, min_reps=3, max_reps=40):
            fn()
            times, t0 = [], time.perf_counter()
            while len(times) < min_reps or (time.perf_counter() - t0 < budget_s and len(times) < max_reps):
                t1 = time.perf_counter()
                fn()
                times.append(time.perf_counter() - t1)
            return round(min(times) * 1e3, 3), len(times)

        # --- configs[0]: relu(conv9x9 3->64) -> relu(conv1x1 64->32) -> tanh(conv5x5 32->3), VALID
        sp = [(rng.normal(0, 0.05, (9, 9, 3, 64)).astype(np.float32), np.zeros(64, np.float32)),
              (rng.normal(0, 0.05, (1, 1, 64, 32)).astype(np.float32), np.zeros(32, np.float32)),
              (rng.normal(0, 0.05, (5, 5, 32, 3)).astype(np.float32), np.zeros(3, np.float32))]
        img = rng.uniform(-1, 1, (1, 243, 243, 3)).astype(np.float32)

        def srcnn_port():
            t = O.c_conv2d_fwd(img, sp[0][0], sp[0][1], 'VALID', 'relu')
            t = O.c_conv2d_fwd(t, sp[1][0], sp[1][1], 'VALID', 'relu')
            return O.c_conv2d_fwd(t, sp[2][0], sp[2][1], 'VALID', 'tanh')
        tw = [(torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), torch.from_numpy(b)) for w, b in sp]
        timg = torch.from_numpy(np.ascontiguousarray(img.transpose(0, 3, 1, 2)))

        def srcnn_lib():
            with torch.no_grad():
                t = F.relu(F.conv2d(timg, tw[0][0], tw[0][1]))
                t = F.relu(F.conv2d(t, tw[1][0], tw[1][1]))
                return torch.tanh(F.conv2d(t, tw[2][0], tw[2][1]))
        pm, pr = best_ms(srcnn_port)
        lm, lr_ = best_ms(srcnn_lib)
        out['srcnn_c1_cpu_ms'] = {'port_ms': pm, 'library_ms': lm, 'cores': cores, 'statistic': 'best of %d / %d repetitions' % (pr, lr_),
                                  'what': 'SRCNN 9-1-5 VALID forward on one 243x243 RGB image -> 231x231 (BASELINE configs[0]): '
                                          'oracle/srx_oracle.c (OpenMP) / torch %s CPU conv2d (oneDNN)' % torch.__version__}
        # --- configs[1]: tanh(conv5x5 3->64) -> tanh(conv3x3 64->32) -> conv3x3 32->27, SAME, then depth-to-space r = 3
        ep = [(rng.normal(0, 0.05, (5, 5, 3, 64)).astype(np.float32), np.zeros(64, np.float32)),
              (rng.normal(0, 0.05, (3, 3, 64, 32)).astype(np.float32), np.zeros(32, np.float32)),
              (rng.normal(0, 0.05, (3, 3, 32, 27)).astype(np.float32), np.zeros(27, np.float32))]
        lrp = rng.uniform(-1, 1, (32, 17, 17, 3)).astype(np.float32)

        def espcn_port():
            t = O.c_conv2d_fwd(lrp, ep[0][0], ep[0][1], 'SAME', 'tanh')
            t = O.c_conv2d_fwd(t, ep[1][0], ep[1][1], 'SAME', 'tanh')
            return O.c_depth_to_space(O.c_conv2d_fwd(t, ep[2][0], ep[2][1], 'SAME', None), 3)
        te = [(torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), torch.from_numpy(b)) for w, b in ep]
        # (TensorFlow's channel order (dy, dx, c) is not pixel_shuffle's (c, dy, dx): the filters are permuted once, outside the timing)
        perm = torch.arange(27).view(3, 3, 3).permute(2, 0, 1).reshape(-1)
        w3, b3 = te[2][0][perm].contiguous(), te[2][1][perm].contiguous()
        tlr = torch.from_numpy(np.ascontiguousarray(lrp.transpose(0, 3, 1, 2)))

        def espcn_lib():
            with torch.no_grad():
                t = torch.tanh(F.conv2d(tlr, te[0][0], te[0][1], padding=2))
                t = torch.tanh(F.conv2d(t, te[1][0], te[1][1], padding=1))
                return F.pixel_shuffle(F.conv2d(t, w3, b3, padding=1), 3)
        pm, pr = best_ms(espcn_port)
        lm, lr_ = best_ms(espcn_lib)
        out['espcn_c2_cpu_ms'] = {'port_ms': pm, 'library_ms': lm, 'cores': cores, 'statistic': 'best of %d / %d repetitions' % (pr, lr_),
                                  'what': 'ESPCN 3x forward + depth-to-space on 32 LR patches of 17x17 (BASELINE configs[1]): '
                                          'oracle/srx_oracle.c (OpenMP) / torch %s CPU conv2d + pixel_shuffle (oneDNN)' % torch.__version__}
    except Exception as exc:     # comparators, not part of the measurement
        out['small_configs_cpu_error'] = repr(exc)
    return out


TRAFFIC_SOURCES = ('conv_kernels.hip.h', 'launchers.h', 'pipe_inst_k3c64.hip', 'srx_api.hip')


def kernel_source_sha():
    """sha256 over the sources that define the dominant kernel and its launch plan (the kernel header, its instance
    file, the planner): ties a number measured on an earlier build to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'ml_super_resolution_amd', 'csrc')
    for name in TRAFFIC_SOURCES:
        h.update(open(os.path.join(csrc, name), 'rb').read())
    return h.hexdigest()[:16]


def measured_traffic():
    """Fallback of live_traffic() (rocprofv3 missing, this process itself under a profiler, N > 1, --no-live-traffic):
    `roofline.traffic` is then NOT measured by this run: it is the
    HBM bytes per launch of the dominant kernel from the committed PMC passes (2*FETCH_SIZE + WRITE_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md), recorded in profiles/traffic.json together with the kernel name and
    the sha of the kernel sources it was measured on.  If the sources have changed since, the value is withheld
    (null) rather than silently carried over."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except (OSError, ValueError):
        return {'traffic': None, 'traffic_source': None}
    sha = kernel_source_sha()
    if rec.get('csrc_sha') != sha:
        return {'traffic': None, 'traffic_source': rec.get('source'), 'traffic_stale': 'measured on csrc %s, this build is %s'
                % (rec.get('csrc_sha'), sha)}
    return {'traffic': rec['traffic_bytes'], 'traffic_algorithmic': rec.get('algorithmic_bytes'),
            'traffic_source': rec.get('source'), 'traffic_kernel': rec.get('kernel'), 'traffic_csrc_sha': sha}


TRAFFIC_KERNEL = 'void srx::conv_pipe_kernel<3, 3, 64, 4, false, 0>(srx::ConvArgs)'


def under_profiler():
    """True when this process itself runs under rocprofv3 (its tool library is preloaded): no second profiler inside."""
    env = os.environ
    return any('rocprof' in env.get(k, '').lower() for k in ('LD_PRELOAD', 'HSA_TOOLS_LIB', 'ROCP_TOOL_LIBRARIES')) or \
        any(k.startswith(('ROCPROF', 'ROCPROFILER_')) for k in env)


def live_traffic(timeout_s=120):
    """`roofline.traffic` measured IN THIS RUN: two child processes `rocprofv3 --kernel-trace --pmc <counter> -- python3
    scripts/prof_conv.py 10 fwd` (FETCH_SIZE and WRITE_SIZE need a pass each; nothing but --kernel-trace beside --pmc),
    started BEFORE this process touches the GPU (a process that has initialised the GPU must not start programs).
    HBM bytes per launch of the dominant kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the counters are in KB and gfx950
    counts a wide streamed read at half its size (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Returns the roofline keys,
    or None (then the caller falls back to the committed profiles/traffic.json and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return None
    vals, launches = {}, 0
    t0 = time.time()
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='srx_pmc_')
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            r = subprocess.run([prof, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', d, '--', 'python3',
                                os.path.join(ROOT, 'scripts', 'prof_conv.py'), '10', 'fwd'], cwd='/tmp', env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return None
            v = [float(row['Counter_Value']) for row in csv.DictReader(open(files[0]))
                 if row['Kernel_Name'] == TRAFFIC_KERNEL and row['Counter_Name'] == ctr]
            if not v:
                return None
            vals[ctr] = sum(v) / len(v)
            launches = len(v)
        except (OSError, subprocess.SubprocessError, ValueError, KeyError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {'traffic': round((2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0),
            'traffic_algorithmic': 2 * 256 * 41 * 41 * 64 * 4,
            'traffic_source': 'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over '
                              'scripts/prof_conv.py 10 fwd, average of %d launches; (2*FETCH_SIZE + WRITE_SIZE) KB' % launches,
            'traffic_kernel': TRAFFIC_KERNEL, 'traffic_fetch_size_kb': round(vals['FETCH_SIZE'], 1),
            'traffic_write_size_kb': round(vals['WRITE_SIZE'], 1), 'traffic_seconds': round(time.time() - t0, 1)}


def extras(model, dev, stream, x64, y64, px):
    """The other north-star numbers, measured after the timed region with HIP events on the launch stream."""
    from ml_super_resolution_amd import ops
    from ml_super_resolution_amd.espcn import model_espcn
    from ml_super_resolution_amd.srcnn import srcnn as srcnn_mod
    out = {}
    # -- dgrad / wgrad of the 3x3 64->64 layer (same algorithmic FLOPs as the forward launch)
    k = model.stack.kernel(5)
    dx = torch.empty_like(x64)
    dw, db = torch.empty_like(k), torch.empty(64, device=dev)
    ws = torch.empty((ops.bwd_filter_workspace_bytes(x64.shape, k.shape) + 3) // 4, device=dev)
    f_dgrad = lambda: ops.conv2d_bwd_data(y64, k, x64.shape, 'same', x_in=x64, in_act='relu', out=dx)
    f_wgrad = lambda: ops.conv2d_bwd_filter(x64, y64, k.shape, 'same', w_for_decay=k, wd_scale=1e-4, dw=dw, dbias=db,
                                            workspace=ws)
    for name, fn in (('dgrad', f_dgrad), ('wgrad', f_wgrad)):
        fn()
        ms = hip_event_time_ms(fn, 20, stream)
        tf = MID_LAYER_FLOP_PER_PX * px / (ms * 1e-3) / 1e12
        out[name] = {'launch_ms': round(ms, 4), 'achieved': round(tf, 2), 'unit': 'TFLOP/s',
                     'frac': round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                     'what': '3x3 64->64 %s at 256x41x41%s' % (name, ' incl. the reduction of the partial filters and the '
                                                             'regulariser term' if name == 'wgrad' else ' + fused ReluGrad')}
    del dx, ws
    out['vdsr_recipe_64x128'] = vdsr_recipe(dev, stream)
    # -- sub-pixel map at the north-star bandwidth shape, 8 rotating buffer pairs (744 MB > Infinity Cache)
    pairs = 8
    ins = [torch.rand((256, 41, 41, 27), device=dev) for _ in range(pairs)]
    outs = [torch.empty((256, 123, 123, 3), device=dev) for _ in range(pairs)]
    state = {'i': 0}

    def d2s():
        i = state['i'] = (state['i'] + 1) % pairs
        ops.depth_to_space(ins[i], 3, out=outs[i])
    for _ in range(pairs):
        d2s()
    ms = min(hip_event_time_ms(d2s, 10 * pairs, stream) for _ in range(3))
    nbytes = 2.0 * ins[0].numel() * 4
    gbps = nbytes / (ms * 1e-3) / 1e9
    # the ceiling of a kernel that only moves these bytes: the library's hand-written streaming copy (nontemporal 16-byte
    # loads / stores, every load of the tensor in flight at once) over the SAME rotating pairs
    flat_out = [o.view(-1) for o in outs]

    def copy():
        i = state['i'] = (state['i'] + 1) % pairs
        ops.stream_copy(ins[i].view(-1), flat_out[i])
    for _ in range(pairs):
        copy()
    cms = min(hip_event_time_ms(copy, 10 * pairs, stream) for _ in range(3))
    cgbps = nbytes / (cms * 1e-3) / 1e9
    out['subpixel'] = {'bound': 'hbm', 'kernel': 'subpixel_even_kernel (standalone depth-to-space [256,41,41,27] -> [256,123,123,3])',
                       'launch_us': round(ms * 1e3, 2), 'bytes': nbytes, 'achieved': round(gbps, 1), 'peak': 8000.0,
                       'unit': 'GB/s', 'frac': round(gbps / 8000.0, 4), 'rotating_pairs': pairs,
                       'copy_ceiling_gbps': round(cgbps, 1), 'copy_ceiling_us': round(cms * 1e3, 2),
                       'copy_ceiling_frac': round(cgbps / 8000.0, 4), 'frac_of_copy_ceiling': round(gbps / cgbps, 4),
                       'copy_ceiling_what': 'srx_stream_copy of the same 92.95 MB over the same 8 rotating pairs'}
    del ins, outs
    # -- BASELINE configs[1]: ESPCN 3x inference, batch 32 of 17x17 LR patches: forward + depth-to-space
    e3 = model_espcn.EspcnModel(3, device=dev, seed=103)
    lr = torch.rand((32, 17, 17, 3), device=dev) * 2 - 1
    sr_fn = lambda: e3.super_resolve(lr)
    for _ in range(5):
        sr_fn()
    us = hip_event_time_ms(sr_fn, 200, stream) * 1e3
    out['espcn_c2_us'] = round(us, 2)
    out['espcn_c2_hr_mpix_per_s'] = round(32 * 51 * 51 / us, 1)
    out['espcn_c2_tflops'] = round(573.5e6 / (us * 1e-6) / 1e12, 2)
    out['espcn_c2_path'] = getattr(e3, 'inference_path', 'eager: 3 conv launches + depth-to-space')
    # -- BASELINE configs[0] shape as the reference would run it: SRCNN 9-1-5 VALID on one 243x243 RGB image
    sm = srcnn_mod.SrcnnModel(device=dev, seed=101)
    for i in range(3):                                   # O(1) activations instead of the reference's sigma 1e-3
        sm.stack.kernel(i).mul_(60.0)
    img = torch.rand((1, 243, 243, 3), device=dev) * 2 - 1
    c1 = lambda: sm.forward(img)
    for _ in range(5):
        c1()
    us = hip_event_time_ms(c1, 100, stream) * 1e3
    out['srcnn_c1_us'] = round(us, 2)
    out['srcnn_c1_tflops'] = round(2200e6 / (us * 1e-6) / 1e12, 2)
    c3 = lambda: sm.forward(img, single_launch=True)
    for _ in range(5):
        c3()
    out['srcnn_c1_one_launch_us'] = round(hip_event_time_ms(c3, 100, stream) * 1e3, 2)
    out['srcnn_c1_path'] = ('three launches: 9x9 3->64 on conv_pack3_kernel, 1x1 64->32, 5x5 32->3 on conv_kwrows_kernel (the default route '
                            'for single images); srcnn_c1_one_launch_us: the three layers chained through LDS per 15x15 output tile '
                            '(srx_srcnn_forward, the route for batches of small patches)')
    del img, lr
    # -- the small networks' TRAIN steps (espcn/makefile:30-36: batch 64 of 17x17 LR patches, 1.6 M steps; srcnn/srcnn.py:14-16,
    #    28-40: batch 64 of 243x243 crops): forward + loss + backward + Adam replayed as one HIP graph per batch shape
    try:
        eb = 64
        elr = torch.rand((eb, 17, 17, 3), device=dev) * 2 - 1
        ehr = torch.rand((eb, 17, 17, 27), device=dev) * 2 - 1
        default_replays = e3.stack.step_graph_max_pixels is None or eb * 17 * 17 <= e3.stack.step_graph_max_pixels
        e3.stack.step_graph_max_pixels = None              # (measure the replay whatever size the model stops replaying at)
        for _ in range(6):
            e3.train_step(elr, ehr, 1e-3)
        st_in = e3.stack.static_step_inputs(elr.shape, ehr.shape)     # batches written straight into the captured step's inputs
        if st_in is not None:
            st_in[0].copy_(elr); st_in[1].copy_(ehr)
            elr, ehr = st_in
        us_g = hip_event_time_ms(lambda: e3.train_step(elr, ehr, 1e-3), 200, stream) * 1e3
        e3.stack.use_step_graph = False
        for _ in range(3):
            e3.train_step(elr, ehr, 1e-3)
        us_e = hip_event_time_ms(lambda: e3.train_step(elr, ehr, 1e-3), 200, stream) * 1e3
        e3.use_single_launch_train = False                  # (round 3's step: the forward pass as three launches)
        for _ in range(3):
            e3.train_step(elr, ehr, 1e-3)
        us_3 = hip_event_time_ms(lambda: e3.train_step(elr, ehr, 1e-3), 200, stream) * 1e3
        best = min(us_g, us_e)
        out['espcn_train_us'] = {'batch': eb, 'patch': '17x17 LR, r = 3', 'graph_replay_us': round(us_g, 2), 'eager_launches_us': round(us_e, 2),
                                 'eager_three_launch_forward_us': round(us_3, 2),
                                 'default_route': 'graph replay' if default_replays else 'eager launches (the model replays batches of up to %s LR pixels)' % os.environ.get('SRX_ESPCN_STEP_GRAPH_MAX_PIXELS', '12000'),
                                 'speedup': round(us_e / us_g, 3), 'patches_per_s': round(eb / (best * 1e-6), 0),
                                 'what': 'ESPCN train step (forward in ONE launch that keeps t1 / t2 / y for backward -- srx_espcn_forward_keep --, MSE, '
                                         '3 wgrad + reduce, 2 dgrad, TF-Adam with device-resident step count) as ONE replayed HIP graph vs the same launches '
                                         'issued eagerly; eager_three_launch_forward_us: the same step with the forward pass as three launches'}
    except Exception as exc:
        out['espcn_train_us'] = {'error': repr(exc)}
    try:
        sb = 64
        ssd = torch.rand((sb, 243, 243, 3), device=dev) * 2 - 1
        shd = torch.rand((sb, 231, 231, 3), device=dev) * 2 - 1
        for _ in range(4):
            sm.train_step(ssd, shd)
        ms_g = hip_event_time_ms(lambda: sm.train_step(ssd, shd), 10, stream)
        # fwd MACs per output pixel: 9*9*3*64 at 235^2, 64*32 at 235^2, 5*5*32*3 at 231^2; backward: wgrad of all three + dgrad of layers 2, 3
        f1, f2, f3 = 2.0 * 81 * 3 * 64 * 235 * 235, 2.0 * 64 * 32 * 235 * 235, 2.0 * 25 * 32 * 3 * 231 * 231
        flop = sb * (2 * f1 + 3 * f2 + 3 * f3)
        out['srcnn_train_ms'] = {'batch': sb, 'crop': '243x243 -> 231x231', 'train_ms': round(ms_g, 3), 'images_per_s': round(sb / (ms_g * 1e-3), 1),
                                 'tflops': round(flop / (ms_g * 1e-3) / 1e12, 2), 'frac_of_fp32_mfma_peak': round(flop / (ms_g * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                 'what': 'SRCNN 9-1-5 train step (row-norm loss, Adam(1e-3, .5, .9)) at the reference\'s batch of 64 crops of 243x243; algorithmic FLOPs (3 input channels unpadded)'}
        del ssd, shd
    except Exception as exc:
        out['srcnn_train_ms'] = {'error': repr(exc)}
    del sm, e3
    # -- BASELINE configs[4] as the reference trains it (enet/enet/experiment_train.py:15-22): EnhanceNet-PAT, batch 64 of
    #    32x32 -> 128x128 patches, VGG-19 perceptual + texture + adversarial losses (random VGG-shaped weights: the real
    #    ones are not available offline -- timing only).  One cycle of the schedule = 1 discriminator + 3 generator runs.
    try:
        from ml_super_resolution_amd.enet import experiment_train as enet_train, model_enet, model_vgg
        nb = 64
        em = model_enet.EnetModel('pat', model_vgg.random_vgg_weights(0), device=dev, seed=1)
        sdb, bqb, hdb = next(enet_train.synthetic_batches(nb, dev))
        em.g_step(sdb, bqb, hdb); em.d_step(sdb, bqb, hdb)
        # (≈ 380 launches per cycle issued from Python: on a host loaded by other tenants the launch thread falls behind the
        # GPU -- 36-39 ms seen against 31.3 -- so the faster of two short loops is reported)
        g_ms = min(hip_event_time_ms(lambda: em.g_step(sdb, bqb, hdb), 3, stream) for _ in range(2))
        d_ms = min(hip_event_time_ms(lambda: em.d_step(sdb, bqb, hdb), 3, stream) for _ in range(2))
        flop = 3 * 12.78e9 * nb + 3 * 2 * 110380.0 * 128 * 128 * nb + 2 * 2 * 0.468e9 * nb
        out['enet_pat'] = {'batch': nb, 'patch': '32->128', 'g_trainer_ms': round(g_ms, 2), 'd_trainer_ms': round(d_ms, 2),
                           'patches_per_s': round(3 * nb / ((d_ms + 3 * g_ms) * 1e-3), 1),
                           'g_trainer_tflops': round(flop / (g_ms * 1e-3) / 1e12, 1),
                           'g_trainer_frac_of_fp32_mfma_peak': round(flop / (g_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3),
                           'weights': 'random VGG-shaped (timing only)', 'statistic': 'faster of two loops of 3 runs'}
        del em, sdb, bqb, hdb
        # the tile shape BASELINE's config names: 512x512 HR tiles, 4 per GPU (the discriminator's first dense layer
        # grows to 16*16*512 inputs)
        try:
            nt = 4
            em = model_enet.EnetModel('pat', model_vgg.random_vgg_weights(0), device=dev, seed=1, image_size=512)
            sdb, bqb, hdb = next(enet_train.synthetic_batches(nt, dev, hd_size=512))
            em.g_step(sdb, bqb, hdb); em.d_step(sdb, bqb, hdb)
            g_ms = min(hip_event_time_ms(lambda: em.g_step(sdb, bqb, hdb), 3, stream) for _ in range(2))
            d_ms = min(hip_event_time_ms(lambda: em.d_step(sdb, bqb, hdb), 3, stream) for _ in range(2))
            out['enet_pat']['tiles_512'] = {'batch': nt, 'patch': '128->512', 'g_trainer_ms': round(g_ms, 2), 'd_trainer_ms': round(d_ms, 2),
                                            'g_trainer_frac_of_fp32_mfma_peak': round(16 * nt / nb * flop / (g_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3)}
            del em
        except Exception as exc:
            out['enet_pat']['tiles_512'] = {'error': repr(exc)}
    except Exception as exc:             # a secondary number must never take the primary line down
        out['enet_pat'] = {'error': repr(exc)}
    return out


def vdsr_recipe(dev, stream, batch=64, size=128):
    """The shape the reference itself trains VDSR-20 at (vdsr/makefile:22-29: --image_size=128 --batch_size=64): rows of
    128 pixels are too wide for full-width LDS tiles, so the body layers run on 32-column strips (conv_pipe_strip_kernel,
    wgrad_pipe_strip_kernel).  Train step and the three body-layer kernels, HIP events on the launch stream."""
    from ml_super_resolution_amd import ops
    from ml_super_resolution_amd.vdsr import model_vdsr
    try:
        m = model_vdsr.VdsrModel(num_layers=20, use_adam=True, device=dev, seed=107)
        g = torch.Generator(device=dev).manual_seed(108)
        hd = torch.rand((batch, size, size, 3), device=dev, generator=g) * 2 - 1
        sd = (hd + 0.1 * torch.randn((batch, size, size, 3), device=dev, generator=g)).clamp(-1, 1)
        for _ in range(2):
            m.train_step(sd, hd, 5e-5)
        ms = hip_event_time_ms(lambda: m.train_step(sd, hd, 5e-5), 5, stream)
        fwd_ms = hip_event_time_ms(lambda: m.forward(sd), 5, stream)
        px = batch * size * size
        tf = TRAIN_FLOP_PER_PX * px / (ms * 1e-3) / 1e12
        out = {'what': 'VDSR-20 train step at the reference recipe\'s shape (vdsr/makefile:22-29): batch %d of %dx%d, fwd + MSE/L2 + bwd + TF-Adam' % (batch, size, size),
               'train_ms': round(ms, 3), 'patches_per_s': round(batch / (ms * 1e-3), 1), 'train_tflops': round(tf, 2),
               'train_frac_of_fp32_mfma_peak': round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
               'fwd_ms': round(fwd_ms, 3), 'fwd_hr_mpix_per_s': round(px / (fwd_ms * 1e-3) / 1e6, 2),
               'fwd_frac_of_fp32_mfma_peak': round(FWD_FLOP_PER_PX * px / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
        k, b = m.stack.kernel(5), m.stack.bias(5)
        del m
        x = torch.rand((batch, size, size, 64), device=dev) * 2 - 1
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        dw, db = torch.empty_like(k), torch.empty(64, device=dev)
        ws = torch.empty((ops.bwd_filter_workspace_bytes(x.shape, k.shape) + 3) // 4, device=dev)
        fns = {'fwd': lambda: ops.conv2d_fwd(x, k, b, 'same', 'relu', out=y),
               'dgrad': lambda: ops.conv2d_bwd_data(y, k, x.shape, 'same', x_in=x, in_act='relu', out=dx),
               'wgrad': lambda: ops.conv2d_bwd_filter(x, y, k.shape, 'same', w_for_decay=k, wd_scale=1e-4, dw=dw, dbias=db, workspace=ws)}
        for name, fn in fns.items():
            fn()
            lms = hip_event_time_ms(fn, 10, stream)
            ltf = MID_LAYER_FLOP_PER_PX * px / (lms * 1e-3) / 1e12
            out[name] = {'launch_ms': round(lms, 4), 'achieved': round(ltf, 2), 'frac': round(ltf / PEAK_FP32_MFMA_TFLOPS, 4)}
        out['kernels'] = 'conv_pipe_strip_kernel (fwd, dgrad + ReluGrad), wgrad_rows_strip_kernel + reduce_partials_kernel: 3x3 64->64 at %dx%dx%d, TFLOP/s of %g peak' % (batch, size, size, PEAK_FP32_MFMA_TFLOPS)
        return out
    except Exception as exc:             # a secondary number must never take the primary line down
        return {'error': repr(exc)}


def launch_ranks(n):
    """Start N ranks of this script as fresh child processes and wait for them.  Runs BEFORE this process has
    made any GPU call (device_count() does not initialise HIP), and never execs: the children are ordinary
    subprocesses with the torchrun environment, the parent only relays their output and return codes."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ndev = torch.cuda.device_count()
    base = dict(os.environ)
    base['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    if ndev < n:
        base.setdefault('SRX_DIST_BACKEND', 'gloo')      # several ranks per GPU: RCCL needs one GPU per rank
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    pending = list(procs)
    failed_at = None
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is not None:
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    failed_at = failed_at or time.time()
        if failed_at is not None and pending and time.time() - failed_at > 20:
            for p in pending:                           # a rank died: its peers would wait in a collective forever
                p.kill()                                # (exactly the PIDs started above)
        time.sleep(0.05)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary north-star numbers')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='do not run the two rocprofv3 --pmc passes; roofline.traffic then comes from profiles/traffic.json')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    # HBM traffic of the dominant kernel, measured by two profiler child processes -- first thing, while this process has
    # not touched the GPU yet (N = 1 only; not when this process is itself being profiled)
    traffic = None
    if args.gpus == 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1 and not args.no_live_traffic and \
            os.environ.get('SRX_BENCH_LIVE_TRAFFIC', '1') != '0' and not under_profiler():
        traffic = live_traffic()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
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
    backend = None
    if world > 1:
        srx_dist.attach(model.stack, world, timed=True)
        backend = torch.distributed.get_backend()

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
    allreduce_ms = srx_dist.allreduce_ms(model.stack, last=args.steps) if world > 1 else None
    if world > 1:
        nccl = backend == 'nccl'
        t = torch.tensor([dt], dtype=torch.float64, device=dev if nccl else 'cpu')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    patches_per_s = world * BATCH * args.steps / dt

    # ---- forward-only rate and the dominant kernels' rooflines (rank 0, HIP events on the launch stream)
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
        cfg = {'workload': 'BASELINE configs[2]: VDSR 20-layer 3x3x64 residual, RGB, batch 256x41x41 '
                           'per GPU, fwd+bwd+TF-Adam, random-init weights',
               'global_batch': world * BATCH, 'patch': SIZE, 'parallelism': 'dp%d' % world}
        if world > 1:
            cfg['collective'] = 'one all_reduce(AVG) of %d floats per step, backend %s' % (model.stack.flat_size, backend)
            if ndev < world:
                cfg['rehearsal'] = '%d ranks share %d GPU(s); gradients exchanged over %s through the host -- a ' \
                                   'functional rehearsal of the N>1 path, not a scaling measurement' % (world, ndev, backend)
        line = {
            'metric': 'VDSR-20 training patches/sec (41x41, batch 256 per GPU); fwd HR megapixels/sec alongside',
            'value': round(patches_per_s, 1), 'unit': 'patches/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': cfg,
            'fwd_hr_mpix_per_s': round(px / (fwd_ms * 1e-3) / 1e6, 2),
            'fwd_ms': round(fwd_ms, 3),
            'train_step_tflops': round(train_tf, 2),
            'train_step_frac_of_fp32_mfma_peak': round(train_tf / PEAK_FP32_MFMA_TFLOPS, 4),
            'roofline': dict({'bound': 'mfma', 'kernel': 'conv_pipe_kernel<3,3,64,4,fwd> (3x3 64->64 fwd+bias+ReLU)',
                              'achieved': round(achieved_tf, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': round(achieved_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                              'launch_ms': round(mid_ms, 4)}, **(traffic if traffic is not None else measured_traffic())),
        }
        if allreduce_ms is not None:
            line['allreduce_ms'] = round(allreduce_ms, 4)
        if not args.no_extras and world == 1:
            # (N = 1 only, like the CPU baselines: the other ranks of an N > 1 run would sit in process-group teardown
            # while rank 0 measured them)
            line.update(extras(model, dev, stream, x64, y64, px))
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(rank)
            line['cpu_library_baseline'] = cpu_library_baseline(rank)
            line.update(cpu_small_configs(rank))
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
