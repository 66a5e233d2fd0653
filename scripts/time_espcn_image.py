#!/usr/bin/env python3
"""ESPCN 3x on whole images (the reference's experiment_test.py runs whole images): LR 128^2, 256^2, 360x640, 720x1280."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd.espcn import model_espcn
dev = torch.device('cuda')
m = model_espcn.EspcnModel(3, device=dev, seed=1)
def timeit(fn, iters=50):
    for _ in range(10): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (n, h, w) in ((1, 128, 128), (1, 170, 170), (1, 256, 256), (1, 300, 300), (1, 362, 362), (1, 360, 640), (1, 720, 1280), (4, 360, 640)):
    x = torch.rand((n, h, w, 3), device=dev) * 2 - 1
    flop = 62016.0 * n * h * w
    t = timeit(lambda: m.super_resolve(x, single_launch=False))
    t2 = timeit(lambda: m.super_resolve_two_step(x))
    t1 = timeit(lambda: m.super_resolve(x, single_launch=True)) if n * h * w <= 300000 else float('nan')
    best = min(t, t1) if t1 == t1 else t
    print('ESPCN 3x %dx%dx%d LR: three launches, sub-pixel store fused %8.1f us | ONE launch (srx_espcn_forward) %8.1f us | best = %6.1f HR-MP/s, %5.1f TFLOP/s '
          '(%4.1f %% of the fp32-MFMA peak) | two-step (standalone d2s) %8.1f us'
          % (n, h, w, t, t1, 9.0 * n * h * w / best, flop / best / 1e6, 100 * flop / best / 1e6 / 157.3, t2), flush=True)
# per layer (the three launches of the graph, timed one by one on the largest shape)
from ml_super_resolution_amd import ops
st = m.stack
for (n, h, w) in ((1, 256, 256), (1, 720, 1280)):
    x = torch.rand((n, h, w, 3), device=dev) * 2 - 1
    t1 = ops.conv2d_fwd(x, st.kernel(0), st.bias(0), 'same', 'tanh')
    t2 = ops.conv2d_fwd(t1, st.kernel(1), st.bias(1), 'same', 'tanh')
    out = torch.empty((n, h * 3, w * 3, 3), device=dev)
    a = timeit(lambda: ops.conv2d_fwd(x, st.kernel(0), st.bias(0), 'same', 'tanh', out=t1))
    b = timeit(lambda: ops.conv2d_fwd(t1, st.kernel(1), st.bias(1), 'same', 'tanh', out=t2))
    c = timeit(lambda: ops.conv2d_fwd(t2, st.kernel(2), st.bias(2), 'same', None, subpixel_r=3, out=out))
    a0 = timeit(lambda: ops.conv2d_fwd(x, st.kernel(0), st.bias(0), 'same', 'relu', out=t1))
    b0 = timeit(lambda: ops.conv2d_fwd(t1, st.kernel(1), st.bias(1), 'same', 'relu', out=t2))
    px = n * h * w
    print('ESPCN layers %dx%dx%d: f1 5x5 3->64 tanh %7.1f us (%5.1f TF; with ReLU instead %7.1f) | f2 3x3 64->32 tanh %7.1f us (%5.1f TF; with ReLU %7.1f) | f3 3x3 32->27 + sub-pixel store %7.1f us (%5.1f TF)'
          % (n, h, w, a, 9600.0 * px / a / 1e6, a0, b, 36864.0 * px / b / 1e6, b0, c, 15552.0 * px / c / 1e6), flush=True)
