#!/usr/bin/env python3
"""SRCNN 9-1-5 (VALID) forward on larger inputs than BASELINE configs[0], three per-layer launches: microseconds per layer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(1)
def rnd(*shape, s=1.0):
    return (torch.rand(shape, device=dev, generator=g) * 2 - 1) * s
w1, b1 = rnd(9, 9, 3, 64, s=0.06), rnd(64, s=0.1)
w2, b2 = rnd(1, 1, 64, 32, s=0.12), rnd(32, s=0.1)
w3, b3 = rnd(5, 5, 32, 3, s=0.03), rnd(3, s=0.1)
def timed(fn, it=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
args = [int(v) for v in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(1, 512, 512), (1, 720, 1280), (16, 128, 128), (64, 33, 33)]
for n, h, w in shapes:
    x = rnd(n, h, w, 3)
    t1 = ops.conv2d_fwd(x, w1, b1, 'valid', 'relu')
    t2 = ops.conv2d_fwd(t1, w2, b2, 'valid', 'relu')
    a = timed(lambda: ops.conv2d_fwd(x, w1, b1, 'valid', 'relu', out=t1))
    b = timed(lambda: ops.conv2d_fwd(t1, w2, b2, 'valid', 'relu', out=t2))
    c = timed(lambda: ops.conv2d_fwd(t2, w3, b3, 'valid', 'tanh'))
    p1 = n * (h - 8) * (w - 8)
    p3 = n * (h - 12) * (w - 12)
    f1, f2, f3 = 2.0 * p1 * 243 * 64, 2.0 * p1 * 64 * 32, 2.0 * p3 * 800 * 3
    print('SRCNN %dx%dx%d: 9x9 3->64 %7.1f us (%5.1f TFLOP/s, writes %.0f MB at %.2f TB/s) | 1x1 64->32 %7.1f us (%.2f TB/s) | 5x5 32->3 %7.1f us (%5.1f TFLOP/s, reads %.0f MB at %.2f TB/s)'
          % (n, h, w, a, f1 / a / 1e6, p1 * 256 / 1e6, p1 * 256 / a / 1e6, b, p1 * 384 / b / 1e6, c, f3 / c / 1e6, p1 * 128 / 1e6, p1 * 128 / c / 1e6), flush=True)
