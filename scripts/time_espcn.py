#!/usr/bin/env python3
"""BASELINE configs[1]: ESPCN 3x inference, batch 32 of 17x17 LR patches (launch-latency-bound by size:
0.57 GFLOP).  Times forward + depth-to-space eagerly and as a replayed HIP graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd.espcn import model_espcn
dev = torch.device('cuda')
m = model_espcn.EspcnModel(3, device=dev, seed=1)
x = torch.rand((32, 17, 17, 3), device=dev) * 2 - 1
def timeit(fn, iters=200):
    for _ in range(20): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3
hr_px = 32 * 51 * 51
t_two = timeit(lambda: m.super_resolve_two_step(x))
t_eager = timeit(lambda: m.super_resolve(x, use_graph=False, single_launch=False))
t_graph = timeit(lambda: m.super_resolve(x, use_graph=True, single_launch=False))
t_one = timeit(lambda: m.super_resolve(x, single_launch=True))
line = 'ESPCN 3x, batch 32x17x17: 4 launches (standalone d2s) %.1f us | 3 launches, fused store, eager %.1f us | the same as a HIP graph %.1f us | ONE launch (layers chained through LDS) %.1f us (%.1f HR-MP/s, %.1f TFLOP/s)' % (
    t_two, t_eager, t_graph, t_one, hr_px / t_one, 573.5 / t_one)
xb = torch.rand((1, 256, 256, 3), device=dev) * 2 - 1
t_img = timeit(lambda: m.super_resolve(xb, single_launch=False), 100)
t_img1 = timeit(lambda: m.super_resolve(xb, single_launch=True), 100)
line += ' | one 256x256 image: %.1f us per-layer, %.1f us single launch' % (t_img, t_img1)
for nb in (64, 128):
    xn = torch.rand((nb, 17, 17, 3), device=dev) * 2 - 1
    line += ' | batch %d: %.1f / %.1f us' % (nb, timeit(lambda: m.super_resolve(xn, single_launch=False)), timeit(lambda: m.super_resolve(xn, single_launch=True)))
xl = torch.rand((256, 41, 41, 3), device=dev) * 2 - 1
t_big2 = timeit(lambda: m.super_resolve_two_step(xl), 50)
t_big = timeit(lambda: m.super_resolve(xl, single_launch=False), 50)
line += ' | 256x41x41: two-step %.1f us, fused %.1f us' % (t_big2, t_big)
print(line)
