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
for (n, h, w) in ((1, 128, 128), (1, 256, 256), (1, 360, 640), (1, 720, 1280), (4, 360, 640)):
    x = torch.rand((n, h, w, 3), device=dev) * 2 - 1
    flop = 62016.0 * n * h * w
    t = timeit(lambda: m.super_resolve(x, single_launch=False))
    t2 = timeit(lambda: m.super_resolve_two_step(x))
    print('ESPCN 3x %dx%dx%d LR: fused store %8.1f us = %6.1f HR-MP/s, %5.1f TFLOP/s (%4.1f %% of the fp32-MFMA peak) | two-step (standalone d2s) %8.1f us'
          % (n, h, w, t, 9.0 * n * h * w / t, flop / t / 1e6, 100 * flop / t / 1e6 / 157.3, t2), flush=True)
