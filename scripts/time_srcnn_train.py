#!/usr/bin/env python3
"""SRCNN 9-1-5 train step at the reference's batch (srcnn/srcnn.py:14-16,28-40: 64 crops of 243 x 243 -> 231 x 231): forward, row-norm
loss, backward, Adam(1e-3, .5, .9).  Usage: time_srcnn_train.py [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd.srcnn import srcnn as srcnn_mod
dev = torch.device('cuda')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sm = srcnn_mod.SrcnnModel(device=dev, seed=101)
for i in range(3):
    sm.stack.kernel(i).mul_(60.0)
sd = torch.rand((b, 243, 243, 3), device=dev) * 2 - 1
hd = torch.rand((b, 231, 231, 3), device=dev) * 2 - 1
for _ in range(4): sm.train_step(sd, hd)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
it = 10
s.record()
for _ in range(it): sm.train_step(sd, hd)
e.record(); e.synchronize()
ms = s.elapsed_time(e) / it
f1, f2, f3 = 2.0 * 81 * 3 * 64 * 235 * 235, 2.0 * 64 * 32 * 235 * 235, 2.0 * 25 * 32 * 3 * 231 * 231
flop = b * (2 * f1 + 3 * f2 + 3 * f3)
print('SRCNN train step, batch %d x 243x243: %.3f ms  %.1f images/s  %.1f TFLOP/s (%.1f %% of the fp32-MFMA peak; algorithmic FLOPs)' % (b, ms, b / ms * 1e3, flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3), flush=True)
