#!/usr/bin/env python3
"""1x1 convolution forward / data gradient (with the ReLU gradient of the layer input) by channel counts and size:
conv_1x1_kernel (default from 100,000 pixels) against conv_mfma_kernel (SRX_CONV_1X1_MIN_PIXELS=-1).
Usage: time_conv1x1.py [N H W ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
args = [int(v) for v in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(16, 128, 128), (64, 128, 128), (1, 720, 1280)]
def timed(fn, it=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
for n, h, w in shapes:
    for cin, cout in ((64, 32), (32, 64), (64, 64), (32, 32)):
        x = torch.rand((n, h, w, cin), device=dev)
        wt = torch.rand((1, 1, cin, cout), device=dev) - 0.5
        b = torch.zeros(cout, device=dev)
        dy = torch.rand((n, h, w, cout), device=dev)
        y = torch.empty((n, h, w, cout), device=dev); dx = torch.empty_like(x)
        tf = timed(lambda: ops.conv2d_fwd(x, wt, b, 'same', 'relu', out=y))
        tb = timed(lambda: ops.conv2d_bwd_data(dy, wt, x.shape, 'same', x_in=x, in_act='relu', out=dx))
        px = n * h * w
        print('%3d x %4d x %4d  %d->%d: forward %7.1f us (%.2f TB/s) | data gradient + ReLU mask %7.1f us (%.2f TB/s)'
              % (n, h, w, cin, cout, tf, px * (cin + cout) * 4 / tf / 1e6, tb, px * (2 * cin + cout) * 4 / tb / 1e6), flush=True)
