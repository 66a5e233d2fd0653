#!/usr/bin/env python3
"""1x1 64->64 convolution (+ residual + ReLU) and its data gradient at the EnhanceNet block shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
for n, hw in ((4, 128), (16, 128), (1, 32)):
    x = torch.rand((n, hw, hw, 64), device=dev); t = torch.rand_like(x); y = torch.empty_like(x)
    w = torch.rand((1, 1, 64, 64), device=dev) * 0.1; b = torch.zeros(64, device=dev)
    def run(fn, it=200):
        for _ in range(10): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); e.synchronize()
        return s.elapsed_time(e) / it * 1e3
    f = run(lambda: ops.conv2d_fwd(x, w, b, 'same', None, skip=t, post_add_relu=True, out=y))
    d = run(lambda: ops.conv2d_bwd_data(x, w, x.shape, 'same', x_in=t, in_act='relu', out=y))
    px = n * hw * hw
    print('%2d x %d^2: fwd+skip+relu %6.1f us (%.2f TB/s)  dgrad+mask %6.1f us' % (n, hw, f, px * 768 / f / 1e6, d), flush=True)
