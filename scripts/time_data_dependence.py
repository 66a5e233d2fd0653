#!/usr/bin/env python3
"""Does the 3x3 64->64 forward kernel's time depend on the VALUES it multiplies (chip power -> clock)?  The same launch
on uniform(-1,1) inputs, on post-ReLU inputs (half of them zero: what the layer sees inside VDSR), on all zeros."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops


def timed(fn, it=40):
    for _ in range(10):
        fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it):
            fn()
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / it)
    return best


g = torch.Generator(device='cuda').manual_seed(3)
w = (torch.randn((3, 3, 64, 64), device='cuda', generator=g) * (2.0 / 576) ** 0.5)
b = torch.zeros(64, device='cuda')
u = torch.rand((256, 41, 41, 64), device='cuda', generator=g) * 2 - 1
cases = [('uniform(-1,1)', u), ('relu(normal): half zeros', torch.relu(torch.randn_like(u))), ('uniform(0,1)', torch.rand_like(u)),
         ('90 % zeros', torch.relu(torch.randn_like(u) - 1.2816)), ('all zeros', torch.zeros_like(u))]
y = torch.empty_like(u)
for rep in range(2):
    for name, x in cases:
        t = timed(lambda: ops.conv2d_fwd(x, w, b, 'same', 'relu', out=y))
        if rep:
            print('%-28s %7.2f us  %5.1f %% of the fp32-MFMA peak' % (name, t, 100 * 2 * 9 * 64 * 64 * 256 * 41 * 41 / t / 1e6 / 157.3), flush=True)
