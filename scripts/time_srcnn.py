#!/usr/bin/env python3
"""BASELINE configs[0]: SRCNN 9-1-5 (VALID) forward on one image, as the reference would run it (256 -> 243 crop,
RGB): 2.2 GFLOP; three launches against ONE (srx_srcnn_forward: the layers chained through LDS per 15x15 output tile)."""
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
def fwd(x):
    t = ops.conv2d_fwd(x, w1, b1, 'valid', 'relu')
    t = ops.conv2d_fwd(t, w2, b2, 'valid', 'relu')
    return ops.conv2d_fwd(t, w3, b3, 'valid', 'tanh')
params = [(w1, b1), (w2, b2), (w3, b3)]
def timed(fn):
    for _ in range(10): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / 100 * 1e3
shapes = [(int(a), int(b)) for a, b in zip(sys.argv[1::2], sys.argv[2::2])] or [(1, 243), (1, 256), (2, 243), (4, 243), (16, 243), (1, 33), (64, 33)]
for n, hw in shapes:
    x = rnd(n, hw, hw, 3)
    us = timed(lambda: fwd(x))
    us1 = timed(lambda: ops.srcnn_forward(x, params))
    same = torch.equal(fwd(x), ops.srcnn_forward(x, params))
    oh = hw - 12
    flop = 2.0 * n * (oh + 4) ** 2 * (243 * 64 + 64 * 32) + 2.0 * n * oh * oh * 800 * 3
    print('SRCNN 9-1-5 forward %2d x %dx%d -> %dx%d: three launches %7.1f us (%6.2f TFLOP/s) | one launch %7.1f us (%6.2f TFLOP/s, %7.2f MP/s out) bit-identical: %s'
          % (n, hw, hw, oh, oh, us, flop / us / 1e6, us1, flop / us1 / 1e6, n * oh * oh / us1, same))
