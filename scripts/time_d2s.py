#!/usr/bin/env python3
"""HIP-event timing of the sub-pixel map at the north-star bandwidth shape
[256,41,41,27] <-> [256,123,123,3] with 4 rotating buffer pairs (372 MB > 256 MiB Infinity Cache)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
bufs = [torch.rand((256, 41, 41, 27), device=dev) for _ in range(4)]
outs = [torch.empty((256, 123, 123, 3), device=dev) for _ in range(4)]
def run(fn, iters=40):
    for i in range(8): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3
t1 = run(lambda i: ops.depth_to_space(bufs[i % 4], 3, out=outs[i % 4]))
t2 = run(lambda i: ops.space_to_depth(outs[i % 4], 3, out=bufs[i % 4]))
t3 = run(lambda i: outs[i % 4].view(-1).copy_(bufs[i % 4].view(-1)))
by = 2 * 256 * 41 * 41 * 27 * 4
print('d2s %.1f us  %.2f TB/s (%.1f%% of 8 TB/s) | s2d %.1f us %.2f TB/s | torch copy_ same bytes %.1f us %.2f TB/s'
      % (t1, by / t1 / 1e6, 100 * by / t1 / 1e6 / 8, t2, by / t2 / 1e6, t3, by / t3 / 1e6))
