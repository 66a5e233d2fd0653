#!/usr/bin/env python3
"""Runs the three dominant VDSR kernels (3x3 64->64 fwd / dgrad / wgrad at 256x41x41) a few
times each, plus the sub-pixel map at the north-star bandwidth shape; meant to be run under
rocprofv3 (--kernel-trace --stats, or --pmc ... in separate passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
what = sys.argv[2] if len(sys.argv) > 2 else 'all'
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((256, 41, 41, 64), device=dev, generator=g) * 2 - 1
dy = torch.rand((256, 41, 41, 64), device=dev, generator=g) * 2 - 1
w = (torch.rand((3, 3, 64, 64), device=dev, generator=g) * 2 - 1) * 0.07
b = torch.zeros(64, device=dev)
y = torch.empty_like(x)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
db = torch.empty(64, device=dev)
ws = torch.empty((ops.bwd_filter_workspace_bytes(x.shape, w.shape) + 3) // 4, device=dev)
for _ in range(iters):
    if what in ('all', 'fwd'):
        ops.conv2d_fwd(x, w, b, 'same', 'relu', out=y)
    if what in ('all', 'dgrad'):
        ops.conv2d_bwd_data(dy, w, x.shape, 'same', x_in=x, in_act='relu', out=dx)
    if what in ('all', 'wgrad'):
        ops.conv2d_bwd_filter(x, dy, w.shape, 'same', w_for_decay=w, wd_scale=1e-4, dw=dw, dbias=db, workspace=ws)
if what in ('all', 'd2s'):
    # rotate over > 256 MiB of buffers so the Infinity Cache cannot serve the reads
    bufs = [torch.rand((256, 41, 41, 27), device=dev, generator=g) for _ in range(4)]
    outs = [torch.empty((256, 123, 123, 3), device=dev) for _ in range(4)]
    for i in range(iters * 4):
        ops.depth_to_space(bufs[i % 4], 3, out=outs[i % 4])
torch.cuda.synchronize()
print('done')
