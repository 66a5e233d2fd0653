#!/usr/bin/env python3
"""Diagnostic: compare the pipelined conv kernels with the two-workgroup kernels element by element."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops, _lib
dev = torch.device('cuda')
N, H, W, C = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (3, 41, 41, 64))]
PAD = sys.argv[5] if len(sys.argv) > 5 else 'same'
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((N, H, W, C), device=dev, generator=g) * 2 - 1
OH, OW = (H, W) if PAD == 'same' else (H - 2, W - 2)
dy = torch.rand((N, OH, OW, C), device=dev, generator=g) * 2 - 1
w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * 0.07
b = torch.rand(C, device=dev, generator=g) * 0.2 - 0.1
lib = _lib.lib()
def run(path):
    lib.srx_set_conv_path(path)
    y = ops.conv2d_fwd(x, w, b, PAD, 'relu')
    dx = ops.conv2d_bwd_data(dy, w, x.shape, PAD, x_in=x, in_act='relu')
    dx0 = ops.conv2d_bwd_data(dy, w, x.shape, PAD)
    torch.cuda.synchronize()
    return y, dx, dx0
ref = run(0)
got = run(1)
for name, r, t in zip(('fwd', 'dgrad+mask', 'dgrad'), ref, got):
    bad = (r != t) & ~(torch.isnan(r) & torch.isnan(t))
    print(name, 'mismatches', int(bad.sum()), 'of', r.numel(), ' max abs diff %.3e' % float((r - t).abs().max()))
    if bad.any():
        idx = bad.nonzero()
        print('  images', sorted(set(idx[:, 0].tolist()))[:10])
        print('  rows', sorted(set(idx[:, 1].tolist()))[:50])
        print('  cols', sorted(set(idx[:, 2].tolist()))[:50])
        print('  chans', sorted(set(idx[:, 3].tolist()))[:70])
        for k in range(min(6, len(idx))):
            i = tuple(idx[k].tolist())
            print('   ', i, float(r[i]), float(t[i]))
