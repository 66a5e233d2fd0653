#!/usr/bin/env python3
"""Diagnostic (needs ml_super_resolution_amd/libsrx_trace.so = a -DSRX_TRACE build): are the slow workgroups of the
3x3 64->64 forward launch the SAME from launch to launch (systematic: a chained multi-layer kernel would gain nothing)
or random (sum over layers of the slowest workgroup > slowest sum: chaining the layers per patch would gain)?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsrx_trace.so')
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
L = 12
xs = [torch.rand((256, 41, 41, 64), device=dev) * 2 - 1 for _ in range(2)]
w = (torch.rand((3, 3, 64, 64), device=dev) * 2 - 1) * 0.07
b = torch.zeros(64, device=dev)
trs = [torch.zeros(512 * 4 * 12, dtype=torch.int64, device=dev) for _ in range(L)]
# the trace pointer is read once per process (with the other knobs, at the first call): one buffer, copied out after every launch
os.environ['SRX_TRACE_PTR'] = hex(trs[0].data_ptr())
for _ in range(3):
    ops.conv2d_fwd(xs[0], w, b, 'same', 'relu', out=xs[1])
life = []
for l in range(L):
    trs[0].zero_()
    ops.conv2d_fwd(xs[l & 1], w, b, 'same', 'relu', out=xs[(l + 1) & 1])
    torch.cuda.synchronize()
    t = trs[0].cpu().numpy().reshape(-1, 12).astype(np.float64)[:256 * 4]
    rt0, rt1 = t[:, 5].reshape(256, 4), t[:, 6].reshape(256, 4)
    start = rt0.min()
    life.append((rt1.max(axis=1) - start) / 100.0)          # us from the launch's first wave start to this workgroup's end
life = np.array(life)                                        # [launch, workgroup]
print('per-launch span (us): ', np.round(life.max(axis=1), 1))
print('per-launch median workgroup end (us): ', np.round(np.median(life, axis=1), 1))
print('sum over launches of the slowest workgroup  %.1f us' % life.max(axis=1).sum())
print('slowest workgroup of the summed times       %.1f us   (what a per-patch chain of the layers would take)' % life.sum(axis=0).max())
print('sum of medians                              %.1f us' % np.median(life, axis=1).sum())
c = np.corrcoef(life)
print('mean correlation of workgroup end times between launches: %.2f' % ((c.sum() - L) / (L * L - L)))
slow = life.argmax(axis=1)
print('slowest workgroup per launch:', slow, ' XCD', slow % 8)
