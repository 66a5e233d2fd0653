#!/usr/bin/env python3
"""Diagnostic (needs a -DSRX_TRACE build of libsrx): per-wave cycle split of the pipelined 3x3 64->64 wgrad."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
x = torch.rand((256, 41, 41, 64), device=dev) * 2 - 1
dy = torch.rand((256, 41, 41, 64), device=dev) * 2 - 1
dw = torch.empty((3, 3, 64, 64), device=dev); db = torch.empty(64, device=dev)
ws = torch.empty((ops.bwd_filter_workspace_bytes(x.shape, dw.shape) + 3) // 4, device=dev)
tr = torch.zeros(512 * 4 * 12, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.conv2d_bwd_filter(x, dy, dw.shape, 'same', dw=dw, dbias=db, workspace=ws)
os.environ['SRX_TRACE_PTR'] = hex(tr.data_ptr())
ops.conv2d_bwd_filter(x, dy, dw.shape, 'same', dw=dw, dbias=db, workspace=ws)
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(-1, 12).astype(np.float64)
t = t[t[:, 1] > 0]
tot = t[:, 1] - t[:, 0]
ideal = 430336 / 16 * 4 * 144 / len(t) * 32
print('waves %d  total cycles median %.0f  (ideal MFMA %.0f -> %.2fx)' % (len(t), np.median(tot), ideal, np.median(tot) / ideal))
print('step loops   median %.0f (%.1f%%)  = %.2fx ideal' % (np.median(t[:, 2]), 100 * np.median(t[:, 2] / tot), np.median(t[:, 2]) / ideal))
print('unit ends    median %.0f (%.1f%%)' % (np.median(t[:, 3]), 100 * np.median(t[:, 3] / tot)))
print('rest (prologue, first tile, partial write) %.1f%%' % (100 * np.median((tot - t[:, 2] - t[:, 3]) / tot)))
