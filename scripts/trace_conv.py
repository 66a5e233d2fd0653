#!/usr/bin/env python3
"""Diagnostic (needs a -DSRX_TRACE build of libsrx): per-wave cycle split of the 3x3 64->64 forward."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
x = torch.rand((256, 41, 41, 64), device=dev) * 2 - 1
w = (torch.rand((3, 3, 64, 64), device=dev) * 2 - 1) * 0.07
b = torch.zeros(64, device=dev)
y = torch.empty_like(x)
tr = torch.zeros(512 * 4 * 12, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.conv2d_fwd(x, w, b, 'same', 'relu', out=y)
os.environ['SRX_TRACE_PTR'] = hex(tr.data_ptr())
ops.conv2d_fwd(x, w, b, 'same', 'relu', out=y)
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(-1, 12).astype(np.float64)
t = t[t[:, 1] > 0]
tot = t[:, 1] - t[:, 0]
print('waves', len(t), 'total cycles median %.0f  min %.0f max %.0f' % (np.median(tot), tot.min(), tot.max()))
print('mfma-section cycles median %.0f (%.1f%% of total)' % (np.median(t[:, 2]), 100 * np.median(t[:, 2] / tot)))
print('stage+barrier cycles median %.0f (%.1f%%)' % (np.median(t[:, 3]), 100 * np.median(t[:, 3] / tot)))
print('rest %.1f%%' % (100 * np.median((tot - t[:, 2] - t[:, 3]) / tot)))
rt0, rt1 = t[:, 5], t[:, 6]
clk = (t[:, 1] - t[:, 7]) / (rt1 - rt0) * 100.0
print('in-kernel clock MHz: median %.0f min %.0f max %.0f' % (np.median(clk), clk.min(), clk.max()))
print('realtime: kernel span %.1f us; wave start skew %.1f us; wave end skew %.1f us; median wave lifetime %.1f us; prologue median %.1f us'
      % ((rt1.max() - rt0.min()) / 100, (rt0.max() - rt0.min()) / 100, (rt1.max() - rt1.min()) / 100, np.median(rt1 - rt0) / 100,
         np.median(t[:, 0] - t[:, 7]) / np.median(clk)))
nm = 430336 / 16 * 4 * 144 / len(t)
print('ideal MFMA cycles per wave (32/MFMA): %.0f ; mfma-section / ideal = %.2f' % (nm * 32, np.median(t[:, 2]) / (nm * 32)))
hw = t[:, 4].astype(np.int64)
print('wave slot ids histogram', np.bincount(hw & 15)[:8], ' simd', np.bincount((hw >> 4) & 3))
slot = hw & 15
ideal = nm * 32
for sl in sorted(set(slot.tolist())):
    m = slot == sl
    life = (rt1[m] - rt0[m]) / 100
    print('slot %d: n=%d lifetime us med %.1f [%.1f..%.1f]  end-time rel. first-start med %.1f us  mfma/ideal med %.2f [%.2f..%.2f]  stage%% %.1f'
          % (sl, m.sum(), np.median(life), life.min(), life.max(), np.median(rt1[m] - rt0.min()) / 100,
             np.median(t[m, 2]) / ideal, (t[m, 2] / ideal).min(), (t[m, 2] / ideal).max(), 100 * np.median(t[m, 3] / tot[m])))
print('stage split (cycles, median): barrier-before %.0f  load+lds-write %.0f  barrier-after %.0f   (per wave, all tiles)'
      % (np.median(t[:, 8]), np.median(t[:, 9]), np.median(t[:, 3] - t[:, 8] - t[:, 9])))
print('group prologue cycles median %.0f (%.1f%%)   epilogue cycles median %.0f (%.1f%%)' % (np.median(t[:, 10]), 100 * np.median(t[:, 10] / tot), np.median(t[:, 11]), 100 * np.median(t[:, 11] / tot)))
