#!/usr/bin/env python3
"""Where does a small launch spend its time?  One 3x3 conv at 17x17 (the ESPCN patch size), by batch size, channel
counts and activation."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops
dev = torch.device('cuda')
def t(n, hw, cin, cout, act, k=3, iters=300):
    x = torch.rand((n, hw, hw, cin), device=dev)
    w = torch.rand((k, k, cin, cout), device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    y = ops.conv2d_fwd(x, w, b, 'same', act)
    for _ in range(20): ops.conv2d_fwd(x, w, b, 'same', act, out=y)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): ops.conv2d_fwd(x, w, b, 'same', act, out=y)
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for n in [int(v) for v in sys.argv[1:]] or (1, 8, 32, 128, 512):
    print('N=%4d 17x17: 64->32 tanh %6.1f us | 64->32 none %6.1f | 64->64 relu %6.1f | 32->27 none %6.1f | 3->64 5x5 tanh %6.1f'
          % (n, t(n, 17, 64, 32, 'tanh'), t(n, 17, 64, 32, None), t(n, 17, 64, 64, 'relu'), t(n, 17, 32, 27, None), t(n, 17, 3, 64, 'tanh', 5)), flush=True)
