#!/usr/bin/env python3
"""srx_conv3x3_blocked (conv_wide_pipe_kernel) on VGG-19's wide layers as EnhanceNet-PAT runs them: forward and data
gradient, microseconds and share of the fp32-MFMA peak.  time_wide.py [images=4] [hd_size=512]
SRX_WIDE_PIPE=0 times the unpipelined kernel.  Two passes over the list, the second one printed: the first launches
of a process run at ramping clocks."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
layers = [('conv2_2', 128, 128, S // 2), ('conv3_1', 128, 256, S // 4), ('conv3_2', 256, 256, S // 4),
          ('conv4_1', 256, 512, S // 8), ('conv4_2', 512, 512, S // 8), ('conv5_1', 512, 512, S // 16)]


def timed(fn):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 100.0


for rep in range(2):
    tot_t = tot_f = 0.0
    for name, cin, cout, hw in layers:
        cib, cob = cin // 64, cout // 64
        x = torch.randn((cib, n, hw, hw, 64), device='cuda')
        dy = torch.randn((cob, n, hw, hw, 64), device='cuda')
        w = torch.randn((cib, cob, 3, 3, 64, 64), device='cuda') * 0.02
        b = torch.zeros(cout, device='cuda')
        y = torch.empty((cob, n, hw, hw, 64), device='cuda')
        dx = torch.empty_like(x)
        flop = 2.0 * 9 * cin * cout * n * hw * hw
        fwd = timed(lambda: ops.conv3x3_blocked(x, w, b, 'relu', out=y))
        bwd = timed(lambda: ops.conv3x3_blocked(dy, w, None, None, transpose=True, out=dx, mask=x, mask_act='relu'))
        tot_t += fwd + bwd
        tot_f += 2 * flop
        if rep == 1:
            print('%-8s %3d->%3d %4dx%-4d x%d  %6.2f GFLOP  fwd %8.1f us (%4.1f%%)  dgrad+mask %8.1f us (%4.1f%%)'
                  % (name, cin, cout, hw, hw, n, flop / 1e9, fwd, 100 * flop / fwd / 1e6 / 157.3, bwd, 100 * flop / bwd / 1e6 / 157.3),
                  flush=True)
    if rep == 1:
        print('all: %.1f us, %.1f%% of the fp32-MFMA peak' % (tot_t, 100 * tot_f / tot_t / 1e6 / 157.3))
