#!/usr/bin/env python3
"""EnhanceNet generator (BASELINE config 5: 4x, 512x512 HR tiles) on one GPU: forward, and forward + backward +
Adam for a given gradient on sr_images (the VGG / discriminator losses that produce it are not built).
110,380 MAC per HR pixel forward; backward = wgrad + dgrad of every layer except the first's dgrad."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd.enet import model_enet  # noqa: E402

dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lr_hw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = model_enet.EnetGenerator(device=dev, seed=1)
sd = torch.rand((B, lr_hw, lr_hw, 3), device=dev) * 2 - 1
bq = torch.rand((B, 4 * lr_hw, 4 * lr_hw, 3), device=dev) * 2 - 1
d_sr = torch.randn((B, 4 * lr_hw, 4 * lr_hw, 3), device=dev) * 1e-3
hr_px = B * 16 * lr_hw * lr_hw
fwd_flop = 2.0 * 110380 * hr_px
# backward MACs per LR pixel: 2 x forward minus the first layer's dgrad (3*64*9)
bwd_flop = 2.0 * fwd_flop - 2.0 * 1728 * B * lr_hw * lr_hw


def timed(fn, it=5):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it


ms = timed(lambda: g.forward(sd, bq))
print('forward      %d x %d^2 HR: %8.2f ms  %7.1f HR-MP/s  %6.1f TFLOP/s (%2.0f%% of the fp32-MFMA peak)'
      % (B, 4 * lr_hw, ms, hr_px / ms / 1e3, fwd_flop / ms / 1e9, 100 * fwd_flop / ms / 1e9 / 157.3), flush=True)
state = {}


def step():
    g.forward(sd, bq, keep=True)
    g.adam_step(g.backward(d_sr), state)


ms = timed(step)
print('fwd+bwd+Adam %d x %d^2 HR: %8.2f ms  %7.1f HR-MP/s  %6.1f TFLOP/s (%2.0f%%)'
      % (B, 4 * lr_hw, ms, hr_px / ms / 1e3, (fwd_flop + bwd_flop) / ms / 1e9, 100 * (fwd_flop + bwd_flop) / ms / 1e9 / 157.3), flush=True)
