#!/usr/bin/env python3
"""Forward-only VDSR-20 on whole images (the inference path of vdsr/experiment_resolve.py): HR megapixels/s by image
size.  41-wide patches run on the pipelined kernel with full-width tiles, wide images on its column-strip variant."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd.vdsr import model_vdsr
dev = torch.device('cuda')
m = model_vdsr.VdsrModel(20, device=dev, seed=1)
for n, h, w in ((256, 41, 41), (16, 96, 96), (32, 64, 64), (4, 256, 256), (1, 512, 512), (1, 1080, 1920)):
    x = torch.rand((n, h, w, 3), device=dev) * 2 - 1
    for _ in range(2):
        m.forward(x)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    s.record()
    for _ in range(it):
        m.forward(x)
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / it
    px = n * h * w
    print('%4d x %4d x %4d : %8.2f ms  %7.1f HR-MP/s  (%.0f%% of the fp32-MFMA peak)' % (n, h, w, ms, px / ms / 1e3, 100 * px * 1334016 / (ms * 1e-3) / 157.3e12))
