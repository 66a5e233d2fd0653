#!/usr/bin/env python3
"""HIP-event timing of the dominant kernels (3x3 64->64 fwd / dgrad / wgrad at 256x41x41) and of a
whole VDSR-20 train step; prints one line.  Used for A/B sweeps (env knobs read by libsrx)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd import ops  # noqa: E402
from ml_super_resolution_amd.vdsr import model_vdsr  # noqa: E402

dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((256, 41, 41, 64), device=dev, generator=g) * 2 - 1
dy = torch.rand((256, 41, 41, 64), device=dev, generator=g) * 2 - 1
w = (torch.rand((3, 3, 64, 64), device=dev, generator=g) * 2 - 1) * 0.07
b = torch.zeros(64, device=dev)
y, dx, dw, db = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w), torch.empty(64, device=dev)
ws = torch.empty((ops.bwd_filter_workspace_bytes(x.shape, w.shape) + 3) // 4, device=dev)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3


t_f = timeit(lambda: ops.conv2d_fwd(x, w, b, 'same', 'relu', out=y))
t_d = timeit(lambda: ops.conv2d_bwd_data(dy, w, x.shape, 'same', x_in=x, in_act='relu', out=dx))
t_w = timeit(lambda: ops.conv2d_bwd_filter(x, dy, w.shape, 'same', w_for_decay=w, wd_scale=1e-4, dw=dw, dbias=db, workspace=ws))
m = model_vdsr.VdsrModel(20, use_adam=True, seed=1)
hd = torch.rand((256, 41, 41, 3), device=dev, generator=g) * 2 - 1
sd = (hd + 0.1 * torch.randn((256, 41, 41, 3), device=dev, generator=g)).clamp(-1, 1)
t_s = timeit(lambda: m.train_step(sd, hd, 5e-5), iters=10, warm=3)
ideal = 430336 * 73728 / 157.3e12 * 1e6
print('%s fwd %.1f us (%.1f%%)  dgrad %.1f us (%.1f%%)  wgrad+reduce %.1f us (%.1f%%)  step %.2f ms -> %.0f patches/s'
      % (os.environ.get('TAG', ''), t_f, 100 * ideal / t_f, t_d, 100 * ideal / t_d, t_w, 100 * ideal / t_w,
         t_s / 1e3, 256 / (t_s * 1e-6)))
