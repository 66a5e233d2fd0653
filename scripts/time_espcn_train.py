#!/usr/bin/env python3
"""ESPCN 3x train step (forward, MSE in sub-pixel space, backward, Adam) at the reference's patch size 17x17."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd.espcn import model_espcn
dev = torch.device('cuda')
for b in [int(v) for v in sys.argv[1:]] or (32, 64, 256):
    m = model_espcn.EspcnModel(3, device=dev, seed=1)
    lr = torch.rand((b, 17, 17, 3), device=dev) * 2 - 1
    hr = torch.rand((b, 17, 17, 27), device=dev) * 2 - 1
    for _ in range(5): m.train_step(lr, hr, 1e-3)
    st = m.stack.static_step_inputs(lr.shape, hr.shape)      # (graph replay: batches written straight into the captured step's inputs)
    if st is not None:
        st[0].copy_(lr); st[1].copy_(hr); lr, hr = st
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 50
    s.record()
    for _ in range(it): m.train_step(lr, hr, 1e-3)
    e.record(); e.synchronize()
    us = s.elapsed_time(e) / it * 1e3
    print('ESPCN 3x train step, batch %4d x 17x17: %7.1f us  (%8.0f patches/s)' % (b, us, b / us * 1e6), flush=True)
