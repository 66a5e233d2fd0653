#!/usr/bin/env python3
"""VDSR-20 at the shape the reference itself trains it (vdsr/makefile:22-29: --image_size=128 --batch_size=64): train step,
forward, and the three body-layer kernels on column strips -- the `vdsr_recipe_64x128` object of the bench line, alone.
Usage: time_vdsr_recipe.py [batch size]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
args = [int(v) for v in sys.argv[1:]]
batch, size = (args + [64, 128])[:2] if args else (64, 128)
print(json.dumps(bench.vdsr_recipe(dev, torch.cuda.current_stream(), batch, size)), flush=True)
