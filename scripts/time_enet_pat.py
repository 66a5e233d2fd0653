#!/usr/bin/env python3
"""BASELINE configs[4] as the reference trains it (enet/enet/experiment_train.py:15-22): EnhanceNet-PAT, batches of
32x32 -> 128x128 patches, VGG-19 perceptual + texture + adversarial losses.  Times one generator run and one
discriminator run (random VGG-shaped weights: timing only).
  time_enet_pat.py [batch=64] [iters=5] [hd_size=128] [g|d|gd]      hd_size 512 = the 512x512 tiles BASELINE's config names"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_super_resolution_amd.enet import model_enet, model_vgg, experiment_train
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
which = sys.argv[4] if len(sys.argv) > 4 else 'gd'      # 'g' / 'd': only that trainer (for per-trainer profiles)
dev = torch.device('cuda')
m = model_enet.EnetModel('pat', model_vgg.random_vgg_weights(0), device=dev, seed=1, image_size=S)
sd, bq, hd = next(experiment_train.synthetic_batches(n, dev, hd_size=S))
def timeit(fn, k):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
tg = timeit(lambda: m.g_step(sd, bq, hd), iters) if 'g' in which else float('nan')
td = timeit(lambda: m.d_step(sd, bq, hd), iters) if 'd' in which else float('nan')
# FLOPs of one generator run: VGG-19's 16 convolutions at 128x128 = 6.39 GMAC = 12.78 GFLOP per image and pass, three
# passes (features of sr and of hd, data gradient of the sr pass); the generator 110,380 MAC per HR pixel forward,
# twice that backward; the discriminator 0.47 GMAC per image forward + its data gradient
area = (S / 128.0) ** 2
vgg = 3 * 12.78e9 * n * area
gen = 3 * 2 * 110380.0 * S * S * n
disc = 2 * 2 * 0.468e9 * area * n          # (convolutions and the first dense layer both grow with the area)
flop = vgg + gen + disc
print('ENet-PAT batch %d x (%d->%d): g_trainer run %.2f ms = %.1f TFLOP/s (%.0f%% of the fp32-MFMA peak; %.2f TFLOP: VGG-19 %.2f, '
      'generator %.2f, discriminator %.2f), d_trainer run %.2f ms; one cycle of the schedule (1 d + 3 g runs) %.1f ms = %.1f patches/s'
      % (n, S // 4, S, tg, flop / tg / 1e9, 100 * flop / tg / 1e9 / 157.3, flop / 1e12, vgg / 1e12, gen / 1e12, disc / 1e12, td, td + 3 * tg,
         3 * n / ((td + 3 * tg) * 1e-3)))
