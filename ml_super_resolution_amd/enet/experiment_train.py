"""
experiment_train.py -- mirror of enet/enet/experiment_train.py: flags `train_dir_path, vgg19_path, ckpt_path,
log_path, model='pat', batch_size=64` (:164-170) and the alternating schedule of its main loop (:105-160):

  step = session.run(model['step'])
  every 3rd step (step % 3 == 0) and if the model has a discriminator: ONE d_trainer run on a batch of its own,
  then always ONE g_trainer run on the next batch (which increments the global step).

A checkpoint (`tf.train.Saver` format and names, model_enet.EnetModel.save_tf_checkpoint) is written whenever
step % 1000 == 999 (:116-117) and the latest one is restored at start (:96-110).
Batches are (sd 32x32, bq 128x128, hd 128x128) in [-1, 1] (experiment_train.py:15-22).  `--train_dir_path`: a directory
of images as in the reference (enet/enet/datasets.py, mirrored in datasets.py: decode and crop on the host, the 25 % /
400 % resizes on the GPU, byte for byte), or an .npz of {'sd','bq','hd'}, or absent (synthetic batches).  VGG-19 weights: the .npz the reference downloads (`--vgg19_path`); when it is absent
and `--allow_random_vgg true`, VGG-shaped random weights (timing / smoke runs only).
With WORLD_SIZE > 1 (torchrun) the batch is sharded and the gradients of BOTH trainers are all-reduced (one flat
buffer each), as SURVEY 8e prescribes for config 5.
"""
import argparse
import json
import os

import numpy as np
import torch

from .. import dist as srx_dist
from . import model_enet, model_vgg


def parse_flags(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--train_dir_path', default=None)
    ap.add_argument('--vgg19_path', default=None)
    ap.add_argument('--ckpt_path', default=None)
    ap.add_argument('--log_path', default=None)
    ap.add_argument('--model', default='pat')
    ap.add_argument('--batch_size', type=int, default=64)
    # not in the reference (its loop never ends): stop after this many generator steps
    ap.add_argument('--stop_training_at_k_step', type=int, default=None)
    ap.add_argument('--allow_random_vgg', default='false')
    ap.add_argument('--save_every', type=int, default=1000)       # the reference's constant (step % 1000 == 999)
    return ap.parse_args(argv)


def synthetic_batches(batch_size, device, seed=0, hd_size=128):
    """hd_size: side of the HR patches (the reference trains on 128: experiment_train.py:15-22; BASELINE config 5
    names 512x512 tiles)."""
    g = torch.Generator(device=device).manual_seed(seed)
    s = hd_size // 4
    while True:
        hd = torch.rand((batch_size, hd_size, hd_size, 3), device=device, generator=g) * 2 - 1
        sd = hd.view(batch_size, s, 4, s, 4, 3).mean(dim=(2, 4))
        bq = sd.repeat_interleave(4, dim=1).repeat_interleave(4, dim=2)
        yield sd.contiguous(), bq.contiguous(), hd


def npz_batches(path, batch_size, device, seed=0):
    z = np.load(path)
    sd, bq, hd = (z[k].astype(np.float32) for k in ('sd', 'bq', 'hd'))
    rng = np.random.default_rng(seed)
    while True:
        idx = rng.integers(0, sd.shape[0], size=batch_size)
        yield tuple(torch.from_numpy(a[idx]).to(device) for a in (sd, bq, hd))


def main(argv=None, log=None):
    """`log`: optional callable receiving one dict per trainer run (tests)."""
    FLAGS = parse_flags(argv)
    if FLAGS.model not in ('p', 'pa', 'pat'):
        FLAGS.model = 'pat'                                              # experiment_train.py:88-89
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    device = torch.device('cuda', torch.cuda.current_device())
    if world > 1:
        srx_dist.init_process_group(rank, world, local_rank)
    if FLAGS.batch_size % world:
        raise SystemExit('batch_size must be divisible by the number of GPUs')
    weights = model_vgg.load_vgg_weights(FLAGS.vgg19_path) if FLAGS.vgg19_path else {}
    if not weights:
        if str(FLAGS.allow_random_vgg).lower() not in ('1', 'true', 'yes'):
            raise SystemExit('VGG-19 weights not found at %r (pass --allow_random_vgg true for a smoke run)' % (FLAGS.vgg19_path,))
        weights = model_vgg.random_vgg_weights(0)
    m = model_enet.EnetModel(FLAGS.model, weights, device=device)
    # source_ckpt_path = tf.train.latest_checkpoint(FLAGS.ckpt_path); restore if there is one (experiment_train.py:96-110)
    source = None
    if FLAGS.ckpt_path and os.path.isdir(FLAGS.ckpt_path):
        from .. import tf_bundle
        source = tf_bundle.latest_checkpoint(FLAGS.ckpt_path, scan=True)     # (state file first, else the newest complete bundle)
    if source is not None:
        m.load_tf_checkpoint(source)
    if world > 1:
        srx_dist.attach_flat(m, world)
    per_rank = FLAGS.batch_size // world
    if FLAGS.train_dir_path and os.path.isdir(FLAGS.train_dir_path):
        # the reference's data path (enet/enet/datasets.py:79-127): crops on the host, both resizes and the float map on
        # the GPU, byte for byte what scipy.misc.imresize gives (each rank walks the directory with its own seed)
        from . import datasets
        batches = datasets.image_batches(FLAGS.train_dir_path, 4, per_rank, device, rng=np.random.RandomState(1234 + rank))
    elif FLAGS.train_dir_path:
        batches = npz_batches(FLAGS.train_dir_path, per_rank, device, seed=rank)
    else:
        batches = synthetic_batches(per_rank, device, seed=rank)
    while True:
        step = m.global_step
        if FLAGS.stop_training_at_k_step is not None and step >= FLAGS.stop_training_at_k_step:
            break
        if step % FLAGS.save_every == FLAGS.save_every - 1 and FLAGS.ckpt_path and rank == 0:
            # saver.save(session, ckpt_path/model.ckpt, global_step=step) when step % 1000 == 999 (:116-117)
            os.makedirs(FLAGS.ckpt_path, exist_ok=True)
            m.save_tf_checkpoint(os.path.join(FLAGS.ckpt_path, 'model.ckpt-%d' % step))
        # NOTE: train discriminator (experiment_train.py:111-131)
        if step % 3 == 0 and m.discriminator is not None:
            a_loss = m.d_step(*next(batches))
            if log is not None:
                log({'step': step, 'trainer': 'd', 'a_loss': a_loss.item()})
        # NOTE: train generator (:133-160)
        losses = m.g_step(*next(batches))
        if log is not None:
            log(dict({'step': step, 'trainer': 'g'}, **{k: v.item() for k, v in losses.items() if k != 'a_loss'}))
        if rank == 0 and (step + 1) % 100 == 0:
            print(json.dumps({'step': step + 1, 'g_loss_all': losses['g_loss_all'].item()}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return m


if __name__ == '__main__':
    main()
