"""
experiment_train.py -- mirror of espcn/espcn/experiment_train.py: flags, LR schedule
lr0 * factor ** (step // decay_steps) (:101-107), resume from the latest checkpoint under ckpt_path when
one exists (:78-88), loop until stop_training_at_k_step, one checkpoint at the end (:130) -- a TensorFlow
V2 bundle `model.ckpt-<step>` with the reference's variable names (f1..f3 kernel / bias, Adam slots,
beta powers, global_step) plus the `checkpoint` state file.  The reference reads pre-shuffled TFRecords; here batches come from a
synthetic generator or an .npz of (lr_patches, hr_patches): HR patches are mapped to the sub-pixel
label layout ON THE GPU with space_to_depth (dataset.py:140-156).
"""
import argparse
import os

import numpy as np
import torch

from .. import ops
from . import model_espcn


def synthetic_batches(batch_size, lr_patch_size, scaling_factor, device, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    hp = lr_patch_size * scaling_factor
    while True:
        hr = torch.rand((batch_size, hp, hp, 3), device=device, generator=g) * 2 - 1
        off = scaling_factor // 2
        lr = hr[:, off::scaling_factor, off::scaling_factor].contiguous()
        yield lr, hr


def npz_batches(path, batch_size, device, seed=0):
    z = np.load(path)
    lr_all, hr_all = z['lr_patches'].astype(np.float32), z['hr_patches'].astype(np.float32)
    rng = np.random.default_rng(seed)
    while True:
        idx = rng.integers(0, lr_all.shape[0], size=batch_size)
        yield torch.from_numpy(lr_all[idx]).to(device), torch.from_numpy(hr_all[idx]).to(device)


def parse_flags(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--data_path', default=None)
    ap.add_argument('--ckpt_path', default=None)
    ap.add_argument('--logs_path', default=None)
    ap.add_argument('--batch_size', type=int, default=64)
    ap.add_argument('--scaling_factor', type=int, default=3)
    ap.add_argument('--lr_patch_size', type=int, default=17)
    ap.add_argument('--initial_learning_rate', type=float, default=0.1)
    ap.add_argument('--learning_rate_decay_factor', type=float, default=0.1)
    ap.add_argument('--learning_rate_decay_steps', type=int, default=2560)
    ap.add_argument('--stop_training_at_k_step', type=int, default=10000)
    return ap.parse_args(argv)


def latest_checkpoint(ckpt_path):
    """tf.train.latest_checkpoint(ckpt_path): the state file first, then the highest-numbered bundle."""
    from .. import tf_bundle
    if not ckpt_path or not os.path.isdir(ckpt_path):
        return None
    found = tf_bundle.latest_checkpoint(ckpt_path)
    if found is not None:
        return found
    steps = {}
    for name in os.listdir(ckpt_path):
        if name.startswith('model.ckpt-') and name.endswith('.index'):
            steps[int(name[len('model.ckpt-'):-len('.index')])] = os.path.join(ckpt_path, name[:-len('.index')])
    return steps[max(steps)] if steps else None


def main(argv=None, log=None):
    """`log`: optional callable receiving one dict per step (tests)."""
    FLAGS = parse_flags(argv)
    device = torch.device('cuda')
    m = model_espcn.EspcnModel(FLAGS.scaling_factor, device=device)
    source = latest_checkpoint(FLAGS.ckpt_path)
    if source is not None:
        m.stack.load_tf_checkpoint(source)               # weights, Adam slots, global_step (:78-88)
    batches = (npz_batches(FLAGS.data_path, FLAGS.batch_size, device) if FLAGS.data_path
               else synthetic_batches(FLAGS.batch_size, FLAGS.lr_patch_size, FLAGS.scaling_factor, device))
    step = m.stack.global_step
    while step < FLAGS.stop_training_at_k_step:
        lr_rate = FLAGS.initial_learning_rate * (FLAGS.learning_rate_decay_factor ** (step // FLAGS.learning_rate_decay_steps))
        lr_patch, hr_patch = next(batches)
        hr_target = ops.space_to_depth(hr_patch, FLAGS.scaling_factor)
        loss = m.train_step(lr_patch, hr_target, lr_rate)
        step = m.stack.global_step
        if log is not None:
            log({'step': step, 'lr': lr_rate, 'loss': loss.item()})
        if step % 100 == 0:
            print('step %d loss %.6f lr %g' % (step, loss.item(), lr_rate), flush=True)
    if FLAGS.ckpt_path:
        os.makedirs(FLAGS.ckpt_path, exist_ok=True)
        # saver.save(session, ckpt_path/model.ckpt, global_step=step) (:130)
        m.stack.save_tf_checkpoint(os.path.join(FLAGS.ckpt_path, 'model.ckpt-%d' % step))
    return m


if __name__ == '__main__':
    main()
