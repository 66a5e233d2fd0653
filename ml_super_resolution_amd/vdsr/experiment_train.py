"""
experiment_train.py -- mirror of vdsr/vdsr/experiment_train.py: same flags and defaults (:157-219),
step-wise learning-rate decay lr0 * factor ** (step // decay_steps) (:130), stop at
stop_training_at_k_step with one checkpoint (:126-128), resume from the latest checkpoint if one
exists (:108,118-121).  Differences that make it an MI355X program: batches are device resident, the
step is one fused sequence of HIP kernels, and with WORLD_SIZE > 1 (torchrun) the batch is sharded
across GPUs with one RCCL all-reduce of the flat gradient per step.

  python -m ml_super_resolution_amd.vdsr.experiment_train --ckpt_path ckpt --stop_training_at_k_step 100
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m \
         ml_super_resolution_amd.vdsr.experiment_train --batch_size 2048 ...
"""
import argparse
import glob
import json
import os
import time

import torch

from .. import dist as srx_dist
from .. import ops
from . import dataset, model_vdsr


def str2bool(v):
    return str(v).lower() in ('1', 'true', 'yes')


def parse_flags(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--data_path', default=None)
    ap.add_argument('--ckpt_path', default=None)
    ap.add_argument('--logs_path', default=None)
    ap.add_argument('--scaling_factors', default='2_3_4')
    ap.add_argument('--image_size', type=int, default=41)
    ap.add_argument('--batch_size', type=int, default=64)
    ap.add_argument('--num_layers', type=int, default=20)
    ap.add_argument('--learning_rate_decay_steps', type=int, default=2560)
    ap.add_argument('--learning_rate_decay_factor', type=float, default=0.1)
    ap.add_argument('--initial_learning_rate', type=float, default=0.1)
    ap.add_argument('--stop_training_at_k_step', type=int, default=12800)
    # tf.app.flags booleans accept a bare `--use_adam` (vdsr/makefile:26) as well as `--use_adam=false`
    ap.add_argument('--use_adam', type=str2bool, nargs='?', const=True, default=True)
    return ap.parse_args(argv)


def latest_checkpoint(ckpt_path):
    """tf.train.latest_checkpoint(ckpt_path) (:108): the `checkpoint` state file if there is one, else the
    highest-numbered bundle in the directory."""
    if not ckpt_path or not os.path.isdir(ckpt_path):
        return None
    from .. import tf_bundle
    state = tf_bundle.latest_checkpoint(ckpt_path)
    if state is not None:
        return state
    # TensorFlow V2 checkpoints (model.ckpt-N.index + .data-*: what the reference's Saver writes and what this
    # script writes too) and, for older runs of this script, torch state dicts (model.ckpt-N.pt)
    found = {}
    for p in glob.glob(os.path.join(ckpt_path, 'model.ckpt-*.index')):
        found[int(p[:-6].rsplit('-', 1)[1])] = p[:-6]
    for p in glob.glob(os.path.join(ckpt_path, 'model.ckpt-*.pt')):
        found.setdefault(int(p.rsplit('-', 1)[1][:-3]), p)
    return found[max(found)] if found else None


def main(argv=None, log=None):
    """`log`: optional callable receiving one dict per step (tests); the reference writes TF summaries."""
    FLAGS = parse_flags(argv)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        srx_dist.init_process_group(rank, world, local_rank)
    if FLAGS.batch_size % world:
        raise SystemExit('batch_size must be divisible by the number of GPUs')
    per_rank = FLAGS.batch_size // world

    model = model_vdsr.VdsrModel(FLAGS.num_layers, FLAGS.use_adam, device=device)
    source = latest_checkpoint(FLAGS.ckpt_path)
    if source is not None:
        model.stack.load_checkpoint(source)
    if world > 1:
        srx_dist.attach(model.stack, world)

    if FLAGS.data_path and os.path.isdir(FLAGS.data_path):
        # the reference's data source: a directory of images; decode on the host, degrade on the GPU
        from PIL import Image
        import numpy as np
        names = [n for n in sorted(os.listdir(FLAGS.data_path)) if n[-4:].lower() in ('.png', '.jpg', '.bmp', 'jpeg')]
        images = [np.asarray(Image.open(os.path.join(FLAGS.data_path, n)).convert('RGB')) for n in names]
        factors = [float(x) for x in str(FLAGS.scaling_factors).split('_')]
        batches = dataset.image_batches(images, factors, FLAGS.image_size, per_rank, device, seed=rank)
    elif FLAGS.data_path:
        batches = dataset.npz_batches(FLAGS.data_path, per_rank, device, seed=rank)
    else:
        batches = dataset.synthetic_batches(FLAGS.image_size, per_rank, device, seed=104 + 10 * rank)
    if FLAGS.logs_path and rank == 0:
        os.makedirs(FLAGS.logs_path, exist_ok=True)
    logfile = open(os.path.join(FLAGS.logs_path, 'train.jsonl'), 'a') if (FLAGS.logs_path and rank == 0) else None
    t0 = time.time()
    while True:
        step = model.stack.global_step
        if step == FLAGS.stop_training_at_k_step:
            if FLAGS.ckpt_path and rank == 0:
                os.makedirs(FLAGS.ckpt_path, exist_ok=True)
                # the reference: saver.save(session, ckpt_path/model.ckpt, global_step) (experiment_train.py:100-121)
                # plus the graph's other global variable, the non-trainable `learning_rate`
                # (vdsr/vdsr/model_vdsr.py:136-141): Saver().restore needs it
                import numpy as np
                model.stack.save_tf_checkpoint(os.path.join(FLAGS.ckpt_path, 'model.ckpt-%d' % step),
                                               extra={'learning_rate': np.float32(model.learning_rate)})
            break
        lr = FLAGS.initial_learning_rate * (FLAGS.learning_rate_decay_factor ** (step // FLAGS.learning_rate_decay_steps))
        sd_images, hd_images = next(batches)
        loss = model.train_step(sd_images, hd_images, lr)
        if log is not None:
            log({'step': step + 1, 'lr': lr, 'loss': loss.item()})
        if (step + 1) % 100 == 0 and rank == 0:
            # the reference's "epoch" summary: loss + mean PSNR(max_val 2.0) of sr vs hd (:80-84)
            psnr = ops.psnr(model.stack.acts[-1], hd_images, 2.0).mean().item()
            rec = {'step': step + 1, 'loss': loss.item(), 'psnr': psnr, 'lr': lr, 'elapsed_s': time.time() - t0}
            print(json.dumps(rec), flush=True)
            if logfile:
                logfile.write(json.dumps(rec) + '\n')
                logfile.flush()
    if world > 1:
        torch.distributed.destroy_process_group()
    return model


if __name__ == '__main__':
    main()
