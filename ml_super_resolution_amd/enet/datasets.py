"""
datasets.py -- mirror of enet/enet/datasets.py (reference): training batches for EnhanceNet from a directory of images.

`image_batches(source_dir_path, scale_factor, batch_size)` (:79-127) yields (sd_images, bq_images, hd_images): per image
a random 128x128 crop (x, y = np.random.randint(128): the images are at least 255 pixels on a side), `sd =
scipy.misc.imresize(hd, 25)` (Pillow BILINEAR, antialiased, to 32x32), `bq = scipy.misc.imresize(sd, 400, 'bicubic')`
(Pillow BICUBIC, back to 128x128), all three as float32 / 127.5 - 1.  Here the file decode and the crop stay on the host
(Pillow), the crops go to the GPU as uint8 and both resizes and the float conversion run there
(`ops.resize_pil_u8`, `ops.u8_to_pm1`): the same bytes as the reference's CPU path (pinned by assets/enet_eagle_bq.png,
tests/test_oracle_pins.py), without the per-image Python resizes on the training loop's critical path.
(`build_image_batch_iterator`, the unused tf.data variant of :33-76, is not mirrored.)  File decoding runs on a small
thread pool a couple of batches ahead of the consumer.
"""
import os

import numpy as np
import torch

from .. import ops


def is_image_name(name):
    return os.path.splitext(name)[1].lower() in ('.png', '.jpg', '.jpeg')


def build_path_generator(dir_path, rng=None):
    """Endless generator over the directory's image paths, reshuffled every pass (:10-30)."""
    rng = rng if rng is not None else np.random
    names = [n for n in sorted(os.listdir(dir_path)) if is_image_name(n)]
    if not names:
        raise ValueError('no .png / .jpg / .jpeg images in %s' % dir_path)

    def paths_generator():
        while True:
            rng.shuffle(names)
            for name in names:
                yield os.path.join(dir_path, name)
    return paths_generator


def degrade_on_device(hd_u8):
    """hd_u8 [N,128,128,3] uint8 on the GPU -> (sd, bq, hd) float32 in [-1, 1]: sd [N,32,32,3], bq and hd [N,128,128,3]."""
    n, h, w, _ = hd_u8.shape
    sd_u8 = ops.resize_pil_u8(hd_u8, h // 4, w // 4, 'bilinear')          # imresize(hd, 25): default interp 'bilinear'
    bq_u8 = ops.resize_pil_u8(sd_u8, (h // 4) * 4, (w // 4) * 4, 'bicubic')   # imresize(sd, 400, 'bicubic')
    return ops.u8_to_pm1(sd_u8), ops.u8_to_pm1(bq_u8), ops.u8_to_pm1(hd_u8)


def _decode_crop(path, x, y):
    from PIL import Image
    hd = np.asarray(Image.open(path).convert('RGB'))
    crop = hd[y:y + 128, x:x + 128, :]
    if crop.shape != (128, 128, 3):
        raise ValueError('%s is smaller than 255 pixels on a side: crop %s' % (path, crop.shape))
    return crop


def image_batches(source_dir_path, scale_factor=4, batch_size=32, device='cuda', rng=None, workers=8, prefetch=2):
    """(sd_images, bq_images, hd_images) device tensors, forever.  scale_factor is accepted and ignored, as in the
    reference (:79: the 25 % / 400 % are literals).
    The reference decodes its batch_size files one after the other inside the training loop (a 256x256 PNG costs a
    millisecond or two: 64 of them are longer than a training step on this GPU).  Here the random numbers are drawn in
    the reference's order on the calling thread, the decode + crop jobs they define run on `workers` threads (Pillow
    releases the GIL) and `prefetch` batches are kept in flight: the same bytes, off the critical path."""
    import collections
    from concurrent.futures import ThreadPoolExecutor
    rng = rng if rng is not None else np.random
    paths = build_path_generator(source_dir_path, rng)()
    pool = ThreadPoolExecutor(max_workers=max(1, workers))

    def submit_batch():
        jobs = []
        for _ in range(batch_size):
            path = next(paths)
            x, y = rng.randint(128), rng.randint(128)          # (:104-105, in this order)
            jobs.append(pool.submit(_decode_crop, path, x, y))
        return jobs
    pending = collections.deque(submit_batch() for _ in range(max(1, prefetch)))
    try:
        while True:
            jobs = pending.popleft()
            pending.append(submit_batch())
            crops = np.stack([j.result() for j in jobs], axis=0)
            yield degrade_on_device(torch.from_numpy(crops).to(device))
    finally:
        pool.shutdown(wait=False, cancel_futures=True)
