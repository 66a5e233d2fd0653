"""
datasets.py -- mirror of enet/enet/datasets.py (reference): training batches for EnhanceNet from a directory of images.

`image_batches(source_dir_path, scale_factor, batch_size)` (:79-127) yields (sd_images, bq_images, hd_images): per image
a random 128x128 crop (x, y = np.random.randint(128): the images are at least 255 pixels on a side), `sd =
scipy.misc.imresize(hd, 25)` (Pillow BILINEAR, antialiased, to 32x32), `bq = scipy.misc.imresize(sd, 400, 'bicubic')`
(Pillow BICUBIC, back to 128x128), all three as float32 / 127.5 - 1.  Here the file decode and the crop stay on the host
(Pillow), the crops go to the GPU as uint8 and both resizes and the float conversion run there
(`ops.resize_pil_u8`, `ops.u8_to_pm1`): the same bytes as the reference's CPU path (pinned by assets/enet_eagle_bq.png,
tests/test_oracle_pins.py), without the per-image Python resizes on the training loop's critical path.
(`build_image_batch_iterator`, the unused tf.data variant of :33-76, is not mirrored.)
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


def image_batches(source_dir_path, scale_factor=4, batch_size=32, device='cuda', rng=None):
    """(sd_images, bq_images, hd_images) device tensors, forever.  scale_factor is accepted and ignored, as in the
    reference (:79: the 25 % / 400 % are literals)."""
    from PIL import Image
    rng = rng if rng is not None else np.random
    paths = build_path_generator(source_dir_path, rng)()
    while True:
        crops = np.empty((batch_size, 128, 128, 3), np.uint8)
        for i in range(batch_size):
            hd = np.asarray(Image.open(next(paths)).convert('RGB'))
            x, y = rng.randint(128), rng.randint(128)
            crop = hd[y:y + 128, x:x + 128, :]
            if crop.shape != (128, 128, 3):
                raise ValueError('image smaller than 255 pixels on a side: crop %s' % (crop.shape,))
            crops[i] = crop
        yield degrade_on_device(torch.from_numpy(crops).to(device))
