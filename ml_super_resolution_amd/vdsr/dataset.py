"""
dataset.py -- batch sources for the VDSR entry points.

The reference's `vdsr/vdsr/dataset.py` builds (sd, hd) pairs on the host with scikit-image
(gaussian blur sigma = 0.5*(s-1), bilinear resize down then up with mode='edge', random 41x41 crop
and flip: dataset.py:13-38,82-128).  scikit-image is not part of this image and that CPU pipeline is
outside the hot path, so this module offers:
  * synthetic_batches: the benchmark's synthetic patches, generated on the GPU;
  * npz_batches: pre-extracted patch pairs ({'sd': [M,h,w,3], 'hd': [M,h,w,3]} in [-1,1]);
  * hd_image_to_sd_image: the same degradation restated with scipy.ndimage (parity with skimage
    unpinned) for the evaluate / resolve entry points.
Every generator yields device tensors: nothing is copied host->device per step.
"""
import numpy as np
import torch


def synthetic_batches(image_size, batch_size, device, seed=104):
    g = torch.Generator(device=device).manual_seed(seed)
    while True:
        hd = torch.rand((batch_size, image_size, image_size, 3), device=device, generator=g) * 2 - 1
        sd = (hd + 0.1 * torch.randn(hd.shape, device=device, generator=g)).clamp(-1, 1)
        yield sd, hd


def npz_batches(path, batch_size, device, seed=0):
    z = np.load(path)
    sd_all, hd_all = z['sd'].astype(np.float32), z['hd'].astype(np.float32)
    rng = np.random.default_rng(seed)
    while True:
        idx = rng.integers(0, sd_all.shape[0], size=batch_size)
        yield torch.from_numpy(sd_all[idx]).to(device), torch.from_numpy(hd_all[idx]).to(device)


def hd_image_to_sd_image(hd_image, scaling_factor):
    """dataset.py:13-38 restated with scipy: blur (nearest borders), bilinear resize down and back up
    (edge mode, no anti-aliasing).  float image in, float image out."""
    from scipy.ndimage import gaussian_filter, map_coordinates
    hd_h, hd_w, _ = hd_image.shape
    sd_h, sd_w = int(hd_h / scaling_factor), int(hd_w / scaling_factor)
    sigma = max(0.0, 0.5 * (scaling_factor - 1.0))
    bl = gaussian_filter(hd_image, sigma=(sigma, sigma, 0), mode='nearest', truncate=4.0) if sigma > 0 else hd_image

    def resize(img, oh, ow):
        ih, iw, c = img.shape
        # skimage.transform.resize (order 1): output pixel centre -> input coordinate, half-pixel mapping
        ys = (np.arange(oh) + 0.5) * ih / oh - 0.5
        xs = (np.arange(ow) + 0.5) * iw / ow - 0.5
        yy, xx = np.meshgrid(ys, xs, indexing='ij')
        out = np.empty((oh, ow, c), img.dtype)
        for ch in range(c):
            out[:, :, ch] = map_coordinates(img[:, :, ch], [yy, xx], order=1, mode='nearest')
        return out

    return resize(resize(bl, sd_h, sd_w), hd_h, hd_w)


def degrade_on_device(hd01, scaling_factor):
    """dataset.py:13-38 on the GPU for a batch [N,H,W,3] of float images in [0,1]: gaussian blur
    sigma = 0.5*(s-1) ('nearest' borders), bilinear resize to (int(H/s), int(W/s)) and back ('edge')."""
    from .. import ops
    N, H, W, C = hd01.shape
    sd_h, sd_w = int(H / scaling_factor), int(W / scaling_factor)
    bl = ops.gaussian_blur(hd01, max(0.0, 0.5 * (scaling_factor - 1.0)))
    return ops.resize_bilinear(ops.resize_bilinear(bl, sd_h, sd_w), H, W)


def image_batches(images_u8, scaling_factors, image_size, batch_size, device, seed=None):
    """The reference's `image_batches` (dataset.py:41-128) with the float work moved to the GPU.
    images_u8: list of decoded uint8 images [h,w,3] (host).  Per patch, as the reference: random crop,
    random horizontal flip (host, uint8 slicing), then ON DEVICE: /255, degrade with a randomly chosen
    scaling factor, map both to [-1,1].  Patches are grouped by scaling factor so that every device op runs
    on a uniform batch.  Yields (sd_images, hd_images) device tensors [batch_size, S, S, 3]."""
    from .. import ops
    if not scaling_factors:
        scaling_factors = [2.0, 3.0, 4.0]
    if any(s <= 1 for s in scaling_factors):
        raise Exception('invalide scaling factors')
    rng = np.random.default_rng(seed)
    images_u8 = [im for im in images_u8 if im.shape[0] >= image_size and im.shape[1] >= image_size and im.shape[2] == 3]
    if not images_u8:
        raise ValueError('no image is at least %dx%d' % (image_size, image_size))
    order = np.arange(len(images_u8))
    pos = len(order)
    while True:
        crops, factors = [], []
        while len(crops) < batch_size:
            if pos == len(order):
                rng.shuffle(order)
                pos = 0
            im = images_u8[order[pos]]
            pos += 1
            h, w, _ = im.shape
            x = rng.integers(0, max(w - image_size, 1))
            y = rng.integers(0, max(h - image_size, 1))
            crop = im[y:y + image_size, x:x + image_size]
            if rng.integers(0, 2) == 1:
                crop = crop[:, ::-1]
            crops.append(crop)
            factors.append(rng.choice(scaling_factors))
        factors = np.asarray(factors)
        idx = np.argsort(factors, kind='stable')
        hd_u8 = torch.from_numpy(np.ascontiguousarray(np.stack(crops)[idx])).to(device)
        hd01 = ops.u8_to_unit_float(hd_u8)
        sd01 = torch.empty_like(hd01)
        f_sorted = factors[idx]
        start = 0
        while start < batch_size:
            end = start
            while end < batch_size and f_sorted[end] == f_sorted[start]:
                end += 1
            sd01[start:end] = degrade_on_device(hd01[start:end].contiguous(), float(f_sorted[start]))
            start = end
        yield ops.affine(sd01, 2.0, -1.0), ops.affine(hd01, 2.0, -1.0)
