"""
make_pin_p7.py -- P7: the reference's own VDSR feature maps pin the convolution of rows A1 / A2.

BUILD CONTAINER ONLY (reads /root/reference/assets, which does not travel).  Output: tests/golden/pin_p7_vdsr_fig2.npz
-- DATA only (fitted numbers and uint8 pixel crops of the reference's PNGs; no reference source text).

What the reference holds: assets/vdsr-fig2-sd_image.png, vdsr-fig2-conv.1.png ... conv.20.png and vdsr-fig2-sr_image.png
are the outputs of ONE run of vdsr/vdsr/model_vdsr.py:47-106 written by
vdsr/vdsr/experiment_feature_map_visualize.py:68-110: every tensor encoded as saturate_cast(x * 127.5 + 127.5, uint8)
(truncation), the 64 maps of a layer laid out as an 8 x 8 mosaic (channel k at tile row k // 8, tile column k % 8), no
per-map normalisation.  The trained weights are not in the repository, but each output channel of a layer has only
3*3*Cin + 1 unknowns (28 for layer 1, 577 for layers 2..20) against ~40-60 thousand pixels, so they are recovered by
least squares from the images themselves:

  * FIT on interior pixels only (never an image-border pixel), OUTSIDE the four held-out corner regions, where the
    target is active and unsaturated (post-ReLU code > 128, < 255; layer 20 has no ReLU: 0 < code < 255);
  * HELD OUT: the four CxC corners of the image.  They contain the image border, where SAME padding acts.

The committed file holds the fitted (W, b) of all 20 layers and the uint8 corner crops of sd, conv.1 .. conv.20 and sr.
tests/test_oracle_pins.py::test_p7_* (CPU, oracle) and tests/test_gpu_pins.py (HIP through the C ABI) then PREDICT
the held-out corners, borders included, from the previous layer's crop: decode -> conv2d SAME + bias + ReLU -> encode.

Run:  python tests/golden/make_pin_p7.py            (about 15 minutes on 8 cores: 64 x 19 least-squares problems of ~50,000 x 577)
      python tests/golden/make_pin_p7.py full20     (seconds: the whole image for the input and the output layer, pin_p7_layer{1,20}_full.npz)
"""
import os
import sys

import numpy as np
from PIL import Image

ASSETS = '/root/reference/assets'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pin_p7_vdsr_fig2.npz')
C = 24                      # side of a held-out corner
S = 256                     # image side
NL = 20


def decode(u8):
    """Midpoint of the code's interval under saturate_cast(x * 127.5 + 127.5) (truncation):
    x * 127.5 + 127.5 in [c, c + 1)  ->  x ~ (c + 0.5 - 127.5) / 127.5 = (c - 127) / 127.5."""
    return (u8.astype(np.float64) - 127.0) / 127.5


def unmosaic(img):
    """[2048, 2048] -> [256, 256, 64]: channel k sits at tile (k // 8, k % 8) (experiment_feature_map_visualize.py:92-99)."""
    return img.reshape(8, S, 8, S).transpose(1, 3, 0, 2).reshape(S, S, 64)


def load():
    maps = {0: np.asarray(Image.open(os.path.join(ASSETS, 'vdsr-fig2-sd_image.png')))}
    for n in range(1, NL):
        maps[n] = unmosaic(np.asarray(Image.open(os.path.join(ASSETS, 'vdsr-fig2-conv.%d.png' % n))))
    maps[NL] = np.asarray(Image.open(os.path.join(ASSETS, 'vdsr-fig2-conv.%d.png' % NL)))
    sr = np.asarray(Image.open(os.path.join(ASSETS, 'vdsr-fig2-sr_image.png')))
    return maps, sr


def patches(x):
    """x [S, S, Cin] -> A [(S-2)*(S-2), 9*Cin + 1] for the interior pixels (rows 1..S-2), tap-major then channel,
    i.e. column (kh*3 + kw)*Cin + ci multiplies W[kh, kw, ci]; last column = 1 (bias)."""
    cin = x.shape[-1]
    cols = [x[kh:kh + S - 2, kw:kw + S - 2, :].reshape(-1, cin) for kh in range(3) for kw in range(3)]
    return np.concatenate(cols + [np.ones(((S - 2) * (S - 2), 1))], axis=1)


def corners(a):
    """The four held-out corners, as a batch [4, C, C, ch]: top-left, top-right, bottom-left, bottom-right."""
    return np.stack([a[:C, :C], a[:C, -C:], a[-C:, :C], a[-C:, -C:]])


def main():
    maps, sr = load()
    held = np.zeros((S, S), bool)
    held[:C, :C] = held[:C, -C:] = held[-C:, :C] = held[-C:, -C:] = True
    held_int = held[1:-1, 1:-1].reshape(-1)
    out = {'sd': corners(maps[0]), 'sr': corners(sr), 'corner': np.int32(C)}
    stats = []
    for n in range(1, NL + 1):
        xin = maps[n - 1]
        x = decode(xin)
        A = patches(x)
        cin = x.shape[-1]
        tgt_u8 = maps[n][1:-1, 1:-1].reshape((S - 2) * (S - 2), -1)
        tgt = decode(tgt_u8)
        # the 3x3 input neighbourhood must be unsaturated too (a saturated input is not the value the layer saw)
        sat = (xin == 255) | (xin == 0) if n > 1 else np.zeros_like(xin, bool)
        sat = sat.any(axis=-1)
        nb = np.zeros((S - 2, S - 2), bool)
        for kh in range(3):
            for kw in range(3):
                nb |= sat[kh:kh + S - 2, kw:kw + S - 2]
        nb = nb.reshape(-1)
        cout = tgt.shape[1]
        W = np.zeros((3, 3, cin, cout))
        b = np.zeros(cout)
        used = []
        for co in range(cout):
            if n < NL:
                ok = (tgt_u8[:, co] > 128) & (tgt_u8[:, co] < 255)
            else:
                ok = (tgt_u8[:, co] > 0) & (tgt_u8[:, co] < 255)
            ok &= ~held_int & ~nb
            used.append(int(ok.sum()))
            if ok.sum() < 4 * A.shape[1]:
                continue                      # a dead (or almost dead) map: leave W = 0, b = 0 and flag it
            Am = A[ok]
            sol, *_ = np.linalg.lstsq(Am, tgt[ok, co], rcond=None)
            W[..., co] = sol[:-1].reshape(3, 3, cin)
            b[co] = sol[-1]
        out['w%d' % n] = W.astype(np.float32)
        out['b%d' % n] = b.astype(np.float32)
        out['n_fit%d' % n] = np.asarray(used, np.int32)
        out['conv%d' % n] = corners(maps[n])
        stats.append((n, min(used), int(np.median(used))))
        print('layer %2d: equations per channel min %d median %d' % stats[-1], flush=True)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


def full_layer20():
    """The whole 256 x 256 image for the output layer: the reference's conv.19 maps, sd, conv.20 and sr as uint8 ->
    pin_p7_layer20_full.npz (1.7 MB).  Used with the (w20, b20) of the fit above: every border pixel of the image (1,020
    of them) was never part of that fit, nor were the four corner regions."""
    maps, sr = load()
    dst = os.path.join(os.path.dirname(OUT), 'pin_p7_layer20_full.npz')
    np.savez_compressed(dst, conv19=maps[19], sd=maps[0], conv20=maps[NL], sr=sr)
    print('wrote', dst, os.path.getsize(dst), 'bytes')
    # ... and for the input layer (3 -> 64): the reference's conv.1 maps of the whole image (sd is in the file above)
    dst = os.path.join(os.path.dirname(OUT), 'pin_p7_layer1_full.npz')
    np.savez_compressed(dst, conv1=maps[1])
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'full20':
        sys.exit(full_layer20())
    sys.exit(main())
