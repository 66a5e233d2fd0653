#!/usr/bin/env python3
"""
Generates the committed fixtures under tests/golden/.  Run in the BUILD container
(it reads /root/reference/assets for the weight-independent pins P2/P3/P4; nothing
under tests/ reads /root/reference at test time).

  python tests/golden/make_golden.py

Outputs (all data, no reference source):
  pins.json        P2: sha256 + min pixel of vdsr-fig2-conv.N / relu.N (N=1,5,19)
                   P4: size of assets/srcnn_000.jpg
  pin_p3_crop.npz  P3: a 96x96 crop of vdsr-fig2-{sd_image,conv.20,sr_image}.png (uint8)
  ops.npz          per-op vectors from the float64 NumPy oracle (seeded inputs stored too)
  nets.npz         whole-net vectors: VDSR-20 fwd+loss+grads+Adam step at [2,41,41,3],
                   ESPCN fwd (+d2s) at [2,17,17,3] r=3,4, SRCNN fwd at [1,33,33,3]
  d2s_maps.npz     exhaustive integer index maps for r in {2,3,4}, non-square

The conv arithmetic in ops.npz / nets.npz comes from the repo's own oracle (TensorFlow
is unavailable): these vectors freeze the oracle, they do not pin it to the reference
("parity unpinned", see oracle/oracle.py).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

ASSETS = '/root/reference/assets'

# (name, k, cin, cout, padding, act, H, W)  -- every (k, Cin, Cout, pad, act) of SURVEY 2's op inventory
OP_CASES = [
    ('vdsr_first', 3, 3, 64, 'SAME', 'relu', 7, 5),
    ('vdsr_mid', 3, 64, 64, 'SAME', 'relu', 7, 5),
    ('vdsr_last', 3, 64, 3, 'SAME', None, 7, 5),
    ('espcn_f1', 5, 3, 64, 'SAME', 'tanh', 7, 5),
    ('espcn_f2', 3, 64, 32, 'SAME', 'tanh', 7, 5),
    ('espcn_f3_r3', 3, 32, 27, 'SAME', None, 7, 5),
    ('espcn_f3_r4', 3, 32, 48, 'SAME', None, 7, 5),
    ('srcnn_f1', 9, 3, 64, 'VALID', 'relu', 13, 11),
    ('srcnn_f2', 1, 64, 32, 'VALID', 'relu', 7, 5),
    ('srcnn_f3', 5, 32, 3, 'VALID', 'tanh', 9, 8),
    ('enet_1x1', 1, 64, 64, 'SAME', None, 7, 5),
    ('enet_lrelu', 3, 3, 32, 'SAME', 'lrelu', 7, 5),
]


def sha256(path):
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def make_pins():
    from PIL import Image
    pins = {'P2': {}, 'P4': {}}
    for n in (1, 5, 19):
        a = os.path.join(ASSETS, 'vdsr-fig2-conv.%d.png' % n)
        b = os.path.join(ASSETS, 'vdsr-fig2-relu.%d.png' % n)
        pins['P2'][str(n)] = {
            'conv_sha256': sha256(a), 'relu_sha256': sha256(b),
            'min_pixel': int(np.asarray(Image.open(a)).min()),
        }
    im = Image.open(os.path.join(ASSETS, 'srcnn_000.jpg'))
    pins['P4'] = {'width': im.size[0], 'height': im.size[1]}
    with open(os.path.join(HERE, 'pins.json'), 'w') as f:
        json.dump(pins, f, indent=1, sort_keys=True)
    crops = {}
    for name in ('sd_image', 'conv.20', 'sr_image'):
        arr = np.asarray(Image.open(os.path.join(ASSETS, 'vdsr-fig2-%s.png' % name)).convert('RGB'))
        crops[name.replace('.', '_')] = arr[80:176, 80:176].copy()
    np.savez_compressed(os.path.join(HERE, 'pin_p3_crop.npz'), **crops)
    make_pin_p5()
    make_pin_p6()


def make_pin_p6():
    """P6: assets/srcnn_000.jpg and srcnn_001.jpg are the panels hd | sd | sr (231x231 each) that srcnn/srcnn.py:169-184,263-278
    writes: sd = resize_bicubic(resize_bicubic(hd_crop, 81), 243) cropped like hd (:89-93,132-136).  The decoded pixels
    of the hd and sd panels are data the reference holds about TensorFlow's bicubic resize; the fixture keeps them
    (both images, uint8)."""
    from PIL import Image
    out = {}
    for j in (0, 1):
        im = np.asarray(Image.open(os.path.join(ASSETS, 'srcnn_%03d.jpg' % j)).convert('RGB'))
        out['hd%d' % j] = im[:, :231].copy()
        out['sd%d' % j] = im[:, 231:462].copy()
    np.savez_compressed(os.path.join(HERE, 'pin_p6_srcnn_panels.npz'), **out)


P5_CROP = (60, 108, 70, 134)     # rows y0:y1, columns x0:x1 of the 224x224 source (eagle's head)


def make_pin_p5():
    """P5: the only end-to-end BYTE pin the reference offers.  assets/enet_eagle_bq.png is the `_bq.png` the
    reference's enet/enet/experiment_resolve.py:61-147 wrote for assets/enet_eagle.png: scipy.misc.imresize(image,
    400, 'bicubic') -> / 127.5 - 1 -> saturate_cast(x * 127.5 + 127.5, uint8) -> PNG.  PIL's bicubic x4 of the source
    reproduces it with 0 differing bytes (checked here on the whole image; both digests go to pins.json), which pins
    (a) imresize == PIL bicubic and (b) that the float round trip is the identity on bytes (multiply and add rounded
    separately).  The fixture keeps a crop of the source and the matching x4 region of the reference's output."""
    from PIL import Image
    src = Image.open(os.path.join(ASSETS, 'enet_eagle.png')).convert('RGB')
    ref = np.asarray(Image.open(os.path.join(ASSETS, 'enet_eagle_bq.png')).convert('RGB'))
    mine = np.asarray(src.resize((src.width * 4, src.height * 4), Image.BICUBIC))
    y0, y1, x0, x1 = P5_CROP
    np.savez_compressed(os.path.join(HERE, 'pin_p5_eagle_crop.npz'),
                        source=np.asarray(src)[y0:y1, x0:x1].copy(),
                        reference_bq=ref[4 * y0:4 * y1, 4 * x0:4 * x1].copy(), crop=np.asarray(P5_CROP))
    path = os.path.join(HERE, 'pins.json')
    pins = json.load(open(path))
    pins['P5'] = {'source': 'assets/enet_eagle.png', 'reference_output': 'assets/enet_eagle_bq.png',
                  'source_size': list(src.size), 'output_shape': list(ref.shape),
                  'reference_bq_pixels_sha256': hashlib.sha256(ref.tobytes()).hexdigest(),
                  'pil_bicubic_x4_pixels_sha256': hashlib.sha256(mine.tobytes()).hexdigest(),
                  'differing_bytes_whole_image': int((ref != mine).sum())}
    with open(path, 'w') as f:
        json.dump(pins, f, indent=1, sort_keys=True)


def make_ops():
    out = {}
    for idx, (name, k, cin, cout, pad, act, H, W) in enumerate(OP_CASES):
        rng = np.random.default_rng(1000 + idx)
        x = rng.uniform(-1, 1, (2, H, W, cin)).astype(np.float32)
        w = rng.normal(0, 1.0 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, (cout,)).astype(np.float32)
        y = O.conv2d_fwd(x, w, b, pad, act)
        dy = rng.normal(0, 1, y.shape).astype(np.float32)
        dpre = dy * O.act_grad_from_y(y, act)
        dx = O.conv2d_bwd_data(dpre, w, (H, W), pad)
        dw, db = O.conv2d_bwd_filter(x, dpre, (k, k), pad)
        for key, val in (('x', x), ('w', w), ('b', b), ('y', y), ('dy', dy), ('dx', dx),
                         ('dw', dw), ('db', db)):
            out['%s.%s' % (name, key)] = np.asarray(val, np.float32)
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)


def vdsr_params(seed, num_layers=20, bias_scale=0.1):
    rng = np.random.default_rng(seed)
    params = []
    for ks, bs in O.vdsr_param_shapes(num_layers):
        params.append((O.xavier_uniform(rng, ks), rng.uniform(-bias_scale, bias_scale, bs).astype(np.float32)))
    return params


def espcn_params(seed, r):
    rng = np.random.default_rng(seed)
    shapes = [(5, 5, 3, 64), (3, 3, 64, 32), (3, 3, 32, 3 * r * r)]
    return [(O.truncated_normal(rng, s, 0.02 * 5), rng.uniform(-0.1, 0.1, (s[-1],)).astype(np.float32))
            for s in shapes]


def srcnn_params(seed):
    rng = np.random.default_rng(seed)
    shapes = [(9, 9, 3, 64), (1, 1, 64, 32), (5, 5, 32, 3)]
    return [(rng.normal(0, 1.0 / np.sqrt(s[0] * s[1] * s[2]), s).astype(np.float32),
             rng.uniform(-0.1, 0.1, (s[-1],)).astype(np.float32)) for s in shapes]


def make_nets():
    out = {}
    # --- VDSR-20 at [2,41,41,3]: weights are regenerated from the seed by the tests
    rng = np.random.default_rng(104)
    hd = rng.uniform(-1, 1, (2, 41, 41, 3)).astype(np.float32)
    sd = np.clip(hd + 0.1 * np.random.default_rng(105).normal(0, 1, hd.shape), -1, 1).astype(np.float32)
    params = vdsr_params(106)
    loss, grads, fwd = O.vdsr_loss_and_grads(sd, hd, params)
    out['vdsr.sd'] = sd
    out['vdsr.hd'] = hd
    out['vdsr.sr'] = fwd['sr_images'].astype(np.float32)
    out['vdsr.conv_1'] = fwd['conv.1'].astype(np.float32)[:, :8, :8]
    out['vdsr.conv_10'] = fwd['conv.10'].astype(np.float32)[:, :8, :8]
    out['vdsr.conv_19'] = fwd['conv.19'].astype(np.float32)[:, :8, :8]
    out['vdsr.loss'] = np.float64(loss)
    for i in (0, 1, 9, 18, 19):
        out['vdsr.dk_%d' % i] = grads[i][0].astype(np.float32)
    out['vdsr.db_all'] = np.concatenate([g[1].ravel() for g in grads]).astype(np.float32)
    out['vdsr.dk_sums'] = np.array([g[0].sum() for g in grads], np.float64)
    out['vdsr.dk_abs_sums'] = np.array([np.abs(g[0]).sum() for g in grads], np.float64)
    # one TF-Adam step (t=1, lr 5e-5 -- vdsr/makefile:27) on every tensor
    new_sums = []
    for (k, b), (dk, db) in zip(params, grads):
        nk, _, _ = O.adam_tf(k.astype(np.float64), dk, 0.0, 0.0, 5e-5, 1)
        nb, _, _ = O.adam_tf(b.astype(np.float64), db, 0.0, 0.0, 5e-5, 1)
        new_sums.append([nk.sum(), nb.sum()])
    out['vdsr.adam_sums'] = np.array(new_sums, np.float64)
    k0 = params[0][0].astype(np.float64)
    out['vdsr.adam_k0'] = O.adam_tf(k0, grads[0][0], 0.0, 0.0, 5e-5, 1)[0].astype(np.float32)

    # --- ESPCN at [2,17,17,3], r = 3, 4
    for r in (3, 4):
        lr = np.random.default_rng(102).uniform(-1, 1, (2, 17, 17, 3)).astype(np.float32)
        p = espcn_params(103 + r, r)
        y = O.espcn_forward(lr, p)
        out['espcn%d.lr' % r] = lr
        out['espcn%d.y' % r] = y.astype(np.float32)
        out['espcn%d.d2s' % r] = O.depth_to_space(y.astype(np.float32), r)

    # --- SRCNN at [1,33,33,3] -> [1,21,21,3]
    lo = np.random.default_rng(101).uniform(-1, 1, (1, 33, 33, 3)).astype(np.float32)
    y = O.srcnn_forward(lo, srcnn_params(107))
    out['srcnn.lo'] = lo
    out['srcnn.y'] = y.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'nets.npz'), **out)


def enet_pat_case(seed=21, vgg_width=8, d_width=32, size=64, units=32, n=2):
    """Seeded inputs and weights of the committed EnhanceNet-PAT fixture (regenerated from the seed by the tests):
    VGG-19-shaped net of width 8, discriminator of the reference's widths on 64x64 images."""
    from oracle import oracle_enet as E
    rng = np.random.default_rng(seed)
    vgg = {}
    for name, (cin, cout) in E.vgg19_channels(vgg_width).items():
        k = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (3, 3, cin, cout)).astype(np.float32)
        if name == 'block1_conv1':
            k /= np.float32(60.0)
        vgg[name] = (k, rng.normal(0, 0.05, cout).astype(np.float32))
    convs, cin = [], 3
    for i in range(5):
        f = d_width * 2 ** i
        for _ in range(2):
            convs.append((rng.normal(0, np.sqrt(1.5 / (9 * cin)), (3, 3, cin, f)).astype(np.float32),
                          rng.normal(0, 0.05, f).astype(np.float32)))
            cin = f
    feat = (size // 32) ** 2 * cin
    dense = [(rng.normal(0, np.sqrt(1.0 / feat), (feat, units)).astype(np.float32), rng.normal(0, 0.05, units).astype(np.float32)),
             (rng.normal(0, np.sqrt(1.0 / units), (units, 1)).astype(np.float32), rng.normal(0, 0.05, 1).astype(np.float32))]
    hd = rng.uniform(-1, 1, (n, size, size, 3)).astype(np.float32)
    sr = np.clip(hd + rng.normal(0, 0.2, hd.shape), -1, 1).astype(np.float32)
    return vgg, convs, dense, sr, hd


def make_enet_pat():
    """enet_pat.npz: the CONTINUOUS quantities of EnhanceNet-PAT's loss side from oracle/oracle_enet.py (VGG-19
    features, discriminator outputs, the five losses of build_enet) for enet_pat_case().  Gradients are compared
    against the oracle at run time (they are discontinuous in the activations: see tests/test_gpu_enet_pat.py)."""
    from oracle import oracle_enet as E
    vgg, convs, dense, sr, hd = enet_pat_case()
    feats = E.vgg19_forward(sr, vgg)
    losses, _ = E.enet_losses_and_sr_gradient(sr, hd, vgg, convs, dense, 'pat')
    out = {'sr': sr, 'hd': hd}
    for name in ('block1_conv1', 'block2_pool', 'block3_conv1', 'block5_pool'):
        out['vgg.' + name] = feats[name].astype(np.float32)
    out['d.fake'] = E.discriminator_forward(sr, convs, dense).astype(np.float32)
    out['d.real'] = E.discriminator_forward(hd, convs, dense).astype(np.float32)
    for k, v in losses.items():
        out['loss.' + k] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, 'enet_pat.npz'), **out)


def make_d2s_maps():
    out = {}
    for r in (2, 3, 4):
        N, H, W, C = 2, 5, 7, 3
        src = np.arange(N * H * W * C * r * r, dtype=np.int64).reshape(N, H, W, C * r * r)
        out['r%d.d2s' % r] = O.depth_to_space(src, r)
        hr = np.arange(N * H * r * W * r * C, dtype=np.int64).reshape(N, H * r, W * r, C)
        out['r%d.s2d' % r] = O.space_to_depth(hr, r)
    np.savez_compressed(os.path.join(HERE, 'd2s_maps.npz'), **out)


if __name__ == '__main__':
    make_pins()
    make_enet_pat()
    make_ops()
    make_nets()
    make_d2s_maps()
    for f in sorted(os.listdir(HERE)):
        print('%8d  %s' % (os.path.getsize(os.path.join(HERE, f)), f))
