#!/usr/bin/env python3
"""HIP-event timing of the sub-pixel map at the north-star bandwidth shape
[256,41,41,27] <-> [256,123,123,3] with 8 rotating buffer pairs (744 MB > 256 MiB Infinity Cache), beside two plain
copies of the same bytes: the library's own streaming copy (srx_stream_copy: nontemporal 16-byte loads / stores,
persistent workgroups -- the ceiling of a byte-moving kernel at this transfer size) and torch's copy_.

  python scripts/time_d2s.py            one line for the current environment
  python scripts/time_d2s.py sweep      the tuning knobs, one fresh process each (they are read once per process)
  python scripts/time_d2s.py shapes     other shapes of the map (both directions), current environment
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWEEP = [{}, {'SRX_SUBPIXEL_THROTTLE': '0'}, {'SRX_SUBPIXEL_THROTTLE': '2'}, {'SRX_SUBPIXEL_THROTTLE': '3'}, {'SRX_SUBPIXEL_THROTTLE': '5'},
         {'SRX_SUBPIXEL_THROTTLE': '6'}, {'SRX_SUBPIXEL_THROTTLE': '8'}]


def one():
    import torch
    from ml_super_resolution_amd import ops
    dev = torch.device('cuda')
    P = 8
    bufs = [torch.rand((256, 41, 41, 27), device=dev) for _ in range(P)]
    outs = [torch.empty((256, 123, 123, 3), device=dev) for _ in range(P)]

    def run(fn, iters=80):
        for i in range(2 * P): fn(i)
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters): fn(i)
            e.record(); e.synchronize()
            best = min(best, s.elapsed_time(e) / iters * 1e3)
        return best
    t1 = run(lambda i: ops.depth_to_space(bufs[i % P], 3, out=outs[i % P]))
    t2 = run(lambda i: ops.space_to_depth(outs[i % P], 3, out=bufs[i % P]))
    t4 = run(lambda i: ops.stream_copy(bufs[i % P], outs[i % P]))
    t3 = run(lambda i: outs[i % P].view(-1).copy_(bufs[i % P].view(-1)))
    by = 2 * 256 * 41 * 41 * 27 * 4
    knobs = ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('SRX_SUBPIXEL'))
    print('%-40s d2s %.2f us %.2f TB/s (%.1f%% of 8 TB/s) | s2d %.2f us %.2f TB/s | srx_stream_copy %.2f us %.2f TB/s | torch copy_ %.2f us %.2f TB/s'
          % (knobs or 'defaults', t1, by / t1 / 1e6, 100 * by / t1 / 1e6 / 8, t2, by / t2 / 1e6, t4, by / t4 / 1e6, t3, by / t3 / 1e6), flush=True)


SHAPES = [(256, 41, 41, 3, 3), (256, 17, 17, 3, 3), (64, 85, 85, 3, 2), (256, 41, 41, 3, 4), (128, 64, 64, 3, 3), (1024, 41, 41, 3, 3),
          (16, 360, 640, 3, 3), (32, 128, 128, 1, 3), (512, 24, 24, 3, 2), (8, 135, 240, 3, 4)]


def shapes():
    import torch
    from ml_super_resolution_amd import ops
    knobs = ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('SRX_SUBPIXEL')) or 'defaults'
    for n, h, w, c, r in SHAPES:
        by = 2 * n * h * w * c * r * r * 4
        P = max(2, min(8, int(800e6 // by)))
        bufs = [torch.rand((n, h, w, c * r * r), device='cuda') for _ in range(P)]
        outs = [torch.empty((n, h * r, w * r, c), device='cuda') for _ in range(P)]

        def run(fn, iters=60):
            for i in range(2 * P): fn(i)
            best = 1e9
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for i in range(iters): fn(i)
                e.record(); e.synchronize()
                best = min(best, s.elapsed_time(e) / iters * 1e3)
            return best
        t1 = run(lambda i: ops.depth_to_space(bufs[i % P], r, out=outs[i % P]))
        t2 = run(lambda i: ops.space_to_depth(outs[i % P], r, out=bufs[i % P]))
        t4 = run(lambda i: ops.stream_copy(bufs[i % P], outs[i % P]))
        print('%-28s [%d,%d,%d,%d] r%d %6.1f MB  d2s %7.2f us %.2f TB/s | s2d %7.2f us %.2f TB/s | copy %7.2f us'
              % (knobs, n, h, w, c * r * r, r, by / 1e6, t1, by / t1 / 1e6, t2, by / t2 / 1e6, t4), flush=True)
        del bufs, outs


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'shapes':
        shapes()
    elif len(sys.argv) > 1 and sys.argv[1] == 'sweep':
        for env in SWEEP:
            subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, **env))
    else:
        one()
