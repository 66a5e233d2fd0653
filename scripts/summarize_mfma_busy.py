#!/usr/bin/env python3
"""profiles/<round>_mfma_busy.csv from the counter passes of scripts/prof_r03_counters.sh / prof_r04.sh (gpurun_out/<round>_{conv,wide,strip}_SQ_VALU_MFMA_BUSY_CYCLES+...csv).
Usage: summarize_mfma_busy.py [round prefix, default r03]

MFMA utilisation from COUNTERS, not from time / FLOPs:
  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
                   busy cycles of the matrix pipes summed over the chip's 1024 SIMDs, over kernel duration in shader cycles
                   (rocprofv3 reports GRBM_GUI_ACTIVE as the sum over the 8 XCDs) x 1024 SIMDs.  Clock-independent: the
                   time-based fractions of bench.py divide by the 2.4 GHz peak, these divide by the cycles that elapsed.
  useful_util    = the same with the algorithmic MFMA cycles (FLOPs / 2048 per 16x16x4 MFMA x 32 cycles) where the launch's
                   FLOPs are known (the VDSR body layer: 73,728 FLOP/px x 430,336 px; the 512x512 strip layer: x 4 x 262,144 px)
  util_resident  = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): busy share while the CU holds waves (no ramp / tail)
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md, per-instruction constants); SQ_BUSY_CU_CYCLES counts quad-cycles."""
import csv, glob, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 else 'r03'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN = {  # kernel-name fragment -> algorithmic FLOPs per launch of the profiled script
    ('conv', 'conv_pipe_kernel<3, 3, 64, 4, false, 0>'): 73728.0 * 430336, ('conv', 'conv_pipe_kernel<3, 3, 64, 4, true, 1>'): 73728.0 * 430336,
    ('conv', 'wgrad_pipe_kernel<3, 3, 64, 4>'): 73728.0 * 430336, ('conv', 'wgrad_rows_full_kernel<3, 3, 64, 4, 41>'): 73728.0 * 430336,
    ('strip', 'conv_pipe_strip_kernel<3, 3, 64, 4, false, 0>'): 73728.0 * 4 * 512 * 512, ('strip', 'conv_pipe_strip_kernel<3, 3, 64, 4, true, 1>'): 73728.0 * 4 * 512 * 512,
    ('strip', 'wgrad_lin_strip_kernel<3, 3, 64, 4, 2>'): 73728.0 * 4 * 512 * 512,
    ('strip', 'wgrad_rows_strip_kernel<3, 3, 64, 4, true>'): 73728.0 * 4 * 512 * 512, ('strip', 'wgrad_rows_strip_kernel<3, 3, 64, 4, false>'): 73728.0 * 4 * 512 * 512,
}
WHAT = {'conv': 'scripts/prof_conv.py 5 all (VDSR body layer 3x3 64->64 at 256x41x41, back to back)',
        'wide': 'scripts/time_wide.py 4 512 (VGG-19 wide layers, 4 x 512^2; averages over the six layer shapes)',
        'espcn_image': 'scripts/time_espcn_image.py (ESPCN 3x on whole images, 128^2 ... 720 x 1280 LR: each kernel averaged over all the sizes it ran at)',
        'strip': 'scripts/time_layer.py 4 512 512 (3x3 64->64 on 4 x 512^2: column strips)' if RND == 'r03' else
                 'scripts/time_layer.py 64 128 128 (3x3 64->64 at the reference recipe\'s 64 x 128^2: column strips; same pixel count as 4 x 512^2)'}
out = [['run', 'kernel', 'dispatches', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_BUSY_CU_CYCLES', 'mfma_util', 'useful_util', 'util_resident', 'what']]
for run in ('conv', 'wide', 'strip', 'espcn_image'):
    fs = glob.glob(os.path.join(ROOT, 'gpurun_out', '%s_%s_SQ_VALU_MFMA_BUSY_CYCLES+*.csv' % (RND, run)))
    if not fs:
        continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        agg.setdefault(r['Kernel_Name'], {})[r['Counter_Name']] = (float(r['Average']), int(r['Dispatches']))
    for k, c in sorted(agg.items()):
        if 'srx::' not in k or c.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0))[0] == 0:
            continue
        busy, n = c['SQ_VALU_MFMA_BUSY_CYCLES']
        gui, cu = c['GRBM_GUI_ACTIVE'][0], c['SQ_BUSY_CU_CYCLES'][0]
        simd_cycles = gui / 8.0 * 1024.0
        useful = ''
        for (rn, frag), flop in KNOWN.items():
            if rn == run and frag in k:
                useful = '%.4f' % (flop / 2048.0 * 32.0 / simd_cycles)
        name = k.replace('void ', '').replace('srx::(anonymous namespace)::', 'srx::').split('(')[0]
        out.append([run, name, n, '%.0f' % busy, '%.0f' % gui, '%.0f' % cu, '%.4f' % (busy / simd_cycles), useful, '%.4f' % (busy / (4.0 * cu)), WHAT[run]])
dst = os.path.join(ROOT, 'profiles', '%s_mfma_busy.csv' % RND)
with open(dst, 'w', newline='') as f:
    csv.writer(f).writerows(out)
for row in out:
    print(' | '.join(str(v) for v in row[:9]))
