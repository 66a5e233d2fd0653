#!/bin/bash
# Round 4: the parity tests that exercise this round's kernels, once under each A/B switch (the fallback behind every switch
# must still match the oracle).  Run on the GPU box from the repo root; one pytest process at a time.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
T="tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_vdsr.py tests/test_gpu_wgrad_strip.py tests/test_gpu_scripts.py tests/test_gpu_subpixel_fused.py"
for kv in SRX_WGRAD_PIPE_STRIP=0 SRX_WGRAD_ROWS_FULL=0 SRX_KWROWS_MIN_PIXELS=-1 SRX_BIG_ROUTE_MIN_PIXELS=-1 SRX_PIPE=0 SRX_WGRAD_1X1=0 SRX_WGRAD_PACK3=0 SRX_STEP_GRAPH=0 SRX_VDSR_STEP_GRAPH=1 SRX_WGRAD_NT=0 SRX_ESPCN_FUSED_MAX_PIXELS=45000 SRX_STRIP_D2S=0 SRX_ESPCN_FUSED_TRAIN=0 SRX_CONV_1X1_MIN_PIXELS=-1 SRX_PACK3_DGRAD=0 SRX_POISON_LDS=1; do
    echo "== $kv"
    env $kv timeout -k 10 600 python3 -m pytest $T -x -q 2>&1 | tail -2
done
