#!/bin/bash
# Round-3 evidence set (run on the GPU box from the repo root; everything lands in gpurun_out/, the summaries are copied
# into profiles/ afterwards on the build side -- scripts/record_traffic.py for traffic.json):
#   kernel stats of the bench run, the un-profiled bench line, HBM counters of the dominant kernels, per-trainer
#   EnhanceNet-PAT kernel stats, the sub-pixel / SRCNN / ESPCN timing scripts
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
scripts/prof_stats.sh r03_bench python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline
cp gpurun_out/r03_bench.out gpurun_out/r03_bench_line_under_rocprof.json
python3 bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
scripts/prof_stats.sh r03_prof_conv python3 scripts/prof_conv.py 10 all
scripts/prof_pmc.sh r03_prof_conv FETCH_SIZE python3 scripts/prof_conv.py 3 all
scripts/prof_pmc.sh r03_prof_conv WRITE_SIZE python3 scripts/prof_conv.py 3 all
scripts/prof_stats.sh r03_pat_g_after python3 scripts/time_enet_pat.py 64 3 128 g
scripts/prof_stats.sh r03_pat_d_after python3 scripts/time_enet_pat.py 64 3 128 d
python3 scripts/time_enet_pat.py 64 5 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_time_enet_pat.txt
python3 scripts/time_enet_pat.py 4 3 512 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_time_enet_pat.txt
python3 scripts/time_d2s.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_time_d2s_final.txt
SRX_SUBPIXEL_EVEN=0 python3 scripts/time_d2s.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_time_d2s_final.txt
python3 scripts/time_d2s.py shapes 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_time_d2s_shapes.txt
SRX_SUBPIXEL_EVEN=0 python3 scripts/time_d2s.py shapes 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_time_d2s_shapes.txt
python3 scripts/time_srcnn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_time_srcnn.txt
python3 scripts/time_espcn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_time_espcn.txt
