#!/bin/bash
# Round-4 final evidence set on the final binary (after wgrad_rows_full_kernel, the SRCNN training kernels, conv_pack3, conv_rows3x3, the
# two-chunk pipelined strips, the 16x16-tile ESPCN kernel and the inlined branch-free tanh): bench kernel stats + line, recipe kernel stats, SRCNN train kernel stats, HBM counters of the
# dominant kernel (traffic.json), MFMA-busy counters of the VDSR body trio, the timing scripts.
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
scripts/prof_stats.sh r04_bench python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-live-traffic
cp gpurun_out/r04_bench.out gpurun_out/r04_bench_line_under_rocprof.json
scripts/prof_stats.sh r04_recipe python3 scripts/time_vdsr_recipe.py
cp gpurun_out/r04_recipe.out gpurun_out/r04_recipe_line_under_rocprof.json
SRX_STEP_GRAPH=0 scripts/prof_stats.sh r04_srcnn_train python3 scripts/time_srcnn_train.py
scripts/prof_pmc.sh r04_conv "$MF" python3 scripts/prof_conv.py 5 all
scripts/prof_pmc.sh r04_prof_conv FETCH_SIZE python3 scripts/prof_conv.py 3 all
scripts/prof_pmc.sh r04_prof_conv WRITE_SIZE python3 scripts/prof_conv.py 3 all
scripts/prof_stats.sh r04_prof_conv python3 scripts/prof_conv.py 10 all
python3 scripts/time_srcnn_image.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_srcnn_image.txt
python3 scripts/time_espcn_image.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_espcn_image.txt
SRX_STEP_GRAPH=0 python3 scripts/time_srcnn_train.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_srcnn_train.txt
python3 scripts/time_vdsr_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_vdsr_batch_eager.txt
python3 scripts/time_layer.py 256 41 41 64 41 41 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_layer_41.txt
SRX_WGRAD_ROWS_FULL=0 python3 scripts/time_layer.py 256 41 41 64 41 41 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_time_layer_41.txt
python3 scripts/time_srcnn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_srcnn.txt
python3 scripts/time_espcn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_espcn.txt
scripts/prof_stats.sh r04_espcn_image python3 scripts/time_espcn_image.py
scripts/prof_pmc.sh r04_espcn_image "$MF" python3 scripts/time_espcn_image.py
python3 scripts/time_espcn_train.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_espcn_train.txt
python3 scripts/time_layer.py 64 128 128 4 512 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_time_layer_strip.txt
python3 bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
