#!/bin/bash
# Round-4 evidence set (run on the GPU box from the repo root; everything lands in gpurun_out/, the summaries are copied into
# profiles/ afterwards on the build side; scripts/record_traffic.py writes traffic.json):
#   * kernel stats of the bench run and of the reference recipe's train step (batch 64 of 128 x 128: column strips);
#   * HBM counters (one pass each) and MFMA-busy counters of the strip trio at the recipe's layer shape;
#   * HBM counters of the dominant kernel of the bench workload (traffic.json);
#   * the timing scripts beside the BASELINE configs.
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
scripts/prof_stats.sh r04_bench python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-live-traffic
cp gpurun_out/r04_bench.out gpurun_out/r04_bench_line_under_rocprof.json
scripts/prof_stats.sh r04_recipe python3 scripts/time_vdsr_recipe.py
cp gpurun_out/r04_recipe.out gpurun_out/r04_recipe_line_under_rocprof.json
scripts/prof_pmc.sh r04_strip FETCH_SIZE python3 scripts/time_layer.py 64 128 128
scripts/prof_pmc.sh r04_strip WRITE_SIZE python3 scripts/time_layer.py 64 128 128
scripts/prof_pmc.sh r04_strip "$MF" python3 scripts/time_layer.py 64 128 128
scripts/prof_pmc.sh r04_conv "$MF" python3 scripts/prof_conv.py 5 all
scripts/prof_pmc.sh r04_prof_conv FETCH_SIZE python3 scripts/prof_conv.py 3 all
scripts/prof_pmc.sh r04_prof_conv WRITE_SIZE python3 scripts/prof_conv.py 3 all
scripts/prof_stats.sh r04_prof_conv python3 scripts/prof_conv.py 10 all
