#!/bin/bash
# After scripts/prof_r04_final.sh has run on the GPU box and gpurun has merged gpurun_out/ back: copy the summaries that are
# committed under profiles/ and rebuild the derived files (traffic.json, r04_mfma_busy.csv).  Run from the repo root, here (no GPU).
set -e
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
MF="SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CU_CYCLES+SQ_WAVE_CYCLES+SQ_INSTS_VALU+SQ_WAIT_INST_ANY+SQ_ACTIVE_INST_ANY+GRBM_GUI_ACTIVE"
for f in r04_bench_kernel_stats.csv r04_bench_line_under_rocprof.json r04_recipe_kernel_stats.csv r04_recipe_line_under_rocprof.json \
         r04_srcnn_train_kernel_stats.csv r04_prof_conv_kernel_stats.csv r04_espcn_image_kernel_stats.csv r04_bench_line.json \
         r04_time_espcn.txt r04_time_espcn_image.txt r04_time_layer_41.txt r04_time_srcnn.txt r04_time_srcnn_image.txt \
         r04_time_srcnn_train.txt r04_time_espcn_train.txt r04_time_vdsr_batch_eager.txt; do
    grep -v 'amdgpu.ids' "$G/$f" > "$P/$f"
done
cp "$G/r04_conv_$MF.csv" "$P/r04_prof_conv_sq_counters.csv"
(head -1 "$G/r04_espcn_image_$MF.csv"; grep 'srx::' "$G/r04_espcn_image_$MF.csv") > "$P/r04_espcn_image_sq_counters.csv"
python3 scripts/record_traffic.py "$G/r04_prof_conv_FETCH_SIZE.csv" "$G/r04_prof_conv_WRITE_SIZE.csv" "$P/r04_prof_conv_hbm_counters.csv" > /dev/null
python3 scripts/summarize_mfma_busy.py r04 > /dev/null
echo "profiles/ updated; traffic.json sha $(python3 -c 'import json; print(json.load(open("profiles/traffic.json"))["csrc_sha"])')"
