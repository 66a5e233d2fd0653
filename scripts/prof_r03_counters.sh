#!/bin/bash
# Round 3 counter passes (each its own rocprofv3 --pmc run, program right after `--`):
#  * MFMA busy cycles against CU-busy / GUI-active cycles for the VDSR body kernels, the wide-layer kernel and the strip kernels
#  * LDS activity / bank conflicts and wave wait buckets of the sub-pixel map
#  * FETCH_SIZE / WRITE_SIZE of the strip kernels on the current binary (the 1.2x wgrad over-read figure was round 1's)
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
scripts/prof_pmc.sh r03_conv "$MF" python3 scripts/prof_conv.py 5 all
scripts/prof_pmc.sh r03_wide "$MF" python3 scripts/time_wide.py 4 512
scripts/prof_pmc.sh r03_strip "$MF" python3 scripts/time_layer.py 4 512 512
scripts/prof_pmc.sh r03_d2s "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" python3 scripts/prof_conv.py 5 d2s
scripts/prof_pmc.sh r03_strip FETCH_SIZE python3 scripts/time_layer.py 4 512 512
scripts/prof_pmc.sh r03_strip WRITE_SIZE python3 scripts/time_layer.py 4 512 512
