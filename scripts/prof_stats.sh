#!/bin/bash
# Usage (on the GPU box, from the repo root): scripts/prof_stats.sh NAME python3 <script> [args]
# Runs the program under `rocprofv3 --kernel-trace --stats` and leaves gpurun_out/NAME_kernel_stats.csv (the per-kernel
# summary) and gpurun_out/NAME.out (the program's stdout).  The program comes right after `--`: no env / bash hop.
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/prof.XXXXXX)
cmd=("$@")
# (scripts are given relative to the repo root)
for i in "${!cmd[@]}"; do [ -e "$root/${cmd[$i]}" ] && cmd[$i]="$root/${cmd[$i]}"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- "${cmd[@]}" > "$out/$name.out" 2> "$out/$name.err" || { tail -20 "$out/$name.err"; exit 1; }
f=$(find "$d" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/${name}_kernel_stats.csv"
head -12 "$out/${name}_kernel_stats.csv" | cut -c1-150
