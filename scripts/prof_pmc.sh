#!/bin/bash
# Usage: scripts/prof_pmc.sh NAME COUNTER python3 <script> [args]   (one pass; COUNTER may be a quoted, space-separated list that
# fits the per-block slots: "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; FETCH_SIZE and WRITE_SIZE need a pass each)
# `rocprofv3 --kernel-trace --pmc COUNTER` only (never combined with other trace domains); leaves
# gpurun_out/NAME_COUNTER.csv = per-kernel averages of the counter.
set -e
name=$1; ctr=$2; shift; shift
tag=${ctr// /+}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/pmc.XXXXXX)
cmd=("$@")
for i in "${!cmd[@]}"; do [ -e "$root/${cmd[$i]}" ] && cmd[$i]="$root/${cmd[$i]}"; done
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$d" -- "${cmd[@]}" > "$out/${name}_$tag.out" 2> "$out/${name}_$tag.err" || { tail -20 "$out/${name}_$tag.err"; exit 1; }
f=$(find "$d" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$out/${name}_$tag.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
with open(sys.argv[2], 'w') as f:
    f.write('"Kernel_Name","Counter_Name","Dispatches","Average","Min","Max"\n')
    for (k, c), v in sorted(agg.items()):
        f.write('"%s","%s",%d,%.1f,%.1f,%.1f\n' % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
PY
head -12 "$out/${name}_$tag.csv" | cut -c1-170
