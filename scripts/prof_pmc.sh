#!/bin/bash
# Usage: scripts/prof_pmc.sh NAME COUNTER python3 <script> [args]   (one counter per pass: FETCH_SIZE, WRITE_SIZE, ...)
# `rocprofv3 --kernel-trace --pmc COUNTER` only (never combined with other trace domains); leaves
# gpurun_out/NAME_COUNTER.csv = per-kernel averages of the counter.
set -e
name=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/pmc.XXXXXX)
cmd=("$@")
for i in "${!cmd[@]}"; do [ -e "$root/${cmd[$i]}" ] && cmd[$i]="$root/${cmd[$i]}"; done
rocprofv3 --kernel-trace --pmc "$ctr" --output-format csv -d "$d" -- "${cmd[@]}" > "$out/${name}_$ctr.out" 2> "$out/${name}_$ctr.err" || { tail -20 "$out/${name}_$ctr.err"; exit 1; }
f=$(find "$d" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$out/${name}_$ctr.csv" <<'PY'
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
head -12 "$out/${name}_$ctr.csv" | cut -c1-170
