#!/bin/bash
# one PMC pass: tools/pmc_one.sh <outdir> "<counters>" [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1; CNT=$2; shift 2
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/p" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 3 --warmup 1 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "map_kernel" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
for k in sorted(agg): print("%-28s per-launch %.6g" % (k, agg[k][0] / max(agg[k][1], 1)))
PY
