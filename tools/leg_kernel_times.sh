#!/bin/bash
# Per-launch durations of map_kernel / map_declined_kernel / order_reads_kernel through a bench run WITH its configuration legs (rocprofv3
# --kernel-trace): which kernel a leg's milliseconds belong to.  tools/leg_kernel_times.sh <tag>  -> gpurun_out/<tag>/leg_kernels.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-legs}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-smaller-batches --steps 5 --warmup 1 > $OUT/bench_legs.json 2> $OUT/bench_legs.err
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - $f > $OUT/leg_kernels.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("map_kernel", "map_declined", "order_reads"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group consecutive launch sequences (order, map, declined) and print one line per sequence
seq, out = [], []
for r in rows:
    nm = "order" if "order_reads" in r["Kernel_Name"] else "declined" if "declined" in r["Kernel_Name"] else "map"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if nm == "order" and seq:
        out.append(seq); seq = []
    seq.append((nm, d, r["Kernel_Name"][:48], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if seq: out.append(seq)
for s in out:
    d = {nm: dd for nm, dd, _, _, _ in s}
    span = (s[-1][4] - s[0][3]) / 1e6
    print("order %.3f  map %.3f  declined %.3f  span %.3f ms   %s" % (d.get("order", 0), d.get("map", 0), d.get("declined", 0), span, [x[2] for x in s if x[0] == "map"][0] if any(x[0] == "map" for x in s) else ""))
PY
rm -rf $OUT/trace
cat $OUT/leg_kernels.txt
