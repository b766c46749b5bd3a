#!/bin/bash
# instruction-cache counters of map_kernel at 1, 2 and 48 reads per wave: tools/icache_probe.sh   (writes gpurun_out/icache/)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*" | sort -u > $OUT/counters.txt
for n in ${ICACHE_SIZES:-4096 8192 196608}; do
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES" "SQ_IFETCH SQ_BUSY_CYCLES"; do
    d=$OUT/n${n}_$(echo $grp | tr ' ' '+')
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 3 --warmup 1 --reads $n > $d.json 2> $d.err < /dev/null
  done
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections, os
out = sys.argv[1]
for n in (4096, 8192, 196608, 1572864):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + "/n%d_*/**/*counter_collection.csv" % n, recursive=True):
        for row in csv.DictReader(open(f)):
            if "map_kernel<64, false, false>" in row.get("Kernel_Name", ""):
                agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
    print("== %d reads per launch" % n)
    for k in sorted(agg): print("  %-30s per launch %.6g  (%d launches)" % (k, agg[k][0] / max(agg[k][1], 1), agg[k][1]))
PY
cat $OUT/counters.txt | tr '\n' ' '
