#!/bin/bash
# HBM traffic of the map path from the TCC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (they do not fit one pass), --kernel-trace only.  Two configurations:
#   fused  the product kernel (map_kernel)                                   -> the traffic figure
#   split  seed_reads_kernel alone streams the read bases with 16-B lane loads: its FETCH_SIZE over the known byte count is
#          the gfx950 calibration of wide coalesced streams (the guide: tallied at half)
# usage: tools/pmc_traffic.sh   (writes gpurun_out/traffic/*.txt; tools/traffic_json.py turns them into profiles/pmc_traffic.json)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/traffic
mkdir -p $OUT
for mode in fused split; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $mode = split ]; then export MQ_PIPELINE=split; else unset MQ_PIPELINE; fi
    $ROOT/tools/pmc_one.sh tr_${mode}_$c "$c" > /dev/null 2>&1 < /dev/null
    python3 - "$ROOT/gpurun_out/tr_${mode}_$c" "$mode" "$c" >> $OUT/raw.txt <<'PY'
import sys, glob, csv, collections, re
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        kn = re.sub(r"^void ", "", re.sub(r"\(.*", "", row.get("Kernel_Name", "")))
        if any(t in kn for t in ("map_kernel<64, false, false>", "seed_reads_kernel", "map_lists_kernel<64, false>", "seed_general")):
            agg[kn][0] += float(row["Counter_Value"]); agg[kn][1] += 1
for k in sorted(agg):
    print("%s %s %s %.6g %d" % (sys.argv[2], sys.argv[3], k.replace(" ", ""), agg[k][0] / max(agg[k][1], 1), agg[k][1]))
PY
  done
done
unset MQ_PIPELINE
cat $OUT/raw.txt
