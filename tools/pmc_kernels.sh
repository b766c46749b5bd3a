#!/bin/bash
# rocprofv3 PMC counters per kernel of bench.py, one counter group per pass (never mixed with trace domains other than
# --kernel-trace).  usage: tools/pmc_kernels.sh <outdir-under-gpurun_out> [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-pmck}
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" ${PMC_EXTRA:-} ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 3 --warmup 1 "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" < /dev/null
  echo "pass $i ($grp): rc=$?"
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        kn = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))
        kn = re.sub(r"^void ", "", kn)
        if not any(t in kn for t in ("map_kernel", "map_declined", "order_reads", "seed_reads", "map_lists", "seed_general")) or "probe_rate" in kn:
            continue
        agg[(kn, row["Counter_Name"])][0] += float(row["Counter_Value"]); agg[(kn, row["Counter_Name"])][1] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        line = "%-36s %-24s per-launch %.6g  (%d launches)" % (k[0], k[1], agg[k][0] / max(agg[k][1], 1), agg[k][1])
        print(line); fo.write(line + "\n")
PY
