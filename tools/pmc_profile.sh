#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py's map_kernel in separate passes (one counter group per run).
# usage: tools/pmc_profile.sh <outdir-under-gpurun_out> [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc}
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-e2e --steps 3 --warmup 1 "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err"
  echo "pass $i ($grp): rc=$?"
done
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "map_kernel" not in row.get("Kernel_Name", ""):
            continue
        k = row["Counter_Name"]
        agg[k][0] += float(row["Counter_Value"])
        agg[k][1] += 1
with open(out + "/summary.txt", "w") as w:
    for k in sorted(agg):
        s, n = agg[k]
        line = "%-28s per-launch %.6g (launches %d)" % (k, s / max(n, 1), n)
        print(line)
        w.write(line + "\n")
PY
