#!/bin/bash
# Diagnostic: instructions and time of the seed phase by stage (split pipeline, truncated seeders MQ_SEED_STOP=1/2) and of the
# map phase.  Results of the truncated runs are not valid mappings; only the counters are read.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for s in 1 2 0; do
  echo "== MQ_SEED_STOP=$s"
  MQ_PIPELINE=split MQ_SEED_STOP=$s PMC_ONLY=1 $ROOT/tools/pmc_kernels.sh st$s > /dev/null 2>&1 < /dev/null
  grep -E "seed_reads.*(SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|GRBM_GUI|SQ_WAIT_ANY |SQ_WAVE_CYCLES|SQ_LDS_BANK|SQ_LDS_IDX)" $ROOT/gpurun_out/st$s/summary.txt
done
grep -E "map_lists_kernel<64, false>.*(SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|GRBM_GUI)" $ROOT/gpurun_out/st0/summary.txt
