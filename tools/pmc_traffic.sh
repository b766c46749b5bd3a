#!/bin/bash
# HBM traffic of map_kernel from the TCC counters, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit together),
# for the full kernel and for the stage-truncated diagnostic runs (MQ_STOP_AFTER) used to calibrate FETCH_SIZE on gfx950.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for s in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "== MQ_STOP_AFTER=$s $c"
    MQ_STOP_AFTER=$s $ROOT/tools/pmc_one.sh tr_${s}_$(echo $c | tr ' ' '_') "$c" | grep -v "^$"
  done
done
