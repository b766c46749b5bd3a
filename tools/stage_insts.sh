#!/bin/bash
# Diagnostic: dynamic instruction counts attributed to stages by stopping each read after stage N (MQ_STOP_AFTER).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for s in 1 2 3 0; do
  echo "== MQ_STOP_AFTER=$s"
  MQ_STOP_AFTER=$s $ROOT/tools/pmc_one.sh si$s "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" | grep -v "^$"
done
