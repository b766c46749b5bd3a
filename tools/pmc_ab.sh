#!/bin/bash
# PMC comparison of library builds: tools/pmc_ab.sh <tag> lib1.so lib2.so ...  ("default" = the product build)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
for L in "$@"; do
  if [ "$L" = "default" ]; then unset MQ_LIB; else export MQ_LIB=$ROOT/mapquik_amd/lib/$L; fi
  echo "== $L"
  $ROOT/tools/pmc_one.sh ${TAG}_${L}_1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
  $ROOT/tools/pmc_one.sh ${TAG}_${L}_2 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
done
