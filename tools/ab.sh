#!/bin/bash
# A/B two builds of the library on one box: tools/ab.sh libbase.so libcand.so   (both under mapquik_amd/lib/)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in 1 2 3; do for L in "$@"; do MQ_LIB=$ROOT/mapquik_amd/lib/$L python bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', j['value'], j['ms_per_step'])"; done; done
for L in "$@"; do echo "== $L"; MQ_LIB=$ROOT/mapquik_amd/lib/$L tools/pmc_one.sh ab_$L "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" | grep -v "^$"; done
MQ_LIB=$ROOT/mapquik_amd/lib/${@: -1} python -m pytest tests -m gpu -x -q 2>&1 | tail -1
