#!/bin/bash
# quick A/B of library builds on one box: tools/abq.sh [lib.so ...]  ("" = the default build); prints Gbases/s, ms/step
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2; do for L in "$@"; do
  if [ "$L" = "default" ]; then unset MQ_LIB; else export MQ_LIB=$ROOT/mapquik_amd/lib/$L; fi
  timeout 200 python bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches ${BENCH_ARGS:-} 2>/dev/null < /dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', j['value'], j['ms_per_step'], j['overflow_reads'], j['mapped_frac'])"
done; done
