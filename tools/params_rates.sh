#!/bin/bash
# kernel-only rates away from the default parameters: tools/params_rates.sh > profiles/r04_params.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
echo "# bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 10 --warmup 2 at several (k, l, density): Gbases/s, ms per launch, mapped fraction, Q60 / wrong, k-min-mers per launch"
for P in "5 31 0.01" "7 31 0.01" "8 16 0.01" "5 16 0.01" "4 14 0.05" "5 24 0.01" "5 27 0.02" "7 21 0.01"; do
  set -- $P
  python bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 10 --warmup 2 --k $1 --l $2 --density $3 2>/dev/null < /dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=%-2s l=%-2s d=%-5s  %8.1f Gbases/s  %.3f ms  mapped %.4f  q60 %d wrong %d  kminmers %d  index build %.1f ms' % ('$1','$2','$3', j['value'], j['ms_per_step'], j['mapped_frac'], j['q60'], j['q60_wrong'], j['kminmers_per_step'], j['index_build']['ms']))"
done
