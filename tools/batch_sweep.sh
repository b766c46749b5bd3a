#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for n in "$@"; do python bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 10 --reads $n 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reads', j['config']['reads_per_step_per_gpu'], j['value'], j['ms_per_step'], j['roofline']['frac'])"; done
