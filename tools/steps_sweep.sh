#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for sw in "10 2" "50 10" "200 50" "10 2"; do set -- $sw; python bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps $1 --warmup $2 --reads ${READS:-49152} 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps', j['steps'], 'warmup', j['warmup'], j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"; done
