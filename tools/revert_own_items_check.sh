#!/bin/bash
# Evidence for tests/test_gpu_poison.py: a library built from a copy of the sources in which map_kernel's take_now() forgets the wave's second
# own work item when its first is the marked entry of a read that went first (the bug fixed by commit 8215505) must FAIL the poisoned-output
# sweeps -- deterministically, by unwritten records, not by luck.
#   tools/revert_own_items_check.sh build      (here: hipcc cross-compiles)  -> mapquik_amd/lib/libmq_revert8215505.so
#   tools/revert_own_items_check.sh run        (on the GPU box)              -> pytest's verdict per sweep, expected: all four FAIL
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
if [ "$1" = "build" ]; then
  T=$(mktemp -d)
  mkdir -p $T/mapquik_amd && cp -r mapquik_amd/csrc $T/mapquik_amd/ && cp -r include $T/
  python3 - "$T/mapquik_amd/csrc/mq_map_kernels.hpp" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "            } else if (own2 != 0xFFFFFFFFu) {"
assert s.count(old) == 1
s = s.replace(old, "            } else if (false) {  // REVERT of 8215505 (test evidence only)")
open(p, "w").write(s)
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-align-mismatch -mllvm -amdgpu-atomic-optimizer-strategy=None \
     -mllvm -pragma-unroll-threshold=65536 -o mapquik_amd/lib/libmq_revert8215505.so $T/mapquik_amd/csrc/mq_capi.hip && ls -la mapquik_amd/lib/libmq_revert8215505.so
  rm -rf $T
else
  MQ_LIB=$ROOT/mapquik_amd/lib/libmq_revert8215505.so python -m pytest tests/test_gpu_poison.py -m gpu -q -k edges 2>&1 | grep -v "^$" | tail -15
fi
