#!/bin/bash
# The round's fuzz campaign on the GPU box: tests/test_gpu_parity.py::test_fuzz_params_and_sequences (random k, l, density, hpc, c, s, g x random
# genomes and reads, damaged references and reads; every third case under a random seeding variant) in the product configuration and under
# the test hooks that force the other code paths.  usage: tools/fuzz_campaign.sh <cases per leg> <outfile>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-3000}; OUT=${2:-$ROOT/gpurun_out/fuzz.txt}
cd $ROOT
: > $OUT
leg() {  # seed, env...
  local seed=$1; shift
  echo "== MQ_FUZZ_ITERS=$N MQ_FUZZ_SEED=$seed $*" >> $OUT
  env MQ_FUZZ_ITERS=$N MQ_FUZZ_SEED=$seed "$@" timeout 3000 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz 2>&1 | tail -n 2 >> $OUT
}
leg 7101 X=1
leg 7102 X=1
leg 7103 X=1
leg 7104 MQ_TABLE_FACTOR=2
leg 7105 MQ_FORCE_GENERAL=1
leg 7106 MQ_REF_CAP=8
leg 7107 MQ_CHAIN_CHUNK=4
leg 7108 MQ_PIPELINE=split
cat $OUT
