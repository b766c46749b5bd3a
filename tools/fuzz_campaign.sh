#!/bin/bash
# The round's fuzz campaign on the GPU box: tests/test_gpu_parity.py::test_fuzz_params_and_sequences (random k, l, density, hpc, c, s, g x random
# genomes and reads, damaged references and reads; every third case under a random seeding variant) in the product configuration and under
# the test hooks that force the other code paths.  usage: tools/fuzz_campaign.sh <cases per leg> <outfile>   (11 x that many cases in all)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-3000}; OUT=${2:-$ROOT/gpurun_out/fuzz.txt}
cd $ROOT
: > $OUT
leg() {  # cases, seed, env...
  local n=$1 seed=$2; shift 2
  echo "== MQ_FUZZ_ITERS=$n MQ_FUZZ_SEED=$seed $*" >> $OUT
  env MQ_FUZZ_ITERS=$n MQ_FUZZ_SEED=$seed "$@" timeout 5000 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz 2>&1 | tail -n 2 >> $OUT
}
S=${FUZZ_SEED_BASE:-8100}
leg $N $((S+1)) X=1
leg $N $((S+2)) X=1
leg $N $((S+3)) X=1
leg $N $((S+4)) MQ_TABLE_FACTOR=2
leg $((3*N)) $((S+5)) MQ_FORCE_GENERAL=1     # every read and every reference segment through the general seeder (the 1-KB walk), N-damaged reads among them
leg $N $((S+6)) MQ_REF_CAP=8
leg $N $((S+7)) MQ_CHAIN_CHUNK=4
leg $N $((S+8)) MQ_PIPELINE=split
leg $N $((S+9)) MQ_FUZZ_FAST_KH=1            # the opt-in tuple hash against the oracle's bit 64
cat $OUT
