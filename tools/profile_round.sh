#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU call: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>/...)
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command (10 timed steps)
#   bench_stats.json   the bench line of that same run (its own HIP-event average must agree with the csv)
#   pmc/summary.txt    SQ / GRBM counters of map_kernel (separate --pmc passes, --kernel-trace only)
#   traffic/raw.txt    FETCH_SIZE / WRITE_SIZE passes (fused + split pipeline) for pmc_traffic.json
#   stage_clocks.txt   wave time per stage (needs mapquik_amd/lib/clk4.so, a -DMQ_STAGE_CLOCKS build)
#   probe_rate.txt     random index probes per second of the memory system
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-configs --no-smaller-batches --steps 10 --warmup 2 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cd $ROOT
rm -rf $ROOT/gpurun_out/traffic
tools/pmc_kernels.sh $TAG/pmc > /dev/null 2>&1
tools/pmc_traffic.sh > /dev/null 2>&1
mkdir -p $OUT/traffic && cp $ROOT/gpurun_out/traffic/raw.txt $OUT/traffic/raw.txt
# the two JSON files bench.py reads for roofline.traffic and roofline.secondary (copy them to profiles/ when they describe the committed kernel)
python3 tools/traffic_json.py $OUT/traffic/raw.txt $OUT/bench_stats.json "$(cat $ROOT/.commit 2>/dev/null || echo unknown)" $OUT/kernel_stats.csv > $OUT/pmc_traffic.json 2> $OUT/traffic_json.err
python3 tools/issue_json.py $OUT/pmc $OUT/bench_stats.json ${CEILING_CPI:-2.51} "${CEILING_SRC:-profiles/r04_valu_enc_stepb.txt: stage-B step as built (10 VALU, SDWA table offset): 25.1 cycles per step at 4 waves per SIMD as two 8-wave workgroups per CU}" "$(cat $ROOT/.commit 2>/dev/null || echo unknown)" > $OUT/pmc_issue.json 2> $OUT/issue_json.err
if [ -f mapquik_amd/lib/clk4.so ]; then MQ_LIB=$ROOT/mapquik_amd/lib/clk4.so python3 tools/stage_clocks.py 2>&1 | grep -v amdgpu.ids > $OUT/stage_clocks.txt; fi
python3 tools/probe_rate.py > $OUT/probe_rate.txt 2>&1
head -3 $OUT/kernel_stats.csv; cat $OUT/pmc/summary.txt | grep "map_kernel<64, false, false>"; cat $OUT/traffic/raw.txt; cat $OUT/stage_clocks.txt; tail -12 $OUT/probe_rate.txt; tail -c 1200 $OUT/bench_stats.json
