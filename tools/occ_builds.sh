#!/bin/bash
# The six- and eight-waves-per-SIMD builds of the library (kept behind macros, with their counters, profiles/NOTES.md):
#   tools/occ_builds.sh   -> mapquik_amd/lib/libmq_w6.so (12-wave workgroups x 2, 80 VGPRs, two-super-row tiles, three lane-batches)
#                            mapquik_amd/lib/libmq_w8.so (16-wave workgroups x 2, 64 VGPRs, one-super-row tiles)
# Same results as the product build (MQ_LIB=... python -m pytest tests/test_gpu_parity.py -m gpu); A/B: tools/abq.sh default libmq_w6.so libmq_w8.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-align-mismatch -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -pragma-unroll-threshold=65536"
$B -DMQ_MAP_WAVES=12 -DMQ_MAP_MIN_WAVES=6 -DMQ_SD_MAX_SR=2 -DMQ_ML_NB=3 -DMQ_SEED_WAVES=12 -DMQ_SEED_MIN_WAVES=6 -DMQ_SD_OWNER_CAP=128 -DMQ_SD_CROSS_PREFETCH=0 \
   -o $ROOT/mapquik_amd/lib/libmq_w6.so $ROOT/mapquik_amd/csrc/mq_capi.hip &
$B -DMQ_MAP_WAVES=16 -DMQ_MAP_MIN_WAVES=8 -DMQ_SD_MAX_SR=1 -DMQ_ML_NB=3 -DMQ_SEED_WAVES=16 -DMQ_SEED_MIN_WAVES=8 -DMQ_SD_OWNER_CAP=64 -DMQ_SD_CROSS_PREFETCH=0 \
   -o $ROOT/mapquik_amd/lib/libmq_w8.so $ROOT/mapquik_amd/csrc/mq_capi.hip &
# (the builds with the read's minimizer list in LDS and with the next read prefetched into LDS -- MQ_LDS_LIST, MQ_LDS_PREFETCH: 1175 against
# 1218, 1237 against 1252 -- left the source with round 6; last at commit cc976e3, numbers in profiles/NOTES.md)
wait
for f in w6 w8; do ls -la $ROOT/mapquik_amd/lib/libmq_$f.so; done
