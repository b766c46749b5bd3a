#!/bin/bash
# The six- and eight-waves-per-SIMD builds of the library (kept behind macros, with their counters, profiles/NOTES.md):
#   tools/occ_builds.sh   -> mapquik_amd/lib/libmq_w6.so (12-wave workgroups x 2, 80 VGPRs, two-super-row tiles, three lane-batches)
#                            mapquik_amd/lib/libmq_w8.so (16-wave workgroups x 2, 64 VGPRs, one-super-row tiles)
#                            mapquik_amd/lib/libmq_ldslist.so (the current read's minimizer list in LDS, see below)
# Same results as the product build (MQ_LIB=... python -m pytest tests/test_gpu_parity.py -m gpu); A/B: tools/abq.sh default libmq_w6.so libmq_w8.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-align-mismatch -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -pragma-unroll-threshold=65536"
$B -DMQ_MAP_WAVES=12 -DMQ_MAP_MIN_WAVES=6 -DMQ_SD_MAX_SR=2 -DMQ_ML_NB=3 -DMQ_SEED_WAVES=12 -DMQ_SEED_MIN_WAVES=6 -DMQ_SD_OWNER_CAP=128 -DMQ_SD_CROSS_PREFETCH=0 \
   -o $ROOT/mapquik_amd/lib/libmq_w6.so $ROOT/mapquik_amd/csrc/mq_capi.hip &
$B -DMQ_MAP_WAVES=16 -DMQ_MAP_MIN_WAVES=8 -DMQ_SD_MAX_SR=1 -DMQ_ML_NB=3 -DMQ_SEED_WAVES=16 -DMQ_SEED_MIN_WAVES=8 -DMQ_SD_OWNER_CAP=64 -DMQ_SD_CROSS_PREFETCH=0 \
   -o $ROOT/mapquik_amd/lib/libmq_w8.so $ROOT/mapquik_amd/csrc/mq_capi.hip &
# the read's minimizer list kept in LDS: one 16-wave workgroup per CU, two-super-row tiles, 256 entries per wave --
# the bench's reads list 351 minimizers on average (4.2 KB), so nearly every list overflows into device memory: 1175 against 1218
$B -DMQ_MAP_WAVES=16 -DMQ_MAP_MIN_WAVES=4 -DMQ_SD_MAX_SR=2 -DMQ_LDS_LIST=256 -o $ROOT/mapquik_amd/lib/libmq_ldslist.so $ROOT/mapquik_amd/csrc/mq_capi.hip &
# the next read's first super-row requested during the map phase, straight into LDS (mq_map_kernels.hpp MQ_LDS_PREFETCH): 1237 against 1252
$B -DMQ_LDS_PREFETCH=1 -o $ROOT/mapquik_amd/lib/libmq_ldsprefetch.so $ROOT/mapquik_amd/csrc/mq_capi.hip &
wait
for f in w6 w8 ldslist ldsprefetch; do ls -la $ROOT/mapquik_amd/lib/libmq_$f.so; done
