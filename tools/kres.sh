#!/bin/bash
# per-kernel register / scratch / LDS usage of the library build: tools/kres.sh [-D...]   (hipcc cross-compiles, no GPU needed)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-align-mismatch -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -pragma-unroll-threshold=65536 "$@" -Rpass-analysis=kernel-resource-usage -c $ROOT/mapquik_amd/csrc/mq_capi.hip -o /tmp/kres.o 2>&1 \
 | grep "Function Name\|  VGPRs:\|Spill\|Scratch\|Occupancy\|LDS Size" | paste - - - - - - - | sed 's/\/[^ ]*\.h[ip]p*:[0-9]*:[0-9]*: remark: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/Function Name: //; s/  */ /g' | grep "${KRES_FILTER:-map_kernelILi64ELb0ELb0\|seed_reads_kernelILi[012]\|map_lists_kernelILi64ELb0}"
