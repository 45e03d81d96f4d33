#!/bin/bash
# Register / LDS / occupancy report of one kernel instantiation without building the library:
#   tools/kres.sh 'acx::k_r1cs_sell_split<Bn254Fr, 0>(const SellSystem*, SellSystem)' [-DACX_K2_PIPE=1 ...]
# (explicit instantiation in a scratch translation unit; hipcc -Rpass-analysis=kernel-resource-usage)
set -e
sig="$1"; shift
d=$(mktemp -d)
cat > $d/k.hip <<EOT
#include "kernels.hip.h"
#include "ntt_r4.hip.h"
using namespace acx;
template __global__ void $sig;
EOT
name=$(echo "$sig" | sed 's/acx:://; s/<.*//')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -I$(dirname $0)/../arithmetic-circuits_amd/csrc -c $d/k.hip -o $d/k.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name: .*$name" | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
rm -rf $d
