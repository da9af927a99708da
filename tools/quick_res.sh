#!/bin/bash
# compile ONE resident instantiation and print its kernels' registers / spills:  tools/quick_res.sh 'double, PwlOp<double, 2, false, true, true>'
cd /root/repo/tramp_amd/csrc
T=$(echo "$1" | cut -d, -f1)
cat > _q.hip <<E2
#include "ew_resident_impl.h"
RES_INSTANTIATE($1)
E2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -c _q.hip -o build/_q.o 2>&1 | grep -E "error" -A3
rm -f _q.hip
/root/repo/tools/kernel_resources.sh _q ew_resident | sed 's/^void ew_resident//' 
rm -f build/_q.o
