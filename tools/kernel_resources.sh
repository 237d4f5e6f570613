#!/bin/bash
# VGPRs / SGPRs / LDS / scratch / occupancy of every kernel in the library (device-only compile, -Rpass-analysis=kernel-resource-usage)
# usage: tools/kernel_resources.sh [name-filter] [extra hipcc flags, e.g. -DSB_LAB=1]
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Iinclude --cuda-device-only -c \
  -Rpass-analysis=kernel-resource-usage ${2} sparsebench_amd/csrc/sbhip.hip -o /dev/null 2>&1 |
  awk '/remark: Function Name:/ {n=$(NF-1)} /remark: +TotalSGPRs:/ {s=$(NF-1)} /remark: +VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
       /Occupancy/ {o=$(NF-1)} /LDS Size/ {print n, "vgpr", v, "sgpr", s, "scratch", sc, "waves/SIMD", o, "lds", $(NF-1)}' |
  c++filt | sed -E 's/\(.*\) vgpr/ vgpr/; s/^sbk:://' | grep -E "${1:-.}"
