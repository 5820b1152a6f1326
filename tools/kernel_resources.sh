#!/bin/bash
# Registers, spills, LDS and waves per SIMD of every kernel in rt_kernels.hip (compiles the file once more).
# usage: tools/kernel_resources.sh [extra compiler flags]
here=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -c "$here/raytracer_amd/csrc/rt_kernels.hip" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|LDS Size|Occupancy" \
    | sed -e 's/^.*remark: [^ ]* *//' -e 's/ \[-Rpass.*$//' \
    | awk '/^Name:/ { if (line) print line; line = $2; next } { gsub(/^ +/, ""); line = line "  " $0 } END { print line }'
