#!/bin/bash
# register / LDS / scratch usage of the kernels of one .hip file: tools/kres.sh <file.hip> [name-regex]
cd "$(dirname "$0")/../seal_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|Spill|Scratch|LDS Size|Occupancy" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//; s/^[^ ]* remark: //' | grep -A7 -E "error|Name: .*(${2:-.})"
