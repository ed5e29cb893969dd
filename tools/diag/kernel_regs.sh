#!/bin/bash
# Register / scratch / occupancy table of the kernels in one HIP source (hipcc's kernel-resource-usage remarks),
# one line per kernel:   tools/diag/kernel_regs.sh pygat_amd/csrc/k1_gemm.hip [name filter]
src="$1"; filt="${2:-.}"
cd "$(dirname "$src")" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$(basename "$src")" -o /dev/null 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/^.*remark: +//; s/ \[-Rpass-analysis=kernel-resource-usage\]$//' | paste -d'|' - - - - - - \
  | sed -E 's/Function Name: //; s/\|/  /g; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/lds/' \
  | while read -r name rest; do echo "$(echo "$name" | c++filt | sed -E 's/pygat:://g; s/\(.*//; s/^void //') $rest"; done | grep -E "$filt"
