#!/bin/bash
# registers / LDS / scratch of every kernel of the built gfx950 code object (from the notes of vf_kernels.o)
set -e
O=${1:-vfind_amd/csrc/vf_kernels.o}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | awk '
/\.name:/ {name=$2}
/\.sgpr_count:/ {s=$2}
/\.group_segment_fixed_size:/ {l=$2}
/\.private_segment_fixed_size:/ {p=$2}
/\.vgpr_count:/ {printf "vgpr %-4s sgpr %-4s lds %-6s scratch %-4s %s\n", $2, s, l, p, name}' | c++filt | sort -k9
# (.vgpr_count is the LAST field of a kernel's note: printing at .symbol, as this script did until round 3, paired every
# kernel with its predecessor's register count)
rm -rf $T
