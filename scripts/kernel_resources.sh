#!/bin/bash
# registers / LDS / scratch of every kernel of the built gfx950 code objects (from the notes of vf_k_*.o, or of the objects named)
set -e
if [ $# -gt 0 ]; then OBJS="$@"; else OBJS=$(ls vfind_amd/csrc/vf_k_front.o vfind_amd/csrc/vf_k_dp.o vfind_amd/csrc/vf_k_table.o vfind_amd/csrc/vf_k_text.o); fi
T=$(mktemp -d)
for O in $OBJS; do
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co; rm -f $T/k.co $T/fat.bin
done | awk '
/\.name:/ {name=$2}
/\.sgpr_count:/ {s=$2}
/\.group_segment_fixed_size:/ {l=$2}
/\.private_segment_fixed_size:/ {p=$2}
/\.vgpr_count:/ {printf "vgpr %-4s sgpr %-4s lds %-6s scratch %-4s %s\n", $2, s, l, p, name}' | c++filt | sort -k9
# (.vgpr_count is the LAST field of a kernel's note: printing at .symbol, as this script did until round 3, paired every
# kernel with its predecessor's register count)
rm -rf $T
