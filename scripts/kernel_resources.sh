#!/bin/bash
# Registers, scratch and occupancy of every kernel in one .hip file, as the compiler reports them (no GPU
# needed):  scripts/kernel_resources.sh owlraytracing_amd/csrc/trueknn_team.hip [filter]
# and, with ISA=1, the gfx950 assembly in /tmp/<name>.s (look for v_cmp_*_i32 + s_and_saveexec in loops:
# wave-uniform values kept in VGPRs).
set -e
src=$1; filt=${2:-.}
dir=$(cd "$(dirname "$src")" && pwd); root=$(cd "$(dirname "$0")/.." && pwd)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I$root/include -I$root/include/owl_shims -I$dir -Wno-unused-result -Wno-bitwise-instead-of-logical"
/opt/rocm/bin/hipcc $flags -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 \
  | grep -A12 "Function Name:" | grep "Function Name\|SGPRs:\|VGPRs:\|ScratchSize\|Occupancy" \
  | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - - | grep "$filt" | sed 's/Function Name: //' | c++filt | cut -c1-230
if [ "${ISA:-0}" = 1 ]; then
  out=/tmp/$(basename "$src" .hip).s
  /opt/rocm/bin/hipcc $flags -S --cuda-device-only -o "$out" "$src" 2>/dev/null
  echo "ISA in $out"
fi
