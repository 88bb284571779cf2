#!/bin/bash
# gfx950 assembly of ONE kernel of a .hip file (no GPU needed):
#   scripts/isa_of.sh owlraytracing_amd/csrc/trueknn_team.hip team_kernelILb0ELi1ELb0E [out.s] [extra hipcc flags...]
# <mangled-substring> selects the function whose "Begin function" line contains it.
set -e
src=$(realpath "$1"); pat=$2; out=${3:-/tmp/isa_of.s}; shift 3 2>/dev/null || shift $#
dir=$(dirname "$src"); root=$(cd "$(dirname "$0")/.." && pwd)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I$root/include -I$root/include/owl_shims -I$dir -Wno-unused-result -Wno-bitwise-instead-of-logical"
/opt/rocm/bin/hipcc $flags "$@" -S --cuda-device-only -o /tmp/isa_of_all.s "$src" 2>/dev/null
awk -v pat="$pat" '/Begin function/ && index($0, pat){f=1} f{print} /End function/{if(f) exit}' /tmp/isa_of_all.s > "$out"
echo "$(wc -l < "$out") lines in $out; $(grep -c s_waitcnt "$out") s_waitcnt, vgpr $(grep -m1 -o 'NumVgprs: [0-9]*' "$out"), scratch $(grep -m1 -o 'ScratchSize: [0-9]*' "$out")"
