#!/bin/bash
# Registers / scratch / occupancy of the team kernels as the compiler reports them (no GPU needed), with the
# flags the Makefile uses for trueknn_team.hip.   scripts/team_resources.sh [extra -D flags]
cd "$(dirname "$0")/../owlraytracing_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize \
  -I../../include -I../../include/owl_shims -I. -Wno-unused-result -Wno-bitwise-instead-of-logical "$@" \
  -Rpass-analysis=kernel-resource-usage -c trueknn_team.hip -o /dev/null 2>&1 \
  | grep -A12 "Function Name:" | grep "Function Name\|  VGPRs:\|ScratchSize\|Occupancy\|LDS Size" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - - \
  | sed 's/Function Name: //' | c++filt | grep "team_\|tie_fix" | sed 's/owlmi::(anonymous namespace):://g; s/(TeamArgs[^)]*)//' | tr -s ' \t' ' ' | cut -c1-160
