#!/bin/bash
# Developer A/B helper: builds owlraytracing_amd/libowl_mi355x_prev.so with the team kernel of a
# given git revision (default HEAD) next to the working-tree library, so both can be timed on the
# same GPU box:  OWL_MI355X_LIB=$PWD/owlraytracing_amd/libowl_mi355x_prev.so python scripts/quick_bench.py ...
set -e
rev=${1:-HEAD}
cd "$(dirname "$0")/../owlraytracing_amd/csrc"
make >/dev/null
mkdir -p diagobj/ab
git show $rev:owlraytracing_amd/csrc/trueknn_team.hip > diagobj/ab/trueknn_team_prev.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden \
  -I../../include -I../../include/owl_shims -I. -Wno-unused-result -Wno-bitwise-instead-of-logical \
  -c diagobj/ab/trueknn_team_prev.hip -o diagobj/ab/team_prev.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^trueknn_team.o$') diagobj/ab/team_prev.o \
  -o ../libowl_mi355x_prev.so
echo built ../libowl_mi355x_prev.so from $rev
