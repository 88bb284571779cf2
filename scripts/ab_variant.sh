#!/bin/bash
# Developer helper: the working tree's team kernel with extra compile flags as owlraytracing_amd/libowl_mi355x_<tag>.so
#   scripts/ab_variant.sh v72 -DTKNN_MAX_PER_QUERY=72
set -e
tag=$1; shift
cd "$(dirname "$0")/../owlraytracing_amd/csrc"
make >/dev/null
mkdir -p diagobj/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize \
  -I../../include -I../../include/owl_shims -I. -Wno-unused-result -Wno-bitwise-instead-of-logical -Wno-unused-variable "$@" \
  -c trueknn_team.hip -o diagobj/ab/team_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^trueknn_team.o$') diagobj/ab/team_$tag.o -o ../libowl_mi355x_$tag.so
echo built ../libowl_mi355x_$tag.so
