#!/bin/bash
# Developer helper: the working tree's dbscan.hip with extra compile flags as owlraytracing_amd/libowl_mi355x_<tag>.so
#   scripts/ab_variant_db.sh b4 -DTKNN_DB_BOXES=4
set -e
tag=$1; shift
cd "$(dirname "$0")/../owlraytracing_amd/csrc"
make >/dev/null
mkdir -p diagobj/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden \
  -I../../include -I../../include/owl_shims -I. -Wno-unused-result -Wno-bitwise-instead-of-logical -Wno-unused-variable "$@" \
  -c dbscan.hip -o diagobj/ab/dbscan_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^dbscan.o$') diagobj/ab/dbscan_$tag.o -o ../libowl_mi355x_$tag.so
echo built ../libowl_mi355x_$tag.so
