#!/bin/bash
# libowl_mi355x_time.so: the diagnostic library (TKNN_DB_DIAG switches and timers) with the PRODUCTION packet-walk stack of
# dbscan.hip -- `make DIAG=1` builds it with TKNN_DB_STACK=320, which makes the walk pop depth-first (the test build of
# the rarely taken paths), so its wave-time shares say nothing about the shipped kernel.  Needs `make DIAG=1` first.
set -e
cd "$(dirname "$0")/../owlraytracing_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include -I../../include/owl_shims -I. -Wno-unused-result -Wno-bitwise-instead-of-logical"
mkdir -p diagobj
/opt/rocm/bin/hipcc $FLAGS -DTKNN_DIAG_BUILD=1 "$@" -c dbscan.hip -o diagobj/dbscan_time.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC diagobj/lbvh.o diagobj/trueknn.o diagobj/trueknn_wave.o diagobj/trueknn_team.o diagobj/dbscan_time.o diagobj/halo_select.o diagobj/owl_runtime.o -o ../libowl_mi355x_time.so
echo built owlraytracing_amd/libowl_mi355x_time.so
