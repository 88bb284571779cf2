#!/bin/bash
# Builds build/owl_tests/sample02-rtdbscan without cmake (the same two steps the cmake macro runs).
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
out=$root/build/owl_tests
mkdir -p "$out"
python3 "$root/tools/owl_embed.py" ptxCode "$here/deviceCode.cu" -o "$out/s02_ptxCode.c" -I "$here"
inc="-I$root/include -I$root/include/owl_shims -I/opt/rocm/include -I$here -D__HIP_PLATFORM_AMD__=1"
g++ -O2 -std=c++17 -Wall -Wno-missing-field-initializers $inc -c "$here/hostCode.cpp" -o "$out/s02_hostCode.o"
gcc -O1 -c "$out/s02_ptxCode.c" -o "$out/s02_ptxCode.o"
g++ "$out/s02_hostCode.o" "$out/s02_ptxCode.o" -o "$out/sample02-rtdbscan" \
    -L"$root/owlraytracing_amd" -lowl_mi355x -L/opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,"$root/owlraytracing_amd" -Wl,-rpath,'$ORIGIN/../../owlraytracing_amd' -Wl,-rpath,/opt/rocm/lib
rm -f "$out/s02_hostCode.o" "$out/s02_ptxCode.o" "$out/s02_ptxCode.c"
echo "built $out/sample02-rtdbscan"
