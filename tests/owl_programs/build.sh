#!/bin/bash
# Builds the test host driver and the test device programs into build/owl_tests/ (git-ignored).
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
out=$root/build/owl_tests
mkdir -p "$out"
python3 "$root/tools/owl_embed.py" radiusCode "$here/radius_programs.cu" -o "$out/radiusCode.c" --keep-hsaco "$out/radius_programs.hsaco"
python3 "$root/tools/owl_embed.py" apiCode "$here/api_programs.cu" -o "$out/apiCode.c" --keep-hsaco "$out/api_programs.hsaco"
# the same programs as built against ANOTHER revision of the device header (its layout word differs): owlBuildPrograms must refuse them
python3 "$root/tools/owl_embed.py" staleCode "$here/radius_programs.cu" -o "$out/staleCode.c" --keep-hsaco "$out/stale_abi_programs.hsaco" -D OWL_MI355X_DEVICE_ABI=65576u
g++ -O2 -std=c++17 -Wall -I"$root/include" -I"$root/include/owl_shims" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__=1 \
    "$here/owl_host_driver.cpp" -o "$out/owl_host_driver" \
    -L"$root/owlraytracing_amd" -lowl_mi355x -L/opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,"$root/owlraytracing_amd" -Wl,-rpath,'$ORIGIN/../../owlraytracing_amd' -Wl,-rpath,/opt/rocm/lib
# the RT-DBSCAN application on the OWL API (samples/s02-rtdbscan: host code + device programs of this repository)
bash "$root/samples/s02-rtdbscan/build.sh" > /dev/null
echo "built $out"
