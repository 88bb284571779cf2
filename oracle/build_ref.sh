#!/bin/bash
# Builds, from the reference's OWN sources where they lie under /root/reference (never copied),
# the unchanged TrueKNN sample against this repo's OWL headers and library:
#     oracle/_ref/sample01-trueknn    (hostCode.cpp + deviceCode.cu of samples/s01-trueknn)
# Outputs go only to oracle/_ref/ (git-ignored, travels to the GPU box with the snapshot).
# This is the drop-in demonstration for SURVEY.md section 8(b): reference application code,
# MI355X runtime.  It is NOT a parity oracle -- the reference's own runtime (OptiX) cannot be built
# here (needs nvcc and the closed OptiX SDK), so parity stays "unpinned" (oracle/trueknn_oracle.c).
set -euo pipefail
here=$(cd "$(dirname "$0")" && pwd)
root=$(dirname "$here")
ref=/root/reference/samples/s01-trueknn
out=$here/_ref
[ -d "$ref" ] || { echo "no reference tree at $ref: nothing to build"; exit 0; }
mkdir -p "$out"
python3 "$root/tools/owl_embed.py" ptxCode "$ref/deviceCode.cu" -o "$out/ptxCode.c" -I "$ref" --keep-hsaco "$out/deviceCode.hsaco"
inc="-I$root/include -I$root/include/owl_shims -I/opt/rocm/include -I$ref -D__HIP_PLATFORM_AMD__=1"
g++ -O2 -std=c++17 -w $inc -c "$ref/hostCode.cpp" -o "$out/hostCode.o"
gcc -O1 -c "$out/ptxCode.c" -o "$out/ptxCode.o"
g++ "$out/hostCode.o" "$out/ptxCode.o" -o "$out/sample01-trueknn" \
    -L"$root/owlraytracing_amd" -lowl_mi355x -L/opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,'$ORIGIN/../../owlraytracing_amd' -Wl,-rpath,/opt/rocm/lib
rm -f "$out/hostCode.o" "$out/ptxCode.o"
echo "built $out/sample01-trueknn"
