#!/bin/bash
# Dev helper: build kernel variants of libtinympc_hip.so into tools/bin/ for A/B runs on the GPU box.
#   tools/build_variants.sh "name1:-DFLAG=1" "name2:-DFLAG=2" ...
# Run a variant with TINYMPC_HIP_LIBRARY=$PWD/tools/bin/libtinympc_hip_<name>.so (e.g. tools/layout_sweep.py).
# For source-level variants copy tinympc-matlab_amd/csrc/ to tools/bin/var_<name>/, edit the copy, and pass
# "name:-Itools/bin/var_<name>" with SRC=tools/bin/var_<name>.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  C="${SRC:-tinympc-matlab_amd/csrc}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -w -Iinclude -I$C $flags $C/*.hip -lhiprtc -o tools/bin/libtinympc_hip_$name.so
  echo "built tools/bin/libtinympc_hip_$name.so ($flags)"
done
