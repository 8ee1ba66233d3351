#!/bin/bash
# Dev helper: build kernel variants of libtinympc_hip.so into tools/bin/ for A/B runs on the GPU box.
#   tools/build_variants.sh "name1:-DFLAG=1" "name2:-DFLAG=2" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
C=tinympc-matlab_amd/csrc
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Iinclude -I$C $flags \
      $C/tinympc_kernels.hip $C/tinympc_solve.hip $C/tinympc_solve_b.hip $C/tinympc_solve_fam.hip $C/tinympc_capi.hip -o tools/bin/libtinympc_hip_$name.so 2>&1 | grep -v "warning\|PRE_LDS\|\^~" || true
  echo "built tools/bin/libtinympc_hip_$name.so ($flags)"
done
