#!/bin/bash
# Dev helper: variants of the large-system kernel only (tinympc_solve_m.hip), linked against the objects of the last build().
#   tools/build_m_variants.sh "name1:-DTINY_EXP_M=1" ...   ->  tools/bin/libtinympc_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
OBJS=$(ls build/hip/*.o | grep -v tinympc_solve_m | grep -v amdgcn | grep -v asan | grep -v "solve_e.o\|solve_f.o")
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -w -Iinclude $flags -c tinympc-matlab_amd/csrc/tinympc_solve_m.hip -o tools/bin/m_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -lhiprtc $OBJS tools/bin/m_$name.o -o tools/bin/libtinympc_hip_$name.so
  echo "built tools/bin/libtinympc_hip_$name.so ($flags)"
done
