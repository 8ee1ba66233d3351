#!/bin/bash
# tools/gpu_profile.sh <tag> -- run ON THE GPU BOX (through gpurun) from the repo root:
#   bench.py (the driver's command), then the same workload under rocprofv3: kernel trace + stats, and the HBM byte
#   counters FETCH_SIZE / WRITE_SIZE in two separate --pmc passes (MI355X_MICROARCH.md: they do not fit one pass; a --pmc pass
#   carries --kernel-trace only -- never --sys-trace / --runtime-trace / the hip, hsa, memory-copy or marker domains, which
#   gpurun refuses next to counters). Everything lands in gpurun_out/; afterwards
#   `python tools/collect_profiles_db.py <tag> ...` copies the judged summaries into profiles/.
set -eo pipefail
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
python3 "$R/bench.py" > "$O/bench_$TAG.json" 2> "$O/bench_$TAG.err"
tail -c 600 "$O/bench_$TAG.json"
cd /tmp && export TMPDIR=/tmp
WL="--no-cpu-baseline --no-single --no-config5 --no-round5-legs --steps 20 --warmup 3"  # (the headline workload only: every launch of k_admm_solve_d in the trace is the SAME 8,192 x 200 solve)
rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG" -o "$TAG" -- python3 "$R/bench.py" $WL > "$O/prof_$TAG.log" 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/pmc_fetch_$TAG" -o f -- python3 "$R/bench.py" $WL --steps 5 > "$O/pmc_fetch_$TAG.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/pmc_write_$TAG" -o w -- python3 "$R/bench.py" $WL --steps 5 > "$O/pmc_write_$TAG.log" 2>&1
echo "write done"
if [ -n "$2" ]; then  # SQ counters of the headline kernel (instruction mix, occupancy), one pass each
  for C in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE; do
    rocprofv3 --pmc $C --kernel-trace -d "$O/pmc_sq_$TAG/$C" -o c -- python3 "$R/bench.py" $WL --steps 3 > "$O/pmc_sq_${TAG}_$C.log" 2>&1 || echo "$C failed"
  done
  echo "sq done"
fi
find "$O" -name "*_results.db" -newer "$O/bench_$TAG.json" | sort
