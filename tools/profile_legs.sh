#!/bin/bash
# tools/profile_legs.sh <tag> [leg ...] -- run ON THE GPU BOX (through gpurun) from the repo root: every bench leg as its own
# process (tools/leg_workload.py) under rocprofv3, four passes each --
#   1. --kernel-trace --stats                      per-kernel durations
#   2. --pmc <8 SQ counters> (one pass: the SQ block has 8 slots) + GRBM_GUI_ACTIVE (GRBM: its own slots)
#   3. --pmc FETCH_SIZE     4. --pmc WRITE_SIZE    (TCC: they do not fit one pass; MI355X_MICROARCH.md)
# Counter passes carry --kernel-trace only (no other trace domain). Databases land in gpurun_out/legs_<tag>/<leg>/;
# `python tools/collect_leg_profiles.py <tag>` turns them into profiles/<tag>_<leg>_stats.csv / _pmc.json.
set -o pipefail
TAG=${1:?tag}; shift
LEGS=${@:-headline rocket_batch rocket_batch_n10 rocket_instance wide_system wide_families long_horizon large_system very_large_system adaptive_rho_batch single_instance}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/legs_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for L in $LEGS; do
  mkdir -p "$O/$L"
  python3 "$R/tools/leg_workload.py" $L 8 > "$O/$L/plain.json" 2> "$O/$L/plain.err" || { echo "$L: workload failed"; tail -3 "$O/$L/plain.err"; continue; }
  cat "$O/$L/plain.json"
  rocprofv3 --kernel-trace --stats -d "$O/$L/trace" -o t -- python3 "$R/tools/leg_workload.py" $L 8 > "$O/$L/trace.log" 2>&1 || echo "$L trace failed"
  rocprofv3 --pmc $SQ --kernel-trace -d "$O/$L/sq" -o s -- python3 "$R/tools/leg_workload.py" $L 4 > "$O/$L/sq.log" 2>&1 || echo "$L sq failed"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/$L/fetch" -o f -- python3 "$R/tools/leg_workload.py" $L 4 > "$O/$L/fetch.log" 2>&1 || echo "$L fetch failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/$L/write" -o w -- python3 "$R/tools/leg_workload.py" $L 4 > "$O/$L/write.log" 2>&1 || echo "$L write failed"
  echo "$L done"
done
find "$O" -name "*_results.db" | wc -l
