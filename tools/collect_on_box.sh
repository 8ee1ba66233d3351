#!/bin/bash
# tools/collect_on_box.sh <tag> -- run ON THE GPU BOX (through gpurun): tools/gpu_profile.sh <tag> sq, then the collectors right there, and only the
# summaries travel back (gpurun merges at most 64 MiB of gpurun_out/; the rocpd databases of a headline run are larger).
#   gpurun_out/<tag>_headline_out/  <-  profiles/<tag>_bench.json, _kernel_stats.csv, _kernel_trace_admm.csv, _pmc.json, _pmc_sq.json, traffic_latest.json
set -o pipefail
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
bash "$R/tools/gpu_profile.sh" "$TAG" sq > "$O/${TAG}_gpu_profile.log" 2>&1
cd "$R"
T=$(find "$O/prof_$TAG" -name "*_results.db" | head -1)
F=$(find "$O/pmc_fetch_$TAG" -name "*_results.db" | head -1)
W=$(find "$O/pmc_write_$TAG" -name "*_results.db" | head -1)
python3 tools/collect_profiles_db.py "$TAG" --bench "$O/bench_$TAG.json" --trace "$T" --fetch "$F" --write "$W" --sq "$O/pmc_sq_$TAG" > "$O/${TAG}_collect_headline.log" 2>&1
mkdir -p "$O/${TAG}_headline_out"
cp profiles/${TAG}_bench.json profiles/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_trace_admm.csv profiles/${TAG}_pmc.json profiles/${TAG}_pmc_sq.json profiles/traffic_latest.json "$O/${TAG}_headline_out/" 2>/dev/null
rm -rf "$O/prof_$TAG" "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG" "$O/pmc_sq_$TAG"
tail -3 "$O/${TAG}_collect_headline.log"; ls "$O/${TAG}_headline_out"
