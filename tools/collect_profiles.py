#!/usr/bin/env python3
"""Copy the judged summaries out of gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/collect_profiles.py r01 --bench gpurun_out/bench_r1.json --trace gpurun_out/prof_r1 \
        --fetch gpurun_out/pmc_fetch_r1 --write gpurun_out/pmc_write_r1 [--sq gpurun_out/pmc_sq_r1]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_bench.json (the bench
line of the same command), <tag>_pmc.json (per-launch counter means for k_admm_solve) and refreshes
profiles/traffic_latest.json, which bench.py reads to fill roofline.traffic.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB and come from
separate --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so
the read side is doubled (calibrated here against the known state-load size: see the "expected" fields).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import statistics as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_admm_solve" in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean": st.mean(v), "min": min(v), "max": max(v)} for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--bench")
    ap.add_argument("--trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    a = ap.parse_args()
    P = os.path.join(ROOT, "profiles")
    os.makedirs(P, exist_ok=True)
    bench = None
    if a.bench:
        lines = [l for l in open(a.bench) if l.startswith("{")]
        bench = json.loads(lines[-1])
        json.dump(bench, open(os.path.join(P, f"{a.tag}_bench.json"), "w"), indent=1)
    if a.trace:
        for f in glob.glob(os.path.join(a.trace, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(P, f"{a.tag}_kernel_stats.csv"))
        for f in glob.glob(os.path.join(a.trace, "**", "*kernel_trace.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_admm_solve" in r["Kernel_Name"]]
            with open(os.path.join(P, f"{a.tag}_kernel_trace_admm.csv"), "w") as o:
                o.write("Kernel_Name,Duration_ns,LDS_Block_Size,VGPR_Count,SGPR_Count,Workgroup_Size_X,Grid_Size_X\n")
                for r in rows:
                    o.write(f"\"{r['Kernel_Name']}\",{int(r['End_Timestamp']) - int(r['Start_Timestamp'])},{r['LDS_Block_Size']},"
                            f"{r['VGPR_Count']},{r['SGPR_Count']},{r['Workgroup_Size_X']},{r['Grid_Size_X']}\n")
    pmc = {}
    for d in (a.fetch, a.write, a.sq):
        if d:
            pmc.update(counters(d))
    if pmc:
        summary = {"counters": pmc}
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            fetch_raw = pmc["FETCH_SIZE"]["mean"] * 1024
            write = pmc["WRITE_SIZE"]["mean"] * 1024
            summary["hbm_read_bytes_raw"] = fetch_raw
            summary["hbm_read_bytes_corrected_x2"] = 2 * fetch_raw
            summary["hbm_write_bytes"] = write
            summary["hbm_bytes_per_launch"] = 2 * fetch_raw + write
            if bench:
                cfg = bench["config"]
                B, it = cfg["batch_per_gpu"], cfg["iters_per_solve"]
                groups, N, nx, nu = (B + 3) // 4, 50, 12, 4
                summary["expected_state_load_bytes"] = groups * (2 * N * 64 + (N - 1) * 4 * nu) * 8 + B * nx * 8
                summary["expected_writeback_bytes"] = groups * (2 * N * 64 + (N - 1) * 4 * nu) * 8 + B * (nx * N + nu * (N - 1)) * 8
                summary["stale_v_stream_store_bytes_issued"] = groups * N * 64 * 8 * it
                summary["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
                json.dump({"batch_per_gpu": B, "iters": it, "horizon": N, "hbm_bytes_per_launch": summary["hbm_bytes_per_launch"],
                           "source": f"profiles/{a.tag}_pmc.json"}, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
        json.dump(summary, open(os.path.join(P, f"{a.tag}_pmc.json"), "w"), indent=1)
    print("profiles/:", sorted(os.listdir(P)))


if __name__ == "__main__":
    main()
