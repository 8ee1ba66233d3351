#!/usr/bin/env python3
"""Copy the judged summaries out of rocprofv3's rocpd databases (gpurun_out/, scratch) into profiles/ (tracked).

    python tools/collect_profiles_db.py r01d --bench gpurun_out/bench_r01d.json --trace gpurun_out/prof_r01d/r01d_results.db \
        --fetch gpurun_out/pmc_fetch_r01d/f_results.db --write gpurun_out/pmc_write_r01d/w_results.db

Same outputs as tools/collect_profiles.py (which reads the CSV output format): <tag>_kernel_stats.csv,
<tag>_kernel_trace_admm.csv, <tag>_bench.json, <tag>_pmc.json and profiles/traffic_latest.json.
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are KiB and come from separate --pmc
passes; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so the read side is doubled."""
from __future__ import annotations

import argparse
import json
import os
import sqlite3
import statistics as st

import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import library_hash  # noqa: E402  (stamps the traffic figure with the kernel sources it was measured on)
HEADLINE = "k_admm_solve_d"  # the kernel bench.py times at 8,192 quadrotor instances (layout D)


def counter_means(db, counter):
    con = sqlite3.connect(db)
    out = {}
    for name, value in con.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        if "k_admm_solve" in name:
            out.setdefault(name, []).append(float(value))
    return {k: {"launches": len(v), "mean": st.mean(v), "min": min(v), "max": max(v)} for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--bench")
    ap.add_argument("--trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq", help="directory with one sub-directory per SQ counter pass (tools/gpu_profile.sh <tag> sq)")
    a = ap.parse_args()
    P = os.path.join(ROOT, "profiles")
    os.makedirs(P, exist_ok=True)
    if a.bench:
        lines = [l for l in open(a.bench) if l.startswith("{")]
        json.dump(json.loads(lines[-1]), open(os.path.join(P, f"{a.tag}_bench.json"), "w"), indent=1)
    if a.trace:
        con = sqlite3.connect(a.trace)
        with open(os.path.join(P, f"{a.tag}_kernel_stats.csv"), "w") as o:
            o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
            for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                o.write('"%s",%d,%.0f,%.0f,%.4f\n' % (name, calls, total * 1e3, avg * 1e3, pct))  # the view reports microseconds
        with open(os.path.join(P, f"{a.tag}_kernel_trace_admm.csv"), "w") as o:
            o.write("Kernel_Name,Duration_ns,LDS_Block_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Workgroup_Size_X,Grid_Size_X\n")
            q = ("select name, duration, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, workgroup_x, grid_x from kernels "
                 "where name like '%k_admm_solve%' order by start")
            for r in con.execute(q):
                o.write('"%s",%d,%d,%d,%d,%d,%d,%d\n' % r)
    pmc = {}
    if a.fetch:
        pmc["FETCH_SIZE_KiB"] = counter_means(a.fetch, "FETCH_SIZE")
    if a.write:
        pmc["WRITE_SIZE_KiB"] = counter_means(a.write, "WRITE_SIZE")
    if pmc:
        def headline(d):
            for k, v in d.items():
                if HEADLINE in k:
                    return v["mean"]
            return None
        f, w = headline(pmc.get("FETCH_SIZE_KiB", {})), headline(pmc.get("WRITE_SIZE_KiB", {}))
        if f is not None and w is not None:
            read_b, write_b = 2.0 * f * 1024.0, w * 1024.0
            pmc["headline_kernel"] = HEADLINE
            pmc["hbm_read_bytes_per_launch"] = read_b
            pmc["hbm_write_bytes_per_launch"] = write_b
            pmc["hbm_bytes_per_launch"] = read_b + write_b
            pmc["note"] = "read side = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md); per launch of 8,192 instances x 200 iterations"
            json.dump({"batch_per_gpu": 8192, "iters": 200, "horizon": 50, "hbm_bytes_per_launch": read_b + write_b,
                       "hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b,
                       "library_hash": library_hash(), "source": f"profiles/{a.tag}_pmc.json"},
                      open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
        json.dump(pmc, open(os.path.join(P, f"{a.tag}_pmc.json"), "w"), indent=1)
    if a.sq:
        import glob
        sq = {}
        for db in sorted(glob.glob(os.path.join(a.sq, "*", "*_results.db"))):
            con = sqlite3.connect(db)
            for name, cn, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
                if HEADLINE in name:
                    sq.setdefault(cn, []).append(float(v))
        c = {k: {"launches": len(v), "mean": st.mean(v)} for k, v in sq.items()}
        out = {"kernel": HEADLINE, "workload": "8,192 quadrotor N=50 instances x 200 iterations per launch", "counters": c}
        if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
            waves = c["SQ_WAVES"]["mean"]
            d = {"waves": waves, "valu_per_wave_iteration": c["SQ_INSTS_VALU"]["mean"] / waves / 200}
            for k, key in (("SQ_INSTS_LDS", "lds_per_wave_iteration"), ("SQ_INSTS_SALU", "salu_per_wave_iteration")):
                if k in c:
                    d[key] = c[k]["mean"] / waves / 200
            if "SQ_WAVE_CYCLES" in c:  # quad-cycles (MI355X_MICROARCH.md, PMC units)
                d["cycles_per_wave_iteration"] = 4 * c["SQ_WAVE_CYCLES"]["mean"] / waves / 200
                d["cycles_per_valu_per_simd_at_2_waves"] = d["cycles_per_wave_iteration"] / 2 / d["valu_per_wave_iteration"]
            out["derived"] = d
        json.dump(out, open(os.path.join(P, f"{a.tag}_pmc_sq.json"), "w"), indent=1)
    print("wrote profiles/%s_*" % a.tag)


if __name__ == "__main__":
    main()
