#!/usr/bin/env python3
"""gpurun_out/legs_<tag>/<leg>/{trace,sq,fetch,write}/*_results.db (tools/profile_legs.sh) -> profiles/<tag>_<leg>_stats.csv (per-kernel
durations of the leg's process) and profiles/<tag>_<leg>_pmc.json (counters of the leg's solve kernel, per launch, and what
follows from them: VALU / LDS / scalar instructions per wavefront-iteration, cycles per VALU instruction per SIMD, the shader
clock from GRBM_GUI_ACTIVE, HBM bytes with the gfx950 read-side correction, fractions of the FP64 and HBM roofs).

    python tools/collect_leg_profiles.py r03 [leg ...]"""
from __future__ import annotations

import glob
import json
import os
import sqlite3
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_FP64, PEAK_HBM = 78.6e12, 8.0e12
FLOPS = {"headline": 60848, "long_horizon": None, "adaptive_rho_batch": 60848}  # others: from the problem, below


def solve_kernel(names):
    ks = [n for n in names if "k_admm_solve" in n or "tinympc_jit_solve" in n or "k_builtin_" in n]
    return ks


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", "legs_" + tag)
    legs = sys.argv[2:] or sorted(os.listdir(base))
    P = os.path.join(ROOT, "profiles")
    for leg in legs:
        d = os.path.join(base, leg)
        if not os.path.isdir(d):
            continue
        plain = None
        try:
            plain = json.loads([l for l in open(os.path.join(d, "plain.json")) if l.startswith("{")][-1])
        except (OSError, IndexError, ValueError):
            pass
        sys.path.insert(0, ROOT)
        from bench import library_hash  # (the kernel sources these counters belong to: bench.py drops them when its library differs)
        out = {"leg": leg, "workload": plain, "library_hash": library_hash()}
        dbs = glob.glob(os.path.join(d, "trace", "**", "*_results.db"), recursive=True)
        kname = None
        if dbs:
            con = sqlite3.connect(dbs[0])
            rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
            with open(os.path.join(P, f"{tag}_{leg}_stats.csv"), "w") as o:
                o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
                for name, calls, total, avg, pct in rows:
                    o.write('"%s",%d,%.0f,%.0f,%.4f\n' % (name, calls, total * 1e3, avg * 1e3, pct))
            ks = [r for r in rows if "k_admm_solve" in r[0] or "tinympc_jit_solve" in r[0] or "k_builtin_" in r[0]]
            if ks:
                k = max(ks, key=lambda r: r[2])
                kname = k[0]
                out["kernel"] = {"name": kname, "calls": k[1], "average_ms_under_the_profiler": k[3] * 1e-3}
                q = "select duration, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, workgroup_x, grid_x from kernels where name = ? order by start"
                tr = list(con.execute(q, (kname,)))
                nlast = int((plain or {}).get("launches") or 0)
                if tr and nlast and len(tr) > nlast:  # (the measured launches: whatever came before them was the clock settling)
                    out["kernel"]["calls_warmup"] = len(tr) - nlast
                    tr = tr[-nlast:]
                    out["kernel"]["average_ms_under_the_profiler"] = sum(r[0] for r in tr) / len(tr) * 1e-6
                    out["kernel"]["calls"] = len(tr)
                if tr:
                    out["kernel"].update({"lds_bytes": tr[-1][1], "vgpr_count": tr[-1][2], "accum_vgpr_count": tr[-1][3], "sgpr_count": tr[-1][4],
                                          "workgroup": tr[-1][5], "grid": tr[-1][6], "durations_ns": [r[0] for r in tr]})
        counters = {}
        for sub in ("sq", "fetch", "write"):
            for db in glob.glob(os.path.join(d, sub, "**", "*_results.db"), recursive=True):
                con = sqlite3.connect(db)
                for name, cn, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
                    if kname and name == kname or (not kname and ("k_admm_solve" in name or "tinympc_jit_solve" in name or "k_builtin_" in name)):
                        counters.setdefault(cn, []).append(float(v))
        nlast = int((plain or {}).get("launches") or 0)
        if nlast:  # (per pass, the counters of its LAST `launches` dispatches of the kernel)
            counters = {k: (v[-nlast:] if len(v) > nlast else v) for k, v in counters.items()}
        c = {k: {"launches": len(v), "mean": st.mean(v)} for k, v in counters.items()}
        out["counters_per_launch"] = c
        if plain and "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
            iters, waves = plain["iterations_per_launch"], c["SQ_WAVES"]["mean"]
            dd = {"waves": waves, "valu_per_wave_iteration": c["SQ_INSTS_VALU"]["mean"] / waves / iters}
            for key, nm in (("SQ_INSTS_LDS", "lds_per_wave_iteration"), ("SQ_INSTS_SALU", "salu_per_wave_iteration")):
                if key in c:
                    dd[nm] = c[key]["mean"] / waves / iters
            if "SQ_WAVE_CYCLES" in c:  # quad-cycles (MI355X_MICROARCH.md)
                dd["cycles_per_wave_iteration"] = 4 * c["SQ_WAVE_CYCLES"]["mean"] / waves / iters
            if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
                dd["valu_active_fraction_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"]["mean"] / c["SQ_WAVE_CYCLES"]["mean"]
            if "SQ_BUSY_CYCLES" in c and "kernel" in out:
                dd["note_busy"] = "SQ_BUSY_CYCLES is summed over the shader engines"
            kms = plain["kernel_ms_median"]
            if "GRBM_GUI_ACTIVE" in c and "kernel" in out:
                prof_ms = out["kernel"]["average_ms_under_the_profiler"]
                dd["shader_clock_ghz_grbm"] = c["GRBM_GUI_ACTIVE"]["mean"] / 8 / (prof_ms * 1e-3) / 1e9
                dd["note_clock"] = "GRBM_GUI_ACTIVE / 8 XCDs / kernel duration under the profiler; reads high on dispatches under ~0.3 ms (MI355X_MICROARCH.md)"
            # VALU issue: instructions x 4 cycles (one wave64 instruction occupies the SIMD's 16 lanes for 4 cycles) / SIMD-cycles available
            if "kernel" in out and "shader_clock_ghz_grbm" in dd:
                simd_cycles = 1024 * dd["shader_clock_ghz_grbm"] * 1e9 * out["kernel"]["average_ms_under_the_profiler"] * 1e-3
                dd["valu_issue_fraction_of_chip"] = c["SQ_INSTS_VALU"]["mean"] * 4 / simd_cycles
            out["derived"] = dd
            out["throughput"] = {"iters_per_s": plain["iters_per_s"], "kernel_ms_median_unprofiled": kms}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and plain:
            rd, wr = 2.0 * c["FETCH_SIZE"]["mean"] * 1024.0, c["WRITE_SIZE"]["mean"] * 1024.0
            out["hbm"] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
                          "bytes_per_instance_iteration": (rd + wr) / plain["instances"] / plain["iterations_per_launch"],
                          "gbs_while_running": (rd + wr) / (plain["kernel_ms_median"] * 1e-3) / 1e9,
                          "frac_of_hbm_peak": (rd + wr) / (plain["kernel_ms_median"] * 1e-3) / PEAK_HBM,
                          "note": "read side = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md), KiB units"}
        json.dump(out, open(os.path.join(P, f"{tag}_{leg}_pmc.json"), "w"), indent=1)
        print("wrote profiles/%s_%s_*" % (tag, leg))


if __name__ == "__main__":
    main()
