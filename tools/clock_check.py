#!/usr/bin/env python3
"""The shader clock the headline kernel really runs at (MI355X_MICROARCH.md, DVFS give-back item 6): a SEPARATE diagnostic build
of the library (-DTINY_CLOCK_STAMP: k_admm_solve_d stamps s_memtime / s_memrealtime once around its iteration loop, per wavefront,
into a buffer of its own; the product build contains no stamp) runs the bench workload -- 8,192 quadrotor N=50 instances x 200
iterations, cold start -- back to back for >= 2 s, then the last launch's stamps are read:
    clock = median over wavefronts of  delta s_memtime / delta s_memrealtime x 100 MHz.
Reconciles the three figures of round 2 (SQ_WAVE_CYCLES: 1.85, GRBM_GUI_ACTIVE / duration: 2.05, GRBM / 8 / un-profiled duration:
2.29 GHz) and prices the kernel against the clock it holds: FP64 issue fraction at that clock.

    python tools/clock_check.py build      (CPU container: hipcc -> tools/bin/libtinympc_hip_clock.so)
    python tools/clock_check.py run        (GPU box, through gpurun)  -> profiles/r03_clock.json"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "clock")
LIB = os.path.join(ROOT, "tools", "bin", "libtinympc_hip_clock.so")  # (git-ignored, but travels to the GPU box; build/ does not)


def build():
    import __graft_entry__ as ge
    ge.build_hip()  # the product's objects: everything but the two stamped sources is linked from there
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    objs = []
    for src in ge.HIP_SOURCES:
        if src in ("tinympc_solve_d.hip", "tinympc_capi.hip", "tinympc_plan.hip"):
            obj = os.path.join(OUT, src.replace(".hip", ".o"))
            extra = [a for a in sys.argv[2:] if a.startswith("-D")]  # (experiments: -DTINY_PRIO_SHIFT=.. -DTINY_PRIO_EVERY=.. -DTINY_PRIO_OFF)
            cmd = [hipcc] + ge.HIP_CFLAGS + ["-DTINY_CLOCK_STAMP=1"] + extra + ["-c", os.path.join(ge.CSRC, src), "-o", obj]
            print("+", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        else:
            obj = os.path.join(ge.BUILD_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
    cmd = [hipcc] + ge.HIP_LDFLAGS + objs + ["-o", LIB]
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def run():
    os.environ["TINYMPC_HIP_LIBRARY"] = LIB
    import numpy as np
    import __graft_entry__ as ge
    pkg = ge.load_package()
    P = pkg.problems
    prob = P.quadrotor(50)
    batch, iters = 8192, 200
    for a in sys.argv:
        if a.startswith("--batch="):
            batch = int(a.split("=")[1])
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(batch)))
    assert s.launch_info()["layout"] == "D"
    L = pkg.load_library()
    L.tinympc_debug_clock_stamps.restype = C.c_int
    L.tinympc_debug_clock_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    groups = (batch + 3) // 4
    rec = (C.c_ulonglong * (8 * groups))()
    samples = []
    if "--bench-like" in sys.argv:  # the driver's pattern: 5 warm-up + 20 timed cold-started launches, nothing before
        ms = []
        for _ in range(25):
            s.reset_workspace()
            ms.append(s.solve_timed())
        print("bench-like: warm-up", ["%.3f" % x for x in ms[:5]], "timed avg %.4f" % (sum(ms[5:]) / 20), "min %.3f max %.3f" % (min(ms[5:]), max(ms[5:])))
    t0 = time.perf_counter()
    launches, ms = 0, []
    while True:
        s.reset_workspace()
        ms.append(s.solve_timed())
        launches += 1
        el = time.perf_counter() - t0
        if el > 2.0 and launches % 50 == 0 or el > 8.0:
            n = L.tinympc_debug_clock_stamps(s._h, rec, groups)
            R = np.array(rec[:8 * n], dtype=np.float64).reshape(n, 8)
            c, t = R[:, 0], R[:, 1]
            ghz = 0.1 * c / t
            first_entry, last_exit = R[:, 5].min(), R[:, 6].max()
            if "--hwid" in sys.argv:  # where every wavefront ran (HW_REG_HW_ID: wave slot 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13; XCC_ID) and how long
                hw = np.array(rec[:8 * n], dtype=np.uint64).reshape(n, 8)[:, 7]
                rows = []
                for i in range(n):
                    h, x = int(hw[i]) & 0xffffffff, int(hw[i]) >> 32
                    rows.append((x & 15, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15, (h >> 4) & 3, h & 15, float(t[i]) / 100.0, i))
                rows.sort()
                with open(os.path.join(ROOT, "gpurun_out", "r03_clock_hwid.txt"), "w") as f:
                    f.write("xcc se sh cu simd slot loop_us wave\n")
                    for r_ in rows:
                        f.write("%d %d %d %d %d %d %.1f %d\n" % r_)
            samples.append(dict(after_s=el, launches=launches, clock_ghz_median=float(np.median(ghz)), clock_ghz_min=float(ghz.min()),
                                clock_ghz_max=float(ghz.max()), loop_cycles_median=float(np.median(c)), loop_us_median=float(np.median(t)) / 100.0,
                                kernel_ms_last=ms[-1],
                                phases_us=dict(entry_to_barrier_median=float(np.median(R[:, 2])) / 100.0, barrier_to_loop_median=float(np.median(R[:, 3])) / 100.0,
                                               loop_median=float(np.median(t)) / 100.0, loop_max=float(t.max()) / 100.0, loop_min=float(t.min()) / 100.0,
                                               after_loop_median=float(np.median(R[:, 4])) / 100.0,
                                               first_entry_to_last_exit=float(last_exit - first_entry) / 100.0,
                                               entry_spread=float(R[:, 5].max() - first_entry) / 100.0, exit_spread=float(last_exit - R[:, 6].min()) / 100.0)))
            if len(samples) >= 5 or el > 8.0:
                break
    s.reset()
    last = samples[-1]
    kernel_ms = float(np.median(ms[len(ms) // 2:]))
    valu_per_wave_iter = None
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "r03_headline_pmc.json")))
        valu_per_wave_iter = sq["derived"]["valu_per_wave_iteration"]
        sq_cycles = sq["derived"]["cycles_per_wave_iteration"]
    except (OSError, KeyError, ValueError):
        sq_cycles = None
    clock = float(np.median([x["clock_ghz_median"] for x in samples]))
    flops = batch * iters * prob.flops_per_iteration()
    out = {
        "what": "in-kernel shader clock of k_admm_solve_d<12,4,50> (diagnostic build, stamps around the iteration loop), bench workload, "
                ">= 2 s of back-to-back cold-started launches",
        "samples": samples, "clock_ghz": clock,
        "kernel_ms_median_hip_events_stamped_build": kernel_ms,
        "loop_cycles_per_wave_iteration": last["loop_cycles_median"] / iters,
        "sq_wave_cycles_per_wave_iteration_profiled": sq_cycles,
        "valu_per_wave_iteration": valu_per_wave_iter,
        "cycles_per_valu_per_simd_two_waves": (last["loop_cycles_median"] / iters / 2 / valu_per_wave_iter) if valu_per_wave_iter else None,
        "fp64_peak_at_this_clock_tflops": 256 * 4 * 32 * clock * 1e9 / 1e12,
        "fp64_frac_at_this_clock": flops / (kernel_ms * 1e-3) / (256 * 4 * 32 * clock * 1e9),
        "fp64_frac_at_nominal_2p4": flops / (kernel_ms * 1e-3) / 78.6e12,
        "note": "loop time covers the iterations only (prologue state load and the final write-back are outside the stamps): "
                "kernel duration - loop time = the launch edge",
        "launch_edge_ms": kernel_ms - last["loop_us_median"] * 1e-3,
    }
    if batch == 8192 and "--no-save" not in sys.argv:
        json.dump(out, open(os.path.join(ROOT, "profiles", "r03_clock.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
