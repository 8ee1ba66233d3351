"""bench.py's round-5 legs, kept out of bench.py itself (which carries the headline and the round 1-4 legs):

  setup          P1, the setup path (tiny_api.cpp:21-190): end-to-end tinympc_setup_batch per BASELINE system, the Riccati loop's own
                 time inside k_precompute_rows, beside the reference's tiny_setup on one host core
  batched_tick   N2, the batched closed-loop tick (examples/cartpole_example_mpc.m:36-44 for B instances at once): tinympc_mpc_step_batch
                 at B = 1 ... 8,192, host x0 in / first controls out (PCIe-inclusive) and with the states resident in HBM, beside the
                 reference core running the same ticks on the host cores the box grants

and the statistics every latency comparison uses: `stats_us()` -- mean, median, p90, max of the SAME kind of sample on both sides.
GPU-side functions take the loaded package; CPU-side functions (`cpu_*`) never touch the GPU and run in bench.py's child processes."""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TICK_BATCHES = (1, 16, 256, 4096, 8192)
TICK_SETTINGS = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1)
TICKS, TICK_SKIP = 45, 5


def stats_us(samples) -> dict:
    """mean / median / p90 / max of per-event durations in microseconds (the same four on the GPU and the CPU side of a comparison)."""
    a = np.asarray(samples, dtype=np.float64)
    if a.size == 0:
        return {}
    return {"mean": float(a.mean()), "median": float(np.median(a)), "p90": float(np.percentile(a, 90)), "max": float(a.max()), "n": int(a.size)}


# ------------------------------------------------------------------------------------------------- CPU side (child processes)
def cpu_setup_and_ticks() -> dict:
    """One host core, the reference's own compiled core (oracle/_ref): tiny_setup per BASELINE system (box-constrained ones: the
    snapshot has no cones / fdyn) and the per-tick samples of the quadrotor closed loop (bench.py's `closed_loop_tick` leg)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import pyoracle as O  # checker / baseline only
    import __graft_entry__ as ge

    if not O.ref_available():
        return {}
    P = ge.load_package().problems
    out = {"setup_us": {}}
    for name, prob in (("cartpole", P.cartpole(20, True)), ("quadrotor", P.quadrotor(50))):
        out["setup_us"][name] = stats_us(O.OracleRef.bench_setup(prob, 60)[10:])
    prob = P.quadrotor(50)
    s = O.OracleRef(prob).load_problem(prob, dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1))
    runs = []
    for _ in range(5):  # (each run starts from the state the last one left: warm, like the GPU leg's handle)
        s.reset_workspace()
        its, sec, _x, per = s.bench_closed_loop_samples(prob.x0, 220, 20)
        runs.append((sec, its, per))
    sec, its, per = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    out["closed_loop_tick_us"] = stats_us(per)
    out["closed_loop_iterations_per_tick"] = its / 200
    return out


def cpu_ticks_worker(first: int, count: int, ticks: int, skip: int) -> dict:
    """`count` quadrotor instances (global indices first ...) ticked on ONE core by the reference core: total seconds of the counted
    ticks for all of them, iterations. (A child process of cpu_batched_ticks, pinned by the parent.)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import pyoracle as O  # checker / baseline only
    import __graft_entry__ as ge

    P = ge.load_package().problems
    prob = P.quadrotor(50)
    cls = O.OracleRef if O.ref_available() else None
    if cls is None or count < 1:
        return {"seconds": 0.0, "iterations": 0, "count": 0}
    solvers = [cls(prob).load_problem(prob, TICK_SETTINGS) for _ in range(count)]
    x0s = P.quadrotor_batch_x0(count, offset=first)
    its, per, _x = cls.bench_ticks_many(solvers, x0s, ticks, skip)
    return {"seconds": 1e-6 * float(per[skip:].sum()), "iterations": its, "count": count, "tick_us": [float(v) for v in per[skip:]]}


def cpu_batched_ticks(cpus, nworkers: int, bench_py: str) -> dict:
    """The batched closed-loop tick on the host: for every batch size, the instances are dealt to min(B, nworkers) single-threaded
    processes (one per granted CPU, pinned), each ticking its share with the reference core; a tick of the whole batch takes as long
    as the slowest process needs for its share (total / ticks: no barrier per tick -- this favours the host side if anything)."""
    out = {}

    def pin(cpu):
        def f():
            try:
                os.sched_setaffinity(0, {cpu})
            except (AttributeError, OSError):
                pass
        return f

    for B in TICK_BATCHES:
        n = min(B, nworkers)
        shares = [(B * i // n, B * (i + 1) // n - B * i // n) for i in range(n)]
        procs = [subprocess.Popen([sys.executable, bench_py, "--cpu-worker-ticks", "--worker-index", str(f), "--worker-count", str(c)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, preexec_fn=pin(cpus[i % len(cpus)]))
                 for i, (f, c) in enumerate(shares)]
        res = []
        for pr in procs:
            o, _ = pr.communicate()
            try:
                res.append(json.loads(o.strip().splitlines()[-1]))
            except (ValueError, IndexError):
                pass
        if len(res) != n:
            continue
        nt = TICKS - TICK_SKIP
        slowest = max(r["seconds"] for r in res)
        # per-tick samples of the batch = per tick, the maximum over the processes (as if they met at a barrier after every tick)
        per = np.max(np.array([r["tick_us"] for r in res]), axis=0) if all("tick_us" in r for r in res) else np.array([])
        out[str(B)] = {"us_per_tick": 1e6 * slowest / nt, "tick_us": stats_us(per), "processes": n,
                       "iterations_per_instance_tick": sum(r["iterations"] for r in res) / (B * nt),
                       "ticks_per_s": B * nt / slowest}
    return out


# ------------------------------------------------------------------------------------------------- GPU side
def setup_leg(pkg, cpu: dict | None) -> dict:
    """End-to-end tinympc_setup_batch (the C call: arenas, upload, k_precompute, wait) per BASELINE system, 40 setup / reset pairs
    after 5 discarded, and the Riccati loop's own duration inside the kernel (in-kernel 100 MHz counter)."""
    P, L, lib = pkg.problems, pkg._lib, pkg.load_library()
    out = {}
    p = lambda a: a.ctypes.data_as(L.c_double_p) if a is not None else None
    for name, prob in (("cartpole", P.cartpole(20, True)), ("quadrotor", P.quadrotor(50)), ("rocket", P.rocket(100))):
        A, B, Q, R = (np.asfortranarray(m, dtype=np.float64) for m in (prob.A, prob.B, prob.Q, prob.R))
        f = np.asfortranarray(prob.fdyn, dtype=np.float64) if prob.fdyn is not None else None
        calls, resets, loops, waits = [], [], [], []
        steps = 0
        for k in range(45):
            h = L.Handle()
            t0 = time.perf_counter()
            rc = lib.tinympc_setup_batch(C.byref(h), p(A), p(B), p(f), p(Q), p(R), prob.rho, prob.nx, prob.nu, prob.N, 1, -1, 0)
            t1 = time.perf_counter()
            L.check(rc)
            ph = np.zeros(10)
            lib.tinympc_debug_setup_timing(h, p(ph))
            t2 = time.perf_counter()
            lib.tinympc_reset(C.byref(h), 0)
            t3 = time.perf_counter()
            if k >= 5:
                calls.append(1e6 * (t1 - t0)); resets.append(1e6 * (t3 - t2)); loops.append(ph[8]); waits.append(ph[5])
            steps = int(ph[9])
        ref = ((cpu or {}).get("setup_us") or {}).get(name)
        out[name] = {"gpu_us": stats_us(calls), "gpu_ms": float(np.median(calls)) * 1e-3, "reset_us": stats_us(resets),
                     "riccati_steps": steps, "k_precompute_riccati_loop_us": float(np.median(loops)), "stream_wait_us": float(np.median(waits)),
                     "cpu_reference_us": ref, "cpu_reference_ms": (ref["median"] * 1e-3 if ref else None)}
    out["what"] = ("tinympc_setup_batch(batch 1) alone, wall clock around the C call: stream + arenas from the process-wide pools (the first setup of a "
                   "process creates them: +1.5 ms, and the HIP runtime's own start), one staged upload, one memset, k_fill_bounds, k_reset_stats, "
                   "k_precompute_rows, one wait; cpu_reference = the reference's own tiny_setup (oracle/_ref, tiny_api.cpp:21-122) in a C loop on one core; "
                   "rocket: the reference snapshot has no fdyn / cones, no CPU figure")
    return out


def batched_tick_leg(pkg, cpu: dict | None, dev_index: int) -> dict:
    """tinympc_mpc_step_batch per batch size: quadrotor N=50, warm start, tol 1e-3, max_iter 100, seeded x0 per instance, plant step
    x+ = A x + B u0 between ticks (outside the timed region on both sides). (a) host x0 in / first controls out (what a host-side
    simulator pays: PCIe inclusive); (b) states resident in HBM: set_x0_batch_device + solve, first controls left on the device."""
    import torch
    P = pkg.problems
    prob = P.quadrotor(50)
    dev = torch.device("cuda", dev_index)
    A_t, B_t = torch.from_numpy(prob.A).to(dev), torch.from_numpy(prob.B).to(dev)
    out, cross = {}, None
    cpu_ticks = (cpu or {}).get("batched_ticks") or {}
    for B in TICK_BATCHES:
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, device=dev_index, rho=prob.rho, **TICK_SETTINGS)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        x = np.asfortranarray(P.quadrotor_batch_x0(B))
        host, its = [], 0
        for k in range(TICKS):
            t0 = time.perf_counter()
            u0 = s.mpc_step(x)
            dt = time.perf_counter() - t0
            if k >= TICK_SKIP:
                host.append(1e6 * dt)
                its += int(np.sum(s.get_stats_batch()["iter"]))
            x = np.asfortranarray(prob.A @ x + prob.B @ u0)
        layout = s.launch_info()["layout"]
        # (b) device-resident: the same ticks again from the same start, cold state
        s.reset_workspace()
        xd = torch.from_numpy(np.ascontiguousarray(P.quadrotor_batch_x0(B).T)).to(dev)  # [B][nx]
        U = prob.nu * (prob.N - 1)
        dptr = C.c_void_p()
        pkg._lib.check(s._L.tinympc_get_solution_device_ptrs(s._h, None, C.byref(dptr)))
        resident = []
        for k in range(TICKS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.set_x0_batch(xd)
            s.solve()
            dt = time.perf_counter() - t0
            if k >= TICK_SKIP:
                resident.append(1e6 * dt)
            # first controls, still on the device: a strided view of sol_u [B][U]
            u_all = _device_view(dptr.value, B * U, dev).view(B, U)
            xd = (xd @ A_t.T + u_all[:, :prob.nu] @ B_t.T).contiguous()
        torch.cuda.synchronize()
        same_final = bool(np.allclose(xd.cpu().numpy().T, x, rtol=0, atol=1e-9))
        s.reset()
        nt = TICKS - TICK_SKIP
        c = cpu_ticks.get(str(B))
        row = {"layout": layout, "host_exchange_us": stats_us(host), "device_resident_us": stats_us(resident),
               "ticks_per_s_host_exchange": B * 1e6 / float(np.mean(host)), "ticks_per_s_device_resident": B * 1e6 / float(np.mean(resident)),
               "iterations_per_instance_tick": its / (B * nt), "device_resident_matches_host_path": same_final,
               "cpu_reference_us": (c or {}).get("tick_us"), "cpu_reference_us_per_tick_mean": (c or {}).get("us_per_tick"),
               "cpu_reference_processes": (c or {}).get("processes")}
        if c:
            row["gpu_over_cpu_host_exchange"] = c["us_per_tick"] / float(np.mean(host))
            if cross is None and float(np.mean(host)) < c["us_per_tick"]:
                cross = B
        out[str(B)] = row
    out["smallest_batch_where_gpu_wins"] = cross
    out["workload"] = ("quadrotor N=50, B instances, warm start, tol 1e-3, max_iter 100, %d ticks (%d untimed), seeded x0 per instance; host_exchange: "
                       "tinympc_mpc_step_batch with host x0 in / first controls out; device_resident: tinympc_set_x0_batch_device + tinympc_solve; "
                       "cpu_reference: the reference core on min(B, granted CPUs) pinned processes, mean = the slowest process's total / ticks; "
                       "statistics: microseconds per tick of the whole batch" % (TICKS, TICK_SKIP))
    return out


def _device_view(ptr: int, count: int, dev):
    """A torch float64 view of `count` doubles of device memory the library owns (no copy; valid until the handle is reset)."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=dev)
