"""bench.py's legs, kept out of bench.py itself (which carries the headline, the contract of the JSON line and the CPU baseline):

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
    # ... and the reference's own closed-loop example shape (examples/interactive_cartpole.m: cartpole, N = 10, |u| <= 5 there; here the bounded
    # cartpole of problems.py), same loop
    cp = P.cartpole(10, True)
    s = O.OracleRef(cp).load_problem(cp, dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100, check_termination=1))
    runs = []
    for _ in range(5):
        s.reset_workspace()
        its, sec, _x, per = s.bench_closed_loop_samples(cp.x0, 220, 20)
        runs.append((sec, its, per))
    sec, its, per = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    out["cartpole10_closed_loop_tick_us"] = stats_us(per)
    out["cartpole10_closed_loop_iterations_per_tick"] = its / 200
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


# ================================================================================================= rounds 1-4 legs (moved out of bench.py in round 5)
# Each function fills `out` (the JSON line's dict) with its legs. `ctx`: what bench.py's main() already has -- the package, the quadrotor
# problem, the device, the parsed arguments, the CPU baseline's numbers, torch / numpy, the headline's flop and byte models.
PEAK_HBM_GBS = 8000.0
PEAK_FP64_TFLOPS = 78.6


class LegContext:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _unpack(ctx):
    return (ctx.pkg, ctx.pkg.problems, ctx.prob, ctx.args, ctx.cpu, ctx.dev_index, ctx.dev, ctx.torch, ctx.np, ctx.out, ctx.flops_iter, ctx.bytes_iter,
            ctx.sol, ctx.parity, ctx.leg_counters)

def config5_legs(ctx) -> None:
    """BASELINE config 5 on ONE GPU (65,536 instances, forced iterations) and the same batch converging (slot refill against plain)."""
    pkg, P, prob, args, cpu, dev_index, dev, torch, np, out, flops_iter, bytes_iter, sol, parity, leg_counters = _unpack(ctx)
    # BASELINE config 5 on ONE GPU: all 65,536 instances in one launch (8 waves per CU x 8 rounds)
    nb = 65536
    big = pkg.TinyMPC()
    big.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=nb, device=dev_index, rho=prob.rho,
              abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=args.iters, check_termination=1)
    big.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    big.set_x0_batch(torch.from_numpy(np.ascontiguousarray(P.quadrotor_batch_x0(nb).T)).to(dev))
    big.synchronize()
    big.reset_workspace()
    big.solve_timed()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    kms = []
    for _ in range(3):
        big.reset_workspace()
        kms.append(big.solve_timed())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    u_first = big.get_solution_batch(0, 64)["controls"]
    out["config5_single_gpu"] = {"workload": "65,536 quadrotor N=%d instances x %d forced iterations on ONE GPU, 3 cold-started steps" % (prob.N, args.iters),
                                 "value": 3 * nb * args.iters / dt, "unit": "ADMM iters/s", "ms_per_step": 1e3 * dt / 3,
                                 "kernel_ms_avg": sum(kms) / 3, "layout": big.launch_info()["layout"],
                                 "fp64_frac": nb * args.iters * flops_iter / (sum(kms) / 3 * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                                 "prefix_matches_8192_run": (bool(np.array_equal(u_first, sol["controls"])) if parity else None),
                                 "kernel": big.jit_info()}
    big.reset()
    # The same 65,536 instances as a CONVERGING batch (tol 1e-3, x0 scaled 0.05 ... 3: 5 to 200 iterations per instance):
    # the plain kernel holds a wavefront until its slowest instance is done; with slot refill (tinympc_solve_d.hip) a
    # row takes the next instance as soon as its own has finished. Same results bit for bit (tests/test_slot_refill_gpu.py).
    rng = np.random.default_rng(0)
    x0c = np.ascontiguousarray((P.quadrotor_batch_x0(nb) * rng.uniform(0.05, 3.0, nb)[None, :]).T)
    conv = {}
    for mode in ("0", None):
        if mode is None:
            os.environ.pop("TINYMPC_REFILL", None)
        else:
            os.environ["TINYMPC_REFILL"] = mode
        cb = pkg.TinyMPC()
        cb.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=nb, device=dev_index, rho=prob.rho,
                 abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=args.iters, check_termination=1)
        cb.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        cb.set_x0_batch(torch.from_numpy(x0c).to(dev))
        kms = []
        for _ in range(4):
            cb.reset_workspace()
            kms.append(cb.solve_timed())
        its = cb.get_stats_batch()["iter"].astype(np.float64)
        conv["plain" if mode == "0" else "default"] = {"kernel_ms": float(np.median(kms[1:])), "kernel": cb.jit_info(), "instance_iterations": float(its.sum())}
        cb.reset()
    os.environ.pop("TINYMPC_REFILL", None)
    forced_rate = nb * args.iters / (out["config5_single_gpu"]["kernel_ms_avg"] * 1e-3)  # instance-iterations/s with every row busy
    d = conv["default"]
    out["converging_batch"] = {"workload": "65,536 quadrotor N=%d instances, tol 1e-3, max_iter %d, x0 scaled 0.05 ... 3 (mean %.0f iterations per instance)"
                                           % (prob.N, args.iters, d["instance_iterations"] / nb),
                               "kernel_ms": d["kernel_ms"], "kernel": d["kernel"], "iters_per_s": d["instance_iterations"] / (d["kernel_ms"] * 1e-3),
                               "fraction_of_forced_iteration_rate": d["instance_iterations"] / (d["kernel_ms"] * 1e-3) / forced_rate,
                               "plain_kernel_ms": conv["plain"]["kernel_ms"],
                               "same_iteration_total": conv["plain"]["instance_iterations"] == d["instance_iterations"]}

def instance_and_family_legs(ctx) -> None:
    """One instance (quadrotor, rocket landing), rocket-landing batches on layout E, adaptive rho on a batch."""
    pkg, P, prob, args, cpu, dev_index, dev, torch, np, out, flops_iter, bytes_iter, sol, parity, leg_counters = _unpack(ctx)
    one = pkg.TinyMPC()
    one.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, device=dev_index, rho=prob.rho,
              abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=args.iters, check_termination=1)
    one.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    one.set_x0(prob.x0)
    ms = []
    for k in range(13):
        one.reset_workspace()
        ms.append(one.solve_timed())
    ms = sorted(ms[3:])
    med = ms[len(ms) // 2]
    out["single_instance"] = {"iters_per_s": args.iters / (med * 1e-3), "us_per_iter": 1e3 * med / args.iters,
                              "kernel_ms": med, "roofline_frac": args.iters * bytes_iter / (med * 1e-3) / 1e9 / PEAK_HBM_GBS,
                              "layout": one.launch_info()["layout"], **leg_counters("single_instance", args.iters / (med * 1e-3))}
    one.reset()
    # BASELINE config 4: one rocket-landing instance, N=100, second-order cones + a linear row + fdyn
    rk = P.rocket(100)
    one = pkg.TinyMPC()
    one.setup(rk.A, rk.B, rk.Q, rk.R, rk.N, batch=1, device=dev_index, rho=rk.rho, fdyn=rk.fdyn,
              abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=args.iters, check_termination=1)
    one.set_bound_constraints(rk.x_min, rk.x_max, rk.u_min, rk.u_max)
    if rk.x_ref is not None:
        one.set_x_ref(rk.x_ref)
    if rk.u_ref is not None:
        one.set_u_ref(rk.u_ref)
    one.set_cone_constraints(**rk.cones)
    if rk.linear:
        one.set_linear_constraints(**rk.linear)
    one.set_x0(rk.x0)
    ms = []
    for k in range(9):
        one.reset_workspace()
        ms.append(one.solve_timed())
    med = sorted(ms[2:])[len(ms[2:]) // 2]
    out["rocket_instance"] = {"workload": "rocket landing nx=6 nu=3 N=100, state + input cones, 1 linear row, fdyn, %d forced iterations" % args.iters,
                              "us_per_iter": 1e3 * med / args.iters, "kernel_ms": med, "layout": one.launch_info()["layout"],
                              "flops_per_instance_iteration": rk.flops_per_iteration_families()["with_box"],
                              "flops_model": "problems.flops_per_iteration_families(): SURVEY section 8a's box-path formula + the cone / half-space projections, "
                                             "their slack / dual / linear-cost terms and fdyn, counted on oracle/tinympc_oracle.c:348-470",
                              "fp64_frac": args.iters * rk.flops_per_iteration_families()["with_box"] / (med * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                              "cpu_port_us_per_iter_single_process": cpu.get("rocket_us_per_iter_single_process") if cpu else None,
                              **leg_counters("rocket_instance", args.iters / (med * 1e-3))}
    one.reset()
    # ... and batches of it: N=100 (BASELINE config 4's horizon: the latency kernel, one workgroup per instance) and
    # N=10 (the horizon of examples/rocket_landing_constraints.m:14: layout D with the families in registers)
    rb = {}
    for rN in (100, 10):
        rkb = P.rocket(rN)
        rB, rit = 4096, 100
        many = pkg.TinyMPC()
        many.setup(rkb.A, rkb.B, rkb.Q, rkb.R, rkb.N, batch=rB, device=dev_index, rho=rkb.rho, fdyn=rkb.fdyn,
                   abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=rit, check_termination=1)
        many.set_bound_constraints(rkb.x_min, rkb.x_max, rkb.u_min, rkb.u_max)
        many.set_x_ref(rkb.x_ref)
        many.set_u_ref(rkb.u_ref)
        many.set_cone_constraints(**rkb.cones)
        many.set_linear_constraints(**rkb.linear)
        many.set_x0_batch(np.asfortranarray(rkb.x0[:, None] * np.linspace(0.6, 1.2, rB)[None, :]))
        ms = []
        for k in range(5):
            many.reset_workspace()
            ms.append(many.solve_timed())
        med = sorted(ms[1:])[len(ms[1:]) // 2]
        rb["N=%d" % rN] = {"iters_per_s": rB * rit / (med * 1e-3), "kernel_ms": med, "layout": many.launch_info()["layout"], "jit": many.jit_info(),
                           "box_part_fp64_frac": rB * rit * rkb.flops_per_iteration() / (med * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                           "flops_per_instance_iteration": rkb.flops_per_iteration_families()["with_box"],
                           "fp64_frac": rB * rit * rkb.flops_per_iteration_families()["with_box"] / (med * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                           **(leg_counters("rocket_batch", rB * rit / (med * 1e-3)) if rN == 100 else {})}
        many.reset()
    out["rocket_batch"] = dict(workload="4096 rocket-landing instances (cones + linear row + fdyn) x 100 forced iterations", **rb)
    # Adaptive rho (admm.cpp:117-174) on a batch: rho, its operator rows and pNref per instance, layout D's ADAPT variant
    ad = pkg.TinyMPC()
    aB, ait = 8192, 100
    ad.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=aB, device=dev_index, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0,
             max_iter=ait, adaptive_rho=True, adaptive_rho_min=0.2, adaptive_rho_max=40.0)
    ad.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    ad.set_sensitivity_matrices(*ad.compute_sensitivity_autograd())
    ad.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(aB)))
    ms = []
    for k in range(5):
        ad.reset_workspace()
        ms.append(ad.solve_timed())
    med = sorted(ms[1:])[len(ms[1:]) // 2]
    out["adaptive_rho_batch"] = {"workload": "quadrotor N=%d, %d instances x %d forced iterations, rho adapted every 5th" % (prob.N, aB, ait),
                                 "iters_per_s": aB * ait / (med * 1e-3), "kernel_ms": med, "layout": ad.launch_info()["layout"],
                                 "fp64_frac_box_part": aB * ait * flops_iter / (med * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                                 "rho_spread": [float(np.min(ad.get_rho_batch())), float(np.max(ad.get_rho_batch()))],
                                 **leg_counters("adaptive_rho_batch", aB * ait / (med * 1e-3))}
    ad.reset()

def shape_and_latency_legs(ctx) -> None:
    """Wide / long / large / very large systems, the cartpole (BASELINE config 1), the closed-loop tick (launched, resident, the reference's three verbs) and the rocket landing's own closed loop."""
    pkg, P, prob, args, cpu, dev_index, dev, torch, np, out, flops_iter, bytes_iter, sol, parity, leg_counters = _unpack(ctx)
    # Wide systems (16 < nx+nu <= 64: dynamic sizes in the reference, types.hpp:16-17): 32 lanes per instance,
    # cross-row swaps + fused DPP chain. Synthetic stable system, box constraints, 100 forced iterations.
    rng = np.random.default_rng(0)
    wnx, wnu, wN, wB, wit = 24, 8, 30, 4096, 100
    wA = np.eye(wnx) + 0.03 * rng.standard_normal((wnx, wnx))
    wBm = 0.1 * rng.standard_normal((wnx, wnu))
    wp = P.Problem("wide", wA, wBm, np.diag(rng.uniform(1, 10, wnx)), np.diag(rng.uniform(0.5, 2, wnu)), wN, 2.0, rng.standard_normal(wnx))
    wide = pkg.TinyMPC()
    wide.setup(wp.A, wp.B, wp.Q, wp.R, wp.N, batch=wB, device=dev_index, rho=wp.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=wit)
    wide.set_bound_constraints(np.full(wnx, -2.0), np.full(wnx, 2.0), np.full(wnu, -0.3), np.full(wnu, 0.3))
    wide.set_x0_batch(np.asfortranarray(np.random.default_rng(1).standard_normal((wnx, wB))))
    ms = []
    for k in range(6):
        wide.reset_workspace()
        ms.append(wide.solve_timed())
    med = sorted(ms[1:])[len(ms[1:]) // 2]
    wtf = wB * wit * wp.flops_per_iteration() / (med * 1e-3) / 1e12
    out["wide_system"] = {"workload": "synthetic nx=%d nu=%d N=%d, box constraints, %d instances x %d forced iterations" % (wnx, wnu, wN, wB, wit),
                          "iters_per_s": wB * wit / (med * 1e-3), "kernel_ms": med, "fp64_tflops": wtf, "fp64_frac": wtf / PEAK_FP64_TFLOPS,
                          "lanes_per_instance": wide.launch_info()["lanes_per_instance"], "layout": wide.launch_info()["layout"],
                          **leg_counters("wide_system", wB * wit / (med * 1e-3))}
    # ... and the same system with one state cone and two linear rows on the states (round 5: the families STREAMED next to layout D's wide sweeps,
    # tinympc_solve_dwide.h; round 4: k_admm_solve_fam, layout A)
    wide.set_cone_constraints(Acx=[0], qcx=[3], cx=[0.7], Acu=[], qcu=[], cu=[])
    wide.set_linear_constraints(Alin_x=np.random.default_rng(1).standard_normal((2, wnx)), blin_x=np.array([1.0, 1.5]), Alin_u=np.zeros((0, wnu)), blin_u=np.zeros(0))
    fms = []
    for k in range(6):
        wide.reset_workspace()
        fms.append(wide.solve_timed())
    fmed = sorted(fms[1:])[2]
    out["wide_system"]["with_families"] = {"workload": "+ one state cone, two linear rows on the states", "kernel_ms": fmed, "iters_per_s": wB * wit / (fmed * 1e-3),
                                           "layout": wide.launch_info()["layout"], "jit": wide.jit_info(), "fraction_of_box_rate": med / fmed}
    wide.reset()
    # Long horizon: the quadrotor at N = 100. The duals of 99 knots do not fit 256 registers, so layout D runs its second
    # plan (one wavefront per SIMD with all 512 registers; the kernel is specialised at run time by tinympc_jit.hip).
    hp = P.quadrotor(100)
    hB, hit = 8192, 100
    longh = pkg.TinyMPC()
    longh.setup(hp.A, hp.B, hp.Q, hp.R, hp.N, batch=hB, device=dev_index, rho=hp.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=hit)
    longh.set_bound_constraints(hp.x_min, hp.x_max, hp.u_min, hp.u_max)
    longh.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(hB)))
    ms = []
    for k in range(5):
        longh.reset_workspace()
        ms.append(longh.solve_timed())
    med = sorted(ms[1:])[len(ms[1:]) // 2]
    htf = hB * hit * hp.flops_per_iteration() / (med * 1e-3) / 1e12
    out["long_horizon"] = {"workload": "quadrotor N=100, box constraints, %d instances x %d forced iterations" % (hB, hit),
                           "iters_per_s": hB * hit / (med * 1e-3), "kernel_ms": med, "fp64_tflops": htf, "fp64_frac": htf / PEAK_FP64_TFLOPS,
                           "layout": longh.launch_info()["layout"], "workgroups": longh.launch_info()["workgroups"],
                           **leg_counters("long_horizon", hB * hit / (med * 1e-3))}
    longh.reset()
    # Large systems (64 < nx+nu <= 512): 16 instances per tile on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), the
    # state streaming through HBM -- the north_star's "MFMA when nx is large enough" clause. HBM-bound: priced on both roofs.
    lnx, lnu, lN, lB, lit = 96, 32, 20, 4096, 50
    rng = np.random.default_rng(lnx)
    lA = np.eye(lnx) * 0.98 + 0.015 * rng.standard_normal((lnx, lnx))
    lBm = 0.08 * rng.standard_normal((lnx, lnu))
    lp = P.Problem("large", lA, lBm, np.diag(rng.uniform(1, 10, lnx)), np.diag(rng.uniform(0.5, 2, lnu)), lN, 2.0, rng.standard_normal(lnx))
    big = pkg.TinyMPC()
    big.setup(lp.A, lp.B, lp.Q, lp.R, lp.N, batch=lB, device=dev_index, rho=lp.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=lit)
    big.set_bound_constraints(np.full(lnx, -2.0), np.full(lnx, 2.0), np.full(lnu, -0.3), np.full(lnu, 0.3))
    big.set_x0_batch(np.asfortranarray(np.random.default_rng(1).standard_normal((lnx, lB))))
    ms = []
    for k in range(4):
        big.reset_workspace()
        ms.append(big.solve_timed())
    med = sorted(ms[1:])[1]
    ltf = lB * lit * lp.flops_per_iteration() / (med * 1e-3) / 1e12
    lgb = lB * lit * lp.bytes_per_iteration() / (med * 1e-3) / 1e9
    lmeas = leg_counters("large_system", lB * lit / (med * 1e-3))
    out["large_system"] = {"workload": "synthetic nx=%d nu=%d N=%d, box constraints, %d instances x %d forced iterations" % (lnx, lnu, lN, lB, lit),
                           "iters_per_s": lB * lit / (med * 1e-3), "kernel_ms": med, "fp64_tflops": ltf, "fp64_frac": ltf / PEAK_FP64_TFLOPS,
                           "hbm_measured_gbs": (lmeas.get("measured") or {}).get("hbm_measured_gbs"),
                           "hbm_measured_frac": (lmeas.get("measured") or {}).get("hbm_measured_frac"),
                           "hbm_model_gbs": lgb, "layout": big.launch_info()["layout"],
                           "hbm_note": "hbm_measured_*: the kernel's PMC-measured traffic per instance and iteration (FETCH_SIZE x 2 + WRITE_SIZE) at "
                                       "this run's rate; hbm_model_gbs: SURVEY.md section 8d's streaming model (9-11 accesses per element), of which "
                                       "the kernel moves about 2/3",
                           "kernel": "k_admm_solve_m (v_mfma_f64_16x16x4_f64, 16 instances per tile)", **lmeas}
    # ... and the same system with a state cone, an input cone and two linear rows per side (round 4: the families' phase of
    # layout M between the sweeps, one knot per wavefront; HBM-bound at a multiple of the box path's bytes)
    frng = np.random.default_rng(5)
    big.set_cone_constraints(Acx=[0, 40], qcx=[3, 6], cx=[0.8, 0.6], Acu=[0], qcu=[3], cu=[0.7])
    big.set_linear_constraints(Alin_x=frng.standard_normal((2, lnx)) / np.sqrt(lnx), blin_x=np.array([0.3, 0.4]),
                               Alin_u=frng.standard_normal((2, lnu)) / np.sqrt(lnu), blin_u=np.array([0.2, 0.25]))
    fms = []
    for k in range(4):
        big.reset_workspace()
        fms.append(big.solve_timed())
    fmed = sorted(fms[1:])[1]
    out["large_system"]["with_families"] = {"workload": "+ 2 state cones, 1 input cone, 2 linear rows per side", "iters_per_s": lB * lit / (fmed * 1e-3),
                                            "kernel_ms": fmed, "layout": big.launch_info()["layout"], "fraction_of_box_rate": med / fmed}
    big.reset()
    # ... and beyond 256 rows (round 3): four row tiles per wavefront, the operator tiles streamed from a tile-major copy in L2;
    # with nxu / 20 flop per byte of state this one is priced on the matrix pipe
    vnx, vnu, vN, vB, vit = 480, 32, 20, 4096, 20
    rng = np.random.default_rng(7)
    vA = 0.6 * np.eye(vnx) + (0.1 / np.sqrt(vnx)) * rng.standard_normal((vnx, vnx))
    vp = P.Problem("very_large", vA, 0.08 * rng.standard_normal((vnx, vnu)), np.diag(rng.uniform(1, 10, vnx)), np.diag(rng.uniform(0.5, 2, vnu)), vN, 2.0,
                   rng.standard_normal(vnx))
    vbig = pkg.TinyMPC()
    vbig.setup(vp.A, vp.B, vp.Q, vp.R, vp.N, batch=vB, device=dev_index, rho=vp.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=vit)
    vbig.set_bound_constraints(np.full(vnx, -2.0), np.full(vnx, 2.0), np.full(vnu, -0.3), np.full(vnu, 0.3))
    vbig.set_x0_batch(np.asfortranarray(np.random.default_rng(1).standard_normal((vnx, vB))))
    ms = []
    for k in range(3):
        vbig.reset_workspace()
        ms.append(vbig.solve_timed())
    med = sorted(ms[1:])[0]
    vtf = vB * vit * vp.flops_per_iteration() / (med * 1e-3) / 1e12
    out["very_large_system"] = {"workload": "synthetic nx=%d nu=%d N=%d, box constraints, %d instances x %d forced iterations" % (vnx, vnu, vN, vB, vit),
                                "iters_per_s": vB * vit / (med * 1e-3), "kernel_ms": med, "fp64_tflops": vtf, "fp64_frac": vtf / PEAK_FP64_TFLOPS,
                                "layout": vbig.launch_info()["layout"], "kernel": "k_admm_solve_m<32> (four row tiles per wavefront, streamed operator tiles)",
                                **leg_counters("very_large_system", vB * vit / (med * 1e-3))}
    vbig.reset()
    # BASELINE config 1: cartpole nx=4 nu=1 N=20, box input constraints, 200 ADMM iterations -- one instance (the reference's
    # example as it stands) and a batch of 8,192 (layout D's compiled-in cartpole shape)
    cp = P.cartpole(20, True)
    cart = {}
    for cB in (1, 8192):
        cs = pkg.TinyMPC()
        cs.setup(cp.A, cp.B, cp.Q, cp.R, cp.N, batch=cB, device=dev_index, rho=cp.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=200)
        cs.set_bound_constraints(cp.x_min, cp.x_max, cp.u_min, cp.u_max)
        if cB == 1:
            cs.set_x0(cp.x0)
        else:
            cs.set_x0_batch(np.asfortranarray(cp.x0[:, None] * np.linspace(0.5, 1.5, cB)[None, :]))
        ms = []
        for k in range(6):
            cs.reset_workspace()
            ms.append(cs.solve_timed())
        med = sorted(ms[1:])[2]
        key = "one_instance" if cB == 1 else "batch_8192"
        cart[key] = {"kernel_ms": med, "iters_per_s": cB * 200 / (med * 1e-3), "us_per_iter": 1e3 * med / 200 if cB == 1 else None,
                     **({"cpu_reference_us_per_iter": cpu.get("cartpole_us_per_iter_single_process") if cpu else None} if cB == 1 else {}),
                     "layout": cs.launch_info()["layout"], "fp64_frac": cB * 200 * cp.flops_per_iteration() / (med * 1e-3) / 1e12 / PEAK_FP64_TFLOPS}
        cs.reset()
    out["cartpole"] = dict(workload="BASELINE config 1: cartpole nx=4 nu=1 N=20, box input constraints, 200 forced iterations", **cart)
    # Closed-loop tick (examples/cartpole_example_mpc.m:36-44 on the quadrotor): x0 in -> warm-started solve -> first
    # controls out, tol 1e-3, 200 ticks of the same trajectory, (a) one launch per tick, (b) resident session kernel.
    # (Mean, median and maximum of the 200 timed ticks: one tick in a few thousand takes milliseconds -- 41 ms once in this
    # file's run -- and moves the mean of 200 by a factor of ten. tools/tick_outliers.py (20,000 ticks, the library's own
    # split per outlier via tinympc_debug_tick_timing): the launch call and the wait of such a tick are the usual
    # 5 + 18 us; the time goes to the calling thread being descheduled, outside the library.)
    tick = {}
    for mode in ("launch", "session"):
        tk = pkg.TinyMPC()
        tk.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, device=dev_index, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
        tk.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if mode == "session":
            tk.session_begin()
        x = prob.x0.copy()
        t_acc, its, dts, slow = 0.0, 0, [], []
        import ctypes as _C
        _L = pkg.load_library()
        _L.tinympc_debug_tick_timing.restype = _C.c_int
        _L.tinympc_debug_tick_timing.argtypes = [_C.c_void_p, _C.POINTER(_C.c_double)]
        split = (_C.c_double * 4)()
        for k in range(220):
            t0 = time.perf_counter()
            u0 = tk.session_step(x) if mode == "session" else tk.mpc_step(x)[:, 0]
            dt = time.perf_counter() - t0
            if k >= 20:
                t_acc += dt
                dts.append(dt)
                its += int(tk.get_stats()["iter"])
                if dt > 1e-3 and mode == "launch" and len(slow) < 4:  # where did a millisecond tick spend its time?
                    _L.tinympc_debug_tick_timing(tk._h, split)
                    slow.append({"tick": k, "us": 1e6 * dt, "library_launch_call_us": split[0], "library_wait_us": split[1],
                                 "polls": int(split[2]), "polling_budget_ran_out": bool(split[3])})
            x = prob.A @ x + prob.B @ u0
        # ... and the same 200 ticks driven from C (tinympc_bench_closed_loop: no Python call inside a tick) -- how the reference
        # core beside it is timed too (oracle/ref_shim.cpp: ref_bench_closed_loop)
        cl = tk.bench_closed_loop(prob.A, prob.B, prob.x0, 220, 20, session=(mode == "session"))
        if mode == "session":
            tk.session_end()
        tick[mode] = {"us_per_tick": 1e6 * t_acc / 200, "us_per_tick_median": 1e6 * float(np.median(dts)), "us_per_tick_max": 1e6 * float(np.max(dts)),
                      "tick_us": stats_us(1e6 * np.asarray(dts)), "iterations_per_tick": its / 200,
                      "c_loop": dict({k: cl[k] for k in ("us_per_tick", "us_per_tick_median", "us_per_tick_max", "iterations_per_tick")},
                                     tick_us=stats_us(cl["tick_us"]))}
        if slow:
            tick[mode]["ticks_above_1ms"] = slow
        tk.reset()
    # ... and the reference's own three verbs per tick (set_x0 + solve + get_solution, the whole solution copied out) from the same C
    # loop: launched solves (what a script written against the reference gets as it stands) and RESIDENT solves (one extra line,
    # tinympc_set_resident: the same verbs on the resident session kernel)
    for mode in ("verbs_launched", "verbs_resident"):
        tk = pkg.TinyMPC()
        tk.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, device=dev_index, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
        tk.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if mode == "verbs_resident":
            tk.set_resident(True)
        cl = tk.bench_closed_loop(prob.A, prob.B, prob.x0, 220, 20, session="verbs")
        tick[mode] = {"c_loop": dict({k: cl[k] for k in ("us_per_tick", "us_per_tick_median", "us_per_tick_max", "iterations_per_tick")}, tick_us=stats_us(cl["tick_us"]))}
        tk.reset()
    # ... and the cartpole at N = 10 (the reference's interactive closed-loop example's shape; no compiled-in specialisation: layout C's resident
    # kernel, what ANY user system gets without tinympc_prepare()), resident session from the same C loop
    cp10 = P.cartpole(10, True)
    tk = pkg.TinyMPC()
    tk.setup(cp10.A, cp10.B, cp10.Q, cp10.R, cp10.N, batch=1, device=dev_index, rho=cp10.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
    tk.set_bound_constraints(cp10.x_min, cp10.x_max, cp10.u_min, cp10.u_max)
    tk.session_begin()
    cl = tk.bench_closed_loop(cp10.A, cp10.B, cp10.x0, 220, 20, session=True)
    c10 = ((cpu or {}).get("latency_extra") or {}).get("cartpole10_closed_loop_tick_us")
    tick["cartpole_N10_session"] = {"layout": tk.launch_info()["layout"], "c_loop": dict(tick_us=stats_us(cl["tick_us"]), iterations_per_tick=cl["iterations_per_tick"]),
                                    "cpu_reference_tick_us": c10,
                                    "cpu_reference_iterations_per_tick": ((cpu or {}).get("latency_extra") or {}).get("cartpole10_closed_loop_iterations_per_tick")}
    tk.session_end()
    tk.reset()
    # The SAME statistics on both sides (round 5): mean, median, p90, max of the 200 per-tick samples -- the GPU's from
    # libtinympc_bench.so's C loop, the reference core's from oracle/ref_shim.cpp's (ref_bench_closed_loop_samples)
    cref = (cpu or {}).get("closed_loop_tick_us_single_process") or {}
    tick["cpu_reference_tick_us"] = cref or None
    tick["cpu_reference_us_per_tick"] = cref.get("mean")
    tick["cpu_reference_us_per_tick_median"] = cref.get("median")
    tick["cpu_reference_iterations_per_tick"] = cpu.get("closed_loop_iterations_per_tick") if cpu else None
    tick["cpu_reference_note"] = ("the reference's own core (oracle/_ref) on one host core: set_x0 + solve + first control per tick, the loop in "
                                  "compiled code; no MATLAB / MEX overhead on its side. Like against like: c_loop.tick_us.{mean,median} against "
                                  "cpu_reference_tick_us.{mean,median} (both loops in C); the Python-mirror numbers carry ctypes calls the host side does not")
    if cref:
        for mode in ("launch", "session", "verbs_launched", "verbs_resident"):
            g_ = tick[mode]["c_loop"]["tick_us"]
            tick[mode]["c_loop"]["gpu_over_cpu_time"] = {"mean": g_["mean"] / cref["mean"], "median": g_["median"] / cref["median"]}
    # ... and BASELINE config 4's own closed loop (rocket_landing_constraints.m:86-121): N = 100, cones + linear row + fdyn, the
    # reference trajectory re-sent every tick (a receding horizon: inside a session only its new last column travels)
    rk = P.rocket(100)
    goal = np.zeros(rk.nx)
    rtick = {}
    for mode in ("launch", "session"):
        tk = pkg.TinyMPC()
        tk.setup(rk.A, rk.B, rk.Q, rk.R, rk.N, batch=1, device=dev_index, rho=rk.rho, fdyn=rk.fdyn, abs_pri_tol=5e-2, abs_dua_tol=5e-2, max_iter=200)
        tk.set_bound_constraints(rk.x_min, rk.x_max, rk.u_min, rk.u_max)
        tk.set_u_ref(rk.u_ref)
        tk.set_cone_constraints(**rk.cones)
        tk.set_linear_constraints(**rk.linear)
        tk.set_x_ref(rk.x_ref)
        tk.prepare()
        layout = tk.launch_info()["layout"]
        if mode == "session":
            tk.session_begin()
        x = rk.x0.copy()
        dts, its = [], 0
        for k in range(110):
            x_ref = np.stack([rk.x0 + (goal - rk.x0) * min(i + k, 140) / 140 for i in range(rk.N)], axis=1)
            t0 = time.perf_counter()
            tk.set_x_ref(x_ref)
            u0 = tk.session_step(x) if mode == "session" else tk.mpc_step(x)[:, 0]
            dt = time.perf_counter() - t0
            if k >= 10:
                dts.append(dt)
                its += int(tk.get_stats()["iter"])
            x = rk.A @ x + rk.B @ u0 + rk.fdyn
        if mode == "session":
            tk.session_end()
        rtick[mode] = {"us_per_tick_median": 1e6 * float(np.median(dts)), "us_per_tick": 1e6 * float(np.mean(dts)), "tick_us": stats_us(1e6 * np.asarray(dts)),
                       "iterations_per_tick": its / 100, "layout": layout}
        tk.reset()
    # (like against like, round 5: both sides time set_x_ref + tick per tick through Python calls of their C libraries and quote the
    # same statistics of the same 100 ticks; the early ticks are long -- 38 iterations on average -- so mean and median differ 2x)
    rref = (cpu or {}).get("rocket_closed_loop_tick_us_single_process") or {}
    rtick["cpu_port_tick_us"] = rref or None
    rtick["cpu_port_us_per_tick"] = rref.get("mean")
    rtick["cpu_port_us_per_tick_median"] = rref.get("median")
    rtick["cpu_port_iterations_per_tick"] = cpu.get("rocket_closed_loop_iterations_per_tick") if cpu else None
    if rref:
        for mode in ("launch", "session"):
            rtick[mode]["cpu_over_gpu_time"] = {"mean": rref["mean"] / rtick[mode]["tick_us"]["mean"], "median": rref["median"] / rtick[mode]["tick_us"]["median"]}
    out["rocket_closed_loop"] = dict(workload="rocket landing N=100, cones + linear row + fdyn, one instance, warm start, tol 5e-2, the reference trajectory "
                                              "re-sent every tick, 100 ticks (set_x_ref + tick, through the Python mirror)", **rtick)
    out["closed_loop_tick"] = dict(workload="quadrotor N=%d, one instance, warm start, tol 1e-3, 200 ticks through the Python mirror of the C ABI" % prob.N, **tick)

