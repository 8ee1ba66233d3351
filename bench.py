#!/usr/bin/env python3
"""bench.py -- ADMM iterations/s of the batched quadrotor solve (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`;
     run directly with --gpus N > 1 it re-launches itself that way before touching the GPU.)

One "step" = one cold-started batched solve on every rank: reset the persistent ADMM state, then ONE
launch of k_admm_solve running 200 forced ADMM iterations (tolerances 0, check_termination 1, so the
residual reductions are always paid) for `--batch-per-gpu` quadrotor instances (nx=12, nu=4, N=50, box
constraints on states and inputs, rho=5) whose x0 are already resident in HBM. Weak scaling: the per-GPU
shard is fixed (8,192 instances), so 8 GPUs reproduce BASELINE config 5 exactly (65,536 instances);
instances are independent, so there is no data-path collective -- only a summary all-reduce after the
timed region.

Prints ONE JSON line on rank 0 (see the keys below). `value` is the whole-job aggregate
instance-iterations per second; `roofline` prices the solve kernel against the roof that BINDS it -- FP64 vector
issue: algorithmic flops (SURVEY.md section 8a's formula, 60,848 per instance-iteration) / the kernel duration measured
live with HIP events on the kernel's stream / 78.6 TFLOP/s. The ADMM state stays on chip for the whole solve, so the
streaming-bytes figure of SURVEY.md section 8d (8*(11U+9X) per instance-iteration against 8 TB/s) exceeds 1 and is
kept only as the secondary block `roofline.hbm_algorithmic`; `roofline.traffic` is the PMC-measured HBM bytes per
launch, reported only when profiles/traffic_latest.json was collected with the library that is running now.
`parity_check` compares instances 0..63 of the timed workload with tests/golden/quadrotor_batch64.npz (the reference
core's own output) after the timed region and fails the run beyond 1e-6. `cpu_baseline` times the reference's own compiled
core (oracle/_ref, kind "reference") -- or this repo's C restatement (kind "port") where that binary is
absent -- on the host cores, on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_HBM_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
PEAK_FP64_TFLOPS = 78.6    # MI355X FP64 vector peak (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=8192)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed TOTAL number of instances split over the GPUs (strong scaling); overrides --batch-per-gpu")
    ap.add_argument("--iters", type=int, default=200, help="forced ADMM iterations per solve")
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-settle", action="store_true", help="time the K steps right behind the W warm-up steps (the GPU's clock is then still ramping up)")
    ap.add_argument("--sync-steps", action="store_true", help="wait for every timed step before launching the next (default: the K steps are queued and waited for once)")
    ap.add_argument("--no-single", action="store_true", help="skip the B=1 latency measurement")
    ap.add_argument("--no-config5", action="store_true", help="skip the 65,536-instance single-GPU leg")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default=os.environ.get("TINYMPC_BENCH_DIST_BACKEND", "nccl"),
                    help="process-group backend of the N > 1 run: nccl (= RCCL over xGMI, one GPU per rank; the default and what the driver "
                         "runs) or gloo (host-side collectives: the only backend that lets several ranks share one device)")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank solves on device 0 (rehearsal of the N > 1 path on a one-GPU box; needs --dist-backend gloo: "
                         "RCCL refuses two ranks on one device)")
    ap.add_argument("--devices", default="", help="comma-separated device index per rank (default: LOCAL_RANK; --share-device = all 0)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--worker-index", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-rocket", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-latency", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-ticks", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--worker-count", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-round5-legs", action="store_true", help="skip the setup and batched_tick legs")
    return ap.parse_args()


def rccl_environment() -> None:
    """Everything the multi-process GPU path needs in the environment, set before anything initialises HIP (and inherited
    by the torchrun children):
      HSA_ENABLE_IPC_MODE_LEGACY=0  the host driver of this pool only supports dmabuf IPC; without it RCCL's
                                    hipIpcGetMemHandle fails with `invalid argument` as soon as two ranks exchange buffers
      MASTER_ADDR=127.0.0.1         single-node rendezvous; the container's hostname may not resolve
    Values already present are respected."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


def relaunch_under_torchrun(args) -> int:
    rccl_environment()
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def cpu_worker(seconds: float, iters: int, horizon: int, index: int) -> int:
    """One baseline process: cold-started `iters`-iteration quadrotor solves on ONE core until the time
    budget is spent. Prints 'kind total_iterations elapsed_seconds'. Never touches the GPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import pyoracle as O  # checker / baseline only
    import __graft_entry__ as ge

    P = ge.load_package().problems
    prob = P.quadrotor(horizon)
    settings = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1)
    cls = O.OracleRef if O.ref_available() else O.OraclePort
    solver = cls(prob).load_problem(prob, settings)
    x0s = P.quadrotor_batch_x0(16, offset=16 * index)
    total = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        total += solver.bench_solves(x0s, 1)
    print(cls.kind, total, time.perf_counter() - t0, flush=True)
    return 0


def cpu_worker_rocket(seconds: float, iters: int) -> int:
    """BASELINE config 4 (rocket landing N=100, cones + linear + fdyn) on ONE host core with this repo's C
    restatement (the reference snapshot has no source for these families). Prints 'iterations seconds'."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import pyoracle as O  # checker / baseline only
    import __graft_entry__ as ge

    import numpy as np
    prob = ge.load_package().problems.rocket(100)
    solver = O.OraclePort(prob).load_problem(prob, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1))
    total = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        solver.reset_workspace()
        solver.set_x0(prob.x0)
        solver.solve()
        total += solver.stats()["iter"]
    elapsed_iter = time.perf_counter() - t0
    # ... and BASELINE config 4 as the reference uses it (rocket_landing_constraints.m:86-121): a closed loop with the reference
    # trajectory re-sent every tick, tol 5e-2, max_iter 200, 100 ticks -- same loop as bench.py's `rocket_closed_loop` leg
    solver = O.OraclePort(prob).load_problem(prob, dict(abs_pri_tol=5e-2, abs_dua_tol=5e-2, max_iter=200, check_termination=1))
    x, goal = prob.x0.copy(), np.zeros(prob.nx)
    per, its = [], 0
    for k in range(110):
        x_ref = np.stack([prob.x0 + (goal - prob.x0) * min(i + k, 140) / 140 for i in range(prob.N)], axis=1)
        t1 = time.perf_counter()
        solver.set_x_ref(x_ref)
        solver.set_x0(x)
        solver.solve()
        u0 = solver.solution()[1][:, 0]
        if k >= 10:
            per.append(1e6 * (time.perf_counter() - t1))
            its += solver.stats()["iter"]
        x = prob.A @ x + prob.B @ u0 + prob.fdyn
    from tools.bench_legs import stats_us
    print(json.dumps({"iterations": total, "seconds": elapsed_iter, "closed_loop_tick_us": stats_us(per), "closed_loop_iterations_per_tick": its / 100}), flush=True)
    return 0


def cpu_worker_latency(seconds: float) -> int:
    """The reference's own core (oracle/_ref) on ONE host core in the reference's own usage, beside the GPU's latency legs:
    (a) BASELINE configs 1/2, cartpole N=20 with input bounds, 200 forced iterations (examples/cartpole_example_one_solve.m:13-31);
    (b) the closed loop of bench.py's `closed_loop_tick` leg -- quadrotor N=50, tol 1e-3, max_iter 100, warm start, 20 untimed + 200
        timed ticks of set_x0 -> solve -> first control (examples/cartpole_example_mpc.m:36-44), the loop itself in compiled code
        (oracle/ref_shim.cpp: ref_bench_closed_loop) so that no Python call sits inside a tick.
    Prints one JSON object. Never touches the GPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import pyoracle as O  # checker / baseline only
    import __graft_entry__ as ge

    if not O.ref_available():
        print(json.dumps({}), flush=True)
        return 0
    P = ge.load_package().problems
    cp = P.cartpole(20, True)
    s = O.OracleRef(cp).load_problem(cp, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=200, check_termination=1))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        n += s.bench_solves(cp.x0.reshape(-1, 1), 20)
    cart_us = 1e6 * (time.perf_counter() - t0) / n
    from tools.bench_legs import cpu_setup_and_ticks
    out = cpu_setup_and_ticks()  # tiny_setup per system + the per-tick samples of the quadrotor closed loop
    out["cartpole_us_per_iter"] = cart_us
    print(json.dumps(out), flush=True)
    return 0


def physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; None where the file does not say."""
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or None
    except OSError:
        return None


def library_hash() -> str:
    """sha256 over the sources of the HIP library (csrc/ + the C-ABI header): stamps PMC traffic figures to the kernels
    they were measured on (the built .so is not tracked, its sources are)."""
    import hashlib
    csrc = os.path.join(ROOT, "tinympc-matlab_amd", "csrc")
    paths = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))]
    paths.append(os.path.join(ROOT, "include", "tinympc_hip.h"))
    h = hashlib.sha256()
    for path in paths:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def usable_cpus():
    """What this process may actually run on: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() ignores both).
    Returns (sorted list of CPU ids, quota in CPUs or None, where the quota was read)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = list(range(os.cpu_count() or 1))
    quota, src = None, None
    try:  # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            quota, src = float(q) / float(per), "/sys/fs/cgroup/cpu.max"
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota, src = q / per, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"
        except (OSError, ValueError):
            pass
    return cpus, quota, src


PROFILE_TAG = "r05"


def leg_counters(leg: str, iters_per_s: float) -> dict:
    """What rocprofv3 measured for this leg's kernel (profiles/<tag>_<leg>_pmc.json, tools/profile_legs.sh + collect_leg_profiles.py):
    VALU issue fraction of the chip, HBM bytes per instance-iteration -> measured GB/s at the LIVE rate of this run. Empty when
    the profile is absent; DROPPED (with a note) when it was collected on other kernel sources than the library running now --
    counters of an older kernel next to a fresh rate would be a stale 'measured' figure."""
    name = "%s_%s_pmc.json" % (PROFILE_TAG, leg)
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return {}
    if d.get("library_hash") != library_hash():
        return {"measured": {"source": "profiles/" + name, "dropped": "collected on other kernel sources (%s, now %s)" % (d.get("library_hash"), library_hash())}}
    out = {"source": "profiles/" + name, "library_hash": d.get("library_hash")}
    dd, hb = d.get("derived") or {}, d.get("hbm") or {}
    for k in ("valu_per_wave_iteration", "valu_issue_fraction_of_chip", "shader_clock_ghz_grbm"):
        if k in dd:
            out[k] = dd[k]
    if "bytes_per_instance_iteration" in hb:
        gbs = hb["bytes_per_instance_iteration"] * iters_per_s / 1e9
        out.update(hbm_bytes_per_instance_iteration=hb["bytes_per_instance_iteration"], hbm_measured_gbs=gbs, hbm_measured_frac=gbs / PEAK_HBM_GBS)
    return {"measured": out}


def cpu_baseline(iters: int, horizon: int, seconds: float, round5: bool = True) -> dict:
    """The reference's own compiled core (oracle/_ref; the C port where that binary is absent) on the host cores this
    process may use: one PROCESS per usable CPU (Eigen's per-operation malloc makes threads of one process contend), each
    pinned to its own CPU of the affinity mask, the count capped by the cgroup quota, each running seeded cold-started
    solves for `seconds`. `cores` is the number of workers that ran; `effective_parallelism` = summed rate / the rate of one
    process alone says what the box really gave them. Runs before this process initialises the GPU."""
    cpus, quota, quota_src = usable_cpus()
    nworkers = len(cpus)
    if quota is not None:
        nworkers = max(1, min(nworkers, int(quota + 0.5)))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-seconds", str(seconds),
           "--iters", str(iters), "--horizon", str(horizon)]

    def pin(cpu):
        def f():
            try:
                os.sched_setaffinity(0, {cpu})
            except (AttributeError, OSError):
                pass
        return f

    # one process alone first, for the per-core anchor -- on an otherwise idle host
    one = subprocess.run(cmd + ["--worker-index", "0", "--cpu-seconds", "2"], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, text=True, preexec_fn=pin(cpus[0])).stdout.split()
    single = int(one[-2]) / float(one[-1]) if len(one) >= 3 else float("nan")
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ["--worker-index", str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              preexec_fn=pin(cpus[i % len(cpus)]))
             for i in range(nworkers)]
    total, kind, rates = 0, "port", []
    for pr in procs:
        out, _ = pr.communicate()
        try:
            k, n, dt = out.split()[-3:]
            kind, total = k, total + int(n)
            rates.append(int(n) / float(dt))
        except (ValueError, IndexError):
            pass
    wall = time.perf_counter() - t0
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    value = sum(rates)
    physical = physical_cores()
    eff = value / single if single == single and single > 0 else None
    note = None
    if eff is not None and len(rates) > 1.2 * eff:
        note = ("%d pinned workers delivered the throughput of %.1f undisturbed single processes: SMT siblings share a core's FP units "
                "(%s physical cores) and %.0f s of work took %.1f s of wall clock" % (len(rates), eff, physical, seconds, wall))
    rocket = {}
    try:
        rocket = json.loads(subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker-rocket", "--cpu-seconds", "2", "--iters", str(iters)],
                                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, preexec_fn=pin(cpus[0])).stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        pass
    rocket_us = 1e6 * rocket["seconds"] / rocket["iterations"] if rocket.get("iterations") else None
    rocket_tick, rocket_tick_its = rocket.get("closed_loop_tick_us"), rocket.get("closed_loop_iterations_per_tick")
    lat = {}
    try:
        lat = json.loads(subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker-latency", "--cpu-seconds", "1"],
                                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, preexec_fn=pin(cpus[0])).stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        pass
    from tools.bench_legs import cpu_batched_ticks
    batched = cpu_batched_ticks(cpus, nworkers, os.path.abspath(__file__)) if round5 else {}
    return {"value": value, "rocket_us_per_iter_single_process": rocket_us,
            "rocket_closed_loop_tick_us_single_process": rocket_tick, "rocket_closed_loop_iterations_per_tick": rocket_tick_its,
            "cartpole_us_per_iter_single_process": lat.get("cartpole_us_per_iter"),
            "closed_loop_tick_us_single_process": lat.get("closed_loop_tick_us"),
            "closed_loop_iterations_per_tick": lat.get("closed_loop_iterations_per_tick"),
            "setup_us": lat.get("setup_us"), "batched_ticks": batched,
            "latency_extra": {k: lat.get(k) for k in ("cartpole10_closed_loop_tick_us", "cartpole10_closed_loop_iterations_per_tick")},
            "unit": "ADMM iters/s", "cores": len(rates),
            "effective_parallelism": eff, "affinity_cpus": len(cpus), "cgroup_quota": quota, "cgroup_quota_source": quota_src,
            "cores_logical": os.cpu_count(), "cores_physical": physical, "kind": kind,
            "sample": f"{total // iters} cold-started {iters}-iteration quadrotor N={horizon} solves, {len(rates)} single-threaded "
                      f"processes, each pinned to one CPU of the affinity mask, x {seconds:.0f} s each ({wall:.1f} s wall; seeded x0, "
                      f"same settings as the GPU run)",
            "single_process_iters_per_s": single, "us_per_iter_single_process": 1e6 / single if single == single else None,
            "physical_cores_x_single_process": (physical * single) if physical and single == single else None,
            "note": note, "cpu_model": cpu_model}


def main() -> int:
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_seconds, args.iters, args.horizon, args.worker_index)
    if args.cpu_worker_rocket:
        return cpu_worker_rocket(args.cpu_seconds, args.iters)
    if args.cpu_worker_latency:
        return cpu_worker_latency(args.cpu_seconds)
    if args.cpu_worker_ticks:
        sys.path.insert(0, ROOT)
        from tools.bench_legs import cpu_ticks_worker, TICKS, TICK_SKIP
        print(json.dumps(cpu_ticks_worker(args.worker_index, args.worker_count, TICKS, TICK_SKIP)), flush=True)
        return 0
    rccl_environment()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        return relaunch_under_torchrun(args)  # nothing has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CPU baseline first (rank 0, single-GPU runs only): child processes, started before this process
    # has initialised the GPU, so the host cores are not shared with the timed GPU region either.
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.iters, args.horizon, args.cpu_seconds, round5=not args.no_round5_legs)

    import numpy as np
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    from tools.bench_legs import (LegContext, batched_tick_leg, config5_legs, instance_and_family_legs, setup_leg, shape_and_latency_legs,
                                  stats_us)
    pkg = ge.load_package()
    P = pkg.problems
    # The interpreter's cyclic collector: a full pass over torch's ~170,000 import-time objects takes 40-80 ms and lands wherever the
    # allocation count says -- round 5 saw it as ONE 55 ms and one 78 ms tick among 400 Python-timed batched ticks (their means went from
    # 125 to 1,501 us). gc.freeze() moves what exists now to the permanent generation: later passes look at the run's own objects only.
    import gc
    gc.collect()
    gc.freeze()
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    # Which device this rank solves on: its LOCAL_RANK (one GPU per rank, what the driver launches), or what --devices /
    # --share-device say (several ranks on one GPU: the N > 1 path rehearsed on a one-GPU box, collectives over gloo).
    if args.devices:
        dev_of_rank = [int(d) for d in args.devices.split(",")]
        if len(dev_of_rank) != world:
            print("bench.py: --devices names %d devices for %d ranks" % (len(dev_of_rank), world), file=sys.stderr)
            return 2
        dev_index = dev_of_rank[rank]
    else:
        dev_index = 0 if args.share_device else local_rank
    shared = world > 1 and (args.share_device or (args.devices and len(set(dev_of_rank)) < world))
    if shared and args.dist_backend == "nccl":
        print("bench.py: several ranks on one device need --dist-backend gloo (RCCL refuses duplicate devices)", file=sys.stderr)
        return 2
    if dev_index < 0 or dev_index >= torch.cuda.device_count():
        print("bench.py: rank %d: device %d does not exist (%d visible)" % (rank, dev_index, torch.cuda.device_count()), file=sys.stderr)
        return 2
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # Under torchrun (RANK / MASTER_ADDR set) the process group is created even for a single rank, so that the
    # collective path can be exercised on a 1-GPU box: `python -m torch.distributed.run --nproc-per-node 1 bench.py`.
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    # the collectives' tensors live where the backend works: HBM for RCCL, host memory for gloo
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    if use_dist:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    prob = P.quadrotor(args.horizon)
    total_instances, first, count, scaling = pkg.batch.job_shard(rank, world, args.batch_per_gpu, args.global_batch)
    strong = scaling == "strong"
    B = total_instances // world if strong else args.batch_per_gpu  # nominal per-GPU share (reports only)
    x0_host = P.quadrotor_batch_x0(count, offset=first)              # (12, count), seeded per global instance index
    x0_dev = torch.from_numpy(np.ascontiguousarray(x0_host.T)).to(dev)  # [count][nx] == nx x count column-major

    solver = pkg.TinyMPC()
    solver.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=count, device=dev_index, rho=prob.rho,
                 abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=args.iters, check_termination=1)
    solver.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    solver.set_x0_batch(x0_dev)  # device-to-device: inputs are HBM-resident before the timed region
    solver.synchronize()

    def step() -> float:
        solver.reset_workspace()      # cold start: every step does identical work
        return solver.solve_timed()   # one launch of k_admm_solve; ms from HIP events on its stream

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps():
        """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides. The K steps are QUEUED
        (tinympc_solve_queued: each records its own event pair around the solve kernel) and waited for once, behind the last one --
        the way a throughput job runs them; a host round trip per step left the GPU idle for ~15 us between a step's end and the
        next step's launch (1.694 against 1.67 ms per step). `--sync-steps` times them one by one. Returns (seconds, kernel ms)."""
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        if args.sync_steps:
            kms = [step() for _ in range(args.steps)]
        else:
            for _ in range(args.steps):
                solver.reset_workspace()  # cold start: every step does identical work (stream-ordered in front of the launch)
                solver.solve_queued()
            kms = solver.collect_kernel_ms()  # (waits for the stream)
            assert len(kms) == args.steps
        barrier()
        return time.perf_counter() - t0, kms

    # (1) AS ASKED: exactly the driver's W warm-up steps, then the K timed steps. This process has just spent ~30 s on the CPU
    # baseline with the GPU idle, and the first launches after that run at a lower shader clock (2.0 ms per launch in the first
    # ten, 1.67 from about the 50th on; in-kernel stamps: tools/clock_check.py), so W = 5 ends inside that ramp.
    # (2) SETTLED: the step is repeated, untimed, until 0.4 s have passed, then W warm-up + K timed steps again -- the
    # steady state a throughput job lives in. `value` is (2); (1) is printed beside it as `value_as_asked`; `--no-settle` runs (1)
    # only and makes it `value`. What the ramp looked like is reported (`settle`: kernel time of the first and last ten launches).
    elapsed_asked, kernel_ms_asked = timed_steps()
    settle = {"launches": 0, "seconds": 0.0}
    if args.no_settle:
        elapsed, kernel_ms = elapsed_asked, kernel_ms_asked
    else:
        ts0 = time.perf_counter()
        sms = []
        while time.perf_counter() - ts0 < 0.4 and len(sms) < 600:
            sms.append(step())
        settle = {"launches": len(sms), "seconds": time.perf_counter() - ts0, "first_launches_kernel_ms": float(np.mean(sms[:10])),
                  "last_launches_kernel_ms": float(np.mean(sms[-10:]))}
        elapsed, kernel_ms = timed_steps()

    # MAX over ranks of the elapsed times (the contract), every rank's own numbers beside it so that a straggler shows, and a
    # fingerprint of what each rank computed: its first controls u[:, 0] (the first four instances verbatim + a SHA-256 over the
    # whole shard) -- sharding must change nothing, bit for bit, and tests/test_multirank_gpu.py checks exactly that.
    import hashlib
    u0_mine = np.ascontiguousarray(solver.get_first_controls_batch().T)  # [count][nu]
    st = solver.get_stats_batch()
    t = torch.tensor([elapsed, sum(kernel_ms) / max(len(kernel_ms), 1), elapsed_asked, sum(kernel_ms_asked) / max(len(kernel_ms_asked), 1)],
                     dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    digest = np.frombuffer(hashlib.sha256(u0_mine.tobytes()).digest(), dtype=np.uint8).astype(np.float64)  # 32 bytes as doubles
    npre = min(4, count)
    prefix = np.zeros(4 * prob.nu)
    prefix[:npre * prob.nu] = u0_mine[:npre].ravel()
    mine = torch.tensor(np.concatenate([[elapsed, sum(kernel_ms) / max(len(kernel_ms), 1), float(count), float(first), float(dev_index),
                                         float(np.sum(st["iter"])), float(npre)], prefix, digest]), dtype=torch.float64, device=cdev)
    if use_dist:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    else:
        gathered = [mine]
    per_rank = []
    for i, g in enumerate(gathered):
        g = g.cpu().numpy()
        k = int(g[6])
        per_rank.append({"rank": i, "device": int(g[4]), "elapsed_s": float(g[0]), "kernel_ms_avg": float(g[1]), "instances": int(g[2]),
                         "first_instance": int(g[3]), "iters_per_s": float(g[2]) * args.iters * args.steps / float(g[0]),
                         "total_iterations": int(g[5]),
                         "first_controls_prefix": g[7:7 + k * prob.nu].reshape(k, prob.nu).tolist(),
                         "first_controls_sha256": bytes(g[7 + 4 * prob.nu:7 + 4 * prob.nu + 32].astype(np.uint8)).hex()})
    elapsed, kernel_ms_avg, elapsed_asked, kernel_ms_asked_avg = (float(v) for v in t.cpu())

    # after the timed region: the one collective of the batched mode (summary statistics)
    summary = pkg.batch.allreduce_summary(pkg.batch.local_summary(st["iter"], st["status"], st["residuals"]), device=cdev)

    inst_iters_per_step = total_instances * args.iters
    value = inst_iters_per_step * args.steps / elapsed
    bytes_iter = prob.bytes_per_iteration()
    flops_iter = prob.flops_per_iteration()
    kernel_s = kernel_ms_avg * 1e-3
    alg_bytes_per_launch = count * args.iters * bytes_iter
    alg_flops_per_launch = count * args.iters * flops_iter
    achieved_gbs = alg_bytes_per_launch / kernel_s / 1e9
    achieved_tflops = alg_flops_per_launch / kernel_s / 1e12

    # Parity of the timed workload itself: instances 0..63 of rank 0 are the x0 of tests/golden/quadrotor_batch64.npz
    # (200 forced iterations of the reference's own core). The last timed step's solutions are still on the device.
    parity = None
    gpath = os.path.join(ROOT, "tests", "golden", "quadrotor_batch64.npz")
    if rank == 0 and os.path.exists(gpath) and args.iters == 200 and args.horizon == 50 and count >= 64:
        g = np.load(gpath)
        if np.array_equal(g["x0s"], x0_host[:, :64]):
            sol = solver.get_solution_batch(0, 64)
            ex = float(np.max(np.abs(sol["states"] - g["sol_x"])) / np.max(np.abs(g["sol_x"])))
            eu = float(np.max(np.abs(sol["controls"] - g["sol_u"])) / np.max(np.abs(g["sol_u"])))
            its = solver.get_stats_batch(0, 64)["iter"]
            parity = {"against": "tests/golden/quadrotor_batch64.npz (reference core, 64 instances x 200 iterations)",
                      "max_rel_err_states": ex, "max_rel_err_controls": eu, "tolerance": 1e-6,
                      "iterations_match": bool(np.array_equal(np.asarray(its), g["iters"])),
                      "ok": bool(ex < 1e-6 and eu < 1e-6 and np.array_equal(np.asarray(its), g["iters"]))}

    out = None
    if rank == 0:
        info = solver.launch_info()
        traffic, traffic_note = None, "no PMC collection for this workload"
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("batch_per_gpu") == B and tj.get("iters") == args.iters and tj.get("horizon") == args.horizon:
                    if tj.get("library_hash") == library_hash():
                        traffic, traffic_note = tj.get("hbm_bytes_per_launch"), ("builder-side PMC on matching sources (library hash %s), not this run: %s"
                                                                                 % (library_hash(), tj.get("source")))
                    else:
                        traffic_note = "dropped: %s was collected on other kernel sources (%s, now %s)" % (
                            tj.get("source"), tj.get("library_hash"), library_hash())
            except (OSError, ValueError):
                traffic = None
        kname = {"A": "k_admm_solve", "B": "k_admm_solve_b", "C": "k_admm_solve_c", "D": "k_admm_solve_d", "E": "k_admm_solve_e (tinympc_jit_solve)", "F": "k_admm_solve_f (tinympc_jit_solve)", "M": "k_admm_solve_m"}.get(info.get("layout"), "k_admm_solve")
        out = {
            "metric": "ADMM iterations/s, batched quadrotor nx=12 nu=4 N=%d (instance-iterations/s, whole job)" % prob.N,
            "value": value, "unit": "ADMM iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "value_as_asked": inst_iters_per_step * args.steps / elapsed_asked, "ms_per_step_as_asked": 1e3 * elapsed_asked / args.steps,
            "value_note": ("value = value_as_asked: --no-settle" if args.no_settle else
                           "value: W warm-up + K timed steps after 0.4 s of untimed repetitions (steady-state clock); value_as_asked: the W warm-up + "
                           "K timed steps run first, right behind the CPU baseline's ~30 s of GPU idling (clock still ramping up)"),
            "ms_per_step": 1e3 * elapsed / args.steps, "steps_queued": not args.sync_steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "vs_baseline_note": "BASELINE.md section 1: the reference publishes no number for this metric",
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "quadrotor hover nx=12 nu=4 N=%d, box x in [-5,5] u in [-0.5,0.5], rho=5, cold start, "
                                   "%d forced ADMM iterations per solve (tol 0, check_termination 1), %d instances per GPU "
                                   "(BASELINE config 5 = 65,536 instances at 8 GPUs)" % (prob.N, args.iters, B),
                       "batch_per_gpu": B, "global_batch": total_instances, "iters_per_solve": args.iters,
                       "parallelism": "independent instances sharded x%d, no data-path collective" % world},
            "solves_per_s": value / args.iters,
            "settle": dict(settle, note="untimed repetitions of the step before the W warm-up steps, until 0.4 s have passed: the GPU idles during the "
                                        "CPU baseline and its first launches run at a lower clock; first / last = mean kernel time of the first / last ten"),
            "roofline": {"bound": "fp64_vector", "achieved": achieved_tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "frac_as_asked": alg_flops_per_launch / (kernel_ms_asked_avg * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                         "kernel_ms_avg_as_asked": kernel_ms_asked_avg, "value_as_asked": inst_iters_per_step * args.steps / elapsed_asked,
                         "kernel": kname + "<%d lanes/instance>" % info["lanes_per_instance"], "kernel_ms_avg": kernel_ms_avg,
                         "algorithmic_flops_per_instance_iteration": flops_iter,
                         "algorithmic_flops_per_launch": alg_flops_per_launch,
                         "note": "binding roof = FP64 vector issue (MI355X: 78.6 TFLOP/s; FP64 MFMA has the same peak and needs "
                                 "16 instances per wavefront whose state does not fit on chip). The ADMM state is register/LDS "
                                 "resident for the whole solve, so HBM is touched at entry and exit only",
                         "hbm_algorithmic": {"achieved": achieved_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                             "frac": achieved_gbs / PEAK_HBM_GBS,
                                             "bytes_per_instance_iteration": bytes_iter, "bytes_per_launch": alg_bytes_per_launch,
                                             "note": "SURVEY.md section 8d streaming model; > 1 means the state never streams "
                                                     "from HBM (on-chip), so this roof does not bind"},
                         "hbm_measured_frac_of_peak": (traffic / kernel_s / 1e9 / PEAK_HBM_GBS) if traffic else None},
            "parity_check": parity,
            "process_group": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                               "ranks_per_device": ("%d ranks share device(s) %s (rehearsal of the N > 1 path on fewer GPUs than ranks: the rates are "
                                                    "not a scaling measurement)" % (world, sorted({p_["device"] for p_ in per_rank}))
                                                    if len({p_["device"] for p_ in per_rank}) < world else 1)} if use_dist else None),
            "legs_run_on": ("rank 0 of a single-rank job only: with n_gpus > 1 the line carries the headline, per_rank, summary and parity_check; "
                            "cpu_baseline, config5_single_gpu, converging_batch and every latency / shape leg are skipped on every rank"),
            "per_rank": per_rank,
            "launch": info,
            "summary": summary,
        }
        ctx = LegContext(pkg=pkg, prob=prob, args=args, cpu=cpu, dev_index=dev_index, dev=dev, torch=torch, np=np, out=out, flops_iter=flops_iter,
                         bytes_iter=bytes_iter, sol=(sol if parity else None), parity=parity, leg_counters=leg_counters)
        # (the legs of rounds 1-4 live in tools/bench_legs.py since round 5: this file keeps the headline, the contract and the flat `legs`)
        if world == 1 and not args.no_config5:
            config5_legs(ctx)
        if world == 1 and not args.no_single:
            instance_and_family_legs(ctx)
            shape_and_latency_legs(ctx)
        if world == 1 and not args.no_round5_legs:
            out["setup"] = setup_leg(pkg, cpu)
            out["batched_tick"] = batched_tick_leg(pkg, cpu, dev_index)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]
        # One headline number per leg, flat: copied into `roofline` as scalars (the driver's record keeps the scalars of the
        # objects it knows and only the NAMES of other keys) and printed as the LAST key of the line (its record keeps the tail).
        def leg(path, scale=1.0):
            d = out
            for k in path.split("/"):
                d = d.get(k) if isinstance(d, dict) else None
                if d is None:
                    return None
            return d * scale if isinstance(d, (int, float)) else None
        legs = {"value_as_asked": out["value_as_asked"],
                "config5_single_gpu_iters_per_s": leg("config5_single_gpu/value"), "config5_single_gpu_fp64_frac": leg("config5_single_gpu/fp64_frac"),
                "converging_batch_kernel_ms": leg("converging_batch/kernel_ms"), "converging_batch_plain_kernel_ms": leg("converging_batch/plain_kernel_ms"),
                "converging_batch_fraction_of_forced_rate": leg("converging_batch/fraction_of_forced_iteration_rate"),
                "single_instance_us_per_iter": leg("single_instance/us_per_iter"), "single_instance_cpu_reference_us_per_iter": leg("cpu_baseline/us_per_iter_single_process"),
                "rocket_instance_us_per_iter": leg("rocket_instance/us_per_iter"), "rocket_instance_cpu_port_us_per_iter": leg("cpu_baseline/rocket_us_per_iter_single_process"),
                "rocket_batch_N100_iters_per_s": leg("rocket_batch/N=100/iters_per_s"), "rocket_batch_N100_fp64_frac": leg("rocket_batch/N=100/fp64_frac"),
                "rocket_batch_N10_iters_per_s": leg("rocket_batch/N=10/iters_per_s"),
                "adaptive_rho_batch_iters_per_s": leg("adaptive_rho_batch/iters_per_s"), "adaptive_rho_batch_fp64_frac_box_part": leg("adaptive_rho_batch/fp64_frac_box_part"),
                "wide_system_fp64_frac": leg("wide_system/fp64_frac"), "wide_system_with_families_kernel_ms": leg("wide_system/with_families/kernel_ms"), "long_horizon_fp64_frac": leg("long_horizon/fp64_frac"),
                "large_system_fp64_frac": leg("large_system/fp64_frac"), "large_system_with_families_iters_per_s": leg("large_system/with_families/iters_per_s"), "large_system_hbm_measured_frac": leg("large_system/hbm_measured_frac"),
                "very_large_system_fp64_frac": leg("very_large_system/fp64_frac"),
                "cartpole_one_instance_us_per_iter": leg("cartpole/one_instance/us_per_iter"), "cartpole_cpu_reference_us_per_iter": leg("cpu_baseline/cartpole_us_per_iter_single_process"),
                "cartpole_batch_8192_iters_per_s": leg("cartpole/batch_8192/iters_per_s"), "cartpole_batch_8192_fp64_frac": leg("cartpole/batch_8192/fp64_frac"),
                # latency comparisons: the SAME statistic on both sides, mean and median each (round 5)
                "closed_loop_tick_launch_c_loop_us_mean": leg("closed_loop_tick/launch/c_loop/tick_us/mean"), "closed_loop_tick_launch_c_loop_us_median": leg("closed_loop_tick/launch/c_loop/tick_us/median"),
                "closed_loop_tick_session_c_loop_us_mean": leg("closed_loop_tick/session/c_loop/tick_us/mean"), "closed_loop_tick_session_c_loop_us_median": leg("closed_loop_tick/session/c_loop/tick_us/median"),
                "closed_loop_tick_verbs_launched_us_mean": leg("closed_loop_tick/verbs_launched/c_loop/tick_us/mean"), "closed_loop_tick_verbs_launched_us_median": leg("closed_loop_tick/verbs_launched/c_loop/tick_us/median"),
                "closed_loop_tick_verbs_resident_us_mean": leg("closed_loop_tick/verbs_resident/c_loop/tick_us/mean"), "closed_loop_tick_verbs_resident_us_median": leg("closed_loop_tick/verbs_resident/c_loop/tick_us/median"),
                "cartpole_N10_session_c_loop_us_mean": leg("closed_loop_tick/cartpole_N10_session/c_loop/tick_us/mean"), "cartpole_N10_session_c_loop_us_median": leg("closed_loop_tick/cartpole_N10_session/c_loop/tick_us/median"),
                "cartpole_N10_cpu_reference_us_mean": leg("closed_loop_tick/cartpole_N10_session/cpu_reference_tick_us/mean"), "cartpole_N10_cpu_reference_us_median": leg("closed_loop_tick/cartpole_N10_session/cpu_reference_tick_us/median"),
                "closed_loop_tick_cpu_reference_us_mean": leg("closed_loop_tick/cpu_reference_tick_us/mean"), "closed_loop_tick_cpu_reference_us_median": leg("closed_loop_tick/cpu_reference_tick_us/median"),
                "closed_loop_tick_session_python_us_mean": leg("closed_loop_tick/session/tick_us/mean"), "closed_loop_tick_session_python_us_median": leg("closed_loop_tick/session/tick_us/median"),
                "rocket_closed_loop_launch_us_mean": leg("rocket_closed_loop/launch/tick_us/mean"), "rocket_closed_loop_launch_us_median": leg("rocket_closed_loop/launch/tick_us/median"),
                "rocket_closed_loop_session_us_mean": leg("rocket_closed_loop/session/tick_us/mean"), "rocket_closed_loop_session_us_median": leg("rocket_closed_loop/session/tick_us/median"),
                "rocket_closed_loop_cpu_port_us_mean": leg("rocket_closed_loop/cpu_port_tick_us/mean"), "rocket_closed_loop_cpu_port_us_median": leg("rocket_closed_loop/cpu_port_tick_us/median"),
                "setup_quadrotor_gpu_ms": leg("setup/quadrotor/gpu_ms"), "setup_quadrotor_cpu_reference_ms": leg("setup/quadrotor/cpu_reference_ms"),
                "setup_cartpole_gpu_ms": leg("setup/cartpole/gpu_ms"), "setup_cartpole_cpu_reference_ms": leg("setup/cartpole/cpu_reference_ms"),
                "setup_rocket_gpu_ms": leg("setup/rocket/gpu_ms"), "setup_quadrotor_riccati_loop_us": leg("setup/quadrotor/k_precompute_riccati_loop_us"),
                "batched_tick_smallest_batch_where_gpu_wins": leg("batched_tick/smallest_batch_where_gpu_wins"),
                **{"batched_tick_B%d_%s" % (b_, k_): leg("batched_tick/%d/%s" % (b_, p_)) for b_ in (1, 16, 256, 4096, 8192)
                   for k_, p_ in (("host_us_mean", "host_exchange_us/mean"), ("resident_us_mean", "device_resident_us/mean"), ("cpu_us_mean", "cpu_reference_us_per_tick_mean"))}}
        legs = {k: v for k, v in legs.items() if v is not None}
        for k, v in legs.items():
            if k != "value_as_asked":
                out["roofline"]["leg_" + k] = v
        out["legs"] = legs
        print(json.dumps(out), flush=True)
    solver.reset()
    if parity is not None and not parity["ok"]:
        print("bench.py: parity_check FAILED: %s" % json.dumps(parity), file=sys.stderr)
        rc_final = 3
    else:
        rc_final = 0
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return rc_final


if __name__ == "__main__":
    sys.exit(main())
