#!/usr/bin/env python3
"""One bench leg as a stand-alone workload, for rocprofv3 (tools/profile_legs.sh): the same handles bench.py builds for its
extra legs, a few launches each, nothing else in the process (no torch import: numpy + the C ABI only).

    python tools/leg_workload.py <leg> [launches]
legs: headline, rocket_batch, rocket_batch_n10, rocket_instance, wide_system, wide_families, long_horizon, large_system, very_large_system, adaptive_rho_batch, single_instance,
      converging_batch, converging_batch_plain
Prints one JSON line: leg, kernel layout, launches, iterations per launch, instances, median kernel ms (HIP events)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
P = pkg.problems


def rocket_handle(N, batch, iters):
    rk = P.rocket(N)
    s = pkg.TinyMPC()
    s.setup(rk.A, rk.B, rk.Q, rk.R, rk.N, batch=batch, rho=rk.rho, fdyn=rk.fdyn, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1)
    s.set_bound_constraints(rk.x_min, rk.x_max, rk.u_min, rk.u_max)
    s.set_x_ref(rk.x_ref)
    s.set_u_ref(rk.u_ref)
    s.set_cone_constraints(**rk.cones)
    s.set_linear_constraints(**rk.linear)
    if batch == 1:
        s.set_x0(rk.x0)
    else:
        s.set_x0_batch(np.asfortranarray(rk.x0[:, None] * np.linspace(0.6, 1.2, batch)[None, :]))
    return s


def synthetic(nx, nu, N, batch, iters, seed, a_scale, b_scale, diag=1.0):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) * diag + a_scale * rng.standard_normal((nx, nx))
    B = b_scale * rng.standard_normal((nx, nu))
    p = P.Problem("synthetic", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    s = pkg.TinyMPC()
    s.setup(p.A, p.B, p.Q, p.R, p.N, batch=batch, rho=p.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters)
    s.set_bound_constraints(np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3))
    s.set_x0_batch(np.asfortranarray(np.random.default_rng(1).standard_normal((nx, batch))))
    return s


def quadrotor(N, batch, iters, **extra):
    prob = P.quadrotor(N)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters, check_termination=1, **extra)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if batch == 1:
        s.set_x0(prob.x0)
    else:
        s.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(batch)))
    return s


def build(leg):
    """-> (handle, instances, iterations per launch)"""
    if leg == "headline":
        return quadrotor(50, 8192, 200), 8192, 200
    if leg == "rocket_batch":
        return rocket_handle(100, 4096, 100), 4096, 100
    if leg == "rocket_instance":
        return rocket_handle(100, 1, 200), 1, 200
    if leg == "rocket_batch_n10":  # (the reference example's own horizon: layout E's uncut form)
        return rocket_handle(10, 4096, 100), 4096, 100
    if leg == "wide_system":
        return synthetic(24, 8, 30, 4096, 100, 0, 0.03, 0.1), 4096, 100
    if leg == "wide_families":  # (round 5: the families streamed next to layout D's wide sweeps; bench.py: wide_system.with_families)
        s = synthetic(24, 8, 30, 4096, 100, 0, 0.03, 0.1)
        s.set_cone_constraints(Acx=[0], qcx=[3], cx=[0.7], Acu=[], qcu=[], cu=[])
        s.set_linear_constraints(Alin_x=np.random.default_rng(1).standard_normal((2, 24)), blin_x=np.array([1.0, 1.5]), Alin_u=np.zeros((0, 8)), blin_u=np.zeros(0))
        return s, 4096, 100
    if leg == "long_horizon":
        return quadrotor(100, 8192, 100), 8192, 100
    if leg == "large_system":
        return synthetic(96, 32, 20, 4096, 50, 96, 0.015, 0.08, 0.98), 4096, 50
    if leg == "very_large_system":  # (beyond 256 rows: four row tiles per wavefront, operator tiles streamed)
        return synthetic(480, 32, 20, 4096, 20, 7, 0.1 / np.sqrt(480.0), 0.08, 0.6), 4096, 20
    if leg == "adaptive_rho_batch":
        s = quadrotor(50, 8192, 100, adaptive_rho=True, adaptive_rho_min=0.2, adaptive_rho_max=40.0)
        s.set_sensitivity_matrices(*s.compute_sensitivity_autograd())
        return s, 8192, 100
    if leg == "single_instance":
        return quadrotor(50, 1, 200), 1, 200
    if leg in ("converging_batch", "converging_batch_plain"):
        # 65,536 quadrotors solved to a tolerance, 5 ... 200 iterations each: the slot-refill variant (tinympc_solve_dr.hip) against
        # the plain kernel (TINYMPC_REFILL=0). "iterations per launch" is the mean over the batch (127.6).
        os.environ["TINYMPC_REFILL"] = "0" if leg.endswith("_plain") else "1"
        prob = P.quadrotor(50)
        nb = 65536
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=nb, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=200, check_termination=1)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        rng = np.random.default_rng(0)
        s.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(nb) * rng.uniform(0.05, 3.0, nb)[None, :]))
        return s, nb, 127.574
    raise SystemExit("unknown leg " + leg)


def main():
    leg = sys.argv[1]
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    s, inst, iters = build(leg)
    # Untimed repetitions first, until the shader clock has settled (round 5: the first launches of a process run 10-25 % slower --
    # profiles/r04_headline_pmc.json had 1.84-2.09 ms where the steady state is 1.68): at least 0.4 s of back-to-back launches and at
    # least 50 of them (fewer if they take more than 4 s). tools/collect_leg_profiles.py looks at the LAST `launches` dispatches only.
    import time
    t0, warm = time.perf_counter(), 0
    while (time.perf_counter() - t0 < 0.4 or warm < 50) and time.perf_counter() - t0 < 4.0:
        s.reset_workspace()
        s.solve_timed()
        warm += 1
    ms = []
    for _ in range(launches):
        s.reset_workspace()
        ms.append(s.solve_timed())
    med = float(np.median(ms))
    print(json.dumps({"leg": leg, "layout": s.launch_info()["layout"], "jit": s.jit_info(), "launches": launches, "warmup_launches": warm, "iterations_per_launch": iters,
                      "instances": inst, "kernel_ms_median": med, "iters_per_s": inst * iters / (med * 1e-3)}), flush=True)
    s.reset()


if __name__ == "__main__":
    main()
