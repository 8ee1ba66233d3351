#!/usr/bin/env python3
"""Dev tool: wall-clock latency of one closed-loop MPC tick through the C ABI (set_x0 -> solve -> u0),
single instance and small batches, warm-started, realistic tolerances."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); P = pkg.problems
for name, prob in (("cartpole N=10", P.cartpole(10, True)), ("quadrotor N=20", P.quadrotor(20)), ("quadrotor N=50", P.quadrotor(50))):
    for B in (1, 64, 1024):
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        x = np.repeat(prob.x0[:, None], B, axis=1) * np.linspace(0.5, 1.0, B)[None, :]
        t_set = t_solve = t_get = 0.0; iters = 0; n = 60
        for k in range(n + 10):
            t0 = time.perf_counter(); s.set_x0_batch(x); t1 = time.perf_counter(); s.solve(); t2 = time.perf_counter()
            u0 = s.get_first_controls_batch(); t3 = time.perf_counter()
            if k >= 10:
                t_set += t1 - t0; t_solve += t2 - t1; t_get += t3 - t2; iters += int(s.get_stats_batch()["iter"].max())
            x = prob.A @ x + prob.B @ u0
        t_fused = 0.0
        for k in range(n + 10):
            t0 = time.perf_counter(); u0 = s.mpc_step(x); t1 = time.perf_counter()
            if k >= 10: t_fused += t1 - t0
            x = prob.A @ x + prob.B @ u0
        print(f"{name:16s} B={B:5d}: fused mpc_step {1e6*t_fused/n:7.1f} us | 3-verb tick {1e6*(t_set+t_solve+t_get)/n:8.1f} us  (set_x0 {1e6*t_set/n:6.1f}, solve {1e6*t_solve/n:7.1f}, get_u0 {1e6*t_get/n:6.1f}; max iters/tick {iters/n:.1f})")
        s.reset()

# The reference's own per-tick sequence (examples/cartpole_example_mpc.m:36-44): set_x0 -> solve -> get_solution on a
# single-instance handle, i.e. what the MEX shim issues. Timed around the three C-ABI calls.
import ctypes as C
L = pkg.load_library()
for name, prob in (("cartpole N=10", P.cartpole(10, True)), ("quadrotor N=50", P.quadrotor(50))):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    x = prob.x0.copy()
    X = np.zeros((prob.nx, prob.N), order="F"); U = np.zeros((prob.nu, prob.N - 1), order="F")
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    t = [0.0, 0.0, 0.0]; n = 200
    for k in range(n + 20):
        xx = np.ascontiguousarray(x)
        t0 = time.perf_counter(); L.tinympc_set_x0(s._h, dp(xx), prob.nx, 0)
        t1 = time.perf_counter(); L.tinympc_solve(s._h, 0)
        t2 = time.perf_counter(); L.tinympc_get_solution(s._h, dp(X), dp(U), 0)
        t3 = time.perf_counter()
        if k >= 20:
            t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2
        x = prob.A @ x + prob.B @ U[:, 0]
    print(f"{name:16s} single-instance verbs: set_x0 {1e6*t[0]/n:5.1f} + solve {1e6*t[1]/n:5.1f} + get_solution {1e6*t[2]/n:5.1f} = {1e6*sum(t)/n:6.1f} us per tick")
    s.reset()

# The tracking loops of the reference re-send the references every tick (rocket_landing_constraints.m:86-121):
# set_x0 -> set_x_ref -> set_u_ref -> solve -> get_solution on a single-instance handle. References go through pinned
# host memory; the solve kernel rebuilds the table rows they determine, so the tick is still one launch.
for name, prob in (("quadrotor N=50", P.quadrotor(50)), ("rocket N=10", P.rocket(10)), ("rocket N=100", P.rocket(100))):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, fdyn=prob.fdyn, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if getattr(prob, "cones", None):
        s.set_cone_constraints(**prob.cones)
    if getattr(prob, "linear", None):
        s.set_linear_constraints(**prob.linear)
    x = prob.x0.copy()
    goal = np.zeros(prob.nx)
    X = np.zeros((prob.nx, prob.N), order="F"); U = np.zeros((prob.nu, prob.N - 1), order="F")
    uref = np.asfortranarray(prob.u_ref if prob.u_ref is not None else np.zeros((prob.nu, prob.N - 1)))
    t = [0.0] * 5; n = 200; total = n + 20 + prob.N
    iters = 0
    for k in range(n + 20):
        xx = np.ascontiguousarray(x)
        xref = np.asfortranarray(np.stack([prob.x0 + (goal - prob.x0) * (i + k) / (total - 1) for i in range(prob.N)], axis=1))
        t0 = time.perf_counter(); L.tinympc_set_x0(s._h, dp(xx), prob.nx, 0)
        t1 = time.perf_counter(); L.tinympc_set_x_ref(s._h, dp(xref), prob.nx, prob.N, 0)
        t2 = time.perf_counter(); L.tinympc_set_u_ref(s._h, dp(uref), prob.nu, prob.N - 1, 0)
        t3 = time.perf_counter(); L.tinympc_solve(s._h, 0)
        t4 = time.perf_counter(); L.tinympc_get_solution(s._h, dp(X), dp(U), 0)
        t5 = time.perf_counter()
        if k >= 20:
            for q, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))): t[q] += b - a
            iters += s.get_stats()["iter"]
        x = prob.A @ x + prob.B @ U[:, 0] + (prob.fdyn if prob.fdyn is not None else 0.0)
    print(f"{name:16s} tick with per-tick references: set_x0 {1e6*t[0]/n:4.1f} + set_x_ref {1e6*t[1]/n:4.1f} + set_u_ref {1e6*t[2]/n:4.1f} + solve {1e6*t[3]/n:5.1f} "
          f"+ get_solution {1e6*t[4]/n:4.1f} = {1e6*sum(t)/n:6.1f} us per tick ({iters/n:.1f} iterations/tick)")
    s.reset()

# Closed-loop SESSION: the kernel stays resident (no launch, no stream synchronisation per tick); x0 goes in and the
# solution comes back through pinned host memory. Same loop as above, tick = tinympc_session_step (+ get_solution).
for name, prob, refs in (("cartpole N=10", P.cartpole(10, True), False), ("quadrotor N=50", P.quadrotor(50), False), ("quadrotor N=50", P.quadrotor(50), True)):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.session_begin()
    x = prob.x0.copy(); goal = np.zeros(prob.nx)
    u0 = np.zeros(prob.nu)
    X = np.zeros((prob.nx, prob.N), order="F"); U = np.zeros((prob.nu, prob.N - 1), order="F")
    t = [0.0, 0.0, 0.0]; n = 300; iters = 0; total = n + 20 + prob.N
    for k in range(n + 20):
        xx = np.ascontiguousarray(x)
        xref = np.asfortranarray(np.stack([prob.x0 + (goal - prob.x0) * (i + k) / (total - 1) for i in range(prob.N)], axis=1))
        t0 = time.perf_counter()
        if refs: L.tinympc_set_x_ref(s._h, dp(xref), prob.nx, prob.N, 0)
        t1 = time.perf_counter(); L.tinympc_session_step(s._h, dp(xx), dp(u0))
        t2 = time.perf_counter(); L.tinympc_get_solution(s._h, dp(X), dp(U), 0)
        t3 = time.perf_counter()
        if k >= 20:
            t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2; iters += s.get_stats()["iter"]
        x = prob.A @ x + prob.B @ u0
    s.session_end()
    print(f"{name:16s} SESSION tick{' with per-tick x_ref' if refs else '':20s}: set_x_ref {1e6*t[0]/n:4.1f} + session_step {1e6*t[1]/n:5.1f} + get_solution {1e6*t[2]/n:4.1f} = {1e6*sum(t)/n:6.1f} us per tick ({iters/n:.1f} iterations/tick)")
    s.reset()
