#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE's own compiled core.

Run in the build container only (needs /root/reference):
    make -C oracle ref && python tests/golden/gen_golden.py

Every fixture holds inputs exactly as they cross the MEX boundary (expanded bounds/refs, the
settings pushed by TinyMPC.m) and the outputs of oracle/_ref/libtinympc_ref.so, i.e. of
/root/reference/src/codegen_src/tinympc/{admm,tiny_api}.cpp compiled where they lie.
A fixture is data only: no reference source text is stored.

Fixtures:
  cartpole_unconstrained   config 1: cartpole_example_one_solve.m (no bounds, wrapper defaults)
  cartpole_box_tol         config 2 variant: |u|<=0.5, tol 1e-4, max_iter 100 (oracle: 51 iterations)
  cartpole_box_200         config 2: tolerances 0, exactly 200 iterations
  quadrotor_box_200        config 3: N=50, rho=5, x in +-5, u in +-0.5, 200 forced iterations
  quadrotor_box_tol        config 3 with tol 1e-3 (in-kernel termination on a 12-state problem)
  quadrotor_batch64        config 5 prefix: 64 instances, seeded x0, cold start, 200 iterations
  cartpole_mpc_loop        warm-start semantics: 12 closed-loop ticks, tol 1e-4 (iters and u0 per tick)
  quadrotor_warm_batch16   warm restarts of a MIXED batch: 16 quadrotor instances scaled to converge at different iterations,
                           three consecutive solves each (cold; warm with the same x0; warm with 1.05 x0): iterations, status, all
                           four residuals and the solution of every solve -- what a kernel that keeps several instances in one
                           wavefront must reproduce per instance (the stale v/z a converged solve leaves behind, admm.cpp:181-197)
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as O  # noqa: E402

spec = importlib.util.spec_from_file_location("tinympc_problems", os.path.join(ROOT, "tinympc-matlab_amd", "problems.py"))
P = importlib.util.module_from_spec(spec)
sys.modules["tinympc_problems"] = P
spec.loader.exec_module(P)

TRACE_ARRAYS = ("x", "u", "znew", "vnew", "y", "g", "r", "q", "p", "d")


def base_inputs(prob, settings):
    xmn, xmx, umn, umx = prob.expanded_bounds()
    nx, nu, N = prob.nx, prob.nu, prob.N
    d = dict(A=prob.A, B=prob.B, Q=prob.Q, R=prob.R, rho=prob.rho, N=N, x0=prob.x0,
             x_min=xmn, x_max=xmx, u_min=umn, u_max=umx, has_bounds=int(prob.has_bounds()),
             Xref=prob.x_ref if prob.x_ref is not None else np.zeros((nx, N)),
             Uref=prob.u_ref if prob.u_ref is not None else np.zeros((nu, N - 1)))
    for k, v in settings.items():
        d["set_" + k] = v
    return d


def cache_outputs(ref):
    return {n: ref.get(n) for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt")}


def trace_first_iterations(prob, settings, iters=3):
    """Replays solve()'s loop body (admm.cpp:129-199) phase by phase on the reference core."""
    ref = O.OracleRef(prob).load_problem(prob, settings)
    out = {}
    for it in range(iters):
        ref.forward_pass()
        ref.update_slack()
        ref.update_dual()
        ref.update_linear_cost()
        ref.set_iter(it + 1)
        ref.termination_condition()
        ref.put("v", ref.get("vnew"))
        ref.put("z", ref.get("znew"))
        ref.backward_pass_grad()
        for n in TRACE_ARRAYS:
            out[f"it{it + 1}_{n}"] = ref.get(n)
        st = ref.stats()
        out[f"it{it + 1}_res"] = np.array([st["pri_x"], st["dua_x"], st["pri_u"], st["dua_u"]])
    return out


def single(name, prob, settings, trace=True):
    ref = O.OracleRef(prob).load_problem(prob, settings)
    data = base_inputs(prob, settings)
    data.update(cache_outputs(ref))
    rc = ref.solve()
    st = ref.stats()
    sx, su = ref.solution()
    data.update(sol_x=sx, sol_u=su, ret=rc, iter=st["iter"], status=st["status"], solved=st["solved"],
                residuals=np.array([st["pri_x"], st["dua_x"], st["pri_u"], st["dua_u"]]))
    # post-solve persistent state (what a warm-started next solve sees)
    for n in ("d", "y", "g", "v", "z"):
        data["post_" + n] = ref.get(n)
    if trace:
        data.update(trace_first_iterations(prob, settings))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{name}: iter={st['iter']} status={st['status']} u0={su[:, 0]} -> {os.path.getsize(path)} B")


def batch64():
    prob = P.quadrotor(50)
    settings = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=200, check_termination=1)
    x0s = P.quadrotor_batch_x0(64)
    ref = O.OracleRef(prob).load_problem(prob, settings)
    nx, nu, N = prob.nx, prob.nu, prob.N
    sx = np.zeros((nx, N, 64), order="F")
    su = np.zeros((nu, N - 1, 64), order="F")
    its = np.zeros(64, dtype=np.int32)
    res = np.zeros((4, 64), order="F")
    for b in range(64):
        ref.reset_workspace()
        ref.set_x0(x0s[:, b])
        ref.solve()
        sx[:, :, b], su[:, :, b] = ref.solution()
        st = ref.stats()
        its[b] = st["iter"]
        res[:, b] = [st["pri_x"], st["dua_x"], st["pri_u"], st["dua_u"]]
    data = base_inputs(prob, settings)
    data.update(x0s=x0s, sol_x=sx, sol_u=su, iters=its, residuals=res, seed=P.BATCH_SEED)
    path = os.path.join(HERE, "quadrotor_batch64.npz")
    np.savez_compressed(path, **data)
    print(f"quadrotor_batch64: iters={its[:4]}.. -> {os.path.getsize(path)} B")


def mpc_loop():
    """Closed loop as in examples/cartpole_example_mpc.m:36-44 but without noise: the workspace
    persists between solves, including the one-iteration-stale v/z that a
    converged solve leaves behind (admm.cpp:181-197)."""
    prob = P.cartpole(10, True)
    prob.u_min, prob.u_max = np.array([-5.0]), np.array([5.0])
    prob.rho = 0.1
    settings = dict(abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=100, check_termination=1)
    ref = O.OracleRef(prob).load_problem(prob, settings)
    x = prob.x0.copy()
    ticks = 12
    iters = np.zeros(ticks, dtype=np.int32)
    u0s = np.zeros((prob.nu, ticks))
    xs = np.zeros((prob.nx, ticks + 1))
    xs[:, 0] = x
    dua = np.zeros((2, ticks))
    for k in range(ticks):
        ref.set_x0(x)
        ref.solve()
        st = ref.stats()
        iters[k] = st["iter"]
        dua[:, k] = [st["dua_x"], st["dua_u"]]
        _, su = ref.solution()
        u0s[:, k] = su[:, 0]
        x = prob.A @ x + prob.B @ su[:, 0]
        xs[:, k + 1] = x
    data = base_inputs(prob, settings)
    data.update(cache_outputs(ref))
    data.update(ticks=ticks, iters=iters, u0s=u0s, xs=xs, dual_residuals=dua)
    path = os.path.join(HERE, "cartpole_mpc_loop.npz")
    np.savez_compressed(path, **data)
    print(f"cartpole_mpc_loop: iters={iters} -> {os.path.getsize(path)} B")


def warm_batch16():
    """Three consecutive solves per instance on the reference core; every instance has its own workspace (a fresh cold start),
    exactly what a batched handle promises per instance."""
    prob = P.quadrotor(50)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=150, check_termination=1)
    B = 16
    scales = np.random.default_rng(11).uniform(0.05, 2.2, B)
    x0s = np.asfortranarray(P.quadrotor_batch_x0(B) * scales[None, :])
    nx, nu, N = prob.nx, prob.nu, prob.N
    its = np.zeros((3, B), dtype=np.int32)
    status = np.zeros((3, B), dtype=np.int32)
    res = np.zeros((3, 4, B))
    u0 = np.zeros((3, nu, B))
    sx = np.zeros((nx, N, B))       # (full trajectories of the THIRD solve only: the fixture stays small)
    su = np.zeros((nu, N - 1, B))
    ref = O.OracleRef(prob).load_problem(prob, settings)
    for b in range(B):
        ref.reset_workspace()
        for k, x0 in enumerate((x0s[:, b], x0s[:, b], 1.05 * x0s[:, b])):
            ref.set_x0(x0)
            ref.solve()
            st = ref.stats()
            its[k, b], status[k, b] = st["iter"], st["status"]
            res[k, :, b] = [st["pri_x"], st["dua_x"], st["pri_u"], st["dua_u"]]
            sx[:, :, b], su[:, :, b] = ref.solution()
            u0[k, :, b] = su[:, 0, b]
    assert len(np.unique(its[0])) >= 6, its[0]
    data = base_inputs(prob, settings)
    data.update(x0s=x0s, iters=its, status=status, residuals=res, u0=u0, sol_x=sx, sol_u=su, third_x0_scale=1.05)
    path = os.path.join(HERE, "quadrotor_warm_batch16.npz")
    np.savez_compressed(path, **data)
    print(f"quadrotor_warm_batch16: iters={its.tolist()} -> {os.path.getsize(path)} B")


def main():
    if not O.ref_available():
        sys.exit("oracle/_ref/libtinympc_ref.so missing: run `make -C oracle ref` first")
    wrapper_defaults = dict(abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=100, check_termination=1)
    forced = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=200, check_termination=1)
    single("cartpole_unconstrained", P.cartpole(20, False), wrapper_defaults)
    single("cartpole_box_tol", P.cartpole(20, True), wrapper_defaults)
    single("cartpole_box_200", P.cartpole(20, True), forced)
    single("quadrotor_box_200", P.quadrotor(50), forced)
    single("quadrotor_box_tol", P.quadrotor(50), dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=500,
                                                      check_termination=1))
    batch64()
    mpc_loop()
    warm_batch16()


if __name__ == "__main__":
    main()
