"""A second, warm-started solve of a batch whose instances converged at DIFFERENT iterations in the first one, against the oracle.

A converged solve leaves the PREVIOUS iterate in v / z (admm.cpp:181-197): that is what the next solve's first dual residual
max|v - vnew| is measured against, hence what its residuals -- and, near the tolerance, its iteration count -- depend on. Kernels
that keep several instances in one wavefront keep sweeping a converged instance's lanes as a zombie until the wavefront is done;
round 3 found (through the slot-refill variant, which finishes an instance the moment it converges and agreed with the oracle
where the plain kernel did not) that those later sweeps overwrote the converged instance's stale copy, so that its canonical
v / z after the solve was a later zombie iterate: same solution, same iteration counts in every test there was, other residuals in
the next solve. Cold solves and batches that converge together cannot see it; this test can."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import golden, problem_from_golden, rel_err, settings_from_golden

import pyoracle as O

pytestmark = pytest.mark.gpu


def _wide_system(P, nx, nu, N, seed=0, a_scale=0.03, b_scale=0.1, diag=1.0):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) * diag + a_scale * rng.standard_normal((nx, nx))
    B = b_scale * rng.standard_normal((nx, nu))
    prob = P.Problem("wide", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
    return prob


CASES = [("A", "rocket", 10, 64), ("D", "rocket", 10, 64), ("E", "rocket", 10, 64), ("E", "rocket", 20, 64), ("E", "rocket", 100, 32), ("F", "rocket", 100, 8),
         ("A", "quadrotor", 50, 64), ("B", "quadrotor", 50, 64), ("C", "quadrotor", 50, 64), ("D", "quadrotor", 50, 64), ("D", "quadrotor", 20, 200),
         ("D", "wide32", 30, 64), ("D", "wide64", 20, 32), ("M", "large", 10, 48), ("E", "quadrotor", 125, 64), ("D", "quadrotor", 100, 64)]


@pytest.mark.parametrize("layout,system,N,B", CASES)
def test_second_solve_of_a_mixed_batch_matches_the_oracle(pkg, monkeypatch, layout, system, N, B):
    P = pkg.problems
    monkeypatch.setenv("TINYMPC_LAYOUT", layout)
    rng = np.random.default_rng(7)
    if system == "quadrotor":
        prob = P.quadrotor(N)
        x0s = np.asfortranarray(P.quadrotor_batch_x0(B) * rng.uniform(0.05, 3.0, B)[None, :])
    elif system == "rocket":  # cones + a linear row + fdyn + references (BASELINE config 4; the oracle is the builder's restatement there)
        prob = P.rocket(N)
        x0s = np.asfortranarray(prob.x0[:, None] * rng.uniform(0.2, 1.3, B)[None, :])
    elif system == "large":  # 84 rows: layout M, sixteen instances per matrix-core tile
        prob = _wide_system(P, 70, 14, N, a_scale=0.015, b_scale=0.08, diag=0.98)
        x0s = np.asfortranarray(rng.standard_normal((70, B)) * rng.uniform(0.02, 1.0, B)[None, :])
    else:
        prob = _wide_system(P, 24, 8, N) if system == "wide32" else _wide_system(P, 40, 8, N)
        x0s = np.asfortranarray(rng.standard_normal((prob.A.shape[0], B)) * rng.uniform(0.02, 1.0, B)[None, :])
    # (the 48- and 84-row synthetic systems and the 100-knot rocket landing converge slowly: looser tolerances, so that instances DO converge)
    tol = 3e-2 if system in ("wide64", "large") else 2e-2 if (system == "rocket" and N == 100) else 1e-3
    settings = dict(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=200 if system == "rocket" else 80, check_termination=1)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.cones:
        s.set_cone_constraints(**prob.cones)
    if prob.linear:
        s.set_linear_constraints(**prob.linear)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref)
    if prob.u_ref is not None:
        s.set_u_ref(prob.u_ref)
    s.set_x0_batch(x0s)
    s.solve()
    it0 = s.get_stats_batch()["iter"].copy()
    if system == "large":
        assert s.launch_info()["layout"] == "M"
    if system != "wide64":  # (64 lanes: one instance per wavefront, nothing to overwrite; the warm semantics are checked all the same)
        assert len(np.unique(it0)) >= 4, "the batch must converge at different iterations for this test to mean anything"
    s.solve()                      # warm, same x0: most instances now converge at the first check
    st1, sol1 = s.get_stats_batch(), s.get_solution_batch()
    s.set_x0_batch(np.asfortranarray(1.05 * x0s))
    s.solve()                      # and once more from that state, with a nudged x0
    st2, sol2 = s.get_stats_batch(), s.get_solution_batch()
    s.reset()
    for i in range(B):
        orc = O.OraclePort(prob).load_problem(prob, settings)
        orc.set_x0(x0s[:, i])
        orc.solve()
        assert orc.stats()["iter"] == it0[i]
        for st, sol, x0 in ((st1, sol1, x0s[:, i]), (st2, sol2, 1.05 * x0s[:, i])):
            orc.set_x0(x0)
            orc.solve()
            o = orc.stats()
            assert st["iter"][i] == o["iter"] and (st["status"][i] == 1) == (o["status"] == 1), (i, st["iter"][i], o)
            ref = np.array([o["pri_x"], o["dua_x"], o["pri_u"], o["dua_u"]])
            np.testing.assert_allclose(st["residuals"][:, i], ref, rtol=1e-6, atol=1e-12, err_msg=f"instance {i} (first solve: {it0[i]} iterations)")
            ox, ou = orc.solution()
            assert rel_err(sol["states"][:, :, i], ox) < 1e-9 and rel_err(sol["controls"][:, :, i], ou) < 1e-9


@pytest.mark.parametrize("layout,B", [("D", 64), ("C", 24), ("A", 40)])
def test_closed_loop_of_a_mixed_batch_matches_the_oracle_tick_by_tick(pkg, monkeypatch, layout, B):
    """The callers' real pattern (cartpole_example_mpc.m:36-44) on a batch: every tick x0 in, warm-started solve, first controls
    out, plant step. Instances start at different distances from the origin, so in every tick they converge at different
    iterations -- each must follow its own oracle closed loop exactly (iteration counts) and to 1e-9 (controls)."""
    P = pkg.problems
    monkeypatch.setenv("TINYMPC_LAYOUT", layout)
    prob = P.quadrotor(50 if layout == "D" else 20)
    rng = np.random.default_rng(3)
    x = np.asfortranarray(P.quadrotor_batch_x0(B) * rng.uniform(0.05, 2.5, B)[None, :])
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=40, check_termination=1)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    oracles = [O.OraclePort(prob).load_problem(prob, settings) for _ in range(B)]
    xo = x.copy()
    spread = 0
    for tick in range(6):
        u0 = s.mpc_step(x)
        its = s.get_stats_batch()["iter"]
        spread = max(spread, len(np.unique(its)))
        for b, o in enumerate(oracles):
            o.set_x0(xo[:, b])
            o.solve()
            assert its[b] == o.stats()["iter"], (tick, b, its[b], o.stats()["iter"])
            uo = o.solution()[1][:, 0]
            assert np.max(np.abs(u0[:, b] - uo)) <= 1e-9 * max(np.max(np.abs(uo)), 1e-3), (tick, b)
            xo[:, b] = prob.A @ xo[:, b] + prob.B @ uo
        x = np.asfortranarray(prob.A @ x + prob.B @ u0)
    assert spread >= 3
    s.reset()


@pytest.mark.parametrize("layout", ["A", "B", "C", "D", "E", "F"])
def test_warm_restarts_of_a_mixed_batch_match_the_reference_core(pkg, monkeypatch, layout):
    """The same property pinned to the REFERENCE's own core rather than to the restatement: tests/golden/quadrotor_warm_batch16.npz
    holds, for 16 quadrotor instances that converge at different iterations (6 ... 54, six of them never), three consecutive solves
    each -- cold, warm with the same x0, warm with 1.05 x0 -- as oracle/_ref computed them one instance at a time. One batched
    handle must reproduce every instance's iteration counts and statuses exactly, its residuals to 1e-6 and its solutions to 1e-9
    in every solve, whatever else shares its wavefront."""
    g = golden("quadrotor_warm_batch16")
    prob = problem_from_golden(pkg, g)
    settings = settings_from_golden(g)
    monkeypatch.setenv("TINYMPC_LAYOUT", layout)
    x0s = np.asfortranarray(g["x0s"])
    B = x0s.shape[1]
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    for k, scale in enumerate((1.0, 1.0, float(g["third_x0_scale"]))):
        s.set_x0_batch(np.asfortranarray(scale * x0s))
        s.solve()
        if k == 0:
            assert s.launch_info()["layout"] == layout
        st, sol = s.get_stats_batch(), s.get_solution_batch()
        np.testing.assert_array_equal(st["iter"], g["iters"][k], err_msg="solve %d" % k)
        np.testing.assert_array_equal(st["status"], g["status"][k], err_msg="solve %d" % k)
        np.testing.assert_allclose(st["residuals"], g["residuals"][k], rtol=1e-6, atol=1e-12, err_msg="solve %d" % k)
        assert rel_err(sol["controls"][:, 0, :], g["u0"][k]) < 1e-9, k
    assert rel_err(sol["states"], g["sol_x"]) < 1e-9 and rel_err(sol["controls"], g["sol_u"]) < 1e-9
    s.reset()
