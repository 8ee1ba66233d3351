"""Adaptive rho (reference admm.cpp:117-174, rho_benchmark.cpp).

Pinning chain: the reference's own core, compiled with zero-initialised automatic variables (oracle/Makefile target
ref_zeroinit -- the plain build reads RhoAdapter::matrices_initialized uninitialised and crashes), pins the plain-C
restatement (CPU tests); the restatement and the reference then check the device kernel (GPU tests)."""
from __future__ import annotations

import os

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

TOL = 1e-9


def _need_zeroinit():
    if not os.path.exists(O.REF_ZEROINIT_LIB):
        pytest.fail("oracle/_ref/libtinympc_ref_zeroinit.so missing: run `python __graft_entry__.py` where /root/reference exists")


def _ref_and_port(prob, settings, adaptive=(1.0, 100.0, True)):
    ref = O.OracleRef(prob, zeroinit=True).load_problem(prob, settings)
    ref.set_adaptive_rho(True, *adaptive)
    port = O.OraclePort(prob).load_problem(prob, settings)
    port.set_adaptive_rho(True, *adaptive)
    port.set_sensitivity(ref.get("dKinf_drho"), ref.get("dPinf_drho"))  # the snapshot's hard-coded quadrotor tables
    return ref, port


# ---------------------------------------------------------------------------------------------
# CPU: restatement vs the reference
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,settings", [(10, dict(max_iter=30, abs_pri_tol=1e-9, abs_dua_tol=1e-9)),
                                        (20, dict(max_iter=200, abs_pri_tol=1e-3, abs_dua_tol=1e-3)),
                                        (50, dict(max_iter=60, abs_pri_tol=0.0, abs_dua_tol=0.0, check_termination=4))])
def test_oracle_adaptive_rho_matches_the_reference_core(pkg, N, settings):
    _need_zeroinit()
    prob = pkg.problems.quadrotor(N)
    ref, port = _ref_and_port(prob, settings)
    for solve in range(3):  # rho and the Taylor-updated cache persist from one solve to the next
        x0 = prob.x0 * (1.0 - 0.3 * solve)
        ref.set_x0(x0)
        port.set_x0(x0)
        assert ref.solve() == port.solve()
        a, b = ref.stats(), port.stats()
        assert (a["iter"], a["status"]) == (b["iter"], b["status"])
        assert abs(a["rho"] - b["rho"]) < 1e-11 * a["rho"]
        assert rel_err(port.solution()[0], ref.solution()[0]) < 1e-10
        assert rel_err(port.solution()[1], ref.solution()[1]) < 1e-10
        assert rel_err(port.get("Kinf"), ref.get("Kinf")) < 1e-11 and rel_err(port.get("Pinf"), ref.get("Pinf")) < 1e-11
    assert ref.stats()["rho"] != prob.rho  # the test would be vacuous if rho never moved


def test_oracle_adaptive_rho_without_clipping_and_bounds_of_the_clip(pkg):
    _need_zeroinit()
    prob = pkg.problems.quadrotor(12)
    settings = dict(max_iter=40, abs_pri_tol=0.0, abs_dua_tol=0.0)
    for adaptive in ((1.0, 100.0, False), (4.5, 5.5, True), (6.0, 50.0, True)):
        ref, port = _ref_and_port(prob, settings, adaptive)
        ref.set_x0(prob.x0)
        port.set_x0(prob.x0)
        ref.solve()
        port.solve()
        assert abs(ref.stats()["rho"] - port.stats()["rho"]) < 1e-11 * ref.stats()["rho"]
        assert rel_err(port.solution()[1], ref.solution()[1]) < 1e-10
        if adaptive[2]:
            assert adaptive[0] <= port.stats()["rho"] <= adaptive[1]


def test_oracle_adaptation_off_is_the_plain_solve(pkg):
    prob = pkg.problems.cartpole(20, True)
    settings = dict(max_iter=50, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
    a = O.OraclePort(prob).load_problem(prob, settings)
    b = O.OraclePort(prob).load_problem(prob, settings)
    b.set_adaptive_rho(False)
    b.set_sensitivity(np.ones((1, 4)), np.ones((4, 4)))
    a.set_x0(prob.x0)
    b.set_x0(prob.x0)
    a.solve()
    b.solve()
    np.testing.assert_array_equal(a.solution()[1], b.solution()[1])


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
def _solver(pkg, prob, batch=1, **settings):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, **settings)
    if prob.has_bounds():
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("N,settings", [(10, dict(max_iter=30, abs_pri_tol=1e-9, abs_dua_tol=1e-9)),
                                        (20, dict(max_iter=200, abs_pri_tol=1e-3, abs_dua_tol=1e-3)),
                                        (50, dict(max_iter=60, abs_pri_tol=0.0, abs_dua_tol=0.0, check_termination=4))])
def test_device_adaptive_rho_matches_the_reference_core(pkg, N, settings):
    _need_zeroinit()
    prob = pkg.problems.quadrotor(N)
    ref = O.OracleRef(prob, zeroinit=True).load_problem(prob, settings)
    ref.set_adaptive_rho(True, 1.0, 100.0, True)
    s = _solver(pkg, prob, adaptive_rho=True, adaptive_rho_min=1.0, adaptive_rho_max=100.0, **settings)
    s.set_sensitivity_matrices(ref.get("dKinf_drho"), ref.get("dPinf_drho"), ref.get("dC1_drho"), ref.get("dC2_drho"))
    for solve in range(3):
        x0 = prob.x0 * (1.0 - 0.3 * solve)
        ref.set_x0(x0)
        s.set_x0(x0)
        ref.solve()
        s.solve()
        st, rs = s.get_stats(), ref.stats()
        assert (st["iter"], st["status"]) == (rs["iter"], rs["status"])
        assert abs(s.get_rho_batch()[0] - rs["rho"]) < 1e-10 * rs["rho"]
        sol = s.get_solution()
        assert rel_err(sol["states"], ref.solution()[0]) < TOL
        assert rel_err(sol["controls"], ref.solution()[1]) < TOL
    assert s.get_rho_batch()[0] != prob.rho
    s.reset()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["cartpole", "quadrotor"])
def test_device_adaptive_rho_batch_with_computed_sensitivities(pkg, which):
    """Every instance of a batch adapts its own rho; sensitivities from compute_sensitivity_autograd; a tracking
    reference exercises the Pinf-dependent terminal term; second solve warm-started with the adapted rho."""
    prob = pkg.problems.cartpole(20, True) if which == "cartpole" else pkg.problems.quadrotor(15)
    nx, nu, N = prob.nx, prob.nu, prob.N
    count = 9
    rng = np.random.default_rng(11)
    x0s = prob.x0[:, None] * rng.uniform(0.2, 1.5, size=(1, count)) * rng.choice([-1.0, 1.0], size=(nx, count))
    Xref = np.tile((0.05 * rng.normal(size=(nx, 1))), (1, N))
    settings = dict(max_iter=45, abs_pri_tol=1e-5, abs_dua_tol=1e-5)
    s = _solver(pkg, prob, batch=count, adaptive_rho=True, adaptive_rho_min=0.2, adaptive_rho_max=40.0, **settings)
    dK, dP, dC1, dC2 = s.compute_sensitivity_autograd()
    s.set_sensitivity_matrices(dK, dP, dC1, dC2)
    s.set_x_ref(Xref)
    oracles = []
    for b in range(count):
        o = O.OraclePort(prob).load_problem(prob, settings)
        o.set_adaptive_rho(True, 0.2, 40.0, True)
        o.set_sensitivity(dK, dP)
        o.set_x_ref(Xref)
        oracles.append(o)
    for solve in range(3):  # (the third: warm from a state in which the instances had converged at different iterations)
        s.set_x0_batch(x0s * (1.0 - 0.4 * min(solve, 1)))
        s.solve()
        sol, st, rho = s.get_solution_batch(), s.get_stats_batch(), s.get_rho_batch()
        for b, o in enumerate(oracles):
            o.set_x0(x0s[:, b] * (1.0 - 0.4 * min(solve, 1)))
            o.solve()
            assert st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"], (solve, b)
            if st["iter"][b] > 0:  # the residuals of the last check: they depend on the canonical v|z the PREVIOUS solve left
                want = np.array([o.stats()[k] for k in ("pri_x", "dua_x", "pri_u", "dua_u")])
                np.testing.assert_allclose(st["residuals"][:, b], want, rtol=1e-6, atol=1e-12, err_msg=f"solve {solve}, instance {b}")
            assert abs(rho[b] - o.stats()["rho"]) < 1e-9 * rho[b], (solve, b)
            assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL, (solve, b)
            assert rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, (solve, b)
    assert np.unique(np.round(rho, 6)).size > 1  # instances really ended on different rho
    s.reset_workspace()
    np.testing.assert_array_equal(s.get_rho_batch(), np.full(count, prob.rho))
    s.reset()


@pytest.mark.gpu
def test_device_adaptive_rho_wide_system_and_long_horizon(pkg):
    """W = 32 and 64 lanes per instance (nx + nu = 20 and 46; the 64-lane form rebuilds its one operator row array at every sweep)
    and a horizon whose state lives in HBM scratch (GMEM variant)."""
    rng = np.random.default_rng(5)

    def wide(nx, nu):
        A = np.eye(nx) + 0.05 * rng.normal(size=(nx, nx))
        A *= 0.97 / max(1.0, np.abs(np.linalg.eigvals(A)).max())
        B = 0.2 * rng.normal(size=(nx, nu))
        return pkg.problems.Problem(name="wide", A=A, B=B, Q=np.diag(rng.uniform(1, 5, nx)), R=np.diag(rng.uniform(0.5, 2, nu)), N=12,
                                    rho=2.0, x0=rng.normal(size=nx), u_min=np.full(nu, -0.4), u_max=np.full(nu, 0.4))

    for prob, iters in ((wide(14, 6), 40), (wide(36, 10), 40), (pkg.problems.cartpole(300, True), 25)):
        settings = dict(max_iter=iters, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s = _solver(pkg, prob, batch=3, adaptive_rho=True, adaptive_rho_min=0.5, adaptive_rho_max=20.0, **settings)
        dK, dP, dC1, dC2 = s.compute_sensitivity_autograd()
        s.set_sensitivity_matrices(dK, dP, dC1, dC2)
        x0s = np.stack([prob.x0, -0.5 * prob.x0, 0.25 * prob.x0], axis=1)
        s.set_x0_batch(x0s)
        s.solve()
        sol, rho = s.get_solution_batch(), s.get_rho_batch()
        for b in range(3):
            o = O.OraclePort(prob).load_problem(prob, settings)
            o.set_adaptive_rho(True, 0.5, 20.0, True)
            o.set_sensitivity(dK, dP)
            o.set_x0(x0s[:, b])
            o.solve()
            assert abs(rho[b] - o.stats()["rho"]) < 1e-9 * rho[b]
            assert rel_err(sol["controls"][:, :, b], o.solution()[1]) < 1e-8
        s.reset()


@pytest.mark.gpu
def test_adaptive_rho_switching_and_unsupported_combination(pkg):
    prob = pkg.problems.cartpole(20, True)
    settings = dict(max_iter=30, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s = _solver(pkg, prob, **settings)
    o = O.OraclePort(prob).load_problem(prob, settings)
    s.set_x0(prob.x0)
    o.set_x0(prob.x0)
    s.solve()  # plain solve first (layout B), then adaptive on the same handle: the persistent state is shared
    o.solve()
    s.update_settings(adaptive_rho=True, adaptive_rho_min=0.1, adaptive_rho_max=10.0)
    o.set_adaptive_rho(True, 0.1, 10.0, True)  # zero sensitivities on both sides: only rho itself moves
    s.solve()
    o.solve()
    assert abs(s.get_rho_batch()[0] - o.stats()["rho"]) < 1e-9 * o.stats()["rho"] and o.stats()["rho"] != prob.rho
    assert rel_err(s.get_solution()["controls"], o.solution()[1]) < TOL
    s.reset()
    rocket = pkg.problems.rocket(10)
    r = pkg.TinyMPC()
    r.setup(rocket.A, rocket.B, rocket.Q, rocket.R, rocket.N, rho=rocket.rho, fdyn=rocket.fdyn, max_iter=5, adaptive_rho=True)
    r.set_cone_constraints(**rocket.cones)
    r.set_x0(rocket.x0)
    with pytest.raises(pkg.TinyMPCError) as ei:
        r.solve()
    assert ei.value.code == pkg._lib.ERR_UNSUPPORTED
    r.reset()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["cartpole20", "quadrotor15", "quadrotor50"])
def test_device_adaptive_rho_on_layout_d(pkg, monkeypatch, which):
    """Large batches adapt rho on the run-time specialised layout D (rho, its operator rows and pNref per lane; the KKT norms
    ride on every fifth forward sweep): every instance against the restatement (iterations, status, final rho, trajectories),
    cold and warm-started with the adapted rho; then the same on k_admm_solve_adapt (TINYMPC_JIT=0)."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    prob = pkg.problems.cartpole(20, True) if which == "cartpole20" else pkg.problems.quadrotor(int(which[9:]))
    nx, nu, N = prob.nx, prob.nu, prob.N
    count = 1301
    rng = np.random.default_rng(3)
    x0s = prob.x0[:, None] * rng.uniform(0.2, 1.5, size=(1, count)) * rng.choice([-1.0, 1.0], size=(nx, count))
    Xref = np.tile((0.05 * rng.normal(size=(nx, 1))), (1, N))
    settings = dict(max_iter=45, abs_pri_tol=1e-5, abs_dua_tol=1e-5, check_termination=1 if which != "quadrotor15" else 2)
    got = {}
    for jit in ("1", "0"):
        monkeypatch.setenv("TINYMPC_JIT", jit)
        s = _solver(pkg, prob, batch=count, adaptive_rho=True, adaptive_rho_min=0.2, adaptive_rho_max=40.0, **settings)
        dK, dP, dC1, dC2 = s.compute_sensitivity_autograd()
        s.set_sensitivity_matrices(dK, dP, dC1, dC2)
        s.set_x_ref(Xref)
        out = []
        for solve in range(2):
            s.set_x0_batch(x0s * (1.0 - 0.4 * solve))
            s.solve()
            out.append((s.get_solution_batch(), s.get_stats_batch(), s.get_rho_batch().copy()))
        assert (s.launch_info()["layout"] == "D") == (jit == "1")
        got[jit] = out
        s.reset()
    sample = list(range(0, count, 13)) + [count - 2, count - 1]
    oracles = {}
    for b in sample:
        o = O.OraclePort(prob).load_problem(prob, settings)
        o.set_adaptive_rho(True, 0.2, 40.0, True)
        o.set_sensitivity(dK, dP)
        o.set_x_ref(Xref)
        oracles[b] = o
    for solve in range(2):
        sol, st, rho = got["1"][solve]
        for b, o in oracles.items():
            o.set_x0(x0s[:, b] * (1.0 - 0.4 * solve))
            o.solve()
            assert st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"], (solve, b)
            assert abs(rho[b] - o.stats()["rho"]) < 1e-9 * rho[b], (solve, b)
            assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL, (solve, b)
            assert rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, (solve, b)
        sol0, st0, rho0 = got["0"][solve]
        np.testing.assert_array_equal(st["iter"], st0["iter"])
        np.testing.assert_array_equal(st["status"], st0["status"])
        np.testing.assert_allclose(rho, rho0, rtol=1e-9)
        assert rel_err(sol["controls"], sol0["controls"]) < TOL
        np.testing.assert_allclose(st["residuals"], st0["residuals"], rtol=1e-6, atol=1e-12)  # (the dual norms carry the rho of the last check)
    assert np.unique(np.round(got["1"][1][2], 6)).size > 1  # instances really ended on different rho
