"""The Riccati helpers the reference implements in MATLAB inside its class (TinyMPC.m:194-241, 336-366;
exercised by the reference's tests/test_cache.m): CPU tests of the restatement, GPU tests of the device path."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import matlab_class_oracle as M


def _systems(pkg):
    P = pkg.problems
    return {"cartpole": P.cartpole(20, True), "quadrotor": P.quadrotor(20), "rocket": P.rocket()}


# ---------------------------------------------------------------------------------------------
# CPU: the restatement is self-consistent (MATLAB absent: see the module header of the oracle)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cartpole", "quadrotor", "rocket"])
def test_oracle_lqr_branches_agree_and_solve_the_dare(pkg, name):
    p = _systems(pkg)[name]
    nx, nu = p.B.shape
    K, P_, C1, C2 = M.solve_lqr(p.A, p.B, p.Q, p.R, p.rho)
    Qr, Rr = p.Q + p.rho * np.eye(nx), p.R + p.rho * np.eye(nu)
    assert M.dare_residual(p.A, p.B, Qr, Rr, P_) < 1e-10
    Ki, Pi, C1i, C2i = M.solve_lqr(p.A, p.B, p.Q, p.R, p.rho, iterative=True)
    assert rel_err(Ki, K) < 1e-6 and rel_err(Pi, P_) < 1e-6 and rel_err(C1i, C1) < 1e-6 and rel_err(C2i, C2) < 1e-6
    assert np.abs(np.linalg.eigvals(C2)).max() < 1.0  # stabilising gain: u = -K x
    Kc, Pc, Qi, Am, it = M.compute_cache_terms(p.A, p.B, p.Q, p.R, p.rho)
    assert 1 < it < 5000
    assert rel_err(Kc, K) < 1e-6 and rel_err(Pc, P_) < 1e-6 and rel_err(Qi, C1) < 1e-6 and rel_err(Am, C2) < 1e-6


def test_oracle_cache_terms_of_the_reference_test_case(pkg):
    """tests/test_cache.m of the reference: cartpole, N = 2, default rho = 1 -- must run and return finite terms."""
    p = pkg.problems.cartpole(2, False)
    K, P_, Qi, Am, it = M.compute_cache_terms(p.A, p.B, p.Q, p.R, 1.0)
    assert K.shape == (1, 4) and P_.shape == (4, 4) and Qi.shape == (1, 1) and Am.shape == (4, 4)
    assert all(np.isfinite(m).all() for m in (K, P_, Qi, Am))


def test_oracle_sensitivity_matches_central_differences(pkg):
    p = pkg.problems.quadrotor(20)
    dK, dP, dC1, dC2 = M.compute_sensitivity(p.A, p.B, p.Q, p.R, p.rho)
    h = 1e-4
    hi = M.solve_lqr(p.A, p.B, p.Q, p.R, p.rho + h)
    lo = M.solve_lqr(p.A, p.B, p.Q, p.R, p.rho - h)
    for got, a, b in zip((dK, dP, dC1, dC2), hi, lo):
        want = (a - b) / (2 * h)
        assert np.abs(got - want).max() < 2e-3 * max(1.0, np.abs(want).max())


# ---------------------------------------------------------------------------------------------
# GPU: device path against the restatement
# ---------------------------------------------------------------------------------------------
def _new(pkg, prob, **kw):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, **kw)
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cartpole", "quadrotor", "rocket"])
def test_compute_cache_terms_matches_the_class_recursion(pkg, name):
    p = _systems(pkg)[name]
    s = _new(pkg, p)
    K, P_, Qi, Am = s.compute_cache_terms()
    Ko, Po, Qio, Amo, it = M.compute_cache_terms(p.A, p.B, p.Q, p.R, p.rho)
    assert abs(s.cache_terms_iters - it) <= 1  # the stopping norm sits at 1e-10: one step of slack
    for got, want in ((K, Ko), (P_, Po), (Qi, Qio), (Am, Amo)):
        assert rel_err(got, want) < 1e-8
    s.reset()


@pytest.mark.gpu
def test_compute_cache_terms_full_q_and_reference_test_case(pkg):
    """Off-diagonal Q and R entries take part here (unlike in setup, which keeps diagonals only), and the
    reference's test_cache.m case (N = 2) runs."""
    p = pkg.problems.cartpole(2, False)
    s = _new(pkg, p)
    K, P_, Qi, Am = s.compute_cache_terms()
    Ko, Po, Qio, Amo, _ = M.compute_cache_terms(p.A, p.B, p.Q, p.R, p.rho)
    assert rel_err(K, Ko) < 1e-8 and rel_err(P_, Po) < 1e-8
    s.reset()
    rng = np.random.default_rng(3)
    q = pkg.problems.quadrotor(10)
    L = rng.normal(size=(12, 12)) * 0.3
    Qf = q.Q + L @ L.T
    Lr = rng.normal(size=(4, 4)) * 0.3
    Rf = q.R + Lr @ Lr.T
    s = pkg.TinyMPC()
    s.setup(q.A, q.B, Qf, Rf, q.N, rho=q.rho)
    K, P_, Qi, Am = s.compute_cache_terms()
    Ko, Po, Qio, Amo, _ = M.compute_cache_terms(q.A, q.B, Qf, Rf, q.rho)
    for got, want in ((K, Ko), (P_, Po), (Qi, Qio), (Am, Amo)):
        assert rel_err(got, want) < 1e-8
    s.reset()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cartpole", "quadrotor", "rocket"])
def test_solve_lqr_is_the_dare_solution(pkg, name):
    p = _systems(pkg)[name]
    nx, nu = p.B.shape
    s = _new(pkg, p)
    for rho in (p.rho, 3.7 * p.rho):
        K, P_, C1, C2 = s.solve_lqr(rho)
        Ko, Po, C1o, C2o = M.solve_lqr(p.A, p.B, p.Q, p.R, rho)
        for got, want in ((K, Ko), (P_, Po), (C1, C1o), (C2, C2o)):
            assert rel_err(got, want) < 1e-9
        assert M.dare_residual(p.A, p.B, p.Q + rho * np.eye(nx), p.R + rho * np.eye(nu), P_) < 1e-11
    s.reset()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cartpole", "quadrotor", "rocket"])
def test_sensitivity_matches_forward_differences(pkg, name):
    """Tolerance: both sides difference two DARE solutions 1e-6 apart, so rounding in P (~1e-13 relative) is
    amplified to ~1e-7 |P| per entry; 1e-3 of the largest derivative entry bounds it with margin."""
    p = _systems(pkg)[name]
    s = _new(pkg, p)
    got = s.compute_sensitivity_autograd()
    want = M.compute_sensitivity(p.A, p.B, p.Q, p.R, p.rho)
    for g, w, n in zip(got, want, ("dK", "dP", "dC1", "dC2")):
        assert g.shape == w.shape
        assert np.abs(g - w).max() < 1e-3 * max(np.abs(w).max(), 1e-3), n
    # the derivatives feed set_sensitivity_matrices unchanged (TinyMPC.m:186-192)
    s.set_sensitivity_matrices(*got)
    s.reset()


@pytest.mark.gpu
def test_cache_terms_round_trip_into_the_solver(pkg):
    """compute_cache_terms -> set_cache_terms -> solve: the solve uses the installed (fully converged) cache
    and still reaches the same optimum as with the truncated one the core computes (looser: 1e-3)."""
    p = pkg.problems.cartpole(20, True)
    a = _new(pkg, p, max_iter=500, abs_pri_tol=1e-6, abs_dua_tol=1e-6)
    b = _new(pkg, p, max_iter=500, abs_pri_tol=1e-6, abs_dua_tol=1e-6)
    for s in (a, b):
        s.set_bound_constraints(p.x_min, p.x_max, p.u_min, p.u_max)
        s.set_x0(p.x0)
    a.solve()
    ref = a.get_solution()
    # the class recursion adds rho once, the core twice: use the core's rho convention
    c = pkg.TinyMPC()
    c.setup(p.A, p.B, p.Q + p.rho * np.eye(4), p.R + p.rho * np.eye(1), p.N, rho=p.rho)
    K, P_, Qi, Am = c.compute_cache_terms()
    b.set_cache_terms(K, P_, Qi, Am)
    b.solve()
    got = b.get_solution()
    assert rel_err(got["controls"], ref["controls"]) < 1e-3
    for s in (a, b, c):
        s.reset()
