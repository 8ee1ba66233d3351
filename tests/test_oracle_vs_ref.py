"""CPU tests: restatement vs the reference's own compiled core (oracle/_ref), phase by phase on
random states. Skipped where oracle/_ref was never built (it needs /root/reference); the golden
fixtures in tests/golden cover the same ground everywhere else."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.skipif(not (O.ref_available() and O.port_available()),
                                reason="oracle/_ref/libtinympc_ref.so not built (needs /root/reference)")


def _pair(prob, settings):
    return (O.OraclePort(prob).load_problem(prob, settings), O.OracleRef(prob).load_problem(prob, settings))


@pytest.mark.parametrize("which", ["cartpole", "quadrotor"])
def test_each_phase_on_random_state(pkg, which):
    P = pkg.problems
    prob = P.cartpole(20, True) if which == "cartpole" else P.quadrotor(50)
    a, b = _pair(prob, dict(max_iter=10))
    rng = np.random.default_rng(7)
    nx, nu, N = prob.nx, prob.nu, prob.N
    for n, shape in (("x", (nx, N)), ("g", (nx, N)), ("v", (nx, N)), ("p", (nx, N)), ("q", (nx, N)),
                     ("u", (nu, N - 1)), ("y", (nu, N - 1)), ("z", (nu, N - 1)), ("d", (nu, N - 1)),
                     ("r", (nu, N - 1)), ("Xref", (nx, N)), ("Uref", (nu, N - 1))):
        arr = rng.standard_normal(shape)
        a.put(n, arr)
        b.put(n, arr)
    for phase, outs in (("forward_pass", ("x", "u")), ("update_slack", ("znew", "vnew")),
                        ("update_dual", ("y", "g")), ("update_linear_cost", ("r", "q", "p")),
                        ("backward_pass_grad", ("d", "p"))):
        getattr(a, phase)()
        getattr(b, phase)()
        for n in outs:
            assert rel_err(a.get(n), b.get(n)) < 1e-12, (phase, n)
            a.put(n, b.get(n))  # re-sync so that every phase is checked in isolation
    a.set_iter(3)
    b.set_iter(3)
    assert a.termination_condition() == b.termination_condition()
    sa, sb = a.stats(), b.stats()
    for k in ("pri_x", "dua_x", "pri_u", "dua_u"):
        assert abs(sa[k] - sb[k]) <= 1e-12 * max(1.0, abs(sb[k])), k


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_stable_systems(pkg, seed):
    """Random problems: cache and full solves agree with the reference core."""
    P = pkg.problems
    rng = np.random.default_rng(seed)
    nx, nu, N = int(rng.integers(2, 9)), int(rng.integers(1, 4)), int(rng.integers(3, 30))
    A = np.eye(nx) + 0.05 * rng.standard_normal((nx, nx))
    B = 0.1 * rng.standard_normal((nx, nu))
    prob = P.Problem("rand", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N,
                     float(rng.uniform(0.5, 3)), rng.standard_normal(nx))
    prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
    prob.x_ref = 0.1 * rng.standard_normal((nx, N))
    prob.u_ref = 0.05 * rng.standard_normal((nu, N - 1))
    a, b = _pair(prob, dict(max_iter=80, abs_pri_tol=1e-5, abs_dua_tol=1e-5))
    for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
        assert rel_err(a.get(n), b.get(n)) < 1e-10, n
    assert a.solve() == b.solve()
    assert a.stats()["iter"] == b.stats()["iter"]
    assert rel_err(a.get("sol_x"), b.get("sol_x")) < 1e-9
    assert rel_err(a.get("sol_u"), b.get("sol_u")) < 1e-9


def test_set_cache_terms_override(pkg):
    """set_cache_terms verb (bindings.cpp:364-405): the sweeps use whatever cache is installed."""
    prob = pkg.problems.cartpole(20, True)
    a, b = _pair(prob, dict(max_iter=40))
    K, Pm, Qi, Am = b.get("Kinf") * 1.01, b.get("Pinf") * 0.99, b.get("Quu_inv") * 1.02, b.get("AmBKt") * 0.995
    a.set_cache_terms(K, Pm, Qi, Am)
    b.set_cache_terms(K, Pm, Qi, Am)
    a.solve()
    b.solve()
    assert a.stats()["iter"] == b.stats()["iter"]
    assert rel_err(a.get("sol_u"), b.get("sol_u")) < 1e-10
