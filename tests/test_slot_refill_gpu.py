"""Slot refill (layout D, `REFILL`; tinympc_solve_d.hip / tinympc_plan.hip): batches larger than the device holds at once, with
tolerances that can be met, run as ONE resident set of wavefronts whose 16-lane rows take the next instance of the batch as soon
as theirs has finished. The arithmetic of an instance does not change, so everything a solve returns must be BIT-IDENTICAL to the
plain kernel's (TINYMPC_REFILL=0), cold and warm, whatever the order in which rows happen to pick instances up; a seeded sample
is checked against the oracle as well (iteration counts exact, 1e-9 on the trajectories: the bar of test_hip_parity.py)."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _batch(pkg, prob, B, settings, x0s, refill, monkeypatch):
    monkeypatch.setenv("TINYMPC_REFILL", "1" if refill else "0")
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(x0s)
    return s


def _everything(s):
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    return sol["states"].copy(), sol["controls"].copy(), st["iter"].copy(), st["status"].copy(), st["residuals"].copy()


@pytest.mark.parametrize("N,B,ct,max_iter", [(50, 9001, 1, 60), (50, 12288, 3, 45), (20, 10000, 1, 60), (33, 9500, 2, 50)])
def test_slot_refill_is_bit_identical_to_the_plain_kernel(pkg, monkeypatch, N, B, ct, max_iter):
    """N=50 is the compiled-in kernel, the other horizons its run-time specialisations. Ragged batch (9,001), check_termination
    that does not divide max_iter, instances that converge after 3 iterations next to ones that hit max_iter; then a second,
    warm-started solve from the state the first one left (the refilled rows load it from HBM)."""
    P = pkg.problems
    prob = P.quadrotor(N)
    rng = np.random.default_rng(B + N)
    x0s = np.asfortranarray(P.quadrotor_batch_x0(B) * rng.uniform(0.05, 3.0, B)[None, :])
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=max_iter, check_termination=ct)
    got = {}
    for refill in (False, True):
        s = _batch(pkg, prob, B, settings, x0s, refill, monkeypatch)
        assert s.launch_info()["layout"] == "D"
        assert ("slot-refill" in s.jit_info()) == refill
        s.solve()
        cold = _everything(s)
        s.set_x0_batch(np.asfortranarray(0.9 * x0s))
        s.solve()
        got[refill] = (cold, _everything(s))
        s.reset()
    for k, name in enumerate(("cold", "warm")):
        for a, b, what in zip(got[False][k], got[True][k], ("states", "controls", "iterations", "status", "residuals")):
            np.testing.assert_array_equal(a, b, err_msg=f"{name} solve: {what}")
    it = got[True][0][2]
    assert it.min() < max_iter and it.max() == max_iter and len(np.unique(it)) > 5  # (a spread worth refilling for)
    assert np.all(got[True][0][3][it < max_iter] == 1)
    # against the oracle, a seeded sample (cold solve)
    sample = np.random.default_rng(1).choice(B, size=16, replace=False)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, ost, _ = orc.solve_batch(x0s[:, sample])
    np.testing.assert_array_equal(got[True][0][2][sample], oit)
    np.testing.assert_array_equal(got[True][0][3][sample] == 1, np.asarray(ost) == 1)
    assert rel_err(got[True][0][0][:, :, sample], ox) < TOL
    assert rel_err(got[True][0][1][:, :, sample], ou) < TOL


def test_slot_refill_is_only_taken_where_it_can_pay(pkg, monkeypatch):
    """Forced iteration counts (tolerances nothing can meet: every instance runs max_iter, nothing to balance), batches the device
    holds at once, and TINYMPC_REFILL=0 keep the plain kernel; TINYMPC_REFILL=1 forces the variant for any batch beyond one
    resident set -- with forced iteration counts it must still return what the plain kernel returns."""
    P = pkg.problems
    prob = P.quadrotor(50)
    B = 9000
    x0s = P.quadrotor_batch_x0(B)
    forced = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=12, check_termination=1)
    monkeypatch.delenv("TINYMPC_REFILL", raising=False)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **forced)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(x0s)
    assert "slot-refill" not in s.jit_info()
    s.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    assert "slot-refill" in s.jit_info()           # default: on for a converging batch of this size
    s.update_settings(abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.solve()
    plain = _everything(s)
    s.reset()
    small = pkg.TinyMPC()
    small.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=8192, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=12)
    assert "slot-refill" not in small.jit_info()   # one resident set: nothing to refill with
    small.reset()
    s = _batch(pkg, prob, B, forced, x0s, True, monkeypatch)
    assert "slot-refill" in s.jit_info()
    s.solve()
    for a, b in zip(plain, _everything(s)):
        np.testing.assert_array_equal(a, b)
    assert np.all(plain[2] == 12) and np.all(plain[3] == 11)
    s.reset()


@pytest.mark.parametrize("B,ct,max_iter,tol", [(8193, 1, 30, 1e-3), (9100, 1, 1, 1e-3), (9100, 5, 3, 1e-3), (10240, 4, 9, 1e-2), (8200, 1, 25, 1e3)])
def test_slot_refill_edges(pkg, monkeypatch, B, ct, max_iter, tol):
    """One instance more than a resident set; max_iter 1; a check_termination that never comes up before max_iter (nobody is
    ever tested: every instance runs max_iter and reports no residuals); a check every fourth iteration with max_iter not a
    multiple of it; tolerances everything meets at the first check. Always: what the plain kernel returns, bit for bit."""
    P = pkg.problems
    prob = P.quadrotor(50)
    rng = np.random.default_rng(B)
    x0s = np.asfortranarray(P.quadrotor_batch_x0(B) * rng.uniform(0.05, 3.0, B)[None, :])
    settings = dict(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=max_iter, check_termination=ct)
    got = {}
    for refill in (False, True):
        s = _batch(pkg, prob, B, settings, x0s, refill, monkeypatch)
        assert ("slot-refill" in s.jit_info()) == refill
        s.solve()
        first = _everything(s)
        s.solve()  # warm, same x0
        got[refill] = (first, _everything(s))
        s.reset()
    for k in range(2):
        for a, b, what in zip(got[False][k], got[True][k], ("states", "controls", "iterations", "status", "residuals")):
            np.testing.assert_array_equal(a, b, err_msg=f"solve {k}: {what}")
    if ct > max_iter:
        assert np.all(got[True][0][2] == max_iter) and np.all(got[True][0][3] != 1)
    if tol >= 1e3:
        assert np.all(got[True][0][2] == ct) and np.all(got[True][0][3] == 1)
