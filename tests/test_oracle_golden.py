"""CPU tests: the plain-C restatement (oracle/tinympc_oracle.c) against the golden fixtures that
were generated from the reference's own compiled core (tests/golden/gen_golden.py), plus the
known-answer values recorded in SURVEY.md section 8(c) / BASELINE.md section 2."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import golden, problem_from_golden, rel_err, settings_from_golden

import pyoracle as O

SINGLE = ["cartpole_unconstrained", "cartpole_box_tol", "cartpole_box_200", "quadrotor_box_200", "quadrotor_box_tol"]
TRACE_ARRAYS = ("x", "u", "znew", "vnew", "y", "g", "r", "q", "p", "d")


@pytest.fixture(scope="module", autouse=True)
def _need_port():
    if not O.port_available():
        pytest.fail("oracle/liboracle_port.so missing: run `python __graft_entry__.py` (build)")


@pytest.mark.parametrize("name", SINGLE)
def test_cache_matches_reference(pkg, name):
    g = golden(name)
    orc = O.OraclePort(problem_from_golden(pkg, g))
    for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
        assert rel_err(orc.get(n), g[n]) < 1e-12, n


@pytest.mark.parametrize("name", SINGLE)
def test_solve_matches_reference(pkg, name):
    g = golden(name)
    prob = problem_from_golden(pkg, g)
    orc = O.OraclePort(prob).load_problem(prob, settings_from_golden(g))
    rc = orc.solve()
    st = orc.stats()
    assert rc == int(g["ret"])
    assert st["iter"] == int(g["iter"])
    assert st["status"] == int(g["status"])
    assert st["solved"] == int(g["solved"])
    sx, su = orc.solution()
    assert rel_err(sx, g["sol_x"]) < 1e-10
    assert rel_err(su, g["sol_u"]) < 1e-10
    res = np.array([st["pri_x"], st["dua_x"], st["pri_u"], st["dua_u"]])
    np.testing.assert_allclose(res, g["residuals"], rtol=1e-6, atol=1e-13)
    for n in ("d", "y", "g", "v", "z"):
        assert rel_err(orc.get(n), g["post_" + n], eps=1e-12) < 1e-9, n


@pytest.mark.parametrize("name", ["cartpole_box_200", "quadrotor_box_200"])
def test_phase_trace_matches_reference(pkg, name):
    """First three ADMM iterations, phase by phase (admm.cpp:129-199 order)."""
    g = golden(name)
    prob = problem_from_golden(pkg, g)
    orc = O.OraclePort(prob).load_problem(prob, settings_from_golden(g))
    for it in range(1, 4):
        orc.forward_pass()
        orc.update_slack()
        orc.update_dual()
        orc.update_linear_cost()
        orc.set_iter(it)
        orc.termination_condition()
        orc.put("v", orc.get("vnew"))
        orc.put("z", orc.get("znew"))
        orc.backward_pass_grad()
        for n in TRACE_ARRAYS:
            assert rel_err(orc.get(n), g[f"it{it}_{n}"], eps=1e-14) < 1e-11, (it, n)
        st = orc.stats()
        res = np.array([st["pri_x"], st["dua_x"], st["pri_u"], st["dua_u"]])
        np.testing.assert_allclose(res, g[f"it{it}_res"], rtol=1e-9, atol=1e-14)


def test_batch64_matches_reference(pkg):
    g = golden("quadrotor_batch64")
    prob = problem_from_golden(pkg, g)
    np.testing.assert_array_equal(pkg.problems.quadrotor_batch_x0(64), g["x0s"])
    orc = O.OraclePort(prob).load_problem(prob, settings_from_golden(g))
    sx, su, its, status, res = orc.solve_batch(g["x0s"])
    np.testing.assert_array_equal(its, g["iters"])
    assert rel_err(sx, g["sol_x"]) < 1e-10
    assert rel_err(su, g["sol_u"]) < 1e-10
    np.testing.assert_allclose(res, g["residuals"], rtol=1e-5, atol=1e-12)


def test_batch_x0_is_shard_invariant(pkg):
    P = pkg.problems
    full = P.quadrotor_batch_x0(96)
    np.testing.assert_array_equal(P.quadrotor_batch_x0(32, offset=64), full[:, 64:])


def test_mpc_loop_warm_start(pkg):
    """Closed loop: iteration counts per tick pin the warm-start semantics, including the
    one-iteration-stale v/z left by a converged solve (admm.cpp:181-197)."""
    g = golden("cartpole_mpc_loop")
    prob = problem_from_golden(pkg, g)
    orc = O.OraclePort(prob).load_problem(prob, settings_from_golden(g))
    x = g["x0"].copy()
    for k in range(int(g["ticks"])):
        orc.set_x0(x)
        orc.solve()
        st = orc.stats()
        assert st["iter"] == int(g["iters"][k]), k
        _, su = orc.solution()
        assert rel_err(su[:, 0], g["u0s"][:, k]) < 1e-9
        np.testing.assert_allclose([st["dua_x"], st["dua_u"]], g["dual_residuals"][:, k], rtol=1e-6, atol=1e-14)
        x = prob.A @ x + prob.B @ su[:, 0]
    assert rel_err(x, g["xs"][:, -1]) < 1e-9


def test_warm_restarts_of_a_mixed_batch_match_reference(pkg):
    """Three consecutive solves per instance (cold; warm, same x0; warm, 1.05 x0) of 16 instances that converge at different
    iterations: the restatement against the reference core's own numbers (tests/golden/gen_golden.py::warm_batch16)."""
    g = golden("quadrotor_warm_batch16")
    prob = problem_from_golden(pkg, g)
    settings = settings_from_golden(g)
    B = g["x0s"].shape[1]
    for b in range(B):
        orc = O.OraclePort(prob).load_problem(prob, settings)
        for k, x0 in enumerate((g["x0s"][:, b], g["x0s"][:, b], float(g["third_x0_scale"]) * g["x0s"][:, b])):
            orc.set_x0(x0)
            orc.solve()
            st = orc.stats()
            assert st["iter"] == g["iters"][k, b] and st["status"] == g["status"][k, b], (b, k)
            np.testing.assert_allclose([st["pri_x"], st["dua_x"], st["pri_u"], st["dua_u"]], g["residuals"][k, :, b], rtol=1e-8, atol=1e-14)
            sx, su = orc.solution()
            assert rel_err(su[:, 0], g["u0"][k, :, b]) < 1e-10
        assert rel_err(sx, g["sol_x"][:, :, b]) < 1e-10 and rel_err(su, g["sol_u"][:, :, b]) < 1e-10


def test_known_answers_from_survey(pkg):
    """values measured on the reference core during the survey."""
    P = pkg.problems
    cp = P.cartpole(20, True)
    orc = O.OraclePort(cp).load_problem(cp, {})
    np.testing.assert_allclose(orc.get("Kinf").ravel(),
                               [-1.8281816031, -2.4111848780, 20.6738188203, 3.3664150316], rtol=0, atol=5e-10)
    assert orc.stats()["riccati_iters"] == 454  # truncated fixed point, tiny_api.cpp:157
    orc.solve()
    assert orc.stats()["iter"] == 51
    assert abs(orc.solution()[1][0, 0] - 0.5) < 1e-12
    qd = P.quadrotor(50)
    orc = O.OraclePort(qd).load_problem(qd, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=200))
    orc.solve()
    np.testing.assert_allclose(orc.solution()[1][:, 0],
                               [-0.054688730, -0.456305153, -0.193144348, -0.500000000], rtol=0, atol=5e-9)


def test_double_rho_and_truncated_riccati(pkg):
    """Parity trap: the cache is the truncated fixed point of the Riccati map with
    Q+2*rho*I, R+2*rho*I -- NOT the DARE solution for Q+rho*I."""
    import scipy.linalg

    cp = pkg.problems.cartpole(20, True)
    orc = O.OraclePort(cp)
    K = orc.get("Kinf")
    Pd = scipy.linalg.solve_discrete_are(cp.A, cp.B, cp.Q + 2 * cp.rho * np.eye(4), cp.R + 2 * cp.rho * np.eye(1))
    Kd = np.linalg.solve(cp.R + 2 * cp.rho * np.eye(1) + cp.B.T @ Pd @ cp.B, cp.B.T @ Pd @ cp.A)
    assert 1e-6 < rel_err(K, Kd) < 1e-3  # close to the double-rho DARE, but measurably truncated
    P1 = scipy.linalg.solve_discrete_are(cp.A, cp.B, cp.Q + cp.rho * np.eye(4), cp.R + cp.rho * np.eye(1))
    K1 = np.linalg.solve(cp.R + cp.rho * np.eye(1) + cp.B.T @ P1 @ cp.B, cp.B.T @ P1 @ cp.A)
    assert rel_err(K, K1) > 1e-2  # single-rho DARE is a different controller


def test_settings_edge_cases(pkg):
    cp = pkg.problems.cartpole(20, True)
    orc = O.OraclePort(cp).load_problem(cp, dict(max_iter=0))
    assert orc.solve() == 1 and orc.stats()["iter"] == 0  # loop never runs: admm.cpp:129, 202-206
    orc = O.OraclePort(cp).load_problem(cp, dict(max_iter=60, check_termination=7))
    orc.solve()
    assert orc.stats()["iter"] % 7 == 0 and orc.stats()["iter"] >= 51  # admm.cpp:91
    orc = O.OraclePort(cp).load_problem(cp, dict(max_iter=30, check_termination=0))
    assert orc.solve() == 1 and orc.stats()["iter"] == 30  # modulo-by-zero in the reference: never check


def test_unpinned_extensions_are_first_principles_sane(pkg):
    """SOC / linear / fdyn have no reference source in tree (parity unpinned): check the restated
    projections by their defining properties instead."""
    rk = pkg.problems.rocket(40)
    orc = O.OraclePort(rk).load_problem(rk, dict(max_iter=400, abs_pri_tol=1e-3, abs_dua_tol=1e-3))
    orc.solve()
    zc = orc.get("zcnew")
    assert np.all(np.linalg.norm(zc[:2], axis=0) <= 0.25 * zc[2] + 1e-9)      # input cone feasibility
    vc = orc.get("vcnew")
    assert np.all(np.linalg.norm(vc[:2], axis=0) <= 0.5 * vc[2] + 1e-9)       # state cone feasibility
    vl = orc.get("vlnew")
    assert np.all(-vl[2] <= 1e-9)                                             # ground-plane half-space
    sx, su = orc.solution()
    assert np.all(su >= rk.u_min[:, None] - 1e-12) and np.all(su <= rk.u_max[:, None] + 1e-12)
    # rollout obeys affine dynamics
    x, u = orc.get("x"), orc.get("u")
    np.testing.assert_allclose(x[:, 1:], rk.A @ x[:, :-1] + rk.B @ u + rk.fdyn[:, None], atol=1e-9)
