"""Large systems, 64 < nx+nu <= 512 (the reference takes any nx, nu: types.hpp:16-17): the step of sixteen instances is a
GEMM on the FP64 matrix cores (tinympc_solve_m.hip, v_mfma_f64_16x16x4_f64), the ADMM state streams through HBM in the
tile's own layout; beyond 128 rows a wavefront owns two to four row tiles and streams its operator tiles from L2. Against the oracle: caches, per-instance termination inside a tile, ragged last tile, warm starts after
converged and unconverged solves, bounds and references that vary over the horizon."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _system(pkg, nx, nu, N, seed, varying):
    P = pkg.problems
    rng = np.random.default_rng(seed)
    A = np.eye(nx) * 0.98 + 0.015 * rng.standard_normal((nx, nx))
    B = 0.08 * rng.standard_normal((nx, nu))
    prob = P.Problem("large", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    if varying:
        prob.x_min = -2.0 - rng.uniform(0, 0.5, (nx, N))
        prob.x_max = 2.0 + rng.uniform(0, 0.5, (nx, N))
        prob.u_min = -0.3 * rng.uniform(0.7, 1.0, (nu, N - 1))
        prob.u_max = 0.3 * rng.uniform(0.7, 1.0, (nu, N - 1))
        prob.x_ref = 0.05 * rng.standard_normal((nx, N))
        prob.u_ref = 0.02 * rng.standard_normal((nu, N - 1))
    else:
        prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
        prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    return prob


def _solver(pkg, prob, settings, batch):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref)
        s.set_u_ref(prob.u_ref)
    return s


@pytest.mark.parametrize("nx,nu,N,varying", [(60, 10, 8, False), (70, 14, 10, True), (90, 20, 6, False), (96, 32, 12, True)])
def test_large_systems_match_the_oracle(pkg, nx, nu, N, varying):
    prob = _system(pkg, nx, nu, N, nx + nu, varying)
    batch = 21  # two tiles of 16, the second one ragged
    rng = np.random.default_rng(3)
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.02, 1.2, batch)[None, :]
    for settings in (dict(max_iter=120, abs_pri_tol=1e-3, abs_dua_tol=1e-3), dict(max_iter=25, abs_pri_tol=0.0, abs_dua_tol=0.0),
                     dict(max_iter=60, abs_pri_tol=1e-4, abs_dua_tol=1e-4, check_termination=4)):
        s = _solver(pkg, prob, settings, batch)
        assert s.launch_info()["layout"] == "M"
        c = s.get_cache()
        orc = [O.OraclePort(prob).load_problem(prob, settings) for _ in range(batch)]
        for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
            assert rel_err(c[n], orc[0].get(n)) < 1e-9, n
        for rnd in range(3):  # cold start, then two warm starts
            xs = x0s * (1.0 - 0.35 * rnd)
            s.set_x0_batch(xs)
            s.solve()
            sol, st = s.get_solution_batch(), s.get_stats_batch()
            for b in range(batch):
                orc[b].set_x0(xs[:, b])
                orc[b].solve()
                assert st["iter"][b] == orc[b].stats()["iter"], (rnd, b)
                assert st["status"][b] == orc[b].stats()["status"], (rnd, b)
                assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL, (rnd, b)
                assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, (rnd, b)
            for b in (0, batch // 2, batch - 1):  # the four inf-norms of the last check
                ob = orc[b].stats()
                np.testing.assert_allclose(st["residuals"][:, b], [ob["pri_x"], ob["dua_x"], ob["pri_u"], ob["dua_u"]], rtol=1e-6, atol=1e-10)
        s.reset()


@pytest.mark.parametrize("nx,nu,N,varying", [(130, 14, 6, False), (160, 32, 8, True), (224, 32, 5, False), (300, 20, 5, True), (480, 32, 4, False)])
def test_systems_beyond_128_rows(pkg, nx, nu, N, varying):
    """R = 9, 12, 16, 20, 32 row tiles: two to four per wavefront, operator tiles streamed (the nx=160, nu=32 case). One oracle per
    checked instance (first tile, tile edge, ragged last tile); cold start and two warm starts."""
    prob = _system(pkg, nx, nu, N, nx + nu, varying)
    rng = np.random.default_rng(7)
    prob.A = 0.95 * np.eye(nx) + (0.15 / np.sqrt(nx)) * rng.standard_normal((nx, nx))  # (spectral radius ~ 1: see the edge-width test below)
    if nx >= 256:  # (a faster-contracting system: the oracle's Riccati iteration at 0.95 takes minutes at nx = 480)
        prob.A = 0.6 * np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    batch = 19
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.02, 1.2, batch)[None, :]
    checked = (0, 15, 16, batch - 1)
    for settings in (dict(max_iter=150, abs_pri_tol=1e-3, abs_dua_tol=1e-3), dict(max_iter=20, abs_pri_tol=0.0, abs_dua_tol=0.0, check_termination=3)):
        s = _solver(pkg, prob, settings, batch)
        assert s.launch_info()["layout"] == "M"
        c = s.get_cache()
        orc = {b: O.OraclePort(prob).load_problem(prob, settings) for b in checked}
        for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
            assert rel_err(c[n], orc[0].get(n)) < 1e-9, n
        for rnd in range(3):
            xs = x0s * (1.0 - 0.35 * rnd)
            s.set_x0_batch(xs)
            s.solve()
            sol, st = s.get_solution_batch(), s.get_stats_batch()
            for b in checked:
                orc[b].set_x0(xs[:, b])
                orc[b].solve()
                ob = orc[b].stats()
                assert st["iter"][b] == ob["iter"] and st["status"][b] == ob["status"], (rnd, b)
                assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL, (rnd, b)
                assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, (rnd, b)
                np.testing.assert_allclose(st["residuals"][:, b], [ob["pri_x"], ob["dua_x"], ob["pri_u"], ob["dua_u"]], rtol=1e-6, atol=1e-10)
        s.reset()
    with pytest.raises(pkg.TinyMPCError) as ei:  # beyond 512 rows: refused, and says so
        pkg.TinyMPC().setup(np.eye(510), np.ones((510, 8)), np.eye(510), np.eye(8), 5, batch=1)
    assert ei.value.code == pkg._lib.ERR_UNSUPPORTED


@pytest.mark.parametrize("nx,nu,N,batch,form", [(96, 32, 12, 21, "one-pass"), (70, 10, 8, 5, "one-pass"), (160, 32, 8, 19, "one-pass"), (300, 20, 5, 17, "one-pass"),
                                                 (96, 32, 12, 21, "general"), (160, 32, 8, 19, "general"), (70, 10, 8, 5, "ten rows"), (300, 20, 5, 17, "ten rows")])
def test_families_on_large_systems(pkg, monkeypatch, nx, nu, N, batch, form):
    """Cones and linear rows beyond 64 rows (bindings.cpp:408-478 take any nx, nu): layout M's families phase (wavefront w evaluates
    knots w, w + NW, ... between the sweeps, tinympc_solve_m.hip). A cone inside one 16-row tile, one across a tile boundary, two
    that share rows (projected one after another), one over three tiles, an input cone that starts in the tile the state rows end
    in; three dense state rows, two input rows; fdyn. R = 8, 5, 12 (streamed operators) and 20 (two row tiles per wavefront).
    Against the restatement: iteration counts, 1e-9 on the trajectories, cold start + two warm starts; then the families switched
    off again (the box path of the same handle).
    Two forms of the phase: up to eight linear rows per side ONE pass computes every row's dot product and the sequence of projections is
    a scalar recurrence over the rows' Gram matrix ("one-pass"); beyond that, or with TINYMPC_M_FAM_FAST=0, a pass per row ("general",
    "ten rows")."""
    if form == "general":
        monkeypatch.setenv("TINYMPC_M_FAM_FAST", "0")
    P = pkg.problems
    rng = np.random.default_rng(nx * 10 + nu)
    A = 0.9 * np.eye(nx) + (0.15 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    if nx >= 256:
        A = 0.6 * np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    Bm = 0.3 * rng.standard_normal((nx, nu))
    prob = P.Problem("largefam", A, Bm, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 1.5, rng.standard_normal(nx))
    prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    prob.u_min, prob.u_max = np.full(nu, -1.0), np.full(nu, 1.0)
    prob.fdyn = 0.01 * rng.standard_normal(nx)
    prob.cones = dict(Acx=[0, 13, 15, 20], qcx=[3, 6, 3, 41], cx=[0.8, 0.6, 1.1, 0.9], Acu=[0, 5], qcu=[3, 4], cu=[0.7, 1.3])
    nlx = 10 if form == "ten rows" else 3
    prob.linear = dict(Alin_x=rng.standard_normal((nlx, nx)) / np.sqrt(nx), blin_x=rng.uniform(0.1, 0.4, nlx),
                       Alin_u=rng.standard_normal((2, nu)) / np.sqrt(nu), blin_u=rng.uniform(0.1, 0.3, 2))
    settings = dict(max_iter=60, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_cone_constraints(**prob.cones)
    s.set_linear_constraints(**prob.linear)
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.1, 1.0, batch)[None, :]
    checked = sorted({0, batch // 2, 15 if batch > 16 else 1, batch - 1})
    orc = {b: O.OraclePort(prob).load_problem(prob, settings) for b in checked}
    moved = 0.0
    for rnd in range(3):
        xs = x0s * (1.0 - 0.3 * rnd)
        s.set_x0_batch(xs)
        s.solve()
        assert s.launch_info()["layout"] == "M"
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in checked:
            orc[b].set_x0(xs[:, b])
            orc[b].solve()
            ob = orc[b].stats()
            assert st["iter"][b] == ob["iter"] and st["status"][b] == ob["status"], (rnd, b, st["iter"][b], ob["iter"])
            assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL, (rnd, b)
            assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, (rnd, b)
            np.testing.assert_allclose(st["residuals"][:, b], [ob["pri_x"], ob["dua_x"], ob["pri_u"], ob["dua_u"]], rtol=1e-6, atol=1e-10)
        if rnd == 0:  # the families do something here: the box-only problem has another answer
            box = pkg.TinyMPC()
            box.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
            box.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            box.set_x0_batch(xs)
            box.solve()
            moved = float(np.abs(box.get_solution_batch()["controls"] - sol["controls"]).max())
            box.reset()
    assert moved > 1e-3, moved
    # families off again: the same handle on the box path, from its warm state, against an oracle brought to the same state
    s.update_settings(en_state_soc=0, en_input_soc=0, en_state_linear=0, en_input_linear=0)
    for b in checked:
        orc[b].update_settings(en_state_soc=0, en_input_soc=0, en_state_linear=0, en_input_linear=0)
    s.solve()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    for b in checked:
        orc[b].solve()
        assert st["iter"][b] == orc[b].stats()["iter"], b
        assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, b
    s.reset()


def test_large_system_single_instance_and_unsupported_features(pkg):
    prob = _system(pkg, 66, 6, 6, 1, False)
    settings = dict(max_iter=80, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    s = _solver(pkg, prob, settings, 1)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    x = prob.x0.copy()
    for k in range(4):  # a short closed loop on the single-instance verbs
        s.set_x0(x)
        s.solve()
        orc.set_x0(x)
        orc.solve()
        assert s.get_stats()["iter"] == orc.stats()["iter"]
        assert rel_err(s.get_solution()["controls"], orc.solution()[1]) < TOL
        x = prob.A @ x + prob.B @ s.get_solution()["controls"][:, 0]
    with pytest.raises(pkg.TinyMPCError):
        s.session_begin()
    s.update_settings(adaptive_rho=1)
    with pytest.raises(pkg.TinyMPCError) as ei:
        s.solve()
    assert ei.value.code == pkg._lib.ERR_UNSUPPORTED
    s.reset()


def test_large_system_lifecycle_edges(pkg):
    """Cold start after reset_workspace reproduces the first solve; a 0-iteration solve leaves everything alone;
    check_termination = 0 never checks; the batched tick (mpc_step) equals set_x0_batch + solve + first controls."""
    prob = _system(pkg, 72, 8, 7, 5, False)
    settings = dict(max_iter=30, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    batch = 18
    rng = np.random.default_rng(9)
    x0s = rng.standard_normal((prob.nx, batch)) * 0.3
    s = _solver(pkg, prob, settings, batch)
    s.set_x0_batch(x0s)
    s.solve()
    first = s.get_solution_batch()
    it1 = s.get_stats_batch()["iter"].copy()
    s.reset_workspace()
    s.solve()
    np.testing.assert_array_equal(s.get_solution_batch()["controls"], first["controls"])
    np.testing.assert_array_equal(s.get_stats_batch()["iter"], it1)
    s.update_settings(max_iter=0)
    s.solve()
    np.testing.assert_array_equal(s.get_solution_batch()["controls"], first["controls"])
    s.update_settings(max_iter=12, check_termination=0)
    s.solve()
    st = s.get_stats_batch()
    assert np.all(st["iter"] == 12) and np.all(st["status"] == 11)
    # tick: one call vs three verbs, warm-started from the same state on two handles
    a, b = _solver(pkg, prob, settings, batch), _solver(pkg, prob, settings, batch)
    x = x0s.copy()
    for k in range(3):
        ua = a.mpc_step(x)
        b.set_x0_batch(x)
        b.solve()
        np.testing.assert_array_equal(ua, b.get_first_controls_batch())
        x = prob.A @ x + prob.B @ ua
    for h in (s, a, b):
        h.reset()


@pytest.mark.parametrize("nx,nu,N", [(17, 2, 10), (15, 2, 10), (33, 2, 6), (62, 2, 5)])
def test_edge_widths_on_the_run_time_specialised_layout_d(pkg, monkeypatch, nx, nu, N):
    """Just above 16 and 32 rows and at 64: the 32- and 64-lane forms of layout D with almost empty upper DPP rows.
    (A stable A and nu = 2: a weakly controllable unstable system makes the Riccati recursion itself ill-conditioned, and
    its round-off then moves the solution by more than the tolerance in every layout alike.)"""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    prob = _system(pkg, nx, nu, N, 11, False)
    rng = np.random.default_rng(5)
    prob.A = 0.85 * np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))  # stable: two inputs cannot hold 60 unstable modes
    settings = dict(max_iter=60, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    batch = 70
    rng = np.random.default_rng(2)
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.05, 1.0, batch)[None, :]
    s = _solver(pkg, prob, settings, batch)
    assert s.launch_info()["layout"] == "D"
    s.set_x0_batch(x0s)
    s.solve()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, ost, _ = orc.solve_batch(x0s)
    np.testing.assert_array_equal(st["iter"], oit)
    np.testing.assert_array_equal(st["status"], ost)
    assert rel_err(sol["states"], ox) < TOL and rel_err(sol["controls"], ou) < TOL
    s.reset()
