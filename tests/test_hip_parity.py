"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(a) the golden fixtures generated from the reference's compiled core and (b) the plain-C oracle on
seeded inputs. Tolerance: 1e-6 relative on states/controls is the bar BASELINE.json states; the
kernels reassociate the two mat-vecs of a sweep step into one, so agreement is ~1e-12, and the tests
assert a tighter 1e-9 to catch regressions early. Iteration counts must match exactly."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import golden, problem_from_golden, rel_err, settings_from_golden

import pyoracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-9        # asserted
TOL_RANDOM = 1e-7  # asserted for the randomized (ill-conditioned) family
TOL_BAR = 1e-6    # the north-star bar (BASELINE.json): TOL must stay below it
assert TOL < TOL_BAR

SINGLE = ["cartpole_unconstrained", "cartpole_box_tol", "cartpole_box_200", "quadrotor_box_200", "quadrotor_box_tol"]


LAYOUTS = ("A", "B", "C", "D")  # (conftest.pytest_generate_tests: the default of this module; @pytest.mark.layouts(...) narrows it per test)


@pytest.fixture(autouse=True)
def kernel_layout(request, monkeypatch):
    """Every test runs against all four solve kernels: layout A (all ADMM state in LDS, tinympc_solve.hip),
    layout B (V L2-resident in HBM, 4-wave workgroups, tinympc_solve_b.hip), layout C (one instance per
    workgroup, horizon swept in 16 concurrent chunks, tinympc_solve_c.hip) and layout D (horizon unrolled at compile
    time, state in registers, two waves per SIMD, tinympc_solve_d.hip). The layout is chosen at setup time;
    where B does not apply (W > 16 or N < 8), C does not (W > 16 or N > 129) or D does not (shape not compiled in,
    time-varying bounds / references) the library falls back. Tests that apply to some of the kernels only say so with
    `@pytest.mark.layouts(...)`: the other combinations are never generated (no skips to wade through)."""
    monkeypatch.setenv("TINYMPC_LAYOUT", request.param)
    return request.param


def make_solver(pkg, prob, settings, batch=1):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho,
            fdyn=prob.fdyn, **{k: v for k, v in settings.items() if k in ("abs_pri_tol", "abs_dua_tol", "max_iter", "check_termination")})
    if prob.has_bounds():
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref)
    if prob.u_ref is not None:
        s.set_u_ref(prob.u_ref)
    extra = {k: v for k, v in settings.items() if k in ("en_state_bound", "en_input_bound")}
    if extra:
        s.update_settings(**extra)
    s.set_x0(prob.x0)
    return s


def test_native_library_is_loaded(pkg):
    lib = pkg.load_library()
    assert lib._name.endswith("tinympc-matlab_amd/libtinympc_hip.so")
    assert pkg.device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libtinympc_hip.so" in f.read()


def test_layout_selection(pkg, kernel_layout, monkeypatch):
    P = pkg.problems
    s = make_solver(pkg, P.quadrotor(50), {})
    assert s.launch_info()["layout"] == kernel_layout  # the env override is honoured where B applies
    s.reset()
    monkeypatch.delenv("TINYMPC_LAYOUT")
    s = make_solver(pkg, P.quadrotor(50), {}, batch=8)
    info = s.launch_info()
    assert info["layout"] == "C" and info["workgroups"] == 8  # default for small batches: the latency kernel
    s.reset()
    s = make_solver(pkg, P.quadrotor(50), {}, batch=2048)
    info = s.launch_info()
    assert info["layout"] == "D" and info["workgroups"] == 128  # large batch of a compiled-in shape: workgroups of four wavefronts
    s.reset()
    s = make_solver(pkg, P.quadrotor(40), {}, batch=2048)
    info = s.launch_info()
    assert info["layout"] == "D" and info["workgroups"] == 128  # other shapes that fit the plan: layout D specialised at run time
    s.reset()
    monkeypatch.setenv("TINYMPC_JIT", "0")
    s = make_solver(pkg, P.quadrotor(40), {}, batch=2048)
    info = s.launch_info()
    assert info["layout"] == "B" and info["workgroups"] == 128 and info["lds_bytes"] <= 160 * 1024  # without it: B where it fits
    s.reset()
    monkeypatch.delenv("TINYMPC_JIT")
    s = make_solver(pkg, P.cartpole(120, True), {}, batch=2048)
    info = s.launch_info()
    assert info["layout"] == "D" and info["workgroups"] == 128  # N = 120: the plan with one wavefront per SIMD (four per workgroup)
    s.reset()
    s = make_solver(pkg, P.cartpole(250, True), {}, batch=2048)
    assert s.launch_info()["layout"] in ("A", "B")  # N = 250 fits neither register plan of layout D ...
    s.prepare()
    assert s.launch_info()["layout"] == "E" and s.launch_info()["workgroups"] == 512  # ... layout E cuts the horizon across a workgroup's wavefronts
    s.reset()
    monkeypatch.setenv("TINYMPC_LAYOUT", "B")
    s = make_solver(pkg, P.cartpole(5, True), {})
    assert s.launch_info()["layout"] == "A"  # N < 8: layout B's 4-deep prefetch ring does not apply
    s.reset()
    monkeypatch.setenv("TINYMPC_LAYOUT", "C")
    s = make_solver(pkg, P.cartpole(200, True), {})
    assert s.launch_info()["layout"] in ("A", "B")  # N > 129: more than 8 steps per chunk, layout C does not apply
    s.reset()


@pytest.mark.parametrize("name", SINGLE)
def test_precompute_kernel_matches_reference_cache(pkg, name):
    """P1: device Riccati fixed point vs the reference's cache, incl. the truncated step count."""
    g = golden(name)
    prob = problem_from_golden(pkg, g)
    s = make_solver(pkg, prob, settings_from_golden(g))
    c = s.get_cache()
    for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
        assert rel_err(c[n], g[n]) < 1e-11, n
    orc = O.OraclePort(prob)
    assert c["riccati_iters"] == orc.stats()["riccati_iters"]
    s.reset()


@pytest.mark.layouts("C")  # (the precompute does not depend on the solve kernel: once, not once per layout)
@pytest.mark.parametrize("nx,nu", [(3, 1), (4, 3), (5, 2), (7, 3), (8, 4), (10, 4), (12, 1), (9, 2), (12, 4), (6, 3), (13, 2), (6, 5)])
def test_register_resident_precompute_on_every_shape_class(pkg, monkeypatch, nx, nu):
    """k_precompute_rows (round 5: the Riccati fixed point in the registers of one wavefront) is instantiated for the BASELINE shapes exactly
    and for zero-padded classes (4,4), (8,4), (12,4); nx > 12 or nu > 4 stay on the one-workgroup LDS kernel. Every class against the
    reference's own core (box problems: oracle/_ref where it is there) incl. the truncated step count, with fdyn against the restatement, and
    against the LDS kernel on the same handle data (TINYMPC_PRECOMPUTE=lds)."""
    P = pkg.problems
    rng = np.random.default_rng(100 * nx + nu)
    A = 0.92 * np.eye(nx) + (0.25 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.3 * rng.standard_normal((nx, nu))
    Q, R = np.diag(rng.uniform(0.5, 20.0, nx)), np.diag(rng.uniform(0.2, 4.0, nu))
    for fdyn in (None, 0.05 * rng.standard_normal(nx)):
        prob = P.Problem("rows", A, B, Q, R, 8, float(rng.uniform(0.3, 6.0)), rng.standard_normal(nx))
        prob.fdyn = fdyn
        caches = {}
        for mode in ("rows", "lds"):
            if mode == "lds":
                monkeypatch.setenv("TINYMPC_PRECOMPUTE", "lds")
            else:
                monkeypatch.delenv("TINYMPC_PRECOMPUTE", raising=False)
            s = pkg.TinyMPC()
            s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, fdyn=prob.fdyn)
            caches[mode] = s.get_cache()
            # (what the affine terms APf / BPf feed: one solve from a non-zero state)
            s.set_x0(prob.x0)
            s.update_settings(max_iter=12, abs_pri_tol=0.0, abs_dua_tol=0.0)
            s.solve()
            caches[mode]["sol"] = s.get_solution()["controls"]
            s.reset()
        monkeypatch.delenv("TINYMPC_PRECOMPUTE", raising=False)
        orc = (O.OracleRef if (fdyn is None and O.ref_available()) else O.OraclePort)(prob)
        steps = O.OraclePort(prob).stats()["riccati_iters"]  # (the restatement counts its Riccati steps; pinned to the reference's cache below)
        for n, key in (("Kinf", "Kinf"), ("Pinf", "Pinf"), ("Quu_inv", "Quu_inv"), ("AmBKt", "AmBKt")):
            ref = orc.get(key)
            assert rel_err(caches["rows"][n], ref.reshape(caches["rows"][n].shape, order="F")) < 1e-10, (n, nx, nu)
            assert rel_err(caches["rows"][n], caches["lds"][n]) < 1e-11, (n, "rows vs lds")
        assert caches["rows"]["riccati_iters"] == caches["lds"]["riccati_iters"] == steps
        assert rel_err(caches["rows"]["sol"], caches["lds"]["sol"]) < 1e-9


@pytest.mark.parametrize("name", SINGLE)
def test_solve_matches_golden(pkg, name):
    g = golden(name)
    prob = problem_from_golden(pkg, g)
    s = make_solver(pkg, prob, settings_from_golden(g))
    assert s.solve() == 0
    st = s.get_stats()
    sol = s.get_solution()
    assert st["iter"] == int(g["iter"])
    assert st["status"] == int(g["status"])
    assert rel_err(sol["states"], g["sol_x"]) < TOL
    assert rel_err(sol["controls"], g["sol_u"]) < TOL
    res = np.array([st["primal_residual_state"], st["dual_residual_state"], st["primal_residual_input"], st["dual_residual_input"]])
    np.testing.assert_allclose(res, g["residuals"], rtol=1e-5, atol=1e-12)
    s.reset()


def test_get_solution_before_solve_is_zero(pkg):
    prob = pkg.problems.cartpole()
    s = make_solver(pkg, prob, {})
    sol = s.get_solution()
    assert not sol["states"].any() and not sol["controls"].any()  # tiny_setup zero-fills the solution (tiny_api.cpp:43-44)
    st = s.get_stats()
    assert st["iter"] == 0
    s.reset()


def test_batch64_matches_golden(pkg):
    g = golden("quadrotor_batch64")
    prob = problem_from_golden(pkg, g)
    s = make_solver(pkg, prob, settings_from_golden(g), batch=64)
    s.set_x0_batch(g["x0s"])
    s.solve()
    sol = s.get_solution_batch()
    st = s.get_stats_batch()
    np.testing.assert_array_equal(st["iter"], g["iters"])
    assert rel_err(sol["states"], g["sol_x"]) < TOL
    assert rel_err(sol["controls"], g["sol_u"]) < TOL
    np.testing.assert_allclose(st["residuals"], g["residuals"], rtol=1e-4, atol=1e-11)
    np.testing.assert_array_equal(s.get_first_controls_batch(), sol["controls"][:, 0, :])
    s.reset()


@pytest.mark.parametrize("batch", [1, 3, 4, 5, 67])
def test_ragged_batches_and_per_instance_termination(pkg, batch):
    """Instances of one wavefront converge at different iterations; batch sizes that do not fill a
    wavefront (4 instances) leave idle lane groups."""
    P = pkg.problems
    prob = P.quadrotor(30)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=300, check_termination=1)
    x0s = P.quadrotor_batch_x0(batch) * np.linspace(0.05, 1.5, batch)[None, :]
    s = make_solver(pkg, prob, settings, batch=batch)
    s.set_x0_batch(x0s)
    s.solve()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, ost, ores = orc.solve_batch(x0s)
    np.testing.assert_array_equal(st["iter"], oit)
    np.testing.assert_array_equal(st["status"], ost)
    assert len(set(oit.tolist())) > 1 or batch == 1  # the case really is heterogeneous
    assert rel_err(sol["states"], ox) < TOL
    assert rel_err(sol["controls"], ou) < TOL
    s.reset()


def test_warm_start_mpc_loop_matches_golden(pkg):
    """Closed-loop ticks: the persistent device state (d, y, g, v, z) must follow the reference's
    warm-start semantics, including the stale v/z after a converged solve (admm.cpp:181-197)."""
    g = golden("cartpole_mpc_loop")
    prob = problem_from_golden(pkg, g)
    s = make_solver(pkg, prob, settings_from_golden(g))
    x = g["x0"].copy()
    for k in range(int(g["ticks"])):
        s.set_x0(x)
        s.solve()
        st = s.get_stats()
        assert st["iter"] == int(g["iters"][k]), k
        u0 = s.get_solution()["controls"][:, 0]
        assert rel_err(u0, g["u0s"][:, k]) < TOL
        np.testing.assert_allclose([st["dual_residual_state"], st["dual_residual_input"]], g["dual_residuals"][:, k],
                                   rtol=1e-6, atol=1e-13)
        x = prob.A @ x + prob.B @ u0
    s.reset()


def test_warm_start_after_unconverged_solve(pkg):
    prob = pkg.problems.quadrotor(20)
    settings = dict(abs_pri_tol=1e-5, abs_dua_tol=1e-5, max_iter=7, check_termination=1)
    s = make_solver(pkg, prob, settings)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    for _ in range(4):  # 4 x 7 iterations, state carried over
        s.solve()
        orc.solve()
        assert s.get_stats()["iter"] == orc.stats()["iter"]
        assert rel_err(s.get_solution()["controls"], orc.solution()[1]) < TOL
    s.reset()


@pytest.mark.parametrize("ct,max_iter", [(7, 60), (0, 30), (1, 0), (3, 2)])
def test_check_termination_and_max_iter_edges(pkg, ct, max_iter):
    prob = pkg.problems.cartpole(20, True)
    settings = dict(max_iter=max_iter, check_termination=ct)
    s = make_solver(pkg, prob, settings)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    s.solve()
    orc.solve()
    st, ost = s.get_stats(), orc.stats()
    assert st["iter"] == ost["iter"] and st["status"] == ost["status"]
    assert rel_err(s.get_solution()["states"], orc.solution()[0], eps=1e-30) < TOL
    s.reset()


def test_references_and_time_varying_bounds(pkg):
    """Non-zero Xref/Uref (terminal Pinf term, admm.cpp:81) and bounds that differ per knot."""
    P = pkg.problems
    rng = np.random.default_rng(3)
    prob = P.quadrotor(25)
    nx, nu, N = prob.nx, prob.nu, prob.N
    prob.x_ref = 0.2 * rng.standard_normal((nx, N))
    prob.u_ref = 0.05 * rng.standard_normal((nu, N - 1))
    prob.x_min = -3.0 - rng.uniform(0, 1, (nx, N))
    prob.x_max = 3.0 + rng.uniform(0, 1, (nx, N))
    prob.u_min = -0.3 - rng.uniform(0, 0.2, (nu, N - 1))
    prob.u_max = 0.3 + rng.uniform(0, 0.2, (nu, N - 1))
    settings = dict(abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=150)
    s = make_solver(pkg, prob, settings)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    s.solve()
    orc.solve()
    assert s.get_stats()["iter"] == orc.stats()["iter"]
    assert rel_err(s.get_solution()["states"], orc.solution()[0]) < TOL
    assert rel_err(s.get_solution()["controls"], orc.solution()[1]) < TOL
    s.reset()


def test_bound_flags_off_means_unclamped(pkg):
    """en_*_bound = 0 with bounds installed: the clamp must be skipped (admm.cpp:49, 55)."""
    prob = pkg.problems.cartpole(20, True)
    settings = dict(max_iter=40, en_state_bound=0, en_input_bound=0)
    s = make_solver(pkg, prob, settings)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    s.solve()
    orc.solve()
    assert np.max(np.abs(orc.solution()[1])) > 0.5  # really unclamped
    assert s.get_stats()["iter"] == orc.stats()["iter"]
    assert rel_err(s.get_solution()["controls"], orc.solution()[1]) < TOL
    s.reset()


def test_set_cache_terms_is_used_by_the_sweeps(pkg):
    prob = pkg.problems.cartpole(20, True)
    settings = dict(max_iter=40)
    s = make_solver(pkg, prob, settings)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    c = s.get_cache()
    K, Pm, Qi, Am = c["Kinf"] * 1.01, c["Pinf"] * 0.99, c["Quu_inv"] * 1.02, c["AmBKt"] * 0.995
    s.set_cache_terms(K, Pm, Qi, Am)
    orc.set_cache_terms(K, Pm, Qi, Am)
    s.solve()
    orc.solve()
    assert s.get_stats()["iter"] == orc.stats()["iter"]
    assert rel_err(s.get_solution()["controls"], orc.solution()[1]) < TOL
    s.reset()


@pytest.mark.parametrize("nx,nu,N", [(2, 1, 5), (7, 2, 12), (9, 3, 40), (13, 3, 30), (20, 6, 15), (30, 10, 9), (40, 12, 8)])
def test_random_problem_shapes_cover_all_lane_widths(pkg, nx, nu, N):
    """W = 16 (KT 8/12/16), W = 32 and W = 64 instantiations, several instances each."""
    P = pkg.problems
    rng = np.random.default_rng(nx * 100 + nu)
    A = np.eye(nx) + 0.05 * rng.standard_normal((nx, nx))
    B = 0.1 * rng.standard_normal((nx, nu))
    prob = P.Problem("rand", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N,
                     float(rng.uniform(0.5, 3)), rng.standard_normal(nx))
    prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
    prob.x_ref = 0.1 * rng.standard_normal((nx, N))
    prob.u_ref = 0.05 * rng.standard_normal((nu, N - 1))
    settings = dict(max_iter=60, abs_pri_tol=1e-5, abs_dua_tol=1e-5)
    batch = 5
    x0s = rng.standard_normal((nx, batch))
    s = make_solver(pkg, prob, settings, batch=batch)
    c = s.get_cache()
    orc = O.OraclePort(prob).load_problem(prob, settings)
    for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
        assert rel_err(c[n], orc.get(n)) < 1e-10, n
    s.set_x0_batch(x0s)
    s.solve()
    ox, ou, oit, ost, _ = orc.solve_batch(x0s)
    st = s.get_stats_batch()
    np.testing.assert_array_equal(st["iter"], oit)
    np.testing.assert_array_equal(st["status"], ost)
    sol = s.get_solution_batch()
    assert rel_err(sol["states"], ox) < TOL
    assert rel_err(sol["controls"], ou) < TOL
    s.reset()


def test_fdyn_affine_dynamics(pkg):
    """fdyn path (PARITY UNPINNED upstream semantics): HIP vs the restated oracle, plus the defining
    property that the converged rollout obeys x+ = A x + B u + f."""
    P = pkg.problems
    rk = P.rocket(30, with_linear=False)
    rk.cones = {}
    settings = dict(max_iter=200, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
    s = make_solver(pkg, rk, settings)
    orc = O.OraclePort(rk).load_problem(rk, settings)
    s.solve()
    orc.solve()
    assert s.get_stats()["iter"] == orc.stats()["iter"]
    assert rel_err(s.get_solution()["states"], orc.solution()[0]) < TOL
    assert rel_err(s.get_solution()["controls"], orc.solution()[1]) < TOL
    s.reset()


def test_full_size_batch_properties(pkg):
    """BASELINE size (8,192 instances = one GPU's shard of config 5): size-independent properties.
    (1) every instance equals the single-instance solve of the same x0 (spot-checked against the
    oracle on a seeded sample), (2) permutation equivariance: reversing the instance order reverses
    the solutions bit for bit, (3) feasibility of the projected solution, (4) idempotence of a
    0-iteration solve, (5) odd symmetry in x0."""
    P = pkg.problems
    prob = P.quadrotor(50)
    settings = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=40, check_termination=1)
    B = 8192
    x0s = P.quadrotor_batch_x0(B)
    s = make_solver(pkg, prob, settings, batch=B)
    s.set_x0_batch(x0s)
    s.solve()
    sol = s.get_solution_batch()
    st = s.get_stats_batch()
    assert np.all(st["iter"] == 40) and np.all(st["status"] == 11)
    assert np.all(sol["controls"] <= 0.5 + 1e-15) and np.all(sol["controls"] >= -0.5 - 1e-15)
    assert np.all(np.abs(sol["states"]) <= 5.0 + 1e-15)
    sample = np.random.default_rng(0).choice(B, size=24, replace=False)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, _, _, _ = orc.solve_batch(x0s[:, sample])
    assert rel_err(sol["states"][:, :, sample], ox) < TOL
    assert rel_err(sol["controls"][:, :, sample], ou) < TOL
    s.reset_workspace()
    s.set_x0_batch(np.ascontiguousarray(x0s[:, ::-1]))
    s.solve()
    sol_r = s.get_solution_batch()
    np.testing.assert_array_equal(sol_r["controls"][:, :, ::-1], sol["controls"])
    np.testing.assert_array_equal(sol_r["states"][:, :, ::-1], sol["states"])
    s.update_settings(max_iter=0)
    s.solve()
    np.testing.assert_array_equal(s.get_solution_batch()["controls"], sol_r["controls"])
    # (5) odd symmetry: with bounds symmetric about zero, zero references and no affine term every operation of the solve is odd
    # in x0 (mat-vecs, the clamp, the dual update; the residual maxima are even) -- the negated batch gives the negated trajectories
    s.update_settings(max_iter=40)
    s.reset_workspace()
    s.set_x0_batch(np.ascontiguousarray(-x0s))
    s.solve()
    sol_n = s.get_solution_batch()
    np.testing.assert_array_equal(sol_n["controls"], -sol["controls"])
    np.testing.assert_array_equal(sol_n["states"], -sol["states"])
    s.reset()


def test_bench_size_solve_matches_golden_and_oracle(pkg, kernel_layout):
    """The configuration bench.py times -- 8,192 quadrotor N=50 instances, cold start, 200 forced iterations -- checked
    at that size (not only through properties): instances 0..63 against the reference core's own output
    (tests/golden/quadrotor_batch64.npz), a seeded sample of 64 instances from the whole batch against the oracle, and
    all iteration counts. Layout C handles one instance per workgroup and is far off its range at 8,192 x 200, so it
    runs the same check on 1,024 instances."""
    P = pkg.problems
    g = golden("quadrotor_batch64")
    prob = P.quadrotor(50)
    settings = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=200, check_termination=1)
    B = 1024 if kernel_layout == "C" else 8192
    x0s = P.quadrotor_batch_x0(B)
    np.testing.assert_array_equal(x0s[:, :64], g["x0s"])  # the bench's seeded x0 ARE the fixture's
    s = make_solver(pkg, prob, settings, batch=B)
    assert s.launch_info()["layout"] == kernel_layout
    s.set_x0_batch(x0s)
    s.solve()
    st = s.get_stats_batch()
    assert np.all(st["iter"] == 200) and np.all(st["status"] == 11)
    head = s.get_solution_batch(0, 64)
    assert rel_err(head["states"], g["sol_x"]) < TOL
    assert rel_err(head["controls"], g["sol_u"]) < TOL
    assert rel_err(st["residuals"][:, :64], g["residuals"]) < 1e-6
    sample = np.sort(np.random.default_rng(7).choice(np.arange(64, B), size=64, replace=False))
    sol = s.get_solution_batch()
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, _, ores = orc.solve_batch(x0s[:, sample])
    assert rel_err(sol["states"][:, :, sample], ox) < TOL
    assert rel_err(sol["controls"][:, :, sample], ou) < TOL
    assert rel_err(st["residuals"][:, sample], ores) < 1e-6
    s.reset()


@pytest.mark.layouts("D")
def test_config5_on_one_gpu_properties(pkg, kernel_layout):
    """BASELINE config 5 in one launch: all 65,536 quadrotor instances on ONE GPU (8 rounds of 8 waves per CU).
    The first 8,192 instances must equal, bit for bit, what the 8,192-instance shard produces (an instance's
    result does not depend on the batch it is solved in or on the wave slot it lands in); the last 64 are checked against
    the oracle; feasibility everywhere."""
    P = pkg.problems
    prob = P.quadrotor(50)
    settings = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=60, check_termination=1)
    B = 65536
    x0s = P.quadrotor_batch_x0(B)
    big = make_solver(pkg, prob, settings, batch=B)
    big.set_x0_batch(x0s)
    big.solve()
    st = big.get_stats_batch()
    assert np.all(st["iter"] == 60) and np.all(st["status"] == 11)
    small = make_solver(pkg, prob, settings, batch=8192)
    small.set_x0_batch(x0s[:, :8192])
    small.solve()
    a, b = big.get_solution_batch(0, 8192), small.get_solution_batch()
    np.testing.assert_array_equal(a["controls"], b["controls"])
    np.testing.assert_array_equal(a["states"], b["states"])
    tail = big.get_solution_batch(B - 64, 64)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, _, _, _ = orc.solve_batch(x0s[:, B - 64:])
    assert rel_err(tail["states"], ox) < TOL and rel_err(tail["controls"], ou) < TOL
    u = big.get_first_controls_batch()
    assert np.all(np.abs(u) <= 0.5 + 1e-15)
    big.reset()
    small.reset()


def test_per_tick_references_stay_one_launch_and_match_the_device_path(pkg):
    """The reference's tracking loops re-send the references every tick (rocket_landing_constraints.m:86-121,
    cartpole_example_mpc_reference_constrained.m:47-55). On a single-instance handle set_x_ref / set_u_ref only fill
    pinned host memory and the solve kernel rebuilds the reference-dependent table rows itself; a batched handle
    uploads them and runs k_build_tables. Both must agree bit for bit, tick by tick, and with the oracle."""
    P = pkg.problems
    prob = P.quadrotor(20)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=80)
    one = make_solver(pkg, prob, settings, batch=1)
    two = make_solver(pkg, prob, settings, batch=2)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    goal = np.array([1.0, -0.5, 0.8, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    x = prob.x0.copy()
    total = 40
    for k in range(10):
        x_ref = np.stack([prob.x0 + (goal - prob.x0) * min(i + k, total - 1) / (total - 1) for i in range(prob.N)], axis=1)
        u_ref = np.full((prob.nu, prob.N - 1), 0.01 * k)
        for h in (one, two, orc):
            h.set_x_ref(x_ref)
            h.set_u_ref(u_ref)
        one.set_x0(x)
        two.set_x0_batch(np.stack([x, x], axis=1))
        orc.set_x0(x)
        one.solve()
        two.solve()
        orc.solve()
        a, b = one.get_solution(), two.get_solution_batch()
        np.testing.assert_array_equal(a["controls"], b["controls"][:, :, 0])
        np.testing.assert_array_equal(a["states"], b["states"][:, :, 1])
        assert one.get_stats()["iter"] == two.get_stats_batch()["iter"][0] == orc.stats()["iter"]
        assert rel_err(a["controls"], orc.solution()[1]) < TOL and rel_err(a["states"], orc.solution()[0]) < TOL
        x = prob.A @ x + prob.B @ a["controls"][:, 0]
    # a reference set while the pinned copy is pending must survive a bounds change (k_build_tables runs first)
    one.set_x_ref(np.tile(goal[:, None], (1, prob.N)))
    one.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min * 0.9, prob.u_max * 0.9)
    two.set_x_ref(np.tile(goal[:, None], (1, prob.N)))
    two.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min * 0.9, prob.u_max * 0.9)
    one.solve()
    two.solve()
    np.testing.assert_array_equal(one.get_solution()["controls"], two.get_solution_batch()["controls"][:, :, 0])
    one.reset()
    two.reset()


@pytest.mark.layouts("A", "D")
@pytest.mark.parametrize("nx,nu,N", [(24, 8, 30), (20, 4, 30), (48, 16, 20), (40, 8, 20)])
def test_wide_systems_on_layout_d(pkg, kernel_layout, nx, nu, N):
    """The shapes compiled into the wide forms of layout D (32 / 64 lanes per instance, duals in registers, two wavefronts
    per SIMD, tinympc_solve_dw.hip / tinympc_solve_dx.hip) against the oracle: per-instance termination inside a wavefront, ragged batch, warm start after
    converged and unconverged solves, forced iteration counts; and layout A on the same handle state in between."""
    P = pkg.problems
    rng = np.random.default_rng(nx * 100 + nu)
    A = np.eye(nx) + 0.03 * rng.standard_normal((nx, nx))
    B = 0.1 * rng.standard_normal((nx, nu))
    prob = P.Problem("wide", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
    prob.x_ref = np.tile(0.1 * rng.standard_normal((nx, 1)), (1, N))   # time-invariant: layout D applies
    prob.u_ref = np.tile(0.05 * rng.standard_normal((nu, 1)), (1, N - 1))
    batch = 7
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.05, 1.5, batch)[None, :]
    for settings in (dict(max_iter=150, abs_pri_tol=1e-3, abs_dua_tol=1e-3), dict(max_iter=40, abs_pri_tol=0.0, abs_dua_tol=0.0),
                     dict(max_iter=90, abs_pri_tol=1e-4, abs_dua_tol=1e-4, check_termination=3)):
        s = make_solver(pkg, prob, settings, batch=batch)
        assert s.launch_info()["layout"] == kernel_layout and s.launch_info()["lanes_per_instance"] == (32 if nx + nu <= 32 else 64)
        orc = [O.OraclePort(prob).load_problem(prob, settings) for _ in range(batch)]
        for rnd in range(3):  # cold start, then two warm starts from perturbed states
            xs = x0s * (1.0 - 0.3 * rnd)
            s.set_x0_batch(xs)
            s.solve()
            sol, st = s.get_solution_batch(), s.get_stats_batch()
            for b in range(batch):
                orc[b].set_x0(xs[:, b])
                orc[b].solve()
                assert st["iter"][b] == orc[b].stats()["iter"], (rnd, b)
                assert st["status"][b] == orc[b].stats()["status"]
                assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL
                assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL
        s.reset()


@pytest.mark.parametrize("nx,nu,N", [(12, 4, 20), (12, 4, 30), (6, 3, 25), (20, 6, 15), (30, 10, 9),
                                     (12, 4, 75), (12, 4, 100), (6, 3, 100), (24, 8, 60), (48, 16, 40)])
@pytest.mark.layouts("D")
def test_run_time_specialised_layout_d(pkg, kernel_layout, monkeypatch, nx, nu, N):
    """Shapes that are NOT compiled into the library get layout D through hiprtc (tinympc_jit.hip): default for large
    batches, same results as the oracle, TINYMPC_JIT=0 falls back to layout B / A. The second row are horizons whose duals
    do not fit 256 registers: the plan with one wavefront per SIMD (512 registers, four wavefronts per workgroup)."""
    monkeypatch.delenv("TINYMPC_LAYOUT")  # the library's own choice
    P = pkg.problems
    rng = np.random.default_rng(nx * 1000 + N)
    if (nx, nu) == (12, 4):
        prob = P.quadrotor(N)
    else:
        A = np.eye(nx) + 0.03 * rng.standard_normal((nx, nx))
        B = 0.1 * rng.standard_normal((nx, nu))
        prob = P.Problem("jit", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
        prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
        prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
    settings = dict(max_iter=120, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    batch = 1500
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.02, 1.0, batch)[None, :]
    s = make_solver(pkg, prob, settings, batch=batch)
    assert s.launch_info()["layout"] == "D"
    s.set_x0_batch(x0s)
    s.solve()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    sample = [0, 1, 2, 3, 700, 1498, 1499]
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, ost, _ = orc.solve_batch(x0s[:, sample])
    np.testing.assert_array_equal(st["iter"][sample], oit)
    np.testing.assert_array_equal(st["status"][sample], ost)
    assert rel_err(sol["states"][:, :, sample], ox) < TOL and rel_err(sol["controls"][:, :, sample], ou) < TOL
    s.reset()
    monkeypatch.setenv("TINYMPC_JIT", "0")
    s = make_solver(pkg, prob, settings, batch=batch)
    assert s.launch_info()["layout"] in ("A", "B")
    s.set_x0_batch(x0s)
    s.solve()
    np.testing.assert_array_equal(s.get_stats_batch()["iter"], st["iter"])  # every instance, both kernels
    assert rel_err(s.get_solution_batch()["controls"], sol["controls"]) < TOL
    s.reset()


@pytest.mark.layouts("D")
@pytest.mark.parametrize("shape", ["cartpole20", "quadrotor25", "quadrotor80", "wide24x8x30", "wide20x6x15", "wide48x16x20", "wide24x8x60"])
def test_layout_d_with_bounds_and_references_that_vary_over_the_horizon(pkg, kernel_layout, monkeypatch, shape):
    """Layout D reads per-knot bounds / references from the workgroup's LDS copy of the tables (the compiled-in cartpole
    shape, run-time specialised ones, the 32- and 64-lane forms -- there this variant is always specialised at run time,
    also for shapes whose constant-table kernel is compiled in): large batch, library's own layout choice."""
    monkeypatch.delenv("TINYMPC_LAYOUT")
    P = pkg.problems
    rng = np.random.default_rng(5)
    if shape.startswith("wide"):
        nx, nu, N = (int(t) for t in shape[4:].split("x"))
        A = np.eye(nx) + 0.03 * rng.standard_normal((nx, nx))
        B = 0.1 * rng.standard_normal((nx, nu))
        prob = P.Problem("wide", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    else:
        prob = P.cartpole(20, True) if shape == "cartpole20" else P.quadrotor(int(shape[9:]))  # (80: one wavefront per SIMD)
    nx, nu, N = prob.nx, prob.nu, prob.N
    xlim, ulim = (2.0, 0.5) if shape == "cartpole20" else (2.0, 0.3) if shape.startswith("wide") else (5.0, 0.5)
    prob.x_min = -xlim - rng.uniform(0, 0.5, (nx, N))
    prob.x_max = xlim + rng.uniform(0, 0.5, (nx, N))
    prob.u_min = -ulim * rng.uniform(0.6, 1.0, (nu, N - 1))
    prob.u_max = ulim * rng.uniform(0.6, 1.0, (nu, N - 1))
    prob.x_ref = 0.05 * rng.standard_normal((nx, N))
    prob.u_ref = 0.02 * rng.standard_normal((nu, N - 1))
    settings = dict(max_iter=150, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    batch = 1300
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.02, 0.6, batch)[None, :]
    s = make_solver(pkg, prob, settings, batch=batch)
    s.set_x0_batch(x0s)
    s.solve()
    assert s.launch_info()["layout"] == "D"
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    sample = [0, 1, 2, 3, 650, 1298, 1299]
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, ost, _ = orc.solve_batch(x0s[:, sample])
    np.testing.assert_array_equal(st["iter"][sample], oit)
    np.testing.assert_array_equal(st["status"][sample], ost)
    assert rel_err(sol["states"][:, :, sample], ox) < TOL and rel_err(sol["controls"][:, :, sample], ou) < TOL
    s.reset()


@pytest.mark.parametrize("batch", [1, 6, 300])
def test_mpc_step_equals_the_three_verb_tick(pkg, batch):
    """tinympc_mpc_step_batch == set_x0_batch + solve + get_first_controls_batch, bit for bit, over a
    warm-started closed loop (examples/cartpole_example_mpc.m:36-44 without the noise). Up to 256 instances the
    tick exchanges x0 / u0 through pinned host memory (no copy engine), beyond that through two async copies; after
    a zero-copy tick the device copy of x0 must be current too (a plain solve() follows below)."""
    P = pkg.problems
    prob = P.quadrotor(20)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60)
    a = make_solver(pkg, prob, settings, batch=batch)
    b = make_solver(pkg, prob, settings, batch=batch)
    x = np.repeat(prob.x0[:, None], batch, axis=1) * np.linspace(0.4, 1.0, batch)[None, :]
    xa, xb = x.copy(), x.copy()
    for _ in range(6):
        ua = a.mpc_step(xa)
        b.set_x0_batch(xb)
        b.solve()
        ub = b.get_first_controls_batch()
        np.testing.assert_array_equal(ua, ub)
        np.testing.assert_array_equal(a.get_stats_batch()["iter"], b.get_stats_batch()["iter"])
        xa = prob.A @ xa + prob.B @ ua
        xb = prob.A @ xb + prob.B @ ub
    # the x0 of the last tick is what a following plain solve() starts from, in both handles
    a.mpc_step(xa)
    b.set_x0_batch(xb)
    b.solve()
    a.solve()
    b.solve()
    np.testing.assert_array_equal(a.get_first_controls_batch(), b.get_first_controls_batch())
    np.testing.assert_array_equal(a.get_stats_batch()["iter"], b.get_stats_batch()["iter"])
    a.reset()
    b.reset()


@pytest.mark.layouts("D")
@pytest.mark.parametrize("case", ["varying_tables", "families"])
def test_first_tick_on_a_layout_d_variant_that_is_decided_late(pkg, kernel_layout, monkeypatch, case):
    """Layout D's per-knot-table and family variants are specialised when first needed. The batched tick asks "layout D?" before
    it launches (small batches exchange x0 / u0 through pinned memory on the other layouts only): the answer must not change
    between that question and the launch, also on the very first tick (TINYMPC_LAYOUT=D forces layout D on a small batch)."""
    P = pkg.problems
    batch = 64
    if case == "families":
        prob = P.rocket(10)
    else:
        prob = P.quadrotor(20)
        rng = np.random.default_rng(4)
        prob.x_ref = 0.05 * rng.standard_normal((prob.nx, prob.N))
        prob.u_ref = 0.02 * rng.standard_normal((prob.nu, prob.N - 1))
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=40)

    def build():
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x_ref(prob.x_ref)
        s.set_u_ref(prob.u_ref)
        if case == "families":
            s.set_cone_constraints(**prob.cones)
            s.set_linear_constraints(**prob.linear)
        return s

    a, b = build(), build()
    x = np.repeat(prob.x0[:, None], batch, axis=1) * np.linspace(0.6, 1.0, batch)[None, :]
    f = prob.fdyn[:, None] if prob.fdyn is not None else 0.0
    for _ in range(3):
        ua = a.mpc_step(x)
        b.set_x0_batch(x)
        b.solve()
        np.testing.assert_array_equal(ua, b.get_first_controls_batch())
        assert a.launch_info()["layout"] == "D" and b.launch_info()["layout"] == "D"
        x = prob.A @ x + prob.B @ ua + f
    a.reset()
    b.reset()


@pytest.mark.parametrize("seed", range(24))
def test_randomized_problems(pkg, seed):
    """Random stable-ish systems: random nx, nu (all W=16 register widths), horizons on both sides of the
    layout-B threshold and of the 4-step unroll remainders, random per-knot bounds and references, random
    batch sizes, tolerances that make instances stop at different iterations. Some of these systems are
    open-loop unstable and poorly damped, so the reassociated mat-vec differs from the oracle by up to ~2e-9
    relative after 90 iterations: asserted at 1e-7 (bar 1e-6), iteration counts and statuses exact."""
    P = pkg.problems
    rng = np.random.default_rng(1000 + seed)
    nx = int(rng.integers(1, 14))
    nu = int(rng.integers(1, min(4, 16 - nx) + 1))
    N = int(rng.choice([2, 3, 5, 8, 9, 10, 11, 17, 26, 31, 44]))
    A = np.eye(nx) * rng.uniform(0.9, 1.02) + 0.04 * rng.standard_normal((nx, nx))
    B = 0.15 * rng.standard_normal((nx, nu))
    prob = P.Problem("rand", A, B, np.diag(rng.uniform(0.5, 20, nx)), np.diag(rng.uniform(0.1, 3, nu)), N,
                     float(rng.uniform(0.2, 6)), rng.standard_normal(nx))
    if rng.random() < 0.8:
        prob.x_min = -1.5 - rng.uniform(0, 1, (nx, N))
        prob.x_max = 1.5 + rng.uniform(0, 1, (nx, N))
        prob.u_min = -0.4 - rng.uniform(0, 0.3, (nu, N - 1))
        prob.u_max = 0.4 + rng.uniform(0, 0.3, (nu, N - 1))
    if rng.random() < 0.6:
        prob.x_ref = 0.3 * rng.standard_normal((nx, N))
        prob.u_ref = 0.1 * rng.standard_normal((nu, N - 1))
    settings = dict(max_iter=int(rng.integers(1, 90)), abs_pri_tol=float(rng.choice([0.0, 1e-2, 1e-3])),
                    abs_dua_tol=float(rng.choice([1e-2, 1e-3])), check_termination=int(rng.choice([1, 1, 2, 5])))
    batch = int(rng.integers(1, 23))
    x0s = rng.standard_normal((nx, batch)) * rng.uniform(0.1, 1.5, (1, batch))
    s = make_solver(pkg, prob, settings, batch=batch)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    s.set_x0_batch(x0s)
    s.solve()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    # a second handle is solved twice: the second solve is warm-started from whatever the first one left
    s2 = make_solver(pkg, prob, settings, batch=batch)
    s2.set_x0_batch(x0s)
    s2.solve()
    s2.solve()
    sol2, st2 = s2.get_solution_batch(), s2.get_stats_batch()
    for b in range(batch):
        o = O.OraclePort(prob).load_problem(prob, settings)
        o.set_x0(x0s[:, b])
        o.solve()
        assert st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"], (b, "cold")
        assert rel_err(sol["states"][:, :, b], o.solution()[0], eps=1e-12) < TOL_RANDOM
        assert rel_err(sol["controls"][:, :, b], o.solution()[1], eps=1e-12) < TOL_RANDOM
        o.solve()
        assert st2["iter"][b] == o.stats()["iter"] and st2["status"][b] == o.stats()["status"], (b, "warm")
        assert rel_err(sol2["states"][:, :, b], o.solution()[0], eps=1e-12) < TOL_RANDOM
        assert rel_err(sol2["controls"][:, :, b], o.solution()[1], eps=1e-12) < TOL_RANDOM
    s.reset()
    s2.reset()


@pytest.mark.parametrize("N", [159, 400])
def test_long_horizon_state_in_global_memory(pkg, N):
    """Horizons whose ADMM state exceeds the 160 KB of LDS (quadrotor: N > 158) run the layout-A kernel on an
    HBM working copy; results and iteration counts are unchanged."""
    P = pkg.problems
    prob = P.quadrotor(N)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=40)
    x0s = P.quadrotor_batch_x0(5)
    s = make_solver(pkg, prob, settings, batch=5)
    assert s.launch_info()["lds_bytes"] == 0 and s.launch_info()["layout"] == "A"
    s.set_x0_batch(x0s)
    s.solve()
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, ost, _ = orc.solve_batch(x0s)
    st, sol = s.get_stats_batch(), s.get_solution_batch()
    np.testing.assert_array_equal(st["iter"], oit)
    np.testing.assert_array_equal(st["status"], ost)
    assert rel_err(sol["states"], ox) < TOL and rel_err(sol["controls"], ou) < TOL
    s.reset()


@pytest.mark.layouts("C")
@pytest.mark.parametrize("N", [2, 3, 16, 17, 18, 33, 34, 49, 65, 66, 100, 129, 130])
def test_chunked_layout_horizon_boundaries(pkg, kernel_layout, N):
    """Layout C cuts the N-1 steps into <= 16 chunks of S = ceil((N-1)/16) steps: horizons on either side of every
    change of S and of the chunk count, the 8-slot variant (N > 65), a partial last chunk, and N = 130 where layout
    C no longer applies. Cold solve, warm-started second solve and a converging solve, against the oracle."""
    P = pkg.problems
    for prob in (P.cartpole(N, True), P.quadrotor(N)):
        settings = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=23, check_termination=2)
        s = make_solver(pkg, prob, settings, batch=3)
        assert (s.launch_info()["layout"] == "C") == (N <= 129)
        x0s = np.stack([prob.x0, -0.6 * prob.x0, 0.3 * prob.x0], axis=1)
        s.set_x0_batch(x0s)
        orcs = [O.OraclePort(prob).load_problem(prob, settings) for _ in range(3)]
        for solve in range(2):
            s.solve()
            sol, st = s.get_solution_batch(), s.get_stats_batch()
            for b, o in enumerate(orcs):
                o.set_x0(x0s[:, b])
                o.solve()
                assert st["iter"][b] == o.stats()["iter"]
                assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL, (N, solve, b)
                assert rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, (N, solve, b)
        s.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=400, check_termination=1)
        s.solve()
        st = s.get_stats_batch()
        for b, o in enumerate(orcs):
            o.update_settings(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=400, check_termination=1)
            o.solve()
            assert (st["iter"][b], st["status"][b]) == (o.stats()["iter"], o.stats()["status"]), (N, b)
        # one more solve from the state a converged solve leaves behind (stale v/z, admm.cpp:181-197)
        s.solve()
        st = s.get_stats_batch()
        for b, o in enumerate(orcs):
            o.solve()
            assert st["iter"][b] == o.stats()["iter"], (N, b)
        s.reset()


@pytest.mark.parametrize("shape", ["quadrotor50", "wide24"])
def test_reset_workspace_is_a_contract_not_a_memset(pkg, monkeypatch, shape):
    """tinympc_reset_workspace no longer writes zeros into G, V, D: layout D starts a cold solve from zero registers
    (SolveParams::cold) and every other kernel gets the zeros written first (materialize_cold_state). Whatever runs between the
    reset and the next real solve -- nothing, a solve of zero iterations (writes nothing back), a launch on another kernel -- the
    cold solve must reproduce the handle's very first solve bit for bit, and a warm start behind it the first warm start."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    P = pkg.problems
    if shape == "quadrotor50":
        prob, batch = P.quadrotor(50), 1300
        x0s = np.asfortranarray(P.quadrotor_batch_x0(batch))
    else:
        rng = np.random.default_rng(0)
        nx, nu, batch = 24, 8, 300
        prob = P.Problem("wide", np.eye(nx) + 0.03 * rng.standard_normal((nx, nx)), 0.1 * rng.standard_normal((nx, nu)), np.diag(rng.uniform(1, 10, nx)),
                         np.diag(rng.uniform(0.5, 2, nu)), 30, 2.0, rng.standard_normal(nx))
        prob.x_min, prob.x_max, prob.u_min, prob.u_max = np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3)
        x0s = np.asfortranarray(prob.x0[:, None] + 0.1 * rng.standard_normal((nx, batch)))
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=25, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(x0s)
    s.solve()
    assert s.launch_info()["layout"] == "D"
    cold = s.get_solution_batch()["controls"].copy()
    s.solve()
    warm = s.get_solution_batch()["controls"].copy()
    assert not np.array_equal(cold, warm)
    for between in ("nothing", "zero_iterations", "other_kernel"):
        s.reset_workspace()
        if between == "zero_iterations":
            s.update_settings(max_iter=0)
            s.solve()
            assert np.all(s.get_solution_batch()["controls"] == 0.0)
            s.update_settings(max_iter=25)
        if between == "other_kernel":  # a zero-iteration... no: one real cold solve on layout B / A, then reset again and back to layout D
            monkeypatch.setenv("TINYMPC_LAYOUT", "A")
            t = pkg.TinyMPC()
            t.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=8, rho=prob.rho, max_iter=25, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
            t.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            t.set_x0_batch(x0s[:, :8])
            t.reset_workspace()
            t.solve()
            np.testing.assert_allclose(t.get_solution_batch()["controls"], cold[:, :, :8], rtol=0, atol=1e-9 * np.max(np.abs(cold)))
            t.reset()
            monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
        s.solve()
        np.testing.assert_array_equal(s.get_solution_batch()["controls"], cold, err_msg=between)
        s.solve()
        np.testing.assert_array_equal(s.get_solution_batch()["controls"], warm, err_msg=between)
    s.reset()
