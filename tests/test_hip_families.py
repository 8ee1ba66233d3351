"""GPU tests of the second-order-cone / linear-inequality / fdyn path (BASELINE config 4, rocket landing).

PARITY UNPINNED: the reference tree holds no source for these families, so the checker
here is this repo's restatement of the upstream algorithm (oracle/tinympc_oracle.c), itself validated
by first principles in tests/test_oracle_golden.py. These tests pin the HIP kernel to that restatement
(1e-9 relative, identical iteration counts) and re-check the defining properties on the GPU results."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


LAYOUTS = ("A", "C")  # (conftest.pytest_generate_tests: the default of this module; @pytest.mark.layouts(...) narrows it per test)


@pytest.fixture(autouse=True)
def kernel_layout(request, monkeypatch):
    """The families run in two kernels: k_admm_solve_fam (layout A: batches, wide systems, long horizons) and the
    FAM variant of the latency kernel k_admm_solve_c (one instance per workgroup); every test runs against both unless its
    `layouts` marker names the one pass it needs (tests that choose their kernels themselves)."""
    monkeypatch.setenv("TINYMPC_LAYOUT", request.param)
    return request.param


def make(pkg, prob, settings, batch=1):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn,
            **{k: v for k, v in settings.items() if k in ("abs_pri_tol", "abs_dua_tol", "max_iter", "check_termination")})
    if prob.has_bounds():
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.cones:
        s.set_cone_constraints(**prob.cones)
    if prob.linear:
        s.set_linear_constraints(**prob.linear)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref)
    if prob.u_ref is not None:
        s.set_u_ref(prob.u_ref)
    s.set_x0(prob.x0)
    return s


def oracle(prob, settings):
    return O.OraclePort(prob).load_problem(prob, settings)


@pytest.mark.parametrize("variant", ["cones", "linear", "both", "state_cone_only", "input_cone_only"])
@pytest.mark.parametrize("N", [10, 100])
def test_rocket_matches_restated_oracle(pkg, variant, N):
    P = pkg.problems
    rk = P.rocket(N, with_linear=variant in ("linear", "both"))
    if variant == "linear":
        rk.cones = {}
    if variant == "state_cone_only":
        rk.cones = dict(Acx=[0], qcx=[3], cx=[0.5], Acu=[], qcu=[], cu=[])
    if variant == "input_cone_only":
        rk.cones = dict(Acx=[], qcx=[], cx=[], Acu=[0], qcu=[3], cu=[0.25])
    settings = dict(max_iter=120, abs_pri_tol=2e-3, abs_dua_tol=1e-4)
    s = make(pkg, rk, settings)
    orc = oracle(rk, settings)
    s.solve()
    orc.solve()
    st, ost = s.get_stats(), orc.stats()
    assert st["iter"] == ost["iter"] and st["status"] == ost["status"]
    sol = s.get_solution()
    assert rel_err(sol["states"], orc.solution()[0]) < TOL
    assert rel_err(sol["controls"], orc.solution()[1]) < TOL
    s.reset()


def test_rocket_closed_loop_warm_start(pkg):
    """examples/rocket_landing_constraints.m:86-121 in miniature: shifted references every tick, warm start."""
    P = pkg.problems
    rk = P.rocket(10, with_linear=True)
    settings = dict(max_iter=100, abs_pri_tol=2e-3, abs_dua_tol=1e-4)
    s = make(pkg, rk, settings)
    orc = oracle(rk, settings)
    x = rk.x0.copy()
    xinit, NT = np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5]), 100
    for k in range(8):
        x_ref = np.stack([xinit + (0 - xinit) * (i + k) / (NT - 1) for i in range(rk.N)], axis=1)
        for solver in (s, orc):
            solver.set_x0(x)
            solver.set_x_ref(x_ref)
        s.solve()
        orc.solve()
        assert s.get_stats()["iter"] == orc.stats()["iter"], k
        u = s.get_solution()["controls"][:, 0]
        assert rel_err(u, orc.solution()[1][:, 0]) < TOL
        x = rk.A @ x + rk.B @ u + rk.fdyn
    s.reset()


def test_families_batch_and_first_principles(pkg):
    P = pkg.problems
    rk = P.rocket(40, with_linear=True)
    settings = dict(max_iter=400, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    B = 9
    rng = np.random.default_rng(5)
    x0s = rk.x0[:, None] * rng.uniform(0.6, 1.2, (1, B)) + 0.1 * rng.standard_normal((6, B))
    s = make(pkg, rk, settings, batch=B)
    s.set_x0_batch(x0s)
    s.solve()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    orc = oracle(rk, settings)
    ox, ou, oit, ost, _ = orc.solve_batch(x0s)
    np.testing.assert_array_equal(st["iter"], oit)
    np.testing.assert_array_equal(st["status"], ost)
    assert rel_err(sol["states"], ox) < TOL and rel_err(sol["controls"], ou) < TOL
    # box feasibility of the returned (projected) solution
    u, x = sol["controls"], sol["states"]
    assert np.all(u >= rk.u_min[:, None, None] - 1e-12) and np.all(u <= rk.u_max[:, None, None] + 1e-12)
    assert np.all(x >= rk.x_min[:, None, None] - 1e-12) and np.all(x <= rk.x_max[:, None, None] + 1e-12)
    # converged instances satisfy the thrust cone up to the ADMM tolerance
    conv = st["status"] == 1
    assert conv.any()
    uc = u[:, :, conv]
    viol = np.linalg.norm(uc[:2], axis=0) - 0.25 * uc[2]
    assert np.max(viol) < 5e-2
    s.reset()


def test_cone_flags_can_be_switched_off_again(pkg):
    """update_settings(en_*_soc=0) after set_cone_constraints must give the box-only result (bit for bit the
    default kernel's), and switching back on must match the oracle again."""
    P = pkg.problems
    rk = P.rocket(20, with_linear=False)
    settings = dict(max_iter=60, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
    s = make(pkg, rk, settings)
    s.update_settings(en_state_soc=False, en_input_soc=False)
    s.solve()
    plain = P.rocket(20, with_linear=False)
    plain.cones = {}
    o2 = oracle(plain, settings)
    o2.solve()
    assert s.get_stats()["iter"] == o2.stats()["iter"]
    assert rel_err(s.get_solution()["controls"], o2.solution()[1]) < TOL
    s.reset()


def test_unsupported_family_shapes_fail_loudly(pkg):
    P = pkg.problems
    rk = P.rocket(10, with_linear=False)
    rk.cones = {}
    s = make(pkg, rk, {})
    with pytest.raises(pkg.TinyMPCError) as ei:  # more cones than the family buffer ever holds (64)
        s.set_cone_constraints([0] * 65, [2] * 65, [0.5] * 65, [], [], [])
    assert ei.value.code == pkg._lib.ERR_UNSUPPORTED
    with pytest.raises(pkg.TinyMPCError) as ei:  # cone outside the state vector
        s.set_cone_constraints([4], [3], [0.5], [], [], [])
    assert ei.value.code == pkg._lib.ERR_INVALID_INPUT
    with pytest.raises(pkg.TinyMPCError) as ei:  # more rows than the family buffer ever holds (128 per side)
        s.set_linear_constraints(np.ones((129, 6)), np.ones(129), np.zeros((0, 3)), np.zeros(0))
    assert ei.value.code == pkg._lib.ERR_UNSUPPORTED
    s.solve()  # nothing was installed by the failed calls: plain box solve still works
    s.reset()
    # (a chain of five cones, each sharing a row with its predecessor -- five rounds --, 33 linear rows: refused until round 3, run on the
    # structure-specialised kernels since round 4: test_shapes_beyond_the_generic_kernels_limits)


def test_large_batch_of_long_horizons_default_kernel_choice(pkg, monkeypatch):
    """Default kernel choice (no TINYMPC_LAYOUT): a long horizon leaves k_admm_solve_fam one wavefront per CU, so the
    families then run in the latency kernel at every batch size -- here 1100 instances (beyond the box path's
    1024-instance switch-over), spot-checked against the restatement."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    prob = pkg.problems.rocket(100)
    settings = dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=12, check_termination=1)
    batch = 1100
    s = make(pkg, prob, settings, batch=batch)
    scale = np.linspace(0.6, 1.2, batch)
    x0s = prob.x0[:, None] * scale[None, :]
    s.set_x0_batch(x0s)
    s.solve()
    sol = s.get_solution_batch()
    for b in (0, 451, 1099):
        o = O.OraclePort(prob).load_problem(prob, settings)
        o.set_x0(x0s[:, b])
        o.solve()
        assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL
        assert rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL
    s.reset()


@pytest.mark.layouts("A")
@pytest.mark.parametrize("variant", ["cones", "linear", "both", "both_constant_references"])
@pytest.mark.parametrize("N", [10, 20, 28])
def test_families_on_layout_d(pkg, kernel_layout, monkeypatch, variant, N):
    """Short horizons, large batches. The library's own choice since round 4: layout E's UNCUT form (one wavefront per group of four
    instances, the families one knot per lane) up to N = 20, its cut form beyond; layout D's families variant (gc, gl, lx in
    registers next to g and v; -DTINY_JIT_FAM=1) remains behind TINYMPC_LAYOUT=D up to N = 22. Cold start and a warm start against
    the restatement for every instance on the library's choice; the same two solves on layout D's variant and on k_admm_solve_fam
    (TINYMPC_JIT=0) against it."""
    monkeypatch.delenv("TINYMPC_LAYOUT")
    P = pkg.problems
    rk = P.rocket(N, with_linear=variant != "cones")
    if variant == "linear":
        rk.cones = {}
    if variant == "both_constant_references":  # (the kernel variant that keeps bounds / references in registers)
        rk.x_ref = np.repeat(rk.x_ref[:, :1], N, axis=1)
    settings = dict(max_iter=150, abs_pri_tol=2e-3, abs_dua_tol=1e-4)
    batch = 1301  # beyond the latency kernel's range; ragged last wavefront and workgroup
    rng = np.random.default_rng(N)
    x0a = rk.x0[:, None] * rng.uniform(0.6, 1.2, (1, batch)) + 0.1 * rng.standard_normal((6, batch))
    x0b = x0a + 0.05 * rng.standard_normal((6, batch))
    results = {}
    for jit in ("1", "D", "0"):
        if jit == "D" and N > 22:
            continue
        monkeypatch.setenv("TINYMPC_JIT", "0" if jit == "0" else "1")
        if jit == "D":
            monkeypatch.setenv("TINYMPC_LAYOUT", "D")
        else:
            monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
        s = make(pkg, rk, settings, batch=batch)
        out = []
        for x0s in (x0a, x0b):
            s.set_x0_batch(x0s)
            s.solve()
            out.append((s.get_solution_batch(), s.get_stats_batch()))
        # (without the specialiser: k_admm_solve_fam, whatever the box path's layout; N = 28 is beyond what layout D holds in
        # registers with the families -- layout E takes it, four wavefronts per workgroup on the 512-register plan)
        if jit == "1":  # (N = 20 with constant tables: the uncut form's register estimate says no -- layout D's variant takes it)
            assert s.launch_info()["layout"] == ("D" if (N == 20 and variant == "both_constant_references") else "E"), s.jit_info()
            assert ("uncut" in s.jit_info()) == (N <= 20 and s.launch_info()["layout"] == "E"), s.jit_info()
        else:
            assert s.launch_info()["layout"] == ("A" if jit == "0" else "D"), s.jit_info()
        results[jit] = out
        s.reset()
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    orc = [oracle(rk, settings) for _ in range(batch)]
    for rnd, x0s in enumerate((x0a, x0b)):
        sol, st = results["1"][rnd]
        for b in range(batch):
            orc[b].set_x0(x0s[:, b])
            orc[b].solve()
        oit = np.array([o.stats()["iter"] for o in orc])
        ost = np.array([o.stats()["status"] for o in orc])
        np.testing.assert_array_equal(st["iter"], oit)
        np.testing.assert_array_equal(st["status"], ost)
        for b in range(0, batch, 7):
            assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL
            assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL
        for other in ("0", "D"):
            if other in results:
                sol0, st0 = results[other][rnd]
                np.testing.assert_array_equal(st["iter"], st0["iter"])
                assert rel_err(sol["controls"], sol0["controls"]) < TOL and rel_err(sol["states"], sol0["states"]) < TOL
    assert (results["1"][0][1]["status"] == 1).any() and (results["1"][1][1]["iter"] < results["1"][0][1]["iter"]).any()


@pytest.mark.layouts("A")
@pytest.mark.parametrize("variant", ["cones", "linear", "both", "both_constant_references", "input_cone_only"])
@pytest.mark.parametrize("N", [100, 44, 10])
def test_families_on_layout_e(pkg, kernel_layout, monkeypatch, variant, N):
    """Long horizons, batches: the families on layout E (tinympc_solve_e.hip: the horizon cut across the eight wavefronts of
    a workgroup, run-time specialised on the cone list and the linear rows per side). Cold start and a warm start against the
    restatement for every instance -- iteration counts and statuses exact --, and against the latency kernel on the same handle
    (the two share the persistent HBM state). N=100: eight chunks of 13, the last one 8; N=44: eight chunks would leave the last
    wavefront one slot, so the plan is four wavefronts (one per SIMD, 512 registers) with chunks of 11 and 10; N=10: the UNCUT form
    (round 4: one wavefront per group, no pass 1, no carries; until round 3 this shape had no plan at all)."""
    monkeypatch.setenv("TINYMPC_LAYOUT", "E")
    P = pkg.problems
    rk = P.rocket(N, with_linear=variant in ("linear", "both", "both_constant_references"))
    if variant == "linear":
        rk.cones = {}
    if variant == "input_cone_only":
        rk.cones = dict(Acx=[], qcx=[], cx=[], Acu=[0], qcu=[3], cu=[0.25])
    if variant == "both_constant_references":
        rk.x_ref = np.repeat(rk.x_ref[:, :1], N, axis=1)
    settings = dict(max_iter=150, abs_pri_tol=2e-3, abs_dua_tol=1e-4)
    batch = 203  # ragged last group of four
    rng = np.random.default_rng(N)
    x0a = rk.x0[:, None] * rng.uniform(0.6, 1.2, (1, batch)) + 0.1 * rng.standard_normal((6, batch))
    x0b = x0a + 0.05 * rng.standard_normal((6, batch))
    s = make(pkg, rk, settings, batch=batch)
    s.prepare()
    info = s.jit_info()
    assert ("uncut" in info) == (N == 10), info
    assert s.launch_info()["layout"] == "E" and info.startswith(("compiled ", "disk-cache ", "compiled-in ")) and "scratch=0" in info, info
    out = []
    for x0s in (x0a, x0b):
        s.set_x0_batch(x0s)
        s.solve()
        out.append((s.get_solution_batch(), s.get_stats_batch()))
    orc = [oracle(rk, settings) for _ in range(batch)]
    for rnd, x0s in enumerate((x0a, x0b)):
        sol, st = out[rnd]
        for b in range(batch):
            orc[b].set_x0(x0s[:, b])
            orc[b].solve()
        np.testing.assert_array_equal(st["iter"], np.array([o.stats()["iter"] for o in orc]))
        np.testing.assert_array_equal(st["status"], np.array([o.stats()["status"] for o in orc]))
        for b in range(batch):
            assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL, (rnd, b)
            assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, (rnd, b)
        res = np.array([[o.stats()[k] for k in ("pri_x", "dua_x", "pri_u", "dua_u")] if "pri_x" in o.stats() else [np.nan] * 4 for o in orc]).T
        if not np.isnan(res).any():
            assert rel_err(st["residuals"], res) < 1e-6
    if variant != "both_constant_references":  # (a constant reference far from x0 does not converge within max_iter)
        assert (out[0][1]["status"] == 1).any() and (out[1][1]["iter"] < out[0][1]["iter"]).any()
    # a third solve on the latency kernel from the state layout E left behind, against the restatement's third solve
    monkeypatch.setenv("TINYMPC_LAYOUT", "C")
    x0c = x0b * 0.97
    s.set_x0_batch(x0c)
    s.solve()
    assert s.launch_info()["layout"] == "C"
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    for b in range(0, batch, 5):
        orc[b].set_x0(x0c[:, b])
        orc[b].solve()
        assert st["iter"][b] == orc[b].stats()["iter"] and rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, b
    # ... and back
    monkeypatch.setenv("TINYMPC_LAYOUT", "E")
    x0d = x0c * 1.02
    s.set_x0_batch(x0d)
    s.solve()
    assert s.launch_info()["layout"] == "E"
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    for b in range(0, batch, 5):
        orc[b].set_x0(x0d[:, b])
        orc[b].solve()
        assert st["iter"][b] == orc[b].stats()["iter"] and rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, b
    s.reset()


@pytest.mark.layouts("A")
def test_prepare_specialises_before_the_first_solve(pkg, kernel_layout, monkeypatch):
    """tinympc_prepare: the variant decision (here: families on layout D) is taken -- and the kernel built -- before the first
    solve, so that launch_info already names it and the first tick does not pay for it."""
    monkeypatch.delenv("TINYMPC_LAYOUT")
    rk = pkg.problems.rocket(10)
    s = make(pkg, rk, dict(max_iter=30, abs_pri_tol=1e-3, abs_dua_tol=1e-3), batch=1500)
    assert s.launch_info()["layout"] not in ("D", "E")  # not decided yet: k_admm_solve_fam would run
    s.prepare()
    assert s.launch_info()["layout"] == "E" and "uncut" in s.jit_info()
    s.set_x0_batch(np.repeat(rk.x0[:, None], 1500, axis=1))
    s.solve()
    assert s.launch_info()["layout"] == "E" and np.all(s.get_stats_batch()["iter"] > 0)
    s.reset()


@pytest.mark.layouts("A")
@pytest.mark.parametrize("what", ["box", "families", "adaptive_rho"])
def test_layout_d_variants_share_the_persistent_state_with_the_other_kernels(pkg, kernel_layout, monkeypatch, what):
    """One handle, four warm-started solves, the kernel switched between them (TINYMPC_JIT on / off / on / off): layout D and its
    FAM / ADAPT variants keep g, v, d, gc, gl and the per-instance rho in the same HBM arrays as layouts A / B,
    k_admm_solve_fam and k_admm_solve_adapt, so the sequence must equal the restatement's four consecutive solves."""
    monkeypatch.delenv("TINYMPC_LAYOUT")
    P = pkg.problems
    batch = 1100
    rng = np.random.default_rng(8)
    if what == "families":
        prob = P.rocket(12)
        settings = dict(max_iter=40, abs_pri_tol=2e-3, abs_dua_tol=1e-4)
        s = make(pkg, prob, settings, batch=batch)
    else:
        prob = P.quadrotor(18)
        settings = dict(max_iter=25, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
        s = pkg.TinyMPC()
        extra = dict(adaptive_rho=True, adaptive_rho_min=0.2, adaptive_rho_max=40.0) if what == "adaptive_rho" else {}
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, **settings, **extra)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if what == "adaptive_rho":
            dK, dP, dC1, dC2 = s.compute_sensitivity_autograd()
            s.set_sensitivity_matrices(dK, dP, dC1, dC2)
    sample = [0, 1, 2, 3, 550, 1098, 1099]
    orcs = {}
    for b in sample:
        o = oracle(prob, settings)
        if what == "adaptive_rho":
            o.set_adaptive_rho(True, 0.2, 40.0, True)
            o.set_sensitivity(dK, dP)
        orcs[b] = o
    x = prob.x0[:, None] * rng.uniform(0.5, 1.2, (1, batch)) + 0.05 * rng.standard_normal((prob.nx, batch))
    seen = []
    for k, jit in enumerate(("1", "0", "1", "0")):
        monkeypatch.setenv("TINYMPC_JIT", jit)
        xs = x * (1.0 - 0.15 * k)
        s.set_x0_batch(xs)
        s.solve()
        seen.append(s.launch_info()["layout"])
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b, o in orcs.items():
            o.set_x0(xs[:, b])
            o.solve()
            assert st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"], (k, b)
            assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL and rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, (k, b)
    if what == "box":  # (the box kernel is chosen at setup; only the variants are decided per launch)
        assert seen == ["D"] * 4, seen
    else:
        fast = "E" if what == "families" else "D"  # (families at this horizon: layout E's uncut form since round 4)
        assert seen[0] == fast and seen[2] == fast and seen[1] not in ("D", "E") and seen[3] not in ("D", "E"), seen
    s.reset()


def _many_rows(rng, n, dim, point, margin):
    """n half-spaces a'k s <= b_k that a given point satisfies with `margin` to spare (so that the set is not empty)."""
    A = rng.standard_normal((n, dim))
    b = A @ point + margin * rng.uniform(0.5, 1.5, n)
    return A, b


@pytest.mark.layouts("A")
@pytest.mark.parametrize("layout,N,batch", [("A", 12, 3), ("C", 20, 1), ("D", 16, 1301), ("E", 100, 203)])
def test_twelve_linear_rows_per_side(pkg, kernel_layout, monkeypatch, layout, N, batch):
    """More linear rows than any kernel keeps in registers (bindings.cpp:408-431 forwards any number): 12 state rows and 5 input
    rows, walked one after another as upstream does, on every kernel that carries the families -- k_admm_solve_fam (rows beyond
    the eighth from L2), the latency kernel and layout D (LDS, run-time loop), layout E (LDS; more than four rows: one copy of the
    row's code behind a loop)."""
    monkeypatch.setenv("TINYMPC_LAYOUT", layout)
    rk = pkg.problems.rocket(N)
    rng = np.random.default_rng(12)
    Ax, bx = _many_rows(rng, 12, 6, np.array([0.0, 0.0, 5.0, 0.0, 0.0, 0.0]), 30.0)
    Au, bu = _many_rows(rng, 5, 3, np.array([0.0, 0.0, 10.0]), 40.0)
    rk.linear = dict(Alin_x=Ax, blin_x=bx, Alin_u=Au, blin_u=bu)
    settings = dict(max_iter=60, abs_pri_tol=1e-3, abs_dua_tol=1e-4)
    s = make(pkg, rk, settings, batch=batch)
    x0s = rk.x0[:, None] * rng.uniform(0.7, 1.1, (1, batch)) + 0.05 * rng.standard_normal((6, batch))
    if batch > 1:
        s.set_x0_batch(x0s)
    else:
        s.set_x0(x0s[:, 0])
    s.solve()
    assert s.launch_info()["layout"] == layout, s.jit_info()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    for b in sorted({0, batch // 2, batch - 1}):
        o = oracle(rk, settings)
        o.set_x0(x0s[:, b])
        o.solve()
        assert st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"], b
        assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL and rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, b
    s.reset()
    with pytest.raises(pkg.TinyMPCError) as ei:  # the hard limit of the buffer (128 rows per side)
        s2 = make(pkg, rk, settings)
        s2.set_linear_constraints(np.ones((129, 6)), np.ones(129), np.zeros((0, 3)), np.zeros(0))
    assert ei.value.code == pkg._lib.ERR_UNSUPPORTED
    s2.reset()


def _rounds_of(cones):
    """The grouping of a cone list into rounds of pairwise-disjoint cones, as the library does it (family_structure)."""
    rounds, used = 0, set()
    for first, dim in cones:
        rows = set(range(first, first + dim))
        if rounds == 0:
            rounds = 1
        if rows & used:
            rounds += 1
            used = set()
        used |= rows
    return rounds


@pytest.mark.layouts("A")
@pytest.mark.parametrize("batch,N", [(1, 30), (5, 100), (300, 100), (300, 20)])
def test_shapes_beyond_the_generic_kernels_limits(pkg, kernel_layout, monkeypatch, batch, N):
    """The reference forwards ANY constraint list (bindings.cpp:408-478, TinyMPC.m:296-317). Up to 32 linear rows per side, 16
    cones and 4 rounds of overlapping cones every families kernel holds; beyond that (round 4) the family buffer is sized for the
    configuration and the structure-specialised kernels run it: 80 state + 40 input linear rows, 20 cones in 7 rounds (state
    cones that share rows, projected one after another as upstream does), on layout F (one instance / small batches) and
    layout E (batches) -- against the restatement's sequential loops."""
    monkeypatch.delenv("TINYMPC_LAYOUT")
    rk = pkg.problems.rocket(N)
    rng = np.random.default_rng(80)
    Ax, bx = _many_rows(rng, 80, 6, np.array([0.0, 0.0, 5.0, 0.0, 0.0, 0.0]), 30.0)
    Au, bu = _many_rows(rng, 40, 3, np.array([0.0, 0.0, 10.0]), 40.0)
    rk.linear = dict(Alin_x=Ax, blin_x=bx, Alin_u=Au, blin_u=bu)
    state_cones = [(0, 3), (3, 3), (1, 3), (4, 2), (0, 2), (2, 3), (5, 1), (0, 4), (4, 2), (1, 2), (3, 3), (0, 3), (2, 2), (4, 2)]
    input_cones = [(0, 3), (0, 2), (1, 2), (2, 1), (0, 3), (1, 2)]
    assert len(state_cones) + len(input_cones) == 20 and _rounds_of(state_cones + [(6 + f, d) for f, d in input_cones]) >= 6
    rk.cones = dict(Acx=[c[0] for c in state_cones], qcx=[c[1] for c in state_cones], cx=list(rng.uniform(0.3, 0.9, len(state_cones))),
                    Acu=[c[0] for c in input_cones], qcu=[c[1] for c in input_cones], cu=list(rng.uniform(0.2, 0.6, len(input_cones))))
    settings = dict(max_iter=40, abs_pri_tol=1e-3, abs_dua_tol=1e-4)
    s = make(pkg, rk, settings, batch=batch)
    x0s = rk.x0[:, None] * rng.uniform(0.7, 1.1, (1, batch)) + 0.05 * rng.standard_normal((6, batch))
    if batch > 1:
        s.set_x0_batch(x0s)
    else:
        s.set_x0(x0s[:, 0])
    s.solve()
    assert s.launch_info()["layout"] == ("E" if batch >= 260 else "F"), s.jit_info()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    s.solve()  # warm, from the state the first solve left (the duals of every family persist)
    sol2, st2 = s.get_solution_batch(), s.get_stats_batch()
    for b in sorted({0, batch // 2, batch - 1}):
        o = oracle(rk, settings)
        o.set_x0(x0s[:, b])
        for so, sto in ((sol, st), (sol2, st2)):
            o.solve()
            assert sto["iter"][b] == o.stats()["iter"] and sto["status"][b] == o.stats()["status"], b
            assert rel_err(so["states"][:, :, b], o.solution()[0]) < TOL and rel_err(so["controls"][:, :, b], o.solution()[1]) < TOL, b
    s.reset()
    # without the run-time specialiser such a configuration has no kernel: a clear refusal, no silent fallback
    monkeypatch.setenv("TINYMPC_JIT", "0")
    s3 = make(pkg, rk, settings, batch=batch)
    with pytest.raises(pkg.TinyMPCError) as ei:
        s3.solve()
    assert ei.value.code == pkg._lib.ERR_UNSUPPORTED and "run-time" in str(ei.value)
    s3.reset()


@pytest.mark.layouts("A")
@pytest.mark.parametrize("layout,N,batch", [("C", 20, 1), ("E", 100, 203)])
def test_equality_constraints_through_the_class_surface(pkg, kernel_layout, monkeypatch, layout, N, batch):
    """set_equality_constraints (TinyMPC.m:296-317): Aeq s == beq becomes the two inequalities [A; -A] s <= [b; -b] -- five
    equality rows are ten linear rows, more than the kernels' register budget of round 2 allowed. The solution must satisfy the
    equalities within the primal tolerance and equal the restatement's."""
    monkeypatch.setenv("TINYMPC_LAYOUT", layout)
    rk = pkg.problems.rocket(N, with_linear=False)
    rk.cones = dict(Acx=[], qcx=[], cx=[], Acu=[0], qcu=[3], cu=[0.25])
    rng = np.random.default_rng(5)
    Aeq = np.vstack([np.eye(6)[[0, 1, 3]] * [[1.0], [1.0], [0.5]], rng.standard_normal((2, 6)) * 0.2])  # 5 rows
    Aeq[3:, 2] = 0.0
    beq = np.zeros(5)  # x = y = 0, vx = 0 and two more combinations: consistent (the origin satisfies them)
    s = make(pkg, rk, dict(max_iter=80, abs_pri_tol=1e-3, abs_dua_tol=1e-4), batch=batch)
    s.set_equality_constraints(Aeq, beq, np.zeros((0, 3)), np.zeros(0))
    x0s = rk.x0[:, None] * rng.uniform(0.8, 1.1, (1, batch))
    if batch > 1:
        s.set_x0_batch(x0s)
    else:
        s.set_x0(x0s[:, 0])
    s.solve()
    assert s.launch_info()["layout"] == layout, s.jit_info()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    rk2 = pkg.problems.rocket(N, with_linear=False)
    rk2.cones = rk.cones
    rk2.linear = dict(Alin_x=np.vstack([Aeq, -Aeq]), blin_x=np.concatenate([beq, -beq]), Alin_u=np.zeros((0, 3)), blin_u=np.zeros(0))
    for b in sorted({0, batch - 1}):
        o = oracle(rk2, dict(max_iter=80, abs_pri_tol=1e-3, abs_dua_tol=1e-4))
        o.set_x0(x0s[:, b])
        o.solve()
        assert st["iter"][b] == o.stats()["iter"], b
        assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL and rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, b
    s.reset()


@pytest.mark.layouts("A")
@pytest.mark.parametrize("layout,N,batch", [("A", 12, 1), ("A", 30, 37), ("E", 100, 203)])
def test_overlapping_cones_are_projected_one_after_another(pkg, kernel_layout, monkeypatch, layout, N, batch):
    """Two state cones that share rows (rows 0-2 and rows 1-4) plus the input cone: upstream projects the cones of a knot in list
    order (each sees its predecessor's result), which differs from projecting them at once. The kernels group the list into
    rounds of pairwise-disjoint cones: k_admm_solve_fam walks the rounds with mask rows from L2 (any batch, any horizon, also
    where the latency kernel would otherwise run), layout E has them compiled in. Against the restatement (sequential loops)."""
    monkeypatch.setenv("TINYMPC_LAYOUT", layout)
    rk = pkg.problems.rocket(N)
    rk.cones = dict(Acx=[0, 1], qcx=[3, 4], cx=[0.2, 0.3], Acu=[0], qcu=[3], cu=[0.25])
    settings = dict(max_iter=70, abs_pri_tol=1e-3, abs_dua_tol=1e-4)
    rng = np.random.default_rng(3)
    s = make(pkg, rk, settings, batch=batch)
    x0s = rk.x0[:, None] * rng.uniform(0.7, 1.1, (1, batch)) + 0.05 * rng.standard_normal((6, batch))
    if batch > 1:
        s.set_x0_batch(x0s)
    else:
        s.set_x0(x0s[:, 0])
    s.solve()
    assert s.launch_info()["layout"] == layout, s.jit_info()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    differs = False
    for b in sorted({0, batch // 2, batch - 1}):
        o = oracle(rk, settings)
        o.set_x0(x0s[:, b])
        o.solve()
        assert st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"], b
        assert rel_err(sol["states"][:, :, b], o.solution()[0]) < TOL and rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, b
        # (and the order matters: the reversed list gives another trajectory)
        rv = pkg.problems.rocket(N)
        rv.cones = dict(Acx=[1, 0], qcx=[4, 3], cx=[0.3, 0.2], Acu=[0], qcu=[3], cu=[0.25])
        o2 = oracle(rv, settings)
        o2.set_x0(x0s[:, b])
        o2.solve()
        differs = differs or rel_err(o2.solution()[0], o.solution()[0]) > 1e-6
    assert differs, "the test problem does not exercise the order of the projections"
    s.reset()


@pytest.mark.parametrize("case", ["rocket100_both", "rocket100_cones", "rocket100_linear", "rocket100_const_refs", "rocket100_overlap", "rocket44_both",
                                  "rocket10_both", "quadrotor50_box", "quadrotor23_box", "cartpole20_box"])
@pytest.mark.layouts("A")
@pytest.mark.parametrize("batch", [1, 5])
def test_layout_f_the_specialised_latency_kernel(pkg, kernel_layout, monkeypatch, case, batch):
    """Layout F (tinympc_solve_f.hip): one instance per workgroup, up to 32 chunks on the DPP rows of up to eight wavefronts, shape /
    chunk plan / structure of the families compiled in. Three consecutive solves (cold, warm after convergence, warm with new
    references' worth of x0) against the restatement: iteration counts, statuses and residuals exact, trajectories 1e-9 --
    through the single-instance verbs (pinned-host x0 / solution / completion stamp) for batch 1, the batched verbs for 5; then the
    handle continues on the latency kernel of round 1 (layout C) from the state layout F left behind.
    Chunk plans: N=100 -> 25 chunks of 4 (last: 3) on 7 wavefronts; N=44 -> 22 chunks of 2 (last: 1); N=10 -> 5 chunks of 2 (last: 1) on
    2 wavefronts; quadrotor N=50 -> 25 chunks of 2 (last: 1); N=23 -> 11 chunks of 2."""
    monkeypatch.setenv("TINYMPC_LAYOUT", "F")
    P = pkg.problems
    name, variant = case.split("_", 1)
    if name.startswith("rocket"):
        N = int(name[6:])
        prob = P.rocket(N, with_linear=variant in ("both", "linear", "const_refs", "overlap"))
        if variant == "linear":
            prob.cones = {}
        if variant == "const_refs":
            prob.x_ref = np.repeat(prob.x_ref[:, :1], N, axis=1)
        if variant == "overlap":
            prob.cones = dict(Acx=[0, 1], qcx=[3, 4], cx=[0.2, 0.3], Acu=[0], qcu=[3], cu=[0.25])
        settings = dict(max_iter=150, abs_pri_tol=2e-3, abs_dua_tol=1e-4)
    elif name.startswith("quadrotor"):
        prob = P.quadrotor(int(name[9:]))
        settings = dict(max_iter=80, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    else:
        prob = P.cartpole(int(name[8:]), True)
        settings = dict(max_iter=100, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
    rng = np.random.default_rng(len(case) + batch)
    nx = prob.nx
    x0a = prob.x0[:, None] * rng.uniform(0.7, 1.1, (1, batch)) + 0.02 * rng.standard_normal((nx, batch))
    s = make(pkg, prob, settings, batch=batch)
    orc = [oracle(prob, settings) for _ in range(batch)]
    for rnd, x0s in enumerate((x0a, x0a * 0.97, x0a * 1.04)):
        if rnd == 2:
            monkeypatch.setenv("TINYMPC_LAYOUT", "C")  # the same handle, the other latency kernel, the same persistent state
        if batch == 1:
            s.set_x0(x0s[:, 0])
        else:
            s.set_x0_batch(x0s)
        s.solve()
        # (cones that share rows: the round-1 latency kernel has no round-by-round projection, k_admm_solve_fam takes over)
        assert s.launch_info()["layout"] == ("F" if rnd < 2 else "A" if variant == "overlap" else "C"), s.jit_info()
        if rnd == 0:
            info = s.jit_info()
            assert info.startswith(("compiled ", "disk-cache ", "compiled-in ")) and "layout=F" in info and "scratch=0" in info, info
        if batch == 1:
            one, st1 = s.get_solution(), s.get_stats()
            sol = dict(states=one["states"][:, :, None], controls=one["controls"][:, :, None])
            st = dict(iter=np.array([st1["iter"]]), status=np.array([st1["status"]]))
        else:
            sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in range(batch):
            orc[b].set_x0(x0s[:, b])
            orc[b].solve()
            assert st["iter"][b] == orc[b].stats()["iter"] and st["status"][b] == orc[b].stats()["status"], (rnd, b, st["iter"][b], orc[b].stats()["iter"])
            assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL, (rnd, b)
            assert rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, (rnd, b)
        if batch > 1:
            res = np.array([[o.stats()[k] for k in ("pri_x", "dua_x", "pri_u", "dua_u")] for o in orc]).T
            assert rel_err(s.get_stats_batch()["residuals"], res) < 1e-6
    s.reset()


@pytest.mark.layouts("A")
@pytest.mark.parametrize("nx,nu,N,batch", [(20, 4, 12, 37), (24, 8, 10, 5), (40, 10, 8, 9), (48, 16, 6, 3), (24, 8, 30, 70), (40, 10, 9, 33), (20, 4, 7, 16)])
def test_families_on_wide_systems(pkg, kernel_layout, monkeypatch, nx, nu, N, batch):
    """32 and 64 lanes per instance (k_admm_solve_fam's reduction form: a cone's ||w||^2 is a group sum over its tail rows, t one
    lane read, a linear row two group sums -- no mask rows in registers): a cone inside one DPP row, one that straddles the
    boundary between two rows of 16 lanes, two cones that share a row (a second round), an input cone, linear rows on both
    sides and fdyn. Against the restatement: identical iteration counts, 1e-9 on the trajectories, over a cold and a warm start."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    P = pkg.problems
    rng = np.random.default_rng(nx * 100 + nu)
    A = 0.9 * np.eye(nx) + (0.15 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    Bm = 0.3 * rng.standard_normal((nx, nu))
    prob = P.Problem("widefam", A, Bm, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 1.5, rng.standard_normal(nx))
    prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    prob.u_min, prob.u_max = np.full(nu, -1.0), np.full(nu, 1.0)
    prob.fdyn = 0.01 * rng.standard_normal(nx)
    prob.cones = dict(Acx=[0, 13, 15], qcx=[3, 5, 3], cx=[0.8, 0.6, 1.1], Acu=[0], qcu=[3], cu=[0.7])  # rows 0-2 | 13-17 | 15-17 (shares rows)
    prob.linear = dict(Alin_x=rng.standard_normal((3, nx)), blin_x=rng.uniform(0.5, 1.5, 3), Alin_u=rng.standard_normal((2, nu)), blin_u=rng.uniform(0.3, 0.8, 2))
    settings = dict(max_iter=60, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    s = make(pkg, prob, settings, batch=batch)
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.1, 1.0, batch)[None, :]
    checked = sorted({0, batch // 2, batch - 1})
    orc = {b: oracle(prob, settings) for b in checked}
    for rnd in range(2):
        xs = x0s * (1.0 - 0.3 * rnd)
        s.set_x0_batch(xs)
        s.solve()
        # (round 5: from 16 instances up the families ride layout D's wide kernels, streamed from HBM -- tinympc_solve_dwide.h; smaller
        # batches stay on k_admm_solve_fam, layout A)
        assert s.launch_info()["layout"] == ("D" if batch >= 16 else "A"), s.jit_info()
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in checked:
            orc[b].set_x0(xs[:, b])
            orc[b].solve()
            assert st["iter"][b] == orc[b].stats()["iter"] and st["status"][b] == orc[b].stats()["status"], (rnd, b)
            assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL and rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, (rnd, b)
    s.reset()


@pytest.mark.layouts("A")
@pytest.mark.parametrize("nx,nu,N,batch", [(48, 14, 6, 70), (20, 4, 7, 16), (40, 10, 8, 33), (24, 8, 21, 40)])
def test_families_on_wide_systems_with_bounds_that_vary_over_the_horizon(pkg, kernel_layout, monkeypatch, nx, nu, N, batch):
    """The streamed-families kernels with the per-knot tables in LDS (!CT). (48, 14, 6) is the specialisation in which round 5's fuzzer
    caught a register used while an asm-issued LDS read was still filling it (profiles/r05_inflight_bug.txt: the next step's bound,
    parked in an AGPR before it had arrived; relative errors of order 10) -- a kernel's code depends on (nx, nu, N, tables, families)
    only, so any data of that shape runs the code that failed. Against the restatement over a cold and a warm start."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    P = pkg.problems
    rng = np.random.default_rng(nx * 1000 + nu * 10 + N)
    A = 0.9 * np.eye(nx) + (0.15 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    Bm = 0.3 * rng.standard_normal((nx, nu))
    prob = P.Problem("widefam_knots", A, Bm, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 1.2, rng.standard_normal(nx))
    prob.fdyn = 0.01 * rng.standard_normal(nx)
    prob.x_min = np.repeat(np.full((nx, 1), -3.0), N, 1) * rng.uniform(0.8, 1.0, (1, N))
    prob.x_max = -prob.x_min
    prob.u_min, prob.u_max = np.full(nu, -1.0), np.full(nu, 1.0)
    prob.cones = dict(Acx=[nx - 6, 1], qcx=[4, 2], cx=[1.3, 0.55], Acu=[0], qcu=[2], cu=[0.7])
    prob.linear = dict(Alin_x=rng.standard_normal((1, nx)), blin_x=rng.uniform(0.5, 1.5, 1), Alin_u=rng.standard_normal((2, nu)), blin_u=rng.uniform(0.3, 0.8, 2))
    settings = dict(max_iter=60, abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=3)
    s = make(pkg, prob, settings, batch=batch)
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.1, 1.0, batch)[None, :]
    checked = sorted({0, batch // 2, batch - 1})
    orc = {b: oracle(prob, settings) for b in checked}
    for rnd in range(2):
        xs = x0s * (1.0 - 0.3 * rnd)
        s.set_x0_batch(xs)
        s.solve()
        assert s.launch_info()["layout"] == "D", s.jit_info()
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in checked:
            orc[b].set_x0(xs[:, b])
            orc[b].solve()
            assert st["iter"][b] == orc[b].stats()["iter"] and st["status"][b] == orc[b].stats()["status"], (rnd, b)
            assert rel_err(sol["states"][:, :, b], orc[b].solution()[0]) < TOL and rel_err(sol["controls"][:, :, b], orc[b].solution()[1]) < TOL, (rnd, b)
    s.reset()


@pytest.mark.layouts("A")
def test_rocket_batch_properties_at_bench_size(pkg, kernel_layout, monkeypatch):
    """BASELINE config 4 in the batch the bench times (4,096 rocket landings, N=100, cones + linear row + fdyn, layout E): properties
    that need no oracle -- reversing the instance order reverses the results bit for bit; equal initial states give equal results
    wherever they sit in the batch; the thrust cone and the box hold on every converged instance; a seeded sample against the
    restatement."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    rk = pkg.problems.rocket(100)
    settings = dict(max_iter=150, abs_pri_tol=1e-2, abs_dua_tol=1e-2)
    B = 4096
    rng = np.random.default_rng(11)
    x0s = rk.x0[:, None] * rng.uniform(0.6, 1.2, (1, B)) + 0.05 * rng.standard_normal((6, B))
    x0s[:, 1000] = x0s[:, 7]      # duplicates far apart (different wavefronts, different workgroups)
    x0s[:, 4095] = x0s[:, 7]
    s = make(pkg, rk, settings, batch=B)
    s.set_x0_batch(x0s)
    s.solve()
    assert s.launch_info()["layout"] == "E", s.jit_info()
    sol, st = s.get_solution_batch(), s.get_stats_batch()
    for b in (1000, 4095):
        np.testing.assert_array_equal(sol["controls"][:, :, b], sol["controls"][:, :, 7])
        assert st["iter"][b] == st["iter"][7]
    conv = st["status"] == 1
    assert conv.sum() > 500  # (a third of the landings converge within 150 iterations at this tolerance)
    u = sol["controls"][:, :, conv]
    assert np.max(np.linalg.norm(u[:2], axis=0) - 0.25 * u[2]) < 5e-2  # the thrust cone, up to the ADMM tolerance
    assert np.all(sol["controls"] >= rk.u_min[:, None, None] - 1e-12) and np.all(sol["controls"] <= rk.u_max[:, None, None] + 1e-12)
    sample = sorted(rng.choice(B, size=6, replace=False))
    for b in sample:
        o = oracle(rk, settings)
        o.set_x0(x0s[:, b])
        o.solve()
        assert st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"], b
        assert rel_err(sol["controls"][:, :, b], o.solution()[1]) < TOL, b
    s.reset_workspace()
    s.set_x0_batch(np.ascontiguousarray(x0s[:, ::-1]))
    s.solve()
    sol_r, st_r = s.get_solution_batch(), s.get_stats_batch()
    np.testing.assert_array_equal(sol_r["controls"][:, :, ::-1], sol["controls"])
    np.testing.assert_array_equal(sol_r["states"][:, :, ::-1], sol["states"])
    np.testing.assert_array_equal(st_r["iter"][::-1], st["iter"])
    s.reset()
