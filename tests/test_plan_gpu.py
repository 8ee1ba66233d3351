"""The kernel of a launch is chosen in ONE function (tinympc_plan.hip: current_plan) from the shape, the batch size and the variant
of the configuration. This table enumerates (problem, batch, variant, environment) -> expected layout, origin of the kernel
(compiled in / run-time specialised) and grid, and checks that a solve on the chosen kernel matches the restatement."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.gpu


def _wide(P, nx, nu, N, seed=0):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) * (0.98 if nx > 64 else 1.0) + (0.015 if nx > 64 else 0.03) * rng.standard_normal((nx, nx))
    B = (0.08 if nx > 64 else 0.1) * rng.standard_normal((nx, nu))
    p = P.Problem("synthetic", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    p.x_min, p.x_max, p.u_min, p.u_max = np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3)
    return p


# (name, problem factory, batch, variant, env, expected layout, expected origin prefix(es), expected workgroups or None)
CASES = [
    # the box path of small batches: layout F (round 4) -- BASELINE configs 2 and 3 compiled in, any other shape specialised because the
    # test calls prepare() (test_box_path_of_small_batches_* below: without it such a shape stays on layout C)
    ("quadrotor50 single", lambda P: P.quadrotor(50), 1, "box", {}, "F", ("compiled-in",), 1),
    ("quadrotor50 x512", lambda P: P.quadrotor(50), 512, "box", {}, "F", ("compiled-in",), 512),
    ("quadrotor50 single per-knot references", lambda P: P.quadrotor(50), 1, "varying", {}, "F", ("compiled-in",), 1),
    ("cartpole20 single", lambda P: P.cartpole(20), 1, "box", {}, "F", ("compiled-in",), 1),
    ("quadrotor50 single no specialiser", lambda P: P.quadrotor(50), 1, "box", {"TINYMPC_JIT": "0"}, "C", ("compiled-in",), 1),
    ("quadrotor23 single", lambda P: P.quadrotor(23), 1, "box", {}, "F", ("compiled ", "disk-cache"), 1),
    ("quadrotor50 x2048", lambda P: P.quadrotor(50), 2048, "box", {}, "D", ("compiled-in",), 128),
    ("quadrotor40 x2048", lambda P: P.quadrotor(40), 2048, "box", {}, "D", ("compiled ", "disk-cache"), 128),
    ("quadrotor40 x2048 per-knot references", lambda P: P.quadrotor(40), 2048, "varying", {}, "D", ("compiled", "disk-cache"), None),
    ("quadrotor40 x2048 no specialiser", lambda P: P.quadrotor(40), 2048, "box", {"TINYMPC_JIT": "0"}, "B", ("refused(TINYMPC_JIT=0)",), 128),
    ("quadrotor200 x2048", lambda P: P.quadrotor(200), 2048, "box", {}, "E", ("compiled", "disk-cache"), 512),
    ("cartpole250 x2048", lambda P: P.cartpole(250, True), 2048, "box", {}, "E", ("compiled", "disk-cache"), 512),
    ("quadrotor50 x8192 adaptive rho", lambda P: P.quadrotor(50), 8192, "adaptive", {}, "D", ("compiled", "disk-cache"), None),
    ("quadrotor50 single adaptive rho", lambda P: P.quadrotor(50), 1, "adaptive", {}, "A", ("compiled-in",), 1),
    # BASELINE config 4 is COMPILED IN (round 4: __graft_entry__.HIP_BUILTINS -- layout F for one instance / small batches, layout E for
    # batches); the same configuration with the compiled-in code switched off, or any other structure, is specialised at run time
    ("rocket100 single", lambda P: P.rocket(100), 1, "families", {}, "F", ("compiled-in",), 1),
    ("rocket100 x200", lambda P: P.rocket(100), 200, "families", {}, "F", ("compiled-in",), 200),
    ("rocket100 x4096", lambda P: P.rocket(100), 4096, "families", {}, "E", ("compiled-in",), 1024),
    ("rocket100 x4096 run-time compiled", lambda P: P.rocket(100), 4096, "families", {"TINYMPC_BUILTIN": "0"}, "E", ("compiled ", "disk-cache"), 1024),
    ("rocket100 single run-time compiled", lambda P: P.rocket(100), 1, "families", {"TINYMPC_BUILTIN": "0"}, "F", ("compiled ", "disk-cache"), 1),
    ("rocket44 x4096", lambda P: P.rocket(44), 4096, "families", {}, "E", ("compiled ", "disk-cache"), 1024),
    ("rocket10 x4096", lambda P: P.rocket(10), 4096, "families", {}, "E", ("compiled ", "disk-cache"), 256),  # (the uncut form: four groups per workgroup)
    ("rocket10 x4096 on layout D", lambda P: P.rocket(10), 4096, "families", {"TINYMPC_LAYOUT": "D"}, "D", ("compiled ", "disk-cache"), None),
    ("rocket100 single no specialiser", lambda P: P.rocket(100), 1, "families", {"TINYMPC_JIT": "0"}, "C", ("refused(TINYMPC_JIT=0)",), 1),
    ("rocket100 x4096 no specialiser", lambda P: P.rocket(100), 4096, "families", {"TINYMPC_JIT": "0"}, "C", ("refused(TINYMPC_JIT=0)",), 4096),
    ("rocket100 overlapping cones single", lambda P: P.rocket(100), 1, "overlap", {}, "F", ("compiled", "disk-cache"), 1),
    ("rocket100 overlapping cones single no specialiser", lambda P: P.rocket(100), 1, "overlap", {"TINYMPC_JIT": "0"}, "A", ("refused(TINYMPC_JIT=0)",), 1),
    ("wide 24+8 x4096", lambda P: _wide(P, 24, 8, 30), 4096, "box", {}, "D", ("compiled-in",), None),
    ("large 96+32 x64", lambda P: _wide(P, 96, 32, 12, 96), 64, "box", {}, "M", ("compiled-in",), 4),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_plan_table(pkg, monkeypatch, case):
    name, factory, batch, variant, env, layout, origins, workgroups = case
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P = pkg.problems
    prob = factory(P)
    settings = dict(max_iter=25, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
    s = pkg.TinyMPC()
    extra = dict(adaptive_rho=True, adaptive_rho_min=0.2, adaptive_rho_max=40.0) if variant == "adaptive" else {}
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=getattr(prob, "fdyn", None), **settings, **extra)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if variant in ("families", "overlap"):
        if variant == "overlap":
            prob.cones = dict(Acx=[0, 1], qcx=[3, 4], cx=[0.2, 0.3], Acu=[0], qcu=[3], cu=[0.25])
        s.set_cone_constraints(**prob.cones)
        s.set_linear_constraints(**prob.linear)
        s.set_x_ref(prob.x_ref)
        s.set_u_ref(prob.u_ref)
    if variant == "varying":
        prob.x_ref = 0.05 * np.sin(np.arange(prob.N))[None, :] * np.ones((prob.nx, 1))
        s.set_x_ref(prob.x_ref)
    if variant == "adaptive":
        dK, dP, dC1, dC2 = s.compute_sensitivity_autograd()
        s.set_sensitivity_matrices(dK, dP, dC1, dC2)
    rng = np.random.default_rng(7)
    x0s = prob.x0[:, None] * rng.uniform(0.8, 1.1, (1, batch))
    if batch == 1:
        s.set_x0(x0s[:, 0])
    else:
        s.set_x0_batch(x0s)
    s.prepare()
    info, origin = s.launch_info(), s.jit_info()
    assert info["layout"] == layout, (name, info, origin)
    assert origin.startswith(origins), (name, origin)
    if workgroups is not None:
        assert info["workgroups"] == workgroups, (name, info)
    s.solve()
    assert s.launch_info()["layout"] == layout
    orc = O.OraclePort(prob).load_problem(prob, settings)  # (the checker takes the families from `prob` when it is constructed)
    if variant == "adaptive":
        orc.set_adaptive_rho(True, 0.2, 40.0, True)
        orc.set_sensitivity(dK, dP)
    b = batch - 1
    orc.set_x0(x0s[:, b])
    orc.solve()
    sol = s.get_solution_batch(b, 1)
    st = s.get_stats_batch(b, 1)
    assert st["iter"][0] == orc.stats()["iter"], name
    assert rel_err(sol["controls"][:, :, 0], orc.solution()[1]) < 1e-9, name
    s.reset()


def test_box_path_of_small_batches_takes_layout_f_only_where_it_costs_nothing_or_was_asked_for(pkg, monkeypatch):
    """decide_layout_f: the compiled-in configurations run on layout F from the first solve; another shape stays on the latency kernel
    (no seconds of run-time compilation behind a setup call) until the caller asks for the specialised kernels with prepare(); results
    agree either way, and the persistent state carries over from one kernel to the other."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    P = pkg.problems
    settings = dict(max_iter=40, abs_pri_tol=1e-4, abs_dua_tol=1e-4)

    def fresh(prob):
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, **settings)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x0(prob.x0)
        return s

    for prob in (P.quadrotor(50), P.cartpole(20)):
        s = fresh(prob)
        s.solve()
        assert s.launch_info()["layout"] == "F" and s.jit_info().startswith("compiled-in"), (prob.name, s.launch_info(), s.jit_info())
        s.reset()
    prob = P.quadrotor(31)  # (no compiled-in kernel)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    s = fresh(prob)
    for rnd in range(4):
        if rnd == 2:
            s.prepare()
        x0 = prob.x0 * (1.0 - 0.2 * rnd)
        s.set_x0(x0)
        s.solve()
        assert s.launch_info()["layout"] == ("C" if rnd < 2 else "F"), (rnd, s.launch_info(), s.jit_info())
        orc.set_x0(x0)
        orc.solve()
        assert s.get_stats()["iter"] == orc.stats()["iter"], rnd
        assert rel_err(s.get_solution()["controls"], orc.solution()[1]) < 1e-9, rnd
    s.reset()
    monkeypatch.setenv("TINYMPC_BUILTIN", "0")  # the compiled-in code switched off: nothing is free any more
    s = fresh(P.quadrotor(50))
    s.solve()
    assert s.launch_info()["layout"] == "C"
    s.reset()


@pytest.mark.parametrize("name,N", [("quadrotor", 8), ("quadrotor", 10), ("quadrotor", 13), ("cartpole", 10), ("quadrotor", 14)])
def test_short_horizons_run_on_one_wavefront_of_layout_f(pkg, monkeypatch, name, N):
    """plan_f (round 4): up to four chunks of at most three slots fit the DPP rows of ONE wavefront -- the carry scan stays inside it, the
    iteration has no barrier (N <= 13); from four slots per chunk on it is two wavefronts again (N = 14). Launched solves against the
    oracle (cold + warm), then a resident session against launched ticks of a twin handle, bit for bit."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    P = pkg.problems
    prob = P.quadrotor(N) if name == "quadrotor" else P.cartpole(N, True)
    settings = dict(max_iter=60, abs_pri_tol=1e-3, abs_dua_tol=1e-3)

    def handle():
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, **settings)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x0(prob.x0)
        s.prepare()
        return s

    a, b = handle(), handle()
    info = a.jit_info()
    assert a.launch_info()["layout"] == "F" and ("wpg=1 " in info) == (N <= 13), info
    orc = O.OraclePort(prob).load_problem(prob, settings)
    for rnd in range(3):
        x0 = prob.x0 * (1.0 - 0.25 * rnd)
        a.set_x0(x0)
        a.solve()
        orc.set_x0(x0)
        orc.solve()
        assert a.get_stats()["iter"] == orc.stats()["iter"] and a.get_stats()["status"] == orc.stats()["status"], rnd
        assert rel_err(a.get_solution()["states"], orc.solution()[0]) < 1e-9 and rel_err(a.get_solution()["controls"], orc.solution()[1]) < 1e-9, rnd
    a.reset_workspace()
    b.reset_workspace()
    a.session_begin()
    assert a.launch_info()["layout"] == "F"
    x = prob.x0.copy()
    for k in range(15):
        ua = a.session_step(x)
        ub = b.mpc_step(x)[:, 0]
        np.testing.assert_array_equal(ua, ub)
        assert a.get_stats()["iter"] == b.get_stats()["iter"], k
        x = prob.A @ x + prob.B @ ua
    a.session_end()
    a.reset()
    b.reset()
