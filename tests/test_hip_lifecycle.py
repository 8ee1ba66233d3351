"""GPU tests of handle lifecycle and verb ordering: the reference keeps ONE process-global solver
(bindings.cpp:17); the C ABI keeps independent handles. These tests pin the behaviours callers rely on."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _new(pkg, prob, batch=1, **settings):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, **settings)
    return s


def test_independent_handles_do_not_share_state(pkg):
    """Two solvers alive at once, solved alternately (in the reference two TinyMPC objects silently share the
    global solver, SURVEY.md section 8b 'Ownership'; here they must not)."""
    P = pkg.problems
    cp, qd = P.cartpole(20, True), P.quadrotor(20)
    a, b = _new(pkg, cp, max_iter=30), _new(pkg, qd, max_iter=30)
    a.set_bound_constraints(cp.x_min, cp.x_max, cp.u_min, cp.u_max)
    b.set_bound_constraints(qd.x_min, qd.x_max, qd.u_min, qd.u_max)
    oa = O.OraclePort(cp).load_problem(cp, dict(max_iter=30))
    ob = O.OraclePort(qd).load_problem(qd, dict(max_iter=30))
    a.set_x0(cp.x0)
    b.set_x0(qd.x0)
    for _ in range(3):
        a.solve(); b.solve(); oa.solve(); ob.solve()
        assert rel_err(a.get_solution()["controls"], oa.solution()[1]) < TOL
        assert rel_err(b.get_solution()["controls"], ob.solution()[1]) < TOL
    a.reset()
    b.solve()  # b survives a's destruction
    ob.solve()
    assert rel_err(b.get_solution()["controls"], ob.solution()[1]) < TOL
    b.reset()


def test_resetup_replaces_the_problem(pkg):
    """setup() on a live object replaces the solver, as the reference's g_solver.reset(ptr) does (bindings.cpp:92)."""
    P = pkg.problems
    s = _new(pkg, P.cartpole(20, True), max_iter=20)
    s.set_x0(P.cartpole().x0)
    s.solve()
    qd = P.quadrotor(15)
    s.setup(qd.A, qd.B, qd.Q, qd.R, qd.N, rho=qd.rho, max_iter=25)
    s.set_bound_constraints(qd.x_min, qd.x_max, qd.u_min, qd.u_max)
    s.set_x0(qd.x0)
    s.solve()
    o = O.OraclePort(qd).load_problem(qd, dict(max_iter=25))
    o.solve()
    assert s.get_stats()["iter"] == o.stats()["iter"]
    assert rel_err(s.get_solution()["states"], o.solution()[0]) < TOL
    s.reset()
    with pytest.raises(pkg.TinyMPCError):
        s.solve()  # TinyMPC:NotSetup after reset (TinyMPC.m:329-334)


def test_changing_bounds_refs_and_settings_between_solves(pkg):
    """Bounds, references and tolerances may change between warm-started solves (the closed-loop examples do)."""
    P = pkg.problems
    rng = np.random.default_rng(11)
    qd = P.quadrotor(20)
    st = dict(max_iter=40, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
    s = _new(pkg, qd, **st)
    o = O.OraclePort(qd).load_problem(qd, st)
    s.set_bound_constraints(qd.x_min, qd.x_max, qd.u_min, qd.u_max)
    s.set_x0(qd.x0)
    for k in range(5):
        if k == 1:
            xr, ur = 0.1 * rng.standard_normal((12, 20)), 0.02 * rng.standard_normal((4, 19))
            s.set_x_ref(xr); s.set_u_ref(ur); o.set_x_ref(xr); o.set_u_ref(ur)
        if k == 2:
            b = (np.full((12, 20), -3.0), np.full((12, 20), 3.0), np.full((4, 19), -0.3), np.full((4, 19), 0.45))
            s.set_bound_constraints(*b); o.set_bound_constraints(*b)
        if k == 3:
            s.update_settings(max_iter=15, abs_pri_tol=1e-2, check_termination=3)
            o.update_settings(max_iter=15, abs_pri_tol=1e-2, check_termination=3)
        if k == 4:
            s.update_settings(en_input_bound=False); o.update_settings(en_input_bound=0)
        s.solve(); o.solve()
        assert s.get_stats()["iter"] == o.stats()["iter"], k
        assert rel_err(s.get_solution()["controls"], o.solution()[1]) < TOL, k
        assert rel_err(s.get_solution()["states"], o.solution()[0]) < TOL, k
    s.reset()


def test_many_handles_created_and_destroyed(pkg):
    """No device-memory leak across handle lifetimes: 1,000 handles of 0.35 GB each (350 GB in total, more
    than the 288 GB of HBM) are created, solved once and destroyed; a leak would surface as a clean
    TINYMPC_ERR_ALLOC from hipMalloc long before the end."""
    P = pkg.problems
    qd = P.quadrotor(50)
    for _ in range(1000):
        s = _new(pkg, qd, batch=16384, max_iter=1)
        s.solve()
        s.reset()


def test_scalar_and_vector_argument_expansion_through_the_boundary(pkg):
    """Scalars / single vectors are broadcast on the caller side exactly as TinyMPC.m does (:378-405)."""
    P = pkg.problems
    qd = P.quadrotor(12)
    a = _new(pkg, qd, max_iter=20)
    b = _new(pkg, qd, max_iter=20)
    a.set_bound_constraints(-5.0, 5.0, np.full(4, -0.5), np.full((1, 4), 0.5))     # scalar / vector / row vector
    b.set_bound_constraints(np.full((12, 12), -5.0), np.full((12, 12), 5.0), np.full((4, 11), -0.5), np.full((4, 11), 0.5))
    a.set_x_ref(np.full(12, 0.05)); b.set_x_ref(np.full((12, 12), 0.05))
    a.set_u_ref(0.01); b.set_u_ref(np.full((4, 11), 0.01))
    for s in (a, b):
        s.set_x0(qd.x0)
        s.solve()
    np.testing.assert_array_equal(a.get_solution()["controls"], b.get_solution()["controls"])
    np.testing.assert_array_equal(a.get_solution()["states"], b.get_solution()["states"])
    a.reset(); b.reset()


def test_single_instance_host_path_states(pkg):
    """Single-instance handles serve set_x0 / get_solution / get_stats from pinned host memory (no device copies around
    the launch). Every ordering of the verbs must still read what the device would have returned."""
    P = pkg.problems
    prob = P.quadrotor(20)
    settings = dict(max_iter=40, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
    s = _new(pkg, prob, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    o = O.OraclePort(prob).load_problem(prob, settings)
    # x0 set twice before the solve: the last one counts; nothing solved yet -> zeros
    s.set_x0(0.3 * prob.x0)
    s.set_x0(prob.x0)
    assert not s.get_solution()["controls"].any() and s.get_stats()["iter"] == 0
    o.set_x0(prob.x0)
    for tick in range(3):
        s.solve_async()          # asynchronous launch: the getters must wait for it
        sol, st = s.get_solution(), s.get_stats()
        o.solve()
        assert st["iter"] == o.stats()["iter"] and st["status"] == o.stats()["status"]
        assert rel_err(sol["states"], o.solution()[0]) < TOL and rel_err(sol["controls"], o.solution()[1]) < TOL
        np.testing.assert_array_equal(s.get_solution_batch()["controls"][:, :, 0], sol["controls"])
        np.testing.assert_array_equal(s.get_first_controls_batch()[:, 0], sol["controls"][:, 0])  # device copy agrees
        x = prob.A @ (prob.x0 * (1 - 0.2 * tick)) + prob.B @ sol["controls"][:, 0]
        s.set_x0(x)
        o.set_x0(x)
    # a 0-iteration solve changes nothing; mpc_step and set_x0_batch override a pending host x0
    before = s.get_solution()["controls"].copy()
    s.update_settings(max_iter=0)
    s.solve()
    np.testing.assert_array_equal(s.get_solution()["controls"], before)
    s.update_settings(max_iter=40)
    s.set_x0(5.0 * prob.x0)                       # pending on the host ...
    s.set_x0_batch(prob.x0[:, None])               # ... replaced through the batch verb
    o.set_x0(prob.x0)
    s.solve()
    o.solve()
    assert rel_err(s.get_solution()["controls"], o.solution()[1]) < TOL
    s.set_x0(5.0 * prob.x0)
    u = s.mpc_step(prob.x0[:, None])               # brings its own x0
    o.solve()
    assert rel_err(u[:, 0], o.solution()[1][:, 0]) < TOL
    assert rel_err(s.get_solution()["controls"], o.solution()[1]) < TOL
    s.reset_workspace()                            # cold start: solution and statistics read as zero again
    assert not s.get_solution()["states"].any() and s.get_stats()["iter"] == 0
    s.reset()


@pytest.mark.gpu
def test_host_threads_share_the_device(pkg):
    """tools/thread_stress.py in small: three host threads cycle through setup -> solve -> reset on handles of their own (pooled streams and
    arenas change hands, specialisations are looked up concurrently) while a fourth keeps a resident session ticking, parked by every setup
    of the others and restarted by its next step. Round 5 found a reader of the session's solution waiting for the sequence number of
    another thread's park command (wait_session_solution); every result must equal the single-threaded one bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "thread_stress.py"), "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "0 error(s)" in out.stdout
    # ... and several sessions at once, each on its own host thread, parking one another with every setup. With ROCm's default of four
    # hardware queues per process the launched twins queued behind resident kernels (2 s per tick); the library's default for
    # GPU_MAX_HW_QUEUES (tinympc_handle.hip) is what this checks. The tool dumps every thread's stack and exits after 45 s of stall.
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "thread_sessions.py"), "3", "3", "80"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "0 error(s)" in out.stdout, out.stdout[-1500:] + out.stderr[-2500:]
