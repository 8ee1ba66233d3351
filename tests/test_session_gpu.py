"""Closed-loop session (tinympc_session_begin / _step / _end): the latency kernel stays resident and takes its ticks from
a mailbox in pinned host memory. A session's ticks must equal, bit for bit, the same ticks issued as ordinary launches
(tinympc_mpc_step_batch) -- same warm-start state, same stale-v semantics after converged solves (admm.cpp:181-197),
same per-tick references -- and the handle must continue with ordinary solves afterwards."""
from __future__ import annotations

import time

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.gpu


def _solver(pkg, prob, settings, families=False):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref)
    if prob.u_ref is not None:
        s.set_u_ref(prob.u_ref)
    if families:
        s.set_cone_constraints(**prob.cones)
        s.set_linear_constraints(**prob.linear)
    return s


@pytest.mark.parametrize("mailbox", ["device", "host"])
@pytest.mark.parametrize("name", ["cartpole", "quadrotor"])
def test_session_ticks_equal_launched_ticks(pkg, monkeypatch, name, mailbox):
    """(mailbox: in device memory behind the PCIe BAR where the device maps it -- round 5's default --, or in pinned host memory, the
    fall-back for devices that do not and TINYMPC_MAILBOX=host: the same protocol over another path, the kernel polling across PCIe)"""
    if mailbox == "host":
        monkeypatch.setenv("TINYMPC_MAILBOX", "host")  # (read when a handle's arenas are acquired: tinympc_setup)
    P = pkg.problems
    prob = P.cartpole(10, True) if name == "cartpole" else P.quadrotor(50)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=80)
    a, b = _solver(pkg, prob, settings), _solver(pkg, prob, settings)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    a.session_begin()
    # (BASELINE config 3 is compiled in on layout F, resident kernel included; the cartpole N=10 has no specialisation: layout C)
    assert a.launch_info()["layout"] == ("F" if name == "quadrotor" else "C")
    xa, xb = prob.x0.copy(), prob.x0.copy()
    for k in range(25):
        ua = a.session_step(xa)
        ub = b.mpc_step(xb)[:, 0]
        orc.set_x0(xa)
        orc.solve()
        np.testing.assert_array_equal(ua, ub)
        assert a.get_stats()["iter"] == b.get_stats()["iter"] == orc.stats()["iter"], k
        np.testing.assert_array_equal(a.get_solution()["states"], b.get_solution()["states"])
        assert rel_err(a.get_solution()["controls"], orc.solution()[1]) < 1e-9
        xa = prob.A @ xa + prob.B @ ua
        xb = prob.A @ xb + prob.B @ ub
    assert b.launch_info()["layout"] == ("F" if name == "quadrotor" else "C")
    a.session_end()
    # the handle goes on with ordinary launches from the session's ADMM state
    ua, ub = a.mpc_step(xa)[:, 0], b.mpc_step(xb)[:, 0]
    np.testing.assert_array_equal(ua, ub)
    assert a.get_stats()["iter"] == b.get_stats()["iter"]
    a.reset()
    b.reset()


@pytest.mark.parametrize("layout,N", [("C", 20), ("F", 10), ("F", 20), ("F", 44)])
def test_session_with_per_tick_references_and_families(pkg, monkeypatch, layout, N):
    """rocket_landing_constraints.m:86-121 inside a session: references re-sent every tick, cones + linear row + fdyn. The session
    is the resident variant of the kernel the handle's launches run on, so its ticks are bit-identical to LAUNCHED ticks: layout F by
    default (round 4; N = 10: one wavefront, 20 / 44: the element form of the families on three / four wavefronts), the latency kernel of
    layout C with TINYMPC_LAYOUT=C."""
    if layout == "C":
        monkeypatch.setenv("TINYMPC_LAYOUT", "C")
    else:
        monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    P = pkg.problems
    prob = P.rocket(N)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60)
    a, b = _solver(pkg, prob, settings, True), _solver(pkg, prob, settings, True)
    a.session_begin()
    assert a.launch_info()["layout"] == layout
    x = prob.x0.copy()
    goal = np.zeros(prob.nx)
    for k in range(12):
        x_ref = np.stack([prob.x0 + (goal - prob.x0) * min(i + k, 99) / 99 for i in range(prob.N)], axis=1)
        for h in (a, b):
            h.set_x_ref(x_ref)
            h.set_u_ref(prob.u_ref)
        ua = a.session_step(x)
        ub = b.mpc_step(x)[:, 0]
        np.testing.assert_array_equal(ua, ub)
        assert a.get_stats()["iter"] == b.get_stats()["iter"], k
        x = prob.A @ x + prob.B @ ua + prob.fdyn
    assert b.launch_info()["layout"] == layout
    a.session_end()
    a.reset()
    b.reset()


def test_session_is_ended_by_other_verbs_and_survives_idling(pkg):
    P = pkg.problems
    prob = P.quadrotor(20)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=40)
    a, b = _solver(pkg, prob, settings), _solver(pkg, prob, settings)
    x = prob.x0.copy()
    a.session_begin()
    np.testing.assert_array_equal(a.session_step(x), b.mpc_step(x)[:, 0])
    # a verb that needs the device ends the session implicitly; the handle keeps working
    a.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min * 0.8, prob.u_max * 0.8)
    b.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min * 0.8, prob.u_max * 0.8)
    with pytest.raises(pkg.TinyMPCError):
        a.session_step(x)
    np.testing.assert_array_equal(a.mpc_step(x)[:, 0], b.mpc_step(x)[:, 0])
    # idle longer than the resident kernel's time-out (2 s): the next step restarts it transparently
    a.session_begin()
    np.testing.assert_array_equal(a.session_step(x), b.mpc_step(x)[:, 0])
    time.sleep(2.6)
    x2 = 0.9 * x
    np.testing.assert_array_equal(a.session_step(x2), b.mpc_step(x2)[:, 0])
    a.session_end()
    a.session_end()  # ending twice is harmless
    a.reset()
    b.reset()


@pytest.mark.parametrize("how", ["parked", "idle"])
def test_reference_jump_after_the_resident_kernel_left(pkg, how):
    """A layout-F session (quadrotor N=50: compiled in) whose resident kernel has gone home -- parked by another handle's setup, or
    after its 2 s idle time-out -- and then receives a NEW reference (a jump, no shift before it): the restart must hand the pending
    full re-read to the new kernel. Layout F's session kernel stages nothing itself; round 4 re-issued the command without the
    reference flag and ran the tick on the old tables (advisor finding, tinympc_session.hip)."""
    P = pkg.problems
    prob = P.quadrotor(50)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60)
    a, b = _solver(pkg, prob, settings), _solver(pkg, prob, settings)
    a.session_begin()
    assert a.launch_info()["layout"] == "F"
    x = prob.x0.copy()
    for _ in range(2):
        ua, ub = a.session_step(x), b.mpc_step(x)[:, 0]
        np.testing.assert_array_equal(ua, ub)
        x = prob.A @ x + prob.B @ ua
    if how == "parked":
        other = _solver(pkg, prob, settings)  # its setup sends the device's resident kernels home
        other.reset()
    else:
        time.sleep(2.6)
    goal = np.array([0.8, -0.6, 0.5, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    x_ref = np.repeat(goal[:, None], prob.N, axis=1)
    x_ref[:, ::2] *= 0.9  # (varies over the horizon: not a constant-table case)
    for h in (a, b):
        h.set_x_ref(x_ref)
    for k in range(3):
        ua, ub = a.session_step(x), b.mpc_step(x)[:, 0]
        np.testing.assert_array_equal(ua, ub, err_msg=f"{how}: tick {k} after the jump")
        assert a.get_stats()["iter"] == b.get_stats()["iter"]
        x = prob.A @ x + prob.B @ ua
    a.session_end()
    np.testing.assert_array_equal(a.mpc_step(x)[:, 0], b.mpc_step(x)[:, 0])
    a.reset()
    b.reset()


def test_receding_horizon_references_travel_as_one_column(pkg):
    """rocket_landing_constraints.m:96-101: the reference re-sent at tick k+1 is the one of tick k moved up by one knot.
    Inside a session the library detects that (bit for bit) and ships only the new last column with the command; the
    result must equal the full re-read (ordinary launches on a second handle), including a tick where the reference is
    NOT a shift, a tick where it is sent twice, and ordinary solves after the session."""
    P = pkg.problems
    prob = P.quadrotor(20)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60)
    a, b = _solver(pkg, prob, settings), _solver(pkg, prob, settings)
    goal = np.array([1.0, -0.5, 0.8, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    ref_at = lambda j: prob.x0 + (goal - prob.x0) * min(j, 60) / 60
    uref_at = lambda j: np.full(prob.nu, 0.002 * j)
    a.session_begin()
    x = prob.x0.copy()
    for k in range(14):
        off = k if k != 7 else 30  # tick 7: a jump, not a shift
        x_ref = np.stack([ref_at(i + off) for i in range(prob.N)], axis=1)
        u_ref = np.stack([uref_at(i + off) for i in range(prob.N - 1)], axis=1)
        for h in (a, b):
            h.set_x_ref(x_ref)
            h.set_u_ref(u_ref)
            if k == 3:
                h.set_x_ref(x_ref)  # sent twice: unchanged the second time
        ua = a.session_step(x)
        ub = b.mpc_step(x)[:, 0]
        np.testing.assert_array_equal(ua, ub, err_msg=f"tick {k}")
        np.testing.assert_array_equal(a.get_solution()["states"], b.get_solution()["states"])
        assert a.get_stats()["iter"] == b.get_stats()["iter"], k
        x = prob.A @ x + prob.B @ ua
    a.session_end()
    np.testing.assert_array_equal(a.mpc_step(x)[:, 0], b.mpc_step(x)[:, 0])  # device copies / tables were restaged
    a.reset()
    b.reset()


def test_setup_and_reset_of_another_handle_do_not_wait_for_the_resident_kernel(pkg):
    """hipMalloc / hipFree synchronise the device; with a resident session kernel spinning they used to block until its idle
    time-out (2 s). The library now sends the device's resident kernels home first; the session stays open, the next step
    restarts the kernel and continues from the same ADMM state: the controls equal those of an undisturbed session."""
    import time
    prob = pkg.problems.quadrotor(20)
    settings = dict(max_iter=60, abs_pri_tol=1e-3, abs_dua_tol=1e-3)

    def handle():
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, **settings)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        return s

    a, ref = handle(), handle()
    a.session_begin()
    ref.session_begin()
    x, us, ur = prob.x0.copy(), [], []
    for k in range(12):
        if k == 5:  # another handle comes and goes while `a`'s kernel is resident
            t0 = time.perf_counter()
            ref.session_end()  # (the reference session steps aside itself: this test is about `a`)
            other = handle()
            other.set_x0(prob.x0)
            other.solve()
            other.reset()
            dt = time.perf_counter() - t0
            assert dt < 1.0, f"setup + solve + reset of another handle took {dt:.2f} s with a session open"
            ref.session_begin()
        ua, ub = a.session_step(x), ref.session_step(x)
        us.append(ua)
        ur.append(ub)
        x = prob.A @ x + prob.B @ ua
    a.session_end()
    ref.session_end()
    np.testing.assert_array_equal(np.array(us), np.array(ur))
    a.reset()
    ref.reset()


def test_rocket_landing_closed_loop_resident_at_the_full_horizon(pkg, monkeypatch):
    """BASELINE config 4 as the reference uses it (rocket_landing_constraints.m:86-121: a closed loop, references re-sent every
    tick) at N = 100: beyond what the latency kernel's session holds with the families (N <= 65), so the resident kernel is layout
    F's (round 4). Its ticks must equal LAUNCHED ticks of layout F bit for bit -- first controls, iteration counts, the whole
    solution -- through receding-horizon references that are a shift (one column travels), a jump (full re-read), a tick without a
    new reference, the idle time-out (the kernel is restarted transparently), and the handle must go on with ordinary launches from
    the session's state; every tick is also checked against the restatement."""
    monkeypatch.delenv("TINYMPC_LAYOUT", raising=False)
    P = pkg.problems
    prob = P.rocket(100)
    settings = dict(abs_pri_tol=5e-2, abs_dua_tol=5e-2, max_iter=200)  # (loose: the 100-knot landing converges slowly, and ticks must CONVERGE for the stale-slack roll-back to be exercised)
    a, b = _solver(pkg, prob, settings, True), _solver(pkg, prob, settings, True)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    goal = np.zeros(prob.nx)
    ref_at = lambda jk: prob.x0 + (goal - prob.x0) * min(jk, 140) / 140
    a.prepare()
    assert a.launch_info()["layout"] == "F" and a.jit_info().startswith(("compiled-in", "compiled ", "disk-cache")), a.jit_info()
    a.session_begin()  # (after jit_info: pushing settings ends a session)
    assert a.launch_info()["layout"] == "F"
    x = prob.x0.copy()
    iters = []
    for k in range(16):
        off = k if k != 6 else 40  # tick 6: a jump, not a shift
        if k != 9:                 # tick 9: no new reference at all
            x_ref = np.stack([ref_at(i + off) for i in range(prob.N)], axis=1)
            u_ref = prob.u_ref + 0.01 * (k if k != 6 else 3)  # (changes every tick, never a shift: the full re-read of the input side)
            for h in (a, b):
                h.set_x_ref(x_ref)
                if k % 2 == 0:
                    h.set_u_ref(u_ref)
            orc.set_x_ref(x_ref)
            if k % 2 == 0:
                orc.set_u_ref(u_ref)
        if k == 12:
            time.sleep(2.6)  # longer than the resident kernel's idle time-out: the next step restarts it
        ua = a.session_step(x)
        ub = b.mpc_step(x)[:, 0]
        orc.set_x0(x)
        orc.solve()
        np.testing.assert_array_equal(ua, ub, err_msg="tick %d" % k)
        assert a.get_stats()["iter"] == b.get_stats()["iter"] == orc.stats()["iter"], (k, a.get_stats(), b.get_stats(), orc.stats())
        np.testing.assert_array_equal(a.get_solution()["states"], b.get_solution()["states"])
        np.testing.assert_array_equal(a.get_solution()["controls"], b.get_solution()["controls"])
        assert rel_err(a.get_solution()["controls"], orc.solution()[1]) < 1e-9 and rel_err(a.get_solution()["states"], orc.solution()[0]) < 1e-9, k
        iters.append(a.get_stats()["iter"])
        x = prob.A @ x + prob.B @ ua + prob.fdyn
    assert len(set(iters)) >= 2  # (warm starts: the ticks do not all take the same number of iterations)
    a.session_end()
    ua, ub = a.mpc_step(x)[:, 0], b.mpc_step(x)[:, 0]  # ordinary launches from the session's ADMM state
    np.testing.assert_array_equal(ua, ub)
    assert a.get_stats()["iter"] == b.get_stats()["iter"]
    a.reset()
    b.reset()


def test_closed_loop_driven_from_c_equals_the_python_loop(pkg):
    """tinympc_bench_closed_loop (the measurement helper bench.py's c_loop numbers come from) runs the same ticks as a loop of
    mpc_step / session_step calls: same final state bit for bit, same iteration total, launched and resident."""
    P = pkg.problems
    prob = P.quadrotor(50)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
    for session in (False, True):
        a, b = _solver(pkg, prob, settings), _solver(pkg, prob, settings)
        if session:
            a.session_begin()
            b.session_begin()
        out = a.bench_closed_loop(prob.A, prob.B, prob.x0, 30, 5, session=session)
        x, its = prob.x0.copy(), 0
        for k in range(30):
            u0 = b.session_step(x) if session else b.mpc_step(x)[:, 0]
            if k >= 5:
                its += b.get_stats()["iter"]
            x = prob.A @ x + prob.B @ u0
        assert out["iterations_per_tick"] * 25 == its and out["us_per_tick"] > 0
        np.testing.assert_allclose(out["x"], x, rtol=1e-12, atol=1e-15)  # (the plant step's summation order differs from numpy's)
        if session:
            a.session_end()
            b.session_end()
        a.reset()
        b.reset()


@pytest.mark.parametrize("name", ["quadrotor", "cartpole10", "rocket"])
def test_resident_solves_are_the_reference_loop_on_the_resident_kernel(pkg, name):
    """tinympc_set_resident (round 5): the reference's per-tick sequence set_x0 -> solve -> get_solution / get_stats
    (examples/cartpole_example_mpc.m:36-44) served by the resident session kernel -- bit-identical to launched solves, through verbs
    that end the session (new bounds) and references that change, and back to launches when switched off."""
    P = pkg.problems
    prob = {"quadrotor": P.quadrotor(50), "cartpole10": P.cartpole(10, True), "rocket": P.rocket(20)}[name]
    fam = name == "rocket"
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=60)
    a, b = _solver(pkg, prob, settings, families=fam), _solver(pkg, prob, settings, families=fam)
    a.set_resident(True)
    x = prob.x0.copy()
    f = prob.fdyn if prob.fdyn is not None else 0.0
    for k in range(16):
        if k == 6:  # a verb that needs the device: the resident kernel goes home, the next solve starts it again
            for h in (a, b):
                h.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min * 0.9, prob.u_max * 0.9)
        if k == 9:
            x_ref = np.repeat((0.3 * prob.x0)[:, None], prob.N, axis=1)
            x_ref[:, ::2] *= 0.5
            for h in (a, b):
                h.set_x_ref(x_ref)
        if k == 12:
            a.set_resident(False)
        for h in (a, b):
            h.set_x0(x)
            h.solve()
        sa, sb = a.get_solution(), b.get_solution()
        np.testing.assert_array_equal(sa["controls"], sb["controls"], err_msg=f"tick {k}")
        np.testing.assert_array_equal(sa["states"], sb["states"])
        assert a.get_stats() == b.get_stats(), k
        x = prob.A @ x + prob.B @ sa["controls"][:, 0] + f
    a.reset()
    b.reset()
    # a batched handle has no resident kernel: the switch is accepted, solves are launched
    c = pkg.TinyMPC()
    c.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=8, rho=prob.rho, fdyn=prob.fdyn, **settings)
    c.set_resident(True)
    c.set_x0_batch(np.asfortranarray(np.repeat(prob.x0[:, None], 8, axis=1)))
    c.solve()
    assert c.get_stats_batch()["iter"].min() >= 1
    c.reset()


def test_early_answer_is_the_solutions_first_column_when_the_controls_saturate(pkg):
    """The tick's first controls travel ahead of the solution in a stamped 64-byte line (SolveParams::host_ans). Round 5's fuzzer caught one
    tick in ~3,000 returning zeros there while the solution itself was right: saturated controls (+b, +b, -b, -b) had the XOR checksum of an
    empty line, and the host read the line's words before its stamp (profiles/r05_session_stamp_bug.txt). A race cannot be pinned by a
    deterministic test; this one is the self-consistency check that would have shown it within a few thousand ticks: session_step's
    controls against the first column of get_solution, tick by tick, on a system whose inputs sit on their bounds with changing signs."""
    rng = np.random.default_rng(3)
    nx, nu, N = 10, 4, 12
    A = 0.9 * np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.3 * rng.standard_normal((nx, nu))
    prob = pkg.problems.Problem("saturating", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 1.5, rng.standard_normal(nx))
    prob.x_min, prob.x_max = np.full(nx, -50.0), np.full(nx, 50.0)
    prob.u_min, prob.u_max = np.full(nu, -0.4), np.full(nu, 0.4)
    s = _solver(pkg, prob, dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=6))
    s.prepare()
    s.session_begin()
    saturated, patterns = 0, set()
    for k in range(4000):
        x = 4.0 * rng.standard_normal(nx)
        u = s.session_step(x)
        full = s.get_solution()["controls"][:, 0]
        assert np.array_equal(u, full), (k, u, full)
        if np.all(np.abs(np.abs(u) - 0.4) < 1e-12):
            saturated += 1
            patterns.add(tuple(np.sign(u).astype(int)))
    s.session_end()
    assert saturated > 1000 and len(patterns) >= 6, (saturated, patterns)  # (the test must exercise what it is about)
    s.reset()
