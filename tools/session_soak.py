"""Long resident sessions under checks no single test can afford: per tick the early answer (session_step's controls) against the first column
of the solution read right after it; the first 3,000 ticks also against launched ticks of a twin handle, bit for bit; every 5,000th tick another handle is
set up and torn down on the device (its allocations park the resident kernel: the next step restarts it); references shifted by one knot every
7th tick where the problem has per-knot references. Shapes: quadrotor N=50 (layout F, compiled in), cartpole N=10 (prepared: layout F; not prepared: layout C), rocket N=20
with cones + a linear row (layout F, specialised), a random system whose four inputs sit on their bounds (layouts F and C).
    python tools/session_soak.py [ticks per shape] > gpurun_out/r05_session_soak.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
twin_ticks = 3000


def saturating():
    rng = np.random.default_rng(3)
    nx, nu, N = 10, 4, 12
    A = 0.9 * np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.3 * rng.standard_normal((nx, nu))
    prob = P.Problem("saturating", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 1.5, rng.standard_normal(nx))
    prob.x_min, prob.x_max = np.full(nx, -50.0), np.full(nx, 50.0)
    prob.u_min, prob.u_max = np.full(nu, -0.4), np.full(nu, 0.4)
    return prob


def handle(prob, settings, fam, prepare=True):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
    if fam:
        s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
    if prepare:
        s.prepare()
    return s


total_bad = 0
for name, prob, fam, scale, prep in (("quadrotor N=50", P.quadrotor(50), False, 1.0, True), ("cartpole N=10, prepared", P.cartpole(10, True), False, 1.0, True),
                                     ("cartpole N=10, not prepared", P.cartpole(10, True), False, 1.0, False), ("rocket N=20, cones + row", P.rocket(20), True, 1.0, True),
                                     ("random 10 x 4, saturating inputs", saturating(), False, 4.0, True), ("random 10 x 4, saturating, not prepared", saturating(), False, 4.0, False)):
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=12)
    rng = np.random.default_rng(7)
    a, b = handle(prob, settings, fam, prep), handle(prob, settings, fam, prep)
    a.session_begin()
    layout = a.launch_info()["layout"]
    x0 = prob.x0.copy()
    bad_early, bad_twin, restarts, shifts, t0 = 0, 0, 0, 0, time.time()
    xr, ur = (prob.x_ref.copy(), prob.u_ref.copy()) if prob.x_ref is not None and np.ndim(prob.x_ref) == 2 else (None, None)
    for k in range(ticks):
        x = x0 * rng.uniform(-1.0, 1.0) * scale + 0.05 * scale * rng.standard_normal(prob.nx)
        if xr is not None and k % 7 == 3:  # receding horizon: the references move up by one knot (both handles)
            xr, ur = np.roll(xr, -1, axis=1), np.roll(ur, -1, axis=1)
            for h in (a, b):
                h.set_x_ref(xr); h.set_u_ref(ur)
            shifts += 1
        u = a.session_step(x)
        full = a.get_solution()["controls"][:, 0]
        if not np.array_equal(u, full):
            bad_early += 1
            if bad_early <= 3:
                print(f"  {name}: tick {k}: early answer {u} != solution's first column {full}", flush=True)
        if k < twin_ticks:  # the twin ticks along (its warm state stays in step only if it solves every tick)
            ub = b.mpc_step(x)[:, 0]
            if not (np.array_equal(u, ub) and a.get_stats()["iter"] == b.get_stats()["iter"]):
                bad_twin += 1
                if bad_twin <= 3:
                    print(f"  {name}: tick {k}: session {u} (iter {a.get_stats()['iter']}) != launched {ub} (iter {b.get_stats()['iter']})", flush=True)
        if k % 5000 == 2500 or k == 1500:
            other = handle(P.cartpole(20, True), settings, False)  # (its allocations park the resident kernel of `a`)
            other.set_x0(P.cartpole(20, True).x0); other.solve(); other.reset()
            restarts += 1
    a.session_end()
    dt = time.time() - t0
    total_bad += bad_early + bad_twin
    print(f"{name:40s} layout {layout}: {ticks} ticks, {1e6 * dt / ticks:5.1f} us per tick through Python with the checks; early answer != solution: {bad_early}; first {twin_ticks} ticks != launched twin: {bad_twin}; "
          f"{restarts} parks + restarts, {shifts} reference shifts", flush=True)
    a.reset(); b.reset()
print(f"# {total_bad} bad tick(s)")
sys.exit(1 if total_bad else 0)
