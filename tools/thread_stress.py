"""Several host threads on one device at once (ctypes releases the interpreter lock inside every call of the C ABI): three threads cycle through
setup -> solve -> compare -> reset on handles of their own (pooled streams and arenas change hands between threads, run-time specialisations are
looked up and built concurrently), a fourth keeps a resident session ticking -- every setup of the others parks its kernel, every next step
restarts it -- and checks each tick's early answer against the solution's first column and against a launched twin. Expected results are
computed single-threaded first; every comparison is bit for bit.
    python tools/thread_stress.py [cycles per thread] > gpurun_out/r05_thread_stress.txt"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=40)


def make(prob, batch):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if getattr(prob, "cones", None):
        s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
    return s


def solve_once(prob, batch, x0s):
    s = make(prob, batch)
    s.set_x0_batch(np.asfortranarray(x0s)) if batch > 1 else s.set_x0(x0s[:, 0])
    s.solve()
    out = (s.get_solution_batch()["controls"].copy(), s.get_stats_batch()["iter"].copy())
    s.reset()
    return out


rng = np.random.default_rng(0)
work = []  # (name, problem, batch, x0s, expected)
for name, prob, batch in (("quadrotor N=50 x 512", P.quadrotor(50), 512), ("cartpole N=20 x 1", P.cartpole(20, True), 1), ("rocket N=20 + families x 64", P.rocket(20), 64),
                          ("quadrotor N=23 x 2000 (specialised)", P.quadrotor(23), 2000), ("rocket N=100 + families x 1", P.rocket(100), 1)):
    x0s = prob.x0[:, None] * rng.uniform(0.3, 1.2, (1, batch))
    work.append((name, prob, batch, x0s, solve_once(prob, batch, x0s)))
print("expected results computed", flush=True)
errors, done = [], [0, 0, 0, 0]


def cycler(tid):
    try:
        for c in range(cycles):
            name, prob, batch, x0s, want = work[(tid + c) % len(work)]
            got = solve_once(prob, batch, x0s)
            if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
                errors.append(f"thread {tid} cycle {c}: {name}: result differs from the single-threaded one (max |du| {np.max(np.abs(got[0] - want[0])):.2e})")
            done[tid] += 1
    except Exception as ex:  # noqa: BLE001
        errors.append(f"thread {tid}: {type(ex).__name__}: {ex}")


def session():
    try:
        prob = P.quadrotor(50)
        a, b = make(prob, 1), make(prob, 1)
        a.session_begin()
        x = prob.x0.copy()
        while any(t.is_alive() for t in threads[:3]):
            u = a.session_step(x)
            full = a.get_solution()["controls"][:, 0]
            ub = b.mpc_step(x)[:, 0]
            if not (np.array_equal(u, full) and np.array_equal(u, ub)):
                errors.append(f"session tick {done[3]}: early {u} solution {full} launched {ub}")
            x = prob.A @ x + prob.B @ u
            if np.max(np.abs(x)) < 1e-3:
                x = prob.x0 * np.random.default_rng(done[3]).uniform(0.3, 1.0)
            done[3] += 1
        a.session_end(); a.reset(); b.reset()
    except Exception as ex:  # noqa: BLE001
        errors.append(f"session thread: {type(ex).__name__}: {ex}")


threads = [threading.Thread(target=cycler, args=(i,)) for i in range(3)] + [threading.Thread(target=session)]
t0 = time.time()
for t in threads: t.start()
for t in threads: t.join()
print(f"{cycles} cycles x 3 threads + a session thread: {done[:3]} cycles, {done[3]} session ticks in {time.time() - t0:.1f} s; {len(errors)} error(s)", flush=True)
for e in errors[:10]: print("  " + e)
sys.exit(1 if errors else 0)
