"""Several resident sessions on one device, each driven by its own host thread: every thread opens a session on a handle of its own, ticks it
(early answer against the solution's first column, every tick; against a launched twin, bit for bit), closes it, tears the handles down and
starts over -- and every setup of one thread parks the resident kernels of all the others, whose next step restarts them.
    python tools/thread_sessions.py [threads] [rounds] [ticks per round] > gpurun_out/r05_thread_sessions.txt"""
import faulthandler, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 150
settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=30)
problems = [("quadrotor N=50", P.quadrotor(50), False), ("cartpole N=10", P.cartpole(10, True), False), ("rocket N=20 + families", P.rocket(20), True), ("quadrotor N=20", P.quadrotor(20), False)]
errors, done = [], [0] * nthreads


def make(prob, fam):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
    if fam:
        s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
    return s


def worker(tid):
    try:
        for rd in range(rounds):
            name, prob, fam = problems[(tid + rd) % len(problems)]
            t_a = time.time()
            a, b = make(prob, fam), make(prob, fam)
            t_b = time.time()
            a.session_begin()
            print(f"  [{time.time() - t0:6.1f} s] thread {tid} round {rd} ({name}): two setups {t_b - t_a:.2f} s, session_begin {time.time() - t_b:.2f} s", flush=True)
            x = prob.x0.copy()
            for k in range(ticks):
                u = a.session_step(x)
                full = a.get_solution()["controls"][:, 0]
                ub = b.mpc_step(x)[:, 0]
                if not (np.array_equal(u, full) and np.array_equal(u, ub) and a.get_stats()["iter"] == b.get_stats()["iter"]):
                    errors.append(f"thread {tid} round {rd} ({name}) tick {k}: early {u} solution {full} launched {ub} iters {a.get_stats()['iter']}/{b.get_stats()['iter']}")
                    break
                x = prob.A @ x + prob.B @ u + (prob.fdyn if prob.fdyn is not None else 0.0)
                done[tid] += 1
            a.session_end(); a.reset(); b.reset()
            print(f"  [{time.time() - t0:6.1f} s] thread {tid} round {rd} ({name}) done", flush=True)
    except Exception as ex:  # noqa: BLE001
        errors.append(f"thread {tid}: {type(ex).__name__}: {ex}")


faulthandler.dump_traceback_later(float(os.environ.get('STRESS_DUMP_AFTER', '45')), exit=True)  # (a stall shows where every thread is)
t0 = time.time()
threads = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
for t in threads: t.start()
for t in threads: t.join()
faulthandler.cancel_dump_traceback_later()
print(f"{nthreads} threads x {rounds} sessions x {ticks} ticks: {done} ticks done in {time.time() - t0:.1f} s; {len(errors)} error(s)", flush=True)
for e in errors[:10]: print("  " + e[:400])
sys.exit(1 if errors else 0)
