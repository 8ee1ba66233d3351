"""How much does a batch lose to instances that converge at different iteration counts? 65,536 quadrotor N=50 instances, tol 1e-3,
max_iter 200: kernel time against the time the same total number of instance-iterations takes in a forced-iteration run."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
prob = P.quadrotor(50)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
x0s = np.asfortranarray(P.quadrotor_batch_x0(B) * rng.uniform(0.05, 3.0, B)[None, :])
def run(tol, iters):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, abs_pri_tol=tol, abs_dua_tol=tol, max_iter=iters)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(x0s)
    ms = []
    for _ in range(5):
        s.reset_workspace(); ms.append(s.solve_timed())
    it = s.get_stats_batch()["iter"].astype(float)
    s.reset()
    return float(np.median(ms[1:])), it
t_forced, _ = run(0.0, 100)
rate = B * 100 / t_forced  # instance-iterations per ms at full occupancy
t, it = run(1e-3, 200)
print(f"{B} instances: iterations min {it.min():.0f} median {np.median(it):.0f} mean {it.mean():.1f} p95 {np.percentile(it, 95):.0f} max {it.max():.0f}")
print(f"kernel {t:.3f} ms; the same {it.sum():.0f} instance-iterations at the forced-iteration rate: {it.sum() / rate:.3f} ms -> efficiency {it.sum() / rate / t:.2f}")
groups = it.reshape(-1, 4).max(axis=1)
wgs = groups.reshape(-1, 4).max(axis=1)
print(f"bounds: wave-granular scheduling (a wave runs as long as its slowest instance) {groups.sum() * 4 / rate:.3f} ms, workgroup-granular {wgs.sum() * 16 / rate:.3f} ms")
