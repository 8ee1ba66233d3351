"""Randomised shapes through the run-time specialised layout D (both register plans, all three lane widths, constant and
per-knot tables, ragged batches, termination inside a wavefront) against the oracle. One-off stress run for the GPU box:
  python tools/fuzz_layout_d.py [count] [seed] > gpurun_out/fuzz_layout_d.txt
  python tools/fuzz_layout_d.py [count] [seed] large      the same for layout M (64 < nx+nu <= 128)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g  # noqa: E402
import pyoracle as O  # noqa: E402  (checker)

pkg = g.load_package()
P = pkg.problems
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
large = len(sys.argv) > 3 and sys.argv[3] == "large"
os.environ.pop("TINYMPC_LAYOUT", None)
worst, fails, on_d = 0.0, 0, 0
for case in range(count):
    width = 128 if large else int(rng.choice([16, 16, 32, 64]))
    nxu = int(rng.integers(3, 17)) if width == 16 else int(rng.integers(width // 2 + 1, width + 1))
    nu = int(rng.integers(1, max(2, nxu // 3 + 1)))
    nx = nxu - nu
    nmax = {16: 110, 32: 70, 64: 45, 128: 22}[width]
    N = int(rng.integers(4, nmax))
    varying = bool(rng.integers(0, 2))
    batch = int(rng.choice([1, 17, 130, 700]) if large else rng.choice([17, 130, 1030, 1500, 2300]))
    A = 0.8 * np.eye(nx) + (0.12 / np.sqrt(nx)) * rng.standard_normal((nx, nx))  # stable: a weakly controllable unstable system makes the
    # Riccati recursion itself ill-conditioned, and the cache's round-off then exceeds the tolerance on every layout alike
    B = 0.2 * rng.standard_normal((nx, nu))
    prob = P.Problem("fuzz", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(0.5, 3.0)), rng.standard_normal(nx))
    if varying:
        prob.x_min = -1.5 - rng.uniform(0, 0.5, (nx, N))
        prob.x_max = 1.5 + rng.uniform(0, 0.5, (nx, N))
        prob.u_min = -0.4 * rng.uniform(0.7, 1.0, (nu, N - 1))
        prob.u_max = 0.4 * rng.uniform(0.7, 1.0, (nu, N - 1))
        prob.x_ref = 0.05 * rng.standard_normal((nx, N))
        prob.u_ref = 0.02 * rng.standard_normal((nu, N - 1))
    else:
        prob.x_min, prob.x_max = np.full(nx, -1.5), np.full(nx, 1.5)
        prob.u_min, prob.u_max = np.full(nu, -0.4), np.full(nu, 0.4)
    settings = dict(max_iter=int(rng.integers(20, 90)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 3])))
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.02, 1.0, batch)[None, :]
    t0 = time.time()
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if varying:
        s.set_x_ref(prob.x_ref)
        s.set_u_ref(prob.u_ref)
    errs = []
    ok = True
    sample = sorted(set(b for b in (0, 1, 2, 3, batch // 2, batch - 2, batch - 1) if 0 <= b < batch))
    for rnd in range(2):  # cold, then warm
        xs = x0s if rnd == 0 else x0s + 0.05 * rng.standard_normal(x0s.shape)
        s.set_x0_batch(xs)
        s.solve()
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        # one oracle per sampled instance keeps the warm-start state apart
        if rnd == 0:
            orcs = {b: O.OraclePort(prob).load_problem(prob, settings) for b in sample}
        for b in sample:
            orcs[b].set_x0(xs[:, b])
            orcs[b].solve()
            ox, ou = orcs[b].solution()
            e = max(np.max(np.abs(sol["states"][:, :, b] - ox)) / max(np.max(np.abs(ox)), 1e-300), np.max(np.abs(sol["controls"][:, :, b] - ou)) / max(np.max(np.abs(ou)), 1e-300))
            errs.append(e)
            if st["iter"][b] != orcs[b].stats()["iter"] or st["status"][b] != orcs[b].stats()["status"]:
                ok = False
    c = s.get_cache()
    cache_err = max(np.max(np.abs(c[n] - orcs[sample[0]].get(n))) / np.max(np.abs(orcs[sample[0]].get(n))) for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"))
    layout = s.launch_info()["layout"]
    on_d += layout == ("M" if large else "D")
    e = max(errs)
    worst = max(worst, e)
    bad = (not ok) or e > 1e-8
    fails += bad
    print(f"case {case:3d}: nx={nx:2d} nu={nu:2d} N={N:3d} batch={batch:5d} varying={int(varying)} ct={settings['check_termination']} -> layout {layout} "
          f"workgroups {s.launch_info()['workgroups']:4d} | cache {cache_err:.0e} | max rel err {e:.1e} iterations {'equal' if ok else 'DIFFER'} | {time.time() - t0:4.1f} s{'   <-- FAIL' if bad else ''}", flush=True)
    s.reset()
print(f"# {count} cases, {on_d} on layout {'M' if large else 'D'}, worst rel err {worst:.1e}, {fails} failure(s)")
sys.exit(1 if fails else 0)
