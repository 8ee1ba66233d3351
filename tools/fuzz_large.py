"""Randomised LARGE systems (65 <= nx+nu <= 512: layout M, every row-tile count R = 5..32 can occur, both the register-resident and the
streamed form; the multi-launch Riccati precompute) against the oracle: caches, iteration counts, statuses, trajectories, over a cold
and a warm start, constant and per-knot bounds / references, ragged last tile.
  python tools/fuzz_large.py [count] [seed] [families] > gpurun_out/fuzz_large.txt
With a third argument every case also gets random cones (anywhere in the state / input vector: inside a row tile, across tiles,
overlapping) and linear rows on either side (round 4: layout M's families phase)."""
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
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails, worst = 0, 0.0
for case in range(count):
    nxu = int(rng.choice([rng.integers(65, 129), rng.integers(129, 257), rng.integers(257, 513)]))
    nu = int(rng.integers(2, min(40, nxu // 3)))
    nx = nxu - nu
    N = int(rng.integers(4, 12))
    batch = int(rng.choice([1, 5, 16, 21, 37]))
    A = (0.6 if nx >= 256 else 0.9) * np.eye(nx) + (0.12 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.1 * rng.standard_normal((nx, nu))
    prob = P.Problem("fuzzlarge", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(0.5, 3.0)), rng.standard_normal(nx))
    varying = bool(rng.integers(0, 2))
    if varying:
        prob.x_min, prob.x_max = -2.0 - rng.uniform(0, 0.5, (nx, N)), 2.0 + rng.uniform(0, 0.5, (nx, N))
        prob.u_min, prob.u_max = -0.3 * rng.uniform(0.7, 1.0, (nu, N - 1)), 0.3 * rng.uniform(0.7, 1.0, (nu, N - 1))
        prob.x_ref, prob.u_ref = 0.05 * rng.standard_normal((nx, N)), 0.02 * rng.standard_normal((nu, N - 1))
    else:
        prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
        prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    prob.fdyn = 0.01 * rng.standard_normal(nx) if rng.integers(0, 2) else None
    fam_note = ""
    if len(sys.argv) > 3:
        def cones(dim, n):
            starts, dims, mus = [], [], []
            for _ in range(n):
                q = int(rng.integers(2, min(dim, 40) + 1))
                starts.append(int(rng.integers(0, dim - q + 1))); dims.append(q); mus.append(float(rng.uniform(0.3, 1.5)))
            return starts, dims, mus
        ncx, ncu = int(rng.integers(0, 5)), int(rng.integers(0, 3))
        ax, qx, cx = cones(nx, ncx)
        au, qu, cu = cones(nu, ncu)
        nlx, nlu = int(rng.integers(0, 5)), int(rng.integers(0, 3))
        if ncx + ncu + nlx + nlu == 0:
            ncx, (ax, qx, cx) = 1, ([0], [3], [0.8])
        prob.cones = dict(Acx=ax, qcx=qx, cx=cx, Acu=au, qcu=qu, cu=cu)
        prob.linear = dict(Alin_x=rng.standard_normal((nlx, nx)) / np.sqrt(nx), blin_x=rng.uniform(0.05, 0.5, nlx),
                           Alin_u=rng.standard_normal((nlu, nu)) / np.sqrt(nu), blin_u=rng.uniform(0.05, 0.3, nlu))
        fam_note = f" cones {ncx}+{ncu} rows {nlx}+{nlu}"
    settings = dict(max_iter=int(rng.integers(15, 60)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 3])))
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.05, 1.0, batch)[None, :]
    t0 = time.time()
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref)
        s.set_u_ref(prob.u_ref)
    if prob.cones:
        s.set_cone_constraints(**prob.cones)
        s.set_linear_constraints(**prob.linear)
    sample = sorted(set(b for b in (0, batch // 2, batch - 1)))
    orcs = {b: O.OraclePort(prob).load_problem(prob, settings) for b in sample}
    c = s.get_cache()
    ce = max(np.max(np.abs(c[n] - orcs[sample[0]].get(n))) / max(np.max(np.abs(orcs[sample[0]].get(n))), 1e-300) for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt"))
    ok, e = True, 0.0
    for rnd in range(2):
        xs = x0s * (1.0 - 0.3 * rnd)
        s.set_x0_batch(xs) if batch > 1 else s.set_x0(xs[:, 0])
        s.solve()
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in sample:
            orcs[b].set_x0(xs[:, b])
            orcs[b].solve()
            ox, ou = orcs[b].solution()
            e = max(e, np.max(np.abs(sol["states"][:, :, b] - ox)) / max(np.max(np.abs(ox)), 1e-300), np.max(np.abs(sol["controls"][:, :, b] - ou)) / max(np.max(np.abs(ou)), 1e-300))
            if st["iter"][b] != orcs[b].stats()["iter"] or st["status"][b] != orcs[b].stats()["status"]:
                ok = False
    layout = s.launch_info()["layout"]
    worst = max(worst, e, ce)
    bad = (not ok) or e > 1e-6 or ce > 1e-6 or layout != "M"
    fails += bad
    print(f"case {case:3d}: nx={nx:3d} nu={nu:2d} (R={(nxu + 15) // 16:2d}) N={N:2d} batch={batch:2d} per-knot tables={int(varying)} fdyn={int(prob.fdyn is not None)}{fam_note} "
          f"check every {settings['check_termination']} -> layout {layout} | caches {ce:.1e} trajectories {e:.1e} iterations {'equal' if ok else 'DIFFER'} | "
          f"{time.time() - t0:5.1f} s{'   <-- FAIL' if bad else ''}", flush=True)
    s.reset()
print(f"# {count} cases, worst rel err {worst:.1e}, {fails} failure(s)")
sys.exit(1 if fails else 0)
