"""Randomised cone / linear-inequality configurations on large batches (layout D's FAM variant where the horizon fits, else
k_admm_solve_fam / the latency kernel) against the restated oracle. One-off stress run for the GPU box:
  python tools/fuzz_families_d.py [count] [seed] > gpurun_out/fuzz_families_d.txt"""
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
count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ.pop("TINYMPC_LAYOUT", None)
fails, on_d, worst = 0, 0, 0.0
for case in range(count):
    nu = int(rng.integers(2, 5))
    nx = int(rng.integers(3, 17 - nu))
    N = int(rng.integers(5, 30))
    batch = int(rng.choice([900, 1300, 2100]))
    A = 0.85 * np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.3 * rng.standard_normal((nx, nu))
    prob = P.Problem("fuzzfam", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(0.5, 3.0)), rng.standard_normal(nx))
    prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    prob.u_min, prob.u_max = np.full(nu, -1.0), np.full(nu, 1.0)
    prob.fdyn = 0.01 * rng.standard_normal(nx) if rng.integers(0, 2) else None
    if rng.integers(0, 2):
        prob.x_ref = 0.05 * rng.standard_normal((nx, N))
        prob.u_ref = 0.02 * rng.standard_normal((nu, N - 1))
    cones = dict(Acx=[], qcx=[], cx=[], Acu=[], qcu=[], cu=[])
    kind = int(rng.integers(0, 4))  # 0 cones, 1 linear, 2 both, 3 both with two state cones
    if kind != 1:
        if nx >= 3:
            q = int(rng.integers(2, min(nx, 4) + 1))
            cones["Acx"], cones["qcx"], cones["cx"] = [0], [q], [float(rng.uniform(0.3, 1.5))]
            if kind == 3 and nx - q >= 2:
                cones["Acx"].append(q); cones["qcx"].append(2); cones["cx"].append(float(rng.uniform(0.3, 1.5)))
        if nu >= 2 and rng.integers(0, 2):
            cones["Acu"], cones["qcu"], cones["cu"] = [0], [nu], [float(rng.uniform(0.3, 1.5))]
    prob.cones = cones if (cones["Acx"] or cones["Acu"]) else {}
    if kind != 0:
        mx, mu_ = int(rng.integers(1, 4)), int(rng.integers(0, 3))
        prob.linear = dict(Alin_x=rng.standard_normal((mx, nx)), blin_x=rng.uniform(0.5, 2.0, mx),
                           Alin_u=rng.standard_normal((mu_, nu)), blin_u=rng.uniform(0.3, 1.0, mu_))
    else:
        prob.linear = {}
    if not prob.cones and not prob.linear:
        prob.linear = dict(Alin_x=rng.standard_normal((1, nx)), blin_x=np.array([1.0]), Alin_u=np.zeros((0, nu)), blin_u=np.zeros(0))
    settings = dict(max_iter=int(rng.integers(20, 80)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 2])))
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.05, 1.0, batch)[None, :]
    t0 = time.time()
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.cones:
        s.set_cone_constraints(**prob.cones)
    if prob.linear:
        s.set_linear_constraints(**prob.linear)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref)
        s.set_u_ref(prob.u_ref)
    sample = sorted(set([0, 1, 2, 3, batch // 2, batch - 2, batch - 1]))
    orcs = {b: O.OraclePort(prob).load_problem(prob, settings) for b in sample}
    ok, e = True, 0.0
    for rnd in range(2):
        xs = x0s if rnd == 0 else x0s + 0.05 * rng.standard_normal(x0s.shape)
        s.set_x0_batch(xs)
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
    on_d += layout == "D"
    worst = max(worst, e)
    bad = (not ok) or e > 1e-8
    fails += bad
    print(f"case {case:3d}: nx={nx:2d} nu={nu} N={N:2d} batch={batch} kind={kind} cones x{len(cones['Acx'])} u{len(cones['Acu'])} linear x{0 if not prob.linear else prob.linear['Alin_x'].shape[0]} "
          f"u{0 if not prob.linear else prob.linear['Alin_u'].shape[0]} fdyn={int(prob.fdyn is not None)} refs={int(prob.x_ref is not None)} -> layout {layout} | max rel err {e:.1e} iterations "
          f"{'equal' if ok else 'DIFFER'} | {time.time() - t0:4.1f} s{'   <-- FAIL' if bad else ''}", flush=True)
    s.reset()
print(f"# {count} cases, {on_d} on layout D, worst rel err {worst:.1e}, {fails} failure(s)")
sys.exit(1 if fails else 0)
