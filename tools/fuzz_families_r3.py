"""Randomised cone / linear-inequality configurations over round 3's kernels and constraint shapes against the restated oracle:
horizons 5..110 and batches 1..1,100 (layout F below 260 instances, layout E / layout D's FAM variant above, the latency kernel
and k_admm_solve_fam where neither has a kernel), up to three state cones of which two may SHARE rows (projected in list order),
up to 12 linear rows per side, equality constraints (as the +/- row pairs the class turns them into). One-off stress run for the
GPU box:  python tools/fuzz_families_r3.py [count] [seed] > gpurun_out/fuzz_families_r3.txt"""
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
only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None  # re-run one case (e.g. under TINYMPC_LAYOUT=...)
if only is None:
    os.environ.pop("TINYMPC_LAYOUT", None)
fails, worst = 0, 0.0
layouts = {}
for case in range(count):
    nu = int(rng.integers(2, 5))
    nx = int(rng.integers(3, 17 - nu))
    N = int(rng.choice([rng.integers(5, 30), rng.integers(30, 70), rng.integers(70, 111)]))
    batch = int(rng.choice([1, 60, 300, 1100]))
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
    kind = int(rng.integers(0, 5))  # 0 cones, 1 linear, 2 both, 3 both with two state cones, 4 both with two state cones that share a row
    if kind != 1:
        if nx >= 3:
            q = int(rng.integers(2, min(nx, 4) + 1))
            cones["Acx"], cones["qcx"], cones["cx"] = [0], [q], [float(rng.uniform(0.3, 1.5))]
            if kind == 3 and nx - q >= 2:
                cones["Acx"].append(q); cones["qcx"].append(2); cones["cx"].append(float(rng.uniform(0.3, 1.5)))
            if kind == 4 and nx - q >= 1:  # the second cone starts on the first one's last row
                cones["Acx"].append(q - 1); cones["qcx"].append(2); cones["cx"].append(float(rng.uniform(0.3, 1.5)))
                if nx - q >= 4 and rng.integers(0, 2):
                    cones["Acx"].append(q + 1); cones["qcx"].append(3); cones["cx"].append(float(rng.uniform(0.3, 1.5)))
        if nu >= 2 and rng.integers(0, 2):
            cones["Acu"], cones["qcu"], cones["cu"] = [0], [nu], [float(rng.uniform(0.3, 1.5))]
    prob.cones = cones if (cones["Acx"] or cones["Acu"]) else {}
    if kind != 0:
        mx, mu_ = int(rng.choice([1, 2, 3, 9, 12])), int(rng.choice([0, 1, 2, 10]))
        Ax, bx = rng.standard_normal((mx, nx)), rng.uniform(0.5, 2.0, mx)
        if rng.integers(0, 3) == 0:  # an equality constraint a'x = b rides along as the pair a'x <= b, -a'x <= -b (TinyMPC.m:296-317)
            ae, be = rng.standard_normal(nx), float(rng.uniform(-0.2, 0.2))
            Ax, bx = np.vstack([Ax, ae, -ae]), np.concatenate([bx, [be, -be]])
        prob.linear = dict(Alin_x=Ax, blin_x=bx, Alin_u=rng.standard_normal((mu_, nu)), blin_u=rng.uniform(0.3, 1.0, mu_))
    else:
        prob.linear = {}
    if not prob.cones and not prob.linear:
        prob.linear = dict(Alin_x=rng.standard_normal((1, nx)), blin_x=np.array([1.0]), Alin_u=np.zeros((0, nu)), blin_u=np.zeros(0))
    settings = dict(max_iter=int(rng.integers(20, 80)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 2])))
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.05, 1.0, batch)[None, :]
    pert = 0.05 * rng.standard_normal(x0s.shape)  # (second round: a warm start from a nearby state)
    if only is not None and case != only:
        continue
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
    s.prepare()
    sample = sorted(set(b for b in (0, 1, 2, 3, batch // 2, batch - 2, batch - 1) if 0 <= b < batch))
    orcs = {b: O.OraclePort(prob).load_problem(prob, settings) for b in sample}
    ok, e = True, 0.0
    for rnd in range(2):
        xs = x0s if rnd == 0 else x0s + pert
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
    layouts[layout] = layouts.get(layout, 0) + 1
    worst = max(worst, e)
    bad = (not ok) or e > 1e-6  # north_star: 1e-6 relative on states / controls; identical iteration counts
    noted = (not bad) and e > 1e-8
    fails += bad
    print(f"case {case:3d}: nx={nx:2d} nu={nu} N={N:2d} batch={batch} kind={kind} cones x{len(cones['Acx'])} u{len(cones['Acu'])} linear x{0 if not prob.linear else prob.linear['Alin_x'].shape[0]} "
          f"u{0 if not prob.linear else prob.linear['Alin_u'].shape[0]} fdyn={int(prob.fdyn is not None)} refs={int(prob.x_ref is not None)} -> layout {layout} | max rel err {e:.1e} iterations "
          f"{'equal' if ok else 'DIFFER'} | {time.time() - t0:4.1f} s{'   <-- FAIL' if bad else '   (above 1e-8: re-run with FUZZ_ONLY on another layout)' if noted else ''}", flush=True)
    s.reset()
print(f"# {count} cases, by layout {dict(sorted(layouts.items()))}, worst rel err {worst:.1e}, {fails} failure(s)")
sys.exit(1 if fails else 0)
