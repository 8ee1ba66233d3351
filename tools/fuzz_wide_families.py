"""Randomised cone / linear-inequality configurations on WIDE systems (17 <= nx+nu <= 64) through layout D's streamed-families kernels
(tinympc_solve_dwide.h, round 5) against the restated oracle: horizons 4..40, batches 16..300, up to three state cones of which two may
share rows (rounds), an input cone, up to 6 linear rows per side, fdyn, constant or per-knot bounds, cold + warm solve.
    python tools/fuzz_wide_families.py [count] [seed] > gpurun_out/r05_fuzz_wide_families.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
import pyoracle as O
pkg = g.load_package(); P = pkg.problems
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ.pop("TINYMPC_LAYOUT", None)
fails, worst, on_d = 0, 0.0, 0
for case in range(count):
    t0 = time.time()
    nxu = int(rng.integers(17, 65))
    nu = int(rng.integers(2, max(3, nxu // 4)))
    nx = nxu - nu
    N = int(rng.integers(4, 41))
    batch = int(rng.choice([16, 33, 70, 300]))
    A = 0.9 * np.eye(nx) + (0.15 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    Bm = 0.3 * rng.standard_normal((nx, nu))
    prob = P.Problem("widefuzz", A, Bm, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(0.5, 3.0)), rng.standard_normal(nx))
    prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    prob.u_min, prob.u_max = np.full(nu, -1.0), np.full(nu, 1.0)
    if rng.integers(0, 2): prob.fdyn = 0.01 * rng.standard_normal(nx)
    ncx = int(rng.integers(0, 4))
    Acx, qcx, cx = [], [], []
    for c in range(ncx):
        q = int(rng.integers(2, 6)); a = int(rng.integers(0, nx - q + 1))
        if c == 2 and Acx: a = min(Acx[0] + 1, nx - q)  # shares rows with the first cone: a second round
        Acx.append(a); qcx.append(q); cx.append(float(rng.uniform(0.4, 1.5)))
    has_cu = bool(rng.integers(0, 2)) and nu >= 2
    qcu = [int(rng.integers(2, min(nu, 4) + 1))] if has_cu else []
    prob.cones = dict(Acx=Acx, qcx=qcx, cx=cx, Acu=[0] if has_cu else [], qcu=qcu, cu=[0.7] if has_cu else [])
    nlx, nlu = int(rng.integers(0, 7)), int(rng.integers(0, 4))
    if not (ncx or has_cu or nlx or nlu): nlx = 1
    prob.linear = dict(Alin_x=rng.standard_normal((nlx, nx)), blin_x=rng.uniform(0.5, 1.5, nlx), Alin_u=rng.standard_normal((nlu, nu)), blin_u=rng.uniform(0.3, 0.8, nlu))
    settings = dict(max_iter=int(rng.integers(20, 80)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 3])))
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
    xmin, xmax = prob.x_min, prob.x_max
    varying = bool(rng.integers(0, 3) == 0)
    if varying:
        xmin = np.repeat(prob.x_min[:, None], N, 1) * rng.uniform(0.8, 1.0, (1, N)); xmax = -xmin
    s.set_bound_constraints(xmin, xmax, prob.u_min, prob.u_max)
    s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.1, 1.0, batch)[None, :]
    checked = sorted({0, batch // 2, batch - 1})
    orcs = {}
    for b in checked:
        o = O.OraclePort(prob).load_problem(prob, settings)
        o.set_bound_constraints(*(np.broadcast_to(np.asarray(v).reshape(len(v), -1), (len(v), n)).copy() for v, n in ((xmin, N), (xmax, N), (prob.u_min, N - 1), (prob.u_max, N - 1))))
        o.set_cone_constraints(**prob.cones); o.set_linear_constraints(**prob.linear)
        orcs[b] = o
    ok, err = True, 0.0
    for rnd in range(2):
        xs = x0s * (1.0 - 0.3 * rnd)
        s.set_x0_batch(np.asfortranarray(xs)); s.solve()
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in checked:
            orcs[b].set_x0(xs[:, b]); orcs[b].solve()
            ox, ou = orcs[b].solution()
            e = max(np.max(np.abs(sol["states"][:, :, b] - ox)) / max(np.max(np.abs(ox)), 1e-300), np.max(np.abs(sol["controls"][:, :, b] - ou)) / max(np.max(np.abs(ou)), 1e-300))
            err = max(err, e)
            if st["iter"][b] != orcs[b].stats()["iter"] or st["status"][b] != orcs[b].stats()["status"] or e > 1e-9: ok = False
    lay = s.launch_info()["layout"]; on_d += lay == "D"; worst = max(worst, err); fails += not ok
    print(f"case {case:3d}: nx={nx:2d} nu={nu:2d} N={N:2d} batch={batch:3d} cones {ncx}+{int(has_cu)} rows {nlx}+{nlu} fdyn={int(prob.fdyn is not None)} per-knot={int(varying)} -> layout {lay} {s.jit_info()[:48]} | rel err {err:.1e} {'ok' if ok else 'FAIL'} | {time.time() - t0:4.1f} s", flush=True)
    s.reset()
print(f"# {count} cases, {on_d} on layout D (streamed families), worst rel err {worst:.1e}, {fails} failure(s)")
