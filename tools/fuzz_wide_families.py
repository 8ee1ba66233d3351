"""Randomised cone / linear-inequality configurations on WIDE systems (17 <= nx+nu <= 64) through layout D's streamed-families kernels
(tinympc_solve_dwide.h, round 5) against the restated oracle: horizons 4..40, batches 16..300, up to three state cones of which two may
share rows (rounds), an input cone, up to 6 linear rows per side, fdyn, constant or per-knot bounds, cold + warm solve.
    python tools/fuzz_wide_families.py [count] [seed] > gpurun_out/r05_fuzz_wide_families.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import pyoracle as O
from wide_case import draw  # (tools/repro_wide_case.py replays a case of this generator)
pkg = g.load_package(); P = pkg.problems
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ.pop("TINYMPC_LAYOUT", None)
fails, worst, on_d = 0, 0.0, 0
for case in range(count):
    t0 = time.time()
    c = draw(rng)
    nx, nu, N, batch = c["nx"], c["nu"], c["N"], c["batch"]
    prob = P.Problem("widefuzz", c["A"], c["B"], c["Q"], c["R"], N, c["rho"], c["xref"])
    prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    prob.u_min, prob.u_max = np.full(nu, -1.0), np.full(nu, 1.0)
    prob.fdyn, prob.cones, prob.linear = c["fdyn"], c["cones"], c["linear"]
    settings, varying, x0s = c["settings"], c["scale"] is not None, c["x0s"]
    ncx, has_cu, nlx, nlu = len(prob.cones["Acx"]), bool(prob.cones["Acu"]), len(prob.linear["blin_x"]), len(prob.linear["blin_u"])
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
    xmin, xmax = prob.x_min, prob.x_max
    if varying:
        xmin = np.repeat(prob.x_min[:, None], N, 1) * c["scale"]; xmax = -xmin
    s.set_bound_constraints(xmin, xmax, prob.u_min, prob.u_max)
    s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
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
