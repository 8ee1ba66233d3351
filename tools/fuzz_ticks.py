"""Randomised closed-loop ticks through tinympc_mpc_step_batch (host x0 in, first controls out; the zero-copy path for small batches, the
copy engines beyond) against the restated oracle run tick by tick: shapes nx+nu 3..16 and 17..40, horizons 5..60, batches 1 .. 700,
box constraints with saturating inputs, per-knot references on some cases, cones / linear rows on some (16-lane shapes), eight warm-started
ticks with the plant stepped in between. Per checked instance and tick: first controls (1e-8 relative to the bound), iteration count, status.
    python tools/fuzz_ticks.py [count] [seed] > gpurun_out/r05_fuzz_ticks.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
import pyoracle as O
pkg = g.load_package(); P = pkg.problems
count = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ.pop("TINYMPC_LAYOUT", None)
fails, worst, layouts = 0, 0.0, {}
for case in range(count):
    t0 = time.time()
    wide = rng.integers(0, 5) == 0
    nxu = int(rng.integers(17, 41)) if wide else int(rng.integers(3, 17))
    nu = int(rng.integers(1, max(2, nxu // 3 + 1)))
    nx = nxu - nu
    N = int(rng.integers(5, 25)) if wide else int(rng.choice([rng.integers(5, 16), rng.integers(16, 40), rng.integers(40, 61)]))
    batch = int(rng.choice([1, 2, 5, 17, 64, 300, 700]))
    A = np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    A *= 0.99 / max(1.0, np.abs(np.linalg.eigvals(A)).max())
    B = 0.25 * rng.standard_normal((nx, nu))
    prob = P.Problem("fuzztick", A, B, np.diag(rng.uniform(1, 8, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(0.5, 4.0)), rng.standard_normal(nx))
    ub = float(rng.choice([0.2, 0.5, 2.0]))
    prob.u_min, prob.u_max = np.full(nu, -ub), np.full(nu, ub)
    if rng.integers(0, 2):
        prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    if rng.integers(0, 3) == 0:
        prob.x_ref, prob.u_ref = 0.05 * rng.standard_normal((nx, N)), 0.02 * rng.standard_normal((nu, N - 1))
    fam = (not wide) and rng.integers(0, 3) == 0
    if fam:
        q = int(rng.integers(2, min(nx, 4) + 1))
        prob.cones = dict(Acx=[int(rng.integers(0, nx - q + 1))], qcx=[q], cx=[float(rng.uniform(0.5, 1.5))], Acu=[], qcu=[], cu=[])
        prob.linear = dict(Alin_x=rng.standard_normal((1, nx)) / np.sqrt(nx), blin_x=rng.uniform(0.3, 0.8, 1), Alin_u=np.zeros((0, nu)), blin_u=np.zeros(0))
    settings = dict(max_iter=int(rng.integers(10, 60)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 2])))
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, **settings)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    if prob.x_ref is not None:
        s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
    if fam:
        s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
    x = rng.standard_normal((nx, batch)) * np.linspace(0.2, 1.5, batch)[None, :]
    checked = sorted({0, batch // 2, batch - 1})
    orcs = {b: O.OraclePort(prob).load_problem(prob, settings) for b in checked}
    ok, err, note = True, 0.0, ""
    try:
        for k in range(8):
            u0 = s.mpc_step(np.asfortranarray(x))
            st = s.get_stats_batch()
            for b in checked:
                o = orcs[b]
                o.set_x0(x[:, b]); o.solve()
                ou0 = o.solution()[1][:, 0]
                e = float(np.max(np.abs(u0[:, b] - ou0)) / ub)
                err = max(err, e)
                if not (st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"] and e < 1e-8):
                    ok = False
                    note = f" [tick {k} instance {b}: iter {st['iter'][b]}/{o.stats()['iter']} status {st['status'][b]}/{o.stats()['status']} u0 err {e:.1e}]"
            x = prob.A @ x + prob.B @ u0
    except pkg.TinyMPCError as ex:
        ok, note = False, f" [{str(ex)[:90]}]"
    lay = s.launch_info()["layout"]
    layouts[lay] = layouts.get(lay, 0) + 1
    worst = max(worst, err); fails += not ok
    print(f"case {case:3d}: nx={nx:2d} nu={nu:2d} N={N:2d} batch={batch:3d} |u|<={ub} refs={int(prob.x_ref is not None)} families={int(fam)} check every {settings['check_termination']} "
          f"-> layout {lay} {s.jit_info()[:40]} | u0 err {err:.1e} {'ok' if ok else 'FAIL' + note} | {time.time() - t0:4.1f} s", flush=True)
    s.reset()
print(f"# {count} cases x 8 ticks, by layout {layouts}, worst u0 err {worst:.1e}, {fails} failure(s)")
sys.exit(1 if fails else 0)
