"""Randomised shapes through the adaptive-rho kernels against the restated oracle (which tests/test_adaptive_rho.py pins to the reference's
own core): nx+nu 3..16 (layout D's run-time specialised variant for batches, k_admm_solve_adapt on layout A's plan otherwise) and 17..40
(the generic kernel on 32 / 64 lanes), horizons 5..60, batches 1 .. 2,000, sensitivities from compute_sensitivity_autograd, clipping on and
off, constant and per-knot references, three solves (rho and the Taylor-updated cache persist from one to the next). Per checked instance:
iteration count, status, final rho (1e-9 relative), trajectories (1e-8).
    python tools/fuzz_adaptive.py [count] [seed] > gpurun_out/r05_fuzz_adaptive.txt"""
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
    batch = int(rng.choice([1, 7, 300] if wide else [1, 7, 300, 2000]))
    A = np.eye(nx) + (0.1 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    A *= 0.97 / max(1.0, np.abs(np.linalg.eigvals(A)).max())
    B = 0.25 * rng.standard_normal((nx, nu))
    prob = P.Problem("fuzzadapt", A, B, np.diag(rng.uniform(1, 8, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(0.5, 4.0)), rng.standard_normal(nx))
    prob.u_min, prob.u_max = np.full(nu, -0.5), np.full(nu, 0.5)
    if rng.integers(0, 2):
        prob.x_min, prob.x_max = np.full(nx, -2.5), np.full(nx, 2.5)
    varying = bool(rng.integers(0, 2))
    xref = 0.05 * rng.standard_normal((nx, N)) if varying else np.tile(0.05 * rng.standard_normal((nx, 1)), (1, N))
    clip = bool(rng.integers(0, 4) != 0)
    rho_min, rho_max = float(rng.uniform(0.1, 0.8)), float(rng.uniform(10.0, 60.0))
    settings = dict(max_iter=int(rng.integers(15, 70)), abs_pri_tol=float(rng.choice([0.0, 1e-5, 1e-3])), check_termination=int(rng.choice([1, 1, 2, 5])))
    settings["abs_dua_tol"] = settings["abs_pri_tol"]
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, adaptive_rho=True, adaptive_rho_min=rho_min, adaptive_rho_max=rho_max,
            adaptive_rho_enable_clipping=clip, **settings)
    if prob.has_bounds():
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    dK, dP, dC1, dC2 = s.compute_sensitivity_autograd()
    s.set_sensitivity_matrices(dK, dP, dC1, dC2)
    s.set_x_ref(xref)
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.2, 1.2, batch)[None, :]
    checked = sorted({0, batch // 2, batch - 1})
    orcs = {}
    for b in checked:
        o = O.OraclePort(prob).load_problem(prob, settings)
        o.set_adaptive_rho(True, rho_min, rho_max, clip)
        o.set_sensitivity(dK, dP)
        o.set_x_ref(xref)
        orcs[b] = o
    ok, err, note = True, 0.0, ""
    try:
        for solve in range(3):
            xs = x0s * (1.0 - 0.35 * min(solve, 1))
            s.set_x0_batch(np.asfortranarray(xs)) if batch > 1 else s.set_x0(xs[:, 0])
            s.solve()
            sol, st, rho = s.get_solution_batch(), s.get_stats_batch(), s.get_rho_batch()
            for b in checked:
                o = orcs[b]
                o.set_x0(xs[:, b]); o.solve()
                ox, ou = o.solution()
                e = max(np.max(np.abs(sol["states"][:, :, b] - ox)) / max(np.max(np.abs(ox)), 1e-300), np.max(np.abs(sol["controls"][:, :, b] - ou)) / max(np.max(np.abs(ou)), 1e-300))
                err = max(err, e)
                same = st["iter"][b] == o.stats()["iter"] and st["status"][b] == o.stats()["status"] and abs(rho[b] - o.stats()["rho"]) < 1e-9 * rho[b]
                if not (same and e < 1e-8):
                    ok = False
                    note = f" [solve {solve} instance {b}: iter {st['iter'][b]}/{o.stats()['iter']} status {st['status'][b]}/{o.stats()['status']} rho {rho[b]:.6g}/{o.stats()['rho']:.6g} err {e:.1e}]"
    except pkg.TinyMPCError as ex:
        ok, note = False, f" [{str(ex)[:90]}]"
    lay = s.launch_info()["layout"]
    layouts[lay] = layouts.get(lay, 0) + 1
    worst = max(worst, err); fails += not ok
    print(f"case {case:3d}: nx={nx:2d} nu={nu:2d} N={N:2d} batch={batch:4d} clip={int(clip)} rho {prob.rho:.2f} in [{rho_min:.2f}, {rho_max:.1f}] per-knot ref={int(varying)} "
          f"check every {settings['check_termination']} tol {settings['abs_pri_tol']:g} -> layout {lay} {s.jit_info()[:44]} | rel err {err:.1e} {'ok' if ok else 'FAIL' + note} | {time.time() - t0:4.1f} s", flush=True)
    s.reset()
print(f"# {count} cases, by layout {layouts}, worst rel err {worst:.1e}, {fails} failure(s)")
sys.exit(1 if fails else 0)
