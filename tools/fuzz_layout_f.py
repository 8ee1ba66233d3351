"""Randomised shapes through layout F as the default of prepared handles (round 4): one instance and small batches, box path and
random cones / linear rows (both forms of the families), constant and per-knot tables, fdyn; launched solves (cold + warm), then -- for
one instance -- a resident session of closed-loop ticks whose results must equal launched ticks of a twin handle bit for bit, with the
references re-sent on some ticks. Against the oracle: iteration counts, statuses, trajectories.
  python tools/fuzz_layout_f.py [count] [seed] > gpurun_out/r04_fuzz_layout_f.txt
  FUZZ_ONLY_CASE=<n>: draw every case (the generator's stream stays the fuzzer's) but solve only case n, with a line per session tick"""
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
count = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ.pop("TINYMPC_LAYOUT", None)
only = int(os.environ["FUZZ_ONLY_CASE"]) if "FUZZ_ONLY_CASE" in os.environ else None
if "FUZZ_LAYOUT" in os.environ:  # (replaying a case on another kernel: FUZZ_LAYOUT=A / C / E)
    os.environ["TINYMPC_LAYOUT"] = os.environ["FUZZ_LAYOUT"]
fails, worst, on_f, sessions = 0, 0.0, 0, 0
for case in range(count):
    nxu = int(rng.integers(3, 17))
    nu = int(rng.integers(1, max(2, nxu // 3 + 1)))
    nx = nxu - nu
    N = int(rng.choice([rng.integers(6, 14), rng.integers(10, 30), rng.integers(30, 70), rng.integers(70, 130)]))  # (6 .. 13: the one-wavefront plans)
    batch = int(rng.choice([1, 1, 1, 3, 40]))
    fam = bool(rng.integers(0, 2))
    varying = bool(rng.integers(0, 2))
    A = 0.85 * np.eye(nx) + (0.12 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.2 * rng.standard_normal((nx, nu))
    prob = P.Problem("fuzzf", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(0.5, 3.0)), rng.standard_normal(nx))
    prob.x_min, prob.x_max = np.full(nx, -1.5), np.full(nx, 1.5)
    prob.u_min, prob.u_max = np.full(nu, -0.4), np.full(nu, 0.4)
    if varying:
        prob.x_ref, prob.u_ref = 0.05 * rng.standard_normal((nx, N)), 0.02 * rng.standard_normal((nu, N - 1))
    prob.fdyn = 0.01 * rng.standard_normal(nx) if rng.integers(0, 2) else None
    note = ""
    if fam:
        def cones(dim, n):
            a, q, c = [], [], []
            for _ in range(n):
                if dim < 2:
                    break
                qq = int(rng.integers(2, min(dim, 5) + 1))
                a.append(int(rng.integers(0, dim - qq + 1))); q.append(qq); c.append(float(rng.uniform(0.3, 1.5)))
            return a, q, c
        ax, qx, cx = cones(nx, int(rng.integers(0, 3)))
        au, qu, cu = cones(nu, int(rng.integers(0, 2)))
        nlx, nlu = int(rng.integers(0, 3)), int(rng.integers(0, 2))
        if len(ax) + len(au) + nlx + nlu == 0:
            nlx = 1
        prob.cones = dict(Acx=ax, qcx=qx, cx=cx, Acu=au, qcu=qu, cu=cu)
        prob.linear = dict(Alin_x=rng.standard_normal((nlx, nx)) / np.sqrt(nx), blin_x=rng.uniform(0.1, 0.6, nlx),
                           Alin_u=rng.standard_normal((nlu, nu)) / np.sqrt(nu), blin_u=rng.uniform(0.1, 0.3, nlu))
        note = f" cones {len(ax)}+{len(au)} rows {nlx}+{nlu}"
    settings = dict(max_iter=int(rng.integers(20, 90)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 3])))

    def handle():
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if prob.x_ref is not None:
            s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
        if fam:
            s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
        s.prepare()
        return s

    t0 = time.time()
    if only is not None and case != only:
        rng.standard_normal((nx, batch))  # (x0s: keeps the stream in step)
        continue
    s = handle()
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.3, 1.0, batch)[None, :]
    sample = sorted({0, batch // 2, batch - 1})
    orcs = {b: O.OraclePort(prob).load_problem(prob, settings) for b in sample}
    ok, e = True, 0.0
    for rnd in range(2):
        xs = x0s * (1.0 - 0.3 * rnd)
        s.set_x0_batch(xs) if batch > 1 else s.set_x0(xs[:, 0])
        s.solve()
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in sample:
            orcs[b].set_x0(xs[:, b]); orcs[b].solve()
            ox, ou = orcs[b].solution()
            e = max(e, np.max(np.abs(sol["states"][:, :, b] - ox)) / max(np.max(np.abs(ox)), 1e-300), np.max(np.abs(sol["controls"][:, :, b] - ou)) / max(np.max(np.abs(ou)), 1e-300))
            ok = ok and st["iter"][b] == orcs[b].stats()["iter"] and st["status"][b] == orcs[b].stats()["status"]
            if only is not None:
                dxs, dus = np.abs(sol["states"][:, :, b] - ox), np.abs(sol["controls"][:, :, b] - ou)
                print(f"    launched solve {rnd} instance {b}: iter {st['iter'][b]} / oracle {orcs[b].stats()['iter']}, status {st['status'][b]} / {orcs[b].stats()['status']}; "
                      f"states {dxs.max() / np.abs(ox).max():.2e} at {np.unravel_index(dxs.argmax(), dxs.shape)} controls {dus.max() / max(np.abs(ou).max(), 1e-300):.2e} at {np.unravel_index(dus.argmax(), dus.shape)} "
                      f"|u|max {np.abs(ou).max():.3e} residuals gpu {st['residuals'][:, b]} oracle {[orcs[b].stats()[k] for k in ('pri_x', 'dua_x', 'pri_u', 'dua_u')]}", flush=True)
    layout = s.launch_info()["layout"]
    on_f += layout == "F"
    sess = ""
    if batch == 1 and (layout == "F" or (layout == "C" and "FUZZ_LAYOUT" in os.environ)):  # resident session against launched ticks of a twin handle, bit for bit
        twin = handle()
        x = x0s[:, 0].copy()
        for h in (s, twin):
            h.reset_workspace()
        try:
            s.session_begin()
            resident = s.launch_info()["layout"]  # (the resident kernel's layout: C where layout F has none for the configuration)
            same, close = True, True
            for k in range(12):
                if prob.x_ref is not None and k % 3 == 1:  # references re-sent (shifted by one knot) on some ticks
                    xr = np.roll(prob.x_ref, -1, axis=1); ur = np.roll(prob.u_ref, -1, axis=1)
                    prob.x_ref, prob.u_ref = xr, ur
                    for h in (s, twin):
                        h.set_x_ref(xr); h.set_u_ref(ur)
                ua = s.session_step(x)
                ub = twin.mpc_step(x)[:, 0]
                same = same and np.array_equal(ua, ub) and s.get_stats()["iter"] == twin.get_stats()["iter"]
                close = close and np.allclose(ua, ub, rtol=1e-7, atol=1e-9)  # (a session on layout C against launches on F: other rounding, same answer)
                if only is not None:
                    print(f"    tick {k:2d}: session iter {s.get_stats()['iter']} status {s.get_stats()['status']} | launched iter {twin.get_stats()['iter']} status {twin.get_stats()['status']} | max |u0 diff| {np.max(np.abs(ua - ub)):.3e}", flush=True)
                x = prob.A @ x + prob.B @ ua + (prob.fdyn if prob.fdyn is not None else 0.0)
            s.session_end()
            sessions += 1
            sess = (" session(%s)==launched" % resident) if same else (" session on layout %s, launches on %s: last bits differ" % (resident, layout) if resident != layout else " SESSION DIFFERS")
            ok = ok and (same or (resident != layout and close))  # (a session on the generic latency kernel: documented, not bit-identical to layout F's launches)
            if resident != layout and not close:
                sess += " AND NOT EVEN CLOSE"
        except pkg.TinyMPCError as err:
            sess = " (no session: %s)" % str(err)[:60]
        twin.reset()
    if only is not None and batch == 1 and "FUZZ_REPEAT" in os.environ:  # the session part again and again: fresh handle + twin, ticks compared one by one
        nrep, nticks, nbad = int(os.environ["FUZZ_REPEAT"]), int(os.environ.get("FUZZ_TICKS", "12")), 0
        for rep in range(nrep):
            a, b = handle(), handle()
            x = x0s[:, 0].copy()
            a.session_begin()
            for k in range(nticks):
                ua = a.session_step(x)
                sa = a.get_stats()
                ub = b.mpc_step(x)[:, 0]
                sb = b.get_stats()
                if not (np.array_equal(ua, ub) and sa["iter"] == sb["iter"]):
                    nbad += 1
                    xa, xb = a.get_solution(), b.get_solution()
                    print(f"    rep {rep} tick {k}: session iter {sa['iter']} status {sa['status']} | launched iter {sb['iter']} status {sb['status']} | max |u0 diff| {np.max(np.abs(ua - ub)):.3e} "
                          f"| whole solution: states {np.max(np.abs(xa['states'] - xb['states'])):.3e} controls {np.max(np.abs(xa['controls'] - xb['controls'])):.3e} | u0 from get_solution vs step {np.max(np.abs(xa['controls'][:, 0] - ua)):.3e}", flush=True)
                    break
                x = prob.A @ x + prob.B @ ua + (prob.fdyn if prob.fdyn is not None else 0.0)
            a.session_end(); a.reset(); b.reset()
        print(f"    {nrep} repetitions of {nticks} ticks: {nbad} with a tick that differs", flush=True)
    worst = max(worst, e)
    bad = (not ok) or e > 1e-6
    fails += bad
    print(f"case {case:3d}: nx={nx:2d} nu={nu} N={N:3d} batch={batch:2d} per-knot refs={int(varying)} fdyn={int(prob.fdyn is not None)}{note} check every {settings['check_termination']} "
          f"-> layout {layout} {s.jit_info()[:44]} | trajectories {e:.1e} iterations {'equal' if ok else 'DIFFER'}{sess} | {time.time() - t0:5.1f} s{'   <-- FAIL' if bad else ''}", flush=True)
    s.reset()
print(f"# {count} cases ({on_f} on layout F, {sessions} sessions), worst rel err {worst:.1e}, {fails} failure(s)")
sys.exit(1 if fails else 0)
