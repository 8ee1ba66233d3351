"""Large systems (64 < nx+nu <= 512) on the FP64 matrix cores (tinympc_solve_m.hip): kernel time, fraction of the FP64 peak,
algorithmic HBM bytes per second, and a parity check of the same run against the oracle.
Usage (GPU box): python tools/large_sweep.py > gpurun_out/large_sweep.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g  # noqa: E402
import pyoracle as O  # noqa: E402  (checker)

pkg = g.load_package()
P = pkg.problems
ITERS = 50
SHAPES = ((60, 10, 20, 4096), (96, 32, 20, 1024), (96, 32, 20, 4096), (96, 32, 20, 8192), (112, 16, 30, 4096),
          (130, 14, 20, 4096), (160, 32, 20, 4096), (160, 32, 20, 8192), (224, 32, 20, 4096), (300, 20, 20, 4096),
          (480, 32, 20, 4096))  # beyond 128 rows: two to four row tiles per wavefront, operator tiles streamed
if "--one" in sys.argv:
    SHAPES = ((96, 32, 20, 4096),)
if "--two" in sys.argv:  # enough tiles for two workgroups per CU
    SHAPES = ((96, 32, 20, 8192), (96, 32, 20, 16384), (160, 32, 20, 8192), (160, 32, 20, 16384), (224, 32, 20, 8192))
if "--big" in sys.argv:
    SHAPES = tuple(sh for sh in SHAPES if sh[0] + sh[1] > 128)
for nx, nu, N, batch in SHAPES:
    rng = np.random.default_rng(nx)
    A = np.eye(nx) * 0.98 + 0.015 * rng.standard_normal((nx, nx)) if nx + nu <= 128 else (0.95 if nx < 256 else 0.6) * np.eye(nx) + ((0.15 if nx < 256 else 0.1) / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.08 * rng.standard_normal((nx, nu))
    prob = P.Problem("large", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    x0s = np.asfortranarray(np.random.default_rng(1).standard_normal((nx, batch)))
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.set_bound_constraints(np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3))
    prob.x_min, prob.x_max, prob.u_min, prob.u_max = np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3)
    s.set_x0_batch(x0s)
    ms = []
    for k in range(4):
        s.reset_workspace()
        ms.append(s.solve_timed())
    t = float(np.median(ms[1:]))
    sample = [0, batch // 2, batch - 1]
    sol = s.get_solution_batch()
    orc = O.OraclePort(prob).load_problem(prob, dict(max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0))
    ox, ou, _, _, _ = orc.solve_batch(x0s[:, sample])
    err = max(np.max(np.abs(sol["states"][:, :, sample] - ox)) / np.max(np.abs(ox)), np.max(np.abs(sol["controls"][:, :, sample] - ou)) / np.max(np.abs(ou)))
    tflops = batch * ITERS * prob.flops_per_iteration() / (t * 1e-3) / 1e12
    gbs = batch * ITERS * prob.bytes_per_iteration() / (t * 1e-3) / 1e9
    print(f"nx={nx:3d} nu={nu:3d} N={N:3d} batch={batch:5d} | layout {s.launch_info()['layout']} | {t:8.3f} ms {1e3 * t / ITERS:7.2f} us/iter {batch * ITERS / t / 1e3:7.2f} M iters/s "
          f"| {tflops:6.2f} TFLOP/s = {tflops / 78.6:5.3f} of the FP64 peak | algorithmic state traffic {gbs:7.0f} GB/s | rel err vs oracle {err:.1e}", flush=True)
    s.reset()
