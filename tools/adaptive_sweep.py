"""Adaptive rho (admm.cpp:117-174) on large batches: the run-time specialised layout D against k_admm_solve_adapt (layout A's
plan). Kernel time for 100 forced iterations (20 of them adapt). Usage (GPU box): python tools/adaptive_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
P = pkg.problems
ITERS = 100
for name, prob in (("cartpole N=20", P.cartpole(20, True)), ("quadrotor N=20", P.quadrotor(20)), ("quadrotor N=50", P.quadrotor(50)), ("quadrotor N=100", P.quadrotor(100))):
    for batch in (2048, 8192):
        row = [f"{name:16s} batch={batch:5d}"]
        rng = np.random.default_rng(1)
        x0s = np.asfortranarray(prob.x0[:, None] * rng.uniform(0.3, 1.2, (1, batch)) + 0.05 * rng.standard_normal((prob.nx, batch)))
        for jit in ("1", "0"):
            os.environ["TINYMPC_JIT"] = jit
            s = pkg.TinyMPC()
            s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0,
                    adaptive_rho=True, adaptive_rho_min=0.2, adaptive_rho_max=40.0)
            s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            dK, dP, dC1, dC2 = s.compute_sensitivity_autograd()
            s.set_sensitivity_matrices(dK, dP, dC1, dC2)
            s.set_x0_batch(x0s)
            ms = []
            for k in range(5):
                s.reset_workspace()
                ms.append(s.solve_timed())
            t = float(np.median(ms[1:]))
            row.append(f"{'layout ' + s.launch_info()['layout'] if jit == '1' else 'k_admm_solve_adapt'}: {t:8.3f} ms {batch * ITERS / t / 1e3:7.1f} M iters/s")
            s.reset()
        print(" | ".join(row), flush=True)
