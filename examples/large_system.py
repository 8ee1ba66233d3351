"""A large system: 160 states, 32 inputs (beyond the 16 / 32 / 64 rows that fit the lanes of a wavefront), 2,048 instances in one launch.
The sweep step of sixteen instances at once is a GEMM on the FP64 matrix cores (layout M); the LQR cache comes from the
multi-launch Riccati precompute. Nothing changes for the caller."""
import time

import numpy as np
from _common import TinyMPC

nx, nu, N, batch = 160, 32, 20, 2048
rng = np.random.default_rng(0)
A = 0.95 * np.eye(nx) + (0.15 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
B = 0.08 * rng.standard_normal((nx, nu))
Q, R = np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu))
solver = TinyMPC()
t0 = time.perf_counter()
solver.setup(A, B, Q, R, N, batch=batch, rho=2.0, max_iter=100, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
print(f"setup (Riccati precompute on the device, nx={nx}): {1e3 * (time.perf_counter() - t0):.0f} ms")
solver.set_bound_constraints(np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3))
solver.set_x0_batch(rng.standard_normal((nx, batch)) * np.linspace(0.05, 1.0, batch)[None, :])
ms = solver.solve_timed()
stats = solver.get_stats_batch()
iters = stats["iter"]
print(f"{batch} solves in {ms:.2f} ms on layout {solver.launch_info()['layout']}; iterations min/median/max = {iters.min()}/{int(np.median(iters))}/{iters.max()}; "
      f"converged {int((stats['status'] == 1).sum())}")
u0 = solver.get_first_controls_batch()
print("first controls of the last instance (first 4 inputs):", np.round(u0[:4, -1], 4), "| all within the bounds:", bool(np.all(np.abs(u0) <= 0.3 + 1e-12)))
solver.reset()
