"""Batched mode: 8,192 independent quadrotor instances (nx=12, nu=4, N=50) in one kernel launch."""
import numpy as np
from _common import TinyMPC, problems

quad = problems.quadrotor(50)
batch = 8192
solver = TinyMPC()
solver.setup(quad.A, quad.B, quad.Q, quad.R, quad.N, batch=batch, rho=quad.rho, max_iter=200, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
solver.set_bound_constraints(quad.x_min, quad.x_max, quad.u_min, quad.u_max)
solver.set_x0_batch(problems.quadrotor_batch_x0(batch))
ms = solver.solve_timed()
stats = solver.get_stats_batch()
iters = stats["iter"]
print(f"{batch} solves in {ms:.2f} ms ({batch / ms / 1e3:.2f} M solves/s); iterations min/median/max = "
      f"{iters.min()}/{int(np.median(iters))}/{iters.max()}; converged {int((stats['status'] == 1).sum())}")
print("first controls of instance 0:", np.round(solver.get_first_controls_batch()[:, 0], 4))
solver.reset()
