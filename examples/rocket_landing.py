"""Rocket soft landing (examples/rocket_landing_constraints.m): affine dynamics (gravity), box bounds, a glide-slope
cone on the position and a thrust-pointing cone on the input, reference shifted along a straight line every tick."""
import numpy as np
from _common import TinyMPC, problems

rk = problems.rocket(10, with_linear=False)  # the reference's NHORIZON
solver = TinyMPC()
solver.setup(rk.A, rk.B, rk.Q, rk.R, rk.N, rho=rk.rho, fdyn=rk.fdyn, max_iter=100, abs_pri_tol=2e-3, abs_dua_tol=1e-4)
solver.set_bound_constraints(rk.x_min, rk.x_max, rk.u_min, rk.u_max)
solver.set_cone_constraints(**rk.cones)

x, x_start, x_goal, n_total = rk.x0.copy(), np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5]), np.zeros(6), 100
for k in range(n_total - rk.N):
    ref = np.stack([x_start + (x_goal - x_start) * min(i + k, n_total - 1) / (n_total - 1) for i in range(rk.N)], axis=1)
    solver.set_x0(x)
    solver.set_x_ref(ref)
    solver.solve()
    u = solver.get_solution()["controls"][:, 0]
    x = rk.A @ x + rk.B @ u + rk.fdyn
    if k % 15 == 0:
        print(f"tick {k:3d}: altitude {x[2]:6.2f} m, thrust cone slack {0.25 * u[2] - np.hypot(u[0], u[1]):+.3f}, iterations {solver.get_stats()['iter']}")
print("final position", np.round(x[:3], 2), "velocity", np.round(x[3:], 2))
solver.reset()
