"""One cartpole solve: the sequence of examples/cartpole_example_one_solve.m (plus input bounds)."""
import numpy as np
from _common import TinyMPC

A = np.array([[1.0, 0.01, 0.0, 0.0], [0.0, 1.0, 0.039, 0.0], [0.0, 0.0, 1.002, 0.01], [0.0, 0.0, 0.458, 1.002]])
B = np.array([[0.0], [0.02], [0.0], [0.067]])
Q, R, N = np.diag([10.0, 1.0, 10.0, 1.0]), np.diag([1.0]), 20

solver = TinyMPC()
solver.setup(A, B, Q, R, N, rho=1.0, max_iter=100, abs_pri_tol=1e-4, abs_dua_tol=1e-4)
solver.set_bound_constraints([], [], -0.5, 0.5)  # scalars / vectors are expanded like the .m class does
solver.set_x0([0.5, 0.0, 0.0, 0.0])
solver.solve()
sol, stats = solver.get_solution(), solver.get_stats()
print(f"converged={bool(stats['solved'])} after {stats['iter']} iterations; first control u0 = {sol['controls'][0, 0]:+.4f}")
print("cart position over the horizon:", np.round(sol["states"][0], 3))
solver.reset()
