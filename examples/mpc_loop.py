"""Closed loop: one warm-started solve per control tick (examples/cartpole_example_mpc.m), here for a batch of
cartpoles started from different angles -- one `mpc_step` call (x in, first controls out) per tick for all of them."""
import time

import numpy as np
from _common import TinyMPC

A = np.array([[1.0, 0.01, 0.0, 0.0], [0.0, 1.0, 0.039, 0.0], [0.0, 0.0, 1.002, 0.01], [0.0, 0.0, 0.458, 1.002]])
B = np.array([[0.0], [0.02], [0.0], [0.067]])
Q, R, N = np.diag([10.0, 1.0, 10.0, 1.0]), np.diag([1.0]), 20
batch, ticks = 32, 300

solver = TinyMPC()
solver.setup(A, B, Q, R, N, batch=batch, rho=1.0, max_iter=50, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
solver.set_bound_constraints([], [], -2.0, 2.0)
x = np.zeros((4, batch))
x[0] = np.linspace(-0.6, 0.6, batch)   # cart offset
x[2] = np.linspace(0.15, -0.15, batch)  # pole angle
t0 = time.perf_counter()
for _ in range(ticks):
    u = solver.mpc_step(x)          # (nu, batch)
    x = A @ x + B @ u
dt = time.perf_counter() - t0
print(f"{ticks} ticks x {batch} cartpoles: {1e6 * dt / ticks:.1f} us per tick, final |x| max = {np.abs(x).max():.2e}")
solver.reset()
