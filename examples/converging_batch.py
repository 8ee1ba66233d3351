"""A batch larger than the GPU holds at once whose instances need very different numbers of iterations: 30,000 quadrotors, some
almost at the origin, some far away, solved to a tolerance. The library runs it as ONE resident set of wavefronts and hands a
16-lane row the next instance as soon as its own has converged ("slot refill", reported by jit_info()); results are bit-identical
to the plain kernel's (TINYMPC_REFILL=0). The solves are queued on the handle's stream and waited for once."""
import os

import numpy as np
from _common import TinyMPC, problems

prob = problems.quadrotor(50)
batch = 30000
rng = np.random.default_rng(0)
x0s = problems.quadrotor_batch_x0(batch) * rng.uniform(0.05, 3.0, batch)[None, :]


def run(refill: bool):
    os.environ["TINYMPC_REFILL"] = "1" if refill else "0"
    solver = TinyMPC()
    solver.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=200)
    solver.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    solver.set_x0_batch(x0s)
    kernel = solver.jit_info()
    for _ in range(3):             # three cold solves, queued back to back
        solver.reset_workspace()
        solver.solve_queued()
    ms = solver.collect_kernel_ms()  # waits once
    out = solver.get_first_controls_batch(), solver.get_stats_batch()["iter"]
    solver.reset()
    return kernel, float(np.median(ms)), out


k0, t0, (u_plain, it_plain) = run(False)
k1, t1, (u_refill, it_refill) = run(True)
print(f"{batch} instances, iterations min/mean/max = {it_plain.min()}/{it_plain.mean():.0f}/{it_plain.max()}")
print(f"plain kernel  [{k0}]: {t0:.2f} ms per solve of the batch")
print(f"slot refill   [{k1}]: {t1:.2f} ms")
print("identical first controls and iteration counts:", bool(np.array_equal(u_plain, u_refill) and np.array_equal(it_plain, it_refill)))
