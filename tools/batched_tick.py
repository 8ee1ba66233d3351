"""The batched closed-loop tick (tinympc_mpc_step_batch: x0s in -> warm-started solves -> first controls out) by batch size, quadrotor
N=50, tol 1e-3: through pinned host memory (the kernel reads x0 and writes u0 itself; TINYMPC_ZERO_COPY_MAX=<batch>) and through the
copy engine (=0). Checks that both give the same controls."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
prob = P.quadrotor(50)
for B in (256, 1024, 4096, 8192, 32768):
    res = {}
    for mode, zc in (("copy engine", "0"), ("pinned memory", str(1 << 30))):
        os.environ["TINYMPC_ZERO_COPY_MAX"] = zc
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        x = np.asfortranarray(P.quadrotor_batch_x0(B)); d = []; us = []
        for k in range(60):
            t0 = time.perf_counter(); u0 = s.mpc_step(x); dt = time.perf_counter() - t0
            if k >= 10: d.append(dt * 1e6)
            us.append(u0.copy())
            x = np.asfortranarray(prob.A @ x + prob.B @ u0)
        res[mode] = (float(np.median(d)), np.array(us), s.launch_info()["layout"])
        s.reset()
    same = np.array_equal(res["copy engine"][1], res["pinned memory"][1])
    print(f"B={B:6d} layout {res['pinned memory'][2]}: copy engine {res['copy engine'][0]:8.1f} us | pinned memory {res['pinned memory'][0]:8.1f} us per tick | same controls: {same}", flush=True)
