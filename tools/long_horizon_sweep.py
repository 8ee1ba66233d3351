"""Long horizons on the batched box path (16 lanes per instance): layout D's two plans (two wavefronts per SIMD; one
wavefront with 512 registers once the duals no longer fit) and layout E (the horizon cut across the eight wavefronts of a
workgroup, two per SIMD at any horizon) against layouts B and A. Kernel time, fraction of the FP64 vector
roof, parity of the same run against the oracle.
Usage (GPU box): python tools/long_horizon_sweep.py > gpurun_out/long_horizon_sweep.txt"""
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
PEAK = 78.6
ITERS = 100
BATCH = 8192



def system(nx, nu, N, seed=0):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) + 0.03 * rng.standard_normal((nx, nx))
    B = 0.1 * rng.standard_normal((nx, nu))
    prob = P.Problem("wide", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
    return prob


print(f"# {ITERS} forced iterations per solve; (12,4): quadrotor, other sizes: random systems as in tools/wide_sweep.py")
SHAPES = ((12, 4, 40, 8192), (12, 4, 50, 8192), (12, 4, 60, 8192), (12, 4, 75, 8192), (12, 4, 100, 8192), (12, 4, 110, 8192), (12, 4, 125, 8192), (12, 4, 200, 8192),
          (6, 3, 100, 8192), (4, 1, 200, 8192), (24, 8, 30, 4096), (24, 8, 60, 4096), (24, 8, 80, 4096), (48, 16, 20, 2048), (48, 16, 40, 2048), (48, 16, 60, 2048))
if "--narrow" in sys.argv:
    SHAPES = tuple(sh for sh in SHAPES if sh[0] + sh[1] <= 16)
for nx, nu, N, BATCH in SHAPES:
    prob = P.quadrotor(N) if (nx, nu) == (12, 4) else P.cartpole(N, True) if (nx, nu) == (4, 1) else system(nx, nu, N)
    rng = np.random.default_rng(1)
    x0s = np.asfortranarray(prob.x0[:, None] + 0.1 * rng.standard_normal((prob.nx, BATCH)))
    LAYOUTS = (None, "D", "E", "B", "A")
    for a in sys.argv:
        if a.startswith("--layouts="):
            LAYOUTS = tuple(x for x in a.split("=", 1)[1].split(","))
    for layout in LAYOUTS:
        if layout == "E" and nx + nu > 16:
            continue
        if layout == "A" and "--narrow" in sys.argv:
            continue
        if layout:
            os.environ["TINYMPC_LAYOUT"] = layout
        else:
            os.environ.pop("TINYMPC_LAYOUT", None)
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=BATCH, rho=prob.rho, max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x0_batch(x0s)
        s.prepare()  # (layout E and the variants of layout D are decided -- and built -- here rather than at the first launch)
        info = s.launch_info()
        if layout and info["layout"] != layout:
            s.reset()
            continue
        ms = []
        for k in range(5):
            s.reset_workspace()
            ms.append(s.solve_timed())
        t = float(np.median(ms[1:]))
        sample = [0, BATCH // 2, BATCH - 1]
        sol = s.get_solution_batch()
        orc = O.OraclePort(prob).load_problem(prob, dict(max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0))
        ox, ou, _, _, _ = orc.solve_batch(x0s[:, sample])
        err = max(np.max(np.abs(sol["states"][:, :, sample] - ox)) / np.max(np.abs(ox)), np.max(np.abs(sol["controls"][:, :, sample] - ou)) / np.max(np.abs(ou)))
        tflops = BATCH * ITERS * prob.flops_per_iteration() / (t * 1e-3) / 1e12
        print(f"nx={nx:2d} nu={nu:2d} N={N:3d} batch={BATCH:5d} | layout {info['layout']} workgroups {info['workgroups']:5d} LDS {info['lds_bytes']:6d} B | {t:8.3f} ms "
              f"{BATCH * ITERS / t / 1e3:8.1f} M iters/s | {tflops:6.2f} TFLOP/s = {tflops / PEAK:5.3f} of FP64 vector peak | rel err vs oracle {err:.1e}", flush=True)
        s.reset()
