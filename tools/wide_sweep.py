"""Wide systems (16 < nx+nu <= 64, dynamic sizes as in types.hpp:16-17): kernel time of the batched solve, the fraction of
the FP64 vector roof it reaches, and a parity check of the same run against the oracle.
Usage (GPU box): python tools/wide_sweep.py > gpurun_out/wide_sweep.txt
  TINYMPC_HIP_LIBRARY=tools/bin/libtinympc_hip_<variant>.so selects a kernel variant (tools/build_variants.sh)."""
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


def system(nx, nu, N, seed=0):
    rng = np.random.default_rng(seed)
    A = np.eye(nx) + 0.03 * rng.standard_normal((nx, nx))
    B = 0.1 * rng.standard_normal((nx, nu))
    prob = P.Problem("wide", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
    prob.u_min, prob.u_max = np.full(nu, -0.3), np.full(nu, 0.3)
    prob.x_min, prob.x_max = np.full(nx, -2.0), np.full(nx, 2.0)
    return prob


print(f"# library: {os.environ.get('TINYMPC_HIP_LIBRARY', 'tinympc-matlab_amd/libtinympc_hip.so')}; {ITERS} forced iterations per solve; "
      f"TINYMPC_JIT={os.environ.get('TINYMPC_JIT', '1')} TINYMPC_LAYOUT={os.environ.get('TINYMPC_LAYOUT', '-')}")
print("# (12,4,50) (20,4,30) (24,8,30) (40,8,20) (48,16,20) are compiled into the library; the other shapes are specialised at run time")
for nx, nu, N, batch in ((12, 4, 50, 4096), (12, 4, 20, 8192), (20, 4, 30, 4096), (24, 8, 30, 4096), (24, 8, 30, 16384), (28, 4, 24, 4096), (40, 8, 20, 4096),
                         (48, 16, 20, 2048), (48, 16, 20, 8192), (36, 12, 16, 4096)):
    prob = system(nx, nu, N)
    rng = np.random.default_rng(1)
    x0s = np.asfortranarray(rng.standard_normal((nx, batch)))
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(x0s)
    info = s.launch_info()
    ms = []
    for k in range(6):
        s.reset_workspace()
        ms.append(s.solve_timed())
    t = float(np.median(ms[1:]))
    sample = [0, batch // 2, batch - 1]
    sol = s.get_solution_batch()
    orc = O.OraclePort(prob).load_problem(prob, dict(max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0))
    ox, ou, _, _, _ = orc.solve_batch(x0s[:, sample])
    err = max(np.max(np.abs(sol["states"][:, :, sample] - ox)) / np.max(np.abs(ox)), np.max(np.abs(sol["controls"][:, :, sample] - ou)) / np.max(np.abs(ou)))
    tflops = batch * ITERS * prob.flops_per_iteration() / (t * 1e-3) / 1e12
    print(f"nx={nx:3d} nu={nu:3d} N={N:3d} batch={batch:6d} | layout {info['layout']} lanes/instance {info['lanes_per_instance']:2d} LDS {info['lds_bytes']:6d} B "
          f"| {t:8.3f} ms {1e3 * t / ITERS:7.2f} us/iter {batch * ITERS / t / 1e3:8.1f} M iters/s | {tflops:6.2f} TFLOP/s = {tflops / PEAK:5.3f} of FP64 vector peak "
          f"| rel err vs oracle {err:.1e}", flush=True)
    s.reset()
