"""Kernel time of the solve layouts over batch sizes (quadrotor N=50, 200 forced iterations).
Usage (GPU box): python tools/layout_sweep.py [--horizon 50] > gpurun_out/layout_sweep.txt"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--horizon", type=int, default=50)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--layouts", type=str, default="A,B,C,D")
ap.add_argument("--batches", type=str, default="1,4,16,64,128,256,512,1024,2048,4096,8192")
args = ap.parse_args()
pkg = g.load_package()
P = pkg.problems
prob = P.quadrotor(args.horizon)
print(f"# quadrotor N={args.horizon}, {args.iters} iterations per solve; kernel ms (HIP events), us/iter, M instance-iters/s")
for batch in [int(b) for b in args.batches.split(",")]:
    row = [f"{batch:6d}"]
    for layout in args.layouts.split(","):
        os.environ["TINYMPC_LAYOUT"] = layout
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=args.iters, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x0_batch(P.quadrotor_batch_x0(batch))
        got = s.launch_info()["layout"]
        for _ in range(2):
            s.reset_workspace()
            s.solve_timed()
        ms = []
        for _ in range(5):
            s.reset_workspace()
            ms.append(s.solve_timed())
        t = float(np.median(ms))
        row.append(f"{got}: {t:8.3f} ms {1e3 * t / args.iters:7.2f} us/it {batch * args.iters / t / 1e3:8.1f} M/s")
        s.reset()
    print(" | ".join(row), flush=True)
