"""A/B of the headline kernel (quadrotor N=50, 8,192 instances, 200 forced iterations) over variant builds of the
library: one child process per variant and round, rounds interleaved so that the box's drift hits all variants alike.
Usage (GPU box): python tools/headline_ab.py name1 name2 ... [--rounds 3]   (tools/bin/libtinympc_hip_<name>.so;
"base" = the in-tree library)"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(batch: int) -> None:
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    P = pkg.problems
    prob = P.quadrotor(50)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(P.quadrotor_batch_x0(batch))
    for _ in range(300):   # clock settle
        s.reset_workspace()
        s.solve_timed()
    ms = []
    for _ in range(60):
        s.reset_workspace()
        ms.append(s.solve_timed())
    print(json.dumps({"median": float(np.median(ms)), "min": float(np.min(ms)), "mean": float(np.mean(ms))}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(a.batch)
        sys.exit(0)
    res = {n: [] for n in a.names}
    for r in range(a.rounds):
        for n in a.names:
            env = dict(os.environ)
            if n != "base":
                env["TINYMPC_HIP_LIBRARY"] = os.path.join(ROOT, "tools", "bin", f"libtinympc_hip_{n}.so")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch)], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(n, "FAILED", out.stderr[-400:], flush=True)
                continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[n].append(d["median"])
            print(f"round {r} {n:12s} median {d['median']:.4f} ms  min {d['min']:.4f}  mean {d['mean']:.4f}", flush=True)
    print("# summary (median of round medians, ms)")
    for n in a.names:
        if res[n]:
            print(f"{n:12s} {float(np.median(res[n])):.4f}   all {['%.4f' % v for v in res[n]]}")
