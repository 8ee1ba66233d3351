"""Slot refill (layout D, REFILL) against the plain kernel on a converging batch: bit-for-bit comparison of everything a solve
returns (two consecutive solves: cold, then warm), and the kernel time of both.
Usage (GPU box): python tools/refill_probe.py [instances] [check_termination] [max_iter] [horizon] > gpurun_out/refill_probe.txt
(horizon 50: the compiled-in quadrotor kernel; any other: its run-time specialisation)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
P = pkg.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ct = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50
prob = P.quadrotor(N)
rng = np.random.default_rng(0)
x0s = np.asfortranarray(P.quadrotor_batch_x0(B) * rng.uniform(0.05, 3.0, B)[None, :])


def run(refill: bool):
    os.environ["TINYMPC_REFILL"] = "1" if refill else "0"
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=max_iter, check_termination=ct)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(x0s)
    info = s.jit_info()
    out = []
    ms = []
    for k in range(2):  # cold, then warm from the first solve's state
        ms.append(s.solve_timed())
        sol = s.get_solution_batch()
        st = s.get_stats_batch()
        out.append((sol["states"].copy(), sol["controls"].copy(), st["iter"].copy(), st["status"].copy(), st["residuals"].copy()))
        s.set_x0_batch(np.asfortranarray(x0s * 0.9))
    t = []
    for _ in range(4):
        s.set_x0_batch(x0s)
        s.reset_workspace()
        t.append(s.solve_timed())
    s.reset()
    return info, out, ms, float(np.median(t[1:]))


ia, a, msa, ta = run(False)
ib, b, msb, tb = run(True)
print(f"{B} instances, N={N}, check_termination {ct}, max_iter {max_iter}: plain [{ia}] {ta:.3f} ms   refill [{ib}] {tb:.3f} ms   ratio {ta / tb:.2f}")
ok = True
for k, name in enumerate(("cold solve", "warm solve")):
    it = a[k][2].astype(float)
    same = [np.array_equal(x, y, equal_nan=True) for x, y in zip(a[k], b[k])]
    ok = ok and all(same)
    print(f"  {name}: iterations min {it.min():.0f} mean {it.mean():.1f} max {it.max():.0f}; solved {int((a[k][3] == 1).sum())}; "
          f"states/controls/iter/status/residuals identical: {same}")
    if not all(same):
        bad = np.nonzero(a[k][2] != b[k][2])[0]
        print("    first instances with different iteration counts:", bad[:10], a[k][2][bad[:10]], b[k][2][bad[:10]])
        d = np.abs(a[k][0] - b[k][0]).max(axis=(0, 1))
        print("    instances with different states:", int((d > 0).sum()), "max abs diff", float(d.max()))
print("RESULT:", "identical" if ok else "DIFFERENT")
sys.exit(0 if ok else 1)
