"""Does the streamed-families kernel of the wide systems pay for its SIZE? Its sweeps are fully unrolled (register arrays indexed at compile
time) and the families' block is inlined once per batch of slots: 17,767 instructions (~125 KB) at N=30 against an instruction cache of
64 KB per pair of CUs. Same system, same families, horizons 6 .. 30: kernel time per slot and iteration, box path beside it.
    python tools/wide_fam_codesize_probe.py > gpurun_out/r05_wide_fam_codesize.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
nx, nu, batch, iters = 24, 8, 4096, 100
for N in (6, 8, 10, 14, 18, 22, 26, 30, 34):
    row = []
    for fam in (False, True):
        rng = np.random.default_rng(0)
        A = np.eye(nx) + 0.03 * rng.standard_normal((nx, nx)); B = 0.1 * rng.standard_normal((nx, nu))
        prob = P.Problem("wide", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=iters, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s.set_bound_constraints(np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3))
        if fam:
            s.set_cone_constraints(Acx=[0], qcx=[3], cx=[0.7], Acu=[], qcu=[], cu=[])
            s.set_linear_constraints(Alin_x=np.random.default_rng(1).standard_normal((2, nx)), blin_x=np.array([1.0, 1.5]), Alin_u=np.zeros((0, nu)), blin_u=np.zeros(0))
        s.set_x0_batch(np.asfortranarray(prob.x0[:, None] + 0.1 * np.random.default_rng(2).standard_normal((nx, batch))))
        ms = []
        for k in range(8):
            s.reset_workspace(); ms.append(s.solve_timed())
        row.append((float(np.median(ms[2:])), s.launch_info()["layout"], s.jit_info()[:58]))
        s.reset()
    (b, lb, jb), (f, lf, jf) = row
    ns = N - 1
    print(f"N={N:2d}: box {b:6.3f} ms ({1e3 * b / iters / ns:6.3f} us per slot-iteration, layout {lb})   families {f:6.3f} ms ({1e3 * f / iters / ns:6.3f} us per slot-iteration, layout {lf} {jf})", flush=True)
