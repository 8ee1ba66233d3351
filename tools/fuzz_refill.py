"""Randomised shapes through the slot-refill variant of layout D (compiled-in and run-time specialised kernels, both register
plans, ragged batches, check_termination 1..4, max_iter that the check does not divide): everything two consecutive solves return
(cold, then warm) must be bit-identical to the plain kernel's (TINYMPC_REFILL=0), and a sample must match the oracle.
  python tools/fuzz_refill.py [count] [seed] > gpurun_out/fuzz_refill.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g  # noqa: E402
import pyoracle as O  # noqa: E402  (checker)

pkg = g.load_package()
P = pkg.problems
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
fails = 0
taken = 0
for case in range(count):
    kind = case % 4
    if kind == 0:      # the compiled-in quadrotor kernel
        prob = P.quadrotor(50)
    elif kind == 1:    # run-time specialisations of the quadrotor, two wavefronts per SIMD
        prob = P.quadrotor(int(rng.integers(8, 50)))
    elif kind == 2:    # the 512-register plan (one wavefront per SIMD: the resident set is half as large)
        prob = P.quadrotor(int(rng.integers(60, 105)))
    else:              # random small systems
        nx, nu = int(rng.integers(2, 13)), int(rng.integers(1, 5))
        N = int(rng.integers(6, 45))
        A = np.eye(nx) + 0.05 * rng.standard_normal((nx, nx))
        Bm = 0.2 * rng.standard_normal((nx, nu))
        prob = P.Problem("rand", A, Bm, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, float(rng.uniform(1, 5)), rng.standard_normal(nx))
        prob.u_min, prob.u_max = np.full(nu, -0.4), np.full(nu, 0.4)
        prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    nx = prob.A.shape[0]
    B = int(rng.integers(4200, 14000))
    ct = int(rng.integers(1, 5))
    max_iter = int(rng.integers(8, 70))
    tol = float(10.0 ** rng.uniform(-3.5, -1.5))
    if prob.name == "rand":
        x0s = np.asfortranarray(rng.standard_normal((nx, B)) * rng.uniform(0.02, 1.5, B)[None, :])
    else:
        x0s = np.asfortranarray(P.quadrotor_batch_x0(B) * rng.uniform(0.05, 3.0, B)[None, :])
    settings = dict(abs_pri_tol=tol, abs_dua_tol=tol, max_iter=max_iter, check_termination=ct)
    t0 = time.time()
    if "FUZZ_ONLY_CASE" in os.environ and case != int(os.environ["FUZZ_ONLY_CASE"]):
        rng.choice(B, size=6, replace=False)  # (the sample: keeps the generator's stream in step)
        continue
    if "FUZZ_ONLY_CASE" in os.environ:  # how well conditioned is the case? the LQR cache against the oracle's, the plant's spectral radius
        os.environ["TINYMPC_REFILL"] = "0"
        h = pkg.TinyMPC(); h.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, **settings)
        o = O.OraclePort(prob).load_problem(prob, settings)
        gc = h.get_cache()
        print("    spectral radius of A %.4f, max|Pinf| %.3e; cache vs oracle:" % (np.abs(np.linalg.eigvals(prob.A)).max(), np.abs(np.asarray(o.get("Pinf"))).max()),
              ", ".join("%s %.1e" % (k, np.max(np.abs(np.asarray(gc[k]) - np.asarray(o.get(k)))) / np.max(np.abs(np.asarray(o.get(k))))) for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt")), flush=True)
        print("    Riccati steps: kernel %d, oracle %s" % (int(h.debug_setup_timing()["riccati_steps"]), o.stats().get("riccati_iters")), flush=True)
        h.reset()
    got, info = {}, {}
    for mode in ("0", "1"):
        os.environ["TINYMPC_REFILL"] = mode
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **settings)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x0_batch(x0s)
        info[mode] = (s.launch_info()["layout"], s.jit_info())
        out = []
        for k in range(2):
            s.solve()
            sol, st = s.get_solution_batch(), s.get_stats_batch()
            out.append((sol["states"].copy(), sol["controls"].copy(), st["iter"].copy(), st["status"].copy(), st["residuals"].copy()))
            s.set_x0_batch(np.asfortranarray(0.93 * x0s))
        got[mode] = out
        s.reset()
    refill = "slot-refill" in info["1"][1]
    taken += refill
    same = all(np.array_equal(a, b, equal_nan=True) for k in range(2) for a, b in zip(got["0"][k], got["1"][k]))
    sample = rng.choice(B, size=6, replace=False)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    ox, ou, oit, ost, _ = orc.solve_batch(x0s[:, sample])
    err = max(np.max(np.abs(got["1"][0][0][:, :, sample] - ox)) / max(np.max(np.abs(ox)), 1e-300), np.max(np.abs(got["1"][0][1][:, :, sample] - ou)) / max(np.max(np.abs(ou)), 1e-300))
    its_ok = np.array_equal(got["1"][0][2][sample], oit)
    ok = same and its_ok and err < 1e-8
    fails += not ok
    it = got["1"][0][2]
    print(f"case {case:3d}: nx={nx:2d} nu={prob.B.shape[1]} N={prob.N:3d} batch={B:6d} ct={ct} max_iter={max_iter:2d} tol={tol:.1e} -> layout {info['1'][0]} "
          f"{'slot-refill' if refill else 'NOT refilled (' + info['1'][1][:40] + ')'} | iterations {it.min()}..{it.max()} | identical to plain: {same} | "
          f"oracle: iterations {'equal' if its_ok else 'DIFFER'}, rel err {err:.1e} | {time.time() - t0:4.1f} s {'' if ok else ' <-- FAIL'}", flush=True)
print(f"# {count} cases, {taken} on the slot-refill variant, {fails} failure(s)")
sys.exit(1 if fails else 0)
