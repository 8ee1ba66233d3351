import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
pkg = ge.load_package(); P = pkg.problems
prob = P.quadrotor(50)
for iters in (1, 3, 5):
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=iters)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0(prob.x0)
    xr = np.asfortranarray(np.random.default_rng(0).standard_normal((12, 50)) * 0.1)
    for mode in ("plain", "refs"):
        ms = []; wall = []
        for k in range(60):
            if mode == "refs":
                s.set_x_ref(xr)
            t0 = time.perf_counter()
            ms.append(s.solve_timed())
            wall.append(time.perf_counter() - t0)
        print(f"iters={iters} {mode:5s}: kernel {1e3*np.median(ms[10:]):6.1f} us, wall of solve_timed {1e6*np.median(wall[10:]):6.1f} us")
    s.reset()
