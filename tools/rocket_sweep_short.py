"""Dev tool: the rocket landing at short horizons (the reference example's own N = 10, and 20, 28), 4,096 instances x 100 forced
iterations: layout D's families variant (TINYMPC_LAYOUT=D) against the default plan (round 4: layout E's uncut form)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import __graft_entry__ as g
    import pyoracle as O
    pkg = g.load_package(); P = pkg.problems
    N, B, it = int(sys.argv[2]), 4096, 100
    prob = P.rocket(N)
    s = pkg.TinyMPC()
    st = dict(max_iter=it, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, fdyn=prob.fdyn, **st)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
    s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
    x0s = np.asfortranarray(prob.x0[:, None] * np.linspace(0.6, 1.2, B)[None, :])
    s.set_x0_batch(x0s)
    ms = []
    for _ in range(6):
        s.reset_workspace(); ms.append(s.solve_timed())
    t = float(np.median(ms[2:]))
    o = O.OraclePort(prob).load_problem(prob, st); o.set_x0(x0s[:, B - 1]); o.solve()
    sol = s.get_solution_batch(B - 1, 1)
    err = np.max(np.abs(sol["controls"][:, :, 0] - o.solution()[1])) / np.max(np.abs(o.solution()[1]))
    print("N=%3d %-10s layout %s  %7.3f ms  %7.1f M iters/s  rel err %.1e  %s" % (N, os.environ.get("TINYMPC_LAYOUT", "(default)"), s.launch_info()["layout"], t, B * it / t / 1e3, err, s.jit_info()[:70]), flush=True)
    s.reset()
    sys.exit(0)
for N in (10, 20, 28):
    for layout in ("D", None):
        env = dict(os.environ)
        env.pop("TINYMPC_LAYOUT", None)
        if layout:
            env["TINYMPC_LAYOUT"] = layout
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(N)], env=env)
