"""Layout F with the families: one knot per lane (KFamilies, default) against one element per lane (TINYMPC_F_KFAM=0), and the chunk
plan (TINYMPC_F_CHUNKS). Rocket landing N=100 / 44 / 20, one instance, 200 forced iterations, microseconds per iteration; parity of
every run against the oracle.   python tools/f_kfam_probe.py   (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import __graft_entry__ as g
    import pyoracle as O
    pkg = g.load_package(); P = pkg.problems
    for N in (100, 44, 20):
        prob = P.rocket(N)
        st = dict(max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
        o = O.OraclePort(prob).load_problem(prob, st); o.set_x0(prob.x0); o.solve(); ox, ou = o.solution()
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, fdyn=prob.fdyn, **st)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
        s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
        s.set_x0(prob.x0); s.prepare()
        ms = []
        for _ in range(8):
            s.reset_workspace(); ms.append(s.solve_timed())
        t = float(np.median(ms[2:]))
        sol = s.get_solution()
        err = max(np.max(np.abs(sol["states"] - ox)) / np.max(np.abs(ox)), np.max(np.abs(sol["controls"] - ou)) / np.max(np.abs(ou)))
        print(f"rocket N={N:3d}  layout {s.launch_info()['layout']} {5 * t:7.3f} us/iter  rel err {err:.1e}  {s.jit_info()[:84]}", flush=True)
        s.reset()
    sys.exit(0)
CONFIGS = ({}, {"TINYMPC_F_KFAM": "0"}, {"TINYMPC_F_CHUNKS": "16"}, {"TINYMPC_F_CHUNKS": "16", "TINYMPC_F_KFAM": "0"}, {"TINYMPC_F_CHUNKS": "24"}, {"TINYMPC_F_CHUNKS": "12"})
if len(sys.argv) > 1 and sys.argv[1] == "--short":
    CONFIGS = ({}, {"TINYMPC_F_S": "3", "TINYMPC_F_KFAM": "0"}, {"TINYMPC_F_S": "3"}, {"TINYMPC_F_S": "5", "TINYMPC_F_KFAM": "0"}, {"TINYMPC_F_S": "5"})
for extra in CONFIGS:
    env = dict(os.environ, **extra)
    print("---- %s" % (extra or "default"), flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
