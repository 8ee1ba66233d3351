"""Layout F (single-instance latency kernel): how many chunks? TINYMPC_F_CHUNKS = 32 (eight wavefronts, the default), 40, 48 (twelve),
64 (sixteen). Rocket landing N=100 with the families, quadrotor N=50 box path (forced onto layout F), one instance, 200 forced iterations;
parity of the same run against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
import pyoracle as O
pkg = g.load_package(); P = pkg.problems
for name, prob in (("rocket N=100", P.rocket(100)), ("rocket N=44", P.rocket(44)), ("quadrotor N=50", P.quadrotor(50))):
    st = dict(max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
    o = O.OraclePort(prob).load_problem(prob, st); o.set_x0(prob.x0); o.solve(); ox, ou = o.solution()
    for ch in ((int(a) for a in sys.argv[1].split(",")) if len(sys.argv) > 1 else (32, 40, 48, 64)):
        os.environ["TINYMPC_F_CHUNKS"] = str(ch)
        os.environ["TINYMPC_LAYOUT"] = "F"
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, fdyn=prob.fdyn, **st)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if prob.cones: s.set_cone_constraints(**prob.cones)
        if prob.linear: s.set_linear_constraints(**prob.linear)
        if prob.x_ref is not None: s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
        s.set_x0(prob.x0); s.prepare()
        ms = []
        for _ in range(6):
            s.reset_workspace(); ms.append(s.solve_timed())
        t = float(np.median(ms[2:]))
        sol = s.get_solution()
        err = max(np.max(np.abs(sol["states"] - ox)) / np.max(np.abs(ox)), np.max(np.abs(sol["controls"] - ou)) / np.max(np.abs(ou)))
        print(f"{name:16s} chunks<={ch}: layout {s.launch_info()['layout']} {s.jit_info()[:70]:70s} {5 * t:7.3f} us/iter  rel err {err:.1e}", flush=True)
        s.reset()
