"""Dev tool: where the iteration of the latency kernel goes for BASELINE config 4 (rocket N=100, one instance).
Run once per library variant (TINYMPC_HIP_LIBRARY=tools/bin/libtinympc_hip_exp<k>.so, built with -DTINY_EXP=k)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
os.environ["TINYMPC_LAYOUT"] = "C"
for N in (100,):
    prob = P.rocket(N)
    for fam in (True, False):
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, fdyn=prob.fdyn, max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
        if fam:
            s.set_cone_constraints(**prob.cones)
            s.set_linear_constraints(**prob.linear)
        s.set_x0(prob.x0)
        ms = []
        for _ in range(8):
            s.reset_workspace(); ms.append(s.solve_timed())
        print(f"{os.environ.get('TINYMPC_HIP_LIBRARY', 'product'):50s} rocket N={N} families={fam}: {5 * float(np.median(ms[2:])):6.2f} us/iter", flush=True)
        s.reset()
