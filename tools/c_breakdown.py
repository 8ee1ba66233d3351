"""Dev tool: kernel time per iteration of the latency kernel for one quadrotor N=50 instance (library variant via
TINYMPC_HIP_LIBRARY, built with -DTINY_EXP=k: 4 = no carry scan, 5 = no pass 1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
os.environ["TINYMPC_LAYOUT"] = "C"
for N in (20, 50, 100):
    prob = P.quadrotor(N)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0(prob.x0)
    ms = []
    for _ in range(8):
        s.reset_workspace(); ms.append(s.solve_timed())
    print(f"{os.path.basename(os.environ.get('TINYMPC_HIP_LIBRARY', 'product')):32s} quadrotor N={N:3d}: {5 * float(np.median(ms[2:])):6.2f} us/iter", flush=True)
    s.reset()
