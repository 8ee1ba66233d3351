"""Dev tool: where layout F's iteration goes -- the shader clock (s_memtime, 100 MHz x ... ticks of the shader clock) at the phase
boundaries of iteration 10, per wavefront (TINYMPC_JIT_DEFS=-DTINY_F_STAMP=1; the states of the solution are overwritten).
    python tools/f_breakdown.py            (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    names = ["fwd pass 1", "fwd scan", "fwd pass 2", "check", "bwd pass 1", "bwd scan", "bwd pass 2"]
    for name, prob in (("quadrotor N=50", P.quadrotor(50)), ("quadrotor N=100", P.quadrotor(100)), ("cartpole N=20", P.cartpole(20)), ("rocket N=100", P.rocket(100))):
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, fdyn=prob.fdyn, max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if prob.cones: s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
        if prob.x_ref is not None: s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
        s.set_x0(prob.x0); s.prepare()
        ms = []
        for _ in range(6):
            s.reset_workspace(); ms.append(s.solve_timed())
        t = float(np.median(ms[2:]))
        info = s.jit_info()
        wpg = int(info.split("wpg=")[1].split()[0])
        st = s.get_solution()["states"].T.ravel()[: 8 * wpg].reshape(wpg, 8)
        print("%s  layout %s  %.3f us per iteration  (%s)" % (name, s.launch_info()["layout"], 5 * t, info[:70]))
        print("   wave " + " ".join("%11s" % n for n in names) + "       total")
        for w in range(wpg):
            d = np.diff(st[w])
            print("   %4d " % w + " ".join("%11.0f" % v for v in d) + "  %10.0f" % (st[w][-1] - st[w][0]))
        s.reset()
    sys.exit(0)
subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, TINYMPC_JIT_DEFS="-DTINY_F_STAMP=1", TINYMPC_LAYOUT="F"))
