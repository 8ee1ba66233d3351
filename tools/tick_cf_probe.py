"""Closed-loop tick, C caller (tinympc_bench_closed_loop): layout C against layout F, launched and resident, same box, same
process tree.   python tools/tick_cf_probe.py   (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    for name, prob in (("quadrotor N=50", P.quadrotor(50)), ("cartpole N=20", P.cartpole(20))):
        for session in (False, True):
            s = pkg.TinyMPC()
            s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, max_iter=100, abs_pri_tol=1e-3, abs_dua_tol=1e-3)
            s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            s.set_x0(prob.x0)
            if session: s.session_begin()
            r = s.bench_closed_loop(prob.A, prob.B, prob.x0, 420, 20, session=session)
            r2 = s.bench_closed_loop(prob.A, prob.B, prob.x0, 420, 20, session=session)
            print(f"{name:16s} {'session' if session else 'launch ':8s} layout {s.launch_info()['layout']}  median {r['us_per_tick_median']:6.2f} / {r2['us_per_tick_median']:6.2f} us  mean {r['us_per_tick']:6.2f} / {r2['us_per_tick']:6.2f}  iterations per tick {r['iterations_per_tick']:.2f} / {r2['iterations_per_tick']:.2f}", flush=True)
            if session: s.session_end()
            s.reset()
    sys.exit(0)
for lay in ("C", None, "C", None):
    env = dict(os.environ)
    env.pop("TINYMPC_LAYOUT", None)
    if lay: env["TINYMPC_LAYOUT"] = lay
    print("---- TINYMPC_LAYOUT=%s" % lay, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
