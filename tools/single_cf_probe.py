"""Single-instance box path: the latency kernel (layout C) against the structure-specialised one (layout F), microseconds per ADMM
iteration over 200 forced iterations (kernel time, HIP events).   python tools/single_cf_probe.py   (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    probs = (("quadrotor N=10", P.quadrotor(10)), ("quadrotor N=20", P.quadrotor(20)), ("quadrotor N=50", P.quadrotor(50)),
             ("quadrotor N=100", P.quadrotor(100)), ("cartpole N=20", P.cartpole(20)))
    if os.environ.get("TINYMPC_F_SHORT"):
        probs = tuple(("quadrotor N=%d" % n, P.quadrotor(n)) for n in (8, 10, 13, 17, 20, 25)) + tuple(("cartpole N=%d" % n, P.cartpole(n)) for n in (10, 20, 25))
    elif os.environ.get("TINYMPC_F_S"):
        probs = tuple(("quadrotor N=%d" % n, P.quadrotor(n)) for n in (10, 15, 20, 30, 40, 50, 65, 80, 100, 130)) + (("cartpole N=20", P.cartpole(20)), ("cartpole N=40", P.cartpole(40)))
    for name, prob in probs:
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        if prob.x_ref is not None: s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
        s.set_x0(prob.x0); s.prepare()
        ms = []
        for _ in range(8):
            s.reset_workspace(); ms.append(s.solve_timed())
        t = float(np.median(ms[2:]))
        print(f"{name:16s} layout {s.launch_info()['layout']} {5 * t:7.3f} us/iter   {s.jit_info()[:80]}", flush=True)
        s.reset()
    sys.exit(0)
CONFIGS = (("C", {}), ("F", {}), ("F", {"TINYMPC_F_CHUNKS": "32"}))
if len(sys.argv) > 1 and sys.argv[1] == "--chunks":  # python tools/single_cf_probe.py --chunks 12,20,24
    CONFIGS = tuple(("F", {"TINYMPC_F_CHUNKS": c, "TINYMPC_BUILTIN": "0"}) for c in sys.argv[2].split(","))
if len(sys.argv) > 1 and sys.argv[1] == "--slots":  # python tools/single_cf_probe.py --slots 2,3,4,5: chunk length S directly
    CONFIGS = tuple(("F", {"TINYMPC_F_S": c, "TINYMPC_BUILTIN": "0"}) for c in sys.argv[2].split(","))
if len(sys.argv) > 1 and sys.argv[1] == "--defs":  # python tools/single_cf_probe.py --defs "-DX=0" "-DX=1": experiment builds of layout F
    CONFIGS = tuple(("F", {"TINYMPC_JIT_DEFS": d}) for d in sys.argv[2:])
if len(sys.argv) > 1 and sys.argv[1] == "--short":  # short horizons: chunk lengths that leave one wavefront (<= 4 chunks) against the default plan
    CONFIGS = (("C", {"TINYMPC_F_SHORT": "1"}), ("F", {"TINYMPC_F_SHORT": "1", "TINYMPC_BUILTIN": "0"})) + tuple(("F", {"TINYMPC_F_S": c, "TINYMPC_F_SHORT": "1", "TINYMPC_BUILTIN": "0"}) for c in ("3", "4", "5", "6", "7"))
for lay, extra in CONFIGS:
    env = dict(os.environ, TINYMPC_LAYOUT=lay, **extra)
    print("---- TINYMPC_LAYOUT=%s %s" % (lay, extra), flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
