"""Dev tool: where layout E's iteration goes on BASELINE config 4 (rocket N=100, cones + linear row + fdyn, 4,096 instances x 100
forced iterations). One child process per experiment build (TINYMPC_JIT_DEFS=-DTINY_E_EXP=k, tinympc_solve_e.hip): the
in-memory kernel cache is keyed without the extra options, so every variant needs a fresh process.
    python tools/e_breakdown.py            (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    N, B, it = int(sys.argv[2]), 4096, 100
    prob = P.rocket(N)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, fdyn=prob.fdyn, max_iter=it, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x_ref(prob.x_ref); s.set_u_ref(prob.u_ref)
    s.set_cone_constraints(**prob.cones); s.set_linear_constraints(**prob.linear)
    s.set_x0_batch(np.asfortranarray(prob.x0[:, None] * np.linspace(0.6, 1.2, B)[None, :]))
    ms = []
    for _ in range(6):
        s.reset_workspace(); ms.append(s.solve_timed())
    t = float(np.median(ms[2:]))
    if "TINY_E_EXP=9" in os.environ.get("TINYMPC_JIT_DEFS", ""):
        st = s.get_solution_batch(0, 1)["states"][:, :, 0].T.ravel()[:80].reshape(8, 10)  # [wavefront][stamp], cycles since the top of iteration 10
        names = ["top", "fwd pass 1", "barrier A", "carry f", "fwd pass 2", "families", "bwd pass 1", "barrier B", "carry b", "bwd pass 2"]
        print("phase durations of iteration 10, shader-clock ticks of s_memtime, per wavefront of workgroup 0 (%.3f ms per launch):" % t)
        print("   wave " + " ".join("%11s" % n for n in names[1:]) + "       total")
        for w in range(8):
            d = np.diff(st[w])
            print("   %4d " % w + " ".join("%11.0f" % v for v in d) + "  %10.0f" % (st[w][-1] - st[w][0]))
        s.reset()
        sys.exit(0)
    ck = float(np.abs(s.get_solution_batch(0, 64)["controls"]).sum())  # (a changed checksum = a build whose results differ)
    print("%-44s layout %s  %7.3f ms  %6.1f M iters/s  sum|u| %.12g  %s" % (os.environ.get("TINYMPC_JIT_DEFS", "(product)"), s.launch_info()["layout"], t, B * it / t / 1e3, ck, s.jit_info()[:60]), flush=True)
    s.reset()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--stamps":
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sys.argv[2] if len(sys.argv) > 2 else "100"], env=dict(os.environ, TINYMPC_JIT_DEFS="-DTINY_E_EXP=9"))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--defs":  # python tools/e_breakdown.py --defs "-DX=1" "-DY=2 -DZ=3" ...: one run per argument
    for defs in sys.argv[2:]:
        env = dict(os.environ)
        env.pop("TINYMPC_JIT_DEFS", None)
        if defs.strip():
            env["TINYMPC_JIT_DEFS"] = defs
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "100"], env=env)
    sys.exit(0)
N = sys.argv[1] if len(sys.argv) > 1 else "100"
for defs in (None, "-DTINY_E_EXP=1", "-DTINY_E_EXP=2", "-DTINY_E_EXP=3", "-DTINY_E_EXP=4", "-DTINY_E_EXP=5"):
    env = dict(os.environ)
    env.pop("TINYMPC_JIT_DEFS", None)
    if defs:
        env["TINYMPC_JIT_DEFS"] = defs
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", N], env=env)
