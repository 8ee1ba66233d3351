"""Session tick A/B on one box, alternating child processes: (a) the round-4 protocol -- mailbox in pinned host memory, plain stores +
system-scope release fence before the stamp (a library built from the previous commit: tools/bin/oldfence/libtinympc_hip.so, if it is
there), (b) round 5's write-through host stores with the mailbox still in host memory (TINYMPC_MAILBOX=host), (c) round 5's default:
the mailbox in fine-grained device memory behind the BAR. Ticks are driven through the Python mirror's fast path (the measurement
loop in libtinympc_bench.so is linked against the default library).   python tools/mailbox_ab.py  (GPU box)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OLD = os.path.join(ROOT, "tools", "bin", "oldfence", "libtinympc_hip.so")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    for name, prob, tol in (("quadrotor N=50", P.quadrotor(50), 1e-3), ("cartpole N=20", P.cartpole(20), 1e-3), ("cartpole N=10", P.cartpole(10), 1e-3)):
        for session in (True, False):
            s = pkg.TinyMPC()
            s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, max_iter=100, abs_pri_tol=tol, abs_dua_tol=tol)
            s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            if session: s.session_begin()
            meds, means, its = [], [], 0
            import ctypes as C
            L = pkg.load_library()
            dbg = np.zeros(4); brk = []
            for rep in range(3):
                x = prob.x0.copy(); d = []; its = 0
                for k in range(620):
                    t0 = time.perf_counter()
                    u = s.session_step(x) if session else s.mpc_step(x)[:, 0]
                    dt = time.perf_counter() - t0
                    if k >= 20:
                        d.append(1e6 * dt); its += s.get_stats()["iter"]
                        if session and hasattr(L, "tinympc_debug_tick_timing"):
                            L.tinympc_debug_tick_timing(s._h, dbg.ctypes.data_as(pkg._lib.c_double_p)); brk.append(dbg[:3].copy())
                    x = prob.A @ x + prob.B @ u
                meds.append(float(np.median(d))); means.append(float(np.mean(d)))
            print(f"{name:16s} {'session' if session else 'launch '} layout {s.launch_info()['layout']}  median {sorted(meds)[1]:6.2f} us  mean {sorted(means)[1]:6.2f}  iterations per tick {its / 600:.2f}" + (("   in the kernel: waited %.2f, iterations %.2f, write-out %.2f us" % tuple(np.median(np.array(brk), axis=0))) if brk and np.any(np.array(brk)) else ""), flush=True)
            if session: s.session_end()
            s.reset()
    sys.exit(0)
modes = (["old"] if os.path.exists(OLD) else []) + ["host", "device"]
for mode in modes * 2:
    env = dict(os.environ)
    env.pop("TINYMPC_MAILBOX", None); env.pop("TINYMPC_HIP_LIBRARY", None)
    if mode == "host": env["TINYMPC_MAILBOX"] = "host"
    if mode == "old": env["TINYMPC_HIP_LIBRARY"] = OLD
    print("---- %s" % {"old": "round-4 protocol (previous commit's library)", "host": "write-through stores, mailbox in host memory", "device": "write-through stores, mailbox in device memory"}[mode], flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env)
