"""BASELINE config 5 on ONE GPU (65,536 quadrotor N=50 instances x 200 FORCED iterations = 8 resident sets of wavefronts): the plain
kernel -- workgroups handed to the eight XCDs round-robin by the dispatcher, each XCD exactly one eighth of them -- against the
slot-refill variant (TINYMPC_REFILL=1: ONE resident set of wavefronts, every 16-lane row takes its next instance from an atomic
counter when its own has finished), i.e. wave-granular dynamic work assignment: a faster XCD (the shader clocks differ by 3.8 %,
profiles/r03_clock_hwid.txt) then simply takes more instances. VERDICT r4 item 5 / r3 item 5.   python tools/config5_refill_ab.py"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    prob = P.quadrotor(50)
    nb = int(sys.argv[2])
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=nb, rho=prob.rho, max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(P.quadrotor_batch_x0(nb))
    for _ in range(40):
        s.reset_workspace(); s.solve_timed()
    ms = []
    for _ in range(20):
        s.reset_workspace(); ms.append(s.solve_timed())
    u = s.get_first_controls_batch()
    import hashlib
    print(json.dumps({"median": float(np.median(ms)), "min": float(np.min(ms)), "kernel": s.jit_info(), "sha": hashlib.sha256(np.ascontiguousarray(u).tobytes()).hexdigest()[:12]}))
    sys.exit(0)
for nb in (65536, 16384):
    res = {}
    for rnd in range(3):
        for mode in ("0", "1"):
            env = dict(os.environ, TINYMPC_REFILL=mode)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(nb)], env=env, capture_output=True, text=True)
            if out.returncode:
                print(mode, "FAILED", out.stderr[-300:]); continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res.setdefault(mode, []).append(d)
            print(f"{nb:6d} instances  round {rnd}  TINYMPC_REFILL={mode}  median {d['median']:.4f} ms  min {d['min']:.4f}  first controls {d['sha']}  {d['kernel']}", flush=True)
    for mode, ds in res.items():
        m = float(np.median([d["median"] for d in ds]))
        print(f"# {nb} instances, TINYMPC_REFILL={mode}: {m:.4f} ms = {nb * 200 / m / 1e6:.1f} G iterations/s... fp64 frac {nb * 200 * 60848 / (m * 1e-3) / 78.6e12:.4f}")
