"""Layout E on BASELINE config 4 as a batch (rocket N=100 x 4,096, 100 forced iterations): plans with three / four wavefronts per SIMD
(TINYMPC_E_WPG=12 / 16, run-time specialised) against the compiled-in plan (eight wavefronts per workgroup, two per SIMD).
    python tools/e_wps_probe.py  (GPU box)"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    rk = P.rocket(100); B = 4096
    s = pkg.TinyMPC()
    s.setup(rk.A, rk.B, rk.Q, rk.R, rk.N, batch=B, rho=rk.rho, fdyn=rk.fdyn, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=100, check_termination=1)
    s.set_bound_constraints(rk.x_min, rk.x_max, rk.u_min, rk.u_max); s.set_x_ref(rk.x_ref); s.set_u_ref(rk.u_ref)
    s.set_cone_constraints(**rk.cones); s.set_linear_constraints(**rk.linear)
    s.set_x0_batch(np.asfortranarray(rk.x0[:, None] * np.linspace(0.6, 1.2, B)[None, :]))
    for _ in range(60):
        s.reset_workspace(); s.solve_timed()
    ms = []
    for _ in range(12):
        s.reset_workspace(); ms.append(s.solve_timed())
    import hashlib
    u = s.get_first_controls_batch()
    print(json.dumps({"ms": float(np.median(ms)), "layout": s.launch_info()["layout"], "jit": s.jit_info(), "wg": s.launch_info()["workgroups"],
                      "sha": hashlib.sha256(np.ascontiguousarray(np.round(u, 9)).tobytes()).hexdigest()[:10]}))
    sys.exit(0)
variants = [("compiled-in (8 per workgroup, 2 per SIMD)", {}),
            ("TINYMPC_E_WPG=12 (3 per SIMD)", {"TINYMPC_E_WPG": "12", "TINYMPC_BUILTIN": "0"}),
            ("TINYMPC_E_WPG=12 forced", {"TINYMPC_E_WPG": "12", "TINYMPC_BUILTIN": "0", "TINYMPC_E_FORCE": "1"}),
            ("TINYMPC_E_WPG=12 forced, knot-per-lane without d in registers", {"TINYMPC_E_WPG": "12", "TINYMPC_BUILTIN": "0", "TINYMPC_E_FORCE": "1", "TINYMPC_E_LDS": "k"}),
            ("TINYMPC_E_WPG=16 forced", {"TINYMPC_E_WPG": "16", "TINYMPC_BUILTIN": "0", "TINYMPC_E_FORCE": "1"}),
            ("TINYMPC_E_WPG=16 forced, no d in registers", {"TINYMPC_E_WPG": "16", "TINYMPC_BUILTIN": "0", "TINYMPC_E_FORCE": "1", "TINYMPC_E_LDS": "k"})]
for name, env in variants + variants[:1]:
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), capture_output=True, text=True)
    line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    try:
        d = json.loads(line)
        print(f"{name:68s} layout {d['layout']}  {d['ms']:.3f} ms  {4096 * 100 / d['ms'] / 1e3:7.1f} M iters/s  workgroups {d['wg']}  u0 {d['sha']}  {d['jit']}", flush=True)
    except Exception:
        print(f"{name:68s} FAILED: {out.stderr[-300:]}", flush=True)
