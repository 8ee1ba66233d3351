"""The streamed families of layout D's wide kernels (tinympc_solve_dwide.h, round 5): variants through TINYMPC_JIT_DEFS, same box.
   nx=24 nu=8 N=30 x 4,096 and nx=48 nu=16 N=20 x 2,048, one state cone + two linear rows on the states, 100 forced iterations."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package(); P = pkg.problems
    out = []
    for nx, nu, N, batch in ((24, 8, 30, 4096), (48, 16, 20, 2048)):
        rng = np.random.default_rng(0)
        A = np.eye(nx) + 0.03 * rng.standard_normal((nx, nx)); B = 0.1 * rng.standard_normal((nx, nu))
        prob = P.Problem("wide", A, B, np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu)), N, 2.0, rng.standard_normal(nx))
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, max_iter=100, abs_pri_tol=0.0, abs_dua_tol=0.0)
        s.set_bound_constraints(np.full(nx, -2.0), np.full(nx, 2.0), np.full(nu, -0.3), np.full(nu, 0.3))
        if os.environ.get("WIDE_BOX") != "1":
            s.set_cone_constraints(Acx=[0], qcx=[3], cx=[0.7], Acu=[], qcu=[], cu=[])
            s.set_linear_constraints(Alin_x=np.random.default_rng(1).standard_normal((2, nx)), blin_x=np.array([1.0, 1.5]), Alin_u=np.zeros((0, nu)), blin_u=np.zeros(0))
        s.set_x0_batch(np.asfortranarray(prob.x0[:, None] + 0.1 * np.random.default_rng(2).standard_normal((nx, batch))))
        ms = []
        for k in range(8):
            s.reset_workspace(); ms.append(s.solve_timed())
        import hashlib
        out.append((float(np.median(ms[2:])), s.launch_info()["layout"], s.jit_info()[:60], hashlib.sha256(np.round(s.get_first_controls_batch(), 9).tobytes()).hexdigest()[:8]))
        s.reset()
    print(json.dumps(out))
    sys.exit(0)
variants = [("box path (no families)", {"WIDE_BOX": "1"}), ("k_admm_solve_fam (round 4)", {"TINYMPC_LAYOUT": "A"}), ("streamed, default", {}),
            ("batch 1, ds_bpermute reductions", {"TINYMPC_JIT_DEFS": "-DTINY_WIDE_FAM_BATCH=1 -DTINY_WIDE_FAM_SWAP=0"}),
            ("batch 2 + swap", {"TINYMPC_JIT_DEFS": "-DTINY_WIDE_FAM_BATCH=2 -DTINY_WIDE_FAM_SWAP=1"}),
            ("batch 3 + swap", {"TINYMPC_JIT_DEFS": "-DTINY_WIDE_FAM_BATCH=3 -DTINY_WIDE_FAM_SWAP=1"}),
            ("batch 4 + swap", {"TINYMPC_JIT_DEFS": "-DTINY_WIDE_FAM_BATCH=4 -DTINY_WIDE_FAM_SWAP=1"}),
            ("batch 4 + swap, ring 8", {"TINYMPC_JIT_DEFS": "-DTINY_WIDE_FAM_BATCH=4 -DTINY_WIDE_FAM_SWAP=1 -DTINY_WIDE_FAM_AHEAD=8"}),
            ("batch 2 + swap, no fences", {"TINYMPC_JIT_DEFS": "-DTINY_WIDE_FAM_BATCH=2 -DTINY_WIDE_FAM_SWAP=1 -DTINY_WIDE_FAM_FENCE=0"}),
            ("batch 6 + swap, ring 8", {"TINYMPC_JIT_DEFS": "-DTINY_WIDE_FAM_BATCH=6 -DTINY_WIDE_FAM_SWAP=1 -DTINY_WIDE_FAM_AHEAD=8"}),
            ("streamed, default (again)", {})]
if len(sys.argv) > 1:  # custom variants: each argument one TINYMPC_JIT_DEFS string ("" = default)
    variants = [("streamed, default", {})] + [(a, {"TINYMPC_JIT_DEFS": a}) for a in sys.argv[1:]] + [("streamed, default (again)", {})]
for name, env in variants:
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), capture_output=True, text=True)
    try:
        d = json.loads(o.stdout.strip().splitlines()[-1])
        print(f"{name:48s} " + "   ".join(f"{m:7.3f} ms layout {l} u0 {h} [{j[9:]}]" for m, l, j, h in d), flush=True)
    except Exception:
        print(name, "FAILED", o.stderr[-300:], flush=True)
