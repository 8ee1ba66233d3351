"""Kernel time per ADMM iteration of BASELINE config 4 (rocket landing, SOC + linear + fdyn) for the kernels that carry
the families: k_admm_solve_fam (TINYMPC_LAYOUT=A), k_admm_solve_c<FAM> (TINYMPC_LAYOUT=C) and, for short horizons, the
run-time specialised layout D with the families in registers (TINYMPC_LAYOUT=D), for long ones layout E (the horizon cut
across the wavefronts of a workgroup, TINYMPC_LAYOUT=E)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
for N in (10, 20, 28, 100):
    prob = P.rocket(N)
    for batch in (1, 256, 512, 1024, 2048, 4096, 16384):
        row = [f"rocket N={N:3d} batch={batch:5d}"]
        for layout in ("A", "C", "D", "E"):
            if layout == "D" and (N > 30 or batch < 256):
                continue
            if layout == "E" and N < 40:
                continue
            if layout == "A" and N >= 100 and "--all" not in sys.argv:
                continue
            os.environ["TINYMPC_LAYOUT"] = layout
            s = pkg.TinyMPC()
            s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, max_iter=200, abs_pri_tol=0.0, abs_dua_tol=0.0)
            s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            if prob.x_ref is not None: s.set_x_ref(prob.x_ref)
            if prob.u_ref is not None: s.set_u_ref(prob.u_ref)
            s.set_cone_constraints(**prob.cones)
            if prob.linear: s.set_linear_constraints(**prob.linear)
            s.set_x0_batch(np.repeat(prob.x0[:, None], batch, axis=1))
            ms = []
            for _ in range(6):
                s.reset_workspace(); ms.append(s.solve_timed())
            t = float(np.median(ms[2:]))
            if layout in "DE" and s.launch_info()["layout"] != layout:
                s.reset()
                continue
            row.append(f"{layout}: {5*t:8.2f} us/it {batch*200/t/1e3:7.1f} M/s")
            s.reset()
        print(" | ".join(row), flush=True)
