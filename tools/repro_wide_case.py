"""One case of tools/fuzz_wide_families.py again, by (seed, case number), with parts of it switched off -- to find what a failing
case's error depends on (the generator is the fuzzer's: tools/wide_case.py).
    python tools/repro_wide_case.py <seed> <case> [variant ...]
variants: base layoutA nocones nolinear nocu nolu nolx const N=<n> batch=<n> iters=<n> noswap"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import pyoracle as O
from wide_case import draw  # (the fuzzer's own generator)
pkg = g.load_package(); P = pkg.problems


def run(c, variant):
    c = dict(c); cones = dict(c["cones"]); linear = dict(c["linear"]); settings = dict(c["settings"])
    nx, nu, N, batch = c["nx"], c["nu"], c["N"], c["batch"]
    scale, x0s = c["scale"], c["x0s"]
    os.environ.pop("TINYMPC_LAYOUT", None); os.environ.pop("TINYMPC_JIT_DEFS", None)
    for v in variant.split("+"):
        if v == "layoutA": os.environ["TINYMPC_LAYOUT"] = "A"
        elif v == "noswap": os.environ["TINYMPC_JIT_DEFS"] = "-DTINY_WIDE_FAM_SWAP=0"
        elif v == "nocones": cones = dict(Acx=[], qcx=[], cx=[], Acu=[], qcu=[], cu=[])
        elif v == "nocu": cones.update(Acu=[], qcu=[], cu=[])
        elif v == "nocx": cones.update(Acx=[], qcx=[], cx=[])
        elif v == "nolinear": linear = dict(Alin_x=np.zeros((0, nx)), blin_x=np.zeros(0), Alin_u=np.zeros((0, nu)), blin_u=np.zeros(0))
        elif v == "nolu": linear.update(Alin_u=np.zeros((0, nu)), blin_u=np.zeros(0))
        elif v == "nolx": linear.update(Alin_x=np.zeros((0, nx)), blin_x=np.zeros(0))
        elif v == "const": scale = None
        elif v.startswith("N="):
            N = int(v[2:]); scale = None if scale is None else np.resize(scale, (1, N))
        elif v.startswith("batch="):
            batch = int(v[6:]); x0s = np.resize(x0s, (nx, batch))
        elif v.startswith("iters="): settings["max_iter"] = int(v[6:])
    prob = P.Problem("widefuzz", c["A"], c["B"], c["Q"], c["R"], N, c["rho"], c["xref"])
    prob.x_min, prob.x_max = np.full(nx, -3.0), np.full(nx, 3.0)
    prob.u_min, prob.u_max = np.full(nu, -1.0), np.full(nu, 1.0)
    prob.fdyn = c["fdyn"]
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, N, batch=batch, rho=prob.rho, fdyn=prob.fdyn, **settings)
    xmin, xmax = prob.x_min, prob.x_max
    if scale is not None:
        xmin = np.repeat(prob.x_min[:, None], N, 1) * scale; xmax = -xmin
    s.set_bound_constraints(xmin, xmax, prob.u_min, prob.u_max)
    s.set_cone_constraints(**cones); s.set_linear_constraints(**linear)
    checked = sorted({0, batch // 2, batch - 1})
    orcs = {}
    for b in checked:
        o = O.OraclePort(prob).load_problem(prob, settings)
        o.set_bound_constraints(*(np.broadcast_to(np.asarray(v).reshape(len(v), -1), (len(v), n)).copy() for v, n in ((xmin, N), (xmax, N), (prob.u_min, N - 1), (prob.u_max, N - 1))))
        o.set_cone_constraints(**cones); o.set_linear_constraints(**linear)
        orcs[b] = o
    out = []
    for rnd in range(2):
        xs = x0s * (1.0 - 0.3 * rnd)
        s.set_x0_batch(np.asfortranarray(xs)); s.solve()
        sol, st = s.get_solution_batch(), s.get_stats_batch()
        for b in checked:
            orcs[b].set_x0(xs[:, b]); orcs[b].solve()
            ox, ou = orcs[b].solution()
            dx, du = np.abs(sol["states"][:, :, b] - ox), np.abs(sol["controls"][:, :, b] - ou)
            ex, eu = dx.max() / max(np.abs(ox).max(), 1e-300), du.max() / max(np.abs(ou).max(), 1e-300)
            wx, wu = np.unravel_index(dx.argmax(), dx.shape), np.unravel_index(du.argmax(), du.shape)
            os_ = orcs[b].stats()
            out.append(f"    solve {rnd} instance {b:3d}: iters {st['iter'][b]}/{os_['iter']} status {st['status'][b]}/{os_['status']} states {ex:.1e} at (row {wx[0]}, knot {wx[1]}) controls {eu:.1e} at (row {wu[0]}, knot {wu[1]}) |x|max {np.abs(ox).max():.2e} |u|max {np.abs(ou).max():.2e}")
    print(f"  {variant:24s} layout {s.launch_info()['layout']} {s.jit_info()[:60]} cones {cones['Acx']}/{cones['qcx']} u {cones['Acu']}/{cones['qcu']} rows {len(linear['blin_x'])}+{len(linear['blin_u'])}", flush=True)
    print("\n".join(out), flush=True)
    s.reset()


if __name__ == "__main__":
    seed, case = int(sys.argv[1]), int(sys.argv[2])
    variants = sys.argv[3:] or ["base"]
    rng = np.random.default_rng(seed)
    for _ in range(case): draw(rng)
    c = draw(rng)
    print(f"seed {seed} case {case}: nx={c['nx']} nu={c['nu']} N={c['N']} batch={c['batch']} fdyn={c['fdyn'] is not None} per-knot={c['scale'] is not None} settings {c['settings']}")
    for v in variants: run(c, v)
