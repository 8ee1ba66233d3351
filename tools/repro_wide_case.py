"""One case of tools/fuzz_wide_families.py again, by (seed, case number), with parts of it switched off -- to find what a failing
case's error depends on. The generator below draws EXACTLY what the fuzzer draws, in its order, so the case is the fuzzer's.
    python tools/repro_wide_case.py <seed> <case> [variant ...]
variants: base layoutA nocones nolinear nocu nolu nolx const N=<n> batch=<n> iters=<n> noswap"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
import pyoracle as O
pkg = g.load_package(); P = pkg.problems


def draw(rng):
    nxu = int(rng.integers(17, 65))
    nu = int(rng.integers(2, max(3, nxu // 4)))
    nx = nxu - nu
    N = int(rng.integers(4, 41))
    batch = int(rng.choice([16, 33, 70, 300]))
    A = 0.9 * np.eye(nx) + (0.15 / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    Bm = 0.3 * rng.standard_normal((nx, nu))
    Q, R = np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu))
    rho, xref = float(rng.uniform(0.5, 3.0)), rng.standard_normal(nx)
    fdyn = 0.01 * rng.standard_normal(nx) if rng.integers(0, 2) else None
    ncx = int(rng.integers(0, 4))
    Acx, qcx, cx = [], [], []
    for c in range(ncx):
        q = int(rng.integers(2, 6)); a = int(rng.integers(0, nx - q + 1))
        if c == 2 and Acx: a = min(Acx[0] + 1, nx - q)
        Acx.append(a); qcx.append(q); cx.append(float(rng.uniform(0.4, 1.5)))
    has_cu = bool(rng.integers(0, 2)) and nu >= 2
    qcu = [int(rng.integers(2, min(nu, 4) + 1))] if has_cu else []
    cones = dict(Acx=Acx, qcx=qcx, cx=cx, Acu=[0] if has_cu else [], qcu=qcu, cu=[0.7] if has_cu else [])
    nlx, nlu = int(rng.integers(0, 7)), int(rng.integers(0, 4))
    if not (ncx or has_cu or nlx or nlu): nlx = 1
    linear = dict(Alin_x=rng.standard_normal((nlx, nx)), blin_x=rng.uniform(0.5, 1.5, nlx), Alin_u=rng.standard_normal((nlu, nu)), blin_u=rng.uniform(0.3, 0.8, nlu))
    settings = dict(max_iter=int(rng.integers(20, 80)), abs_pri_tol=1e-3, abs_dua_tol=1e-3, check_termination=int(rng.choice([1, 1, 3])))
    varying = bool(rng.integers(0, 3) == 0)
    scale = rng.uniform(0.8, 1.0, (1, N)) if varying else None
    x0s = rng.standard_normal((nx, batch)) * np.linspace(0.1, 1.0, batch)[None, :]
    return dict(nx=nx, nu=nu, N=N, batch=batch, A=A, B=Bm, Q=Q, R=R, rho=rho, xref=xref, fdyn=fdyn, cones=cones, linear=linear, settings=settings, scale=scale, x0s=x0s)


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
