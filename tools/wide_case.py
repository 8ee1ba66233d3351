"""One random cone / linear-inequality configuration on a WIDE system (17 <= nx+nu <= 64): what tools/fuzz_wide_families.py solves per case and
tools/repro_wide_case.py replays by (seed, case number) -- ONE generator, so a replayed case is the fuzzer's case. Horizons 4..40, batches 16..300,
up to three state cones of which two may share rows (rounds), an input cone, up to 6 linear rows per side, fdyn, constant or per-knot bounds."""
import numpy as np


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
