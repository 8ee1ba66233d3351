import os, sys
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import pyoracle as O
from wide_case import draw
pkg = g.load_package(); P = pkg.problems
rng = np.random.default_rng(92)
for _ in range(8): draw(rng)
c = draw(rng)
prob = P.Problem("w", c["A"], c["B"], c["Q"], c["R"], c["N"], c["rho"], c["xref"])
s = pkg.TinyMPC(); s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, **c["settings"])
o = O.OraclePort(prob).load_problem(prob, c["settings"])
gc = s.get_cache()
for k in ("Kinf", "Pinf", "Quu_inv", "AmBKt"):
    a, b = np.asarray(gc[k]), np.asarray(o.get(k))
    print(k, "rel err", np.max(np.abs(a - b)) / np.max(np.abs(b)), "max", np.max(np.abs(b)))
print("riccati iterations (oracle)", o.stats().get("riccati_iters"), "cond(A)", np.linalg.cond(prob.A), "spectral radius", np.abs(np.linalg.eigvals(prob.A)).max())
