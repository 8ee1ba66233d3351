"""Setup time (the LQR cache precompute k_precompute: Riccati fixed point on the device) by system size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
for nx, nu in ((12, 4), (24, 8), (48, 16), (96, 32), (160, 32), (224, 32), (300, 20), (480, 32)):
    rng = np.random.default_rng(7)
    A = (0.95 if nx < 256 else 0.6) * np.eye(nx) + ((0.15 if nx < 256 else 0.1) / np.sqrt(nx)) * rng.standard_normal((nx, nx))
    B = 0.08 * rng.standard_normal((nx, nu))
    Q, R = np.diag(rng.uniform(1, 10, nx)), np.diag(rng.uniform(0.5, 2, nu))
    ts = []
    for k in range(2):
        s = pkg.TinyMPC()
        t0 = time.perf_counter()
        s.setup(A, B, Q, R, 10, batch=16, rho=2.0, max_iter=5)
        c = s.get_cache()
        ts.append(time.perf_counter() - t0)
        s.reset()
    print(f"nx={nx:3d} nu={nu:2d}: setup + get_cache {1e3 * min(ts):9.1f} ms", flush=True)
