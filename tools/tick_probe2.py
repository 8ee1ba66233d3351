import os, sys, time
import numpy as np
if '--torch-first' in sys.argv:
    import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
prob = P.quadrotor(50)
def ticks(label, n=220):
    tk = pkg.TinyMPC()
    tk.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
    tk.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    x = prob.x0.copy(); d = []
    for k in range(n):
        t0 = time.perf_counter(); u0 = tk.mpc_step(x)[:, 0]; dt = time.perf_counter() - t0
        if k >= 20: d.append(dt * 1e6)
        x = prob.A @ x + prob.B @ u0
    d = np.array(d)
    print(f"{label:50s} mean {d.mean():8.1f} median {np.median(d):7.1f} p90 {np.percentile(d,90):7.1f} max {d.max():9.1f} us; count>100us {int((d>100).sum())}", flush=True)
    tk.reset()
ticks("fresh")
import torch
ticks("after import torch")
ticks("again")
if '--torch-first' not in sys.argv:
    sys.exit(0)
torch.cuda.set_device(0); torch.cuda.synchronize()
ticks("after torch.cuda.synchronize()")
ticks("again")
t = torch.zeros(10, device="cuda"); torch.cuda.synchronize()
ticks("after a torch tensor on the device")
ticks("again")
