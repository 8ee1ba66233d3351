"""Why does the launched single-instance tick read ~190 us inside bench.py and ~20 us on its own? Times 200 ticks of the quadrotor
(tol 1e-3, warm start) after each of the things the bench does before that leg."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
prob = P.quadrotor(50)

def ticks(label, stats=False):
    tk = pkg.TinyMPC()
    tk.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
    tk.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    x = prob.x0.copy(); acc = 0.0
    for k in range(220):
        t0 = time.perf_counter(); u0 = tk.mpc_step(x)[:, 0]; dt = time.perf_counter() - t0
        if k >= 20: acc += dt
        if stats: tk.get_stats()
        x = prob.A @ x + prob.B @ u0
    print(f"{label:55s} {1e6 * acc / 200:8.1f} us per tick  layout {tk.launch_info()['layout']}", flush=True)
    tk.reset()

ticks("fresh process")
ticks("fresh process, get_stats() between the ticks", True)
import torch
ticks("after import torch")
ticks("after import torch, get_stats() between the ticks", True)
big = pkg.TinyMPC()
big.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=8192, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=50)
big.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
big.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(8192))); big.solve()
ticks("with an 8,192-instance handle alive")
rk = P.rocket(100)
r = pkg.TinyMPC()
r.setup(rk.A, rk.B, rk.Q, rk.R, rk.N, batch=4096, rho=rk.rho, fdyn=rk.fdyn, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=20)
r.set_bound_constraints(rk.x_min, rk.x_max, rk.u_min, rk.u_max); r.set_cone_constraints(**rk.cones); r.set_linear_constraints(**rk.linear)
r.set_x0_batch(np.asfortranarray(rk.x0[:, None] * np.linspace(0.6, 1.2, 4096)[None, :])); r.solve()
ticks("after a run-time specialised kernel ran (layout " + r.launch_info()["layout"] + ")")
r.reset()
ticks("after that handle was reset")
nx, nu = 96, 32
rng = np.random.default_rng(0)
m = pkg.TinyMPC()
m.setup(np.eye(nx) * 0.9, 0.1 * rng.standard_normal((nx, nu)), np.eye(nx), np.eye(nu), 10, batch=512, rho=1.0, max_iter=10)
m.set_x0_batch(np.asfortranarray(rng.standard_normal((nx, 512)))); m.solve(); m.reset()
ticks("after a layout-M handle ran and was reset")
s1 = pkg.TinyMPC()
s1.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
s1.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
s1.set_x0(prob.x0)
for k in range(5): s1.solve_timed()
ticks("after solve_timed on another single-instance handle")
big.reset()
ticks("after the big handle was reset")
