"""Where does an occasional 40 ms tick come from? 20,000 launched ticks of the quadrotor (tol 1e-3, warm start); every tick above
1 ms is printed with the library's own split: time in the launch call, time waiting, number of polls, whether the polling
budget ran out (tinympc_debug_tick_timing)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
L = pkg.load_library()
L.tinympc_debug_tick_timing.restype = C.c_int
L.tinympc_debug_tick_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
prob = P.quadrotor(50)
if "--busy-first" in sys.argv:  # what bench.py does before its tick leg
    big = pkg.TinyMPC()
    big.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=65536, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=50)
    big.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    big.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(65536))); big.solve(); big.reset()
tk = pkg.TinyMPC()
tk.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=1, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100)
tk.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
x = prob.x0.copy(); d = []; out4 = (C.c_double * 4)()
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
for k in range(n):
    t0 = time.perf_counter(); u0 = tk.mpc_step(x)[:, 0]; dt = (time.perf_counter() - t0) * 1e6
    d.append(dt)
    if dt > 1000.0 and k > 5:
        L.tinympc_debug_tick_timing(tk._h, out4)
        print(f"tick {k}: {dt:9.1f} us | launch call {out4[0]:9.1f} us, wait {out4[1]:9.1f} us, polls {int(out4[2])}, budget ran out {int(out4[3])}", flush=True)
    x = prob.A @ x + prob.B @ u0
    if k % 500 == 499: x = prob.x0.copy()
d = np.array(d[20:])
print(f"{len(d)} ticks: mean {d.mean():.1f} median {np.median(d):.1f} p99 {np.percentile(d, 99):.1f} p99.9 {np.percentile(d, 99.9):.1f} max {d.max():.1f} us; above 1 ms: {int((d > 1000).sum())}")
