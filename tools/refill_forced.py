"""Slot refill with FORCED iteration counts (nothing to balance): kernel time of the plain kernel (TINYMPC_REFILL=0) and of the
variant (=1) over batch sizes -- where the resident set pays by itself. Usage (GPU box): python tools/refill_forced.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); P = pkg.problems
prob = P.quadrotor(50)
for B in (16384, 32768, 65536):
  x0s = P.quadrotor_batch_x0(B)
  for mode in ("0", "1"):
      os.environ["TINYMPC_REFILL"] = mode
      s = pkg.TinyMPC()
      s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=200)
      s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
      s.set_x0_batch(x0s)
      ms = []
      for _ in range(6):
          s.reset_workspace(); ms.append(s.solve_timed())
      print(B, mode, s.jit_info(), "forced 200 iterations:", float(np.median(ms[2:])), "ms")
      s.reset()
