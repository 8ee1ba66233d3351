"""50 setup / reset pairs per BASELINE system (for rocprofv3 --kernel-trace --stats: the duration of k_precompute_rows and of the
setup's other stream work)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
P = pkg.problems
for prob in (P.cartpole(20, True), P.quadrotor(50), P.rocket(100)):
    for _ in range(50):
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, fdyn=prob.fdyn)
        s.reset()
print("done")
