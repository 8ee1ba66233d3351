"""Setup time, phase by phase (tinympc_debug_setup_timing): the C call tinympc_setup_batch alone, the Python mirror's setup()
(+ update_settings), and the precompute kernel by HIP events -- beside the reference's own tiny_setup on one host core.
    python tools/setup_time.py [> profiles/r05_setup_time.txt]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g

pkg = g.load_package()
P = pkg.problems
L = pkg._lib
lib = pkg.load_library()


def c_setup(prob, batch, reps=30):
    """-> list of (wall us of the C call, phases) over `reps` setup / reset pairs"""
    A, B, Q, R = (np.asfortranarray(m, dtype=np.float64) for m in (prob.A, prob.B, prob.Q, prob.R))
    f = np.asfortranarray(prob.fdyn, dtype=np.float64) if getattr(prob, "fdyn", None) is not None else None
    p = lambda a: a.ctypes.data_as(L.c_double_p) if a is not None else None
    out = []
    for _ in range(reps):
        h = L.Handle()
        t0 = time.perf_counter()
        rc = lib.tinympc_setup_batch(C.byref(h), p(A), p(B), p(f), p(Q), p(R), prob.rho, prob.nx, prob.nu, prob.N, batch, -1, 0)
        t1 = time.perf_counter()
        assert rc == 0, L.last_error()
        ph = np.zeros(10)
        lib.tinympc_debug_setup_timing(h, ph.ctypes.data_as(L.c_double_p))
        t2 = time.perf_counter()
        lib.tinympc_reset(C.byref(h), 0)
        t3 = time.perf_counter()
        out.append((1e6 * (t1 - t0), ph.copy(), 1e6 * (t3 - t2)))
    return out


def main():
    names = ("prologue", "dev_arena", "pin_arena", "stage+queue", "queue_pre", "wait", "total")
    print("tinympc_setup_batch, host microseconds per phase (median of 30 after 3 discarded; first call listed separately)")
    print("%-22s %8s | %s | %8s" % ("problem", "call", " ".join("%11s" % n for n in names), "reset"))
    for name, prob, batch in (("cartpole N=20", P.cartpole(20, True), 1), ("quadrotor N=50", P.quadrotor(50), 1), ("rocket N=100", P.rocket(100), 1),
                              ("quadrotor N=50 x8192", P.quadrotor(50), 8192)):
        r = c_setup(prob, batch, 33)
        first = r[0]
        r = r[3:]
        med = lambda xs: float(np.median(xs))
        print("%-22s %8.0f | %s | %8.0f" % (name, med([x[0] for x in r]), " ".join("%11.0f" % med([x[1][k] for x in r]) for k in range(7)), med([x[2] for x in r])))
        print("%-22s %8.0f | %s | %8.0f" % ("  (first call)", first[0], " ".join("%11.0f" % first[1][k] for k in range(7)), first[2]))
        clk, us, steps = med([x[1][7] for x in r]), med([x[1][8] for x in r]), med([x[1][9] for x in r])
        if clk > 0:
            print("%-22s Riccati loop in the kernel: %d steps, %.1f us, %.0f shader clocks per step at %.2f GHz" % ("", steps, us, clk / steps, 1e-3 * clk / us))
    # through the Python mirror of the .m class (setup + update_settings), as tools/precompute_time.py measured in round 4
    for name, prob in (("cartpole N=20", P.cartpole(20, True)), ("quadrotor N=50", P.quadrotor(50)), ("rocket N=100", P.rocket(100))):
        ts = []
        for _ in range(12):
            s = pkg.TinyMPC()
            t0 = time.perf_counter()
            s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, fdyn=getattr(prob, "fdyn", None))
            ts.append(1e3 * (time.perf_counter() - t0))
            steps = s.get_cache()["riccati_iters"] if isinstance(s.get_cache(), dict) and "riccati_iters" in s.get_cache() else None
            s.reset()
        print("TinyMPC.setup (Python mirror) %-16s median %.3f ms  min %.3f ms  Riccati steps %s" % (name, float(np.median(ts[2:])), min(ts), steps))
    try:
        import pyoracle
        for name, prob in (("cartpole N=20", P.cartpole(20, True)), ("quadrotor N=50", P.quadrotor(50))):
            us = pyoracle.OracleRef.bench_setup(prob, 30)[3:]
            print("reference tiny_setup (oracle/_ref, one host core) %-16s median %.3f ms  mean %.3f ms  min %.3f ms" % (name, 1e-3 * float(np.median(us)), 1e-3 * float(np.mean(us)), 1e-3 * float(us.min())))
    except Exception as e:  # (the checker is optional here)
        print("reference tiny_setup: not measured (%s)" % e)


if __name__ == "__main__":
    main()
