"""One large-system workload (nx=96, nu=32, N=20 x 4,096 instances x 50 forced iterations) for rocprofv3 counter passes on
k_admm_solve_m. Usage (GPU box, from /tmp):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <out> -o f -- python3 tools/large_pmc.py
  ... WRITE_SIZE / SQ_* likewise; then `python tools/large_pmc.py --collect <dir with one sub-directory per pass>`"""
import glob
import json
import os
import sqlite3
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NX, NU, N, BATCH, ITERS = 96, 32, 20, 4096, 50


def run():
    import numpy as np
    import __graft_entry__ as g
    pkg = g.load_package()
    rng = np.random.default_rng(NX)
    A = np.eye(NX) * 0.98 + 0.015 * rng.standard_normal((NX, NX))
    B = 0.08 * rng.standard_normal((NX, NU))
    s = pkg.TinyMPC()
    s.setup(A, B, np.diag(rng.uniform(1, 10, NX)), np.diag(rng.uniform(0.5, 2, NU)), N, batch=BATCH, rho=2.0, max_iter=ITERS, abs_pri_tol=0.0, abs_dua_tol=0.0)
    s.set_bound_constraints(np.full(NX, -2.0), np.full(NX, 2.0), np.full(NU, -0.3), np.full(NU, 0.3))
    s.set_x0_batch(np.asfortranarray(np.random.default_rng(1).standard_normal((NX, BATCH))))
    for k in range(4):
        s.reset_workspace()
        ms = s.solve_timed()
    print("kernel ms", ms)
    s.reset()


def collect(d):
    out = {}
    for db in sorted(glob.glob(os.path.join(d, "*", "*_results.db"))):
        con = sqlite3.connect(db)
        for name, cn, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
            if "k_admm_solve_m" in name:
                out.setdefault(cn, []).append(float(v))
    c = {k: st.mean(v) for k, v in out.items()}
    res = {"kernel": "k_admm_solve_m", "workload": f"nx={NX} nu={NU} N={N}, {BATCH} instances x {ITERS} iterations per launch", "counters": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0  # read side x2: gfx950 correction (MI355X_MICROARCH.md)
        res["hbm_read_bytes_per_launch"], res["hbm_write_bytes_per_launch"] = rd, wr
        res["hbm_bytes_per_instance_iteration"] = (rd + wr) / BATCH / ITERS
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--collect":
        collect(sys.argv[2])
    else:
        run()
