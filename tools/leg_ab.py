"""A/B of bench legs over variant builds of the library (tools/bin/libtinympc_hip_<name>.so; "base" = the in-tree library): one child
process (tools/leg_workload.py) per leg, variant and round, rounds interleaved.
    python tools/leg_ab.py <variant> [--legs a,b,c] [--rounds 2] [--launches 12]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("variant")
ap.add_argument("--legs", default="rocket_batch,rocket_batch_n10,rocket_instance,wide_system,wide_families,long_horizon,large_system,adaptive_rho_batch,single_instance")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--launches", type=int, default=12)
a = ap.parse_args()
res = {}
for leg in a.legs.split(","):
    for r in range(a.rounds):
        for n in ("base", a.variant):
            env = dict(os.environ)
            if n != "base": env["TINYMPC_HIP_LIBRARY"] = os.path.join(ROOT, "tools", "bin", f"libtinympc_hip_{n}.so")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "leg_workload.py"), leg, str(a.launches)], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(leg, n, "FAILED", out.stderr[-300:], flush=True); continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res.setdefault((leg, n), []).append(d["kernel_ms_median"]); lay = d.get("layout")
    b, v = res.get((leg, "base"), []), res.get((leg, a.variant), [])
    if b and v: print(f"{leg:22s} layout {lay}: base {min(b):9.4f} ms   {a.variant} {min(v):9.4f} ms   base/{a.variant} = {min(b) / min(v):.4f}   (best of {a.rounds} medians)", flush=True)
