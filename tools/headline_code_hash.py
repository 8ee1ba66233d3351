"""The code of the headline kernel (k_admm_solve_d<12, 4, 50, true, 4, 25, false>, the plain 16-lane quadrotor N=50 kernel) as the
build produced it: sha256 over its instructions (labels renumbered, comments and directives dropped), plus the compiler's version.
The kernel's speed depends on details of the generated code that no source-level reasoning predicts (+-2.5 % between builds with the
same instruction mix: profiles/r03_dgroup_ab.txt), so a change of this hash means: run tools/headline_ab.py against the previous
build on ONE box before believing any number, then record the new hash.
    python tools/headline_code_hash.py            print the hash of the current build
    python tools/headline_code_hash.py --record   write tests/golden/headline_kernel_code.json (after the A/B)"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN7tinympc14k_admm_solve_dILi12ELi4ELi50ELb1ELi4ELi25ELb0EEEvNS_11SolveParamsE"
RECORD = os.path.join(ROOT, "tests", "golden", "headline_kernel_code.json")


def compiler_version() -> str:
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return "unknown"
    m = re.search(r"clang version [^\n]+", out)
    return m.group(0).strip() if m else out.strip().splitlines()[0]


def current_hash() -> dict | None:
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    path = ge.device_asm_path("tinympc_solve_d.hip")
    if not os.path.exists(path):
        return None
    text = open(path).read()
    m = re.search(r"^%s:(.*?)^\.Lfunc_end" % re.escape(KERNEL), text, re.S | re.M)
    if not m:
        return None
    lines = []
    for line in m.group(1).split("\n"):
        t = line.split(";")[0].rstrip()
        if t.startswith(".LBB"):
            lines.append("L:")
        elif t.startswith("\t") and not t.strip().startswith("."):
            lines.append(re.sub(r"\.LBB\d+_\d+", ".LBB", t.strip()))
    return {"kernel": KERNEL, "instructions": sum(1 for x in lines if x != "L:"), "sha256": hashlib.sha256("\n".join(lines).encode()).hexdigest(),
            "compiler": compiler_version()}


if __name__ == "__main__":
    h = current_hash()
    if h is None:
        sys.exit("no build assembly: run __graft_entry__.build() first")
    print(json.dumps(h, indent=1))
    if "--record" in sys.argv:
        h["measured"] = ("kernel 1.68-1.69 ms (8,192 x 200 iterations, bench.py on MI355X; profiles/r05_kernel_stats.csv); tools/headline_ab.py, one box, against the "
                         "build before the sweeps' LDS reads and waits became compiler-tracked (4,349 instructions, asm-issued reads): 1.6872 vs 1.6820 ms "
                         "(+0.3 %, three interleaved rounds each), profiles/r05_tracked_reads_headline_ab.txt")
        with open(RECORD, "w") as f:
            json.dump(h, f, indent=1)
            f.write("\n")
        print("recorded", RECORD)
