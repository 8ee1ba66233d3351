"""Build-time lint of the generated gfx950 code: the DPP data hazard of the fused mat-vec chain.

`v_fmac_f64_dpp acc, w, m row_newbcast:k` reads `w` through the DPP crossbar; a VGPR written by a VALU instruction
must not be read through DPP within the next 2 wait states, and neither the hardware nor hipcc (the instructions live in
inline asm) guards that. The chain is emitted in blocks of four so that the scheduler can fill the gaps between them
(tinympc_sweep.h); only the first block opens with `s_nop 1`. This test compiles every solve kernel to assembly and
checks each DPP FMA: walking back from it, 2 wait states must pass before any VALU instruction that writes the
registers of `w` (or a label, behind which the predecessors are unknown) is met."""
from __future__ import annotations

import os
import re
import shutil
import subprocess

import pytest
from conftest import ROOT

CSRC = os.path.join(ROOT, "tinympc-matlab_amd", "csrc")
SOURCES = ["tinympc_solve.hip", "tinympc_solve_b.hip", "tinympc_solve_c.hip", "tinympc_solve_fam.hip", "tinympc_solve_adapt.hip"]
REG = re.compile(r"^v\[(\d+):(\d+)\]$|^v(\d+)$")


def _regs(tok: str):
    m = REG.match(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def _lint(asm_text: str):
    """Returns (number of DPP FMAs checked, list of violations)."""
    window = []  # recent items, newest last: ("instr", mnemonic, dest_regs, wait_states) or ("label",)
    checked, bad = 0, []
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;") else ""
        if not line or line.startswith(".") and not line.endswith(":"):
            continue
        if line.endswith(":"):
            window.append(("label",))
            continue
        parts = line.split(None, 1)
        mnem, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        toks = [t.strip() for t in ops.split(",")]
        if mnem == "v_fmac_f64_dpp":
            checked += 1
            src = _regs(toks[1].split()[0])
            waited = 0
            for item in reversed(window):
                if waited >= 2:
                    break
                if item[0] == "label":
                    bad.append((ln, raw.strip(), "branch target inside the hazard window"))
                    break
                _, m2, dest, ws = item
                if m2.startswith("v_") and dest & src:
                    bad.append((ln, raw.strip(), f"{m2} writes {sorted(dest & src)} {waited} wait state(s) earlier"))
                    break
                waited += ws
        dest = _regs(toks[0].split()[0]) if mnem.startswith("v_") and toks and toks[0] else set()
        ws = 1
        if mnem == "s_nop":
            ws = int(toks[0]) + 1
        window.append(("instr", mnem, dest, ws))
        if len(window) > 8:
            window.pop(0)
    return checked, bad


def test_lint_catches_a_planted_hazard():
    good = "s_nop 1\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
    assert _lint("v_mov_b32_e32 v4, v9\n" + good) == (1, [])
    n, bad = _lint("v_mov_b32_e32 v4, v9\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:0 row_mask:0xf bank_mask:0xf\n")
    assert n == 1 and len(bad) == 1
    n, bad = _lint("v_mov_b32_e32 v5, v9\n s_nop 0\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    assert len(bad) == 1  # one wait state is not enough
    n, bad = _lint(".LBB0_1:\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    assert len(bad) == 1
    assert _lint("v_mov_b32_e32 v4, v9\n v_add_u32_e32 v1, v1, v1\n v_add_u32_e32 v1, v1, v1\n" + good.split("\n")[1] + "\n")[1] == []


@pytest.mark.parametrize("source", SOURCES)
def test_generated_code_has_no_dpp_hazard(source, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / (source + ".s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, source)], check=True, timeout=900)
    checked, bad = _lint(out.read_text())
    assert checked > 100, "no fused DPP mat-vec found: did the kernel change?"
    assert not bad, f"{len(bad)} DPP hazard(s) in {source}, first: {bad[:3]}"
