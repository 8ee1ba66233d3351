"""ISA lint of the generated gfx950 code: the DPP data hazard of the fused mat-vec chain.

`v_fmac_f64_dpp acc, w, m row_newbcast:k` (and `v_mov_b64_dpp acc, w row_newbcast:k`) reads `w` through the DPP crossbar; a VGPR written by a VALU instruction
must not be read through DPP within the next 2 wait states, and neither the hardware nor hipcc (the instructions live in
inline asm) guards that. `lint(asm_text)` walks back from every DPP FMA: 2 wait states must pass before any VALU
instruction that writes the registers of `w` (or a label, behind which the predecessors are unknown) is met.

Used by __graft_entry__.build() on the assembly of the very objects that are linked into libtinympc_hip.so (the build
fails on a violation) and by tests/test_isa_hazards.py."""
from __future__ import annotations

import re

REG = re.compile(r"^v\[(\d+):(\d+)\]$|^v(\d+)$")


def _regs(tok: str):
    m = REG.match(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def lint(asm_text: str):
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
        if mnem in ("v_fmac_f64_dpp", "v_mov_b64_dpp"):  # (the move: layout E's gathers under an EXEC mask)
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
        if mnem.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap_b")) and len(toks) > 1:
            dest |= _regs(toks[1].split()[0])  # the swaps write both of their operands
        ws = 1
        if mnem == "s_nop":
            ws = int(toks[0]) + 1
        window.append(("instr", mnem, dest, ws))
        if len(window) > 8:
            window.pop(0)
    return checked, bad




def lint_file(path: str):
    with open(path) as f:
        return lint(f.read())
