"""ISA lint of the generated gfx950 code: the DPP data hazard of the fused mat-vec chain.

`v_fmac_f64_dpp acc, w, m row_newbcast:k` (and `v_mov_b64_dpp acc, w row_newbcast:k`) reads `w` through the DPP crossbar; a VGPR written by a VALU instruction
must not be read through DPP within the next 2 wait states, and neither the hardware nor hipcc (the instructions live in
inline asm) guards that. `lint(asm_text)` walks back from every DPP FMA: 2 wait states must pass before any VALU
instruction that writes the registers of `w` is met -- along EVERY path: a label inside the window is followed to the branches
that jump to it (round 4; before, any label inside the window counted as a violation).

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
    # pass 1: the instruction stream as items -- ("label", name) | ("instr", mnemonic, dest_regs, wait_states, branch_target, line, text)
    items = []
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;") else ""
        if not line or line.startswith(".") and not line.endswith(":"):
            continue
        if line.endswith(":"):
            items.append(("label", line[:-1]))
            continue
        parts = line.split(None, 1)
        mnem, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        toks = [t.strip() for t in ops.split(",")]
        dest = _regs(toks[0].split()[0]) if mnem.startswith("v_") and toks and toks[0] else set()
        if mnem.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap_b")) and len(toks) > 1:
            dest |= _regs(toks[1].split()[0])  # the swaps write both of their operands
        ws = int(toks[0]) + 1 if mnem == "s_nop" else 1
        target = toks[0] if mnem.startswith(("s_cbranch", "s_branch")) and toks and toks[0] else None
        items.append(("instr", mnem, dest, ws, target, ln, raw.strip()))
    sites = {}  # label -> indices of the branches that jump to it
    for i, it in enumerate(items):
        if it[0] == "instr" and it[4]:
            sites.setdefault(it[4], []).append(i)

    def walk(i, waited, src, depth):
        """Walks back from item i (inclusive) along every path that can reach the DPP instruction: None if `src` is not written by a
        VALU instruction within the 2 wait states, else the reason. A branch target is followed to its branch sites (a branch is one
        wait state itself) and, unless the instruction above it cannot fall through, straight up."""
        while i >= 0 and waited < 2:
            it = items[i]
            if it[0] == "label":
                if depth >= 3:
                    return "branch target inside the hazard window (paths nested too deep to follow)"
                for b in sites.get(it[1], []):
                    why = walk(b, waited, src, depth + 1)
                    if why:
                        return why + " (on the path through the branch at line %d)" % items[b][5]
                if not sites.get(it[1]) and i == 0:
                    return "branch target inside the hazard window (nothing known about what runs before it)"
                i -= 1
                if i >= 0 and items[i][0] == "instr" and items[i][1] in ("s_branch", "s_endpgm", "s_setpc_b64"):
                    return None  # nothing falls through into the label
                continue
            _, m2, dest, ws, _, _, _ = it
            if m2.startswith("v_") and dest & src:
                return f"{m2} writes {sorted(dest & src)} {waited} wait state(s) earlier"
            if m2 in ("s_setpc_b64", "s_swappc_b64"):
                return "indirect branch inside the hazard window"
            waited += ws
            i -= 1
        return None

    checked, bad = 0, []
    for i, it in enumerate(items):
        if it[0] == "instr" and it[1] in ("v_fmac_f64_dpp", "v_mov_b64_dpp"):  # (the move: layout E's gathers under an EXEC mask)
            checked += 1
            ops = it[6].split(None, 1)[1]
            src = _regs(ops.split(",")[1].split()[0])
            why = walk(i - 1, 0, src, 0)
            if why:
                bad.append((it[5], it[6], why))
    return checked, bad


def lint_file(path: str):
    with open(path) as f:
        return lint(f.read())
