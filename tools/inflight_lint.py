"""A register that an LDS read is still filling must not be touched before the s_waitcnt that retires the read.

The compiler guarantees that for the reads it knows. Through round 5 the sweeps of layouts D / E requested their operands with
`asm volatile("ds_read_b64 ...")`, which it does NOT know: it believed the destination valid at once and, under register pressure, parked
the still-empty register in an AGPR and reused it (profiles/r05_inflight_bug.txt). The sources now use reads and waits the compiler
tracks (tinympc_sweep.h: lds_read_issued_here / lds_reads_landed); this lint keeps the old form from coming back:
it follows every LDS read issued from INLINE ASM (between #ASMSTART / #ASMEND) -- linear scan over a .s file (branches are ignored:
the sweeps are straight-line code), outstanding LGKM operations in order, `s_waitcnt lgkmcnt(n)` keeps the last n.
    python tools/inflight_lint.py [--all] file.s ...  -> every instruction that reads or writes an in-flight register
    (--all: the compiler's own reads too -- a diagnostic only: without a control-flow graph it reports reads and waits that sit on
    different paths)"""
import re, sys

def regs_of(tok):
    """VGPRs and AGPRs named in a piece of operand text, as ("v" | "a", number) pairs."""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1): out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else: out.add((m.group(4), int(m.group(5))))
    return out

def lint(path, asm_only=True):
    with open(path) as f:
        return lint_text(f.read(), asm_only)


def lint_text(text, asm_only=True):
    inasm, outstanding, bad = False, [], []   # outstanding: list of (set of dest regs or None, line number, text)
    for n, line in enumerate(text.splitlines(), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): inasm = True; continue
        if t.startswith(";;#ASMEND"): inasm = False; continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        t = t.split(";")[0].strip()
        op = t.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                k = int(m.group(1))
                outstanding = outstanding[len(outstanding) - k:] if k else []
            continue
        pend = set().union(*[o[0] for o in outstanding if o[0]]) if outstanding else set()
        touched = regs_of(t) & pend
        if touched:
            src = [o for o in outstanding if o[0] and o[0] & touched]
            bad.append((n, t, sorted(touched), src[0][1], src[0][2]))
        if op.startswith(("ds_", "s_load", "s_buffer_load", "s_store", "s_buffer_store", "s_sendmsg", "s_dcache", "s_memtime", "s_memrealtime")) or (op.startswith(("flat_", "scratch_")) and False):
            dest = None
            if (inasm or not asm_only) and op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")):
                dest = regs_of(t.split(",")[0])
            outstanding.append((dest, n, t))
    return bad

if __name__ == "__main__":
    rc = 0
    asm_only = "--all" not in sys.argv
    for p in [a for a in sys.argv[1:] if a != "--all"]:
        bad = lint(p, asm_only)
        print(f"{p}: {len(bad)} access(es) to in-flight registers")
        for n, t, regs, n0, t0 in bad[:40]:
            print(f"  line {n}: {t}\n      touches {regs} in flight since line {n0}: {t0}")
        rc |= bool(bad)
    sys.exit(rc)
