"""Build-time lint of the generated gfx950 code: the DPP data hazard of the fused mat-vec chain (tools/isa_lint.py).

The chain is emitted in blocks so that the scheduler can fill the gaps between them (tinympc_sweep.h,
tinympc_solve_d_chain.h). __graft_entry__.build() runs the same lint on the assembly of the objects it links and fails
on a violation; this test checks the lint itself and every solve kernel's assembly (the build's own `.s` files when
they are there, a fresh `hipcc -S` otherwise)."""
from __future__ import annotations

import os
import re
import shutil
import subprocess

import pytest
from conftest import ROOT

from tools.inflight_lint import lint_text as _inflight
from tools.isa_lint import lint as _lint

CSRC = os.path.join(ROOT, "tinympc-matlab_amd", "csrc")
SOURCES = ["tinympc_solve.hip", "tinympc_solve_b.hip", "tinympc_solve_c.hip", "tinympc_solve_fam.hip", "tinympc_solve_adapt.hip",
           "tinympc_solve_d.hip", "tinympc_solve_dr.hip", "tinympc_solve_dw.hip", "tinympc_solve_dx.hip"]


def test_lint_catches_a_planted_hazard():
    good = "s_nop 1\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
    assert _lint("v_mov_b32_e32 v4, v9\n" + good) == (1, [])
    n, bad = _lint("v_mov_b32_e32 v4, v9\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:0 row_mask:0xf bank_mask:0xf\n")
    assert n == 1 and len(bad) == 1
    n, bad = _lint("v_mov_b32_e32 v5, v9\n s_nop 0\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    assert len(bad) == 1  # one wait state is not enough
    n, bad = _lint(".LBB0_1:\n v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    assert len(bad) == 1
    # a branch target inside the window is followed to the branches that jump to it (round 4): safe on both paths ...
    dpp = " v_fmac_f64_dpp v[2:3], v[4:5], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
    both = "v_mov_b32_e32 v4, v9\n s_nop 0\n s_cbranch_execz .LBB0_2\n v_add_u32_e32 v1, v1, v1\n v_add_u32_e32 v1, v1, v1\n.LBB0_2:\n" + dpp
    assert _lint(both) == (1, [])
    # ... a write one wait state in front of the branch is not (the branch itself is the only wait state on that path) ...
    n, bad = _lint("v_mov_b32_e32 v4, v9\n s_cbranch_execz .LBB0_2\n v_add_u32_e32 v1, v1, v1\n v_add_u32_e32 v1, v1, v1\n.LBB0_2:\n" + dpp)
    assert len(bad) == 1 and "branch at line" in bad[0][2]
    # ... nor is a write on the fall-through path right above the label
    n, bad = _lint("s_cbranch_execz .LBB0_2\n s_nop 1\n v_mov_b32_e32 v5, v9\n.LBB0_2:\n" + dpp)
    assert len(bad) == 1
    # (nothing falls through an unconditional branch)
    assert _lint("s_nop 1\n s_cbranch_execz .LBB0_2\n v_mov_b32_e32 v5, v9\n s_branch .LBB0_3\n.LBB0_2:\n" + dpp + ".LBB0_3:\n s_endpgm\n") == (1, [])
    assert _lint("v_mov_b32_e32 v4, v9\n v_add_u32_e32 v1, v1, v1\n v_add_u32_e32 v1, v1, v1\n" + good.split("\n")[1] + "\n")[1] == []
    # the 64-bit DPP move (layout E's masked gathers) reads its source through the crossbar too
    n, bad = _lint("v_mul_f64 v[4:5], v[8:9], v[8:9]\n v_mov_b64_dpp v[2:3], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf\n")
    assert n == 1 and len(bad) == 1
    assert _lint("v_mul_f64 v[4:5], v[8:9], v[8:9]\n s_and_saveexec_b64 s[0:1], s[2:3]\n s_nop 1\n v_mov_b64_dpp v[2:3], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf\n") == (1, [])


def test_inflight_lint_catches_a_register_touched_before_its_read_landed():
    """What tools/fuzz_wide_families.py found in round 5 (profiles/r05_inflight_bug.txt): an LDS read issued from inline asm, whose
    destination the compiler parked in an AGPR and reused before the block's s_waitcnt."""
    read = ";;#ASMSTART\n ds_read_b64 v[0:1], v186 offset:0xa00\n ;;#ASMEND\n"
    wait = ";;#ASMSTART\n v_add_f64 v[8:9], v[8:9], v[8:9]\n s_waitcnt lgkmcnt(0)\n ;;#ASMEND\n"
    assert _inflight(read + wait + "v_mov_b32_e32 v2, v0\n") == []
    bad = _inflight(read + "v_accvgpr_write_b32 a49, v1\n" + wait)
    assert len(bad) == 1 and bad[0][2] == [("v", 1)]
    assert len(_inflight(read + "v_cndmask_b32_e64 v0, v6, v192, s[0:1]\n" + wait)) == 1   # a write is as bad as a read
    # LDS operations retire in order: lgkmcnt(1) behind a second read has retired the first one only
    two = read + ";;#ASMSTART\n ds_read_b64 v[2:3], v186 offset:0xb00\n ;;#ASMEND\n s_waitcnt lgkmcnt(1)\n"
    assert _inflight(two + "v_mov_b32_e32 v9, v0\n") == [] and len(_inflight(two + "v_mov_b32_e32 v9, v2\n")) == 1
    # the compiler's own reads are its own business (it waits where it must) unless asked for
    own = "ds_read_b64 v[0:1], v186 offset:2560\n v_mov_b32_e32 v9, v0\n"
    assert _inflight(own) == [] and len(_inflight(own, asm_only=False)) == 1


def test_no_kernel_source_issues_an_lds_read_the_compiler_cannot_see():
    """Every LDS read with a result goes through tinympc_sweep.h's lds_read_issued_here (a volatile load: its place is pinned, its
    arrival tracked) and every wait it relies on through lds_reads_landed (__builtin_amdgcn_s_waitcnt) -- no `ds_read` and no
    `s_waitcnt lgkmcnt` in inline asm text, except fused with a workgroup barrier."""
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".hip", ".h")):
            continue
        code = "\n".join(line.split("//")[0] for line in open(os.path.join(CSRC, name)).read().splitlines())
        strings = " ".join(re.findall(r'"([^"\n]*)"', code))  # (asm text holds no escaped quotes)
        assert "ds_read" not in strings and "ds_bpermute" not in strings, name
        # (a wait FUSED with a workgroup barrier is the one exception: it guards this wavefront's LDS stores, not a read's result)
        assert "lgkmcnt" not in strings.replace("s_waitcnt lgkmcnt(0)\\n\\ts_barrier", ""), name


def test_every_solve_source_is_linted_by_the_build():
    import __graft_entry__ as ge
    assert set(SOURCES) <= set(ge.HIP_LINTED), "a solve kernel source is missing from the build's ISA lint"
    on_disk = {f for f in os.listdir(CSRC) if f.startswith("tinympc_solve") and f.endswith(".hip")}
    no_dpp_chain = {"tinympc_solve_m.hip"}  # the matrix-core kernel: no DPP operand anywhere
    # layouts E and F exist as specialisations only: the ones BASELINE config 4 needs are compiled in and linted by the build
    # (HIP_BUILTINS), every other one is compiled at run time from the same source with a hazard s_nop in front of every chain block
    run_time_only = {e[1] for e in ge.HIP_BUILTINS}
    assert run_time_only == {"tinympc_solve_e.hip", "tinympc_solve_f.hip"}
    assert [e[0] for e in ge.HIP_BUILTINS][:3] == ["k_builtin_e_rocket100", "k_builtin_f_rocket100", "k_builtin_f_rocket100_session"]
    assert {e[0] for e in ge.HIP_BUILTINS[3:]} == {"k_builtin_f_%s%s" % (n, v) for n in ("cartpole20", "quadrotor50") for v in ("", "_var", "_session")}
    assert ge.builtin_guarded() <= {e[0] for e in ge.HIP_BUILTINS}
    assert on_disk - no_dpp_chain - run_time_only == set(SOURCES), "new solve kernel source: add it to SOURCES here and to HIP_LINTED in __graft_entry__.py"
    for f in no_dpp_chain:
        assert "_dpp" not in open(os.path.join(CSRC, f)).read()


@pytest.mark.parametrize("source", SOURCES)
def test_generated_code_has_no_dpp_hazard(source, tmp_path):
    import __graft_entry__ as ge
    built = ge.device_asm_path(source)
    if os.path.exists(built) and os.path.getmtime(built) >= os.path.getmtime(os.path.join(CSRC, source)):
        text = open(built).read()
    else:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("hipcc not available and no build assembly")
        out = tmp_path / (source + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"),
                        "-I" + CSRC, "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, source)], check=True, timeout=900)
        text = out.read_text()
    checked, bad = _lint(text)
    assert checked > 100, "no fused DPP mat-vec found: did the kernel change?"
    assert not bad, f"{len(bad)} DPP hazard(s) in {source}, first: {bad[:3]}"


@pytest.mark.parametrize("source,nx,nu,N,vreg,wps,ct", [("tinympc_solve_d.hip", 12, 4, 20, 19, 2, 1), ("tinympc_solve_d.hip", 6, 3, 30, 29, 2, 1), ("tinympc_solve_d.hip", 4, 1, 45, 30, 2, 1),
                                                       ("tinympc_solve_dw.hip", 20, 6, 15, 14, 2, 1), ("tinympc_solve_dw.hip", 13, 4, 25, 16, 2, 1),
                                                       ("tinympc_solve_dx.hip", 30, 10, 9, 8, 2, 1), ("tinympc_solve_dx.hip", 40, 12, 8, 7, 2, 1),
                                                       ("tinympc_solve_d.hip", 12, 4, 100, 86, 1, 1), ("tinympc_solve_dw.hip", 24, 8, 60, 59, 1, 1),
                                                       ("tinympc_solve_dx.hip", 48, 16, 40, 39, 1, 1),
                                                       ("tinympc_solve_d.hip", 12, 4, 25, 24, 2, 0), ("tinympc_solve_dw.hip", 24, 8, 30, 10, 2, 0),
                                                       ("tinympc_solve_dx.hip", 48, 16, 20, 19, 1, 0)])
def test_run_time_specialisations_have_no_dpp_hazard(source, nx, nu, N, vreg, wps, ct, tmp_path):
    """tinympc_jit.hip compiles these sources with -DTINY_JIT ... through hiprtc on the GPU box, where nothing lints the
    result; the same specialisations are compiled here with hipcc (same front end, same flags) and linted."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "jit.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    "-DTINY_JIT=1", f"-DTINY_JIT_NX={nx}", f"-DTINY_JIT_NU={nu}", f"-DTINY_JIT_N={N}", f"-DTINY_JIT_VREG={vreg}",
                    f"-DTINY_JIT_WPS={wps}", f"-DTINY_JIT_CT={ct}", "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, source)], check=True, timeout=900)
    text = out.read_text()
    checked, bad = _lint(text)
    assert checked > 50 and not bad, bad[:3]
    if wps == 2:
        assert "vgpr_spill_count: 0" in text, "the specialisation spills"
    else:  # 512 registers: values beyond the 256 architectural ones sit in accumulation registers, counted as spills; no scratch
        assert ".private_segment_fixed_size: 0" in text, "the specialisation spills to scratch memory"


@pytest.mark.parametrize("nx,nu,N,vreg,ct,variant", [(6, 3, 10, 9, 0, "FAM"), (6, 3, 20, 19, 1, "FAM"), (12, 4, 15, 14, 0, "FAM"),
                                                     (12, 4, 50, 49, 1, "ADAPT"), (4, 1, 20, 19, 0, "ADAPT"), (12, 4, 15, 14, 1, "ADAPT")])
def test_family_specialisations_of_layout_d_have_no_dpp_hazard(nx, nu, N, vreg, ct, variant, tmp_path):
    """Layout D with the cone / linear families (-DTINY_JIT_FAM=1) or adaptive rho (-DTINY_JIT_ADAPT=1), one wavefront per
    SIMD: their code adds mat-vecs (the same fused DPP chain) between the sweep blocks -- lint what hiprtc will build on
    the GPU box."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "jit_fam.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    "-DTINY_JIT=1", f"-DTINY_JIT_NX={nx}", f"-DTINY_JIT_NU={nu}", f"-DTINY_JIT_N={N}", f"-DTINY_JIT_VREG={vreg}",
                    "-DTINY_JIT_WPS=1", f"-DTINY_JIT_CT={ct}", f"-DTINY_JIT_{variant}=1", "-S", "--cuda-device-only", "-o", str(out),
                    os.path.join(CSRC, "tinympc_solve_d.hip")], check=True, timeout=900)
    text = out.read_text()
    checked, bad = _lint(text)
    assert checked > 200 and not bad, bad[:3]
    assert ".private_segment_fixed_size: 0" in text, "the specialisation spills to scratch memory"


def test_compiled_in_specialisations_are_linted_and_do_not_spill():
    """BASELINE config 4's kernels (layout E for batches, layout F for one instance) and layout F for configs 2 and 3 are compiled in with
    the BARE chain blocks: the build's ISA lint is what stands between them and the DPP hazard (builtin_guarded(): the ones it refused
    bare), and their descriptors must show no scratch."""
    import __graft_entry__ as ge
    for name, source, defs in ge.HIP_BUILTINS:
        path = ge.builtin_asm_path(name)
        if not os.path.exists(path):
            pytest.skip("no build assembly (run __graft_entry__.build())")
        text = open(path).read()
        checked, bad = _lint(text)
        assert checked > 100 and not bad, (name, bad[:3])
        assert ".amdhsa_kernel " + name in text, name
        if name.endswith("_session"):
            # the resident session kernel may spill in its RARE paths (mailbox, reference refresh, the write-back when it leaves:
            # once per tick or per session); its sweeps -- the blocks made of DPP FMAs -- must not touch scratch
            body = re.search(r"^%s:(.*?)^\.Lfunc_end" % re.escape(name), text, re.S | re.M).group(1)
            sweeps = [b for b in re.split(r"^\.LBB\d+_\d+:", body, flags=re.M) if b.count("v_fmac_f64_dpp") >= 9]
            assert len(sweeps) >= 4, name
            for b in sweeps:
                assert "scratch_" not in b, f"{name}: a sweep block of the session kernel uses scratch"
        else:
            assert ".private_segment_fixed_size: 0" in text and "vgpr_spill_count: 0" in text, name
        # bare blocks: only the back-to-back chains of pass 1 / the carry recurrences (fwd_plain) keep their own s_nop; the sweep
        # blocks proper start without one (the run-time builds put one in front of every chain)
        chains = len(re.findall(r"v_fmac_f64_dpp [^\n]* row_newbcast:0 ", text))
        with_nop = len(re.findall(r"s_nop 1\n\t\.p2align 3\n\tv_fmac_f64_dpp", text))
        if name in ge.builtin_guarded():  # (the lint found a hazard in its bare form: it keeps the guard in front of every chain)
            assert with_nop >= chains // 2, (name, chains, with_nop)  # (the backward chains start behind their own v_mov)
        else:
            assert chains >= 2 * with_nop, (name, chains, with_nop)


E_ROCKET = dict(nround=1, ncone=2, cones="{0,0,2},{0,6,8}", nlx=1, nlu=0)
# lds: the element form's placement (gc, gl, lx in LDS), or "k" = the families one knot per lane (KFamilies), "kd" = ... and d in registers
@pytest.mark.parametrize("nx,nu,N,ct,wpg,S,fam,lds", [
    (6, 3, 100, 0, 8, 13, E_ROCKET, "kd"),        # BASELINE config 4 as bench.py runs it (compiled in with these options, but bare blocks)
    (6, 3, 100, 1, 8, 13, E_ROCKET, "kd"),        # ... with constant tables
    (6, 3, 100, 0, 8, 13, E_ROCKET, (1, 1, 1)),   # the element form of the families, all three arrays in LDS
    (6, 3, 44, 0, 4, 11, E_ROCKET, "k"),          # four wavefronts per workgroup
    (12, 4, 60, 1, 8, 8, dict(nround=2, ncone=3, cones="{0,0,2},{0,12,15},{1,1,4}", nlx=2, nlu=1), (0, 1, 1)),  # overlapping cones, rows on both sides
    (8, 4, 60, 1, 8, 8, dict(nround=2, ncone=3, cones="{0,0,2},{0,8,11},{1,1,4}", nlx=2, nlu=1), "k"),          # ... one knot per lane (12 rows, d in LDS: what the plan picks; it refuses 16 rows)
    (6, 3, 100, 1, 8, 13, dict(nround=1, ncone=1, cones="{0,6,8}", nlx=12, nlu=5), "k"),                        # many linear rows: the run-time loop (with d in registers it spills: the next placement)
    (12, 4, 200, 1, 8, 25, None, (0, 0, 0)),      # box path only, a horizon beyond layout D's plans
])
def test_layout_e_specialisations_have_no_dpp_hazard_and_no_scratch(nx, nu, N, ct, wpg, S, fam, lds, tmp_path):
    """tinympc_solve_e.hip exists only as run-time specialisations (tinympc_jit.hip -> hiprtc on the GPU box); the same
    specialisations compiled here with hipcc: DPP hazards (the sweep chains AND the EXEC-masked gathers of the families),
    no scratch, register count within the plan."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    f = fam or dict(nround=0, ncone=0, cones="{-1,0,0}", nlx=0, nlu=0)
    out = tmp_path / "jit_e.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    "-DTINY_JIT=1", f"-DTINY_JIT_NX={nx}", f"-DTINY_JIT_NU={nu}", f"-DTINY_JIT_N={N}", f"-DTINY_JIT_CT={ct}", f"-DTINY_JIT_FAM={1 if fam else 0}",
                    f"-DTINY_JIT_E_WPG={wpg}", f"-DTINY_JIT_E_WPS={wpg // 4}", f"-DTINY_JIT_E_S={S}", f"-DTINY_JIT_E_NROUND={f['nround']}", f"-DTINY_JIT_E_NCONE={f['ncone']}",
                    f"-DTINY_JIT_E_CONES={f['cones']}", f"-DTINY_JIT_E_NLX={f['nlx']}", f"-DTINY_JIT_E_NLU={f['nlu']}",
                    *([f"-DTINY_JIT_E_GC_LDS=0", "-DTINY_JIT_E_GL_LDS=0", "-DTINY_JIT_E_LX_LDS=0", "-DTINY_JIT_E_KFAM=1", f"-DTINY_JIT_E_DREG={1 if lds == 'kd' else 0}"]
                      if isinstance(lds, str) else [f"-DTINY_JIT_E_GC_LDS={lds[0]}", f"-DTINY_JIT_E_GL_LDS={lds[1]}", f"-DTINY_JIT_E_LX_LDS={lds[2]}"]),
                    "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "tinympc_solve_e.hip")], check=True, timeout=900)
    text = out.read_text()
    checked, bad = _lint(text)
    assert checked > 200 and not bad, bad[:3]
    assert ".private_segment_fixed_size: 0" in text, "the specialisation spills to scratch memory"
    if wpg == 8:
        assert "vgpr_spill_count: 0" in text


def test_embedded_kernel_sources_are_the_sources_on_disk():
    """libtinympc_hip.so specialises the sources embedded at build time (build/hip/tinympc_jit_embedded.inc): what build() wrote
    must be the files of csrc/, byte for byte, and every file the run-time translation units include must be on the list."""
    import re
    import __graft_entry__ as ge
    inc = ge.write_embedded_sources()
    text = open(inc).read()
    chunks = re.split(r"static const char kEmbeddedText\d+\[\] =\n", text)[1:]
    assert len(chunks) == len(ge.JIT_EMBEDDED)
    for name, chunk in zip(ge.JIT_EMBEDDED, chunks):
        src = open(os.path.join(CSRC, name)).read()
        body = "".join(re.findall(r'R"TINYSRC\((.*?)\)TINYSRC"', chunk, re.S))
        assert body == src, name
        for inc_name in re.findall(r'#include "([^"]+)"', src):
            assert inc_name in ge.JIT_EMBEDDED, f"{name} includes {inc_name}, which is not embedded"


@pytest.mark.parametrize("nx,nu,N,ct,wpg,S,fam", [
    (6, 3, 100, 0, 7, 4, E_ROCKET),     # BASELINE config 4, one instance (the build lints the same instance)
    (6, 3, 100, 1, 7, 4, dict(nround=2, ncone=3, cones="{0,0,2},{0,6,8},{1,1,4}", nlx=12, nlu=5)),  # overlapping cones, many rows (run-time loop)
    (6, 3, 10, 0, 2, 2, E_ROCKET),      # short horizon: five chunks of two slots, a one-slot last chunk
    (12, 4, 50, 1, 7, 2, None),         # box path
    (4, 1, 120, 1, 8, 4, None),         # eight wavefronts
])
def test_layout_f_specialisations_have_no_dpp_hazard_and_no_scratch(nx, nu, N, ct, wpg, S, fam, tmp_path):
    """tinympc_solve_f.hip (the specialised latency kernel) as hiprtc will build it on the GPU box, compiled here with hipcc: DPP
    hazards in the sweep chains, the carry scans' mat-vecs (the early-clobber accumulator: a shared register there was this lint's
    find) and the EXEC-masked gathers; no scratch."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    f = fam or dict(nround=0, ncone=0, cones="{-1,0,0}", nlx=0, nlu=0)
    out = tmp_path / "jit_f.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    "-DTINY_JIT=1", f"-DTINY_JIT_NX={nx}", f"-DTINY_JIT_NU={nu}", f"-DTINY_JIT_N={N}", f"-DTINY_JIT_CT={ct}", f"-DTINY_JIT_FAM={1 if fam else 0}",
                    f"-DTINY_JIT_F_WPG={wpg}", f"-DTINY_JIT_F_S={S}", f"-DTINY_JIT_E_NROUND={f['nround']}", f"-DTINY_JIT_E_NCONE={f['ncone']}",
                    f"-DTINY_JIT_E_CONES={f['cones']}", f"-DTINY_JIT_E_NLX={f['nlx']}", f"-DTINY_JIT_E_NLU={f['nlu']}",
                    "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "tinympc_solve_f.hip")], check=True, timeout=900)
    text = out.read_text()
    checked, bad = _lint(text)
    assert checked > 100 and not bad, bad[:3]
    assert ".private_segment_fixed_size: 0" in text and "vgpr_spill_count: 0" in text, "the specialisation spills"


def test_compiled_in_kernels_do_not_live_in_scratch():
    """Register arrays that outgrow the register file end up in scratch memory silently (round 3 found the 64-lane forms of the
    families and adaptive-rho kernels there: 200 spilled registers, 10-60x slower). The build keeps the device assembly of every
    object it links; its kernel descriptors say how much scratch each kernel uses. Allowed: nothing, except the entries below."""
    import re

    import __graft_entry__ as ge
    allowed = {
        # k_admm_solve_fam<64, 64, false, false>: 14 registers in the variant that reads its tables from L2 (44 bytes per lane)
        ("tinympc_solve_fam.hip", "_ZN7tinympc16k_admm_solve_famILi64ELi64ELb0ELb0EEEvNS_11SolveParamsE"): 64,
        # k_admm_solve_c<16, 16, 4, SESSION, FAM>: the resident variant with the families at 16 rows sits at 512 registers; five dwords of
        # its prologue's values live in scratch across the tick loop (three stores at the start, three loads per tick in the write-out,
        # none in a sweep -- checked by hand in the assembly when the mailbox's checksum code was added, round 5)
        ("tinympc_solve_c.hip", "_ZN7tinympc14k_admm_solve_cILi16ELi16ELi4ELb1ELb1EEEvNS_11SolveParamsE"): 32,
        # k_admm_solve_m<13..15, false>: the per-knot-table variants at sixteen wavefronts per workgroup (128 registers): 14 registers
        **{("tinympc_solve_m.hip", "_ZN7tinympc14k_admm_solve_mILi%dELb0ELb0EEEvNS_11SolveParamsE" % r): 64 for r in (13, 14, 15)},
        # k_admm_solve_m<R, false, true>: the families variants (HBM-bound; the phase between the sweeps keeps four slots' loads in
        # flight) hold a few values of the iteration's outer scope in scratch -- 3-11 registers up to R = 12, 20-32 at sixteen
        # wavefronts per workgroup; none of the blocks that hold matrix instructions touches scratch (checked below). (The phase as a
        # separate function -- its own register allocation -- was slower: 81-136 registers saved and restored around every call.)
        **{("tinympc_solve_m.hip", "_ZN7tinympc14k_admm_solve_mILi%dELb0ELb1EEEvNS_11SolveParamsE" % r): 128 for r in range(5, 33)},
    }
    seen = refill = 0
    for source in ge.HIP_SOURCES + [e[0] for e in ge.HIP_BUILTINS]:
        built = ge.device_asm_path(source) if source.endswith(".hip") else ge.builtin_asm_path(source)
        if not os.path.exists(built):
            pytest.skip("no build assembly (run __graft_entry__.build())")
        text = open(built).read()
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
            seen += 1
            size = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
            if source == "tinympc_solve_dr.hip" and "k_admm_solve_d_refill" in m.group(1):
                # k_admm_solve_d_refill<...> (tinympc_solve_dr.hip): the slot-refill variants spill in their RARE paths (write-back
                # and refill of a row: once per instance); their sweeps -- the blocks made of DPP FMAs -- must not touch scratch
                refill += 1
                body = re.search(r"^%s:(.*?)^\.Lfunc_end" % re.escape(m.group(1)), text, re.S | re.M).group(1)
                blocks = re.split(r"^\.LBB\d+_\d+:", body, flags=re.M)
                sweeps = [b for b in blocks if b.count("v_fmac_f64_dpp") >= 15]
                assert len(sweeps) >= 2, m.group(1)
                for b in sweeps:
                    assert "scratch_" not in b, f"{m.group(1)}: a sweep block of the slot-refill variant uses scratch"
                continue
            if m.group(1).endswith("_session"):
                continue  # (rare paths may spill: test_compiled_in_specialisations_are_linted_and_do_not_spill checks its sweeps)
            if source == "tinympc_solve_m.hip" and size > 0:
                body = re.search(r"^%s:(.*?)^\.Lfunc_end" % re.escape(m.group(1)), text, re.S | re.M).group(1)
                for b in re.split(r"^\.LBB\d+_\d+:", body, flags=re.M):
                    assert not ("v_mfma" in b and "scratch_" in b), f"{m.group(1)}: a GEMM block uses scratch"
            assert size <= allowed.get((source, m.group(1)), 0), f"{source}: {m.group(1)} uses {size} bytes of scratch per lane"
    assert seen > 50 and refill >= 3


def _hiprtc_compile(source_text: str, options: list[str]) -> tuple[int, str]:
    """hiprtcCompileProgram through ctypes (no GPU needed): -> (result code, log)."""
    import ctypes as C
    lib = None
    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        try:
            lib = C.CDLL(name)
            break
        except OSError:
            continue
    if lib is None:
        pytest.skip("libhiprtc.so not found")
    prog = C.c_void_p()
    lib.hiprtcCreateProgram.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    assert lib.hiprtcCreateProgram(C.byref(prog), source_text.encode(), b"spec.hip", 0, None, None) == 0
    arr = (C.c_char_p * len(options))(*[o.encode() for o in options])
    lib.hiprtcCompileProgram.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]
    rc = lib.hiprtcCompileProgram(prog, len(options), arr)
    n = C.c_size_t()
    lib.hiprtcGetProgramLogSize.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.hiprtcGetProgramLogSize(prog, C.byref(n))
    buf = C.create_string_buffer(n.value + 1)
    lib.hiprtcGetProgramLog.argtypes = [C.c_void_p, C.c_char_p]
    lib.hiprtcGetProgramLog(prog, buf)
    lib.hiprtcDestroyProgram.argtypes = [C.POINTER(C.c_void_p)]
    lib.hiprtcDestroyProgram(C.byref(prog))
    return rc, buf.value.decode(errors="replace")


@pytest.mark.parametrize("source,defs", [
    ("tinympc_solve_d.hip", "-DTINY_JIT_NX=6 -DTINY_JIT_NU=3 -DTINY_JIT_N=30 -DTINY_JIT_VREG=29 -DTINY_JIT_WPS=2 -DTINY_JIT_WPG=4 -DTINY_JIT_CT=1 -DTINY_JIT_FAM=0 -DTINY_JIT_ADAPT=0"),
    ("tinympc_solve_d.hip", "-DTINY_JIT_NX=12 -DTINY_JIT_NU=4 -DTINY_JIT_N=20 -DTINY_JIT_VREG=19 -DTINY_JIT_WPS=2 -DTINY_JIT_WPG=4 -DTINY_JIT_CT=1 -DTINY_JIT_FAM=0 -DTINY_JIT_ADAPT=0 -DTINY_JIT_REFILL=1"),
    ("tinympc_solve_dw.hip", "-DTINY_JIT_NX=17 -DTINY_JIT_NU=2 -DTINY_JIT_N=10 -DTINY_JIT_VREG=9 -DTINY_JIT_WPS=2 -DTINY_JIT_WPG=4 -DTINY_JIT_CT=1"),
    ("tinympc_solve_dx.hip", "-DTINY_JIT_NX=40 -DTINY_JIT_NU=12 -DTINY_JIT_N=8 -DTINY_JIT_VREG=7 -DTINY_JIT_WPS=2 -DTINY_JIT_WPG=4 -DTINY_JIT_CT=1"),
])
def test_run_time_specialisations_compile_under_hiprtc(source, defs):
    """The run-time specialised kernels are compiled by hiprtc, not hipcc: its built-in headers are a subset (round 3: a host
    declaration outside the __HIPCC_RTC__ guard and a __double2loint in a shared header both passed every hipcc build and
    failed on the GPU box only, where the shape then fell back to a slower kernel). hiprtc needs no GPU: compile here."""
    rc, log = _hiprtc_compile('#include "%s"\n' % source, ["--offload-arch=gfx950", "-O3", "-std=c++17", "-DTINY_JIT=1", "-I" + CSRC] + defs.split())
    assert rc == 0, log[-2000:]
