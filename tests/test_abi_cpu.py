"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/tinympc_hip.h
declares, the ctypes table matches the header, the host-side TinyMPC mirror reproduces the reference
class's expansion / option / error behaviour, and -- without a GPU -- the product fails loudly
instead of falling back to anything."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np
import pytest
from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "tinympc_hip.h")
BENCH_HEADER = os.path.join(ROOT, "include", "tinympc_hip_bench.h")


def _declared_symbols(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tinympc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in tinympc_hip.h but not exported"
    assert sorted(pkg._lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert pkg.abi_version() == 1


def test_measurement_helpers_live_outside_the_boundary(pkg):
    """include/tinympc_hip_bench.h: the closed-loop measurement loop is a separate library calling public verbs; the two diagnostics
    are exported by the product library; none of them is declared in the drop-in header."""
    L = pkg._lib
    declared = _declared_symbols(BENCH_HEADER)
    assert declared == sorted(list(L.BENCH_SIGNATURES) + list(L.DEBUG_SIGNATURES))
    assert not set(declared) & set(_declared_symbols())
    lib, bench = pkg.load_library(), L.load_bench_library()
    for name in L.DEBUG_SIGNATURES:
        assert hasattr(lib, name)
    for name in L.BENCH_SIGNATURES:
        assert hasattr(bench, name) and not hasattr(lib, name), name
    # argument validation of the loop needs no GPU: skip >= ticks is refused (it used to return seconds = 0)
    x = np.zeros(4)
    p = x.ctypes.data_as(L.c_double_p)
    h = C.c_void_p(1)  # (never dereferenced: validation comes first)
    assert bench.tinympc_bench_closed_loop(h, 2, 1, 5, p, p, None, p, 5, 5, 0, None, None, None) == L.ERR_INVALID_INPUT
    assert bench.tinympc_bench_closed_loop(None, 2, 1, 5, p, p, None, p, 5, 1, 0, None, None, None) == L.ERR_INVALID_INPUT


def test_session_line_stamp_tells_saturated_controls_from_an_empty_line(pkg):
    """The session protocol accepts a 64-byte line when its stamp fits the seven words read with it. Through most of round 5 the checksum
    was the folded XOR of the words: equal words cancelled, so four saturated controls (+0.4, +0.4, -0.4, -0.4) had the checksum of a
    line of zeros, and a host that read a line's payload just before the resident kernel's answer landed and its stamp just after took
    the zeros (one tick in ~3,000 of tools/fuzz_layout_f.py; profiles/r05_session_stamp_bug.txt). Now order-dependent and multiplicative."""
    lib = pkg.load_library()

    def stamp(seq, words):
        w = np.zeros(7)
        w[:len(words)] = words
        return lib.tinympc_debug_mail_stamp(float(seq), w.ctypes.data_as(pkg._lib.c_double_p))

    zero = stamp(1, [])
    assert 1.0 <= zero < 2.0
    sat = [[0.4, 0.4, -0.4, -0.4], [0.4, -0.4, 0.4, -0.4], [-0.4, -0.4, 0.4, 0.4], [0.4, 0.4], [0.25, 0.25, 0.25, 0.25], [1.0, 1.0, -1.0, -1.0, 0.3, 0.3]]
    for u in sat:
        assert stamp(1, u) != zero, u
    assert stamp(1, [0.4, -0.4]) != stamp(1, [-0.4, 0.4])             # the order of the words counts
    assert stamp(7, [0.1, 0.2]) - 7 != stamp(8, [0.1, 0.2]) - 8      # ... and the sequence number (a line of another tick with the same words)
    assert len({stamp(3, [0.4 * a, 0.4 * b, 0.4 * c, 0.4 * d]) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1) for d in (-1, 1)}) >= 15
    big = 2.0 ** 35 + 12345.0
    assert int(stamp(big, [0.3])) == int(big) and 0.0 <= stamp(big, [0.3]) - big < 1.0  # exact below 2^36
    rng = np.random.default_rng(0)
    seen = {stamp(5, rng.standard_normal(7)) for _ in range(2000)}
    assert len(seen) > 1900  # 16 bits: ~30 collisions expected among 2,000


def test_seventeen_mex_verbs_have_entry_points(pkg):
    """One C entry point per verb of the reference dispatcher (bindings.cpp:641-692)."""
    verbs = ["setup", "set_x0", "set_x_ref", "set_u_ref", "solve", "get_solution", "get_stats", "codegen", "reset",
             "set_bound_constraints", "set_sensitivity_matrices", "set_cache_terms", "codegen_with_sensitivity",
             "update_settings", "print_problem_data", "set_linear_constraints", "set_cone_constraints"]
    assert len(verbs) == 17
    lib = pkg.load_library()
    for v in verbs:
        assert hasattr(lib, "tinympc_" + v), v


def test_null_handle_is_not_initialized(pkg):
    lib = pkg.load_library()
    L = pkg._lib
    x = np.zeros(4)
    assert lib.tinympc_set_x0(None, x.ctypes.data_as(L.c_double_p), 4, 0) == L.ERR_NOT_INITIALIZED
    assert "not initialized" in L.last_error().lower()
    assert lib.tinympc_solve(None, 0) == L.ERR_NOT_INITIALIZED
    h = L.Handle()
    assert lib.tinympc_reset(C.byref(h), 0) == L.OK  # resetting nothing is fine (bindings.cpp:539)


def test_setup_validation_needs_no_gpu(pkg):
    lib, L = pkg.load_library(), pkg._lib
    h = L.Handle()
    a = np.eye(2)
    p = a.ctypes.data_as(L.c_double_p)
    assert lib.tinympc_setup(C.byref(h), p, p, None, p, p, 1.0, 2, 2, 1, 0) == L.ERR_INVALID_INPUT  # N >= 2
    assert lib.tinympc_setup(C.byref(h), None, p, None, p, p, 1.0, 2, 2, 5, 0) == L.ERR_INVALID_INPUT
    assert lib.tinympc_setup(C.byref(h), p, p, None, p, p, 1.0, 500, 20, 5, 0) == L.ERR_UNSUPPORTED  # nx+nu > 512
    assert not h


@pytest.mark.skipif(os.environ.get("TINYMPC_EXPECT_GPU") == "1", reason="GPU box")
def test_no_gpu_fails_loudly_no_cpu_fallback(pkg):
    """In the CPU container the product must refuse to run rather than compute on the host."""
    if pkg.device_count() > 0:
        pytest.skip("a HIP device is visible")
    P = pkg.problems
    prob = P.cartpole()
    s = pkg.TinyMPC()
    with pytest.raises(pkg.TinyMPCError) as ei:
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho)
    assert ei.value.code == pkg._lib.ERR_NO_DEVICE
    assert not s.is_setup
    with pytest.raises(pkg.TinyMPCError) as ei:
        s.solve()
    assert ei.value.code == pkg._lib.ERR_NOT_INITIALIZED  # TinyMPC:NotSetup path (TinyMPC.m:329-334)


def test_missing_library_fails_loudly(pkg, monkeypatch):
    L = pkg._lib
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        L.load_library()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg_dir = os.path.join(ROOT, "tinympc-matlab_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".m")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and "tinympc_oracle" not in text, f
                assert "libtinympc_ref" not in text, f


def test_expand_bounds_and_refs_match_matlab_rules(pkg):
    """expand_bounds / expand_matrix (TinyMPC.m:378-405) against the expanded arrays stored in the
    golden fixtures (what the reference class sent through the MEX boundary)."""
    from conftest import golden

    T = pkg.TinyMPC
    g = golden("quadrotor_box_200")
    nx, nu, N = 12, 4, 50
    np.testing.assert_array_equal(T._expand_bounds(np.full(nx, -5.0), nx, N, -1e17), g["x_min"])
    np.testing.assert_array_equal(T._expand_bounds(0.5, nu, N - 1, 1e17), g["u_max"])
    np.testing.assert_array_equal(T._expand_bounds(np.full((1, nu), -0.5), nu, N - 1, -1e17), g["u_min"])
    g = golden("cartpole_box_tol")
    np.testing.assert_array_equal(T._expand_bounds([], 4, 20, -1e17), g["x_min"])
    np.testing.assert_array_equal(T._expand_bounds(None, 4, 20, 1e17), g["x_max"])
    full = np.arange(12.0).reshape(4, 3)
    assert T._expand_bounds(full, 4, 3, 0.0) is not None and np.array_equal(T._expand_bounds(full, 4, 3, 0.0), full)
    np.testing.assert_array_equal(T._expand_matrix(2.0, 3, 4), np.full((3, 4), 2.0))
    np.testing.assert_array_equal(T._expand_matrix([1.0, 2.0, 3.0], 3, 2), np.array([[1, 1], [2, 2], [3, 3.0]]))
    np.testing.assert_array_equal(T._expand_matrix(np.array([[1.0, 2.0, 3.0]]), 3, 2), np.array([[1, 1], [2, 2], [3, 3.0]]))


def test_constructor_defaults_and_option_parsing(pkg):
    s = pkg.TinyMPC()
    assert s.settings["abs_pri_tol"] == 1e-4 and s.settings["abs_dua_tol"] == 1e-4  # TinyMPC.m:26-27
    assert s.settings["max_iter"] == 100 and s.settings["check_termination"] == 1      # :28-29
    assert not s.settings["en_state_bound"] and not s.settings["en_input_bound"]       # :30-31
    assert s.settings["adaptive_rho_min"] == 0.1 and s.settings["adaptive_rho_max"] == 10.0
    opts = s._parse_options(dict(rho=1.0, verbose=False), dict(rho=2.5, u_min=-1, bogus=3))
    assert opts == dict(rho=2.5, verbose=False)  # unknown keys silently dropped (TinyMPC.m:372)
    with pytest.raises(AssertionError, match="N must be >= 2"):
        s.setup(np.eye(2), np.ones((2, 1)), np.eye(2), np.eye(1), 1)
    with pytest.raises(AssertionError, match="A must be square"):
        s.setup(np.ones((2, 3)), np.ones((2, 1)), np.eye(2), np.eye(1), 5)


def test_shard_range_partitions_exactly(pkg):
    B = pkg.batch
    for total, world in ((65536, 8), (10, 3), (7, 8), (1, 1)):
        spans = [B.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
    assert B.shard_range(65536, 3, 8) == (3 * 8192, 8192)
    with pytest.raises(ValueError):
        B.shard_range(4, 4, 4)


def test_families_flop_model(pkg):
    """problems.flops_per_iteration_families(): the roof bench.py prices the rocket-landing legs against. Hand count for BASELINE
    config 4 (nx=6, nu=3, N=100, one 3-cone per side, one state half-space, fdyn): per side 6 E for slack + dual + linear cost,
    15 per cone and knot, 6 dim + 3 per linear row and knot, 3 nx for the terminal p of each state family, 2 nx + nu per fdyn step."""
    P = pkg.problems
    r = P.rocket(100)
    X, U = 600, 297
    f = r.flops_per_iteration_families()
    assert f["state_cones"] == 6 * X + 18 + 100 * 15 and f["input_cones"] == 6 * U + 99 * 15
    assert f["state_linear"] == 6 * X + 18 + 100 * 39 and f["input_linear"] == 0 and f["fdyn"] == 99 * 15
    assert f["total"] == 17388 and f["with_box"] == 17388 + r.flops_per_iteration() == 61215
    q = P.quadrotor(50).flops_per_iteration_families()
    assert q["total"] == 0 and q["with_box"] == 60848  # the box path alone: SURVEY.md section 8(a)
