"""The MEX shim (tinympc-matlab_amd/matlab/tinympc_matlab_mex.cpp) driven verb by verb through a mock of
the MEX C API (tests/mock_mex): the string-verb dispatcher, argument counts and error identifiers of the
reference's MEX function (bindings.cpp:641-692), without MATLAB. CPU tests cover dispatch and errors;
the gpu-marked test runs the reference's one-solve script flow end to end against the golden fixture."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import pytest
from conftest import ROOT, golden, rel_err

# (TINYMPC_MOCK_MEX_LIB: the sanitizer build of shim + mock, tools/asan_check.py)
MOCK = os.environ.get("TINYMPC_MOCK_MEX_LIB") or os.path.join(ROOT, "tests", "mock_mex", "libtinympc_matlab_mock.so")


class Mex:
    def __init__(self):
        if not os.path.exists(MOCK):
            pytest.fail("tests/mock_mex/libtinympc_matlab_mock.so missing: run `python __graft_entry__.py`")
        L = C.CDLL(MOCK)
        self.L = L
        L.mxCreateDoubleMatrix.restype = C.c_void_p
        L.mxCreateDoubleMatrix.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        L.mxCreateInt32Matrix.restype = C.c_void_p
        L.mxCreateInt32Matrix.argtypes = [C.c_size_t, C.c_size_t]
        L.mxCreateString.restype = C.c_void_p
        L.mxCreateString.argtypes = [C.c_char_p]
        L.mxGetPr.restype = C.POINTER(C.c_double)
        L.mxGetPr.argtypes = [C.c_void_p]
        L.mxGetData.restype = C.c_void_p
        L.mxGetData.argtypes = [C.c_void_p]
        L.mxGetM.restype = C.c_size_t
        L.mxGetM.argtypes = [C.c_void_p]
        L.mxGetN.restype = C.c_size_t
        L.mxGetN.argtypes = [C.c_void_p]
        L.mxDestroyArray.argtypes = [C.c_void_p]
        L.mock_mex_call.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        L.mock_mex_last_id.restype = C.c_char_p
        L.mock_mex_last_msg.restype = C.c_char_p

    def to_mx(self, v):
        if isinstance(v, str):
            return self.L.mxCreateString(v.encode())
        a = np.asarray(v)
        if a.dtype == np.int32:
            a = np.atleast_1d(a)
            h = self.L.mxCreateInt32Matrix(a.size, 1)
            C.memmove(self.L.mxGetData(h), a.ctypes.data, a.nbytes)
            return h
        a = np.asfortranarray(np.atleast_2d(np.asarray(v, dtype=np.float64)))
        if np.ndim(v) == 1:
            a = np.asfortranarray(a.reshape(-1, 1))
        m, n = (a.shape if a.size else (0, 0))
        h = self.L.mxCreateDoubleMatrix(m, n, 0)
        if a.size:
            C.memmove(self.L.mxGetPr(h), a.ctypes.data, a.nbytes)
        return h

    def call(self, verb, *args, nlhs=0):
        """tinympc_matlab(verb, args...) -> (error_id or None, [outputs as numpy])"""
        ins = [self.to_mx(verb)] + [self.to_mx(a) for a in args]
        prhs = (C.c_void_p * len(ins))(*ins)
        plhs = (C.c_void_p * max(nlhs, 4))()
        rc = self.L.mock_mex_call(nlhs, plhs, len(ins), prhs)
        outs = []
        if rc == 0:
            for k in range(nlhs):
                h = plhs[k]
                m, n = self.L.mxGetM(h), self.L.mxGetN(h)
                outs.append(np.ctypeslib.as_array(self.L.mxGetPr(h), shape=(n, m)).T.copy())
        for h in ins:
            self.L.mxDestroyArray(h)
        return (self.L.mock_mex_last_id().decode() if rc else None), outs


@pytest.fixture(scope="module")
def mex():
    return Mex()


def test_unknown_verb_and_missing_verb(mex):
    err, _ = mex.call("frobnicate")
    assert err == "TinyMPC:InvalidFunction"  # bindings.cpp:685
    prhs = (C.c_void_p * 1)()
    plhs = (C.c_void_p * 1)()
    assert mex.L.mock_mex_call(0, plhs, 0, prhs) == 1
    assert mex.L.mock_mex_last_id() == b"TinyMPC:InvalidInput"  # bindings.cpp:642-644


def test_verbs_before_setup_are_not_initialized(mex):
    mex.call("reset", 0.0)
    for verb, args in (("set_x0", (np.zeros(4), 0.0)), ("solve", (0.0,)), ("get_solution", (0.0,)), ("get_stats", (0.0,)),
                       ("set_x_ref", (np.zeros((4, 20)), 0.0)), ("update_settings", tuple([0.0] * 15))):
        err, _ = mex.call(verb, *args)
        assert err == "TinyMPC:NotInitialized", verb  # bindings.cpp:112-114 etc.


def test_argument_count_errors(mex):
    assert mex.call("setup", np.eye(2))[0] == "TinyMPC:InvalidInput"          # bindings.cpp:49-51
    assert mex.call("set_x0", np.zeros(4))[0] == "TinyMPC:InvalidInput"       # :108-110
    assert mex.call("solve")[0] == "TinyMPC:InvalidInput"                     # :213-215
    assert mex.call("update_settings", 1.0, 2.0)[0] == "TinyMPC:InvalidInput"  # :549-551
    assert mex.call("reset", 0.0)[0] is None                                  # resetting nothing is fine (:539)


def test_setup_rejects_arrays_that_do_not_match_nx_nu(mex, pkg):
    """The C ABI reads nx*nx, nx*nu ... doubles behind the raw pointers; the shim must not forward arrays of another
    shape (the reference gets that from its Eigen conversions + tiny_setup's checks, bindings.cpp:60-85)."""
    p = pkg.problems.cartpole()
    ok = [p.A, p.B, np.zeros((4, 1)), p.Q, p.R, 1.0, 4.0, 1.0, 20.0, 0.0]
    for idx, bad in ((0, np.eye(3)), (1, np.zeros((4, 2))), (2, np.zeros((3, 1))), (3, np.eye(5)), (4, np.eye(2))):
        args = list(ok)
        args[idx] = bad
        assert mex.call("setup", *args, nlhs=1)[0] == "TinyMPC:InvalidInput", idx


def test_setup_without_gpu_raises_setup_failed(mex, pkg):
    if pkg.device_count() > 0:
        pytest.skip("a HIP device is visible")
    p = pkg.problems.cartpole()
    err, _ = mex.call("setup", p.A, p.B, np.zeros((4, 1)), p.Q, p.R, 1.0, 4.0, 1.0, 20.0, 0.0, nlhs=1)
    assert err == "TinyMPC:SetupFailed"  # no device: fail loudly, no CPU fallback


@pytest.mark.gpu
def test_one_solve_script_through_the_mex_verbs(mex, pkg):
    """examples/cartpole_example_one_solve.m + bounds, expressed as the MEX calls TinyMPC.m makes."""
    g = golden("cartpole_box_tol")
    nx, nu, N = 4, 1, 20
    err, out = mex.call("setup", g["A"], g["B"], np.zeros((nx, 1)), g["Q"], g["R"], float(g["rho"]), float(nx), float(nu),
                        float(N), 0.0, nlhs=1)
    assert err is None and out[0][0, 0] == 0
    settings = [1e-4, 1e-4, 100.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.1, 10.0, 1.0, 0.0]
    assert mex.call("update_settings", *settings)[0] is None
    assert mex.call("set_bound_constraints", g["x_min"], g["x_max"], g["u_min"], g["u_max"], 0.0)[0] is None
    settings[4] = settings[5] = 1.0
    assert mex.call("update_settings", *settings)[0] is None
    assert mex.call("set_x0", g["x0"], 0.0)[0] is None
    err, out = mex.call("solve", 0.0, nlhs=1)
    assert err is None and out[0][0, 0] == 0
    err, (x, u) = mex.call("get_solution", 0.0, nlhs=2)
    assert err is None and x.shape == (nx, N) and u.shape == (nu, N - 1)
    assert rel_err(x, g["sol_x"]) < 1e-9 and rel_err(u, g["sol_u"]) < 1e-9
    err, (it, status, pri_x, pri_u) = mex.call("get_stats", 0.0, nlhs=4)
    assert int(it[0, 0]) == int(g["iter"]) and int(status[0, 0]) == 1
    np.testing.assert_allclose([pri_x[0, 0], pri_u[0, 0]], g["residuals"][[0, 2]], rtol=1e-5, atol=1e-12)
    # wrong-shaped bounds and references are rejected with the reference's identifiers
    assert mex.call("set_bound_constraints", g["x_min"][:, :5], g["x_max"], g["u_min"], g["u_max"], 0.0)[0] == "TinyMPC:SetBoundConstraintsFailed"
    assert mex.call("set_x_ref", np.zeros((3, N)), 0.0)[0] == "TinyMPC:SetXRefFailed"
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        err, out = mex.call("codegen", os.path.join(tmp, "gen"), 0.0, nlhs=1)
        assert err is None and out[0][0, 0] == 0 and os.path.exists(os.path.join(tmp, "gen", "src", "tiny_data.cpp"))
        open(os.path.join(tmp, "blocker"), "w").close()
        err, out = mex.call("codegen", os.path.join(tmp, "blocker"), 0.0, nlhs=1)
        assert err is None and out[0][0, 0] != 0  # status != 0 -> TinyMPC.m raises TinyMPC:CodegenFailed
    assert mex.call("reset", 0.0)[0] is None
    assert mex.call("solve", 0.0)[0] == "TinyMPC:NotInitialized"


@pytest.mark.gpu
def test_resident_solves_through_the_mex_verbs(mex, pkg):
    """The 'set_resident' verb (extension): the same three-verb loop, the solves on the resident kernel; equal to the launched loop."""
    prob = pkg.problems.quadrotor(50)
    nx, nu, N = prob.nx, prob.nu, prob.N
    settings = [1e-3, 1e-3, 60.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.1, 10.0, 1.0, 0.0]
    runs = []
    for resident in (False, True):
        err, _ = mex.call("setup", prob.A, prob.B, np.zeros((nx, 1)), prob.Q, prob.R, float(prob.rho), float(nx), float(nu), float(N), 0.0, nlhs=1)
        assert err is None
        xmin, xmax = np.repeat(prob.x_min[:, None], N, 1), np.repeat(prob.x_max[:, None], N, 1)
        umin, umax = np.repeat(prob.u_min[:, None], N - 1, 1), np.repeat(prob.u_max[:, None], N - 1, 1)
        assert mex.call("set_bound_constraints", xmin, xmax, umin, umax, 0.0)[0] is None
        assert mex.call("update_settings", *settings)[0] is None
        if resident:
            assert mex.call("set_resident", 1.0)[0] is None
        x, us = prob.x0.copy(), []
        for k in range(8):
            assert mex.call("set_x0", x, 0.0)[0] is None
            assert mex.call("solve", 0.0, nlhs=1)[0] is None
            err, (xs, u) = mex.call("get_solution", 0.0, nlhs=2)
            assert err is None
            us.append(u.copy())
            x = prob.A @ x + prob.B @ u[:, 0]
        runs.append(np.array(us))
        assert mex.call("reset", 0.0)[0] is None
    np.testing.assert_array_equal(runs[0], runs[1])
    assert mex.call("set_resident", 1.0)[0] == "TinyMPC:NotInitialized"


@pytest.mark.gpu
def test_cache_script_through_the_mex_verbs(mex, pkg):
    """tests/test_cache.m of the reference (setup with N = 2, then compute_cache_terms), plus the two other
    class-side recursions, as the MEX calls this build's TinyMPC.m makes for them."""
    import matlab_class_oracle as M
    p = pkg.problems.cartpole(2, False)
    err, out = mex.call("setup", p.A, p.B, np.zeros((4, 1)), p.Q, p.R, 1.0, 4.0, 1.0, 2.0, 0.0, nlhs=1)
    assert err is None and out[0][0, 0] == 0
    err, (K, P_, Qi, Am) = mex.call("compute_cache_terms", 0.0, nlhs=4)
    assert err is None and K.shape == (1, 4) and P_.shape == (4, 4) and Qi.shape == (1, 1) and Am.shape == (4, 4)
    Ko, Po, Qio, Amo, _ = M.compute_cache_terms(p.A, p.B, p.Q, p.R, 1.0)
    assert rel_err(K, Ko) < 1e-8 and rel_err(P_, Po) < 1e-8 and rel_err(Qi, Qio) < 1e-8 and rel_err(Am, Amo) < 1e-8
    err, (K2, P2, C1, C2) = mex.call("solve_lqr", 2.5, 0.0, nlhs=4)
    assert err is None and rel_err(P2, M.solve_lqr(p.A, p.B, p.Q, p.R, 2.5)[1]) < 1e-9
    err, outs = mex.call("compute_sensitivity", 0.0, nlhs=4)
    assert err is None and [o.shape for o in outs] == [(1, 4), (4, 4), (1, 1), (4, 4)]
    assert mex.call("compute_cache_terms")[0] == "TinyMPC:InvalidInput"
    assert mex.call("reset", 0.0)[0] is None
    assert mex.call("compute_cache_terms", 0.0, nlhs=4)[0] == "TinyMPC:NotInitialized"


def test_missing_trailing_arguments_the_reference_does_not_check(mex):
    """bindings.cpp:188-195, 408-414, 450-459 read prhs[0..] without looking at nrhs: a call with too few arguments dereferences
    past the argument list there (undefined behaviour, typically a MATLAB crash). The shim raises TinyMPC:InvalidInput instead --
    before it looks at the handle, so this needs no GPU. (INTEGRATION.md section 2a lists every such deviation.)"""
    mex.call("reset", 0.0)
    z = np.zeros((4, 20))
    assert mex.call("set_bound_constraints", z, z, np.zeros((1, 19)))[0] in ("TinyMPC:InvalidInput", "TinyMPC:NotInitialized")
    assert mex.call("set_linear_constraints", np.zeros((1, 4)), np.zeros(1))[0] in ("TinyMPC:InvalidInput", "TinyMPC:NotInitialized")
    assert mex.call("set_cone_constraints", np.array([0], dtype=np.int32), np.array([3], dtype=np.int32), np.array([0.5]))[0] in (
        "TinyMPC:InvalidInput", "TinyMPC:NotInitialized")


@pytest.mark.gpu
def test_sloppy_inputs_the_reference_lets_through(mex, pkg):
    """One case per deviation of INTEGRATION.md section 2a -- inputs the reference accepts (it prints and assigns, or reads past its
    argument list) and this build refuses, with the identifier a MATLAB caller sees:
      wrong-length x0            tiny_api.cpp:238-240 perror()s and assigns        -> TinyMPC:SetX0Failed
      wrong-shape Xref / Uref    tiny_api.cpp:250-254, 262-266 print and assign   -> TinyMPC:SetXRefFailed / SetURefFailed
      too few arguments          bindings.cpp:188-195, 408-414, 450-459 unchecked -> TinyMPC:InvalidInput
      cone index / length lists of different lengths (bindings.cpp:450-466 forwards them) -> TinyMPC:InvalidInput
    and after every refusal the solver is untouched: the solve that follows equals the golden one."""
    g = golden("cartpole_box_tol")
    nx, nu, N = 4, 1, 20
    err, out = mex.call("setup", g["A"], g["B"], np.zeros((nx, 1)), g["Q"], g["R"], float(g["rho"]), float(nx), float(nu), float(N), 0.0, nlhs=1)
    assert err is None and out[0][0, 0] == 0
    settings = [1e-4, 1e-4, 100.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.1, 10.0, 1.0, 0.0]
    assert mex.call("set_bound_constraints", g["x_min"], g["x_max"], g["u_min"], g["u_max"], 0.0)[0] is None
    assert mex.call("update_settings", *settings)[0] is None
    assert mex.call("set_x0", g["x0"], 0.0)[0] is None
    # --- the refusals
    assert mex.call("set_x0", np.zeros(nx + 1), 0.0)[0] == "TinyMPC:SetX0Failed"
    assert mex.call("set_x0", np.zeros(nx - 1), 0.0)[0] == "TinyMPC:SetX0Failed"
    assert mex.call("set_x_ref", np.zeros((nx, N - 1)), 0.0)[0] == "TinyMPC:SetXRefFailed"
    assert mex.call("set_x_ref", np.zeros((nx + 1, N)), 0.0)[0] == "TinyMPC:SetXRefFailed"
    assert mex.call("set_u_ref", np.zeros((nu, N)), 0.0)[0] == "TinyMPC:SetURefFailed"
    assert mex.call("set_u_ref", np.zeros((nu + 1, N - 1)), 0.0)[0] == "TinyMPC:SetURefFailed"
    assert mex.call("set_bound_constraints", g["x_min"], g["x_max"], g["u_min"])[0] == "TinyMPC:InvalidInput"
    assert mex.call("set_linear_constraints", np.zeros((1, nx)), np.zeros(1), np.zeros((0, 0)))[0] == "TinyMPC:InvalidInput"
    i32 = lambda *v: np.array(v, dtype=np.int32)
    assert mex.call("set_cone_constraints", i32(0), i32(3), np.array([0.5]), i32(), i32())[0] == "TinyMPC:InvalidInput"
    assert mex.call("set_cone_constraints", i32(0, 1), i32(3), np.array([0.5]), i32(), i32(), np.zeros(0))[0] == "TinyMPC:InvalidInput"
    # --- nothing of the above reached the solver
    err, out = mex.call("solve", 0.0, nlhs=1)
    assert err is None and out[0][0, 0] == 0
    err, (x, u) = mex.call("get_solution", 0.0, nlhs=2)
    assert rel_err(x, g["sol_x"]) < 1e-9 and rel_err(u, g["sol_u"]) < 1e-9
    err, (it, status, _, _) = mex.call("get_stats", 0.0, nlhs=4)
    assert int(it[0, 0]) == int(g["iter"]) and int(status[0, 0]) == 1
    assert mex.call("reset", 0.0)[0] is None
