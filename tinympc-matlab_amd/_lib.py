"""ctypes binding of libtinympc_hip.so (include/tinympc_hip.h). No torch, no numpy math: plumbing only."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TINYMPC_HIP_LIBRARY lets a developer point at an alternative build of the SAME library (kernel A/B runs).
LIB_PATH = os.environ.get("TINYMPC_HIP_LIBRARY") or os.path.join(_HERE, "libtinympc_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
Handle = C.c_void_p

OK = 0
ERR_INVALID_INPUT = -1
ERR_NOT_INITIALIZED = -2
ERR_HIP = -3
ERR_UNSUPPORTED = -4
ERR_NOT_IMPLEMENTED = -5
ERR_NO_DEVICE = -6
ERR_ALLOC = -7

# MEX error identifiers the reference raises for the same conditions (bindings.cpp)
MEX_ERROR_IDS = {
    ERR_INVALID_INPUT: "TinyMPC:InvalidInput",
    ERR_NOT_INITIALIZED: "TinyMPC:NotInitialized",
    ERR_HIP: "TinyMPC:Exception",
    ERR_UNSUPPORTED: "TinyMPC:SetupFailed",
    ERR_NOT_IMPLEMENTED: "TinyMPC:InvalidFunction",
    ERR_NO_DEVICE: "TinyMPC:SetupFailed",
    ERR_ALLOC: "TinyMPC:SetupFailed",
}

class CodegenData(C.Structure):
    """struct tinympc_codegen_data (include/tinympc_hip.h), for the host-only tinympc_codegen_emit()."""
    _fields_ = ([("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int), ("iter", C.c_int), ("solved", C.c_int), ("rho", C.c_double)]
                + [(n, c_double_p) for n in ("Kinf", "Pinf", "Quu_inv", "AmBKt", "dKinf_drho", "dPinf_drho", "dC1_drho", "dC2_drho")]
                + [("abs_pri_tol", C.c_double), ("abs_dua_tol", C.c_double)]
                + [(n, C.c_int) for n in ("max_iter", "check_termination", "en_state_bound", "en_input_bound", "adaptive_rho")]
                + [(n, c_double_p) for n in ("Q", "R", "Adyn", "Bdyn", "x_min", "x_max", "u_min", "u_max")])


# name -> (restype, argtypes); mirrors include/tinympc_hip.h one to one
SIGNATURES = {
    "tinympc_last_error": (C.c_char_p, []),
    "tinympc_abi_version": (C.c_int, []),
    "tinympc_device_count": (C.c_int, []),
    "tinympc_setup": (C.c_int, [C.POINTER(Handle), c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tinympc_set_x0": (C.c_int, [Handle, c_double_p, C.c_int, C.c_int]),
    "tinympc_set_x_ref": (C.c_int, [Handle, c_double_p, C.c_int, C.c_int, C.c_int]),
    "tinympc_set_u_ref": (C.c_int, [Handle, c_double_p, C.c_int, C.c_int, C.c_int]),
    "tinympc_set_bound_constraints": (C.c_int, [Handle, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
    "tinympc_solve": (C.c_int, [Handle, C.c_int]),
    "tinympc_get_solution": (C.c_int, [Handle, c_double_p, c_double_p, C.c_int]),
    "tinympc_get_stats": (C.c_int, [Handle, c_int_p, c_int_p, c_double_p, c_double_p, C.c_int]),
    "tinympc_codegen": (C.c_int, [Handle, C.c_char_p, C.c_int]),
    "tinympc_set_sensitivity_matrices": (C.c_int, [Handle, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
    "tinympc_set_cache_terms": (C.c_int, [Handle, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
    "tinympc_set_linear_constraints": (C.c_int, [Handle, c_double_p, c_double_p, C.c_int, c_double_p, c_double_p, C.c_int]),
    "tinympc_set_cone_constraints": (C.c_int, [Handle, c_int_p, c_int_p, c_double_p, C.c_int,
                                               c_int_p, c_int_p, c_double_p, C.c_int]),
    "tinympc_codegen_with_sensitivity": (C.c_int, [Handle, C.c_char_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
    "tinympc_reset": (C.c_int, [C.POINTER(Handle), C.c_int]),
    "tinympc_update_settings": (C.c_int, [Handle, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]),
    "tinympc_print_problem_data": (C.c_int, [Handle]),
    "tinympc_get_cache": (C.c_int, [Handle, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p]),
    "tinympc_get_residuals": (C.c_int, [Handle, c_double_p]),
    "tinympc_codegen_emit": (C.c_int, [C.POINTER(CodegenData), C.c_char_p, C.c_int]),
    "tinympc_compute_cache_terms": (C.c_int, [Handle, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p, C.c_int]),
    "tinympc_solve_lqr": (C.c_int, [Handle, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p]),
    "tinympc_compute_sensitivity": (C.c_int, [Handle, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int]),
    "tinympc_setup_batch": (C.c_int, [C.POINTER(Handle), c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                      C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tinympc_set_x0_batch": (C.c_int, [Handle, c_double_p, C.c_int, C.c_int]),
    "tinympc_set_x0_batch_device": (C.c_int, [Handle, C.c_void_p, C.c_int, C.c_int]),
    "tinympc_reset_workspace": (C.c_int, [Handle]),
    "tinympc_get_rho_batch": (C.c_int, [Handle, c_double_p, C.c_int, C.c_int]),
    "tinympc_get_solution_batch": (C.c_int, [Handle, c_double_p, c_double_p, C.c_int, C.c_int]),
    "tinympc_get_first_controls_batch": (C.c_int, [Handle, c_double_p, C.c_int, C.c_int]),
    "tinympc_get_stats_batch": (C.c_int, [Handle, c_int_p, c_int_p, c_double_p, C.c_int, C.c_int]),
    "tinympc_get_solution_device_ptrs": (C.c_int, [Handle, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "tinympc_solve_async": (C.c_int, [Handle]),
    "tinympc_synchronize": (C.c_int, [Handle]),
    "tinympc_solve_timed": (C.c_int, [Handle, C.POINTER(C.c_float)]),
    "tinympc_solve_queued": (C.c_int, [Handle]),
    "tinympc_collect_kernel_ms": (C.c_int, [Handle, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "tinympc_mpc_step_batch": (C.c_int, [Handle, c_double_p, c_double_p]),
    "tinympc_session_begin": (C.c_int, [Handle]),
    "tinympc_session_step": (C.c_int, [Handle, c_double_p, c_double_p]),
    "tinympc_session_end": (C.c_int, [Handle]),
    "tinympc_set_resident": (C.c_int, [Handle, C.c_int]),
    "tinympc_get_launch_info": (C.c_int, [Handle, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "tinympc_get_layout": (C.c_int, [Handle]),
    "tinympc_prepare": (C.c_int, [Handle]),
    "tinympc_get_jit_info": (C.c_int, [Handle, C.c_char_p, C.c_int]),
    "tinympc_get_stream": (C.c_void_p, [Handle]),
}

# include/tinympc_hip_bench.h: measurement helpers / diagnostics, not part of the drop-in boundary
BENCH_LIB_PATH = os.path.join(_HERE, "libtinympc_bench.so")
BENCH_SIGNATURES = {
    "tinympc_bench_closed_loop": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int, C.c_int, C.c_int,
                                            c_double_p, C.POINTER(C.c_long), c_double_p]),
}
DEBUG_SIGNATURES = {  # (exported by the product library itself: they read the handle's diagnostic counters)
    "tinympc_debug_tick_timing": (C.c_int, [Handle, c_double_p]),
    "tinympc_debug_setup_timing": (C.c_int, [Handle, c_double_p]),
    "tinympc_debug_mail_stamp": (C.c_double, [C.c_double, c_double_p]),
}

_lib = None
_bench_lib = None


class TinyMPCError(RuntimeError):
    """Raised for any non-zero status of the C ABI; `.identifier` is the MEX error id the reference
    would have raised (bindings.cpp), `.code` the TINYMPC_ERR_* value."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.identifier = MEX_ERROR_IDS.get(code, "TinyMPC:Exception")
        super().__init__(f"{self.identifier}: {message}")


def load_library() -> C.CDLL:
    """dlopen libtinympc_hip.so and type every exported entry point. Fails loudly when the library
    (the HIP extension) is missing: there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "The HIP library is the only implementation; there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    missing = [n for n in list(SIGNATURES) + list(DEBUG_SIGNATURES) if not hasattr(lib, n)]
    # (an alternative build named by TINYMPC_HIP_LIBRARY may be an OLDER one -- A/B runs against a previous round: what it lacks stays unbound)
    if missing and not os.environ.get("TINYMPC_HIP_LIBRARY"):
        raise ImportError(f"{LIB_PATH} does not export: {missing}")
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        if name in missing:
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_bench_library() -> C.CDLL:
    """libtinympc_bench.so (csrc/bench/): the closed-loop measurement loop, a caller of the product library's public verbs."""
    global _bench_lib
    if _bench_lib is not None:
        return _bench_lib
    load_library()  # (the product library first: the bench library resolves its verbs against the SAME loaded copy)
    if not os.path.exists(BENCH_LIB_PATH):
        raise FileNotFoundError(f"{BENCH_LIB_PATH} not found: build it with `python __graft_entry__.py`")
    lib = C.CDLL(BENCH_LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in BENCH_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _bench_lib = lib
    return lib


def fast_tick_functions():
    """The two per-tick entry points bound a second time with plain addresses as arguments: a closed-loop tick of ~10 us should not
    spend 2-3 of them building ctypes pointer objects (tinympc.py keeps persistent buffers and passes their addresses)."""
    lib = load_library()
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
    return proto(("tinympc_session_step", lib)), proto(("tinympc_mpc_step_batch", lib))


def last_error() -> str:
    msg = load_library().tinympc_last_error()
    return msg.decode() if msg else ""


def check(code: int) -> None:
    if code != OK:
        raise TinyMPCError(code, last_error())


def abi_version() -> int:
    return int(load_library().tinympc_abi_version())


def device_count() -> int:
    return int(load_library().tinympc_device_count())
