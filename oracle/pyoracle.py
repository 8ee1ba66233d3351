"""ctypes wrappers around the two CPU checkers -- TEST INFRASTRUCTURE ONLY.

  OraclePort : oracle/liboracle_port.so   (this repo's plain-C restatement, tinympc_oracle.c)
  OracleRef  : oracle/_ref/libtinympc_ref.so (the reference's own core, built by `make -C oracle ref`)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Both classes expose the same surface so a test can drive either with the same code:
  setup-from-Problem, set_x0 / set_x_ref / set_u_ref / set_bound_constraints / update_settings,
  solve, phase functions, get(name), put(name, array), stats().
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TINYMPC_ORACLE_PORT_LIB: the sanitizer build of the same source, tools/asan_check.py)
PORT_LIB = os.environ.get("TINYMPC_ORACLE_PORT_LIB") or os.path.join(_HERE, "liboracle_port.so")
REF_LIB = os.path.join(_HERE, "_ref", "libtinympc_ref.so")
REF_ZEROINIT_LIB = os.path.join(_HERE, "_ref", "libtinympc_ref_zeroinit.so")  # see oracle/Makefile: adaptive rho only

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _f(a) -> np.ndarray:
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def _shape_of(name: str, nx: int, nu: int, N: int):
    xs = {"x", "q", "p", "v", "vnew", "g", "x_min", "x_max", "Xref", "sol_x", "vcnew", "gc", "vlnew", "gl"}
    us = {"u", "r", "d", "z", "znew", "y", "u_min", "u_max", "Uref", "sol_u", "zcnew", "yc", "zlnew", "yl"}
    if name in xs:
        return (nx, N)
    if name in us:
        return (nu, N - 1)
    return {"Kinf": (nu, nx), "Pinf": (nx, nx), "Quu_inv": (nu, nu), "AmBKt": (nx, nx), "C1": (nu, nu),
            "C2": (nx, nx), "Adyn": (nx, nx), "Bdyn": (nx, nu), "Q": (nx,), "R": (nu,), "fdyn": (nx,),
            "APf": (nx,), "BPf": (nu,), "dKinf_drho": (nu, nx), "dPinf_drho": (nx, nx), "dC1_drho": (nu, nu),
            "dC2_drho": (nx, nx)}[name]


def port_available() -> bool:
    return os.path.exists(PORT_LIB)


def ref_available() -> bool:
    return os.path.exists(REF_LIB)


class _Base:
    nx: int
    nu: int
    N: int

    def get(self, name: str) -> np.ndarray:
        shape = _shape_of(name, self.nx, self.nu, self.N)
        out = np.zeros(shape, dtype=np.float64, order="F")
        n = self._get(name.encode(), _p(out), out.size)
        if n != out.size:
            raise KeyError(f"{name}: accessor returned {n}, expected {out.size}")
        return out

    def put(self, name: str, arr) -> None:
        a = _f(arr)
        rc = self._put(name.encode(), _p(a), a.size)
        if rc != 0:
            raise KeyError(f"{name}: put returned {rc}")

    def solution(self):
        return self.get("sol_x"), self.get("sol_u")

    def load_problem(self, prob, settings: dict | None = None):
        """Drive the checker the way TinyMPC.m drives the MEX (TinyMPC.m:42-124, 256-278)."""
        st = dict(abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=100, check_termination=1,
                  en_state_bound=0, en_input_bound=0)
        if prob.has_bounds():
            self.set_bound_constraints(*prob.expanded_bounds())
            st["en_state_bound"] = 1
            st["en_input_bound"] = 1
        if settings:
            st.update(settings)
        self.update_settings(**st)
        if prob.x_ref is not None:
            self.set_x_ref(prob.x_ref)
        if prob.u_ref is not None:
            self.set_u_ref(prob.u_ref)
        self.set_x0(prob.x0)
        return self


class OraclePort(_Base):
    """Plain-C restatement (oracle/tinympc_oracle.c)."""

    kind = "port"

    def __init__(self, prob):
        if not port_available():
            raise FileNotFoundError(f"{PORT_LIB} missing: run `make -C oracle port` (or __graft_entry__.build())")
        L = C.CDLL(PORT_LIB)
        self.L = L
        L.orc_setup.restype = C.c_void_p
        L.orc_setup.argtypes = [_dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        for fn in ("orc_set_x0", "orc_set_x_ref", "orc_set_u_ref"):
            getattr(L, fn).argtypes = [C.c_void_p, _dp]
        L.orc_set_bound_constraints.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.orc_set_cache_terms.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.orc_set_cone_constraints.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp]
        L.orc_set_linear_constraints.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_update_settings.argtypes = [C.c_void_p, C.c_double, C.c_double] + [C.c_int] * 8
        for fn in ("orc_forward_pass", "orc_update_slack", "orc_update_dual", "orc_update_linear_cost",
                   "orc_backward_pass_grad", "orc_reset_workspace"):
            getattr(L, fn).argtypes = [C.c_void_p]
            getattr(L, fn).restype = None
        L.orc_termination_condition.argtypes = [C.c_void_p]
        L.orc_solve.argtypes = [C.c_void_p]
        L.orc_get.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        L.orc_put.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        L.orc_get_stats.argtypes = [C.c_void_p, _ip, _dp]
        L.orc_set_iter.argtypes = [C.c_void_p, C.c_int]
        L.orc_bench_solves.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int]
        L.orc_bench_solves.restype = C.c_long
        L.orc_solve_batch.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp, _ip, _ip, _dp]
        self.nx, self.nu, self.N = prob.nx, prob.nu, prob.N
        A, B, Q, R = _f(prob.A), _f(prob.B), _f(prob.Q), _f(prob.R)
        fd = _f(prob.fdyn) if prob.fdyn is not None else None
        self.h = L.orc_setup(_p(A), _p(B), _p(fd) if fd is not None else None, _p(Q), _p(R),
                             float(prob.rho), prob.nx, prob.nu, prob.N)
        if not self.h:
            raise RuntimeError("orc_setup failed")
        self._settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=1000, check_termination=1,
                              en_state_bound=1, en_input_bound=1, en_state_soc=0, en_input_soc=0,
                              en_state_linear=0, en_input_linear=0)
        if prob.cones:
            self.set_cone_constraints(**prob.cones)
        if prob.linear:
            self.set_linear_constraints(**prob.linear)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    def _get(self, name, ptr, cap):
        return self.L.orc_get(self.h, name, ptr, cap)

    def _put(self, name, ptr, n):
        return self.L.orc_put(self.h, name, ptr, n)

    def set_x0(self, x0):
        a = _f(x0)
        return self.L.orc_set_x0(self.h, _p(a))

    def set_x_ref(self, xr):
        a = _f(xr)
        return self.L.orc_set_x_ref(self.h, _p(a))

    def set_u_ref(self, ur):
        a = _f(ur)
        return self.L.orc_set_u_ref(self.h, _p(a))

    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        a, b, c, d = _f(x_min), _f(x_max), _f(u_min), _f(u_max)
        self._settings["en_state_bound"] = 1
        self._settings["en_input_bound"] = 1
        return self.L.orc_set_bound_constraints(self.h, _p(a), _p(b), _p(c), _p(d))

    def set_cache_terms(self, Kinf, Pinf, Quu_inv, AmBKt):
        a, b, c, d = _f(Kinf), _f(Pinf), _f(Quu_inv), _f(AmBKt)
        return self.L.orc_set_cache_terms(self.h, _p(a), _p(b), _p(c), _p(d))

    def set_cone_constraints(self, Acx, qcx, cx, Acu, qcu, cu):
        ax, qx = np.asarray(Acx, dtype=np.int32), np.asarray(qcx, dtype=np.int32)
        au, qu = np.asarray(Acu, dtype=np.int32), np.asarray(qcu, dtype=np.int32)
        cxa, cua = _f(cx), _f(cu)
        if ax.size:
            self._settings["en_state_soc"] = 1
        if au.size:
            self._settings["en_input_soc"] = 1
        return self.L.orc_set_cone_constraints(
            self.h, ax.size, ax.ctypes.data_as(_ip), qx.ctypes.data_as(_ip), _p(cxa),
            au.size, au.ctypes.data_as(_ip), qu.ctypes.data_as(_ip), _p(cua))

    def set_linear_constraints(self, Alin_x, blin_x, Alin_u, blin_u):
        ax, bx, au, bu = _f(Alin_x), _f(blin_x), _f(Alin_u), _f(blin_u)
        nlx = ax.shape[0] if ax.size else 0
        nlu = au.shape[0] if au.size else 0
        if nlx:
            self._settings["en_state_linear"] = 1
        if nlu:
            self._settings["en_input_linear"] = 1
        return self.L.orc_set_linear_constraints(self.h, nlx, _p(ax), _p(bx), nlu, _p(au), _p(bu))

    def update_settings(self, **kw):
        self._settings.update(kw)
        s = self._settings
        self.L.orc_update_settings(self.h, float(s["abs_pri_tol"]), float(s["abs_dua_tol"]), int(s["max_iter"]),
                                   int(s["check_termination"]), int(s["en_state_bound"]), int(s["en_input_bound"]),
                                   int(s["en_state_soc"]), int(s["en_input_soc"]), int(s["en_state_linear"]),
                                   int(s["en_input_linear"]))

    def reset_workspace(self):
        self.L.orc_reset_workspace(self.h)

    def set_adaptive_rho(self, enabled: bool, rho_min: float = 1.0, rho_max: float = 100.0, clip: bool = True):
        self.L.orc_set_adaptive_rho.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        self.L.orc_set_adaptive_rho(self.h, int(enabled), float(rho_min), float(rho_max), int(clip))

    def set_sensitivity(self, dK, dP):
        a, b = _f(dK), _f(dP)
        self.L.orc_set_sensitivity.argtypes = [C.c_void_p, _dp, _dp]
        self.L.orc_set_sensitivity(self.h, _p(a), _p(b))

    def rho_adaptation(self) -> float:
        self.L.orc_rho_adaptation.restype = C.c_double
        self.L.orc_rho_adaptation.argtypes = [C.c_void_p]
        return float(self.L.orc_rho_adaptation(self.h))

    def solve(self) -> int:
        return self.L.orc_solve(self.h)

    def forward_pass(self):
        self.L.orc_forward_pass(self.h)

    def update_slack(self):
        self.L.orc_update_slack(self.h)

    def update_dual(self):
        self.L.orc_update_dual(self.h)

    def update_linear_cost(self):
        self.L.orc_update_linear_cost(self.h)

    def backward_pass_grad(self):
        self.L.orc_backward_pass_grad(self.h)

    def termination_condition(self) -> int:
        return self.L.orc_termination_condition(self.h)

    def set_iter(self, it: int):
        self.L.orc_set_iter(self.h, it)

    def stats(self) -> dict:
        i = (C.c_int * 8)()
        d = (C.c_double * 8)()
        self.L.orc_get_stats(self.h, i, d)
        return dict(iter=i[0], status=i[1], solved=i[2], sol_iter=i[3], riccati_iters=i[4],
                    pri_x=d[0], dua_x=d[1], pri_u=d[2], dua_u=d[3], rho=d[4])

    def bench_solves(self, x0s, reps: int) -> int:
        a = _f(x0s)
        return int(self.L.orc_bench_solves(self.h, _p(a), a.shape[1], reps))

    def solve_batch(self, x0s):
        a = _f(x0s)
        cnt = a.shape[1]
        sx = np.zeros((self.nx, self.N, cnt), order="F")
        su = np.zeros((self.nu, self.N - 1, cnt), order="F")
        it = np.zeros(cnt, dtype=np.int32)
        stt = np.zeros(cnt, dtype=np.int32)
        res = np.zeros((4, cnt), order="F")
        self.L.orc_solve_batch(self.h, _p(a), cnt, _p(sx), _p(su), it.ctypes.data_as(_ip),
                               stt.ctypes.data_as(_ip), _p(res))
        return sx, su, it, stt, res


class OracleRef(_Base):
    """The reference's own compiled core (box constraints only; old snapshot API)."""

    kind = "reference"

    def codegen(self, output_dir: str) -> int:
        """The reference's own tiny_codegen() (codegen.cpp:56-68) on this solver."""
        return int(self.L.ref_codegen(self.h, str(output_dir).encode()))

    def set_adaptive_rho(self, enabled: bool, rho_min: float = 1.0, rho_max: float = 100.0, clip: bool = True):
        self.L.ref_set_adaptive_rho(self.h, int(enabled), float(rho_min), float(rho_max), int(clip))

    def __init__(self, prob, zeroinit: bool = False):
        lib = REF_ZEROINIT_LIB if zeroinit else REF_LIB
        if not os.path.exists(lib):
            raise FileNotFoundError(f"{lib} missing: run `make -C oracle ref ref_zeroinit` where /root/reference exists")
        if prob.fdyn is not None and np.any(prob.fdyn != 0) or prob.cones or prob.linear:
            raise ValueError("the reference snapshot has no fdyn / cone / linear support (SURVEY.md section 0.1)")
        L = C.CDLL(lib)
        self.L = L
        L.ref_setup.restype = C.c_void_p
        L.ref_setup.argtypes = [_dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int]
        L.ref_free.argtypes = [C.c_void_p]
        for fn in ("ref_set_x0", "ref_set_x_ref", "ref_set_u_ref"):
            getattr(L, fn).argtypes = [C.c_void_p, _dp]
        L.ref_set_bounds.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.ref_set_cache_terms.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.ref_update_settings.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_solve.argtypes = [C.c_void_p, C.c_int]
        for fn in ("ref_forward_pass", "ref_update_slack", "ref_update_dual", "ref_update_linear_cost",
                   "ref_backward_pass_grad"):
            getattr(L, fn).argtypes = [C.c_void_p]
            getattr(L, fn).restype = None
        L.ref_termination_condition.argtypes = [C.c_void_p]
        L.ref_set_iter.argtypes = [C.c_void_p, C.c_int]
        L.ref_get.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        L.ref_put.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        L.ref_get_stats.argtypes = [C.c_void_p, _ip, _dp]
        L.ref_bench_solves.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int]
        L.ref_bench_solves.restype = C.c_long
        if hasattr(L, "ref_bench_closed_loop"):
            L.ref_bench_closed_loop.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp]
            L.ref_bench_closed_loop.restype = C.c_long
        if hasattr(L, "ref_bench_closed_loop_samples"):
            L.ref_bench_closed_loop_samples.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp, _dp]
            L.ref_bench_closed_loop_samples.restype = C.c_long
        if hasattr(L, "ref_codegen"):
            L.ref_codegen.argtypes = [C.c_void_p, C.c_char_p]
            L.ref_set_adaptive_rho.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        self.nx, self.nu, self.N = prob.nx, prob.nu, prob.N
        nx, nu, N = self.nx, self.nu, self.N
        A, B, Q, R = _f(prob.A), _f(prob.B), _f(prob.Q), _f(prob.R)
        big = 1e17
        xmn, xmx = _f(np.full((nx, N), -big)), _f(np.full((nx, N), big))
        umn, umx = _f(np.full((nu, N - 1), -big)), _f(np.full((nu, N - 1), big))
        self.h = L.ref_setup(_p(A), _p(B), _p(Q), _p(R), float(prob.rho), nx, nu, N,
                             _p(xmn), _p(xmx), _p(umn), _p(umx), 0)
        if not self.h:
            raise RuntimeError("ref_setup failed")
        self._settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=1000, check_termination=1,
                              en_state_bound=1, en_input_bound=1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_free(self.h)
            self.h = None

    def _get(self, name, ptr, cap):
        return self.L.ref_get(self.h, name, ptr, cap)

    def _put(self, name, ptr, n):
        return self.L.ref_put(self.h, name, ptr, n)

    def set_x0(self, x0):
        a = _f(x0)
        return self.L.ref_set_x0(self.h, _p(a))

    def set_x_ref(self, xr):
        a = _f(xr)
        return self.L.ref_set_x_ref(self.h, _p(a))

    def set_u_ref(self, ur):
        a = _f(ur)
        return self.L.ref_set_u_ref(self.h, _p(a))

    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        a, b, c, d = _f(x_min), _f(x_max), _f(u_min), _f(u_max)
        self._settings["en_state_bound"] = 1
        self._settings["en_input_bound"] = 1
        return self.L.ref_set_bounds(self.h, _p(a), _p(b), _p(c), _p(d))

    def set_cache_terms(self, Kinf, Pinf, Quu_inv, AmBKt):
        a, b, c, d = _f(Kinf), _f(Pinf), _f(Quu_inv), _f(AmBKt)
        return self.L.ref_set_cache_terms(self.h, _p(a), _p(b), _p(c), _p(d))

    def update_settings(self, **kw):
        self._settings.update({k: v for k, v in kw.items() if k in self._settings})
        s = self._settings
        self.L.ref_update_settings(self.h, float(s["abs_pri_tol"]), float(s["abs_dua_tol"]), int(s["max_iter"]),
                                   int(s["check_termination"]), int(s["en_state_bound"]), int(s["en_input_bound"]))

    def reset_workspace(self):
        nx, nu, N = self.nx, self.nu, self.N
        for n in ("x", "q", "p", "v", "vnew", "g"):
            self.put(n, np.zeros((nx, N)))
        for n in ("u", "r", "d", "z", "znew", "y"):
            self.put(n, np.zeros((nu, N - 1)))

    def solve(self) -> int:
        return self.L.ref_solve(self.h, 0)

    def forward_pass(self):
        self.L.ref_forward_pass(self.h)

    def update_slack(self):
        self.L.ref_update_slack(self.h)

    def update_dual(self):
        self.L.ref_update_dual(self.h)

    def update_linear_cost(self):
        self.L.ref_update_linear_cost(self.h)

    def backward_pass_grad(self):
        self.L.ref_backward_pass_grad(self.h)

    def termination_condition(self) -> int:
        return self.L.ref_termination_condition(self.h)

    def set_iter(self, it: int):
        self.L.ref_set_iter(self.h, it)

    def stats(self) -> dict:
        i = (C.c_int * 8)()
        d = (C.c_double * 8)()
        self.L.ref_get_stats(self.h, i, d)
        return dict(iter=i[0], status=i[1], solved=i[2], sol_iter=i[3],
                    pri_x=d[0], dua_x=d[1], pri_u=d[2], dua_u=d[3], rho=d[4])

    def bench_solves(self, x0s, reps: int) -> int:
        a = _f(x0s)
        return int(self.L.ref_bench_solves(self.h, _p(a), a.shape[1], reps))

    def bench_closed_loop(self, x0, ticks: int, skip: int = 0):
        """`ticks` warm-started ticks x+ = A x + B u0 inside the compiled shim; returns (iterations, seconds) of the timed ones."""
        a = _f(np.asarray(x0, dtype=np.float64).reshape(-1, 1).copy())
        sec = C.c_double(0.0)
        its = int(self.L.ref_bench_closed_loop(self.h, _p(a), int(ticks), int(skip), C.byref(sec)))
        return its, float(sec.value), a.ravel()

    def bench_closed_loop_samples(self, x0, ticks: int, skip: int = 0):
        """As bench_closed_loop, plus the per-tick durations (us) of the counted ticks: (iterations, seconds, x, tick_us)."""
        a = _f(np.asarray(x0, dtype=np.float64).reshape(-1, 1).copy())
        sec = C.c_double(0.0)
        per = np.zeros(int(ticks))
        its = int(self.L.ref_bench_closed_loop_samples(self.h, _p(a), int(ticks), int(skip), C.byref(sec), _p(per)))
        return its, float(sec.value), a.ravel(), per[int(skip):].copy()

    @staticmethod
    def bench_ticks_many(solvers, x0s, ticks: int, skip: int = 0):
        """`ticks` closed-loop ticks of len(solvers) independent instances on one thread (ref_bench_ticks_many): returns
        (iterations of the counted ticks, per-tick microseconds for all instances together [ticks], final states)."""
        L = solvers[0].L
        L.ref_bench_ticks_many.argtypes = [C.POINTER(C.c_void_p), C.c_int, _dp, C.c_int, C.c_int, _dp]
        L.ref_bench_ticks_many.restype = C.c_long
        hs = (C.c_void_p * len(solvers))(*[s.h for s in solvers])
        x = _f(np.asarray(x0s, dtype=np.float64).copy())
        per = np.zeros(int(ticks))
        its = int(L.ref_bench_ticks_many(hs, len(solvers), _p(x), int(ticks), int(skip), _p(per)))
        return its, per, x

    @staticmethod
    def bench_setup(prob, reps: int = 20):
        """The reference's tiny_setup (tiny_api.cpp:21-122, precompute included) `reps` times on one thread: per-call microseconds."""
        L = C.CDLL(REF_LIB)
        L.ref_bench_setup.argtypes = [_dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp]
        nx, nu, N = prob.nx, prob.nu, prob.N
        A, B, Q, R = _f(prob.A), _f(prob.B), _f(prob.Q), _f(prob.R)
        big = 1e17
        xmn, xmx = _f(np.full((nx, N), -big)), _f(np.full((nx, N), big))
        umn, umx = _f(np.full((nu, N - 1), -big)), _f(np.full((nu, N - 1), big))
        us = np.zeros(int(reps))
        ok = L.ref_bench_setup(_p(A), _p(B), _p(Q), _p(R), float(prob.rho), nx, nu, N, _p(xmn), _p(xmx), _p(umn), _p(umx), int(reps), _p(us))
        assert ok == reps
        return us
