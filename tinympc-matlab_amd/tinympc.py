"""TinyMPC -- host-side mirror of the reference's MATLAB class (/root/reference/src/TinyMPC.m).

Same method names, argument meaning, defaults and error behaviour as the `.m` class, over the C ABI
of libtinympc_hip.so instead of the MEX function. MATLAB itself is not available in the build image;
the drop-in `.m` class and MEX shim live in tinympc-matlab_amd/matlab/ and mirror this file.

Reference behaviours reproduced here (file:line in /root/reference/src/TinyMPC.m):
  constructor defaults                     :24-40   tol 1e-4, max_iter 100, check_termination 1, bounds off
  setup(A,B,Q,R,N,'rho',..)                :42-104  dimension asserts, unknown options silently ignored (:368-376),
                                                     bound flags forced off at setup (:71-73), settings pushed (:94-98)
  set_x0 / set_x_ref / set_u_ref           :106-124 x0(:) column; refs broadcast by expand_matrix (:393-405)
  update_settings                          :126-139 only known fields, then pushed
  solve -> always 0                        :141-147
  get_solution -> states / controls        :149-157
  set_bound_constraints                    :256-278 expand with -/+1e17 fill (:378-391), both flags auto-enabled
  set_linear / cone / equality constraints :243-317
  reset                                    :319-325
  check_setup -> TinyMPC:NotSetup          :328-334
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TinyMPCError, c_double_p, c_int_p

BOUND_INF = 1e17


def _f(a) -> np.ndarray:
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


class TinyMPC:
    def __init__(self):
        self.nx = self.nu = self.N = 0
        self.A = self.B = self.Q = self.R = None
        self.rho = 1.0
        self.is_setup = False
        self.batch = 1
        self.settings = dict(abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=100, check_termination=1,
                             en_state_bound=False, en_input_bound=False, en_state_soc=False, en_input_soc=False,
                             en_state_linear=False, en_input_linear=False, adaptive_rho=False,
                             adaptive_rho_min=0.1, adaptive_rho_max=10.0, adaptive_rho_enable_clipping=True)
        self.x_min = self.x_max = self.u_min = self.u_max = None
        self._h = _lib.Handle()
        self._tick = None  # persistent buffers of mpc_step / session_step
        self._L = None

    # ------------------------------------------------------------------ setup
    def setup(self, A, B, Q, R, N, batch: int = 1, device: int = -1, **varargin):
        A, B, Q, R = _f(A), _f(B), _f(Q), _f(R)
        if B.ndim == 1:
            B = _f(B.reshape(-1, 1))
        assert A.ndim == 2 and A.shape[0] == A.shape[1], "A must be square"
        assert A.shape[0] == B.shape[0], "A and B row dimensions must match"
        assert Q.shape[0] == A.shape[0], "Q must match A dimensions"
        assert R.shape[0] == B.shape[1], "R must match B column dimension"
        assert N >= 2, "N must be >= 2"
        self.nx, self.nu, self.N = A.shape[0], B.shape[1], int(N)
        self.A, self.B, self.Q, self.R = A, B, Q, R
        opts = self._parse_options(dict(rho=1.0, fdyn=None, verbose=False, abs_pri_tol=1e-4, abs_dua_tol=1e-4,
                                        max_iter=100, check_termination=1, en_state_bound=False,
                                        en_input_bound=False, adaptive_rho=False, adaptive_rho_min=0.1,
                                        adaptive_rho_max=10.0, adaptive_rho_enable_clipping=True), varargin)
        self.rho = float(opts["rho"])
        for k in ("abs_pri_tol", "abs_dua_tol", "max_iter", "check_termination", "adaptive_rho",
                  "adaptive_rho_min", "adaptive_rho_max", "adaptive_rho_enable_clipping"):
            self.settings[k] = opts[k]
        # Do not enable bound constraints in setup; only via set_bound_constraints (TinyMPC.m:71-73)
        self.settings["en_state_bound"] = False
        self.settings["en_input_bound"] = False
        fdyn = np.zeros(self.nx) if opts["fdyn"] is None else _f(opts["fdyn"]).reshape(-1)
        assert fdyn.size == self.nx, "fdyn must have nx entries"
        fdyn = _f(fdyn)

        self._L = _lib.load_library()
        if self._h:
            self._L.tinympc_reset(C.byref(self._h), 0)  # the reference replaces its global solver (bindings.cpp:92)
        self.batch = int(batch)
        status = self._L.tinympc_setup_batch(C.byref(self._h), _p(A), _p(B), _p(fdyn), _p(Q), _p(R), self.rho,
                                             self.nx, self.nu, self.N, self.batch, int(device), int(bool(opts["verbose"])))
        if status != 0:
            raise TinyMPCError(status, f"Setup failed with status {status}: {_lib.last_error()}")
        self.is_setup = True
        self._push_settings()
        if opts["verbose"]:
            print(f"TinyMPC solver setup successful (nx={self.nx}, nu={self.nu}, N={self.N})")

    # ------------------------------------------------------------------ ingest
    def set_x0(self, x0):
        self._check_setup()
        x0 = _f(np.asarray(x0, dtype=np.float64).reshape(-1))
        _lib.check(self._L.tinympc_set_x0(self._h, _p(x0), x0.size, 0))

    def set_x_ref(self, x_ref):
        self._check_setup()
        xr = _f(self._expand_matrix(x_ref, self.nx, self.N))
        rows, cols = (xr.shape + (1,))[:2] if xr.ndim else (1, 1)
        _lib.check(self._L.tinympc_set_x_ref(self._h, _p(xr), rows, cols, 0))

    def set_u_ref(self, u_ref):
        self._check_setup()
        ur = _f(self._expand_matrix(u_ref, self.nu, self.N - 1))
        rows, cols = (ur.shape + (1,))[:2] if ur.ndim else (1, 1)
        _lib.check(self._L.tinympc_set_u_ref(self._h, _p(ur), rows, cols, 0))

    def update_settings(self, **kw):
        self._check_setup()
        for k, v in kw.items():
            if k in self.settings:  # unknown names are ignored, as isfield() does (TinyMPC.m:129-133)
                self.settings[k] = v
        self._push_settings()

    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        self._check_setup()
        self.x_min = _f(self._expand_bounds(x_min, self.nx, self.N, -BOUND_INF))
        self.x_max = _f(self._expand_bounds(x_max, self.nx, self.N, +BOUND_INF))
        self.u_min = _f(self._expand_bounds(u_min, self.nu, self.N - 1, -BOUND_INF))
        self.u_max = _f(self._expand_bounds(u_max, self.nu, self.N - 1, +BOUND_INF))
        for a, shape in ((self.x_min, (self.nx, self.N)), (self.x_max, (self.nx, self.N)),
                         (self.u_min, (self.nu, self.N - 1)), (self.u_max, (self.nu, self.N - 1))):
            if a.shape != shape:
                raise TinyMPCError(_lib.ERR_INVALID_INPUT, f"bound array has shape {a.shape}, expected {shape}")
        _lib.check(self._L.tinympc_set_bound_constraints(self._h, _p(self.x_min), _p(self.x_max),
                                                         _p(self.u_min), _p(self.u_max), 0))
        self.settings["en_state_bound"] = True
        self.settings["en_input_bound"] = True
        self._push_settings()

    def set_linear_constraints(self, Alin_x, blin_x, Alin_u, blin_u):
        self._check_setup()
        ax, bx = self._mat(Alin_x, self.nx), _f(np.asarray(blin_x, dtype=np.float64).reshape(-1))
        au, bu = self._mat(Alin_u, self.nu), _f(np.asarray(blin_u, dtype=np.float64).reshape(-1))
        nlx = ax.shape[0] if ax.size else 0
        nlu = au.shape[0] if au.size else 0
        _lib.check(self._L.tinympc_set_linear_constraints(self._h, _p(ax), _p(bx), nlx, _p(au), _p(bu), nlu))
        self.settings["en_state_linear"] = bool(nlx)
        self.settings["en_input_linear"] = bool(nlu)
        if nlx or nlu:
            self._push_settings()

    def set_cone_constraints(self, Acx, qcx, cx, Acu, qcu, cu):
        self._check_setup()
        ax, qx = np.asarray(Acx, dtype=np.int32).reshape(-1), np.asarray(qcx, dtype=np.int32).reshape(-1)
        au, qu = np.asarray(Acu, dtype=np.int32).reshape(-1), np.asarray(qcu, dtype=np.int32).reshape(-1)
        cxa, cua = _f(np.asarray(cx, dtype=np.float64).reshape(-1)), _f(np.asarray(cu, dtype=np.float64).reshape(-1))
        _lib.check(self._L.tinympc_set_cone_constraints(
            self._h, ax.ctypes.data_as(c_int_p), qx.ctypes.data_as(c_int_p), _p(cxa), ax.size,
            au.ctypes.data_as(c_int_p), qu.ctypes.data_as(c_int_p), _p(cua), au.size))
        self.settings["en_state_soc"] = bool(ax.size and qx.size and cxa.size)
        self.settings["en_input_soc"] = bool(au.size and qu.size and cua.size)
        if self.settings["en_state_soc"] or self.settings["en_input_soc"]:
            self._push_settings()

    def set_equality_constraints(self, Aeq_x, beq_x, Aeq_u, beq_u):
        """Aeq*s == beq as the two inequalities [A; -A] s <= [b; -b] (TinyMPC.m:296-317)."""
        self._check_setup()
        ax, bx = self._mat(Aeq_x, self.nx), np.asarray(beq_x, dtype=np.float64).reshape(-1)
        au, bu = self._mat(Aeq_u, self.nu), np.asarray(beq_u, dtype=np.float64).reshape(-1)
        lx = (np.vstack([ax, -ax]), np.concatenate([bx, -bx])) if ax.size else (np.zeros((0, self.nx)), np.zeros(0))
        lu = (np.vstack([au, -au]), np.concatenate([bu, -bu])) if au.size else (np.zeros((0, self.nu)), np.zeros(0))
        self.set_linear_constraints(lx[0], lx[1], lu[0], lu[1])

    def set_cache_terms(self, Kinf, Pinf, Quu_inv, AmBKt):
        self._check_setup()
        k, p_, q, a = _f(Kinf), _f(Pinf), _f(Quu_inv), _f(AmBKt)
        assert k.shape == (self.nu, self.nx) and p_.shape == (self.nx, self.nx)
        assert q.shape == (self.nu, self.nu) and a.shape == (self.nx, self.nx)
        _lib.check(self._L.tinympc_set_cache_terms(self._h, _p(k), _p(p_), _p(q), _p(a), 0))

    def set_sensitivity_matrices(self, dK, dP, dC1, dC2):
        self._check_setup()
        dK, dP, dC1, dC2 = _f(dK), _f(dP), _f(dC1), _f(dC2)
        assert dK.shape == (self.nu, self.nx), "dK must be nu x nx"
        assert dP.shape == (self.nx, self.nx), "dP must be nx x nx"
        assert dC1.shape == (self.nu, self.nu), "dC1 must be nu x nu"
        assert dC2.shape == (self.nx, self.nx), "dC2 must be nx x nx"
        _lib.check(self._L.tinympc_set_sensitivity_matrices(self._h, _p(dK), _p(dP), _p(dC1), _p(dC2), 0))

    # ------------------------------------------------------------------ solve / egress
    def solve(self) -> int:
        """Always returns 0, like the reference (TinyMPC.m:146); see get_stats() for the core status."""
        self._check_setup()
        _lib.check(self._L.tinympc_solve(self._h, 0))
        return 0

    def get_solution(self) -> dict:
        self._check_setup()
        x = np.zeros((self.nx, self.N), order="F")
        u = np.zeros((self.nu, self.N - 1), order="F")
        _lib.check(self._L.tinympc_get_solution(self._h, _p(x), _p(u), 0))
        return {"states": x, "controls": u}

    def get_stats(self) -> dict:
        """The MEX verb get_stats (bindings.cpp:264-285; not surfaced by TinyMPC.m) plus all residuals."""
        self._check_setup()
        it, st = C.c_int(), C.c_int()
        ps, pi = C.c_double(), C.c_double()
        _lib.check(self._L.tinympc_get_stats(self._h, C.byref(it), C.byref(st), C.byref(ps), C.byref(pi), 0))
        res = (C.c_double * 4)()
        _lib.check(self._L.tinympc_get_residuals(self._h, res))
        return dict(iter=it.value, status=st.value, primal_residual_state=ps.value, primal_residual_input=pi.value,
                    dual_residual_state=res[1], dual_residual_input=res[3], solved=int(st.value == 1))

    def get_cache(self) -> dict:
        """Device-computed LQR cache (what compute_cache_terms would be compared against)."""
        self._check_setup()
        K = np.zeros((self.nu, self.nx), order="F")
        P = np.zeros((self.nx, self.nx), order="F")
        Qi = np.zeros((self.nu, self.nu), order="F")
        Am = np.zeros((self.nx, self.nx), order="F")
        it = C.c_int()
        _lib.check(self._L.tinympc_get_cache(self._h, _p(K), _p(P), _p(Qi), _p(Am), C.byref(it)))
        return dict(Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am, C1=Qi, C2=Am, riccati_iters=it.value)

    def codegen(self, output_dir, source_dir=None):
        """Write the embedded project's data files from the cache on the device, then -- as the class does after
        the MEX call (TinyMPC.m:159-169, 415-434) -- copy the solver sources next to them and create build/.
        The sources are the `codegen_src` tree of a TinyMPC checkout: `source_dir`, else $TINYMPC_CODEGEN_SRC,
        else a `codegen_src` directory beside this file; without one only the generated files are written."""
        self._check_setup()
        status = self._L.tinympc_codegen(self._h, str(output_dir).encode(), 0)
        if status != 0:
            raise TinyMPCError(status, f"Code generation failed with status: {status}: {_lib.last_error()}")
        self._copy_build_artifacts(output_dir, source_dir)

    def codegen_with_sensitivity(self, output_dir, dK, dP, dC1, dC2, source_dir=None):
        self._check_setup()
        self.set_sensitivity_matrices(dK, dP, dC1, dC2)
        dK, dP, dC1, dC2 = _f(dK), _f(dP), _f(dC1), _f(dC2)
        status = self._L.tinympc_codegen_with_sensitivity(self._h, str(output_dir).encode(), _p(dK), _p(dP), _p(dC1), _p(dC2), 0)
        if status != 0:
            raise TinyMPCError(status, f"Code generation with sensitivity failed with status: {status}: {_lib.last_error()}")
        self._copy_build_artifacts(output_dir, source_dir)

    @staticmethod
    def _copy_build_artifacts(output_dir, source_dir=None):
        import os
        import shutil
        src = source_dir or os.environ.get("TINYMPC_CODEGEN_SRC") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "codegen_src")
        if os.path.isdir(src):
            shutil.copytree(src, str(output_dir), dirs_exist_ok=True)
        os.makedirs(os.path.join(str(output_dir), "build"), exist_ok=True)

    def _cache_buffers(self):
        nx, nu = self.nx, self.nu
        return (np.zeros((nu, nx), order="F"), np.zeros((nx, nx), order="F"),
                np.zeros((nu, nu), order="F"), np.zeros((nx, nx), order="F"))

    def compute_cache_terms(self):
        """[Kinf, Pinf, Quu_inv, AmBKt] of the class's own Riccati loop (TinyMPC.m:194-221: full Q and R, rho
        added once, P0 = Q, 1e-8 regulariser, norm(K - Kprev) < 1e-10, at most 5000 steps), run on the device."""
        self._check_setup()
        K, P, Qi, Am = self._cache_buffers()
        it = C.c_int()
        _lib.check(self._L.tinympc_compute_cache_terms(self._h, _p(K), _p(P), _p(Qi), _p(Am), C.byref(it), 0))
        self.cache_terms_iters = it.value
        return K, P, Qi, Am

    def solve_lqr(self, rho_val: float):
        """[K, P, C1, C2] for Q + rho_val*I, R + rho_val*I (TinyMPC.m:336-366; idare there, a device-side
        fixed-point recursion to stationarity here)."""
        self._check_setup()
        K, P, C1, C2 = self._cache_buffers()
        _lib.check(self._L.tinympc_solve_lqr(self._h, float(rho_val), _p(K), _p(P), _p(C1), _p(C2), None))
        return K, P, C1, C2

    def compute_sensitivity_autograd(self):
        """[dK, dP, dC1, dC2]: forward differences of solve_lqr in rho, h = 1e-6 (TinyMPC.m:223-241)."""
        self._check_setup()
        dK, dP, dC1, dC2 = self._cache_buffers()
        _lib.check(self._L.tinympc_compute_sensitivity(self._h, _p(dK), _p(dP), _p(dC1), _p(dC2), 0))
        return dK, dP, dC1, dC2

    def print_problem_data(self):
        self._check_setup()
        _lib.check(self._L.tinympc_print_problem_data(self._h))

    def reset(self):
        if self.is_setup:
            self._L.tinympc_reset(C.byref(self._h), 0)
            self.is_setup = False

    def __del__(self):
        try:
            if self.is_setup and self._L is not None:
                self._L.tinympc_reset(C.byref(self._h), 0)
                self.is_setup = False
        except Exception:
            pass

    # ------------------------------------------------------------------ batched extensions
    def set_x0_batch(self, x0s, first: int = 0):
        """x0s: nx x count (column b = instance first+b), numpy array or a CUDA torch tensor (count x nx
        contiguous, i.e. the same memory layout)."""
        self._check_setup()
        if hasattr(x0s, "data_ptr") and getattr(x0s, "is_cuda", False):
            assert x0s.dtype.itemsize == 8 and x0s.is_contiguous() and x0s.numel() % self.nx == 0
            count = x0s.numel() // self.nx
            # the C ABI's contract: the producer of the tensor has finished (the handle's stream is ordered against no
            # torch stream); the copy itself completes inside the call
            import torch
            torch.cuda.current_stream(x0s.device).synchronize()
            _lib.check(self._L.tinympc_set_x0_batch_device(self._h, C.c_void_p(x0s.data_ptr()), first, count))
            return
        a = _f(x0s)
        assert a.shape[0] == self.nx
        _lib.check(self._L.tinympc_set_x0_batch(self._h, _p(a), first, a.shape[1]))

    def _tick_buffers(self):
        """Persistent buffers of the per-tick verbs (addresses cached: see _lib.fast_tick_functions)."""
        if self._tick is None or self._tick[0].shape != (self.nx, self.batch):
            xb = np.zeros((self.nx, self.batch), order="F")
            ub = np.zeros((self.nu, self.batch), order="F")
            f_session, f_step = _lib.fast_tick_functions()
            self._tick = (xb, ub, xb.ctypes.data, ub.ctypes.data, f_session, f_step)
        return self._tick

    def mpc_step(self, x0s) -> np.ndarray:
        """One closed-loop tick for every instance: measured states in (nx x batch, or an nx-vector for
        batch 1), warm-started solve, first controls out (nu x batch). One call, one synchronisation."""
        self._check_setup()
        xb, ub, xa, ua, _, f_step = self._tick_buffers()
        a = np.asarray(x0s, dtype=np.float64)
        assert a.size == self.nx * self.batch, f"x0s must be {self.nx} x {self.batch}"
        xb[...] = a.reshape(self.nx, self.batch)
        rc = f_step(self._h.value, xa, ua)
        if rc:
            _lib.check(rc)
        return ub.copy(order="F")

    # ------------------------------------------------------------------ closed-loop session (resident kernel)
    def session_begin(self):
        """Launch the solve kernel once and keep it resident: `session_step` then costs no kernel launch and no stream
        synchronisation per tick (include/tinympc_hip.h). Single-instance handles."""
        self._check_setup()
        _lib.check(self._L.tinympc_session_begin(self._h))

    def session_step(self, x0) -> np.ndarray:
        """One tick inside a session: measured state in, warm-started solve, first controls out."""
        self._check_setup()
        xb, ub, xa, ua, f_session, _ = self._tick_buffers()
        a = np.asarray(x0, dtype=np.float64)
        assert a.size == self.nx
        xb[:, 0] = a.reshape(-1)
        rc = f_session(self._h.value, xa, ua)
        if rc:
            _lib.check(rc)
        return ub[:, 0].copy()

    def bench_closed_loop(self, A, B, x0, ticks: int, skip: int = 0, session=False, fdyn=None) -> dict:
        """`ticks` closed-loop ticks driven from C (libtinympc_bench.so: tinympc_bench_closed_loop, include/tinympc_hip_bench.h): what a
        caller written in C pays per tick -- no Python call inside a tick. With session=True the caller has opened the session; session="verbs": the reference's own three verbs per
        tick (set_x0 + solve + get_solution), launched or -- after set_resident(True) -- resident.
        Returns the per-tick durations (us) of the counted ticks (`tick_us`), their mean / median / maximum, the iterations per tick
        and the final state. A measurement helper, not part of the reference class's surface."""
        self._check_setup()
        if not 0 <= int(skip) < int(ticks):
            raise ValueError(f"bench_closed_loop: need 0 <= skip < ticks (got skip={skip}, ticks={ticks})")
        a, b = _f(A), _f(B)
        x = _f(np.asarray(x0, dtype=np.float64).reshape(-1, 1).copy())
        f = _f(np.asarray(fdyn, dtype=np.float64).reshape(-1, 1)) if fdyn is not None else None
        sec, its = C.c_double(0.0), C.c_long(0)
        per = np.zeros(ticks)
        rc = _lib.load_bench_library().tinympc_bench_closed_loop(self._h, self.nx, self.nu, self.N, _p(a), _p(b), _p(f) if f is not None else None, _p(x), int(ticks),
                                                                 int(skip), 2 if session == "verbs" else int(bool(session)), C.byref(sec), C.byref(its), _p(per))
        _lib.check(rc)
        n = ticks - skip
        return dict(us_per_tick=1e6 * sec.value / n, us_per_tick_median=float(np.median(per[skip:])), us_per_tick_max=float(np.max(per[skip:])),
                    iterations_per_tick=its.value / n, x=x.ravel().copy(), tick_us=per[skip:].copy())

    def debug_setup_timing(self) -> dict:
        """Host microseconds of the phases of this handle's setup (tinympc_debug_setup_timing, include/tinympc_hip_bench.h)."""
        self._check_setup()
        out = np.zeros(10)
        _lib.check(self._L.tinympc_debug_setup_timing(self._h, _p(out)))
        return dict(zip(("prologue_us", "device_arena_us", "pinned_arena_us", "stage_and_queue_us", "queue_precompute_us", "wait_us", "total_us",
                         "riccati_loop_clocks", "riccati_loop_us", "riccati_steps"), out))

    def set_resident(self, on: bool = True):
        """Resident solves (tinympc_set_resident): set_x0 / solve / get_solution served by the resident session kernel instead of a
        launch per solve -- bit-identical results, a 2.5x shorter tick. Off by default (include/tinympc_hip.h says why)."""
        self._check_setup()
        _lib.check(self._L.tinympc_set_resident(self._h, int(bool(on))))

    def session_end(self):
        self._check_setup()
        _lib.check(self._L.tinympc_session_end(self._h))

    def reset_workspace(self):
        self._check_setup()
        _lib.check(self._L.tinympc_reset_workspace(self._h))

    def solve_timed(self) -> float:
        """Synchronous solve; returns the kernel duration in ms (HIP events on the handle's stream)."""
        self._check_setup()
        ms = C.c_float()
        _lib.check(self._L.tinympc_solve_timed(self._h, C.byref(ms)))
        return float(ms.value)

    def solve_queued(self):
        """Queue a timed solve on the handle's stream and return at once (tinympc_solve_queued); collect_kernel_ms() waits and
        returns the kernel durations of everything queued since the last collect."""
        self._check_setup()
        _lib.check(self._L.tinympc_solve_queued(self._h))

    def collect_kernel_ms(self) -> list[float]:
        self._check_setup()
        buf = (C.c_float * 4096)()
        n = C.c_int()
        _lib.check(self._L.tinympc_collect_kernel_ms(self._h, buf, 4096, C.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def solve_async(self):
        self._check_setup()
        _lib.check(self._L.tinympc_solve_async(self._h))

    def synchronize(self):
        self._check_setup()
        _lib.check(self._L.tinympc_synchronize(self._h))

    def get_solution_batch(self, first: int = 0, count: int | None = None) -> dict:
        self._check_setup()
        count = self.batch - first if count is None else count
        x = np.zeros((self.nx, self.N, count), order="F")
        u = np.zeros((self.nu, self.N - 1, count), order="F")
        _lib.check(self._L.tinympc_get_solution_batch(self._h, _p(x), _p(u), first, count))
        return {"states": x, "controls": u}

    def get_first_controls_batch(self, first: int = 0, count: int | None = None) -> np.ndarray:
        self._check_setup()
        count = self.batch - first if count is None else count
        u0 = np.zeros((self.nu, count), order="F")
        _lib.check(self._L.tinympc_get_first_controls_batch(self._h, _p(u0), first, count))
        return u0

    def get_stats_batch(self, first: int = 0, count: int | None = None) -> dict:
        self._check_setup()
        count = self.batch - first if count is None else count
        it = np.zeros(count, dtype=np.int32)
        st = np.zeros(count, dtype=np.int32)
        res = np.zeros((4, count), order="F")
        _lib.check(self._L.tinympc_get_stats_batch(self._h, it.ctypes.data_as(c_int_p), st.ctypes.data_as(c_int_p),
                                                   _p(res), first, count))
        return dict(iter=it, status=st, residuals=res)

    def get_rho_batch(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """Current rho of every instance (adaptive rho adapts it per instance and keeps it across solves)."""
        self._check_setup()
        count = self.batch - first if count is None else count
        rho = np.zeros(count)
        _lib.check(self._L.tinympc_get_rho_batch(self._h, _p(rho), first, count))
        return rho

    def prepare(self):
        """Choose (and, if needed, specialise) the solve kernel for the current constraints / settings now, instead of at the
        first solve that needs it (seconds the first time a shape is seen)."""
        self._check_setup()
        self._push_settings()
        _lib.check(self._L.tinympc_prepare(self._h))

    def launch_info(self) -> dict:
        self._check_setup()
        v = [C.c_int() for _ in range(5)]
        _lib.check(self._L.tinympc_get_launch_info(self._h, *[C.byref(x) for x in v]))
        return dict(lanes_per_instance=v[0].value, instances_per_wave=v[1].value, workgroups=v[2].value,
                    lds_bytes=v[3].value, tables_in_lds=bool(v[4].value), layout=chr(self._L.tinympc_get_layout(self._h)))

    def jit_info(self) -> str:
        """Origin of the kernel of the current configuration: 'compiled-in ...', 'compiled ...', 'disk-cache ...' or
        'refused(<reason>)' (tinympc_get_jit_info)."""
        self._check_setup()
        self._push_settings()
        buf = C.create_string_buffer(1024)
        _lib.check(self._L.tinympc_get_jit_info(self._h, buf, len(buf)))
        return buf.value.decode()

    # ------------------------------------------------------------------ private helpers
    def _check_setup(self):
        if not self.is_setup:
            raise TinyMPCError(_lib.ERR_NOT_INITIALIZED, "Solver not setup. Call setup() first.") from None

    def _push_settings(self):
        s = self.settings
        _lib.check(self._L.tinympc_update_settings(
            self._h, float(s["abs_pri_tol"]), float(s["abs_dua_tol"]), int(s["max_iter"]), int(s["check_termination"]),
            int(bool(s["en_state_bound"])), int(bool(s["en_input_bound"])), int(bool(s["en_state_soc"])),
            int(bool(s["en_input_soc"])), int(bool(s["en_state_linear"])), int(bool(s["en_input_linear"])),
            int(bool(s["adaptive_rho"])), float(s["adaptive_rho_min"]), float(s["adaptive_rho_max"]),
            int(bool(s["adaptive_rho_enable_clipping"])), 0))

    @staticmethod
    def _parse_options(defaults: dict, given: dict) -> dict:
        opts = dict(defaults)
        for k, v in given.items():
            if k in opts:  # unknown keys are silently dropped (TinyMPC.m:372)
                opts[k] = v
        return opts

    @staticmethod
    def _mat(a, cols: int) -> np.ndarray:
        a = np.asarray(a, dtype=np.float64)
        if a.size == 0:
            return _f(np.zeros((0, cols)))
        return _f(a.reshape(-1, cols) if a.ndim == 1 else a)

    @staticmethod
    def _expand_bounds(inp, dim: int, horizon: int, default: float) -> np.ndarray:
        """expand_bounds, TinyMPC.m:378-391."""
        if inp is None or (hasattr(inp, "__len__") and np.size(inp) == 0):
            return np.full((dim, horizon), default)
        a = np.asarray(inp, dtype=np.float64)
        if a.ndim == 0 or a.size == 1:
            return np.full((dim, horizon), float(a.reshape(-1)[0]))
        if a.shape == (dim,) or a.shape == (dim, 1) or a.shape == (1, dim):
            return np.repeat(a.reshape(dim, 1), horizon, axis=1)
        return a  # assume the user provides correct dimensions

    @staticmethod
    def _expand_matrix(ref, dim: int, horizon: int) -> np.ndarray:
        """expand_matrix, TinyMPC.m:393-405."""
        a = np.asarray(ref, dtype=np.float64)
        if a.ndim == 0 or a.size == 1:
            return np.full((dim, horizon), float(a.reshape(-1)[0]))
        if a.shape == (dim,) or a.shape == (dim, 1) or a.shape == (1, dim):
            return np.repeat(a.reshape(dim, 1), horizon, axis=1)
        return a
