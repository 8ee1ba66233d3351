"""CPU restatement of the Riccati helpers the reference implements in MATLAB inside its class
-- TEST INFRASTRUCTURE ONLY (imported by tests/ only; the product runs these on the device).

Follows /root/reference/src/TinyMPC.m:
  compute_cache_terms            :194-221
  compute_sensitivity_autograd   :223-241
  solve_lqr                      :336-366   (MATLAB's idare -> scipy.linalg.solve_discrete_are, the same DARE)

PARITY: MATLAB is not available in this image, so no output of the reference's own run pins these; they are
pinned instead by (i) the DARE residual of the result being ~0, (ii) scipy's independent DARE solver, and
(iii) agreement between the two branches of solve_lqr (closed-form vs the class's iterative fallback).
"""
from __future__ import annotations

import numpy as np


def compute_cache_terms(A, B, Q, R, rho):
    """TinyMPC.m:194-221. Returns Kinf, Pinf, Quu_inv, AmBKt, iterations."""
    A, B, Q, R = (np.asarray(m, dtype=np.float64) for m in (A, B, Q, R))
    nx, nu = B.shape
    Q_rho = Q + rho * np.eye(nx)                                            # :199
    R_rho = R + rho * np.eye(nu)                                            # :200
    Kinf = np.zeros((nu, nx))                                               # :203
    Pinf = Q.copy()                                                         # :204
    it = 0
    for it in range(1, 5001):                                               # :207
        Kinf_prev = Kinf
        Kinf = np.linalg.solve(R_rho + B.T @ Pinf @ B + 1e-8 * np.eye(nu), B.T @ Pinf @ A)   # :209
        Pinf = Q_rho + A.T @ Pinf @ (A - B @ Kinf)                          # :210
        if np.linalg.norm(Kinf - Kinf_prev, 2) < 1e-10:                     # :211  (MATLAB norm = spectral)
            break
    AmBKt = (A - B @ Kinf).T                                                # :216
    Quu_inv = np.linalg.inv(R_rho + B.T @ Pinf @ B)                         # :217
    return Kinf, Pinf, Quu_inv, AmBKt, it


def solve_lqr(A, B, Q, R, rho_val, iterative=False):
    """TinyMPC.m:336-366. Returns K, P, C1, C2."""
    A, B, Q, R = (np.asarray(m, dtype=np.float64) for m in (A, B, Q, R))
    nx, nu = B.shape
    Q_rho = Q + rho_val * np.eye(nx)                                        # :341
    R_rho = R + rho_val * np.eye(nu)                                        # :342
    if not iterative:
        from scipy.linalg import solve_discrete_are
        P = solve_discrete_are(A, B, Q_rho, R_rho)                          # :347  idare
        # Sign convention: u = -K x, i.e. the K of the class's iterative branch (:354), of compute_cache_terms
        # and of the solver core. (The reference negates idare's K at :348; idare already returns this K, so
        # on a MATLAB with the Control System Toolbox its K -- and with it dK and C2 -- come out with the
        # opposite sign to its own fallback branch. The build follows the fallback branch.)
        K = np.linalg.solve(R_rho + B.T @ P @ B, B.T @ P @ A)
    else:
        P = Q_rho.copy()                                                    # :351
        K = np.zeros((nu, nx))
        for it in range(1, 5001):                                           # :353
            K_prev = K
            K = np.linalg.solve(R_rho + B.T @ P @ B + 1e-8 * np.eye(nu), B.T @ P @ A)
            P = Q_rho + A.T @ P @ (A - B @ K)
            if it > 1 and np.linalg.norm(K - K_prev, 2) < 1e-10:            # :356
                break
    C1 = np.linalg.inv(R_rho + B.T @ P @ B)                                 # :363
    C2 = (A - B @ K).T                                                      # :364
    return K, P, C1, C2


def compute_sensitivity(A, B, Q, R, rho, h=1e-6):
    """TinyMPC.m:223-241: forward differences of solve_lqr."""
    K0, P0, C10, C20 = solve_lqr(A, B, Q, R, rho)
    K1, P1, C11, C21 = solve_lqr(A, B, Q, R, rho + h)
    return (K1 - K0) / h, (P1 - P0) / h, (C11 - C10) / h, (C21 - C20) / h


def dare_residual(A, B, Q_rho, R_rho, P):
    """|| P - (Q + A'PA - A'PB (R + B'PB)^-1 B'PA) ||_max / ||P||_max"""
    G = np.linalg.solve(R_rho + B.T @ P @ B, B.T @ P @ A)
    return np.abs(P - (Q_rho + A.T @ P @ A - A.T @ P @ B @ G)).max() / np.abs(P).max()
