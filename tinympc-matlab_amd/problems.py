"""Problem data for the BASELINE.json configurations (plain numbers, no algorithm).

Sources (reference files, read for their numeric values only):
  cartpole   examples/cartpole_example_one_solve.m:13-21, bounds examples/cartpole_example_code_generation.m:23-33
  quadrotor  examples/quadrotor_hover_code_generation.m:17-49 (same in tests/test_quadrotor_codegen.m:9-43)
  rocket     examples/rocket_landing_constraints.m:17-47, refs :72-75
SURVEY.md section 8(d) fixes the horizon lengths, bounds, x0 and the batch seed used here.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

BOUND_INF = 1e17  # TinyMPC.m:261-264 fill value for missing bounds


@dataclass
class Problem:
    name: str
    A: np.ndarray
    B: np.ndarray
    Q: np.ndarray
    R: np.ndarray
    N: int
    rho: float
    x0: np.ndarray
    x_min: np.ndarray | None = None  # nx-vector or None
    x_max: np.ndarray | None = None
    u_min: np.ndarray | None = None
    u_max: np.ndarray | None = None
    fdyn: np.ndarray | None = None
    x_ref: np.ndarray | None = None  # nx x N or None (zeros)
    u_ref: np.ndarray | None = None  # nu x (N-1) or None (zeros)
    cones: dict = field(default_factory=dict)   # Acx,qcx,cx,Acu,qcu,cu
    linear: dict = field(default_factory=dict)  # Alin_x, blin_x, Alin_u, blin_u

    @property
    def nx(self) -> int:
        return self.A.shape[0]

    @property
    def nu(self) -> int:
        return self.B.shape[1]

    def has_bounds(self) -> bool:
        return any(b is not None for b in (self.x_min, self.x_max, self.u_min, self.u_max))

    def expanded_bounds(self):
        """What TinyMPC.m:256-264 / expand_bounds sends through the boundary."""
        nx, nu, N = self.nx, self.nu, self.N

        def ex(v, dim, hor, default):
            if v is None:
                return np.full((dim, hor), default, dtype=np.float64)
            v = np.asarray(v, dtype=np.float64)
            if v.ndim == 0:
                return np.full((dim, hor), float(v))
            if v.ndim == 1:
                return np.repeat(v.reshape(dim, 1), hor, axis=1)
            return v.copy()

        return (ex(self.x_min, nx, N, -BOUND_INF), ex(self.x_max, nx, N, BOUND_INF),
                ex(self.u_min, nu, N - 1, -BOUND_INF), ex(self.u_max, nu, N - 1, BOUND_INF))

    def bytes_per_iteration(self) -> int:
        """Algorithmic HBM bytes per instance-iteration, SURVEY.md section 8(d): 8*(11U + 9X)."""
        X = self.nx * self.N
        U = self.nu * (self.N - 1)
        return 8 * (11 * U + 9 * X)

    def flops_per_iteration(self) -> int:
        """SURVEY.md section 8(a) general flop model (box-only)."""
        nx, nu, N = self.nx, self.nu, self.N
        X, U = nx * N, nu * (N - 1)
        fwd = 2 * nu * nx + nu + 2 * nx * nx + 2 * nx * nu
        bwd = 2 * nx * nu + nu + 2 * nu * nu + 2 * nx * nx + 2 * nx * nu + 2 * nx
        return (N - 1) * (fwd + bwd) + 13 * (X + U) + 2 * nx * nx + 3 * nx

    def flops_per_iteration_families(self) -> dict:
        """Flop model of what the cone / linear-inequality families and `fdyn` ADD to an ADMM iteration, counted on the
        published upstream algorithm (the semantics behind bindings.cpp:84-85, 408-478; the repo's CPU restatement has the loops) the way
        SURVEY.md section 8(a) counts the box path: every add, subtract, multiply, compare, divide and square root is one flop,
        the projections at their expensive branch (a cone point outside the cone, a violated half-space). Per enabled family and
        side, with E = nx N (state side) or nu (N-1) (input side) elements over K = N or N-1 knots:
          slack copy  s = x + dual                       E
          dual ascent dual += x - s                      2 E
          linear cost q -= rho (s - dual)                3 E   (+ 3 nx for the terminal p on the state side)
          cone of dimension n per knot (project_soc)     u0 = mu t: 1; |w|^2: 2(n-1); sqrt: 1; two compares: 2;
                                                         scale = 0.5 (1 + u0 / a): 3; w *= scale: n-1; t = scale (a / mu): 2
                                                         = 3 n + 6
          linear row per knot (project_halfspaces)       a's: 2 dim; |a|^2: 2 dim; compare: 1; (a's - b) / |a|^2: 2;
                                                         s -= dist a: 2 dim   = 6 dim + 3
          fdyn                                            x+ += f: nx per forward step; p += APf, (.. + BPf): nx + nu per backward step
        Returns the parts and their sum (`total`), and `with_box` = total + flops_per_iteration()."""
        nx, nu, N = self.nx, self.nu, self.N
        X, U = nx * N, nu * (N - 1)
        parts = {"state_cones": 0, "input_cones": 0, "state_linear": 0, "input_linear": 0, "fdyn": 0}
        c = self.cones or {}
        if len(c.get("qcx", [])):
            parts["state_cones"] = 6 * X + 3 * nx + N * sum(3 * int(n) + 6 for n in c["qcx"])
        if len(c.get("qcu", [])):
            parts["input_cones"] = 6 * U + (N - 1) * sum(3 * int(n) + 6 for n in c["qcu"])
        lin = self.linear or {}
        mx = int(np.asarray(lin.get("blin_x", [])).size)
        mu_rows = int(np.asarray(lin.get("blin_u", [])).size)
        if mx:
            parts["state_linear"] = 6 * X + 3 * nx + N * mx * (6 * nx + 3)
        if mu_rows:
            parts["input_linear"] = 6 * U + (N - 1) * mu_rows * (6 * nu + 3)
        if self.fdyn is not None and np.any(np.asarray(self.fdyn) != 0):
            parts["fdyn"] = (N - 1) * (2 * nx + nu)
        total = sum(parts.values())
        return dict(parts, total=total, with_box=total + self.flops_per_iteration())


def cartpole(N: int = 20, bounded: bool = True) -> Problem:
    A = np.array([[1.0, 0.01, 0.0, 0.0],
                  [0.0, 1.0, 0.039, 0.0],
                  [0.0, 0.0, 1.002, 0.01],
                  [0.0, 0.0, 0.458, 1.002]])
    B = np.array([[0.0], [0.02], [0.0], [0.067]])
    Q = np.diag([10.0, 1.0, 10.0, 1.0])
    R = np.diag([1.0])
    p = Problem("cartpole", A, B, Q, R, N, 1.0, np.array([0.5, 0.0, 0.0, 0.0]))
    if bounded:
        p.u_min = np.array([-0.5])
        p.u_max = np.array([0.5])
    return p


_QUAD_A = np.array([
    [1.0, 0.0, 0.0, 0.0, 0.0245250, 0.0, 0.05, 0.0, 0.0, 0.0, 0.0002044, 0.0],
    [0.0, 1.0, 0.0, -0.0245250, 0.0, 0.0, 0.0, 0.05, 0.0, -0.0002044, 0.0, 0.0],
    [0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.05, 0.0, 0.0, 0.0],
    [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.025, 0.0, 0.0],
    [0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.025, 0.0],
    [0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.025],
    [0.0, 0.0, 0.0, 0.0, 0.981, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0122625, 0.0],
    [0.0, 0.0, 0.0, -0.981, 0.0, 0.0, 0.0, 1.0, 0.0, -0.0122625, 0.0, 0.0],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0],
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])

_QUAD_B = np.array([
    [-0.0007069, 0.0007773, 0.0007091, -0.0007795],
    [0.0007034, 0.0007747, -0.0007042, -0.0007739],
    [0.0052554, 0.0052554, 0.0052554, 0.0052554],
    [-0.1720966, -0.1895213, 0.1722891, 0.1893288],
    [-0.1729419, 0.1901740, 0.1734809, -0.1907131],
    [0.0123423, -0.0045148, -0.0174024, 0.0095748],
    [-0.0565520, 0.0621869, 0.0567283, -0.0623632],
    [0.0562756, 0.0619735, -0.0563386, -0.0619105],
    [0.2102143, 0.2102143, 0.2102143, 0.2102143],
    [-13.7677303, -15.1617018, 13.7831318, 15.1463003],
    [-13.8353509, 15.2139209, 13.8784751, -15.2570451],
    [0.9873856, -0.3611820, -1.3921880, 0.7659845]])

_QUAD_QDIAG = np.array([100.0, 100.0, 100.0, 4.0, 4.0, 400.0, 4.0, 4.0, 4.0, 2.0408163, 2.0408163, 4.0])
QUAD_X0 = np.array([0.5, -0.4, 0.3, 0.05, -0.05, 0.1, 0.2, -0.2, 0.1, 0.1, -0.1, 0.05])
QUAD_X0_SCALE = np.array([1, 1, 1, .2, .2, .2, .5, .5, .5, .5, .5, .5], dtype=np.float64)
BATCH_SEED = 20250905


def quadrotor(N: int = 50) -> Problem:
    """Config 3: hover, rho=5, x in [-5,5], u in [-0.5,0.5]."""
    p = Problem("quadrotor", _QUAD_A.copy(), _QUAD_B.copy(), np.diag(_QUAD_QDIAG), np.diag([4.0] * 4),
                N, 5.0, QUAD_X0.copy())
    p.x_min = np.full(12, -5.0)
    p.x_max = np.full(12, 5.0)
    p.u_min = np.full(4, -0.5)
    p.u_max = np.full(4, 0.5)
    return p


def quadrotor_batch_x0(count: int, offset: int = 0) -> np.ndarray:
    """Config 5 initial states: x0[b] = s * xi_b, xi_b ~ U(-1,1)^12, rng(20250905).
    Returns a (12, count) column-major-friendly array for instances [offset, offset+count).
    The stream is consumed instance by instance so any shard reproduces the same numbers."""
    rng = np.random.default_rng(BATCH_SEED)
    xi = rng.uniform(-1.0, 1.0, size=(offset + count, 12))[offset:]
    return np.asfortranarray((xi * QUAD_X0_SCALE).T)


def rocket(N: int = 100, with_linear: bool = True) -> Problem:
    """Config 4: rocket landing with SOC thrust/glide cones (+ synthetic ground-plane half-space)."""
    A = np.eye(6)
    A[0, 3] = A[1, 4] = A[2, 5] = 0.05
    B = np.zeros((6, 3))
    B[0, 0] = B[1, 1] = B[2, 2] = 0.000125
    B[3, 0] = B[4, 1] = B[5, 2] = 0.005
    fdyn = np.array([0.0, 0.0, -0.0122625, 0.0, 0.0, -0.4905])
    xinit = np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5])
    xgoal = np.zeros(6)
    NTOTAL = 100
    p = Problem("rocket", A, B, np.diag([101.0] * 6), np.diag([2.0] * 3), N, 1.0, xinit * 1.1)
    p.fdyn = fdyn
    p.x_min = np.array([-5.0, -5.0, -0.5, -10.0, -10.0, -20.0])
    p.x_max = np.array([5.0, 5.0, 100.0, 10.0, 10.0, 20.0])
    p.u_min = np.array([-10.0, -10.0, -10.0])
    p.u_max = np.array([105.0, 105.0, 105.0])
    x_ref = np.zeros((6, N))
    for i in range(N):
        x_ref[:, i] = xinit + (xgoal - xinit) * min(i, NTOTAL - 1) / (NTOTAL - 1)
    u_ref = np.zeros((3, N - 1))
    u_ref[2, :] = 10.0
    p.x_ref, p.u_ref = x_ref, u_ref
    p.cones = dict(Acx=[0], qcx=[3], cx=[0.5], Acu=[0], qcu=[3], cu=[0.25])
    if with_linear:
        p.linear = dict(Alin_x=np.array([[0.0, 0.0, -1.0, 0.0, 0.0, 0.0]]), blin_x=np.array([0.0]),
                        Alin_u=np.zeros((0, 3)), blin_u=np.zeros(0))
    return p
