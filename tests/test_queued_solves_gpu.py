"""tinympc_solve_queued / tinympc_collect_kernel_ms: solves queued on the handle's stream (what bench.py times) give what the same
solves give one by one, and every queued launch reports its own kernel duration."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_queued_solves_equal_synchronous_ones(pkg):
    P = pkg.problems
    prob = P.quadrotor(50)
    B = 2048
    x0s = P.quadrotor_batch_x0(B)
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=25)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    s.set_x0_batch(x0s)
    # three warm-started solves, one by one
    ref = []
    for _ in range(3):
        s.solve()
        ref.append((s.get_solution_batch()["controls"].copy(), s.get_stats_batch()["iter"].copy()))
    # the same three, queued behind a reset, collected once
    s.reset_workspace()
    for _ in range(3):
        s.solve_queued()
    ms = s.collect_kernel_ms()
    assert len(ms) == 3 and all(0.0 < t < 100.0 for t in ms)
    np.testing.assert_array_equal(s.get_solution_batch()["controls"], ref[2][0])
    np.testing.assert_array_equal(s.get_stats_batch()["iter"], ref[2][1])
    assert s.collect_kernel_ms() == []  # nothing queued since
    # cold starts queued back to back (the bench's timed loop): each step returns what a single cold solve returns
    for _ in range(4):
        s.reset_workspace()
        s.solve_queued()
    assert len(s.collect_kernel_ms()) == 4
    np.testing.assert_array_equal(s.get_solution_batch()["controls"], ref[0][0])
    np.testing.assert_array_equal(s.get_stats_batch()["iter"], ref[0][1])
    s.reset()


def test_queued_solves_are_refused_where_they_make_no_sense(pkg):
    P = pkg.problems
    prob = P.quadrotor(50)
    one = pkg.TinyMPC()
    one.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, max_iter=5)
    one.set_x0(prob.x0)
    with pytest.raises(pkg.TinyMPCError):
        one.solve_queued()  # single-instance handles exchange through pinned host memory: synchronous by construction
    one.reset()
