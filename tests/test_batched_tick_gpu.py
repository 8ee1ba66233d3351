"""The batched closed-loop tick (examples/cartpole_example_mpc.m:36-44 for B instances at once): tinympc_mpc_step_batch
against the oracle ticking every instance on its own -- the workload of bench.py's `batched_tick` leg (tools/bench_legs.py)."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import rel_err

import pyoracle as O

pytestmark = pytest.mark.gpu


def test_tick_three_of_256_instances_matches_the_oracle(pkg):
    from tools.bench_legs import TICK_SETTINGS
    P = pkg.problems
    prob, B = P.quadrotor(50), 256
    s = pkg.TinyMPC()
    s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **TICK_SETTINGS)
    s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    x = np.asfortranarray(P.quadrotor_batch_x0(B))
    cls = O.OracleRef if O.ref_available() else O.OraclePort
    orcs = [cls(prob).load_problem(prob, TICK_SETTINGS) for _ in range(B)]
    xo = x.copy()
    for k in range(4):  # ticks 0 .. 3
        u = s.mpc_step(x)
        st = s.get_stats_batch()
        uo = np.zeros_like(u)
        for b, orc in enumerate(orcs):
            orc.set_x0(xo[:, b])
            orc.solve()
            uo[:, b] = orc.solution()[1][:, 0]
            assert st["iter"][b] == orc.stats()["iter"], (k, b)
        assert rel_err(u, uo) < 1e-9, k
        x = np.asfortranarray(prob.A @ x + prob.B @ u)
        xo = prob.A @ xo + prob.B @ uo
    assert rel_err(x, xo) < 1e-9
    # the states left on the device: the same tick with x0 resident in HBM gives the same controls
    import torch
    if torch.cuda.is_available():
        t = pkg.TinyMPC()
        t.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **TICK_SETTINGS)
        t.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        x0 = P.quadrotor_batch_x0(B)
        t.set_x0_batch(torch.from_numpy(np.ascontiguousarray(x0.T)).cuda())
        t.solve()
        r = pkg.TinyMPC()
        r.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=B, rho=prob.rho, **TICK_SETTINGS)
        r.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        np.testing.assert_array_equal(t.get_first_controls_batch(), r.mpc_step(np.asfortranarray(x0)))
        t.reset()
        r.reset()
    s.reset()


def test_setup_reset_cycles_reuse_pooled_streams_and_arenas(pkg):
    """Round 5: streams, events and small arenas are pooled per device. Fifty setup / solve / reset cycles of different shapes must give
    the same results as fresh handles did (nothing of a previous tenant may leak into a reused arena), and two live handles never share."""
    P = pkg.problems
    probs = [P.cartpole(20, True), P.quadrotor(20), P.quadrotor(50), P.cartpole(10, True)]
    first = {}
    for k in range(48):
        prob = probs[k % len(probs)]
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=60)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        # a fresh handle: zero solution, zero statistics, zero references (tiny_api.cpp:73-88), whatever the arena held before
        z = s.get_solution()
        assert not z["states"].any() and not z["controls"].any() and s.get_stats()["iter"] == 0
        s.set_x0(prob.x0 * (1.0 + 0.01 * (k % len(probs))))
        s.solve()
        key = k % len(probs)
        res = (s.get_solution()["controls"].copy(), s.get_stats()["iter"], s.get_cache()["Kinf"].copy())
        if key in first:
            np.testing.assert_array_equal(res[0], first[key][0])
            assert res[1] == first[key][1]
            np.testing.assert_array_equal(res[2], first[key][2])
        else:
            first[key] = res
        s.reset()
    a, b = pkg.TinyMPC(), pkg.TinyMPC()
    for h, key in ((a, 0), (b, 2)):
        prob = probs[key]
        h.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, rho=prob.rho, abs_pri_tol=1e-4, abs_dua_tol=1e-4, max_iter=60)
        h.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        h.set_x0(prob.x0 * (1.0 + 0.01 * key))
    a.solve(); b.solve()
    np.testing.assert_array_equal(a.get_solution()["controls"], first[0][0])
    np.testing.assert_array_equal(b.get_solution()["controls"], first[2][0])
    a.reset(); b.reset()
