// tinympc_bench_loop.cpp -- libtinympc_bench.so: the closed-loop measurement loop of include/tinympc_hip_bench.h. A CALLER of the
// product library (public verbs of include/tinympc_hip.h only), kept out of it: bench.py's `c_loop` numbers come from here.
#include "tinympc_hip_bench.h"

#include <chrono>
#include <cstddef>
#include <vector>

extern "C" {

int tinympc_bench_closed_loop(tinympc_solver *s, int nx, int nu, int N, const double *A, const double *B, const double *f, double *x, int ticks, int skip, int session,
                              double *seconds, long *iterations, double *tick_us) {
    if (!s || nx < 1 || nu < 1 || (session == 2 && N < 2) || !A || !B || !x || ticks < 1 || skip < 0 || skip >= ticks) return TINYMPC_ERR_INVALID_INPUT;  // (0 <= skip < ticks)
    int rc;
    std::vector<double> u0(nu), xn(nx), sol_x((size_t)nx * (N > 0 ? N : 1)), sol_u((size_t)nu * (N > 1 ? N - 1 : 1));
    double acc = 0.0;
    long its = 0;
    for (int k = 0; k < ticks; ++k) {
        const auto t0 = std::chrono::steady_clock::now();
        if (session == 2) {
            // the reference's own per-tick sequence (examples/cartpole_example_mpc.m:36-44): three verbs, the whole solution copied out
            rc = tinympc_set_x0(s, x, nx, 0);
            if (!rc) rc = tinympc_solve(s, 0);
            if (!rc) rc = tinympc_get_solution(s, sol_x.data(), sol_u.data(), 0);
            for (int q = 0; q < nu; ++q) u0[q] = sol_u[q];
        } else {
            rc = session ? tinympc_session_step(s, x, u0.data()) : tinympc_mpc_step_batch(s, x, u0.data());
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (rc) return rc;
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        if (tick_us) tick_us[k] = us;
        if (k >= skip) {
            acc += 1e-6 * us;
            int it = 0;
            if ((rc = tinympc_get_stats(s, &it, nullptr, nullptr, nullptr, 0))) return rc;  // (a host copy after a tick: outside the timed region)
            its += it;
        }
        for (int i = 0; i < nx; ++i) {
            double v = f ? f[i] : 0.0;
            for (int q = 0; q < nx; ++q) v += A[i + (size_t)q * nx] * x[q];
            for (int q = 0; q < nu; ++q) v += B[i + (size_t)q * nx] * u0[q];
            xn[i] = v;
        }
        for (int i = 0; i < nx; ++i) x[i] = xn[i];
    }
    if (seconds) *seconds = acc;
    if (iterations) *iterations = its;
    return TINYMPC_OK;
}

}  // extern "C"
