// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin C-ABI driver around the REFERENCE's own vendored core, compiled from the sources
// where they lie under /root/reference/src/codegen_src (see oracle/Makefile, target `ref`).
// Nothing from the reference is copied into this repository: this file only #includes the
// reference headers at build time and forwards to
//   tiny_setup / tiny_set_x0 / tiny_set_x_ref / tiny_set_u_ref / tiny_solve
//     (reference: src/codegen_src/tinympc/tiny_api.hpp:10-41)
//   forward_pass / update_slack / update_dual / update_linear_cost /
//   termination_condition / backward_pass_grad
//     (reference: src/codegen_src/tinympc/admm.hpp:9-17)
// so that (a) the plain-C restatement in oracle/tinympc_oracle.c can be pinned against the real
// reference, (b) golden fixtures can be generated (tests/golden/gen_golden.py), and (c) bench.py
// can time the real reference CPU path as `cpu_baseline.kind = "reference"`.
//
// The output (oracle/_ref/libtinympc_ref.so) is git-ignored and travels to the GPU box as a
// prebuilt binary only.
#include <vector>
#include <cstring>
#include <ctime>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>

#include "tinympc/codegen.hpp"
#include "tinympc/tiny_api.hpp"
#include "tinympc/types.hpp"

namespace {

tinyMatrix to_mat(const double *p, int rows, int cols) {
    return Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic>>(p, rows, cols);
}

// The reference core chats on std::cout (tiny_api.cpp:138-179, admm.cpp:190); silence it
// unless the caller asked for verbose output.
// Process-wide and reference-counted, so that bench.py's baseline threads (one solver object each)
// can overlap: the first one in swaps the buffer, the last one out restores it.
struct CoutSilencer {
    static std::mutex &mu() { static std::mutex m; return m; }
    static int &depth() { static int d = 0; return d; }
    static std::streambuf *&saved() { static std::streambuf *p = nullptr; return p; }
    static std::ostringstream &sink() { static std::ostringstream s; return s; }
    bool on;
    explicit CoutSilencer(bool quiet) : on(quiet) {
        if (!on) return;
        std::lock_guard<std::mutex> lk(mu());
        if (depth()++ == 0) saved() = std::cout.rdbuf(sink().rdbuf());
    }
    ~CoutSilencer() {
        if (!on) return;
        std::lock_guard<std::mutex> lk(mu());
        if (--depth() == 0) {
            std::cout.rdbuf(saved());
            sink().str(std::string());
        }
    }
};

const tinyMatrix *find_matrix(TinySolver *s, const std::string &n) {
    TinyWorkspace *w = s->work;
    TinyCache *c = s->cache;
    if (n == "x") return &w->x;
    if (n == "u") return &w->u;
    if (n == "q") return &w->q;
    if (n == "r") return &w->r;
    if (n == "p") return &w->p;
    if (n == "d") return &w->d;
    if (n == "v") return &w->v;
    if (n == "vnew") return &w->vnew;
    if (n == "z") return &w->z;
    if (n == "znew") return &w->znew;
    if (n == "g") return &w->g;
    if (n == "y") return &w->y;
    if (n == "Adyn") return &w->Adyn;
    if (n == "Bdyn") return &w->Bdyn;
    if (n == "x_min") return &w->x_min;
    if (n == "x_max") return &w->x_max;
    if (n == "u_min") return &w->u_min;
    if (n == "u_max") return &w->u_max;
    if (n == "Xref") return &w->Xref;
    if (n == "Uref") return &w->Uref;
    if (n == "Kinf") return &c->Kinf;
    if (n == "Pinf") return &c->Pinf;
    if (n == "Quu_inv") return &c->Quu_inv;
    if (n == "AmBKt") return &c->AmBKt;
    if (n == "C1") return &c->C1;
    if (n == "C2") return &c->C2;
    if (n == "dKinf_drho") return &c->dKinf_drho;
    if (n == "dPinf_drho") return &c->dPinf_drho;
    if (n == "dC1_drho") return &c->dC1_drho;
    if (n == "dC2_drho") return &c->dC2_drho;
    if (n == "sol_x") return &s->solution->x;
    if (n == "sol_u") return &s->solution->u;
    return nullptr;
}

}  // namespace

extern "C" {

// tiny_setup of the old snapshot takes the bounds at setup time (tiny_api.cpp:21-25).
void *ref_setup(const double *A, const double *B, const double *Q, const double *R, double rho,
                int nx, int nu, int N, const double *x_min, const double *x_max,
                const double *u_min, const double *u_max, int verbose) {
    CoutSilencer quiet(!verbose);
    TinySolver *solver = nullptr;
    int status = tiny_setup(&solver, to_mat(A, nx, nx), to_mat(B, nx, nu), to_mat(Q, nx, nx),
                            to_mat(R, nu, nu), rho, nx, nu, N, to_mat(x_min, nx, N),
                            to_mat(x_max, nx, N), to_mat(u_min, nu, N - 1),
                            to_mat(u_max, nu, N - 1), verbose);
    if (status != 0) return nullptr;
    return solver;
}

void ref_free(void *h) {
    TinySolver *s = static_cast<TinySolver *>(h);
    if (!s) return;
    delete s->solution;
    delete s->cache;
    delete s->settings;
    delete s->work;
    delete s;
}

int ref_set_x0(void *h, const double *x0) {
    TinySolver *s = static_cast<TinySolver *>(h);
    tinyVector v = Eigen::Map<const Eigen::VectorXd>(x0, s->work->nx);
    return tiny_set_x0(s, v);
}

int ref_set_x_ref(void *h, const double *xr) {
    TinySolver *s = static_cast<TinySolver *>(h);
    return tiny_set_x_ref(s, to_mat(xr, s->work->nx, s->work->N));
}

int ref_set_u_ref(void *h, const double *ur) {
    TinySolver *s = static_cast<TinySolver *>(h);
    return tiny_set_u_ref(s, to_mat(ur, s->work->nu, s->work->N - 1));
}

// What the newer binding's set_bound_constraints verb does (bindings.cpp:188-209): replace the
// four bound arrays; the flags are driven separately through ref_update_settings.
int ref_set_bounds(void *h, const double *x_min, const double *x_max, const double *u_min,
                   const double *u_max) {
    TinySolver *s = static_cast<TinySolver *>(h);
    int nx = s->work->nx, nu = s->work->nu, N = s->work->N;
    s->work->x_min = to_mat(x_min, nx, N);
    s->work->x_max = to_mat(x_max, nx, N);
    s->work->u_min = to_mat(u_min, nu, N - 1);
    s->work->u_max = to_mat(u_max, nu, N - 1);
    return 0;
}

int ref_update_settings(void *h, double abs_pri_tol, double abs_dua_tol, int max_iter,
                        int check_termination, int en_state_bound, int en_input_bound) {
    TinySolver *s = static_cast<TinySolver *>(h);
    return tiny_update_settings(s->settings, abs_pri_tol, abs_dua_tol, max_iter,
                                check_termination, en_state_bound, en_input_bound);
}

// set_cache_terms verb (bindings.cpp:364-405).
int ref_set_cache_terms(void *h, const double *Kinf, const double *Pinf, const double *Quu_inv,
                        const double *AmBKt) {
    TinySolver *s = static_cast<TinySolver *>(h);
    int nx = s->work->nx, nu = s->work->nu;
    s->cache->Kinf = to_mat(Kinf, nu, nx);
    s->cache->Pinf = to_mat(Pinf, nx, nx);
    s->cache->Quu_inv = to_mat(Quu_inv, nu, nu);
    s->cache->AmBKt = to_mat(AmBKt, nx, nx);
    s->cache->C1 = s->cache->Quu_inv;
    s->cache->C2 = s->cache->AmBKt;
    return 0;
}

int ref_solve(void *h, int verbose) {
    CoutSilencer quiet(!verbose);
    return tiny_solve(static_cast<TinySolver *>(h));
}

// The reference's own emitter (codegen.cpp:56-68), so that tests can compare the data file this repository's
// emitter writes with the one the reference writes from the same solver state.
int ref_codegen(void *h, const char *output_dir) {
    CoutSilencer quiet(true);
    return tiny_codegen(static_cast<TinySolver *>(h), output_dir, 0);
}

int ref_set_adaptive_rho(void *h, int enabled, double rho_min, double rho_max, int clip) {
    auto *s = static_cast<TinySolver *>(h);
    s->settings->adaptive_rho = enabled;
    s->settings->adaptive_rho_min = rho_min;
    s->settings->adaptive_rho_max = rho_max;
    s->settings->adaptive_rho_enable_clipping = clip;
    return 0;
}

// Single phase functions, for phase-by-phase pinning of the restatement.
void ref_forward_pass(void *h) { forward_pass(static_cast<TinySolver *>(h)); }
void ref_update_slack(void *h) { update_slack(static_cast<TinySolver *>(h)); }
void ref_update_dual(void *h) { update_dual(static_cast<TinySolver *>(h)); }
void ref_update_linear_cost(void *h) { update_linear_cost(static_cast<TinySolver *>(h)); }
void ref_backward_pass_grad(void *h) { backward_pass_grad(static_cast<TinySolver *>(h)); }
int ref_termination_condition(void *h) {
    TinySolver *s = static_cast<TinySolver *>(h);
    return termination_condition(s) ? 1 : 0;
}
void ref_set_iter(void *h, int iter) { static_cast<TinySolver *>(h)->work->iter = iter; }

// Copy a named matrix out (column-major, as stored). Returns rows*cols, or -1 if unknown.
int ref_get(void *h, const char *name, double *out, int capacity) {
    TinySolver *s = static_cast<TinySolver *>(h);
    std::string n(name);
    if (n == "Q" || n == "R") {
        const tinyVector &v = (n == "Q") ? s->work->Q : s->work->R;
        if ((int)v.size() > capacity) return -2;
        std::memcpy(out, v.data(), sizeof(double) * v.size());
        return (int)v.size();
    }
    const tinyMatrix *m = find_matrix(s, n);
    if (!m) return -1;
    if ((int)m->size() > capacity) return -2;
    std::memcpy(out, m->data(), sizeof(double) * m->size());
    return (int)m->size();
}

// Overwrite a named workspace matrix (used to seed identical states for phase pinning).
int ref_put(void *h, const char *name, const double *in, int count) {
    TinySolver *s = static_cast<TinySolver *>(h);
    tinyMatrix *m = const_cast<tinyMatrix *>(find_matrix(s, std::string(name)));
    if (!m) return -1;
    if ((int)m->size() != count) return -2;
    std::memcpy(m->data(), in, sizeof(double) * count);
    return 0;
}

// iter, status, solved, solution iter, then the four residuals.
void ref_get_stats(void *h, int *istats, double *dstats) {
    TinySolver *s = static_cast<TinySolver *>(h);
    istats[0] = s->work->iter;
    istats[1] = s->work->status;
    istats[2] = s->solution->solved;
    istats[3] = s->solution->iter;
    dstats[0] = s->work->primal_residual_state;
    dstats[1] = s->work->dual_residual_state;
    dstats[2] = s->work->primal_residual_input;
    dstats[3] = s->work->dual_residual_input;
    dstats[4] = s->cache->rho;
}

// CPU-baseline helper for bench.py: run `reps` cold-started solves back to back on one thread
// and return the total ADMM iterations executed. x0s is nx*count (one column per instance).
long ref_bench_solves(void *h, const double *x0s, int count, int reps) {
    TinySolver *s = static_cast<TinySolver *>(h);
    CoutSilencer quiet(true);
    int nx = s->work->nx, nu = s->work->nu, N = s->work->N;
    long iters = 0;
    for (int r = 0; r < reps; ++r) {
        for (int b = 0; b < count; ++b) {
            // cold start: what tiny_setup leaves behind (tiny_api.cpp:73-88)
            s->work->x.setZero();
            s->work->u.setZero();
            s->work->q.setZero();
            s->work->r.setZero();
            s->work->p.setZero();
            s->work->d.setZero();
            s->work->v.setZero();
            s->work->vnew.setZero();
            s->work->z.setZero();
            s->work->znew.setZero();
            s->work->g.setZero();
            s->work->y.setZero();
            tinyVector v = Eigen::Map<const Eigen::VectorXd>(x0s + (size_t)b * nx, nx);
            tiny_set_x0(s, v);
            tiny_solve(s);
            iters += s->work->iter;
        }
    }
    (void)nu;
    (void)N;
    return iters;
}

// CPU-baseline helper for bench.py's closed-loop leg (the reference's own usage, examples/cartpole_example_mpc.m:36-44:
// set_x0 -> solve -> first control -> simulate one step): `ticks` warm-started ticks of x+ = A x + B u0 on one thread,
// the first `skip` of them untimed. Only the three core calls of a tick are inside the timed region (what the MEX verbs
// set_x0 / solve / get_solution reach, bindings.cpp:107-131, 212-261); the plant step is outside it. Returns the total
// ADMM iterations of the timed ticks; *seconds receives the timed seconds, x0 is overwritten with the final state.
// ... tick_us (may be NULL) receives every tick's duration in microseconds, counted or not (the same samples the GPU side's
// tinympc_bench_closed_loop returns: bench.py quotes the same statistic on both sides).
long ref_bench_closed_loop_samples(void *h, double *x0, int ticks, int skip, double *seconds, double *tick_us) {
    TinySolver *s = static_cast<TinySolver *>(h);
    CoutSilencer quiet(true);
    int nx = s->work->nx;
    tinyVector x = Eigen::Map<const Eigen::VectorXd>(x0, nx);
    long iters = 0;
    double acc = 0.0;
    for (int k = 0; k < ticks; ++k) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        tiny_set_x0(s, x);
        tiny_solve(s);
        tinyVector u0 = s->solution->u.col(0);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (tick_us) tick_us[k] = 1e6 * (double)(t1.tv_sec - t0.tv_sec) + 1e-3 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (k >= skip) {
            acc += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            iters += s->work->iter;
        }
        x = s->work->Adyn * x + s->work->Bdyn * u0;
    }
    Eigen::Map<Eigen::VectorXd>(x0, nx) = x;
    if (seconds) *seconds = acc;
    return iters;
}

long ref_bench_closed_loop(void *h, double *x0, int ticks, int skip, double *seconds) {
    return ref_bench_closed_loop_samples(h, x0, ticks, skip, seconds, nullptr);
}

// CPU-baseline helper for bench.py's `batched_tick` leg: `count` independent solvers (one per MPC instance: the reference keeps one
// instance's warm-start state per TinySolver) advanced tick by tick on ONE thread -- for every tick, for every instance: set_x0 ->
// solve -> first control -> plant step x+ = A x + B u0. x0s is nx x count (column = instance), overwritten with the final states;
// tick_us[k] receives the microseconds tick k took for ALL `count` instances (plant steps excluded, as on the GPU side); the first
// `skip` ticks are not counted in the returned iteration total.
long ref_bench_ticks_many(void **hs, int count, double *x0s, int ticks, int skip, double *tick_us) {
    CoutSilencer quiet(true);
    if (count < 1) return 0;
    const int nx = static_cast<TinySolver *>(hs[0])->work->nx;
    long iters = 0;
    std::vector<tinyVector> xs(count), us(count);
    for (int b = 0; b < count; ++b) xs[b] = Eigen::Map<const Eigen::VectorXd>(x0s + (size_t)b * nx, nx);
    for (int k = 0; k < ticks; ++k) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int b = 0; b < count; ++b) {
            TinySolver *s = static_cast<TinySolver *>(hs[b]);
            tiny_set_x0(s, xs[b]);
            tiny_solve(s);
            us[b] = s->solution->u.col(0);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (tick_us) tick_us[k] = 1e6 * (double)(t1.tv_sec - t0.tv_sec) + 1e-3 * (double)(t1.tv_nsec - t0.tv_nsec);
        for (int b = 0; b < count; ++b) {
            TinySolver *s = static_cast<TinySolver *>(hs[b]);
            if (k >= skip) iters += s->work->iter;
            xs[b] = s->work->Adyn * xs[b] + s->work->Bdyn * us[b];
        }
    }
    for (int b = 0; b < count; ++b) Eigen::Map<Eigen::VectorXd>(x0s + (size_t)b * nx, nx) = xs[b];
    return iters;
}

// CPU-baseline helper for bench.py's `setup` leg: tiny_setup (tiny_api.cpp:21-122, precompute included) + the teardown the MEX's
// reset does, `reps` times on one thread; us_out[r] receives the microseconds of the r-th tiny_setup alone. Returns the number of
// successful setups.
int ref_bench_setup(const double *A, const double *B, const double *Q, const double *R, double rho, int nx, int nu, int N,
                    const double *x_min, const double *x_max, const double *u_min, const double *u_max, int reps, double *us_out) {
    int ok = 0;
    for (int r = 0; r < reps; ++r) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        void *h = ref_setup(A, B, Q, R, rho, nx, nu, N, x_min, x_max, u_min, u_max, 0);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (us_out) us_out[r] = 1e6 * (double)(t1.tv_sec - t0.tv_sec) + 1e-3 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (h) ok += 1;
        ref_free(h);
    }
    return ok;
}

}  // extern "C"
