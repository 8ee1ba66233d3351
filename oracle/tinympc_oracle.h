/* oracle/tinympc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (no Eigen) CPU restatement of the reference's ADMM hot path. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it -- and only as the
 * checker. The product (libtinympc_hip.so) never links, loads or calls this.
 *
 * Pinned part (checked against the reference's own compiled core, oracle/_ref, and against
 * the fixtures in tests/golden/ generated from it):
 *   box-constrained path == /root/reference/src/codegen_src/tinympc/{admm.cpp,tiny_api.cpp}.
 * PARITY UNPINNED part (no source in the reference tree):
 *   affine dynamics term fdyn, second-order-cone and linear-inequality slack families.
 *   These restate the published upstream TinyMPC/TinyMPC `main` algorithm as reachable from the
 *   call sites in /root/reference/src/bindings.cpp:84-85,408-478.
 *
 * All matrices are FP64, column-major, column = knot point (reference types.hpp:15-17).
 */
#ifndef TINYMPC_ORACLE_H
#define TINYMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_solver {
    int nx, nu, N;

    /* TinyCache (types.hpp:42-56) */
    double rho;
    double *Kinf;    /* nu x nx */
    double *Pinf;    /* nx x nx */
    double *Quu_inv; /* nu x nu */
    double *AmBKt;   /* nx x nx */
    double *APf;     /* nx   (unpinned: affine term) */
    double *BPf;     /* nu   (unpinned: affine term) */
    int riccati_iters;

    /* TinyWorkspace (types.hpp:79-136) */
    double *x, *u, *q, *r, *p, *d, *v, *vnew, *z, *znew, *g, *y;
    double *Q, *R;       /* diagonals, already + rho (tiny_api.cpp:90-91) */
    double *Adyn, *Bdyn; /* nx x nx, nx x nu */
    double *fdyn;        /* nx (unpinned) */
    double *x_min, *x_max, *u_min, *u_max, *Xref, *Uref;
    double primal_residual_state, primal_residual_input;
    double dual_residual_state, dual_residual_input;
    int status, iter;

    /* unpinned: cone + linear slack families */
    int n_cone_x, n_cone_u;
    int *Acx, *qcx, *Acu, *qcu;
    double *cx, *cu;
    double *vc, *vcnew, *gc, *zc, *zcnew, *yc;
    int n_lin_x, n_lin_u;
    double *Alin_x, *blin_x, *Alin_u, *blin_u; /* n_lin_x x nx (col-major), n_lin_x */
    double *vl, *vlnew, *gl, *zl, *zlnew, *yl;

    /* TinySettings (types.hpp:61-74 + bindings.cpp:583-586) */
    double abs_pri_tol, abs_dua_tol;
    int max_iter, check_termination;
    int en_state_bound, en_input_bound;
    int en_state_soc, en_input_soc, en_state_linear, en_input_linear;

    /* adaptive rho (types.hpp:69-73, 50-54; admm.cpp:117-174; rho_benchmark.cpp). Pinned against the
     * reference's own core compiled with zero-initialised automatic variables (oracle/Makefile, target
     * ref_zeroinit): the snapshot reads RhoAdapter::matrices_initialized uninitialised (admm.cpp:118). */
    int adaptive_rho;
    double adaptive_rho_min, adaptive_rho_max;
    int adaptive_rho_enable_clipping;
    double *dKinf_drho, *dPinf_drho; /* nu x nx, nx x nx (dC1/dC2 only touch C1/C2, which no solve phase reads) */

    /* TinySolution (types.hpp:32-37) */
    int sol_iter, solved;
    double *sol_x, *sol_u;
} orc_solver;

/* tiny_setup (tiny_api.cpp:21-122); fdyn may be NULL (zeros). Bounds start at -/+1e17 with the
 * core's default flags (tiny_api_constants.hpp:5-10). Returns NULL on allocation failure. */
orc_solver *orc_setup(const double *A, const double *B, const double *fdyn, const double *Q,
                      const double *R, double rho, int nx, int nu, int N);
void orc_free(orc_solver *s);

/* tiny_precompute_and_set_cache (tiny_api.cpp:124-190); Qd/Rd are the already-augmented
 * diagonals, exactly as tiny_setup passes them (tiny_api.cpp:113). */
int orc_precompute_and_set_cache(orc_solver *s, const double *Qd, const double *Rd);

int orc_set_x0(orc_solver *s, const double *x0);                 /* tiny_api.cpp:233-243 */
int orc_set_x_ref(orc_solver *s, const double *Xref);            /* tiny_api.cpp:245-255 */
int orc_set_u_ref(orc_solver *s, const double *Uref);            /* tiny_api.cpp:257-267 */
int orc_set_bound_constraints(orc_solver *s, const double *x_min, const double *x_max,
                              const double *u_min, const double *u_max); /* bindings.cpp:188-209 */
int orc_set_cache_terms(orc_solver *s, const double *Kinf, const double *Pinf,
                        const double *Quu_inv, const double *AmBKt); /* bindings.cpp:364-405 */
int orc_set_cone_constraints(orc_solver *s, int ncx, const int *Acx, const int *qcx,
                             const double *cx, int ncu, const int *Acu, const int *qcu,
                             const double *cu); /* bindings.cpp:433-478 */
int orc_set_linear_constraints(orc_solver *s, int nlx, const double *Alin_x, const double *blin_x,
                               int nlu, const double *Alin_u, const double *blin_u); /* :408-431 */
void orc_reset_workspace(orc_solver *s); /* zero what tiny_setup zeroes (tiny_api.cpp:73-88) */

/* admm.cpp phase functions */
void orc_forward_pass(orc_solver *s);         /* admm.cpp:25-35  */
void orc_update_slack(orc_solver *s);         /* admm.cpp:43-59  */
void orc_update_dual(orc_solver *s);          /* admm.cpp:65-69  */
void orc_update_linear_cost(orc_solver *s);   /* admm.cpp:75-83  */
int orc_termination_condition(orc_solver *s); /* admm.cpp:89-107 */
void orc_backward_pass_grad(orc_solver *s);   /* admm.cpp:13-20  */
int orc_solve(orc_solver *s);                 /* admm.cpp:109-207 */

/* CPU-baseline helper: `reps` rounds of `count` cold-started solves, one thread.
 * Returns the number of ADMM iterations executed. */
long orc_bench_solves(orc_solver *s, const double *x0s, int count, int reps);

/* Batched oracle: solve `count` instances (x0s nx*count) cold-started, writing
 * sol_x (nx*N*count), sol_u (nu*(N-1)*count), iters (count), residuals (4*count). */
void orc_solve_batch(orc_solver *s, const double *x0s, int count, double *sol_x, double *sol_u,
                     int *iters, int *status, double *residuals);

/* Name-based array access for the Python tests (column-major copies). orc_get returns the
 * element count, -1 for an unknown name, -2 if `capacity` is too small. */
int orc_get(orc_solver *s, const char *name, double *out, int capacity);
int orc_put(orc_solver *s, const char *name, const double *in, int count);
void orc_update_settings(orc_solver *s, double abs_pri_tol, double abs_dua_tol, int max_iter,
                         int check_termination, int en_state_bound, int en_input_bound,
                         int en_state_soc, int en_input_soc, int en_state_linear,
                         int en_input_linear); /* bindings.cpp:548-603 */
void orc_get_stats(orc_solver *s, int *istats, double *dstats);
/* adaptive rho: settings (tiny_api.cpp:225-229 defaults: off, 1.0, 100.0, clip) and sensitivity matrices */
void orc_set_adaptive_rho(orc_solver *s, int enabled, double rho_min, double rho_max, int clip);
int orc_set_sensitivity(orc_solver *s, const double *dKinf_drho, const double *dPinf_drho);
/* benchmark_rho_adaptation (rho_benchmark.cpp:200-249) on the current workspace; returns the new rho */
double orc_rho_adaptation(orc_solver *s);
void orc_set_iter(orc_solver *s, int iter);

#ifdef __cplusplus
}
#endif
#endif
