/* oracle/tinympc_oracle.c -- TEST INFRASTRUCTURE ONLY (see tinympc_oracle.h).
 *
 * Function-by-function plain-C restatement of the reference ADMM path. Every function cites the
 * reference lines it follows; paths are relative to /root/reference/src/codegen_src/tinympc/.
 * Association order of every product/sum follows the reference's Eigen expressions so that the
 * restatement agrees with the compiled reference to ~1e-15 relative (tests/test_oracle_vs_ref.py).
 *
 * PARITY UNPINNED: everything guarded by fdyn / *_soc / *_linear restates the upstream
 * TinyMPC/TinyMPC `main` algorithm, which is not present in the reference tree.
 */
#include "tinympc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_BOUND_DEFAULT 1e17 /* TinyMPC.m:261-264 */

static double *zalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

/* C(m x n) = A(m x k) * B(k x n), column-major, plain triple loop */
static void matmul(double *C, const double *A, const double *B, int m, int k, int n) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
            double acc = 0.0;
            for (int l = 0; l < k; ++l) acc += A[i + (size_t)l * m] * B[l + (size_t)j * k];
            C[i + (size_t)j * m] = acc;
        }
}

static void transpose(double *T, const double *A, int m, int n) { /* T = A^T, A is m x n */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) T[j + (size_t)i * n] = A[i + (size_t)j * m];
}

/* Inverse by partial-pivot LU + solve against the identity: what Eigen's dynamic-size
 * MatrixBase::inverse() does (PartialPivLU), used at tiny_api.cpp:154 and :169. */
static int lu_inverse(double *inv, const double *M, int n) {
    double *lu = (double *)malloc(sizeof(double) * n * n);
    int *perm = (int *)malloc(sizeof(int) * n);
    if (!lu || !perm) return 1;
    memcpy(lu, M, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(lu[k + (size_t)k * n]);
        for (int i = k + 1; i < n; ++i) {
            double a = fabs(lu[i + (size_t)k * n]);
            if (a > best) {
                best = a;
                piv = i;
            }
        }
        if (piv != k) {
            for (int j = 0; j < n; ++j) {
                double t = lu[k + (size_t)j * n];
                lu[k + (size_t)j * n] = lu[piv + (size_t)j * n];
                lu[piv + (size_t)j * n] = t;
            }
            int t = perm[k];
            perm[k] = perm[piv];
            perm[piv] = t;
        }
        double pivot = lu[k + (size_t)k * n];
        for (int i = k + 1; i < n; ++i) lu[i + (size_t)k * n] /= pivot;
        for (int j = k + 1; j < n; ++j) {
            double ukj = lu[k + (size_t)j * n];
            for (int i = k + 1; i < n; ++i) lu[i + (size_t)j * n] -= lu[i + (size_t)k * n] * ukj;
        }
    }
    for (int c = 0; c < n; ++c) {
        double *col = inv + (size_t)c * n;
        for (int i = 0; i < n; ++i) col[i] = (perm[i] == c) ? 1.0 : 0.0; /* P * e_c */
        for (int i = 0; i < n; ++i) /* unit-lower forward substitution */
            for (int j = 0; j < i; ++j) col[i] -= lu[i + (size_t)j * n] * col[j];
        for (int i = n - 1; i >= 0; --i) { /* upper back substitution */
            for (int j = i + 1; j < n; ++j) col[i] -= lu[i + (size_t)j * n] * col[j];
            col[i] /= lu[i + (size_t)i * n];
        }
    }
    free(lu);
    free(perm);
    return 0;
}

/* ---------------------------------------------------------------- precompute (P1) */
/* tiny_api.cpp:124-190. Note the two parity traps: rho is added a second
 * time here (:134-135) on top of tiny_setup's (:90-91), and the recursion stops at
 * max|K - Kprev| < 1e-5 (:157) keeping that iteration's K and P. */
int orc_precompute_and_set_cache(orc_solver *s, const double *Qd, const double *Rd) {
    const int nx = s->nx, nu = s->nu;
    const double rho = s->rho;
    const double *A = s->Adyn, *B = s->Bdyn;
    double *Q1 = zalloc((size_t)nx * nx), *R1 = zalloc((size_t)nu * nu);
    double *Ktp1 = zalloc((size_t)nu * nx), *Ptp1 = zalloc((size_t)nx * nx);
    double *Kinf = zalloc((size_t)nu * nx), *Pinf = zalloc((size_t)nx * nx);
    double *Bt = zalloc((size_t)nu * nx), *At = zalloc((size_t)nx * nx);
    double *BtP = zalloc((size_t)nu * nx), *S = zalloc((size_t)nu * nu), *Sinv = zalloc((size_t)nu * nu);
    double *T1 = zalloc((size_t)nu * nx), *T2 = zalloc((size_t)nu * nx);
    double *AtP = zalloc((size_t)nx * nx), *BK = zalloc((size_t)nx * nx), *AmBK = zalloc((size_t)nx * nx);
    double *T3 = zalloc((size_t)nx * nx);
    for (int i = 0; i < nx; ++i) Q1[i + (size_t)i * nx] = Qd[i] + rho; /* :134 */
    for (int i = 0; i < nu; ++i) R1[i + (size_t)i * nu] = Rd[i] + rho; /* :135 */
    for (int i = 0; i < nx; ++i) Ptp1[i + (size_t)i * nx] = rho;       /* :148 */
    transpose(Bt, B, nx, nu);
    transpose(At, A, nx, nx);
    s->riccati_iters = 1000;
    for (int it = 0; it < 1000; ++it) { /* :152 */
        /* :154  Kinf = (R1 + B'*P*B).inverse() * B' * P * A   (left to right) */
        matmul(BtP, Bt, Ptp1, nu, nx, nx);
        matmul(S, BtP, B, nu, nx, nu);
        for (int i = 0; i < nu * nu; ++i) S[i] = R1[i] + S[i];
        lu_inverse(Sinv, S, nu);
        matmul(T1, Sinv, Bt, nu, nu, nx);
        matmul(T2, T1, Ptp1, nu, nx, nx);
        matmul(Kinf, T2, A, nu, nx, nx);
        /* :155  Pinf = Q1 + A'*P*(A - B*Kinf) */
        matmul(AtP, At, Ptp1, nx, nx, nx);
        matmul(BK, B, Kinf, nx, nu, nx);
        for (int i = 0; i < nx * nx; ++i) AmBK[i] = A[i] - BK[i];
        matmul(T3, AtP, AmBK, nx, nx, nx);
        for (int i = 0; i < nx * nx; ++i) Pinf[i] = Q1[i] + T3[i];
        /* :157 */
        double md = 0.0;
        for (int i = 0; i < nu * nx; ++i) {
            double a = fabs(Kinf[i] - Ktp1[i]);
            if (a > md) md = a;
        }
        if (md < 1e-5) {
            s->riccati_iters = it + 1;
            break;
        }
        memcpy(Ktp1, Kinf, sizeof(double) * nu * nx); /* :164-165 */
        memcpy(Ptp1, Pinf, sizeof(double) * nx * nx);
    }
    /* :169  Quu_inv = (R1 + B'*Pinf*B).inverse() */
    matmul(BtP, Bt, Pinf, nu, nx, nx);
    matmul(S, BtP, B, nu, nx, nu);
    for (int i = 0; i < nu * nu; ++i) S[i] = R1[i] + S[i];
    lu_inverse(s->Quu_inv, S, nu);
    /* :170  AmBKt = (A - B*Kinf)' */
    matmul(BK, B, Kinf, nx, nu, nx);
    for (int i = 0; i < nx * nx; ++i) AmBK[i] = A[i] - BK[i];
    transpose(s->AmBKt, AmBK, nx, nx);
    memcpy(s->Kinf, Kinf, sizeof(double) * nu * nx);
    memcpy(s->Pinf, Pinf, sizeof(double) * nx * nx);
    /* UNPINNED (upstream main): APf = AmBKt*Pinf*fdyn, BPf = B'*Pinf*fdyn */
    {
        double *Pf = zalloc(nx);
        matmul(Pf, s->Pinf, s->fdyn, nx, nx, 1);
        matmul(s->APf, s->AmBKt, Pf, nx, nx, 1);
        matmul(s->BPf, Bt, Pf, nu, nx, 1);
        free(Pf);
    }
    free(Q1); free(R1); free(Ktp1); free(Ptp1); free(Kinf); free(Pinf); free(Bt); free(At);
    free(BtP); free(S); free(Sinv); free(T1); free(T2); free(AtP); free(BK); free(AmBK); free(T3);
    return 0;
}

/* ---------------------------------------------------------------- setup / setters */
static void fill(double *a, size_t n, double v) {
    for (size_t i = 0; i < n; ++i) a[i] = v;
}

/* tiny_api.cpp:21-122 (old snapshot) with the newer binding's surface (bindings.cpp:47-104):
 * bounds are not setup arguments; they start "infinite" and are replaced by
 * orc_set_bound_constraints. Only the diagonals of Q and R are kept (:90-91). */
orc_solver *orc_setup(const double *A, const double *B, const double *fdyn, const double *Q,
                      const double *R, double rho, int nx, int nu, int N) {
    orc_solver *s = (orc_solver *)calloc(1, sizeof(orc_solver));
    if (!s) return NULL;
    const size_t X = (size_t)nx * N, U = (size_t)nu * (N - 1);
    s->nx = nx; s->nu = nu; s->N = N; s->rho = rho;
    s->Kinf = zalloc((size_t)nu * nx); s->Pinf = zalloc((size_t)nx * nx);
    s->Quu_inv = zalloc((size_t)nu * nu); s->AmBKt = zalloc((size_t)nx * nx);
    s->APf = zalloc(nx); s->BPf = zalloc(nu);
    s->x = zalloc(X); s->q = zalloc(X); s->p = zalloc(X); s->v = zalloc(X); s->vnew = zalloc(X); s->g = zalloc(X);
    s->u = zalloc(U); s->r = zalloc(U); s->d = zalloc(U); s->z = zalloc(U); s->znew = zalloc(U); s->y = zalloc(U);
    s->Q = zalloc(nx); s->R = zalloc(nu);
    s->Adyn = zalloc((size_t)nx * nx); s->Bdyn = zalloc((size_t)nx * nu); s->fdyn = zalloc(nx);
    s->x_min = zalloc(X); s->x_max = zalloc(X); s->u_min = zalloc(U); s->u_max = zalloc(U);
    s->Xref = zalloc(X); s->Uref = zalloc(U);
    s->vc = zalloc(X); s->vcnew = zalloc(X); s->gc = zalloc(X);
    s->zc = zalloc(U); s->zcnew = zalloc(U); s->yc = zalloc(U);
    s->vl = zalloc(X); s->vlnew = zalloc(X); s->gl = zalloc(X);
    s->zl = zalloc(U); s->zlnew = zalloc(U); s->yl = zalloc(U);
    s->sol_x = zalloc(X); s->sol_u = zalloc(U);
    s->dKinf_drho = zalloc((size_t)nu * nx); s->dPinf_drho = zalloc((size_t)nx * nx);
    s->adaptive_rho = 0; s->adaptive_rho_min = 1.0; s->adaptive_rho_max = 100.0; /* tiny_api.cpp:226-229 */
    s->adaptive_rho_enable_clipping = 1;
    memcpy(s->Adyn, A, sizeof(double) * nx * nx);
    memcpy(s->Bdyn, B, sizeof(double) * nx * nu);
    if (fdyn) memcpy(s->fdyn, fdyn, sizeof(double) * nx);
    for (int i = 0; i < nx; ++i) s->Q[i] = Q[i + (size_t)i * nx] + rho; /* :90 */
    for (int i = 0; i < nu; ++i) s->R[i] = R[i + (size_t)i * nu] + rho; /* :91 */
    fill(s->x_min, X, -ORC_BOUND_DEFAULT); fill(s->x_max, X, ORC_BOUND_DEFAULT);
    fill(s->u_min, U, -ORC_BOUND_DEFAULT); fill(s->u_max, U, ORC_BOUND_DEFAULT);
    /* tiny_set_default_settings (tiny_api.cpp:213-231, tiny_api_constants.hpp:5-10) */
    s->abs_pri_tol = 1e-3; s->abs_dua_tol = 1e-3; s->max_iter = 1000; s->check_termination = 1;
    s->en_state_bound = 1; s->en_input_bound = 1;
    orc_precompute_and_set_cache(s, s->Q, s->R); /* :113 */
    return s;
}

void orc_free(orc_solver *s) {
    if (!s) return;
    double *ptrs[] = {s->Kinf, s->Pinf, s->Quu_inv, s->AmBKt, s->APf, s->BPf, s->x, s->u, s->q, s->r,
                      s->p, s->d, s->v, s->vnew, s->z, s->znew, s->g, s->y, s->Q, s->R, s->Adyn,
                      s->Bdyn, s->fdyn, s->x_min, s->x_max, s->u_min, s->u_max, s->Xref, s->Uref,
                      s->cx, s->cu, s->vc, s->vcnew, s->gc, s->zc, s->zcnew, s->yc, s->Alin_x,
                      s->blin_x, s->Alin_u, s->blin_u, s->vl, s->vlnew, s->gl, s->zl, s->zlnew,
                      s->yl, s->sol_x, s->sol_u, s->dKinf_drho, s->dPinf_drho};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) free(ptrs[i]);
    free(s->Acx); free(s->qcx); free(s->Acu); free(s->qcu);
    free(s);
}

void orc_reset_workspace(orc_solver *s) { /* tiny_api.cpp:73-88 */
    const size_t X = (size_t)s->nx * s->N, U = (size_t)s->nu * (s->N - 1);
    double *xs[] = {s->x, s->q, s->p, s->v, s->vnew, s->g, s->vc, s->vcnew, s->gc, s->vl, s->vlnew, s->gl};
    double *us[] = {s->u, s->r, s->d, s->z, s->znew, s->y, s->zc, s->zcnew, s->yc, s->zl, s->zlnew, s->yl};
    for (size_t i = 0; i < sizeof(xs) / sizeof(xs[0]); ++i) memset(xs[i], 0, sizeof(double) * X);
    for (size_t i = 0; i < sizeof(us) / sizeof(us[0]); ++i) memset(us[i], 0, sizeof(double) * U);
}

int orc_set_x0(orc_solver *s, const double *x0) { /* tiny_api.cpp:241 */
    memcpy(s->x, x0, sizeof(double) * s->nx);
    return 0;
}
int orc_set_x_ref(orc_solver *s, const double *Xref) { /* :253 */
    memcpy(s->Xref, Xref, sizeof(double) * s->nx * s->N);
    return 0;
}
int orc_set_u_ref(orc_solver *s, const double *Uref) { /* :265 */
    memcpy(s->Uref, Uref, sizeof(double) * s->nu * (s->N - 1));
    return 0;
}
int orc_set_bound_constraints(orc_solver *s, const double *x_min, const double *x_max,
                              const double *u_min, const double *u_max) {
    const size_t X = (size_t)s->nx * s->N, U = (size_t)s->nu * (s->N - 1);
    memcpy(s->x_min, x_min, sizeof(double) * X); memcpy(s->x_max, x_max, sizeof(double) * X);
    memcpy(s->u_min, u_min, sizeof(double) * U); memcpy(s->u_max, u_max, sizeof(double) * U);
    s->en_state_bound = 1; s->en_input_bound = 1; /* bindings.cpp:206-207 */
    return 0;
}
int orc_set_cache_terms(orc_solver *s, const double *Kinf, const double *Pinf,
                        const double *Quu_inv, const double *AmBKt) {
    memcpy(s->Kinf, Kinf, sizeof(double) * s->nu * s->nx);
    memcpy(s->Pinf, Pinf, sizeof(double) * s->nx * s->nx);
    memcpy(s->Quu_inv, Quu_inv, sizeof(double) * s->nu * s->nu);
    memcpy(s->AmBKt, AmBKt, sizeof(double) * s->nx * s->nx);
    return 0;
}

static int *icopy(const int *src, int n) {
    int *p = (int *)malloc(sizeof(int) * (n ? n : 1));
    if (n) memcpy(p, src, sizeof(int) * n);
    return p;
}
static double *dcopy(const double *src, size_t n) {
    double *p = zalloc(n);
    if (n) memcpy(p, src, sizeof(double) * n);
    return p;
}

/* UNPINNED. bindings.cpp:433-478: (Ac start row 0-based, qc cone dimension, c slope) per cone,
 * applied at every knot; a side is auto-enabled when it is non-empty (:468-476). */
int orc_set_cone_constraints(orc_solver *s, int ncx, const int *Acx, const int *qcx,
                             const double *cx, int ncu, const int *Acu, const int *qcu,
                             const double *cu) {
    free(s->Acx); free(s->qcx); free(s->cx); free(s->Acu); free(s->qcu); free(s->cu);
    s->n_cone_x = ncx; s->Acx = icopy(Acx, ncx); s->qcx = icopy(qcx, ncx); s->cx = dcopy(cx, ncx);
    s->n_cone_u = ncu; s->Acu = icopy(Acu, ncu); s->qcu = icopy(qcu, ncu); s->cu = dcopy(cu, ncu);
    if (ncx > 0) s->en_state_soc = 1;
    if (ncu > 0) s->en_input_soc = 1;
    return 0;
}

/* UNPINNED. bindings.cpp:408-431: rows of Alin*s <= blin applied at every knot. */
int orc_set_linear_constraints(orc_solver *s, int nlx, const double *Alin_x, const double *blin_x,
                               int nlu, const double *Alin_u, const double *blin_u) {
    free(s->Alin_x); free(s->blin_x); free(s->Alin_u); free(s->blin_u);
    s->n_lin_x = nlx; s->Alin_x = dcopy(Alin_x, (size_t)nlx * s->nx); s->blin_x = dcopy(blin_x, nlx);
    s->n_lin_u = nlu; s->Alin_u = dcopy(Alin_u, (size_t)nlu * s->nu); s->blin_u = dcopy(blin_u, nlu);
    if (nlx > 0) s->en_state_linear = 1;
    if (nlu > 0) s->en_input_linear = 1;
    return 0;
}

/* ---------------------------------------------------------------- ADMM phases */

/* B1: admm.cpp:13-20
 *   d_i = Quu_inv * (B' * p_{i+1} + r_i [+ BPf])
 *   p_i = q_i + AmBKt * p_{i+1} - Kinf' * r_i [+ APf]            (bracketed: UNPINNED) */
void orc_backward_pass_grad(orc_solver *s) {
    const int nx = s->nx, nu = s->nu, N = s->N;
    double tmp[64];
    double *t = (nu <= 64) ? tmp : (double *)malloc(sizeof(double) * nu);
    for (int i = N - 2; i >= 0; --i) {
        const double *pn = s->p + (size_t)(i + 1) * nx;
        const double *ri = s->r + (size_t)i * nu;
        const double *qi = s->q + (size_t)i * nx;
        double *di = s->d + (size_t)i * nu;
        double *pi = s->p + (size_t)i * nx;
        for (int j = 0; j < nu; ++j) {
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc += s->Bdyn[k + (size_t)j * nx] * pn[k];
            t[j] = acc + ri[j] + s->BPf[j];
        }
        for (int j = 0; j < nu; ++j) {
            double acc = 0.0;
            for (int l = 0; l < nu; ++l) acc += s->Quu_inv[j + (size_t)l * nu] * t[l];
            di[j] = acc;
        }
        for (int rr = 0; rr < nx; ++rr) {
            double a1 = 0.0, a2 = 0.0;
            for (int k = 0; k < nx; ++k) a1 += s->AmBKt[rr + (size_t)k * nx] * pn[k];
            for (int j = 0; j < nu; ++j) a2 += s->Kinf[j + (size_t)rr * nu] * ri[j];
            pi[rr] = qi[rr] + a1 - a2 + s->APf[rr];
        }
    }
    if (t != tmp) free(t);
}

/* F1: admm.cpp:25-35
 *   u_i = -Kinf * x_i - d_i ;  x_{i+1} = A * x_i + B * u_i [+ fdyn]   (bracketed: UNPINNED) */
void orc_forward_pass(orc_solver *s) {
    const int nx = s->nx, nu = s->nu, N = s->N;
    for (int i = 0; i < N - 1; ++i) {
        const double *xi = s->x + (size_t)i * nx;
        double *ui = s->u + (size_t)i * nu;
        double *xn = s->x + (size_t)(i + 1) * nx;
        const double *di = s->d + (size_t)i * nu;
        for (int j = 0; j < nu; ++j) {
            double acc = 0.0;
            for (int k = 0; k < nx; ++k) acc += s->Kinf[j + (size_t)k * nu] * xi[k];
            ui[j] = -acc - di[j];
        }
        for (int rr = 0; rr < nx; ++rr) {
            double a1 = 0.0, a2 = 0.0;
            for (int k = 0; k < nx; ++k) a1 += s->Adyn[rr + (size_t)k * nx] * xi[k];
            for (int j = 0; j < nu; ++j) a2 += s->Bdyn[rr + (size_t)j * nx] * ui[j];
            xn[rr] = a1 + a2 + s->fdyn[rr];
        }
    }
}

/* UNPINNED: upstream project_soc -- cone { (w, t) : ||w||_2 <= mu * t }, t = last entry. */
static void project_soc(double *sv, int n, double mu) {
    double u0 = sv[n - 1] * mu;
    double a = 0.0;
    for (int i = 0; i < n - 1; ++i) a += sv[i] * sv[i];
    a = sqrt(a);
    if (a <= -u0) {
        for (int i = 0; i < n; ++i) sv[i] = 0.0;
    } else if (a <= u0) {
        /* inside */
    } else {
        double scale = 0.5 * (1.0 + u0 / a);
        for (int i = 0; i < n - 1; ++i) sv[i] = scale * sv[i];
        sv[n - 1] = scale * (a / mu);
    }
}

/* UNPINNED: half-space projection a's <= b, rows applied one after another. */
static void project_halfspaces(double *sv, int n, int rows, const double *Alin, const double *blin) {
    for (int k = 0; k < rows; ++k) {
        double dot = 0.0, nrm = 0.0;
        for (int c = 0; c < n; ++c) {
            double a = Alin[k + (size_t)c * rows];
            dot += a * sv[c];
            nrm += a * a;
        }
        if (dot > blin[k]) {
            double dist = (dot - blin[k]) / nrm;
            for (int c = 0; c < n; ++c) sv[c] -= dist * Alin[k + (size_t)c * rows];
        }
    }
}

/* S1: admm.cpp:43-59 */
void orc_update_slack(orc_solver *s) {
    const int nx = s->nx, nu = s->nu, N = s->N;
    const size_t X = (size_t)nx * N, U = (size_t)nu * (N - 1);
    for (size_t i = 0; i < U; ++i) s->znew[i] = s->u[i] + s->y[i]; /* :45 */
    for (size_t i = 0; i < X; ++i) s->vnew[i] = s->x[i] + s->g[i]; /* :46 */
    if (s->en_input_bound) /* :49-52  u_max.cwiseMin(u_min.cwiseMax(znew)) */
        for (size_t i = 0; i < U; ++i) {
            double t = s->znew[i] > s->u_min[i] ? s->znew[i] : s->u_min[i];
            s->znew[i] = t < s->u_max[i] ? t : s->u_max[i];
        }
    if (s->en_state_bound) /* :55-58 */
        for (size_t i = 0; i < X; ++i) {
            double t = s->vnew[i] > s->x_min[i] ? s->vnew[i] : s->x_min[i];
            s->vnew[i] = t < s->x_max[i] ? t : s->x_max[i];
        }
    /* UNPINNED below */
    if (s->en_state_soc && s->n_cone_x > 0) {
        for (size_t i = 0; i < X; ++i) s->vcnew[i] = s->x[i] + s->gc[i];
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < s->n_cone_x; ++k)
                project_soc(s->vcnew + (size_t)i * nx + s->Acx[k], s->qcx[k], s->cx[k]);
    }
    if (s->en_input_soc && s->n_cone_u > 0) {
        for (size_t i = 0; i < U; ++i) s->zcnew[i] = s->u[i] + s->yc[i];
        for (int i = 0; i < N - 1; ++i)
            for (int k = 0; k < s->n_cone_u; ++k)
                project_soc(s->zcnew + (size_t)i * nu + s->Acu[k], s->qcu[k], s->cu[k]);
    }
    if (s->en_state_linear && s->n_lin_x > 0) {
        for (size_t i = 0; i < X; ++i) s->vlnew[i] = s->x[i] + s->gl[i];
        for (int i = 0; i < N; ++i)
            project_halfspaces(s->vlnew + (size_t)i * nx, nx, s->n_lin_x, s->Alin_x, s->blin_x);
    }
    if (s->en_input_linear && s->n_lin_u > 0) {
        for (size_t i = 0; i < U; ++i) s->zlnew[i] = s->u[i] + s->yl[i];
        for (int i = 0; i < N - 1; ++i)
            project_halfspaces(s->zlnew + (size_t)i * nu, nu, s->n_lin_u, s->Alin_u, s->blin_u);
    }
}

/* D1: admm.cpp:65-69   y = y + u - znew ; g = g + x - vnew */
void orc_update_dual(orc_solver *s) {
    const size_t X = (size_t)s->nx * s->N, U = (size_t)s->nu * (s->N - 1);
    for (size_t i = 0; i < U; ++i) s->y[i] = s->y[i] + s->u[i] - s->znew[i];
    for (size_t i = 0; i < X; ++i) s->g[i] = s->g[i] + s->x[i] - s->vnew[i];
    if (s->en_state_soc && s->n_cone_x > 0)
        for (size_t i = 0; i < X; ++i) s->gc[i] = s->gc[i] + s->x[i] - s->vcnew[i];
    if (s->en_input_soc && s->n_cone_u > 0)
        for (size_t i = 0; i < U; ++i) s->yc[i] = s->yc[i] + s->u[i] - s->zcnew[i];
    if (s->en_state_linear && s->n_lin_x > 0)
        for (size_t i = 0; i < X; ++i) s->gl[i] = s->gl[i] + s->x[i] - s->vlnew[i];
    if (s->en_input_linear && s->n_lin_u > 0)
        for (size_t i = 0; i < U; ++i) s->yl[i] = s->yl[i] + s->u[i] - s->zlnew[i];
}

/* L1: admm.cpp:75-83 */
void orc_update_linear_cost(orc_solver *s) {
    const int nx = s->nx, nu = s->nu, N = s->N;
    const double rho = s->rho;
    for (int i = 0; i < N - 1; ++i)
        for (int j = 0; j < nu; ++j) {
            size_t e = (size_t)i * nu + j;
            double rr = -(s->Uref[e] * s->R[j]);  /* :77 */
            rr -= rho * (s->znew[e] - s->y[e]);   /* :78 */
            if (s->en_input_soc && s->n_cone_u > 0) rr -= rho * (s->zcnew[e] - s->yc[e]);
            if (s->en_input_linear && s->n_lin_u > 0) rr -= rho * (s->zlnew[e] - s->yl[e]);
            s->r[e] = rr;
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < nx; ++j) {
            size_t e = (size_t)i * nx + j;
            double qq = -(s->Xref[e] * s->Q[j]);  /* :79 */
            qq -= rho * (s->vnew[e] - s->g[e]);   /* :80 */
            if (s->en_state_soc && s->n_cone_x > 0) qq -= rho * (s->vcnew[e] - s->gc[e]);
            if (s->en_state_linear && s->n_lin_x > 0) qq -= rho * (s->vlnew[e] - s->gl[e]);
            s->q[e] = qq;
        }
    /* :81-82  p_{N-1} = -(Xref_{N-1}' * Pinf)' - rho*(vnew_{N-1} - g_{N-1}) */
    const size_t o = (size_t)(N - 1) * nx;
    for (int c = 0; c < nx; ++c) {
        double acc = 0.0;
        for (int k = 0; k < nx; ++k) acc += s->Xref[o + k] * s->Pinf[k + (size_t)c * nx];
        double pp = -acc;
        pp -= rho * (s->vnew[o + c] - s->g[o + c]);
        if (s->en_state_soc && s->n_cone_x > 0) pp -= rho * (s->vcnew[o + c] - s->gc[o + c]);
        if (s->en_state_linear && s->n_lin_x > 0) pp -= rho * (s->vlnew[o + c] - s->gl[o + c]);
        s->p[o + c] = pp;
    }
}

static double max_abs_diff(const double *a, const double *b, size_t n) {
    double m = 0.0; /* Eigen maxCoeff of cwiseAbs: first element seeds the max; all >= 0 */
    for (size_t i = 0; i < n; ++i) {
        double t = fabs(a[i] - b[i]);
        if (t > m) m = t;
    }
    return m;
}

/* R1: admm.cpp:89-107 -- inf-norms, strict '<', only when iter % check_termination == 0.
 * check_termination <= 0 is undefined behaviour in the reference (modulo by zero); the
 * restatement treats it as "never check". */
int orc_termination_condition(orc_solver *s) {
    const size_t X = (size_t)s->nx * s->N, U = (size_t)s->nu * (s->N - 1);
    if (s->check_termination <= 0) return 0;
    if (s->iter % s->check_termination == 0) {
        s->primal_residual_state = max_abs_diff(s->x, s->vnew, X);
        s->dual_residual_state = max_abs_diff(s->v, s->vnew, X) * s->rho;
        s->primal_residual_input = max_abs_diff(s->u, s->znew, U);
        s->dual_residual_input = max_abs_diff(s->z, s->znew, U) * s->rho;
        if (s->primal_residual_state < s->abs_pri_tol && s->primal_residual_input < s->abs_pri_tol &&
            s->dual_residual_state < s->abs_dua_tol && s->dual_residual_input < s->abs_dua_tol)
            return 1;
    }
    return 0;
}

/* M1: admm.cpp:109-207 (adaptive rho, :117-174, is out of scope: off by default) */
int orc_solve(orc_solver *s) {
    const size_t X = (size_t)s->nx * s->N, U = (size_t)s->nu * (s->N - 1);
    s->solved = 0; s->sol_iter = 0; s->status = 11; s->iter = 0; /* :112-115 */
    for (int i = 0; i < s->max_iter; ++i) {
        orc_forward_pass(s);        /* :132 */
        orc_update_slack(s);        /* :135 */
        orc_update_dual(s);         /* :138 */
        orc_update_linear_cost(s);  /* :141 */
        s->iter += 1;               /* :143 */
        if (s->adaptive_rho && i > 0 && i % 5 == 0) orc_rho_adaptation(s); /* :147-174 */
        if (orc_termination_condition(s)) { /* :181 */
            s->status = 1;
            s->sol_iter = s->iter; s->solved = 1;
            memcpy(s->sol_x, s->vnew, sizeof(double) * X); /* :187 */
            memcpy(s->sol_u, s->znew, sizeof(double) * U); /* :188 */
            return 0;
        }
        memcpy(s->v, s->vnew, sizeof(double) * X); /* :196 */
        memcpy(s->z, s->znew, sizeof(double) * U); /* :197 */
        orc_backward_pass_grad(s);                 /* :199 */
    }
    s->sol_iter = s->iter; s->solved = 0; /* :202-205 */
    memcpy(s->sol_x, s->vnew, sizeof(double) * X);
    memcpy(s->sol_u, s->znew, sizeof(double) * U);
    return 1;
}

/* ---------------------------------------------------------------- adaptive rho
 * benchmark_rho_adaptation (rho_benchmark.cpp:200-249) = format_matrices (:44-150) + compute_residuals (:152-180)
 * + predict_rho (:182-198) + update_matrices_with_derivatives (:200-216 in the snapshot's numbering), evaluated on
 * the block structure of the matrices the reference assembles densely:
 *   x_decision = [x_0; u_0; x_1; u_1; ...; x_{N-1}]                                              (:62-71)
 *   A_matrix   = [ rows (N-1)*nu : u_i ] ; [ rows (N-1)*nx : A x_i + B u_i - x_{i+1} ]            (:77-94)
 *   z_vector   = [ znew_i ; vnew_{i+1} ],   y_vector = [ y_i ; g_{i+1} ]                          (:97-103)
 *   P_matrix   = blkdiag(Q, R, Q, R, ..., Pinf)  with Q, R = the rho-augmented diagonals of the workspace (:106-124)
 *   q_vector   = [ Q.*x_i ; R.*u_i ]  (zero reference)                                            (:127-147)
 */
static double amax(double m, double v) { v = fabs(v); return v > m ? v : m; }

double orc_rho_adaptation(orc_solver *s) {
    const int nx = s->nx, nu = s->nu, N = s->N;
    double pri_res = 0.0, ax_max = 0.0, z_max = 0.0;          /* compute_residuals :160-164 */
    double dual_res = 0.0, px_max = 0.0, aty_max = 0.0, q_max = 0.0; /* :167-179 */
    for (int i = 0; i < N - 1; ++i) {
        const double *xi = s->x + (size_t)i * nx, *xn = s->x + (size_t)(i + 1) * nx, *ui = s->u + (size_t)i * nu;
        for (int j = 0; j < nu; ++j) { /* input rows: Ax = u_i, z = znew_i */
            const double z = s->znew[(size_t)i * nu + j];
            pri_res = amax(pri_res, ui[j] - z); ax_max = amax(ax_max, ui[j]); z_max = amax(z_max, z);
        }
        for (int r = 0; r < nx; ++r) { /* dynamics rows: Ax = A x_i + B u_i - x_{i+1}, z = vnew_{i+1} */
            double ax = 0.0;
            for (int c = 0; c < nx; ++c) ax += s->Adyn[r + (size_t)c * nx] * xi[c];
            for (int c = 0; c < nu; ++c) ax += s->Bdyn[r + (size_t)c * nx] * ui[c];
            ax -= xn[r];
            const double z = s->vnew[(size_t)(i + 1) * nx + r];
            pri_res = amax(pri_res, ax - z); ax_max = amax(ax_max, ax); z_max = amax(z_max, z);
        }
    }
    for (int i = 0; i < N; ++i) {
        const double *xi = s->x + (size_t)i * nx;
        for (int r = 0; r < nx; ++r) { /* column block of x_i */
            double px;
            if (i == N - 1) { /* Pinf block (:112) */
                px = 0.0;
                for (int c = 0; c < nx; ++c) px += s->Pinf[r + (size_t)c * nx] * xi[c];
            } else {
                px = s->Q[r] * xi[r]; /* :114 */
            }
            const double qv = s->Q[r] * xi[r]; /* :133 */
            double aty = 0.0;               /* A_matrix' * y_vector */
            if (i < N - 1) {                /* A' g_{i+1} from dynamics row block i */
                const double *gn = s->g + (size_t)(i + 1) * nx;
                for (int c = 0; c < nx; ++c) aty += s->Adyn[c + (size_t)r * nx] * gn[c];
            }
            if (i > 0) aty -= s->g[(size_t)i * nx + r]; /* -I of dynamics row block i-1 */
            px_max = amax(px_max, px); q_max = amax(q_max, qv); aty_max = amax(aty_max, aty);
            dual_res = amax(dual_res, px + qv + aty);
        }
        if (i < N - 1) {
            const double *ui = s->u + (size_t)i * nu, *gn = s->g + (size_t)(i + 1) * nx;
            for (int j = 0; j < nu; ++j) { /* column block of u_i */
                const double px = s->R[j] * ui[j], qv = px;
                double aty = s->y[(size_t)i * nu + j];
                for (int c = 0; c < nx; ++c) aty += s->Bdyn[c + (size_t)j * nx] * gn[c];
                px_max = amax(px_max, px); q_max = amax(q_max, qv); aty_max = amax(aty_max, aty);
                dual_res = amax(dual_res, px + qv + aty);
            }
        }
    }
    const double pri_norm = ax_max > z_max ? ax_max : z_max;
    double dual_norm = px_max > aty_max ? px_max : aty_max;
    if (q_max > dual_norm) dual_norm = q_max;
    /* predict_rho (rho_benchmark.cpp:182-198) */
    const double eps = 1e-10;
    const double normalized_pri = pri_res / (pri_norm + eps);
    const double normalized_dual = dual_res / (dual_norm + eps);
    const double ratio = normalized_pri / (normalized_dual + eps);
    double new_rho = s->rho * sqrt(ratio);
    if (s->adaptive_rho_enable_clipping) {
        if (new_rho < s->adaptive_rho_min) new_rho = s->adaptive_rho_min;
        if (new_rho > s->adaptive_rho_max) new_rho = s->adaptive_rho_max;
    }
    /* update_matrices_with_derivatives: first-order update of Kinf and Pinf (C1/C2 are not read by any phase) */
    const double delta = new_rho - s->rho;
    for (int i = 0; i < nu * nx; ++i) s->Kinf[i] += delta * s->dKinf_drho[i];
    for (int i = 0; i < nx * nx; ++i) s->Pinf[i] += delta * s->dPinf_drho[i];
    s->rho = new_rho;
    return new_rho;
}

void orc_set_adaptive_rho(orc_solver *s, int enabled, double rho_min, double rho_max, int clip) {
    s->adaptive_rho = enabled; s->adaptive_rho_min = rho_min; s->adaptive_rho_max = rho_max;
    s->adaptive_rho_enable_clipping = clip;
}

int orc_set_sensitivity(orc_solver *s, const double *dK, const double *dP) {
    memcpy(s->dKinf_drho, dK, sizeof(double) * s->nu * s->nx);
    memcpy(s->dPinf_drho, dP, sizeof(double) * s->nx * s->nx);
    return 0;
}

long orc_bench_solves(orc_solver *s, const double *x0s, int count, int reps) {
    long iters = 0;
    for (int r = 0; r < reps; ++r)
        for (int b = 0; b < count; ++b) {
            orc_reset_workspace(s);
            orc_set_x0(s, x0s + (size_t)b * s->nx);
            orc_solve(s);
            iters += s->iter;
        }
    return iters;
}

void orc_solve_batch(orc_solver *s, const double *x0s, int count, double *sol_x, double *sol_u,
                     int *iters, int *status, double *residuals) {
    const size_t X = (size_t)s->nx * s->N, U = (size_t)s->nu * (s->N - 1);
    for (int b = 0; b < count; ++b) {
        orc_reset_workspace(s);
        orc_set_x0(s, x0s + (size_t)b * s->nx);
        orc_solve(s);
        memcpy(sol_x + (size_t)b * X, s->sol_x, sizeof(double) * X);
        memcpy(sol_u + (size_t)b * U, s->sol_u, sizeof(double) * U);
        if (iters) iters[b] = s->iter;
        if (status) status[b] = s->status;
        if (residuals) {
            residuals[4 * (size_t)b + 0] = s->primal_residual_state;
            residuals[4 * (size_t)b + 1] = s->dual_residual_state;
            residuals[4 * (size_t)b + 2] = s->primal_residual_input;
            residuals[4 * (size_t)b + 3] = s->dual_residual_input;
        }
    }
}

/* ---------------------------------------------------------------- test accessors */
static double *find_array(orc_solver *s, const char *n, size_t *count) {
    const size_t X = (size_t)s->nx * s->N, U = (size_t)s->nu * (s->N - 1);
    const size_t nx = s->nx, nu = s->nu;
#define ORC_ARR(nm, ptr, cnt) if (!strcmp(n, nm)) { *count = (cnt); return (ptr); }
    ORC_ARR("x", s->x, X) ORC_ARR("u", s->u, U) ORC_ARR("q", s->q, X) ORC_ARR("r", s->r, U)
    ORC_ARR("p", s->p, X) ORC_ARR("d", s->d, U) ORC_ARR("v", s->v, X) ORC_ARR("vnew", s->vnew, X)
    ORC_ARR("z", s->z, U) ORC_ARR("znew", s->znew, U) ORC_ARR("g", s->g, X) ORC_ARR("y", s->y, U)
    ORC_ARR("Q", s->Q, nx) ORC_ARR("R", s->R, nu) ORC_ARR("Adyn", s->Adyn, nx * nx)
    ORC_ARR("Bdyn", s->Bdyn, nx * nu) ORC_ARR("fdyn", s->fdyn, nx)
    ORC_ARR("x_min", s->x_min, X) ORC_ARR("x_max", s->x_max, X) ORC_ARR("u_min", s->u_min, U)
    ORC_ARR("u_max", s->u_max, U) ORC_ARR("Xref", s->Xref, X) ORC_ARR("Uref", s->Uref, U)
    ORC_ARR("Kinf", s->Kinf, nu * nx) ORC_ARR("Pinf", s->Pinf, nx * nx)
    ORC_ARR("Quu_inv", s->Quu_inv, nu * nu) ORC_ARR("AmBKt", s->AmBKt, nx * nx)
    ORC_ARR("C1", s->Quu_inv, nu * nu) ORC_ARR("C2", s->AmBKt, nx * nx)
    ORC_ARR("APf", s->APf, nx) ORC_ARR("BPf", s->BPf, nu)
    ORC_ARR("dKinf_drho", s->dKinf_drho, nu * nx) ORC_ARR("dPinf_drho", s->dPinf_drho, nx * nx)
    ORC_ARR("sol_x", s->sol_x, X) ORC_ARR("sol_u", s->sol_u, U)
    ORC_ARR("vcnew", s->vcnew, X) ORC_ARR("gc", s->gc, X) ORC_ARR("zcnew", s->zcnew, U) ORC_ARR("yc", s->yc, U)
    ORC_ARR("vlnew", s->vlnew, X) ORC_ARR("gl", s->gl, X) ORC_ARR("zlnew", s->zlnew, U) ORC_ARR("yl", s->yl, U)
#undef ORC_ARR
    return NULL;
}

int orc_get(orc_solver *s, const char *name, double *out, int capacity) {
    size_t n = 0;
    double *p = find_array(s, name, &n);
    if (!p) return -1;
    if ((size_t)capacity < n) return -2;
    memcpy(out, p, sizeof(double) * n);
    return (int)n;
}

int orc_put(orc_solver *s, const char *name, const double *in, int count) {
    size_t n = 0;
    double *p = find_array(s, name, &n);
    if (!p) return -1;
    if ((size_t)count != n) return -2;
    memcpy(p, in, sizeof(double) * n);
    return 0;
}

void orc_update_settings(orc_solver *s, double abs_pri_tol, double abs_dua_tol, int max_iter,
                         int check_termination, int en_state_bound, int en_input_bound,
                         int en_state_soc, int en_input_soc, int en_state_linear,
                         int en_input_linear) {
    s->abs_pri_tol = abs_pri_tol; s->abs_dua_tol = abs_dua_tol;
    s->max_iter = max_iter; s->check_termination = check_termination;
    s->en_state_bound = en_state_bound; s->en_input_bound = en_input_bound;
    s->en_state_soc = en_state_soc; s->en_input_soc = en_input_soc;
    s->en_state_linear = en_state_linear; s->en_input_linear = en_input_linear;
}

void orc_get_stats(orc_solver *s, int *istats, double *dstats) {
    istats[0] = s->iter; istats[1] = s->status; istats[2] = s->solved; istats[3] = s->sol_iter;
    istats[4] = s->riccati_iters;
    dstats[0] = s->primal_residual_state; dstats[1] = s->dual_residual_state;
    dstats[2] = s->primal_residual_input; dstats[3] = s->dual_residual_input;
    dstats[4] = s->rho;
}

void orc_set_iter(orc_solver *s, int iter) { s->iter = iter; }
