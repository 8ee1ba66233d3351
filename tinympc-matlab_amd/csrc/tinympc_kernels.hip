// tinympc_kernels.hip -- setup-time HIP kernels for gfx950 (MI355X, CDNA4), FP64.
//
//   k_precompute        P1  tiny_precompute_and_set_cache         (reference tiny_api.cpp:124-190)
//   k_build_operators       fuses the cache into the two 1-step sweep operators used by the solve
//   k_build_tables          per-knot clamp bounds / linear-cost reference terms
//
// The solve kernel itself (M1: F1, S1, D1, L1, R1, C1, B1) is in tinympc_solve.hip.
#include "tinympc_device.h"

namespace tinympc {

// =====================================================================================
// P1: LQR cache precompute -- one workgroup, matrices in LDS (global scratch if too big)
// =====================================================================================
constexpr int PRE_THREADS = 256;

// C (m x n) = op(A) (m x k) * op(B) (k x n), column-major. TA: A is stored k x m. TB: B is stored n x k.
// k-loop order l = 0..k-1 per output element, as in the oracle's matmul.
template <bool TA, bool TB>
__device__ void wg_gemm(double *C, const double *A, const double *B, int m, int k, int n) {
    for (int idx = threadIdx.x; idx < m * n; idx += PRE_THREADS) {
        const int i = idx % m, j = idx / m;
        double acc = 0.0;
        for (int l = 0; l < k; ++l) {
            const double a = TA ? A[l + (size_t)i * k] : A[i + (size_t)l * m];
            const double b = TB ? B[j + (size_t)l * n] : B[l + (size_t)j * k];
            acc += a * b;
        }
        C[idx] = acc;
    }
    __syncthreads();
}

// In-workgroup inverse by partial-pivot LU and a solve against the identity (what Eigen's
// dynamic-size inverse() does at tiny_api.cpp:154,169). `lu` holds M on entry and is destroyed.
__device__ void wg_lu_inverse(double *inv, double *lu, int *perm, int n) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += PRE_THREADS) perm[i] = i;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        if (tid == 0) {
            int piv = k;
            double best = fabs(lu[k + (size_t)k * n]);
            for (int i = k + 1; i < n; ++i) {
                const double a = fabs(lu[i + (size_t)k * n]);
                if (a > best) {
                    best = a;
                    piv = i;
                }
            }
            perm[n] = piv;
        }
        __syncthreads();
        const int piv = perm[n];
        if (piv != k) {
            for (int j = tid; j < n; j += PRE_THREADS) {
                const double t = lu[k + (size_t)j * n];
                lu[k + (size_t)j * n] = lu[piv + (size_t)j * n];
                lu[piv + (size_t)j * n] = t;
            }
            if (tid == 0) {
                const int t = perm[k];
                perm[k] = perm[piv];
                perm[piv] = t;
            }
        }
        __syncthreads();
        const double pivot = lu[k + (size_t)k * n];
        for (int i = k + 1 + tid; i < n; i += PRE_THREADS) lu[i + (size_t)k * n] /= pivot;
        __syncthreads();
        const int rem = n - k - 1;
        for (int idx = tid; idx < rem * rem; idx += PRE_THREADS) {
            const int i = k + 1 + idx % rem, j = k + 1 + idx / rem;
            lu[i + (size_t)j * n] -= lu[i + (size_t)k * n] * lu[k + (size_t)j * n];
        }
        __syncthreads();
    }
    for (int c = tid; c < n; c += PRE_THREADS) {
        double *col = inv + (size_t)c * n;
        for (int i = 0; i < n; ++i) col[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) col[i] -= lu[i + (size_t)j * n] * col[j];
        for (int i = n - 1; i >= 0; --i) {
            for (int j = i + 1; j < n; ++j) col[i] -= lu[i + (size_t)j * n] * col[j];
            col[i] /= lu[i + (size_t)i * n];
        }
    }
    __syncthreads();
}

__device__ double wg_max_abs_diff(const double *a, const double *b, int n, double *red) {
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += PRE_THREADS) m = fmax(m, fabs(a[i] - b[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = PRE_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

size_t precompute_scratch_doubles(int nx, int nu) {
    // Ktp1,Kinf,BtP,T1,T2 (nu*nx) + Ptp1,Pinf,AtP,AmBK,T3 (nx*nx) + S,Sinv (nu*nu) + Q1d,R1d,Pf + perm
    return (size_t)5 * nu * nx + (size_t)5 * nx * nx + (size_t)2 * nu * nu + 2 * nx + nu + (nu + 2);
}

// Follows tiny_api.cpp:124-190 step for step, including its two parity traps: rho is added to the
// (already augmented) diagonals a second time (:134-135) and the fixed-point iteration is truncated
// at max|K - Kprev| < 1e-5 keeping THAT iteration's K and P (:157).
__global__ void __launch_bounds__(PRE_THREADS) k_precompute(const PrecomputeParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *red = lds;  // PRE_THREADS doubles, always in LDS
    double *w = p.use_lds ? (lds + PRE_THREADS) : p.scratch;
    const int nx = p.nx, nu = p.nu, tid = threadIdx.x;
    const double rho = p.rho;
    double *Ktp1 = w;                w += nu * nx;
    double *Kinf = w;                w += nu * nx;
    double *BtP = w;                 w += nu * nx;
    double *T1 = w;                  w += nu * nx;
    double *T2 = w;                  w += nu * nx;
    double *Ptp1 = w;                w += nx * nx;
    double *Pinf = w;                w += nx * nx;
    double *AtP = w;                 w += nx * nx;
    double *AmBK = w;                w += nx * nx;
    double *T3 = w;                  w += nx * nx;
    double *S = w;                   w += nu * nu;
    double *Sinv = w;                w += nu * nu;
    double *Q1d = w;                 w += nx;
    double *R1d = w;                 w += nu;
    double *Pf = w;                  w += nx;
    int *perm = reinterpret_cast<int *>(w);  // nu + 1 ints
    const double *A = p.A, *B = p.B;

    for (int i = tid; i < nx; i += PRE_THREADS) Q1d[i] = p.Qd[i] + rho;  // :134
    for (int i = tid; i < nu; i += PRE_THREADS) R1d[i] = p.Rd[i] + rho;  // :135
    for (int i = tid; i < nu * nx; i += PRE_THREADS) Ktp1[i] = 0.0;
    for (int i = tid; i < nx * nx; i += PRE_THREADS) Ptp1[i] = (i % nx == i / nx) ? rho : 0.0;  // :148
    __syncthreads();

    int steps = 1000;
    for (int it = 0; it < 1000; ++it) {  // :152
        // :154  Kinf = (R1 + B'*P*B).inverse() * B' * P * A      (evaluated left to right)
        wg_gemm<true, false>(BtP, B, Ptp1, nu, nx, nx);
        wg_gemm<false, false>(S, BtP, B, nu, nx, nu);
        for (int i = tid; i < nu * nu; i += PRE_THREADS) S[i] = ((i % nu == i / nu) ? R1d[i % nu] : 0.0) + S[i];
        __syncthreads();
        wg_lu_inverse(Sinv, S, perm, nu);
        wg_gemm<false, true>(T1, Sinv, B, nu, nu, nx);
        wg_gemm<false, false>(T2, T1, Ptp1, nu, nx, nx);
        wg_gemm<false, false>(Kinf, T2, A, nu, nx, nx);
        // :155  Pinf = Q1 + A'*P*(A - B*Kinf)
        wg_gemm<true, false>(AtP, A, Ptp1, nx, nx, nx);
        wg_gemm<false, false>(T3, B, Kinf, nx, nu, nx);
        for (int i = tid; i < nx * nx; i += PRE_THREADS) AmBK[i] = A[i] - T3[i];
        __syncthreads();
        wg_gemm<false, false>(T3, AtP, AmBK, nx, nx, nx);
        for (int i = tid; i < nx * nx; i += PRE_THREADS) Pinf[i] = ((i % nx == i / nx) ? Q1d[i % nx] : 0.0) + T3[i];
        __syncthreads();
        // :157
        const double md = wg_max_abs_diff(Kinf, Ktp1, nu * nx, red);
        if (md < 1e-5) {
            steps = it + 1;
            break;
        }
        for (int i = tid; i < nu * nx; i += PRE_THREADS) Ktp1[i] = Kinf[i];  // :164
        for (int i = tid; i < nx * nx; i += PRE_THREADS) Ptp1[i] = Pinf[i];  // :165
        __syncthreads();
    }
    // :169  Quu_inv = (R1 + B'*Pinf*B).inverse()
    wg_gemm<true, false>(BtP, B, Pinf, nu, nx, nx);
    wg_gemm<false, false>(S, BtP, B, nu, nx, nu);
    for (int i = tid; i < nu * nu; i += PRE_THREADS) S[i] = ((i % nu == i / nu) ? R1d[i % nu] : 0.0) + S[i];
    __syncthreads();
    wg_lu_inverse(Sinv, S, perm, nu);
    // :170  AmBKt = (A - B*Kinf)'
    wg_gemm<false, false>(T3, B, Kinf, nx, nu, nx);
    for (int i = tid; i < nx * nx; i += PRE_THREADS) AmBK[i] = A[i] - T3[i];
    __syncthreads();
    for (int i = tid; i < nu * nu; i += PRE_THREADS) p.Quu_inv[i] = Sinv[i];
    for (int i = tid; i < nu * nx; i += PRE_THREADS) p.Kinf[i] = Kinf[i];
    for (int i = tid; i < nx * nx; i += PRE_THREADS) {
        p.Pinf[i] = Pinf[i];
        p.AmBKt[(i / nx) + (size_t)(i % nx) * nx] = AmBK[i];
    }
    // Affine-dynamics terms (upstream TinyMPC main; PARITY UNPINNED): APf = AmBKt*Pinf*f, BPf = B'*Pinf*f
    wg_gemm<false, false>(Pf, Pinf, p.fdyn, nx, nx, 1);
    wg_gemm<true, false>(p.APf, AmBK, Pf, nx, nx, 1);
    wg_gemm<true, false>(p.BPf, B, Pf, nu, nx, 1);
    if (tid == 0) p.info[0] = steps;
}

hipError_t launch_precompute(const PrecomputeParams &p, hipStream_t stream) {
    size_t lds = sizeof(double) * PRE_THREADS;
    if (p.use_lds) lds += sizeof(double) * precompute_scratch_doubles(p.nx, p.nu);
    hipLaunchKernelGGL(k_precompute, dim3(1), dim3(PRE_THREADS), lds, stream, p);
    return hipGetLastError();
}

// =====================================================================================
// The .m class's Riccati recursions (compute_cache_terms / solve_lqr), same workgroup scheme
// =====================================================================================
// Largest singular value of D (m x n, column-major) = sqrt(lambda_max) of the smaller Gram matrix,
// by cyclic Jacobi rotations (thread 0; the matrix is at most 32 x 32). `gram` holds min(m,n)^2 doubles.
__device__ double wg_spectral_norm(const double *D, int m, int n, double *gram, double *red) {
    const int g = m < n ? m : n;
    for (int idx = threadIdx.x; idx < g * g; idx += PRE_THREADS) {
        const int i = idx % g, j = idx / g;
        double acc = 0.0;
        if (m <= n)
            for (int l = 0; l < n; ++l) acc += D[i + (size_t)l * m] * D[j + (size_t)l * m];  // D D'
        else
            for (int l = 0; l < m; ++l) acc += D[l + (size_t)i * m] * D[l + (size_t)j * m];  // D' D
        gram[idx] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int sweep = 0; sweep < 40; ++sweep) {
            double off = 0.0, diag = 0.0;
            for (int j = 0; j < g; ++j)
                for (int i = 0; i < g; ++i) (i == j ? diag : off) += gram[i + j * g] * gram[i + j * g];
            if (off <= 1e-30 * diag || off == 0.0) break;
            for (int p = 0; p < g - 1; ++p)
                for (int q = p + 1; q < g; ++q) {
                    const double apq = gram[p + q * g];
                    if (apq == 0.0) continue;
                    const double theta = (gram[q + q * g] - gram[p + p * g]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                    for (int k = 0; k < g; ++k) {  // columns p, q
                        const double akp = gram[k + p * g], akq = gram[k + q * g];
                        gram[k + p * g] = c * akp - sn * akq;
                        gram[k + q * g] = sn * akp + c * akq;
                    }
                    for (int k = 0; k < g; ++k) {  // rows p, q
                        const double apk = gram[p + k * g], aqk = gram[q + k * g];
                        gram[p + k * g] = c * apk - sn * aqk;
                        gram[q + k * g] = sn * apk + c * aqk;
                    }
                }
        }
        double lmax = 0.0;
        for (int i = 0; i < g; ++i) lmax = fmax(lmax, gram[i + i * g]);
        red[0] = sqrt(lmax);
    }
    __syncthreads();
    const double r = red[0];
    __syncthreads();
    return r;
}

__device__ double wg_sum_sq_diff(const double *a, const double *b, int n, double *red) {
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += PRE_THREADS) m += (a[i] - b[i]) * (a[i] - b[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = PRE_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__device__ double wg_max_abs(const double *a, int n, double *red) {
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += PRE_THREADS) m = fmax(m, fabs(a[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = PRE_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

size_t lqr_scratch_doubles(int nx, int nu) {
    const int g = nx < nu ? nx : nu;
    // Kprev,K,BtP,T1,dK (nu*nx) + Pprev,P,AtP,AmBK,T3 (nx*nx) + S,Sinv (nu*nu) + gram + perm
    return (size_t)5 * nu * nx + (size_t)5 * nx * nx + (size_t)2 * nu * nu + (size_t)g * g + (nu + 2);
}

// TinyMPC.m:194-221 (compute_cache_terms) and :336-366 (solve_lqr, iterative branch run to stationarity in
// place of MATLAB's idare): K = (R_rho + B'PB + reg I)^-1 B'PA ; P = Q_rho + A'P(A - BK), from P0.
__global__ void __launch_bounds__(PRE_THREADS) k_lqr(const LqrParams p) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *red = lds;
    double *w = p.use_lds ? (lds + PRE_THREADS) : p.scratch;
    const int nx = p.nx, nu = p.nu, tid = threadIdx.x;
    double *Kprev = w;               w += nu * nx;
    double *K = w;                   w += nu * nx;
    double *BtP = w;                 w += nu * nx;
    double *T1 = w;                  w += nu * nx;
    double *dK = w;                  w += nu * nx;
    double *Pprev = w;               w += nx * nx;
    double *P = w;                   w += nx * nx;
    double *AtP = w;                 w += nx * nx;
    double *AmBK = w;                w += nx * nx;
    double *T3 = w;                  w += nx * nx;
    double *S = w;                   w += nu * nu;
    double *Sinv = w;                w += nu * nu;
    double *gram = w;                w += (nx < nu ? nx : nu) * (nx < nu ? nx : nu);
    int *perm = reinterpret_cast<int *>(w);
    const double *A = p.A, *B = p.B;
    auto Qrho = [&](int i) { return p.Q[i] + ((i % nx == i / nx) ? p.rho : 0.0); };
    auto Rrho = [&](int i) { return p.R[i] + ((i % nu == i / nu) ? p.rho : 0.0); };

    for (int i = tid; i < nu * nx; i += PRE_THREADS) Kprev[i] = 0.0;
    for (int i = tid; i < nx * nx; i += PRE_THREADS) Pprev[i] = p.p0_augmented ? Qrho(i) : p.Q[i];
    __syncthreads();

    int steps = p.max_iter;
    double best = 1e300;
    int since_best = 0;
    const double rank_bound = sqrt((double)(nx < nu ? nx : nu));
    for (int it = 1; it <= p.max_iter; ++it) {
        wg_gemm<true, false>(BtP, B, Pprev, nu, nx, nx);
        wg_gemm<false, false>(S, BtP, B, nu, nx, nu);
        for (int i = tid; i < nu * nu; i += PRE_THREADS) S[i] = Rrho(i) + S[i] + ((i % nu == i / nu) ? p.reg : 0.0);
        __syncthreads();
        wg_lu_inverse(Sinv, S, perm, nu);
        wg_gemm<false, false>(T1, BtP, A, nu, nx, nx);
        wg_gemm<false, false>(K, Sinv, T1, nu, nu, nx);
        wg_gemm<true, false>(AtP, A, Pprev, nx, nx, nx);
        wg_gemm<false, false>(T3, B, K, nx, nu, nx);
        for (int i = tid; i < nx * nx; i += PRE_THREADS) AmBK[i] = A[i] - T3[i];
        __syncthreads();
        wg_gemm<false, false>(T3, AtP, AmBK, nx, nx, nx);
        for (int i = tid; i < nx * nx; i += PRE_THREADS) P[i] = Qrho(i) + T3[i];
        __syncthreads();
        bool stop = false;
        if (p.norm_kind == 2) {
            // ||.||_2 <= ||.||_F <= sqrt(rank) ||.||_2 decides almost every iteration without the eigen-solve
            const double fro = sqrt(wg_sum_sq_diff(K, Kprev, nu * nx, red));
            if (fro < p.tol) {
                stop = true;
            } else if (fro < p.tol * rank_bound) {
                for (int i = tid; i < nu * nx; i += PRE_THREADS) dK[i] = K[i] - Kprev[i];
                __syncthreads();
                stop = wg_spectral_norm(dK, nu, nx, gram, red) < p.tol;
            }
        } else {
            const double md = wg_max_abs_diff(K, Kprev, nu * nx, red);
            const double scale = fmax(1.0, wg_max_abs(K, nu * nx, red));
            stop = md < p.tol * scale;
            // rounding noise can sit above a very small tol: a converging recursion sets a new minimum of the
            // change every step, so 50 steps without one mean the noise floor has been reached
            if (md < best) {
                best = md;
                since_best = 0;
            } else if (++since_best >= 50 && md < 1e-9 * scale) {
                stop = true;
            }
        }
        for (int i = tid; i < nu * nx; i += PRE_THREADS) Kprev[i] = K[i];
        for (int i = tid; i < nx * nx; i += PRE_THREADS) Pprev[i] = P[i];
        __syncthreads();
        if (stop && it >= p.min_iter) {
            steps = it;
            break;
        }
    }
    // C1 = inv(R_rho + B'PB) (no regulariser, :219/:363), C2 = (A - BK)' with the final K (:218/:364)
    wg_gemm<true, false>(BtP, B, P, nu, nx, nx);
    wg_gemm<false, false>(S, BtP, B, nu, nx, nu);
    for (int i = tid; i < nu * nu; i += PRE_THREADS) S[i] = Rrho(i) + S[i];
    __syncthreads();
    wg_lu_inverse(Sinv, S, perm, nu);
    wg_gemm<false, false>(T3, B, K, nx, nu, nx);
    for (int i = tid; i < nu * nu; i += PRE_THREADS) p.C1[i] = Sinv[i];
    for (int i = tid; i < nu * nx; i += PRE_THREADS) p.K[i] = K[i];
    for (int i = tid; i < nx * nx; i += PRE_THREADS) {
        p.P[i] = P[i];
        p.C2[(i / nx) + (size_t)(i % nx) * nx] = A[i] - T3[i];
    }
    if (tid == 0) p.info[0] = steps;
}

hipError_t launch_lqr(const LqrParams &p, hipStream_t stream) {
    size_t lds = sizeof(double) * PRE_THREADS;
    if (p.use_lds) lds += sizeof(double) * lqr_scratch_doubles(p.nx, p.nu);
    hipLaunchKernelGGL(k_lqr, dim3(1), dim3(PRE_THREADS), lds, stream, p);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_finite_diff(const FiniteDiffParams p) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.count; i += gridDim.x * 256) p.out[i] = (p.hi[i] - p.lo[i]) / p.h;
}

__global__ void __launch_bounds__(256) k_fill(double *dst, size_t count, double value) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) dst[i] = value;
}

// The small per-instance arrays of a workspace reset in ONE launch: iteration count / status <- 0, the four residual norms <- 0, rho <- the
// setup value (three separate operations were three launch gaps of a 1.76 ms cold-started step).
__global__ void __launch_bounds__(256) k_reset_stats(int *istats, double *dstats, double *rho_inst, int batch, double rho) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < batch; i += gridDim.x * 256) {
        istats[2 * i] = 0;
        istats[2 * i + 1] = 0;
        dstats[4 * i] = 0.0;
        dstats[4 * i + 1] = 0.0;
        dstats[4 * i + 2] = 0.0;
        dstats[4 * i + 3] = 0.0;
        rho_inst[i] = rho;
    }
}
hipError_t launch_reset_stats(int *istats, double *dstats, double *rho_inst, int batch, double rho, hipStream_t stream) {
    const int blocks = (batch + 255) / 256;
    hipLaunchKernelGGL(k_reset_stats, dim3(blocks < 1024 ? (blocks ? blocks : 1) : 1024), dim3(256), 0, stream, istats, dstats, rho_inst, batch, rho);
    return hipGetLastError();
}

// Setup of a SMALL handle in ONE launch (round 5): the problem data from its pinned staging copy into the device arena (the kernel reads host
// memory itself: no copy-engine hop), the zero-initialised span, the bounds' +-1e17, the per-instance statistics and rho, the session's mailbox
// -- five stream operations (copy, two memsets, two fills) of ~3-5 us each in front of k_precompute.
__global__ void __launch_bounds__(256) k_setup_init(const SetupInitParams p) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < p.upload_doubles; i += nth) p.upload_dst[i] = p.stage[i];
    for (size_t i = tid; i < p.zero_doubles; i += nth) p.zero[i] = 0.0;
    for (size_t i = tid; i < p.X + p.U; i += nth) {
        if (i < p.X) { p.xmin[i] = -p.inf; p.xmax[i] = p.inf; }
        else { p.umin[i - p.X] = -p.inf; p.umax[i - p.X] = p.inf; }
    }
    for (size_t i = tid; i < (size_t)p.batch; i += nth) p.rho_inst[i] = p.rho;  // (iteration counts, statuses, residuals: inside the zeroed span)
    if (p.mail)
        for (size_t i = tid; i < 64; i += nth) p.mail[i] = 0.0;
}
hipError_t launch_setup_init(const SetupInitParams &p, hipStream_t stream) {
    const size_t work = p.zero_doubles > p.upload_doubles ? p.zero_doubles : p.upload_doubles;
    const size_t blocks = (work + 255) / 256;
    hipLaunchKernelGGL(k_setup_init, dim3((unsigned)(blocks < 256 ? (blocks ? blocks : 1) : 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Setup: the four bound arrays <- -inf / +inf (TinyMPC.m:261-264's 1e17) in ONE launch
__global__ void __launch_bounds__(256) k_fill_bounds(double *xmin, double *xmax, size_t X, double *umin, double *umax, size_t U, double inf) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < X + U; i += (size_t)gridDim.x * 256) {
        if (i < X) { xmin[i] = -inf; xmax[i] = inf; }
        else { umin[i - X] = -inf; umax[i - X] = inf; }
    }
}
hipError_t launch_fill_bounds(double *xmin, double *xmax, size_t X, double *umin, double *umax, size_t U, double inf, hipStream_t stream) {
    const size_t blocks = (X + U + 255) / 256;
    hipLaunchKernelGGL(k_fill_bounds, dim3((unsigned)(blocks < 1024 ? (blocks ? blocks : 1) : 1024)), dim3(256), 0, stream, xmin, xmax, X, umin, umax, U, inf);
    return hipGetLastError();
}

hipError_t launch_fill(double *dst, size_t count, double value, hipStream_t stream) {
    const unsigned blocks = (unsigned)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_fill, dim3(blocks ? blocks : 1), dim3(256), 0, stream, dst, count, value);
    return hipGetLastError();
}

hipError_t launch_finite_diff(const FiniteDiffParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_finite_diff, dim3(1), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// =====================================================================================
// Sweep operators: one (nx+nu)x(nx+nu) mat-vec per step instead of the reference's 2-3
// =====================================================================================
//  forward  (admm.cpp:29,33):  u_i = -Kinf x_i - d_i ;  x_{i+1} = A x_i + B u_i + f
//           = [ A-B*Kinf  -B ] [x_i]   [f]
//             [ -Kinf     -I ] [d_i] + [0]
//  backward (admm.cpp:17-18):  d_i = Quu_inv (B' p_{i+1} + r_i + BPf) ; p_i = q_i + AmBKt p_{i+1} - Kinf' r_i + APf
//           = [ AmBKt        -Kinf'   ] [p_{i+1}]   [q_i + APf      ]
//             [ Quu_inv*B'   Quu_inv  ] [r_i    ] + [Quu_inv * BPf  ]
// A-B*Kinf is formed from the installed Kinf (not from AmBKt) because the reference's forward pass
// uses Adyn, Bdyn and Kinf; the backward operator uses the installed AmBKt, as the reference does.
__global__ void __launch_bounds__(256) k_build_operators(const OperatorParams p) {
    const int nx = p.nx, nu = p.nu, W = p.W, KT = p.KT, nxu = nx + nu;
    double *Mf = p.ops, *Mb = p.ops + (size_t)W * KT, *cf = p.ops + (size_t)2 * W * KT, *cb = cf + W, *dg = cb + W;
    for (int idx = threadIdx.x; idx < W * KT; idx += 256) {
        const int r = idx / KT, k = idx % KT;
        double f = 0.0, b = 0.0;
        if (r < nx && k < nx) {
            double bk = 0.0;
            for (int j = 0; j < nu; ++j) bk += p.B[r + (size_t)j * nx] * p.Kinf[j + (size_t)k * nu];
            f = p.A[r + (size_t)k * nx] - bk;
            b = p.AmBKt[r + (size_t)k * nx];
        } else if (r < nx && k < nxu) {
            f = -p.B[r + (size_t)(k - nx) * nx];
            b = -p.Kinf[(k - nx) + (size_t)r * nu];
        } else if (r < nxu && k < nx) {
            const int j = r - nx;
            f = -p.Kinf[j + (size_t)k * nu];
            double qb = 0.0;
            for (int l = 0; l < nu; ++l) qb += p.Quu_inv[j + (size_t)l * nu] * p.B[k + (size_t)l * nx];
            b = qb;
        } else if (r < nxu && k < nxu) {
            const int j = r - nx;
            f = (k - nx == j) ? -1.0 : 0.0;
            b = p.Quu_inv[j + (size_t)(k - nx) * nu];
        }
        Mf[idx] = f;
        Mb[idx] = b;
    }
    for (int r = threadIdx.x; r < W; r += 256) {
        double vf = 0.0, vb = 0.0, vd = 0.0;
        if (r < nx) {
            vf = p.fdyn[r];
            vb = p.APf[r];
            vd = p.Qd[r];
        } else if (r < nxu) {
            const int j = r - nx;
            for (int l = 0; l < nu; ++l) vb += p.Quu_inv[j + (size_t)l * nu] * p.BPf[l];
            vd = p.Rd[j];
        }
        cf[r] = vf;
        cb[r] = vb;
        dg[r] = vd;
    }
}

hipError_t launch_build_operators(const OperatorParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_operators, dim3(1), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Per-knot tables in the solve kernel's [knot][row-lane] layout. A disabled bound family becomes
// (-inf, +inf): min(hi, max(lo, s)) is then the identity, which is what the reference does when
// en_state_bound / en_input_bound are off (admm.cpp:49, 55).
// Each table has N+2 rows: row k+1 is knot k; rows 0 and N+1 are padding that the solve kernel's
// one-step-ahead operand prefetch may touch at either end of a sweep (values never used).
__global__ void __launch_bounds__(256) k_build_tables(const TableParams p) {
    const int nx = p.nx, nu = p.nu, N = p.N, W = p.W, nxu = nx + nu;
    const size_t TR = (size_t)(N + 2) * W;
    double *lo = p.tables, *hi = lo + TR, *lr = hi + TR, *pn = lr + TR;
    const double *dg = p.ops + (size_t)2 * W * p.KT + 2 * W;
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < (N + 2) * W; idx += gridDim.x * 256) {
        const int kn = idx / W - 1, r = idx % W;
        double l = -inf, h = inf, ref = 0.0;
        if (kn < 0 || kn >= N) {
            // padding row
        } else if (r < nx) {
            if (p.en_state_bound) {
                l = p.x_min[r + (size_t)kn * nx];
                h = p.x_max[r + (size_t)kn * nx];
            }
            ref = -(p.Xref[r + (size_t)kn * nx] * dg[r]);  // admm.cpp:79
        } else if (r < nxu && kn < N - 1) {
            const int j = r - nx;
            if (p.en_input_bound) {
                l = p.u_min[j + (size_t)kn * nu];
                h = p.u_max[j + (size_t)kn * nu];
            }
            ref = -(p.Uref[j + (size_t)kn * nu] * dg[r]);  // admm.cpp:77
        }
        lo[idx] = l;
        hi[idx] = h;
        lr[idx] = ref;
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < W; c += 256) {
            double acc = 0.0;
            if (c < nx) {
                for (int k = 0; k < nx; ++k) acc += p.Xref[k + (size_t)(N - 1) * nx] * p.Pinf[k + (size_t)c * nx];
                acc = -acc;  // admm.cpp:81
            }
            pn[c] = acc;
        }
    }
}

hipError_t launch_build_tables(const TableParams &p, hipStream_t stream) {
    const int blocks = ((p.N + 2) * p.W + 255) / 256;
    hipLaunchKernelGGL(k_build_tables, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, p);
    return hipGetLastError();
}


}  // namespace tinympc
