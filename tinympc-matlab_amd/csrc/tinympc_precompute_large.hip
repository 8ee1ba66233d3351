// tinympc_precompute_large.hip -- P1, the LQR cache precompute (reference tiny_api.cpp:124-190), for LARGE systems (nx+nu > 64).
//
// k_precompute (tinympc_kernels.hip) runs the whole Riccati fixed point in ONE workgroup: right for the quadrotor (0.6 ms),
// hopeless at the sizes layout M solves -- 2 nx^3 multiply-adds per iteration on 256 threads: 2.8 s of setup at nx=224, a
// minute at nx=480. Here every matrix product of an iteration is its own launch over the whole chip, on the FP64 matrix cores:
//   k_dgemm_mfma   C = op(A) op(B) (+ an epilogue: C = X - AB, or + a diagonal), one 16x16 tile of C per wavefront,
//                  v_mfma_f64_16x16x4_f64 over k in blocks of 4, operands straight from L2 (the matrices are <= 2 MB)
//   k_gain_solve   the nu x nu part of an iteration in one workgroup: S = R1 + (B'P)B, its inverse by partial-pivot LU (what
//                  Eigen's dynamic-size inverse() does), T1 = S^-1 B'
//   k_riccati_step the truncation test max|K - Kprev| < 1e-5 (tiny_api.cpp:157) and, if it fails, Kprev <- K, P <- Pnew
// The iteration count is data-dependent, so the host enqueues iterations in chunks and reads a `done` word between chunks;
// every kernel of an iteration that is enqueued behind the converged one sees `done` and returns at once. The same two parity
// traps as k_precompute: rho is added to the already augmented diagonals a second time (:134-135), and the iteration that meets
// the test keeps ITS K and P (:157).
// Summation order: an MFMA adds its four products and then the accumulator, the reference (and k_precompute) one product at a
// time -- differences of an ulp per product that the contraction of the Riccati map does not amplify (caches against the
// oracle: 1e-9 in tests/test_large_systems_gpu.py, as for k_precompute).
#include <cstdlib>

#include "tinympc_device.h"

namespace tinympc {

typedef double double4_pl __attribute__((ext_vector_type(4)));

enum PlEpilogue { PL_NONE = 0, PL_X_MINUS = 1, PL_PLUS_DIAG = 2 };

// C (m x n) = op(A) (m x k) op(B) (k x n), column-major; TA: A is stored k x m; TB: B is stored n x k.
// EPI: PL_X_MINUS: C = X - AB (X like C); PL_PLUS_DIAG: C = AB + diag(dg) (m == n).
template <bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(256) k_dgemm_mfma(double *C, const double *A, const double *B, int m, int k, int n, const double *X, const double *dg,
                                                     const int *done) {
    if (done && *done) return;
    // ONE tile of C per workgroup, k split over its four wavefronts (round 4: at the sizes of a setup -- 36 tiles at nx = 96 -- a
    // product is a chain of L2 round trips, one per sixteen columns of k; four wavefronts walk a quarter of it each, and a grid of
    // `tiles` workgroups instead of tiles / 4 spreads over more of the chip) and summed in wavefront order through LDS.
    __shared__ double part[3][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_m = (m + 15) / 16;
    const int tile = blockIdx.x;
    const int i0 = (tile % tiles_m) * 16, j0 = (tile / tiles_m) * 16;
    const int li = lane & 15, kk = lane >> 4;
    const int ia = i0 + li, jb = j0 + li;  // this lane's row of op(A) / column of op(B)
    const bool ia_ok = ia < m, jb_ok = jb < n;
    double4_pl acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    auto ld_a = [&](int l) -> double { return (ia_ok && l < k) ? (TA ? A[l + (size_t)ia * k] : A[ia + (size_t)l * m]) : 0.0; };
    auto ld_b = [&](int l) -> double { return (jb_ok && l < k) ? (TB ? B[jb + (size_t)l * n] : B[l + (size_t)jb * k]) : 0.0; };
    // sixteen columns of k per trip: four loads each in flight, two accumulation chains (a dependent FP64 MFMA only issues when
    // its predecessor has left the pipe); acc0 takes blocks 0 and 2, acc1 blocks 1 and 3 -- added in block order at the end
    for (int l0 = 16 * wave; l0 < k; l0 += 64) {
        double a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = ld_a(l0 + 4 * q + kk);
            b[q] = ld_b(l0 + 4 * q + kk);
        }
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave - 1][lane][r] = acc0[r] + acc1[r];
    }
    __syncthreads();
    if (wave > 0) return;
    // result register r of a lane: row i0 + (lane >> 4) + 4 r, column j0 + (lane & 15)
    const int j = j0 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + kk + 4 * r;
        if (i < m && j < n) {
            double v = ((acc0[r] + acc1[r]) + part[0][lane][r]) + (part[1][lane][r] + part[2][lane][r]);
            if constexpr (EPI == PL_X_MINUS) v = X[i + (size_t)j * m] - v;
            if constexpr (EPI == PL_PLUS_DIAG) v = ((i == j) ? dg[i] : 0.0) + v;
            C[i + (size_t)j * m] = v;
        }
    }
}

template <bool TA, bool TB, int EPI>
static void dgemm(double *C, const double *A, const double *B, int m, int k, int n, const double *X, const double *dg, const int *done, hipStream_t st) {
    const int tiles = ((m + 15) / 16) * ((n + 15) / 16);
    hipLaunchKernelGGL((k_dgemm_mfma<TA, TB, EPI>), dim3(tiles), dim3(256), 0, st, C, A, B, m, k, n, X, dg, done);
}

// One workgroup: S = R1 + BtP B (nu x nu), Sinv by partial-pivot LU, T1 = Sinv B' (nu x nx). (B'P was a dgemm; nu <= 64 here.)
__global__ void __launch_bounds__(256) k_gain_solve(double *T1, double *Sinv, double *S, int *perm, const double *BtP, const double *B, const double *R1d, int nx,
                                                     int nu, const int *done) {
    if (done && *done) return;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < nu * nu; idx += 256) {  // S = BtP (nu x nx) * B (nx x nu) + diag(R1)
        const int i = idx % nu, j = idx / nu;
        double acc = 0.0;
        for (int l = 0; l < nx; ++l) acc += BtP[i + (size_t)l * nu] * B[l + (size_t)j * nx];
        S[idx] = ((i == j) ? R1d[i] : 0.0) + acc;
    }
    __syncthreads();
    // (partial-pivot LU and a solve against the identity: wg_lu_inverse of tinympc_kernels.hip, restated for this file)
    for (int i = tid; i < nu; i += 256) perm[i] = i;
    __syncthreads();
    for (int kq = 0; kq < nu; ++kq) {
        if (tid == 0) {
            int piv = kq;
            double best = fabs(S[kq + (size_t)kq * nu]);
            for (int i = kq + 1; i < nu; ++i) {
                const double a = fabs(S[i + (size_t)kq * nu]);
                if (a > best) {
                    best = a;
                    piv = i;
                }
            }
            perm[nu] = piv;
        }
        __syncthreads();
        const int piv = perm[nu];
        if (piv != kq) {
            for (int j = tid; j < nu; j += 256) {
                const double t = S[kq + (size_t)j * nu];
                S[kq + (size_t)j * nu] = S[piv + (size_t)j * nu];
                S[piv + (size_t)j * nu] = t;
            }
            if (tid == 0) {
                const int t = perm[kq];
                perm[kq] = perm[piv];
                perm[piv] = t;
            }
        }
        __syncthreads();
        const double pivot = S[kq + (size_t)kq * nu];
        for (int i = kq + 1 + tid; i < nu; i += 256) S[i + (size_t)kq * nu] /= pivot;
        __syncthreads();
        const int rem = nu - kq - 1;
        for (int idx = tid; idx < rem * rem; idx += 256) {
            const int i = kq + 1 + idx % rem, j = kq + 1 + idx / rem;
            S[i + (size_t)j * nu] -= S[i + (size_t)kq * nu] * S[kq + (size_t)j * nu];
        }
        __syncthreads();
    }
    for (int c = tid; c < nu; c += 256) {
        double *col = Sinv + (size_t)c * nu;
        for (int i = 0; i < nu; ++i) col[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < nu; ++i)
            for (int j = 0; j < i; ++j) col[i] -= S[i + (size_t)j * nu] * col[j];
        for (int i = nu - 1; i >= 0; --i) {
            for (int j = i + 1; j < nu; ++j) col[i] -= S[i + (size_t)j * nu] * col[j];
            col[i] /= S[i + (size_t)i * nu];
        }
    }
    __syncthreads();
    if (T1) {
        for (int idx = tid; idx < nu * nx; idx += 256) {  // T1 = Sinv (nu x nu) * B' (nu x nx)
            const int i = idx % nu, j = idx / nu;
            double acc = 0.0;
            for (int l = 0; l < nu; ++l) acc += Sinv[i + (size_t)l * nu] * B[j + (size_t)l * nx];
            T1[idx] = acc;
        }
    }
}

// The same inverse for nu <= 64 (round 4): ONE wavefront, S and its inverse in LDS, lane j = column j. k_gain_solve above keeps S in
// global memory and walks it with one or a few threads -- every pivot search, row swap and substitution step a chain of L2 round
// trips: 307 us per call at nu = 32, 129 calls = 40 of the 44 ms a setup at nx = 96 took (profiles/r03_large_system_stats.csv).
// Here the products around it are launches on the matrix cores like every other product of the iteration (S = B'P B + R1,
// T1 = S^-1 B') and the LU lives in LDS: partial pivoting (largest magnitude, lowest row on ties, as Eigen's PartialPivLU),
// the elimination column by column with the arithmetic of k_gain_solve in its order, then the solve against the permuted identity,
// one right-hand side per lane. Rows of the LDS copies are padded to an odd length: lanes walk columns AND rows without conflicts.
__global__ void __launch_bounds__(64) k_lu_inverse(double *Sinv, const double *S, int nu, const int *done) {
    if (done && *done) return;
    constexpr int LD = 65;
    __shared__ double sA[64 * LD];
    __shared__ double sX[64 * LD];
    __shared__ int sPerm[64];
    const int lane = threadIdx.x;
    const bool col = lane < nu;
    if (col)
        for (int i = 0; i < nu; ++i) sA[i * LD + lane] = S[i + (size_t)lane * nu];
    sPerm[lane] = lane;
    __syncthreads();
    for (int k = 0; k < nu; ++k) {
        // pivot of column k among rows k .. nu-1
        double v = (lane >= k && col) ? fabs(sA[lane * LD + k]) : -1.0;
        int idx = lane;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const double ov = __shfl_xor(v, m, 64);
            const int oi = __shfl_xor(idx, m, 64);
            if (ov > v || (ov == v && oi < idx)) {
                v = ov;
                idx = oi;
            }
        }
        const int piv = __builtin_amdgcn_readfirstlane(idx);
        if (piv != k) {
            if (col) {
                const double t = sA[k * LD + lane];
                sA[k * LD + lane] = sA[piv * LD + lane];
                sA[piv * LD + lane] = t;
            }
            if (lane == 0) {
                const int t = sPerm[k];
                sPerm[k] = sPerm[piv];
                sPerm[piv] = t;
            }
        }
        __syncthreads();
        const double pivot = sA[k * LD + k];
        if (lane > k && col) sA[lane * LD + k] /= pivot;
        __syncthreads();
        if (lane > k && col) {
            const double u = sA[k * LD + lane];
            for (int i = k + 1; i < nu; ++i) sA[i * LD + lane] -= sA[i * LD + k] * u;
        }
        __syncthreads();
    }
    if (col) {  // column `lane` of the inverse: L U x = P e_lane
        for (int i = 0; i < nu; ++i) {
            double x = (sPerm[i] == lane) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) x -= sA[i * LD + j] * sX[j * LD + lane];
            sX[i * LD + lane] = x;
        }
        for (int i = nu - 1; i >= 0; --i) {
            double x = sX[i * LD + lane];
            for (int j = i + 1; j < nu; ++j) x -= sA[i * LD + j] * sX[j * LD + lane];
            sX[i * LD + lane] = x / sA[i * LD + i];
        }
        for (int i = 0; i < nu; ++i) Sinv[i + (size_t)lane * nu] = sX[i * LD + lane];
    }
}

// The truncation test and the hand-over to the next iteration. ctl[0] = done, ctl[1] = steps taken.
// (the iteration index lives in ctl[2]: every iteration's launches have the same arguments, so a chunk of iterations is ONE graph)
__global__ void __launch_bounds__(1024) k_riccati_step(const double *K, double *Kprev, const double *Pnew, double *P, int nx, int nu, int *ctl) {
    if (ctl[0]) return;
    const int it = ctl[2];
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    double mx = 0.0;
    for (int i = tid; i < nu * nx; i += 1024) mx = fmax(mx, fabs(K[i] - Kprev[i]));
    red[tid] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (red[0] < 1e-5 || it == 999) {  // :157 (the 1000th iteration ends the loop whatever the test says, :152)
        if (tid == 0) {
            ctl[1] = red[0] < 1e-5 ? it + 1 : 1000;
            __threadfence();
            ctl[0] = 1;
        }
        return;  // K and Pnew are this solve's Kinf and Pinf
    }
    for (int i = tid; i < nu * nx; i += 1024) Kprev[i] = K[i];       // :164
    for (int i = tid; i < nx * nx; i += 1024) P[i] = Pnew[i];       // :165
    if (tid == 0) ctl[2] = it + 1;
}

__global__ void __launch_bounds__(256) k_pl_init(double *Q1d, double *R1d, double *Kprev, double *P, const double *Qd, const double *Rd, double rho, int nx, int nu,
                                                  int *ctl) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = g; i < (size_t)nx; i += stride) Q1d[i] = Qd[i] + rho;  // :134
    for (size_t i = g; i < (size_t)nu; i += stride) R1d[i] = Rd[i] + rho;  // :135
    for (size_t i = g; i < (size_t)nu * nx; i += stride) Kprev[i] = 0.0;
    for (size_t i = g; i < (size_t)nx * nx; i += stride) P[i] = (i % nx == i / nx) ? rho : 0.0;  // :148
    if (g == 0) {
        ctl[0] = 0;
        ctl[1] = 1000;
        ctl[2] = 0;
    }
}

__global__ void __launch_bounds__(256) k_pl_finish(PrecomputeParams p, const double *K, const double *P, const double *Sinv, const double *AmBK, const int *ctl) {
    const int nx = p.nx, nu = p.nu;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = g; i < (size_t)nu * nu; i += stride) p.Quu_inv[i] = Sinv[i];
    for (size_t i = g; i < (size_t)nu * nx; i += stride) p.Kinf[i] = K[i];
    for (size_t i = g; i < (size_t)nx * nx; i += stride) {
        p.Pinf[i] = P[i];
        p.AmBKt[(i / nx) + (size_t)(i % nx) * nx] = AmBK[i];  // :170
    }
    if (g == 0) p.info[0] = ctl[1];
}

// y (m) = op(M) x, one workgroup (the three affine-dynamics vectors at the end)
template <bool T>
__global__ void __launch_bounds__(256) k_pl_matvec(double *y, const double *M, const double *x, int m, int k) {
    for (int i = threadIdx.x; i < m; i += 256) {
        double acc = 0.0;
        for (int l = 0; l < k; ++l) acc += (T ? M[l + (size_t)i * k] : M[i + (size_t)l * m]) * x[l];
        y[i] = acc;
    }
}

size_t precompute_large_scratch_doubles(int nx, int nu) {
    // K, Kprev, BtP, T1, T2 (nu*nx) + P, Pnew, AtP, AmBK (nx*nx) + S, Sinv (nu*nu) + Q1d, R1d, Pf + perm (nu+1 ints) + ctl (3 ints)
    return (size_t)5 * nu * nx + (size_t)4 * nx * nx + (size_t)2 * nu * nu + 2 * nx + nu + (nu + 2) + 4;
}

hipError_t launch_precompute_large(const PrecomputeParams &p, hipStream_t st) {
    const int nx = p.nx, nu = p.nu;
    double *w = p.scratch;
    double *K = w;      w += (size_t)nu * nx;
    double *Kprev = w;  w += (size_t)nu * nx;
    double *BtP = w;    w += (size_t)nu * nx;
    double *T1 = w;     w += (size_t)nu * nx;
    double *T2 = w;     w += (size_t)nu * nx;
    double *P = w;      w += (size_t)nx * nx;
    double *Pnew = w;   w += (size_t)nx * nx;
    double *AtP = w;    w += (size_t)nx * nx;
    double *AmBK = w;   w += (size_t)nx * nx;
    double *S = w;      w += (size_t)nu * nu;
    double *Sinv = w;   w += (size_t)nu * nu;
    double *Q1d = w;    w += nx;
    double *R1d = w;    w += nu;
    double *Pf = w;     w += nx;
    int *perm = reinterpret_cast<int *>(w);  w += (nu + 2);  // nu + 1 ints (rounded up to doubles)
    int *ctl = reinterpret_cast<int *>(w);
    const double *A = p.A, *B = p.B;
    hipLaunchKernelGGL(k_pl_init, dim3(64), dim3(256), 0, st, Q1d, R1d, Kprev, P, p.Qd, p.Rd, p.rho, nx, nu, ctl);
    // Iterations are enqueued in chunks of CHUNK between two looks at `done`. (Captured into a HIP graph and replayed, a chunk was
    // no faster -- 17.1 against 16.2 ms at nx = 96: what an iteration costs is the ten DEPENDENT kernels' start-to-end latency on the
    // GPU, ~12 us each at these sizes, not the host's launch calls.)
    constexpr int CHUNK = 16;
    int host_ctl[2] = {0, 1000};
    for (int it = 0; it < 1000 && !host_ctl[0];) {
        for (int c = 0; c < CHUNK && it < 1000; ++c, ++it) {
            // :154  K = (R1 + B'PB)^-1 B' P A, evaluated left to right
            dgemm<true, false, PL_NONE>(BtP, B, P, nu, nx, nx, nullptr, nullptr, ctl, st);
            if (nu <= 64) {  // S = B'P B + R1 and T1 = S^-1 B' on the matrix cores, the inverse in LDS
                dgemm<false, false, PL_PLUS_DIAG>(S, BtP, B, nu, nx, nu, nullptr, R1d, ctl, st);
                hipLaunchKernelGGL(k_lu_inverse, dim3(1), dim3(64), 0, st, Sinv, S, nu, ctl);
                dgemm<false, true, PL_NONE>(T1, Sinv, B, nu, nu, nx, nullptr, nullptr, ctl, st);
            } else {
                hipLaunchKernelGGL(k_gain_solve, dim3(1), dim3(256), 0, st, T1, Sinv, S, perm, BtP, B, R1d, nx, nu, ctl);
            }
            dgemm<false, false, PL_NONE>(T2, T1, P, nu, nx, nx, nullptr, nullptr, ctl, st);
            dgemm<false, false, PL_NONE>(K, T2, A, nu, nx, nx, nullptr, nullptr, ctl, st);
            // :155  Pnew = Q1 + A'P (A - B K)
            dgemm<true, false, PL_NONE>(AtP, A, P, nx, nx, nx, nullptr, nullptr, ctl, st);
            dgemm<false, false, PL_X_MINUS>(AmBK, B, K, nx, nu, nx, A, nullptr, ctl, st);
            dgemm<false, false, PL_PLUS_DIAG>(Pnew, AtP, AmBK, nx, nx, nx, nullptr, Q1d, ctl, st);
            hipLaunchKernelGGL(k_riccati_step, dim3(1), dim3(1024), 0, st, K, Kprev, Pnew, P, nx, nu, ctl);  // :157, :164-165
        }
        hipError_t e = hipMemcpyAsync(host_ctl, ctl, sizeof(host_ctl), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    }
    // :169  Quu_inv = (R1 + B' Pinf B)^-1 ; :170  AmBKt = (A - B Kinf)'   (K, Pnew: the iteration that met the test)
    dgemm<true, false, PL_NONE>(BtP, B, Pnew, nu, nx, nx, nullptr, nullptr, nullptr, st);
    if (nu <= 64) {
        dgemm<false, false, PL_PLUS_DIAG>(S, BtP, B, nu, nx, nu, nullptr, R1d, nullptr, st);
        hipLaunchKernelGGL(k_lu_inverse, dim3(1), dim3(64), 0, st, Sinv, S, nu, (const int *)nullptr);
    } else {
        hipLaunchKernelGGL(k_gain_solve, dim3(1), dim3(256), 0, st, (double *)nullptr, Sinv, S, perm, BtP, B, R1d, nx, nu, (const int *)nullptr);
    }
    dgemm<false, false, PL_X_MINUS>(AmBK, B, K, nx, nu, nx, A, nullptr, nullptr, st);
    hipLaunchKernelGGL(k_pl_finish, dim3(64), dim3(256), 0, st, p, K, Pnew, Sinv, AmBK, ctl);
    // affine-dynamics terms (upstream TinyMPC main; PARITY UNPINNED): APf = AmBKt Pinf f, BPf = B' Pinf f
    hipLaunchKernelGGL(k_pl_matvec<false>, dim3(1), dim3(256), 0, st, Pf, Pnew, p.fdyn, nx, nx);
    hipLaunchKernelGGL(k_pl_matvec<true>, dim3(1), dim3(256), 0, st, p.APf, AmBK, Pf, nx, nx);
    hipLaunchKernelGGL(k_pl_matvec<true>, dim3(1), dim3(256), 0, st, p.BPf, B, Pf, nu, nx);
    return hipGetLastError();
}

}  // namespace tinympc
