// tinympc_precompute_rows.hip -- P1 (tiny_precompute_and_set_cache, reference tiny_api.cpp:124-190) for small systems, nx + nu <= 16:
// the whole Riccati fixed point in the REGISTERS of one wavefront, one matrix ROW per lane, every product a chain of fused
// `v_fmac_f64_dpp ... row_newbcast:k` instructions (the idiom of the solve kernels, tinympc_sweep.h). No LDS, no barrier, no
// memory access inside the loop.
//
// Why (round 5): the one-workgroup LDS kernel (k_precompute, tinympc_kernels.hip) spends ~40 barriers and ~10 LDS round trips per
// Riccati step -- 11 us per step, 0.6 ms for the quadrotor's 55 steps, 2.2 ms for the cartpole's 454, where the reference's
// whole tiny_setup takes 0.07 / 0.17 ms on one host core (profiles/r05_setup_time.txt). A step is a CHAIN of nine small dependent
// products; what bounds it on a GPU is the latency of each link, not arithmetic.
//
// Layout (one DPP row of 16 lanes; the wavefront's other three rows hold copies and compute the same):
//   lane r <  NXP        "x-lane": row r of A', A, B, P            XA[k] = A[k][r]   Ar[j] = A[r][j]   Br[j] = B[r][j]   P[j]
//   lane NXP + u, u<NUP  "u-lane": row u of B', R                  XA[k] = B[k][u]
// With lane k holding row k of Y, a product C = X Y is, per output column j, ONE chain over k:
//   C[r][j] = sum_k X[r][k] * Y[k][j]      acc_j += X_k(own lane) * (Y_j of lane k)      v_fmac_f64_dpp acc_j, Y_j, X_k row_newbcast:k
// and the result -- row r in lane r -- is in place both as the next product's left operand (own row) and as a right operand
// (broadcast from lane k). The x- and u-lanes run the SAME instructions on different rows, so [A'; B'] P is one product
// (the (nx+nu)^2 fusion of the solve kernels' sweeps), and the chain  S = B'P B + R1 -> S^-1 -> ((S^-1 B') P) A = K  lives in the
// u-lanes while  A - B K -> Q1 + A'P (A - B K)  lives in the x-lanes, K crossing over through the broadcast alone.
// Evaluation order = the reference's (:154 left to right, :155), summation index ascending, as in k_precompute and the oracle.
// The nu x nu inverse: Gauss-Jordan with partial pivoting on [S | I], rows in the u-lanes, the pivot row fetched with v_readlane.
//
// Shapes: instantiated for the BASELINE systems exactly -- (4,1) cartpole, (12,4) quadrotor, (6,3) rocket -- and for zero-padded
// classes (4,4), (8,4), (12,4) that take any nx <= NXP, nu <= NUP (the padding stays exactly zero through the recursion; padded
// rows of S carry a unit diagonal). Everything else (nu > 4, nx > 12 with nx + nu <= 64) stays on k_precompute.
#include "tinympc_device.h"

namespace tinympc {
namespace {

#define ROWS_FMA(i) "v_fmac_f64_dpp %[a], %[w], %[m" #i "] row_newbcast:%[l" #i "] row_mask:0xf bank_mask:0xf\n\t"

// acc += sum_{k < K} m[k] * (w of lane L0 + k of the DPP row), k ascending; blocks of at most four instructions (tinympc_sweep.h:
// separate blocks leave the scheduler room, one statement per instruction would cost boundary nops). `w` must not have been written
// by the two VALU instructions in front of the first block: the `s_nop 1` guarantees it whatever the scheduler does (the build's ISA
// lint checks every DPP FMA of the generated code, tools/isa_lint.py).
template <int L0, int K, int K0 = 0>
__device__ __forceinline__ void chain(double &a, double w, const double *m) {
    if constexpr (K0 < K) {
        constexpr int n = (K - K0) >= 4 ? 4 : (K - K0);
        if constexpr (n == 4) {
            if constexpr (K0 == 0)
                asm("s_nop 1\n\t" ROWS_FMA(0) ROWS_FMA(1) ROWS_FMA(2) ROWS_FMA(3)
                    : [a] "+&v"(a)
                    : [w] "v"(w), [m0] "v"(m[K0]), [m1] "v"(m[K0 + 1]), [m2] "v"(m[K0 + 2]), [m3] "v"(m[K0 + 3]), [l0] "n"(L0 + K0), [l1] "n"(L0 + K0 + 1),
                      [l2] "n"(L0 + K0 + 2), [l3] "n"(L0 + K0 + 3));
            else
                asm(ROWS_FMA(0) ROWS_FMA(1) ROWS_FMA(2) ROWS_FMA(3)
                    : [a] "+&v"(a)
                    : [w] "v"(w), [m0] "v"(m[K0]), [m1] "v"(m[K0 + 1]), [m2] "v"(m[K0 + 2]), [m3] "v"(m[K0 + 3]), [l0] "n"(L0 + K0), [l1] "n"(L0 + K0 + 1),
                      [l2] "n"(L0 + K0 + 2), [l3] "n"(L0 + K0 + 3));
        } else if constexpr (n == 3) {
            if constexpr (K0 == 0)
                asm("s_nop 1\n\t" ROWS_FMA(0) ROWS_FMA(1) ROWS_FMA(2)
                    : [a] "+&v"(a)
                    : [w] "v"(w), [m0] "v"(m[K0]), [m1] "v"(m[K0 + 1]), [m2] "v"(m[K0 + 2]), [l0] "n"(L0 + K0), [l1] "n"(L0 + K0 + 1), [l2] "n"(L0 + K0 + 2));
            else
                asm(ROWS_FMA(0) ROWS_FMA(1) ROWS_FMA(2)
                    : [a] "+&v"(a)
                    : [w] "v"(w), [m0] "v"(m[K0]), [m1] "v"(m[K0 + 1]), [m2] "v"(m[K0 + 2]), [l0] "n"(L0 + K0), [l1] "n"(L0 + K0 + 1), [l2] "n"(L0 + K0 + 2));
        } else if constexpr (n == 2) {
            if constexpr (K0 == 0)
                asm("s_nop 1\n\t" ROWS_FMA(0) ROWS_FMA(1) : [a] "+&v"(a) : [w] "v"(w), [m0] "v"(m[K0]), [m1] "v"(m[K0 + 1]), [l0] "n"(L0 + K0), [l1] "n"(L0 + K0 + 1));
            else
                asm(ROWS_FMA(0) ROWS_FMA(1) : [a] "+&v"(a) : [w] "v"(w), [m0] "v"(m[K0]), [m1] "v"(m[K0 + 1]), [l0] "n"(L0 + K0), [l1] "n"(L0 + K0 + 1));
        } else {
            if constexpr (K0 == 0)
                asm("s_nop 1\n\t" ROWS_FMA(0) : [a] "+&v"(a) : [w] "v"(w), [m0] "v"(m[K0]), [l0] "n"(L0 + K0));
            else
                asm(ROWS_FMA(0) : [a] "+&v"(a) : [w] "v"(w), [m0] "v"(m[K0]), [l0] "n"(L0 + K0));
        }
        chain<L0, K, K0 + n>(a, w, m);
    }
}

// The same for G = 2, 3, 4 accumulators at once, one k per statement: a LONE wavefront waits for every dependent FP64 FMA (a chain on
// one accumulator issues every ~8 cycles, not every 4); with the columns of a product interleaved the next instruction is always
// independent of the last three.
#define ROWS_FMAG(g) "v_fmac_f64_dpp %[a" #g "], %[w" #g "], %[m] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\t"
template <int L0, int K, int K0 = 0>
__device__ __forceinline__ void chain4(double &a0, double &a1, double &a2, double &a3, double w0, double w1, double w2, double w3, const double *m) {
    if constexpr (K0 < K) {
        if constexpr (K0 == 0)
            asm("s_nop 1\n\t" ROWS_FMAG(0) ROWS_FMAG(1) ROWS_FMAG(2) ROWS_FMAG(3)
                : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2), [a3] "+&v"(a3)
                : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [m] "v"(m[K0]), [l] "n"(L0 + K0));
        else
            asm(ROWS_FMAG(0) ROWS_FMAG(1) ROWS_FMAG(2) ROWS_FMAG(3)
                : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2), [a3] "+&v"(a3)
                : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [m] "v"(m[K0]), [l] "n"(L0 + K0));
        chain4<L0, K, K0 + 1>(a0, a1, a2, a3, w0, w1, w2, w3, m);
    }
}
template <int L0, int K, int K0 = 0>
__device__ __forceinline__ void chain3(double &a0, double &a1, double &a2, double w0, double w1, double w2, const double *m) {
    if constexpr (K0 < K) {
        if constexpr (K0 == 0)
            asm("s_nop 1\n\t" ROWS_FMAG(0) ROWS_FMAG(1) ROWS_FMAG(2)
                : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2)
                : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [m] "v"(m[K0]), [l] "n"(L0 + K0));
        else
            asm(ROWS_FMAG(0) ROWS_FMAG(1) ROWS_FMAG(2)
                : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2)
                : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [m] "v"(m[K0]), [l] "n"(L0 + K0));
        chain3<L0, K, K0 + 1>(a0, a1, a2, w0, w1, w2, m);
    }
}
template <int L0, int K, int K0 = 0>
__device__ __forceinline__ void chain2(double &a0, double &a1, double w0, double w1, const double *m) {
    if constexpr (K0 < K) {
        if constexpr (K0 == 0)
            asm("s_nop 1\n\t" ROWS_FMAG(0) ROWS_FMAG(1) : [a0] "+&v"(a0), [a1] "+&v"(a1) : [w0] "v"(w0), [w1] "v"(w1), [m] "v"(m[K0]), [l] "n"(L0 + K0));
        else
            asm(ROWS_FMAG(0) ROWS_FMAG(1) : [a0] "+&v"(a0), [a1] "+&v"(a1) : [w0] "v"(w0), [w1] "v"(w1), [m] "v"(m[K0]), [l] "n"(L0 + K0));
        chain2<L0, K, K0 + 1>(a0, a1, w0, w1, m);
    }
}

// C[j] = sum_{k < K} X[k] * (Y[j] of lane L0 + k), j < NC: the product of the own row X with the rows Y of lanes L0 .. L0 + K - 1,
// four columns at a time (each column's sum in the order k = 0, 1, ...: bit-identical to one chain per column)
template <int L0, int K, int NC, int J0 = 0>
__device__ __forceinline__ void rows_product(double (&C)[NC], const double *X, const double (&Y)[NC]) {
    if constexpr (J0 < NC) {
        constexpr int g = (NC - J0) >= 4 ? 4 : (NC - J0);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if constexpr (g == 4) chain4<L0, K>(a0, a1, a2, a3, Y[J0], Y[J0 + 1], Y[J0 + 2], Y[J0 + 3], X);
        else if constexpr (g == 3) chain3<L0, K>(a0, a1, a2, Y[J0], Y[J0 + 1], Y[J0 + 2], X);
        else if constexpr (g == 2) chain2<L0, K>(a0, a1, Y[J0], Y[J0 + 1], X);
        else chain<L0, K>(a0, Y[J0], X);
        C[J0] = a0;
        if constexpr (g > 1) C[J0 + 1] = a1;
        if constexpr (g > 2) C[J0 + 2] = a2;
        if constexpr (g > 3) C[J0 + 3] = a3;
        rows_product<L0, K, NC, J0 + g>(C, X, Y);
    }
}

// the value of lane `lane` (wave-uniform) in every lane
__device__ __forceinline__ double lane_value(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Sinv = S^-1 for the NUP x NUP matrix whose row u sits in lane NXP + u (`u` < 0: not a u-lane). Gauss-Jordan with partial
// pivoting on [S | I] (the pivot of column k: the largest |S[.][k]| among the rows not used yet, the first of equals -- the choice
// of the partial-pivot LU behind Eigen's inverse(), tiny_api.cpp:154); rows stay where they are, the pivot row travels by
// v_readlane, and at the end row k of the inverse is fetched from the lane that was column k's pivot. S is destroyed.
// 1 / x to within an ulp or two: v_rcp_f64 (about 2^-25 relative) and two Newton steps -- a third of the IEEE division's dependent
// chain, which sits in front of everything a pivot step does. (x: a pivot of S = R + 2 rho I + B'PB, far from the range ends.)
__device__ __forceinline__ double reciprocal(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

template <int NXP, int NUP>
__device__ __forceinline__ void rows_inverse(double (&Sinv)[NUP], double (&S)[NUP], int u) {
    if constexpr (NUP == 1) {
        Sinv[0] = reciprocal(S[0]);
        return;
    }
    double E[NUP];
#pragma unroll
    for (int j = 0; j < NUP; ++j) E[j] = (j == u) ? 1.0 : 0.0;
    bool used = false;
    int mine = 0;  // the pivot row of column `u`
#pragma unroll
    for (int k = 0; k < NUP; ++k) {
        const double cand = (u >= 0 && !used) ? fabs(S[k]) : -1.0;
        double v[NUP];
        double best = -1.0;
#pragma unroll
        for (int q = 0; q < NUP; ++q) {
            v[q] = lane_value(cand, NXP + q);
            best = fmax(best, v[q]);
        }
        int piv = NUP - 1;
#pragma unroll
        for (int q = NUP - 2; q >= 0; --q) piv = (v[q] == best) ? q : piv;  // the first of equals
        const int pl = NXP + __builtin_amdgcn_readfirstlane(piv);
        const double inv = reciprocal(lane_value(S[k], pl));
        const double g = -S[k] * inv;  // row r != pivot row:  row_r -= (S[r][k] / pivot) * pivot row
        const bool me = (u == piv);
#pragma unroll
        for (int j = k + 1; j < NUP; ++j) {
            const double pr = lane_value(S[j], pl);
            S[j] = me ? pr * inv : fma(g, pr, S[j]);
        }
        S[k] = me ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < NUP; ++j) {
            const double pr = lane_value(E[j], pl);
            E[j] = me ? pr * inv : fma(g, pr, E[j]);
        }
        used = used || me;
        if (u == k) mine = piv;
    }
    const int src = NXP + mine;  // (row 0 of the wavefront: all four DPP rows hold the same)
#pragma unroll
    for (int j = 0; j < NUP; ++j) Sinv[j] = __shfl(E[j], src, 64);
}

template <int NXP, int NUP>
__global__ void __launch_bounds__(64) k_precompute_rows(const PrecomputeParams p) {
    static_assert(NXP + NUP <= 16, "one DPP row");
    const int nx = p.nx, nu = p.nu;
    const int r = (int)threadIdx.x & 15;
    const bool xl = r < nx;                          // x-lane with a real row
    const int u = (r >= NXP && r < NXP + NUP) ? r - NXP : -1;  // u-lane (padded rows included)
    const bool ul = u >= 0 && u < nu;                // ... with a real row
    const double rho = p.rho;
    const double *A = p.A, *B = p.B;

    double XA[NXP], Ar[NXP], Br[NUP], P[NXP];
#pragma unroll
    for (int k = 0; k < NXP; ++k) {
        XA[k] = 0.0;
        if (k < nx) XA[k] = xl ? A[k + (size_t)r * nx] : (ul ? B[k + (size_t)u * nx] : 0.0);  // A'[r][k] | B'[u][k]
        Ar[k] = (xl && k < nx) ? A[r + (size_t)k * nx] : 0.0;
        P[k] = (xl && k == r) ? rho : 0.0;  // :148
    }
#pragma unroll
    for (int j = 0; j < NUP; ++j) Br[j] = (xl && j < nu) ? B[r + (size_t)j * nx] : 0.0;
    const double q1 = xl ? p.Qd[r] + rho : 0.0;                   // :134 (the diagonal handed over already carries rho once, :90)
    const double r1 = ul ? p.Rd[u] + rho : (u >= 0 ? 1.0 : 0.0);  // :135; padded rows of S: unit diagonal

    double M[NXP], S[NUP], Sinv[NUP], T1[NXP], T2[NXP], K[NXP], Kp[NXP], D[NXP], Pn[NXP];
#pragma unroll
    for (int j = 0; j < NXP; ++j) { Kp[j] = 0.0; K[j] = 0.0; D[j] = 0.0; Pn[j] = 0.0; }
    int steps = 1000;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();  // (diagnostics: info[1], info[2])
#pragma unroll 1
    for (int it = 0; it < 1000; ++it) {  // :152
        // :154  Kinf = (R1 + B'*P*B).inverse() * B' * P * A      (evaluated left to right)
        rows_product<0, NXP>(M, XA, P);                       // x-lanes: A'P   u-lanes: B'P
        rows_product<0, NXP>(S, M, Br);                       // u-lanes: B'P B
#pragma unroll
        for (int j = 0; j < NUP; ++j) S[j] = ((j == u) ? r1 : 0.0) + S[j];
        rows_inverse<NXP, NUP>(Sinv, S, u);
        rows_product<NXP, NUP>(T1, Sinv, XA);                 // u-lanes: S^-1 B'       (B' rows: lanes NXP ..)
        rows_product<0, NXP>(T2, T1, P);                      // u-lanes: (S^-1 B') P
        rows_product<0, NXP>(K, T2, Ar);                      // u-lanes: ((S^-1 B') P) A = Kinf
        // :155  Pinf = Q1 + A'*P*(A - B*Kinf)
        rows_product<NXP, NUP>(D, Br, K);                     // x-lanes: B Kinf        (Kinf rows: lanes NXP ..)
#pragma unroll
        for (int j = 0; j < NXP; ++j) D[j] = Ar[j] - D[j];
        rows_product<0, NXP>(Pn, M, D);                       // x-lanes: A'P (A - B Kinf)
#pragma unroll
        for (int j = 0; j < NXP; ++j) Pn[j] = ((j == r) ? q1 : 0.0) + Pn[j];
        // :157  max|Kinf - Ktp1| < 1e-5 -> stop, keeping THIS iteration's Kinf and Pinf
        double dm = 0.0;
#pragma unroll
        for (int j = 0; j < NXP; ++j) dm = fmax(dm, fabs(K[j] - Kp[j]));
        if (!ul) dm = 0.0;
        double md = 0.0;
#pragma unroll
        for (int q = 0; q < NUP; ++q) md = fmax(md, lane_value(dm, NXP + q));
        if (md < 1e-5) {
            steps = it + 1;
            break;
        }
#pragma unroll
        for (int j = 0; j < NXP; ++j) { Kp[j] = K[j]; P[j] = Pn[j]; }  // :164-165
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
    // :169  Quu_inv = (R1 + B'*Pinf*B).inverse()
    rows_product<0, NXP>(M, XA, Pn);
    rows_product<0, NXP>(S, M, Br);
#pragma unroll
    for (int j = 0; j < NUP; ++j) S[j] = ((j == u) ? r1 : 0.0) + S[j];
    rows_inverse<NXP, NUP>(Sinv, S, u);
    // :170  AmBKt = (A - B*Kinf)': D holds A - B Kinf of the last iteration = of the final Kinf
    const bool row0 = threadIdx.x < 16;
    if (row0 && ul) {
#pragma unroll
        for (int j = 0; j < NUP; ++j)
            if (j < nu) p.Quu_inv[u + (size_t)j * nu] = Sinv[j];
#pragma unroll
        for (int j = 0; j < NXP; ++j)
            if (j < nx) p.Kinf[u + (size_t)j * nu] = K[j];
    }
    if (row0 && xl) {
#pragma unroll
        for (int j = 0; j < NXP; ++j)
            if (j < nx) {
                p.Pinf[r + (size_t)j * nx] = Pn[j];
                p.AmBKt[j + (size_t)r * nx] = D[j];
            }
    }
    // Affine-dynamics terms (upstream TinyMPC main; PARITY UNPINNED): Pf = Pinf f, APf = (A - B Kinf)' Pf, BPf = B' Pf -- the sums
    // of k_precompute, in its order: Pf in the own lane, then its entries as the multipliers of broadcast columns
    double Pf = 0.0;
#pragma unroll
    for (int l = 0; l < NXP; ++l)
        if (l < nx) Pf = fma(Pn[l], p.fdyn[l], Pf);
    double PfB[NXP];
#pragma unroll
    for (int l = 0; l < NXP; ++l) PfB[l] = lane_value(Pf, l);
    double APf[NXP], BPf[NUP];
    rows_product<0, NXP>(APf, PfB, D);   // APf[i] = sum_l (A - B Kinf)[l][i] Pf[l]   (every lane: the whole vector)
    rows_product<0, NXP>(BPf, PfB, Br);  // BPf[i] = sum_l B[l][i] Pf[l]
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NXP; ++j)
            if (j < nx) p.APf[j] = APf[j];
#pragma unroll
        for (int j = 0; j < NUP; ++j)
            if (j < nu) p.BPf[j] = BPf[j];
        p.info[0] = steps;
        p.info[1] = (int)(c1 - c0);  // shader clocks of the Riccati loop
        p.info[2] = (int)(t1 - t0);  // ... and its duration in ticks of the 100 MHz counter
    }
}

}  // namespace

bool precompute_rows_supported(int nx, int nu) { return nx >= 1 && nu >= 1 && nx <= 12 && nu <= 4; }

// The exact shape where it is instantiated, else the smallest zero-padded class that holds it.
hipError_t launch_precompute_rows(const PrecomputeParams &p, hipStream_t stream) {
#define ROWS_LAUNCH(NXP, NUP) hipLaunchKernelGGL((k_precompute_rows<NXP, NUP>), dim3(1), dim3(64), 0, stream, p)
    const int nx = p.nx, nu = p.nu;
    if (nx == 4 && nu == 1) ROWS_LAUNCH(4, 1);
    else if (nx == 6 && nu == 3) ROWS_LAUNCH(6, 3);
    else if (nx <= 4) ROWS_LAUNCH(4, 4);
    else if (nx <= 8) ROWS_LAUNCH(8, 4);
    else ROWS_LAUNCH(12, 4);
#undef ROWS_LAUNCH
    return hipGetLastError();
}

}  // namespace tinympc
