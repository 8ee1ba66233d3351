// tinympc_solve_m.hip -- k_admm_solve_m ("layout M"): LARGE systems, 64 < nx+nu <= 512, on the FP64 matrix cores.
//
// Same algorithm as the other solve kernels (tinympc_solve.hip has the reference citations: M1 solve admm.cpp:109-207 = F1
// :25-35, S1 :43-59, D1 :65-69, L1 :75-83, R1 :89-107, C1 :196-197, B1 :13-20). Up to 64 rows an instance fits the lanes of
// a wavefront and the sweep step is a mat-vec on the VALU with the operator row in registers (layouts A-D); beyond that a lane
// would need several operand entries and an operator row no register file holds. This is the regime BASELINE.json's
// north_star reserves MFMA for: the step of SIXTEEN instances at once is a GEMM,
//     [x_{i+1}; u_i] (nxu x 16) = Mf (nxu x nxu) * [x_i; d_i] (nxu x 16) + cf,
// on v_mfma_f64_16x16x4_f64 tiles -- FP64 MFMA has the VALU's peak on MI355X (tools/microbench_mfma_f64.hip), but it shares
// the operator tile between 16 instances and needs no per-column operand broadcast.
//
// Up to 128 rows: one workgroup = 8 wavefronts = one tile of 16 instances. Wavefront w owns the 16-row output tile t = w (R =
// ceil(nxu / 16) <= 8 tiles; wavefronts beyond R only take part in the barriers) and keeps its operator tiles register-resident for a whole
// sweep: A[kb] = the 16 x 4 block (rows 16w.., columns 4kb..), one double per lane, 4R <= 32 doubles. (The first version gave
// a wavefront two row tiles and a workgroup four wavefronts: 408 VGPRs, one wavefront per SIMD, and every wait for state or
// for the exchange left its SIMD idle -- 18 M iterations/s at nx=96 against 22.5 M now. Half the rows per wavefront halve
// every per-lane array; two wavefronts share a SIMD.) The MFMA layouts (MI355X_MICROARCH.md) make the
// data flow closed: a result register D_t[reg] holds out[16t + (lane>>4) + 4 reg][instance lane&15], and the B operand of
// k-block kb = 4t + reg wants x[4kb + (lane>>4)][instance lane&15] -- the same lane, the same value. So the operand vector
// of the next step is the result of this one, exchanged between the wavefronts of the tile through a double-buffered LDS array
// Xb[kb][lane] behind ONE barrier per step; nothing is ever transposed.
// Everything row-local (slack projection, dual ascent, residual maxima, linear cost) happens on the result registers, 4
// (row, instance) entries per lane. The ADMM state does not fit on chip at these sizes (16 instances x 128 rows x N knots x
// (g, v) = 650 KB at N = 20) and streams through HBM once per sweep, in the tile's own layout
//     G, V, V2, D : [tile][knot][4R (= t, reg)][64 lanes]      (512-byte lines)
// which only this kernel reads (these sizes run on no other layout). The slack is a ping-pong pair V / V2 by iteration
// parity: a converged solve must keep the PREVIOUS iterate (admm.cpp:181-197), which is then simply the buffer last read.
// Arithmetic intensity ~ nxu / 20 flop per byte of state: HBM-bound below nxu ~ 128, at a few tenths of the FP64 peak.
#include "tinympc_device.h"
#include "tinympc_sweep.h"  // soc_project_element, halfspace_project_element

#ifndef TINY_EXP_M
#define TINY_EXP_M 0  // timing experiments (tools/build_m_variants.sh): 1 = no state traffic in the sweeps, 2 = no MFMAs
#endif

namespace tinympc {

typedef double double4_m __attribute__((ext_vector_type(4)));

// The barrier of a sweep step: only the LDS operand exchange crosses it (a lane's HBM state is read back by that lane alone),
// so it must not wait for the step's global stores the way __syncthreads()' fence does -- that wait sat in front of every
// GEMM.
__device__ __forceinline__ void lds_exchange_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int M_INST = 16;       // instances per tile (the N dimension of the MFMA)
// wavefronts per workgroup: eight up to 128 rows; R of them for 9 <= R <= 16 row tiles (one tile each: no wavefront carries two
// while its neighbours carry one), sixteen beyond
__host__ __device__ constexpr int m_waves(int R) {
#ifdef TINY_M_EIGHT_WAVES
    return 8;
#else
    return R <= 8 ? 8 : (R <= 16 ? R : 16);
#endif
}

// doubles per (tile, knot) of a state array
__host__ __device__ constexpr size_t m_knot_doubles(int R) { return (size_t)4 * R * 64; }
size_t solve_m_state_doubles(int nx, int nu, int N, int tiles) {
    const int R = (nx + nu + 15) / 16;
    return (size_t)tiles * N * m_knot_doubles(R);
}
bool solve_m_supported(int nx, int nu) { return nx + nu > 64 && nx + nu <= 512 && nx >= 1 && nu >= 1; }
// geometry (W = KT) of the operators and tables these sizes are built with
__host__ __device__ constexpr int m_geometry(int R) { return R > 16 ? 512 : R > 8 ? 256 : 128; }
int solve_m_geometry(int nx, int nu) { return m_geometry((nx + nu + 15) / 16); }

// CT: bounds and references are the same at every knot (p.const_tables): they are served from an LDS copy instead of
// the L2-resident per-knot tables -- 24 L2 round trips less behind every GEMM. (A compile-time switch: as a run-time one it
// pushed the kernel over its register file. Likewise, requesting the state a whole step ahead instead of right before the
// step's own GEMM, and interleaving two instance tiles per workgroup over shared operator tiles, both cost more in spills
// than they hid in latency; round 3's software-pipelined form is in tools/experiments, its measurements in
// profiles/r03_large_m_experiments.txt.)
//
// R <= 8 (nx+nu <= 128): one row tile per wavefront, its operator tiles register-resident (above).
// R = 9..32 (nx+nu <= 512, round 3): the operator tiles are STREAMED from L2 a batch ahead of the matrix instructions that consume
// them -- 4R doubles per lane and row tile no longer fit next to anything else -- from a tile-major copy of the operators
// (k_tile_operators_m: one 512-byte line per load). Up to sixteen row tiles a workgroup has R wavefronts with ONE tile each
// (nine tiles on eight wavefronts left seven of them waiting for the one that carried two); beyond, sixteen wavefronts carry two
// tiles each, handled one after the other inside a step -- state in, GEMM, row-local phase, state out, twice, then the one
// barrier. Sixteen wavefronts are four per SIMD: 128 registers, which the streamed form meets (123-128) because its tile loops
// are never unrolled and never start from a constant the compiler can see (see the body). A step reads each operator tile once
// per workgroup (<= 2 MB per operator, L2-resident) while the matrix pipe works 4R x 64 cycles per tile on it.
//
// FAM (round 4): the second-order-cone and linear-inequality families (bindings.cpp:408-478; the restated algorithm of
// oracle/, tinympc_solve_fam.hip has the bookkeeping) at these sizes. A cone or a linear row couples rows of ONE knot that may sit
// in any row tile, i.e. in any wavefront's result registers -- so the families are not part of the sweep step: the forward sweep
// also leaves its rollout x | u in HBM (p.scratch, the tile's own layout), and between the sweeps wavefront w takes knots
// w, w + NW, ... of the tile's sixteen instances: the four lanes (kq = 0..3) of an instance share the knot's rows exactly as the
// state layout stores them -- every access of the phase is a full 512-byte line --, a cone's ||w||^2 and a row's a_k' s are
// per-lane partial sums + two cross-lane adds. Duals gc | gl persist in the tile layout (p.GC, p.GL); the linear-cost term goes
// to p.LX, which the backward sweep reads next to V and G. Cones are projected one after another in list order (overlapping
// cones need no rounds here), linear rows likewise. The description is compact (m_fam_*: counts, cones as first / last / slope,
// one coefficient vector per linear row) -- the mask matrices of the other layouts would be 6 MB at 512 rows.
__host__ __device__ constexpr size_t m_fam_cone_offset() { return 8; }                                   // [c][first, last, mu]
__host__ __device__ constexpr size_t m_fam_lin_offset() { return 8 + (size_t)3 * HARD_MAX_CONES; }      // [k][GW coefficients | b_x, b_u, 1/|a_x|^2, 1/|a_u|^2]
__host__ __device__ constexpr size_t m_fam_lin_stride(int GW) { return (size_t)GW + 4; }
// Up to M_FAM_FAST_ROWS linear rows per side (header entry 4 says so) the description also carries the rows' Gram matrices, state side
// then input side, [k][j] = a_k' a_j, behind the rows: the phase then needs ONE pass for all dot products (m_families_fast).
constexpr int M_FAM_FAST_ROWS = 8;
size_t solve_m_fam_doubles(int nx, int nu, int nl) {
    return m_fam_lin_offset() + (size_t)nl * m_fam_lin_stride(solve_m_geometry(nx, nu)) + (nl <= M_FAM_FAST_ROWS ? (size_t)2 * nl * nl : 0);
}
int solve_m_fam_fast_rows() { return M_FAM_FAST_ROWS; }
size_t solve_m_fam_cone_offset() { return m_fam_cone_offset(); }
size_t solve_m_fam_lin_offset() { return m_fam_lin_offset(); }

// The families of one iterate for the tile's instances (see FAM above). X: rollout x | u in, garbage out; GCa / GLa: duals in and
// out; LXa: out. Called by every wavefront of the workgroup between two __syncthreads(); nothing in here crosses wavefronts, and
// lanes only communicate through the two cross-lane adds of a reduction (program order covers a lane's own stores and loads).
// Per knot:   A   s_c = x + gc -> X and GC (GC then holds vcnew while the cones work on it), s_l = x + gl -> GL and LX (LX: vlnew)
//             B   cones in list order on GC: ||w||^2, t, then the rows (soc_project_element);   C   linear rows in order on LX
//             Z   gc <- s_c - vcnew, gl <- s_l - vlnew, LX <- -rho (vcnew - gc) - rho (vlnew - gl)   (admm.cpp:65-69, :77-80 with the
//                 families' terms as the oracle restates them)
template <int R, int NW>
__device__ __forceinline__ void m_families(const SolveParams &p, double *X, double *GCa, double *GLa, double *LXa, int wv, int lane, bool active) {
    constexpr int GW = m_geometry(R), NS = 4 * R;
    const size_t KD = m_knot_doubles(R);
    const double *F = p.fam;
    const int nx = p.nx, nxu = p.nx + p.nu, N = p.N;
    const int ncx = __builtin_amdgcn_readfirstlane((int)F[0]), ncu = __builtin_amdgcn_readfirstlane((int)F[1]);
    const int nlx = __builtin_amdgcn_readfirstlane((int)F[2]), nlu = __builtin_amdgcn_readfirstlane((int)F[3]);
    const int nc = ncx + ncu, nl = nlx > nlu ? nlx : nlu;
    const int kq = lane >> 4;
    const double rho = p.rho;
    auto sum4 = [](double v) -> double {  // over the four lanes of an instance; the same bits in all four (a + b = b + a)
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
    for (int kn = wv; kn < N; kn += NW) {
        double *x = X + (size_t)kn * KD, *gc = GCa + (size_t)kn * KD, *gl = GLa + (size_t)kn * KD, *lx = LXa + (size_t)kn * KD;
        const bool has_u = kn < N - 1;  // (uniform) input rows: knots 0 .. N-2
        auto row_at = [&](int sl) -> int { return 16 * (sl >> 2) + kq + 4 * (sl & 3); };
        auto cone_on = [&](int r) -> bool { return r < nx ? ncx > 0 : (r < nxu && has_u && ncu > 0); };
        auto lin_on = [&](int r) -> bool { return r < nx ? nlx > 0 : (r < nxu && has_u && nlu > 0); };
        // ---- A
#pragma unroll 4
        for (int sl = 0; sl < NS; ++sl) {
            const int r = row_at(sl);
            const unsigned o = (unsigned)(sl * 64 + lane);
            const double xv = x[o], gcv = nc > 0 ? gc[o] : 0.0, glv = nl > 0 ? gl[o] : 0.0;  // (uniform: a family nobody uses costs no traffic)
            if (active && cone_on(r)) {
                const double sc = xv + gcv;
                x[o] = sc;
                gc[o] = sc;
            }
            if (active && lin_on(r)) {
                const double s0 = xv + glv;
                gl[o] = s0;
                lx[o] = s0;
            }
        }
        // ---- B: cones one after another (project_soc per cone in list order; state cones, then input cones)
        for (int c = 0; c < nc; ++c) {
            const double *cd = F + m_fam_cone_offset() + 3 * c;
            const int first = __builtin_amdgcn_readfirstlane((int)cd[0]), last = __builtin_amdgcn_readfirstlane((int)cd[1]);
            const double mu = cd[2];
            if (first >= nx && !has_u) continue;
            const double inv_mu = (mu != 0.0) ? 1.0 / mu : 0.0;
            const int s_first = 4 * (first >> 4) + ((first & 15) >> 2), s_last = 4 * (last >> 4) + ((last & 15) >> 2);
            double a2 = 0.0, t = 0.0;
            for (int sl = s_first; sl <= s_last; ++sl) {
                const int r = row_at(sl);
                const double v = gc[(unsigned)(sl * 64 + lane)];
                a2 += (r >= first && r < last) ? v * v : 0.0;
                t += (r == last) ? v : 0.0;
            }
            a2 = sum4(a2);
            t = sum4(t);
            for (int sl = s_first; sl <= s_last; ++sl) {
                const int r = row_at(sl);
                const unsigned o = (unsigned)(sl * 64 + lane);
                const double v = gc[o];
                if (active && r >= first && r <= last) gc[o] = soc_project_element(v, a2, t, mu, inv_mu, r == last ? 2 : 1);
            }
        }
        // ---- C: half-spaces one after another (row k of the state side and row k of the input side in one pass: disjoint rows)
        for (int k = 0; k < nl; ++k) {
            const double *ak = F + m_fam_lin_offset() + (size_t)k * m_fam_lin_stride(GW);
            const double bx = ak[GW], bu = ak[GW + 1], inx = ak[GW + 2], inu = ak[GW + 3];
            double dx = 0.0, du = 0.0;
#pragma unroll 4
            for (int sl = 0; sl < NS; ++sl) {
                const int r = row_at(sl);
                const double a = ak[r];
                const double v = lx[(unsigned)(sl * 64 + lane)];
                const double prod = lin_on(r) ? a * v : 0.0;
                dx += (r < nx) ? prod : 0.0;
                du += (r < nx) ? 0.0 : prod;
            }
            dx = sum4(dx);
            du = sum4(du);
            const bool viol = (dx > bx) || (du > bu);
            if (__ballot(viol && active) == 0ull) continue;  // (uniform) nothing to move
#pragma unroll 4
            for (int sl = 0; sl < NS; ++sl) {
                const int r = row_at(sl);
                const unsigned o = (unsigned)(sl * 64 + lane);
                const double a = ak[r];
                const double v = lx[o];
                const bool sx = r < nx;
                if (active && lin_on(r)) lx[o] = halfspace_project_element(v, sx ? dx : du, a, sx ? bx : bu, sx ? inx : inu);
            }
        }
        // ---- Z
#pragma unroll 4
        for (int sl = 0; sl < NS; ++sl) {
            const int r = row_at(sl);
            const unsigned o = (unsigned)(sl * 64 + lane);
            double sc = 0.0, vc = 0.0, s0 = 0.0, vl = 0.0;
            if (nc > 0) {
                sc = x[o];
                vc = gc[o];
            }
            if (nl > 0) {
                s0 = gl[o];
                vl = lx[o];
            }
            const bool real = r < nx || (r < nxu && has_u);
            double l = 0.0;
            if (cone_on(r)) {
                const double gcn = sc - vc;  // gc + x - vcnew
                l -= rho * (vc - gcn);
                if (active) gc[o] = gcn;
            }
            if (lin_on(r)) {
                const double gln = s0 - vl;
                l -= rho * (vl - gln);
                if (active) gl[o] = gln;
            }
            if (active && real) lx[o] = l;
        }
    }
}

// The same phase with fewer bytes (the sweeps and this phase are HBM-bound: what counts is how often an array is walked). Cones touch
// few rows: only the slots that hold a cone row (`cmask`, wave-uniform) get a temporary (in LX) -- every other row's projected value IS
// its x + gc. Linear rows (up to M_FAM_FAST_ROWS per side): the dot products of ALL rows with the unprojected s = x + gl in one pass;
// what the earlier projections of the sequence change in row k's product follows from the rows' Gram matrix,
//     a_k' (s - sum_{j<k} dist_j a_j) = a_k' s - sum_{j<k} dist_j (a_k' a_j),
// a scalar recurrence per instance; the projected vector is rebuilt in the last pass, s -> s - dist_k a_k in the order of the rows
// (the reference's own sequence of updates). Per knot: x, gl once (dots), x, gc, gl once more and gc, gl, LX out -- 5 reads + 3 writes
// against 9-13 + 7 of m_families. Same results up to the rounding of the products (parity tests: 1e-9, iteration counts exact).
template <int R, int NW>
__device__ __forceinline__ void m_families_fast(const SolveParams &p, double *X, double *GCa, double *GLa, double *LXa, int wv, int lane, bool active,
                                                unsigned long long cm_lo, unsigned long long cm_hi) {
    constexpr int GW = m_geometry(R), NS = 4 * R, KM = M_FAM_FAST_ROWS;
    const size_t KD = m_knot_doubles(R);
    const double *F = p.fam;
    const int nx = p.nx, nxu = p.nx + p.nu, N = p.N;
    const int ncx = __builtin_amdgcn_readfirstlane((int)F[0]), ncu = __builtin_amdgcn_readfirstlane((int)F[1]);
    const int nlx = __builtin_amdgcn_readfirstlane((int)F[2]), nlu = __builtin_amdgcn_readfirstlane((int)F[3]);
    const int nc = ncx + ncu, nl = nlx > nlu ? nlx : nlu;
    const int kq = lane >> 4;
    const double rho = p.rho;
    const double *const rows = F + m_fam_lin_offset();
    const double *const Gx = rows + (size_t)nl * m_fam_lin_stride(GW), *const Gu = Gx + (size_t)nl * nl;
    auto sum4 = [](double v) -> double {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
    for (int kn = wv; kn < N; kn += NW) {
        double *x = X + (size_t)kn * KD, *gc = GCa + (size_t)kn * KD, *gl = GLa + (size_t)kn * KD, *lx = LXa + (size_t)kn * KD;
        const bool has_u = kn < N - 1;
        auto row_at = [&](int sl) -> int { return 16 * (sl >> 2) + kq + 4 * (sl & 3); };
        auto cone_on = [&](int r) -> bool { return r < nx ? ncx > 0 : (r < nxu && has_u && ncu > 0); };
        auto lin_on = [&](int r) -> bool { return r < nx ? nlx > 0 : (r < nxu && has_u && nlu > 0); };
        auto cone_slot = [&](int sl) -> bool { return (((sl < 64) ? (cm_lo >> sl) : (cm_hi >> (sl - 64))) & 1ull) != 0ull; };  // (uniform)
        // ---- 1: the rows' dot products with s = x + gl; the cone slots' temporaries
        double dx[KM], du[KM];  // a_k' s per side; from the recurrence on: dist_k
#pragma unroll
        for (int k = 0; k < KM; ++k) dx[k] = du[k] = 0.0;
        constexpr int U = 4;  // slots whose loads are in flight together (see pass 4)
        static_assert(NS % U == 0, "slots come in fours");
        for (int s0 = 0; s0 < NS; s0 += U) {
            bool any = nl > 0;
#pragma unroll
            for (int u = 0; u < U; ++u) any = any || (nc > 0 && cone_slot(s0 + u));
            if (!any) continue;  // (uniform)
            double xv[U], gcv[U], glv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned o = (unsigned)((s0 + u) * 64 + lane);
                const bool cs = nc > 0 && cone_slot(s0 + u);
                xv[u] = (nl > 0 || cs) ? x[o] : 0.0;
                gcv[u] = cs ? gc[o] : 0.0;
                glv[u] = nl > 0 ? gl[o] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sl = s0 + u, r = row_at(sl);
                const unsigned o = (unsigned)(sl * 64 + lane);
                const bool cs = nc > 0 && cone_slot(sl);
                if (cs && active && cone_on(r)) lx[o] = xv[u] + gcv[u];
                if (nl > 0) {
                    const double sv0 = lin_on(r) ? xv[u] + glv[u] : 0.0;
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        if (k < nl) {  // (uniform)
                            const double prod = rows[(size_t)k * m_fam_lin_stride(GW) + r] * sv0;
                            dx[k] += (r < nx) ? prod : 0.0;
                            du[k] += (r < nx) ? 0.0 : prod;
                        }
                    }
                }
            }
        }
        // ---- 2: the sequence of half-spaces as a scalar recurrence (project_halfspaces: row k sees what rows 0 .. k-1 moved)
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k < nl) {
                const double *rk = rows + (size_t)k * m_fam_lin_stride(GW);
                double dotx = sum4(dx[k]), dotu = sum4(du[k]);
#pragma unroll
                for (int j = 0; j < KM; ++j) {
                    if (j < k) {
                        dotx = fma(-dx[j], Gx[(size_t)k * nl + j], dotx);
                        dotu = fma(-du[j], Gu[(size_t)k * nl + j], dotu);
                    }
                }
                dx[k] = dotx > rk[GW] ? (dotx - rk[GW]) * rk[GW + 2] : 0.0;      // dist_k, state side (0: not violated, nothing moves)
                du[k] = dotu > rk[GW + 1] ? (dotu - rk[GW + 1]) * rk[GW + 3] : 0.0;
            }
        }
        // ---- 3: cones one after another on the temporaries
        for (int c = 0; c < nc; ++c) {
            const double *cd = F + m_fam_cone_offset() + 3 * c;
            const int first = __builtin_amdgcn_readfirstlane((int)cd[0]), last = __builtin_amdgcn_readfirstlane((int)cd[1]);
            const double mu = cd[2];
            if (first >= nx && !has_u) continue;
            const double inv_mu = (mu != 0.0) ? 1.0 / mu : 0.0;
            const int s_first = 4 * (first >> 4) + ((first & 15) >> 2), s_last = 4 * (last >> 4) + ((last & 15) >> 2);
            double a2 = 0.0, t = 0.0;
            for (int sl = s_first; sl <= s_last; ++sl) {
                const int r = row_at(sl);
                const double v = lx[(unsigned)(sl * 64 + lane)];
                a2 += (r >= first && r < last) ? v * v : 0.0;
                t += (r == last) ? v : 0.0;
            }
            a2 = sum4(a2);
            t = sum4(t);
            for (int sl = s_first; sl <= s_last; ++sl) {
                const int r = row_at(sl);
                const unsigned o = (unsigned)(sl * 64 + lane);
                const double v = lx[o];
                if (active && r >= first && r <= last) lx[o] = soc_project_element(v, a2, t, mu, inv_mu, r == last ? 2 : 1);
            }
        }
        // ---- 4: duals and the linear-cost term. U slots' loads in flight before the first of their stores: the arrays may alias as far
        // as the compiler knows (a load behind a store waits for it), and with one slot in flight per wavefront the phase ran at a
        // third of the sweeps' bandwidth.
        for (int s0 = 0; s0 < NS; s0 += U) {
            double xv[U], gcv[U], glv[U], tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned o = (unsigned)((s0 + u) * 64 + lane);
                const bool cs = nc > 0 && cone_slot(s0 + u);
                xv[u] = x[o];
                gcv[u] = nc > 0 ? gc[o] : 0.0;
                glv[u] = nl > 0 ? gl[o] : 0.0;
                tv[u] = cs ? lx[o] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sl = s0 + u;
                const bool cs = nc > 0 && cone_slot(sl);
                const int r = row_at(sl);
                const unsigned o = (unsigned)(sl * 64 + lane);
                const bool real = r < nx || (r < nxu && has_u);
                double l = 0.0;
                if (cone_on(r)) {
                    const double sc = xv[u] + gcv[u];
                    const double vc = cs ? tv[u] : sc;  // (a row outside every cone: its projected value is s itself)
                    const double gcn = sc - vc;
                    l -= rho * (vc - gcn);
                    if (active) gc[o] = gcn;
                }
                if (lin_on(r)) {
                    const double sv0 = xv[u] + glv[u];
                    double vl = sv0;
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        if (k < nl) vl = fma(-((r < nx) ? dx[k] : du[k]), rows[(size_t)k * m_fam_lin_stride(GW) + r], vl);
                    }
                    const double gln = sv0 - vl;
                    l -= rho * (vl - gln);
                    if (active) gl[o] = gln;
                }
                if (active && real) lx[o] = l;
            }
        }
    }
}

template <int R, bool CT, bool FAM = false>
__global__ void __launch_bounds__(64 * m_waves(R)) k_admm_solve_m(const SolveParams p) {
    constexpr int KB = 4 * R;            // k-blocks of 4 operand rows (columns beyond nxu are zero in the operator)
    constexpr int NW = m_waves(R);                 // wavefronts per workgroup
    constexpr int TPW = (R + NW - 1) / NW;          // row tiles per wavefront (1 or 2)
    constexpr bool STREAM = R > 8;       // operator tiles from L2 instead of registers
    constexpr int GW = m_geometry(R);    // ops / tables geometry of these sizes: W = KT
    __shared__ __attribute__((aligned(16))) double sX[2][KB][64];  // operand vector of the step, double-buffered
    __shared__ unsigned sFlag[2][NW];                          // per-wave "instance still below tolerance" masks
    __shared__ double sTab[CT ? 3 : 1][GW];                         // CT: lo | hi | linref of every row
    __shared__ double sC[STREAM ? 2 : 1][STREAM ? GW : 1];          // two row tiles per wavefront: the sweep constants cf | cb
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (an SGPR: branches on it are scalar branches, not EXEC-masked regions)
    const int nx = p.nx, nu = p.nu, N = p.N, nxu = nx + nu, T = N - 1;
    const int W = GW, KT = GW;
    const long tile = blockIdx.x;
    const int jn = lane & 15, kq = lane >> 4;  // instance within the tile, row within a k-block / result quad
    const long inst = tile * M_INST + jn;
    const bool inst_ok = inst < p.batch;
    const int TOFF = (N + 2) * W;
    const size_t KD = m_knot_doubles(R);
    // Addressing: every state access is (tile- and knot-uniform base, in SGPRs) + (a 32-bit per-lane offset that depends on
    // the entry only): no 64-bit VGPR address per (array, entry) -- those cost the first version 200 VGPRs and its spills.
    double *const gG = p.G + (size_t)tile * N * KD;
    double *const gVa = p.V + (size_t)tile * N * KD;
    double *const gVb = p.V2 + (size_t)tile * N * KD;
    double *const gD = p.D + (size_t)tile * N * KD;
    // FAM: rollout x | u (transient), cone / linear duals (persistent), linear-cost term (forward -> backward), same layout
    double *const gXU = FAM ? p.scratch + (size_t)tile * N * KD : nullptr;
    double *const gGC = FAM ? p.GC + (size_t)tile * N * KD : nullptr;
    double *const gGL = FAM ? p.GL + (size_t)tile * N * KD : nullptr;
    double *const gLX = FAM ? p.LX + (size_t)tile * N * KD : nullptr;

    // this wave's result entries of its row tile tw (t = wv + 8 tw): e = 0..3 <-> reg = e, row = 16 t + kq + 4 e. Rows and their
    // kind are recomputed from the tile index where they are needed (cheap integer work) instead of living in masks.
    auto tile_of = [&](int tw) -> int { return wv + NW * tw; };
    auto has_tile = [&](int tw) -> bool { return (R == NW * TPW) || tile_of(tw) < R; };  // (uniform) this wave owns row tile tw
    auto row_of = [&](int tw, int e) -> int { return 16 * tile_of(tw) + kq + 4 * e; };
    auto kind_of = [&](int tw, int e) -> int { const int r = row_of(tw, e); return !has_tile(tw) ? 0 : (r < nx ? 1 : (r < nxu ? 2 : 0)); };  // 1 state, 2 input, 0 padding
    auto slot = [&](int tw, int e) -> unsigned { return (unsigned)((4 * tile_of(tw) + e) * 64 + lane); };  // offset of entry e inside a knot
    auto is_x = [&](int tw, int e) -> bool { return kind_of(tw, e) == 1; };
    auto is_u = [&](int tw, int e) -> bool { return kind_of(tw, e) == 2; };
    auto k_x = [&](int tw, int e) -> int { return is_x(tw, e) ? 1 : 0; };  // state rows own knot i + 1 of step i, input rows knot i
    const double rho = p.rho;
    const int ct = p.check_termination;
    // lo / hi / linref of (row, knot): table row kn + 1
    if constexpr (CT) {
        for (int i = tid; i < 3 * GW; i += 64 * NW) sTab[i / GW][i % GW] = p.tables[(unsigned)((i / GW) * TOFF + W + (i % GW))];
        __syncthreads();
    }
    auto tab = [&](int which, int kn, int tw, int e) -> double {
        if constexpr (CT) return sTab[which][row_of(tw, e)];
        else return p.tables[(unsigned)(which * TOFF + (kn + 1) * W + row_of(tw, e))];  // (uniform base + 32-bit offset)
    };
    const double *const cf_tab = p.ops + (size_t)2 * W * KT, *const cb_tab = cf_tab + W;
    if constexpr (STREAM) {
        for (int i = tid; i < 2 * GW; i += 64 * NW) sC[i / GW][i % GW] = cf_tab[i];
        __syncthreads();
    }
    // the constant term of a sweep's rows: in registers for the sweep (one row tile per wavefront), or from LDS at each step
    double start0[4] = {0.0, 0.0, 0.0, 0.0};
    auto load_start = [&](int which) {
        if constexpr (!STREAM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) start0[e] = kind_of(0, e) ? (which ? cb_tab : cf_tab)[row_of(0, e)] : 0.0;
        }
    };
    auto start_of = [&](int which, int tw, int e) -> double {
        if constexpr (STREAM) return kind_of(tw, e) ? sC[which][row_of(tw, e)] : 0.0;
        else return start0[e];
    };

    // A tiles of the sweep operator `which` (0: Mf, 1: Mb): register-resident for one sweep (R <= 8), or streamed (R > 8)
    double A0[STREAM ? 1 : KB];
    auto load_A = [&](int which) {
        if constexpr (!STREAM) {
            const double *M = p.ops + (size_t)which * W * KT;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) A0[kb] = has_tile(0) ? M[(size_t)(16 * wv + jn) * KT + 4 * kb + kq] : 0.0;
        }
    };
    // out[e] = start[e] + (rows of tile tw of the operator) * (operand vector in sX[buf]). Two accumulation chains per wavefront
    // (even / odd k-blocks), i.e. four per SIMD: a dependent FP64 MFMA only issues when its predecessor has left the pipe.
    auto gemm = [&](int buf, int which, int tw, const double (&start)[4], double (&out)[4]) {
        double4_m c0 = {start[0], start[1], start[2], start[3]}, d0 = {0.0, 0.0, 0.0, 0.0};
        if (has_tile(tw)) {
            // operand reads run a batch of eight k-blocks ahead of the matrix instructions that consume them (left to the
            // scheduler they ran two ahead, and an MFMA issued every 78 cycles instead of every 64)
            constexpr int BATCH = STREAM ? ((NW > 12 && (TPW == 1 || R == NW * TPW)) ? 2 : 4) : (CT ? 8 : (R == 8 ? 2 : 4));  // (streamed, sixteen wavefronts: 128 registers)  // (per-knot tables: fewer registers to spare)
            static_assert(KB % 4 == 0, "k-blocks come in fours");
            double b[2][BATCH], a[2][STREAM ? BATCH : 1];
            // (streamed: the tile-major copy of the operators, one 512-byte line per (row tile, k-block) -- read row-major, an
            // instruction touched 16 rows x 32 bytes, every cache line came in four times and the stream ran at the L2 -> CU limit)
            const double *Arow = p.ctab + ((size_t)(which * R + (STREAM ? tile_of(tw) : 0)) * KB) * 64 + lane;
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                b[0][q] = q < KB ? sX[buf][q][lane] : 0.0;
                if constexpr (STREAM) a[0][q] = q < KB ? Arow[64 * q] : 0.0;
            }
#pragma unroll
            for (int k0 = 0; k0 < KB; k0 += BATCH) {
                const int cur = (k0 / BATCH) & 1;
#pragma unroll
                for (int q = 0; q < BATCH; ++q)
                    if (k0 + BATCH + q < KB) {
                        b[cur ^ 1][q] = sX[buf][k0 + BATCH + q][lane];
                        if constexpr (STREAM) a[cur ^ 1][q] = Arow[64 * (k0 + BATCH + q)];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < BATCH; q += 2) {
                    if (k0 + q < KB) {
                        const double a_e = STREAM ? a[STREAM ? cur : 0][STREAM ? q : 0] : A0[STREAM ? 0 : k0 + q];
                        const double a_o = STREAM ? a[STREAM ? cur : 0][STREAM ? q + 1 : 0] : A0[STREAM ? 0 : k0 + q + 1];
#if TINY_EXP_M == 2  // timing experiment: the operand reads without the matrix instructions
                        c0[0] += a_e * b[cur][q];
                        d0[0] += a_o * b[cur][q + 1];
#else
                        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_e, b[cur][q], c0, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_o, b[cur][q + 1], d0, 0, 0, 0);
#endif
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = c0[q] + d0[q];
    };

    // (streamed kernels with ONE row tile per wavefront: the tile loops start from a zero the compiler cannot see through -- with a
    // constant tile index it hoists every address of the sweeps out of them, 100+ registers, and spills)
    int tw0 = 0;
    if constexpr (STREAM && TPW == 1) asm volatile("" : "+s"(tw0));
    bool active = inst_ok;      // (per lane: its instance is still iterating)
    int it_done = 0, status = 11;
    bool res_valid = false;
    double snap_pri_x = 0.0, snap_pri_u = 0.0, snap_dua_x = 0.0, snap_dua_u = 0.0;
    int par = 0;                // the slack buffer this iteration READS: 0 = V, 1 = V2
    int buf = 0;

    // FAM: few enough linear rows for the one-pass form (m_families_fast)? and which slots of a knot hold a cone row (uniform)
    bool fam_fast = false;
    unsigned long long cmask_lo = 0ull, cmask_hi = 0ull;
    if constexpr (FAM) {
        fam_fast = __builtin_amdgcn_readfirstlane((int)p.fam[4]) != 0;
        const int ncones = __builtin_amdgcn_readfirstlane((int)p.fam[0]) + __builtin_amdgcn_readfirstlane((int)p.fam[1]);
        for (int c = 0; c < ncones; ++c) {
            const int first = __builtin_amdgcn_readfirstlane((int)p.fam[m_fam_cone_offset() + 3 * c]);
            const int last = __builtin_amdgcn_readfirstlane((int)p.fam[m_fam_cone_offset() + 3 * c + 1]);
            for (int sl = 4 * (first >> 4) + ((first & 15) >> 2); sl <= 4 * (last >> 4) + ((last & 15) >> 2); ++sl) {
                if (sl < 64) cmask_lo |= 1ull << sl;
                else cmask_hi |= 1ull << (sl - 64);
            }
        }
    }
    for (int it = 0; it < p.max_iter; ++it) {  // admm.cpp:129
        if (__syncthreads_or(active ? 1 : 0) == 0) break;
        const bool check = (ct > 0) && (((it + 1) % ct) == 0);
        double *const Vr = par ? gVb : gVa, *const Vw = par ? gVa : gVb;
        double pri_x = 0.0, pri_u = 0.0, dua_x = 0.0, dua_u = 0.0;

        // ================= forward sweep (F1) with S1 + D1 + R1 fused in =================
        load_A(0);
        load_start(0);
        // operand of step 0: [x_0; d_0]; and knot 0 of the state rows: x_0 is given, only projected
#pragma nounroll
        for (int tw = tw0; tw < TPW; ++tw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kd = kind_of(tw, e);
                double w = 0.0;
                if (kd == 1) {
                    const double x0 = inst_ok ? p.x0[inst * nx + row_of(tw, e)] : 0.0;
                    const double g = gG[slot(tw, e)], vold = Vr[slot(tw, e)];
                    const double s = x0 + g;
                    const double snew = fmin(tab(1, 0, tw, e), fmax(tab(0, 0, tw, e), s));
                    pri_x = fmax(pri_x, fabs(x0 - snew));
                    dua_x = fmax(dua_x, fabs(vold - snew));
                    if (active) {
                        gG[slot(tw, e)] = s - snew;
                        Vw[slot(tw, e)] = snew;
                        if constexpr (FAM) gXU[slot(tw, e)] = x0;
                    }
                    w = x0;
                } else if (kd == 2) {
                    w = gD[slot(tw, e)];
                }
                if (has_tile(tw)) sX[buf][4 * tile_of(tw) + e][lane] = w;
            }
        }
        __syncthreads();
        // The row-local operands of a step (dual, old slack, the next step's feed-forward entry) do not depend on its GEMM: they
        // are requested in front of it and arrive while the matrix cores work -- the state streams through HBM at these sizes
        // (the kernel without its MFMAs runs at the HBM roof), and a step that waited for its operands AFTER its MFMAs ran at
        // a fifth of this speed.
        // BRANCH-FREE on purpose: every lane of a wavefront that owns a row tile loads and stores all four of its entries
        // (padding rows have slots of their own; lanes without a feed-forward entry all read one dummy address). With the
        // loads inside `if (row is real)` regions the compiler put an s_waitcnt vmcnt(0) in front of every entry's address
        // arithmetic -- four serialised HBM round trips per step instead of one.
        for (int i = 0; i < T; ++i) {
#pragma nounroll
            for (int tw = tw0; tw < TPW; ++tw) {  // (one copy of the code: unrolled, the second tile's addresses were hoisted and spilled)
                if (!has_tile(tw)) continue;
                double pg[4], pv[4], pd[4], start[4];
                const bool more = i + 1 < T;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    start[e] = start_of(0, tw, e);
                    if (TINY_EXP_M != 1) {
                        const unsigned o = (unsigned)((i + k_x(tw, e)) * (int)KD) + slot(tw, e);
                        pg[e] = gG[o];
                        pv[e] = Vr[o];
                        pd[e] = gD[(is_u(tw, e) && more) ? (unsigned)((i + 1) * (int)KD) + slot(tw, e) : 0u];
                    } else {
                        pg[e] = pv[e] = pd[e] = 0.0;
                    }
                }
                double out[4];
                gemm(buf, 0, tw, start, out);  // state rows: x_{i+1}; input rows: u_i
                double gn[4], sn[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kn = i + k_x(tw, e);
                    const double s = out[e] + pg[e];
                    const double snew = fmin(tab(1, kn, tw, e), fmax(tab(0, kn, tw, e), s));  // (bounds: LDS copy, or the L2-resident table)
                    const double tp = fabs(out[e] - snew), td = fabs(pv[e] - snew);
                    pri_x = fmax(pri_x, is_x(tw, e) ? tp : 0.0);
                    dua_x = fmax(dua_x, is_x(tw, e) ? td : 0.0);
                    pri_u = fmax(pri_u, is_u(tw, e) ? tp : 0.0);
                    dua_u = fmax(dua_u, is_u(tw, e) ? td : 0.0);
                    gn[e] = s - snew;
                    sn[e] = snew;
                    // next operand: state rows carry x_{i+1}, input rows bring d_{i+1}, padding rows stay zero
                    sX[buf ^ 1][4 * tile_of(tw) + e][lane] = is_u(tw, e) ? pd[e] : (is_x(tw, e) ? out[e] : 0.0);
                }
                if (active && TINY_EXP_M != 1) {  // (one masked region, stores only)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned o = (unsigned)((i + k_x(tw, e)) * (int)KD) + slot(tw, e);
                        gG[o] = gn[e];
                        Vw[o] = sn[e];
                        if constexpr (FAM) gXU[o] = out[e];
                    }
                }
            }
            buf ^= 1;
            lds_exchange_barrier();
        }
        // ================= the cone / linear families of this iterate (instances that entered the iteration active) =================
        if constexpr (FAM) {
            __syncthreads();  // the rollout's rows come from every wavefront of the tile (waits for the sweep's stores)
            if (fam_fast) m_families_fast<R, NW>(p, gXU, gGC, gGL, gLX, wv, lane, active, cmask_lo, cmask_hi);
            else m_families<R, NW>(p, gXU, gGC, gGL, gLX, wv, lane, active);
            __syncthreads();  // the backward sweep reads LX by row tile
        }
        if (active) it_done = it + 1;  // admm.cpp:143

        // ================= R1: termination (admm.cpp:93-101), per instance over all rows and all wavefronts of the tile =================
        bool conv = false;
        if (check) {
            const bool below = (pri_x < p.abs_pri_tol) && (pri_u < p.abs_pri_tol) && (dua_x * rho < p.abs_dua_tol) && (dua_u * rho < p.abs_dua_tol);
            const unsigned long long m = __ballot(below);
            const unsigned q = (unsigned)(m & (m >> 16) & (m >> 32) & (m >> 48)) & 0xffffu;  // instance j: all four row quads of this wave
            if (lane == 0) sFlag[it & 1][wv] = q;
            __syncthreads();
            unsigned all = 0xffffu;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) all &= sFlag[it & 1][w8];
            conv = ((all >> jn) & 1u) != 0u;
            // the four inf-norms of this instance: maxima over its lanes in this wave now, over the waves after the loop
            if (active) {
                snap_pri_x = pri_x;
                snap_pri_u = pri_u;
                snap_dua_x = dua_x;
                snap_dua_u = dua_u;
                res_valid = true;
                if (conv) {
                    status = 1;  // TINY_SOLVED: stops before the backward pass; its canonical slack is the buffer just READ
                    active = false;
                }
            }
        }
        const int par_read = par;
        par ^= 1;

        // ================= backward sweep (B1, admm.cpp:13-20); linear cost (L1, :77-82) from V (just written), G =================
        load_A(1);
        load_start(1);
        double *const Vn = par_read ? gVa : gVb;  // the slack written by this iteration's forward sweep
#pragma nounroll
        for (int tw = tw0; tw < TPW; ++tw) {  // operand of step N-2: [p_{N-1}; r_{N-2}]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kd = kind_of(tw, e);
                double w = 0.0;
                if (kd == 1) {
                    const unsigned o = (unsigned)((N - 1) * (int)KD) + slot(tw, e);
                    w = p.tables[(size_t)3 * TOFF + row_of(tw, e)] - rho * (Vn[o] - gG[o]);  // p_{N-1}, admm.cpp:81-82
                    if constexpr (FAM) w += gLX[o];
                } else if (kd == 2) {
                    const unsigned o = (unsigned)((N - 2) * (int)KD) + slot(tw, e);
                    w = tab(2, N - 2, tw, e) - rho * (Vn[o] - gG[o]);  // r_{N-2}, admm.cpp:77-78
                    if constexpr (FAM) w += gLX[o];
                }
                if (has_tile(tw)) sX[buf][4 * tile_of(tw) + e][lane] = w;
            }
        }
        __syncthreads();
        // q_i (state rows, knot i) and r_{i-1} (input rows, knot i-1) from V, G and the table: requested before the GEMM like the
        // forward operands (branch-free: padding rows and the input rows of step 0 load their own slot and drop it)
        for (int i = T - 1; i >= 0; --i) {
#pragma nounroll
            for (int tw = tw0; tw < TPW; ++tw) {
                if (!has_tile(tw)) continue;
                double lv[4], lg[4], start[4], lx[FAM ? 4 : 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    start[e] = start_of(1, tw, e);
                    if (TINY_EXP_M != 1) {
                        const int kn = (is_u(tw, e) && i >= 1) ? i - 1 : i;
                        const unsigned o = (unsigned)(kn * (int)KD) + slot(tw, e);
                        lv[e] = Vn[o];
                        lg[e] = gG[o];
                        if constexpr (FAM) lx[e] = gLX[o];
                    } else {
                        lv[e] = lg[e] = 0.0;
                        if constexpr (FAM) lx[e] = 0.0;
                    }
                }
                double out[4];
                gemm(buf, 1, tw, start, out);  // state rows: AmBKt p_{i+1} - Kinf' r_i (+ APf); input rows: d_i
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kn = (is_u(tw, e) && i >= 1) ? i - 1 : i;
                    const bool real = is_x(tw, e) || (is_u(tw, e) && i >= 1);
                    double lin = (real && TINY_EXP_M != 1) ? tab(2, kn, tw, e) - rho * (lv[e] - lg[e]) : 0.0;  // admm.cpp:77-80
                    if constexpr (FAM) lin = real ? lin + lx[e] : 0.0;
                    sX[buf ^ 1][4 * tile_of(tw) + e][lane] = is_x(tw, e) ? lin + out[e] : lin;                             // p_i = q_i + ...
                }
                if (active && TINY_EXP_M != 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (is_u(tw, e)) gD[(unsigned)(i * (int)KD) + slot(tw, e)] = out[e];  // d_i (a converged instance keeps its last real d)
                }
            }
            buf ^= 1;
            lds_exchange_barrier();
        }
    }

    // ---- canonical slack: an instance that stopped at max_iter has v <- vnew (admm.cpp:196-197): the buffer `par` now points
    // at; a converged one keeps the previous iterate: the buffer it READ in its last iteration. Both must end up in p.V.
    // `par` is block-uniform, so a lane goes by its own iteration count: after k forward sweeps the last-written buffer is V2 if
    // k is odd, V if even.
    if (p.max_iter > 0 && inst_ok) {
        const bool last_written_is_b = (it_done & 1) != 0;
        double *const Vsol = last_written_is_b ? gVb : gVa;
        double *const Vold = last_written_is_b ? gVa : gVb;
#pragma nounroll
        for (int tw = tw0; tw < TPW; ++tw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kd = kind_of(tw, e), rw = row_of(tw, e);
                if (kd == 0) continue;
                const int knots = kd == 1 ? N : N - 1;
                for (int kn = 0; kn < knots; ++kn) {
                    const unsigned o = (unsigned)(kn * (int)KD) + slot(tw, e);
                    const double sol = Vsol[o];
                    if (kd == 1) p.sol_x[((size_t)inst * N + kn) * nx + rw] = sol;
                    else p.sol_u[((size_t)inst * (N - 1) + kn) * nu + (rw - nx)] = sol;
                    const double canon = (status == 1) ? Vold[o] : sol;
                    if (it_done > 0) {
                        gVa[o] = canon;  // p.V is the canonical copy between solves (both stores are by this lane, in order)
                    }
                }
            }
        }
    }
    // residual norms of the last check: max over this instance's lanes in the wave, then over the wavefronts of the tile
    __syncthreads();
    double *sR = &sX[0][0][0];  // reuse: [NW waves][4 norms][16 instances]
    {
        double v4[4] = {snap_pri_x, snap_dua_x, snap_pri_u, snap_dua_u};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = v4[q];
            v = fmax(v, __shfl_xor(v, 16));
            v = fmax(v, __shfl_xor(v, 32));
            if (kq == 0) sR[(wv * 4 + q) * 16 + jn] = v;
        }
    }
    __syncthreads();
    if (wv == 0 && kq == 0 && inst_ok) {
        p.istats[inst * 2 + 0] = it_done;
        p.istats[inst * 2 + 1] = status;
        if (res_valid) {
            for (int q = 0; q < 4; ++q) {
                double v = 0.0;
                for (int w8 = 0; w8 < NW; ++w8) v = fmax(v, sR[(w8 * 4 + q) * 16 + jn]);
                p.dstats[inst * 4 + q] = (q == 1 || q == 3) ? v * rho : v;
            }
        }
    }
}

// Tile-major copy of the two sweep operators for R > 8 (see gemm): out[which][row tile t][k-block kb][lane] = the MFMA A operand
// of that lane, M_which[16 t + (lane & 15)][4 kb + (lane >> 4)].
__global__ void __launch_bounds__(256) k_tile_operators_m(const double *ops, double *out, int R, int KT) {
    const size_t total = (size_t)2 * R * 4 * R * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const size_t blk = i >> 6;
        const int kb = (int)(blk % (4 * R)), t = (int)((blk / (4 * R)) % R), which = (int)(blk / ((size_t)4 * R * R));
        out[i] = ops[(size_t)which * KT * KT + (size_t)(16 * t + (lane & 15)) * KT + 4 * kb + (lane >> 4)];
    }
}
size_t solve_m_tiled_ops_doubles(int nx, int nu) {
    const int R = (nx + nu + 15) / 16;
    return R > 8 ? (size_t)2 * R * 4 * R * 64 : 0;
}
hipError_t launch_tile_operators_m(const double *ops, double *out, int nx, int nu, hipStream_t stream) {
    const int R = (nx + nu + 15) / 16;
    if (R <= 8) return hipSuccess;
    hipLaunchKernelGGL(k_tile_operators_m, dim3(64), dim3(256), 0, stream, ops, out, R, m_geometry(R));
    return hipGetLastError();
}

hipError_t launch_solve_m(const SolveParams &p, hipStream_t stream) {
    const int R = (p.nx + p.nu + 15) / 16;
    const int tiles = (p.batch + M_INST - 1) / M_INST;
#define TINY_M_LAUNCH(R_)                                                                                                \
    case R_:                                                                                                              \
        if (p.families) hipLaunchKernelGGL((k_admm_solve_m<R_, false, true>), dim3(tiles), dim3(64 * m_waves(R_)), 0, stream, p); \
        else if (p.const_tables) hipLaunchKernelGGL((k_admm_solve_m<R_, true>), dim3(tiles), dim3(64 * m_waves(R_)), 0, stream, p); \
        else hipLaunchKernelGGL((k_admm_solve_m<R_, false>), dim3(tiles), dim3(64 * m_waves(R_)), 0, stream, p);              \
        break;
    switch (R) {
        TINY_M_LAUNCH(5)
        TINY_M_LAUNCH(6)
        TINY_M_LAUNCH(7)
        TINY_M_LAUNCH(8)
        TINY_M_LAUNCH(9)
        TINY_M_LAUNCH(10)
        TINY_M_LAUNCH(11)
        TINY_M_LAUNCH(12)
        TINY_M_LAUNCH(13)
        TINY_M_LAUNCH(14)
        TINY_M_LAUNCH(15)
        TINY_M_LAUNCH(16)
        TINY_M_LAUNCH(17)
        TINY_M_LAUNCH(18)
        TINY_M_LAUNCH(19)
        TINY_M_LAUNCH(20)
        TINY_M_LAUNCH(21)
        TINY_M_LAUNCH(22)
        TINY_M_LAUNCH(23)
        TINY_M_LAUNCH(24)
        TINY_M_LAUNCH(25)
        TINY_M_LAUNCH(26)
        TINY_M_LAUNCH(27)
        TINY_M_LAUNCH(28)
        TINY_M_LAUNCH(29)
        TINY_M_LAUNCH(30)
        TINY_M_LAUNCH(31)
        TINY_M_LAUNCH(32)
        default: return hipErrorInvalidValue;
    }
#undef TINY_M_LAUNCH
    return hipGetLastError();
}

}  // namespace tinympc
