// tinympc_fam_red.h -- the cone / linear-inequality families for ONE (row, knot) element per lane on WIDE systems (32 / 64 lanes per
// instance), cross-row quantities by group REDUCTIONS: the evaluation k_admm_solve_fam uses on these widths (tinympc_solve_fam.hip, its
// RED branch -- same formulas, same order of operations, same family buffer p.fam), as a struct so that layout D's wide kernels
// (tinympc_solve_dwide.h, round 5) can ride it on their register-resident sweeps. PARITY UNPINNED upstream semantics (see there).
//   cone family    s_c = x + gc, projected onto the row's cone(s), round by round (cones that share rows: one after the other);
//                  a cone's ||w||^2 is one group sum over its tail rows, t one lane read
//   linear family  s_l = x + gl pushed through the half-spaces a_k' s <= b_k one after another; a row's a_k' s is one group sum per
//                  side that HAS rows (the other side's sum is skipped: round 5)
//   returns lx = -rho (vcnew - gc_new) - rho (vlnew - gl_new), the element's contribution to the linear cost; new duals in gc_new / gl_new
#pragma once
#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {

// Sum over the W lanes of an instance, the same bits in every lane -- group_sum<W> (tinympc_sweep.h) with its cross-row step(s) on the
// gfx950 swaps (v_permlane16_swap / v_permlane32_swap: VALU) instead of ds_bpermute (an LDS-crossbar round trip per step, on the serial
// path of every cone and every linear row of every sweep step). Same additions, same operands: bit-identical.
template <int W>
__device__ __forceinline__ double group_sum_swap(double v) {
    static_assert(W == 16 || W == 32 || W == 64, "groups of 16, 32 or 64 lanes");
    v += dpp_exchange<0xB1>(v);
    v += dpp_exchange<0x4E>(v);
    v += dpp_exchange<0x141>(v);
    v += dpp_exchange<0x140>(v);  // every lane: the sum of its 16-lane row
    if constexpr (W == 64) {
        double lo, hi;
        cross_row_pair<0>(v, lo, hi);  // [r0 r1 r0 r1], [r2 r3 r2 r3]
        v = lo + hi;                   // rows 0, 2: r0 + r2; rows 1, 3: r1 + r3
    }
    if constexpr (W >= 32) {
        double e, o;
        cross_row_pair<1>(v, e, o);    // even-row / odd-row copies on both rows of a pair
        v = e + o;
    }
    return v;
}

#ifndef TINY_WIDE_FAM_SWAP
#define TINY_WIDE_FAM_SWAP 1  // 0 (experiments, TINYMPC_JIT_DEFS): ds_bpermute for the cross-row steps. One element at a time the swap form is 5-8 % SLOWER
                              // (its split / swap / recombine sits on the serial path); with several elements in flight (eval_batch) it wins 12-25 %
#endif
template <int W>
__device__ __forceinline__ double red_group_sum(double v) {
    if constexpr (TINY_WIDE_FAM_SWAP != 0 && W > 16) return group_sum_swap<W>(v);
    else return group_sum<W>(v);
}

template <int W>
struct RedFamilies {
    int r_role[MAX_ROUNDS], r_head[MAX_ROUNDS];
    double r_mu[MAX_ROUNDS], r_imu[MAX_ROUNDS];
    unsigned long long r_cones[MAX_ROUNDS];  // per round: the cones' last rows as a bit mask of group-relative lane numbers (uniform)
    const double *lin;   // nl | per row: a_k[W] | b_k[W] | ||a_k||^2 [W] (the family buffer in L2), or an LDS copy holding the RECIPROCAL norms
    bool lin_recip;      // ... which of the two
    double rho;
    int r, nround, nl;
    bool is_x, is_u, famc, faml, any_cone, lin_x_on, lin_u_on;

    __device__ __forceinline__ void init(const double *F, int nxu, int r_, bool is_x_, bool is_u_, double rho_, const double *lin_lds = nullptr) {
        constexpr int KT = W;
        r = r_; is_x = is_x_; is_u = is_u_; rho = rho_;
        nround = __builtin_amdgcn_readfirstlane((int)F[fam_nround_offset(W, KT)]);
#pragma unroll
        for (int q = 0; q < MAX_ROUNDS; ++q) {
            r_role[q] = 0; r_head[q] = -1; r_mu[q] = 0.0; r_imu[q] = 0.0; r_cones[q] = 0ull;
            if (q < nround) {  // (uniform)
                const double *rd = q == 0 ? F : F + fam_round_offset(W, KT, q);
                const double *ctq = q == 0 ? F + 4 * W + (size_t)W * KT : rd + 2 * W + (size_t)W * KT;
                r_role[q] = (int)rd[r];
                r_mu[q] = rd[W + r];
                r_imu[q] = (r_mu[q] != 0.0) ? 1.0 / r_mu[q] : 0.0;
                for (int k = 0; k < nxu; ++k)
                    if (ctq[(size_t)r * KT + k] != 0.0) r_head[q] = k;
                const unsigned long long heads = __ballot(r_head[q] == r);
                r_cones[q] = (W == 64) ? heads : (heads & ((1ull << (W % 64)) - 1ull));  // (every instance of the wave has the same cones)
            }
        }
        famc = F[2 * W + r] != 0.0;
        faml = F[3 * W + r] != 0.0;
        lin = lin_lds ? lin_lds : F + 4 * W + (size_t)3 * W * KT;
        lin_recip = lin_lds != nullptr;
        nl = __builtin_amdgcn_readfirstlane((int)lin[0]);
        any_cone = __ballot(famc) != 0ull;
        lin_x_on = __ballot(faml && is_x) != 0ull;
        lin_u_on = __ballot(faml && is_u) != 0ull;
    }

    __device__ __forceinline__ double eval(double val, double gc_old, double gl_old, double &gc_new, double &gl_new) const {
        double lx = 0.0;
        gc_new = gc_old;
        gl_new = gl_old;
        if (any_cone) {
            const double sv = val + gc_old;  // vcnew = x + gc (all rows of an enabled side)
            double vc = sv;
#pragma unroll
            for (int q = 0; q < MAX_ROUNDS; ++q) {
                if (q < nround) {  // (uniform)
                    double a2 = 0.0;
                    for (unsigned long long m = r_cones[q]; m != 0ull; m &= m - 1ull) {  // one cone of the round after the other (uniform)
                        const int hc = __builtin_ctzll(m);
                        const bool mine = r_head[q] == hc;
                        const double tail2 = red_group_sum<W>((mine && r_role[q] == 1) ? vc * vc : 0.0);  // ||w||^2 of that cone
                        a2 = mine ? tail2 : a2;
                    }
                    const double t = __shfl(vc, r_head[q] >= 0 ? r_head[q] : r, W);  // last entry of the row's cone
                    vc = soc_project_element(vc, a2, t, r_mu[q], r_imu[q], r_role[q]);
                }
            }
            const double gcn = sv - vc;  // gc + x - vcnew
            if (famc) {
                gc_new = gcn;
                lx -= rho * (vc - gcn);
            }
        }
        if (lin_x_on || lin_u_on) {
            const double s0 = val + gl_old;
            double sv = s0;
#pragma unroll 1
            for (int k = 0; k < nl; ++k) {  // (uniform trip count; the rows' coefficients from the family buffer in L2)
                const double a_k = lin[1 + (size_t)(3 * k + 0) * W + r], b_k = lin[1 + (size_t)(3 * k + 1) * W + r];
                const double nk = lin[1 + (size_t)(3 * k + 2) * W + r];
                const double in_k = lin_recip ? nk : 1.0 / nk;
                const double prod = a_k * sv;
                const double dx = lin_x_on ? red_group_sum<W>(is_x ? prod : 0.0) : 0.0;  // a_k' x
                const double du = lin_u_on ? red_group_sum<W>(is_u ? prod : 0.0) : 0.0;  // a_k' u
                sv = halfspace_project_element(sv, is_x ? dx : du, a_k, b_k, in_k);
            }
            const double gln = s0 - sv;
            if (faml) {
                gl_new = gln;
                lx -= rho * (sv - gln);
            }
        }
        return lx;
    }

    // The same for NB elements (NB slots of a sweep) at once, the loop over the elements INNERMOST: inside one cone / one linear row the NB
    // reductions and projections are independent instruction streams of ONE basic block, which the scheduler interleaves -- eval() called
    // NB times is NB sequences of blocks (its loops over cones and rows are control flow) and nothing overlaps. Per element the operations
    // and their order are eval()'s: bit-identical.
    template <int NB>
    __device__ __forceinline__ void eval_batch(const double (&val)[NB], const double (&gc_old)[NB], const double (&gl_old)[NB], double (&gc_new)[NB],
                                               double (&gl_new)[NB], double (&lx)[NB]) const {
#pragma unroll
        for (int b = 0; b < NB; ++b) { lx[b] = 0.0; gc_new[b] = gc_old[b]; gl_new[b] = gl_old[b]; }
        if (any_cone) {
            double sv[NB], vc[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { sv[b] = val[b] + gc_old[b]; vc[b] = sv[b]; }
#pragma unroll
            for (int q = 0; q < MAX_ROUNDS; ++q) {
                if (q < nround) {  // (uniform)
                    double a2[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) a2[b] = 0.0;
                    for (unsigned long long m = r_cones[q]; m != 0ull; m &= m - 1ull) {  // one cone of the round after the other (uniform)
                        const int hc = __builtin_ctzll(m);
                        const bool mine = r_head[q] == hc;
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const double tail2 = red_group_sum<W>((mine && r_role[q] == 1) ? vc[b] * vc[b] : 0.0);
                            a2[b] = mine ? tail2 : a2[b];
                        }
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const double t = __shfl(vc[b], r_head[q] >= 0 ? r_head[q] : r, W);
                        vc[b] = soc_project_element(vc[b], a2[b], t, r_mu[q], r_imu[q], r_role[q]);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double gcn = sv[b] - vc[b];
                if (famc) {
                    gc_new[b] = gcn;
                    lx[b] -= rho * (vc[b] - gcn);
                }
            }
        }
        if (lin_x_on || lin_u_on) {
            double s0[NB], sv[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { s0[b] = val[b] + gl_old[b]; sv[b] = s0[b]; }
#pragma unroll 1
            for (int k = 0; k < nl; ++k) {
                const double a_k = lin[1 + (size_t)(3 * k + 0) * W + r], b_k = lin[1 + (size_t)(3 * k + 1) * W + r];
                const double nk = lin[1 + (size_t)(3 * k + 2) * W + r];
                const double in_k = lin_recip ? nk : 1.0 / nk;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const double prod = a_k * sv[b];
                    const double dx = lin_x_on ? red_group_sum<W>(is_x ? prod : 0.0) : 0.0;
                    const double du = lin_u_on ? red_group_sum<W>(is_u ? prod : 0.0) : 0.0;
                    sv[b] = halfspace_project_element(sv[b], is_x ? dx : du, a_k, b_k, in_k);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double gln = s0[b] - sv[b];
                if (faml) {
                    gl_new[b] = gln;
                    lx[b] -= rho * (sv[b] - gln);
                }
            }
        }
    }
};

}  // namespace tinympc
