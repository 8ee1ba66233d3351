// tinympc_solve_e_common.h -- what the structure-specialised kernels share (layout E, tinympc_solve_e.hip, and layout F,
// tinympc_solve_f.hip; both exist only as run-time specialisations): compile-time loops, LDS accesses placed by hand, the
// cross-lane sums of the cone / linear families under an EXEC mask, and the families' row-local evaluation specialised on their
// STRUCTURE (PARITY UNPINNED upstream semantics, see tinympc_solve_fam.hip).
//
// Structure = the -D options TINY_JIT_E_NROUND / _NCONE / _CONES / _NLX / _NLU (tinympc_jit.hip from FamilyStructure): the cone
// list in list order as {round, first lane, last lane (the cone's t row)} -- lane = row for state rows, NX + row for input rows;
// cones of one round are pairwise disjoint and are projected together, a cone that overlaps an earlier one of its round opens the
// next round (= upstream's one-after-another projection) -- and the number of linear rows per side. Cross-lane sums then need no
// mask rows (72 VGPRs in the generic kernels): a cone's ||w||^2 and t are gathered by `dim` DPP instructions under an EXEC mask
// of the cone's lanes, a linear row's dot product by one DPP instruction per row of its side.
#pragma once
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {

template <int I, int E, class F>
__device__ __forceinline__ void e_static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        e_static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ unsigned e_lds_addr(const double *p) { return lds_address(p); }
template <int OFF>
__device__ __forceinline__ double e_lds_read_async(unsigned addr) {  // issued HERE, tracked by the compiler (tinympc_sweep.h); valid after the next lds_reads_landed()
    return lds_read_issued_here<OFF>(addr);
}
template <int OFF>
__device__ __forceinline__ void e_lds_write_async(unsigned addr, double v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void e_lds_write_masked(unsigned addr, double v, unsigned long long mask) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "ds_write_b64 %[a], %[v] offset:%[o]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved)
                 : [m] "s"(mask), [a] "v"(addr), [v] "v"(v), [o] "n"(OFF)
                 : "memory", "scc");
}
// dst <- v on the lanes of `mask` only (a register-resident array element that some instances of the wavefront must keep)
__device__ __forceinline__ void e_reg_write_masked(double &dst, double v, unsigned long long mask) {
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "v_mov_b64 %[d], %[v]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved), [d] "+v"(dst)
                 : [m] "s"(mask), [v] "v"(v)
                 : "scc");
}
__device__ __forceinline__ void e_lds_wait() {
    lds_reads_landed();
    asm volatile("" ::: "memory");
}
// ... the same wait, with the values of e_lds_read_async that ordinary C++ arithmetic consumes next passed THROUGH it. (History: while
// reads and waits were asm text the compiler saw no dependence between them and scheduled arithmetic on a read's result in front of the
// wait -- round 4, two thirds of a batch's iteration counts went wrong; these pass-throughs were the fix. Round 5 found the same class once
// more, as a register copy, and removed its cause: reads and waits are tracked by the compiler now, tinympc_sweep.h. The pass-throughs stay
// as scheduling points: they keep consumers BEHIND the hand-placed wait instead of having the compiler add one of its own earlier.)
__device__ __forceinline__ void e_lds_arrived() {}
template <class... T>
__device__ __forceinline__ void e_lds_arrived(double &v, T &...rest) {
    asm volatile("" : "+v"(v));
    e_lds_arrived(rest...);
}
template <int NV>
__device__ __forceinline__ void e_lds_wait_for(double (&v)[NV]) {
    lds_reads_landed();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));
}
// workgroup barrier that waits for this wavefront's LDS traffic only (not for its global stores, as __syncthreads() would)
#if defined(TINY_E_EXP) && (TINY_E_EXP == 4 || TINY_E_EXP == 5)  // (timing experiment: no barrier)
__device__ __forceinline__ void e_barrier() {
    lds_reads_landed();
    asm volatile("" ::: "memory");
}
#else
__device__ __forceinline__ void e_barrier() {
    // (the wait once for the compiler's scoreboard and once FUSED with the barrier: between two statements the compiler may place code --
    // it did put ds_bpermutes there --, and a store of this wavefront that other wavefronts read behind the barrier must have landed)
    lds_reads_landed();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif

__device__ __forceinline__ bool e_wave_may_converge(unsigned long long bad, unsigned long long live) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long b = (bad >> (j * 16)) & 0xffffull, l = (live >> (j * 16)) & 0xffffull;
        any = any || (l != 0ull && b == 0ull);
    }
    return any;
}

// ---- cross-lane sums of the families under an EXEC mask --------------------------------------------------------------------
// acc (lanes of `mask` only) = sum of w over the CNT lanes F .. F+CNT-1 of the lane's own DPP row; with HAS_T also
// t = s of lane TL. Every source lane belongs to the mask itself (a cone's members and its t row; the rows of one side), so no
// DPP read crosses into a disabled lane. ONE asm statement: nothing the compiler schedules may run under the narrowed EXEC.
// Hazards: `w` / `s` are fresh VALU results (2 wait states before a DPP read) -- s_and_saveexec + `s_nop 1` in front.
#define TINY_EG_HEAD "s_and_saveexec_b64 %[sx], %[mk]\n\ts_nop 1\n\t"
#define TINY_EG_MOV "v_mov_b64_dpp %[acc], %[w] row_newbcast:%[c0] row_mask:0xf bank_mask:0xf\n\t"
#define TINY_EG_FM(j) "v_fmac_f64_dpp %[acc], %[w], %[one] row_newbcast:%[c" #j "] row_mask:0xf bank_mask:0xf\n\t"
#define TINY_EG_T "v_mov_b64_dpp %[t], %[s] row_newbcast:%[tl] row_mask:0xf bank_mask:0xf\n\t"
#define TINY_EG_TAIL "s_mov_b64 exec, %[sx]"
#define TINY_EG_1 TINY_EG_MOV
#define TINY_EG_2 TINY_EG_1 TINY_EG_FM(1)
#define TINY_EG_3 TINY_EG_2 TINY_EG_FM(2)
#define TINY_EG_4 TINY_EG_3 TINY_EG_FM(3)
#define TINY_EG_5 TINY_EG_4 TINY_EG_FM(4)
#define TINY_EG_6 TINY_EG_5 TINY_EG_FM(5)
#define TINY_EG_7 TINY_EG_6 TINY_EG_FM(6)
#define TINY_EG_8 TINY_EG_7 TINY_EG_FM(7)
#define TINY_EG_9 TINY_EG_8 TINY_EG_FM(8)
#define TINY_EG_10 TINY_EG_9 TINY_EG_FM(9)
#define TINY_EG_11 TINY_EG_10 TINY_EG_FM(10)
#define TINY_EG_12 TINY_EG_11 TINY_EG_FM(11)
#define TINY_EG_13 TINY_EG_12 TINY_EG_FM(12)
#define TINY_EG_14 TINY_EG_13 TINY_EG_FM(13)
#define TINY_EG_15 TINY_EG_14 TINY_EG_FM(14)
#define TINY_EG_OPS(F)                                                                                                             \
    [c0] "n"(F), [c1] "n"((F + 1) & 15), [c2] "n"((F + 2) & 15), [c3] "n"((F + 3) & 15), [c4] "n"((F + 4) & 15), [c5] "n"((F + 5) & 15), \
        [c6] "n"((F + 6) & 15), [c7] "n"((F + 7) & 15), [c8] "n"((F + 8) & 15), [c9] "n"((F + 9) & 15), [c10] "n"((F + 10) & 15),        \
        [c11] "n"((F + 11) & 15), [c12] "n"((F + 12) & 15), [c13] "n"((F + 13) & 15), [c14] "n"((F + 14) & 15)
template <int F, int CNT, bool HAS_T, int TL>
__device__ __forceinline__ void e_masked_gather(unsigned long long mask, double w, double s, double one, double &acc, double &t) {
    static_assert(F >= 0 && CNT >= 0 && CNT <= 15 && F + CNT <= 16 && TL >= 0 && TL < 16, "lanes of one DPP row");
    unsigned long long sx;
#define TINY_EG_CASE(N_, BODY)                                                                                         \
    else if constexpr (CNT == N_ && HAS_T) asm volatile(TINY_EG_HEAD BODY TINY_EG_T TINY_EG_TAIL                        \
                                                        : [acc] "+v"(acc), [t] "+v"(t), [sx] "=&s"(sx)                \
                                                        : [mk] "s"(mask), [w] "v"(w), [s] "v"(s), [one] "v"(one), [tl] "n"(TL), TINY_EG_OPS(F) \
                                                        : "scc");                                                     \
    else if constexpr (CNT == N_ && !HAS_T) asm volatile(TINY_EG_HEAD BODY TINY_EG_TAIL                                 \
                                                         : [acc] "+v"(acc), [sx] "=&s"(sx)                            \
                                                         : [mk] "s"(mask), [w] "v"(w), [one] "v"(one), TINY_EG_OPS(F)  \
                                                         : "scc");
    if constexpr (CNT == 0 && HAS_T)  // a one-row cone: no norm members (acc keeps its 0), only t
        asm volatile(TINY_EG_HEAD TINY_EG_T TINY_EG_TAIL : [t] "+v"(t), [sx] "=&s"(sx) : [mk] "s"(mask), [s] "v"(s), [tl] "n"(TL) : "scc");
    else if constexpr (CNT == 0) {}
    TINY_EG_CASE(1, TINY_EG_1) TINY_EG_CASE(2, TINY_EG_2) TINY_EG_CASE(3, TINY_EG_3) TINY_EG_CASE(4, TINY_EG_4) TINY_EG_CASE(5, TINY_EG_5)
    TINY_EG_CASE(6, TINY_EG_6) TINY_EG_CASE(7, TINY_EG_7) TINY_EG_CASE(8, TINY_EG_8) TINY_EG_CASE(9, TINY_EG_9) TINY_EG_CASE(10, TINY_EG_10)
    TINY_EG_CASE(11, TINY_EG_11) TINY_EG_CASE(12, TINY_EG_12) TINY_EG_CASE(13, TINY_EG_13) TINY_EG_CASE(14, TINY_EG_14) TINY_EG_CASE(15, TINY_EG_15)
#undef TINY_EG_CASE
}

// The cone list of the specialisation: {round, first lane, last lane (the cone's t row)}, in list order (state cones, then
// input cones -- lane = row for state rows, NX + row for input rows). Rounds: cones of one round are pairwise disjoint and are
// projected together; a cone that overlaps an earlier one of its round opens the next round (built by the host).
struct EConeDesc {
    int round, first, last;
};
#if TINY_JIT_E_NCONE > 0
constexpr EConeDesc E_CONES[] = {TINY_JIT_E_CONES};
#else
constexpr EConeDesc E_CONES[] = {{-1, 0, 0}};
#endif
constexpr int E_NCONE = TINY_JIT_E_NCONE, E_NROUND = TINY_JIT_E_NROUND, E_NLX = TINY_JIT_E_NLX, E_NLU = TINY_JIT_E_NLU;
constexpr int E_NL = E_NLX > E_NLU ? E_NLX : E_NLU;
static_assert(E_NCONE <= HARD_MAX_CONES && E_NROUND <= (E_NCONE > 0 ? E_NCONE : 1) && E_NL <= HARD_MAX_LIN_ROWS, "cone list / linear rows");
// the capacities of the family buffer for THIS structure (tinympc_device.h: the defaults, or the structure's own counts beyond them)
constexpr int E_LIN_CAP = fam_lin_cap(E_NL), E_CONE_CAP = fam_cone_cap(E_NCONE);


// The families' per-lane data and their row-local evaluation for ONE (row, knot) element (admm.cpp's update_slack / update_dual /
// update_linear_cost pattern on the families' own slack and dual).
template <int NX, int NU>
struct EFamilies {
    static constexpr int W = 16, NR = E_NROUND > 0 ? E_NROUND : 1, NC = E_NCONE > 0 ? E_NCONE : 1;
    int role_r[NR];
    double mu_r[NR], imu_r[NR];
    unsigned long long cmask[NC];
    unsigned long long mask_x, mask_u;
    bool famc, faml;
    double one, rho;
    const double *sLin;  // LDS: [E_NL][3][16]  a_k | b_k | 1/||a_k||^2
    int r;

    // fam: the family buffer (fam_doubles()); KT: its row stride; lin: the LDS copy of the linear rows (built by the caller)
    __device__ __forceinline__ void init(const double *fam, int KT, const double *lin, int lane_r, double rho_) {
        r = lane_r;
        rho = rho_;
        one = 1.0;
        sLin = lin;
        mask_x = __ballot(r < NX);
        mask_u = __ballot(r >= NX && r < NX + NU);
        famc = fam[2 * W + r] != 0.0;
        faml = fam[3 * W + r] != 0.0;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            role_r[q] = 0;
            mu_r[q] = 0.0;
            imu_r[q] = 0.0;
        }
        const double *cone_mu = fam + fam_cone_mu_offset(W, KT, E_LIN_CAP);
        e_static_for<0, E_NCONE>([&](auto Cc) {
            constexpr EConeDesc cd = E_CONES[Cc.value];
            const bool in = (r >= cd.first) && (r <= cd.last);
            cmask[Cc.value] = __ballot(in);
            const double mu = cone_mu[Cc.value];  // (uniform)
            if (in) {
                role_r[cd.round] = (r == cd.last) ? 2 : 1;
                mu_r[cd.round] = mu;
                imu_r[cd.round] = 1.0 / mu;
            }
        });
    }
    // returns the element's contribution to the linear cost and the new duals. Every enabled lane takes part; lanes outside every
    // cone / of a side without linear rows fall through unchanged.
    __device__ __forceinline__ double eval(double val, double gc_old, double gl_old, double &gc_new, double &gl_new) const {
        double lxv = 0.0;
        gc_new = gc_old;
        gl_new = gl_old;
        if constexpr (E_NCONE > 0) {
            const double s_in = val + gc_old;
            double sv = s_in;
            e_static_for<0, E_NROUND>([&](auto R) {
                const double w = sv * sv;
                double a2 = 0.0, t = 0.0;
                e_static_for<0, E_NCONE>([&](auto Cc) {
                    constexpr EConeDesc cd = E_CONES[Cc.value];
                    if constexpr (cd.round == R.value)
                        e_masked_gather<cd.first, cd.last - cd.first, true, cd.last>(cmask[Cc.value], w, sv, one, a2, t);
                });
                sv = soc_project_element(sv, a2, t, mu_r[R.value], imu_r[R.value], role_r[R.value]);
            });
            const double gcn = s_in - sv;
            if (famc) {
                gc_new = gcn;
                lxv -= rho * (sv - gcn);
            }
        }
        if constexpr (E_NL > 0) {
            const double s_in = val + gl_old;
            double sv = s_in;
            auto row = [&](int k) {
                const double a_k = sLin[(3 * k + 0) * W + r], b_k = sLin[(3 * k + 1) * W + r], in_k = sLin[(3 * k + 2) * W + r];
                const double w = a_k * sv;
                double dot = 0.0, unused = 0.0;
                if (k < E_NLX) e_masked_gather<0, NX, false, 0>(mask_x, w, w, one, dot, unused);
                if (k < E_NLU) e_masked_gather<NX, NU, false, 0>(mask_u, w, w, one, dot, unused);
                sv = halfspace_project_element(sv, dot, a_k, b_k, in_k);
            };
            if constexpr (E_NL <= 4) {  // a few rows: straight-line code (the row index folds into the branches above)
                e_static_for<0, E_NL>([&](auto K) { row(K.value); });
            } else {                    // many rows (an equality constraint counts twice): one copy of the row's code
#pragma unroll 1
                for (int k = 0; k < E_NL; ++k) row(k);
            }
            const double gln = s_in - sv;
            if (faml) {
                gl_new = gln;
                lxv -= rho * (sv - gln);
            }
        }
        return lxv;
    }
    // the LDS copy of the linear rows' coefficients, by all threads of the workgroup (a barrier must follow)
    static __device__ __forceinline__ void stage_linear_rows(const double *fam, int KT, double *lin, int tid, int nthreads) {
        if constexpr (E_NL > 0) {  // layout of fam_doubles(): ... | nl | per linear row k: a_k[W] b_k[W] ||a_k||^2[W]
            const double *lin_rows = fam + 4 * W + (size_t)3 * W * KT;
            for (int i = tid; i < 3 * E_NL * 16; i += nthreads) {
                const int k3 = i / W, rr = i % W;
                const double v = lin_rows[1 + (size_t)k3 * W + rr];
                lin[i] = (k3 % 3 == 2) ? 1.0 / v : v;  // 1 / ||a_k||^2
            }
        }
    }
};

// ---- the families, one KNOT per lane ("transposed" evaluation; layout E, round 4) ------------------------------------------------
// EFamilies above evaluates ONE (row, knot) element per lane: every slot of a sweep pays the whole evaluation -- ~70 FP64
// instructions for the rocket landing, half of layout E's iteration (profiles/r04_e_breakdown.txt) -- for the 4 x (nx+nu) real
// elements of its 64 lanes, and needs cross-lane gathers for every norm and dot product. But the families are local to a KNOT:
// a cone couples rows of one knot, a half-space likewise. KFamilies turns the data round: after the forward sweep has left the
// wavefront's S slots of x | u in an LDS buffer, lane (instance j, entry t) picks up ALL nx+nu rows of its knot into registers and
// evaluates every cone and every linear row of that knot in plain per-lane arithmetic -- no DPP, no EXEC masks, one square root per
// cone and knot instead of one per lane -- ONCE per ADMM iteration for all slots of the wavefront at the same time (4 x (S+1) of
// the 64 lanes busy), and hands the families' linear-cost term back through the same buffer. The families' duals gc, gl are only
// ever touched here, so they LIVE in this layout (2 x (nx+nu) register pairs per lane) for the whole solve.
// Entry t of a wavefront: t = 0 is knot 0 of the state rows (bottom wavefront only; x_0, its duals persist, its linear-cost term
// reaches nothing); t >= 1 is slot t-1, i.e. state rows of knot s0+t and input rows of knot s0+t-1 (tinympc_solve_e.hip).
// Semantics are EFamilies' (= upstream's update_slack / update_dual / update_linear_cost pattern, PARITY UNPINNED): rows of an
// enabled side take part whether or not a cone contains them; cones of one round at once, rounds one after another; linear rows
// one after another, row k of the state side and row k of the input side independently.
template <int NX, int NU>
struct KFamilies {
    static constexpr int W = 16, NXU = NX + NU;
    double gc[NXU], gl[NXU];  // the families' duals of this lane's knot
    double rho;
    bool cone_x, cone_u, lin_x, lin_u;  // (uniform) the family is enabled for the side
    // ... which the structure also says at compile time (it lists the ACTIVE cones and rows only: a side is enabled exactly when it has
    // one). Small structures use that -- a disabled side's rows drop out of the loops below (rocket N=100 x 4,096 on layout E: 2.62 ->
    // 2.52 ms) --; large ones keep the run-time flags (20 cones + 120 rows: the folded form needed 50-80 registers more and spilled).
    static constexpr bool has_cone(bool input_side) {
        for (int c = 0; c < E_NCONE; ++c)
            if ((E_CONES[c].first >= NX) == input_side) return true;
        return false;
    }
    static constexpr bool FOLD = E_NCONE <= 4 && E_NL <= 4;
    static constexpr bool CX = has_cone(false), CU = has_cone(true), LX_ = E_NLX > 0, LU_ = E_NLU > 0;
    const double *sMu;        // LDS: [2][E_NCONE] slope of cone c of the list | its reciprocal (uniform reads: no registers for up to 64 cones)
    const double *sLin;       // LDS: [E_NL][3][16]  a_k | b_k | 1/||a_k||^2, per ROW-layout lane (state lanes: the state side's row k)

    // (by the first E_NCONE threads of the workgroup; a barrier must follow)
    static __device__ __forceinline__ void stage_cone_slopes(const double *fam, int KT, double *mu, int tid) {
        if constexpr (E_NCONE > 0) {
            if (tid < E_NCONE) {
                const double m = fam[fam_cone_mu_offset(W, KT, E_LIN_CAP) + tid];
                mu[tid] = m;
                mu[E_NCONE + tid] = 1.0 / m;
            }
        }
    }
    __device__ __forceinline__ void init(const double *fam, int KT, const double *lin, const double *mu, double rho_) {
        rho = rho_;
        sLin = lin;
        sMu = mu;
        cone_x = fam[2 * W + 0] != 0.0;
        cone_u = fam[2 * W + NX] != 0.0;
        lin_x = fam[3 * W + 0] != 0.0;
        lin_u = fam[3 * W + NX] != 0.0;
        (void)KT;
    }
    // one cone {F .. L}, L its t row, on the knot's slack vector (soc_project_element's arithmetic, once per knot)
    template <int F, int L>
    static __device__ __forceinline__ void project_cone(double (&sv)[NXU], double mu, double inv_mu) {
        double a2 = 0.0;
        e_static_for<F, L>([&](auto K) { a2 = (K.value == F) ? sv[K.value] * sv[K.value] : __builtin_fma(sv[K.value], sv[K.value], a2); });
        const double t = sv[L];
        const double u0 = t * mu;
        const double a2c = fmax(a2, 1e-300);  // a2 = 0 -> the `inside` / `apex` cases below decide, never the quotient
        double y = __builtin_amdgcn_rsq(a2c);
        double g = a2c * y, h = 0.5 * y;
        double rr = fma(-g, h, 0.5);
        g = fma(g, rr, g);
        h = fma(h, rr, h);
        rr = fma(-g, h, 0.5);
        g = fma(g, rr, g);
        h = fma(h, rr, h);
        const double d = fma(-g, g, a2c);
        const double a = fma(d, h, g);   // sqrt(a2)
        const double inv_a = h + h;      // 1 / sqrt(a2)
        const double scale = 0.5 * (1.0 + u0 * inv_a);
        const bool apex = a <= -u0, inside = a <= u0;
        const double tproj = scale * (a * inv_mu);
        e_static_for<F, L>([&](auto K) { sv[K.value] = apex ? 0.0 : (inside ? sv[K.value] : scale * sv[K.value]); });
        sv[L] = apex ? 0.0 : (inside ? t : tproj);
    }
    // val: the knot's x | u rows; lx: the families' contribution to the linear cost of every row (overwritten)
    __device__ __forceinline__ void eval(const double (&val)[NXU], double (&lx)[NXU]) {
        e_static_for<0, NXU>([&](auto R) { lx[R.value] = 0.0; });
        if constexpr (E_NCONE > 0) {
            double sv[NXU];
            e_static_for<0, NXU>([&](auto R) { sv[R.value] = val[R.value] + gc[R.value]; });
            e_static_for<0, E_NROUND>([&](auto Rd) {
                e_static_for<0, E_NCONE>([&](auto Cc) {
                    constexpr EConeDesc cd = E_CONES[Cc.value];
                    if constexpr (cd.round == Rd.value) project_cone<cd.first, cd.last>(sv, sMu[Cc.value], sMu[E_NCONE + Cc.value]);
                });
            });
            e_static_for<0, NXU>([&](auto R) {
                constexpr int r = R.value;
                const bool on = FOLD ? (r < NX ? bool(CX) : bool(CU)) : (r < NX ? cone_x : cone_u);
                const double gcn = (val[r] + gc[r]) - sv[r];
                if (on) {
                    gc[r] = gcn;
                    lx[r] -= rho * (sv[r] - gcn);
                }
            });
        }
        if constexpr (E_NL > 0) {
            double sv[NXU];
            e_static_for<0, NXU>([&](auto R) { sv[R.value] = val[R.value] + gl[R.value]; });
            auto row = [&](int k) {
                const double *ak = sLin + (size_t)(3 * k + 0) * W, *bk = sLin + (size_t)(3 * k + 1) * W, *ik = sLin + (size_t)(3 * k + 2) * W;
                if (k < E_NLX) {
                    double dot = 0.0;
                    e_static_for<0, NX>([&](auto R) { dot = (R.value == 0) ? ak[R.value] * sv[R.value] : __builtin_fma(ak[R.value], sv[R.value], dot); });
                    const double dist = (dot - bk[0]) * ik[0];
                    const bool viol = dot > bk[0];
                    e_static_for<0, NX>([&](auto R) { sv[R.value] = viol ? fma(-dist, ak[R.value], sv[R.value]) : sv[R.value]; });
                }
                if (k < E_NLU) {
                    double dot = 0.0;
                    e_static_for<NX, NXU>([&](auto R) { dot = (R.value == NX) ? ak[R.value] * sv[R.value] : __builtin_fma(ak[R.value], sv[R.value], dot); });
                    const double dist = (dot - bk[NX]) * ik[NX];
                    const bool viol = dot > bk[NX];
                    e_static_for<NX, NXU>([&](auto R) { sv[R.value] = viol ? fma(-dist, ak[R.value], sv[R.value]) : sv[R.value]; });
                }
            };
            if constexpr (E_NL <= 4) {
                e_static_for<0, E_NL>([&](auto K) { row(K.value); });
            } else {
#pragma unroll 1
                for (int k = 0; k < E_NL; ++k) row(k);
            }
            e_static_for<0, NXU>([&](auto R) {
                constexpr int r = R.value;
                const bool on = FOLD ? (r < NX ? bool(LX_) : bool(LU_)) : (r < NX ? lin_x : lin_u);
                const double gln = (val[r] + gl[r]) - sv[r];
                if (on) {
                    gl[r] = gln;
                    lx[r] -= rho * (sv[r] - gln);
                }
            });
        }
    }
};

}  // namespace tinympc
