// tinympc_solve_dwide.h -- layout D for WIDE systems, ONE body for both widths: W = 32 lanes per instance (16 < nx+nu <= 32, two
// instances per wavefront; front end tinympc_solve_dw.hip, chain blocks tinympc_solve_dw_chain.h) and W = 64 (32 < nx+nu <= 64,
// one instance per wavefront; tinympc_solve_dx.hip, tinympc_solve_dx_chain.h). (Round 2 kept two copies of this body, 80 %
// line-identical.)
//
// Same plan as tinympc_solve_d.hip -- horizon a compile-time constant, both sweeps fully unrolled, the duals g|y in registers
// (2*(N-1)+2 VGPRs), the slack v|z split between registers and LDS, LDS otherwise only for the feed-forward d and one copy
// of the two sweep operators per workgroup, <= 256 VGPRs -> two wavefronts per SIMD -- and the same reference semantics
// (per-instance termination, `iter % check_termination`, solution = vnew / znew, stale v|z after a converged solve:
// admm.cpp:109-207). What differs is the mat-vec: an instance spans W / 16 DPP rows, so each step first replicates the operand
// vector across them (cross-row swaps, tinympc_sweep.h) and then runs the fused DPP chain in W / 16 blocks of 16 columns -- the
// width-specific part, behind WideStep<W, NX, NU> (defined by the front ends from their chain headers):
//   Operand                                    the replicated operand vector (2 or 4 registers)
//   replicate(w, op)                           cross-row swaps
//   fwd_head(op, m, cf) -> a                   a = cf + the first blocks of the chain
//   fwd_tail_reg / fwd_tail_lds(a, op, ...)    the last block + the row-local phase (slack in a register / in LDS)
//   bwd(a, op, ...), bwd_last(a, op, m)        the backward step (+ the tail that prepares the next ones)
//
// FAM (round 5, run-time specialised only): the cone / linear-inequality families on wide systems, STREAMED. Their duals gc | gl and the
// linear-cost term lx do not fit next to the box path's state (three more arrays of the slack's size: registers are full, LDS would hold
// them for one wavefront in four), so they stay in HBM / L2 in V's layout: the forward sweep reads a slot's duals WIDE_FAM_AHEAD steps
// ahead of its use (a ring of registers), evaluates the element with the group-reduction form of tinympc_fam_red.h between two chain
// blocks, and stores the new duals and lx; the backward sweep reads lx back the same way, descending. Before this variant the families on
// 32 / 64 lanes ran on k_admm_solve_fam (layout A: state in LDS, ONE wavefront per SIMD): 9.1 ms where the box path takes 1.2 (nx=24,
// nu=8, N=30 x 4,096, one cone + two linear rows; tools/wide_families_probe.py).
#pragma once
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"
#include "tinympc_fam_red.h"

namespace tinympc {

template <int W, int NX, int NU>
struct WideStep;  // specialised per (W, nx, nu) by tinympc_solve_dw.hip / tinympc_solve_dx.hip

namespace {
template <int I, int E, class F>
__device__ __forceinline__ void static_for_w(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for_w<I + 1, E>(f);
    }
}

#ifdef TINY_WIDE_GROUP
constexpr int WIDE_GROUP = TINY_WIDE_GROUP;  // (experiments)
constexpr int WIDE_FIRST = TINY_WIDE_FIRST;
#else
constexpr int WIDE_GROUP = 8;        // forward steps between two "can this sweep still converge" tests
constexpr int WIDE_FIRST = 8;        // ... and before the first one inside the sweep (after the one on knot 0)
#endif
constexpr int WIDE_LDS_PER_CU = 160 * 1024;
#ifdef TINY_WIDE_FAM_AHEAD
constexpr int WIDE_FAM_AHEAD = TINY_WIDE_FAM_AHEAD;  // (experiments)
#else
constexpr int WIDE_FAM_AHEAD = 4;    // FAM: sweep steps between the request of a slot's duals / lx and their use
#endif
#ifdef TINY_WIDE_FAM_BATCH
constexpr int WIDE_FAM_BATCH = TINY_WIDE_FAM_BATCH;  // (experiments)
#else
constexpr int WIDE_FAM_BATCH = 4;    // FAM: slots whose families are evaluated together (measured 1 .. 6: profiles/r05_wide_families.txt)
#endif
#ifndef TINY_WIDE_FAM_FENCE
#define TINY_WIDE_FAM_FENCE 1        // FAM: scheduling fences around a step's families block (experiments: 0)
#endif
// ---- LDS plan per workgroup, in doubles: operators [2][W k][W r] | tables (!ct) | per wave: V[VL][64], D[(N-1) * (64 / W) * nu]
__host__ __device__ constexpr int wide_ops_doubles(int W) { return 2 * W * W; }
__host__ __device__ constexpr int wide_d_doubles(int W, int nu, int N) { return ((N - 1) * (64 / W) * nu + 1) & ~1; }
__host__ __device__ constexpr int wide_tab_doubles(int W, int N) { return 3 * (N + 2) * W + W; }  // the workgroup's copy of the per-knot tables (!CT)
// number of slack slots in LDS; -1 if the shape does not fit the plan (cu_waves wavefronts per CU: 8, or 4 with 512 registers each)
__host__ __device__ constexpr int wide_vl(int W, int vreg_max, int nu, int N, bool ct, int wpg, int cu_waves = 8) {
    const int ns = N - 1;
    const int wg_doubles = WIDE_LDS_PER_CU / 8 * wpg / cu_waves - wide_ops_doubles(W) - (ct ? 0 : wide_tab_doubles(W, N));
    const int wave_doubles = wg_doubles / wpg - wide_d_doubles(W, nu, N);
    if (wave_doubles < 0) return -1;
    const int vlmax = wave_doubles / 64;
    const int want = ns > vreg_max ? ns - vreg_max : 0;
    return want <= vlmax ? want : -1;
}
__host__ __device__ constexpr size_t wide_lds_bytes(int W, int nu, int N, bool ct, int wpg, int vl) {
    return sizeof(double) * ((size_t)wide_ops_doubles(W) + (ct ? 0 : wide_tab_doubles(W, N)) + (size_t)wpg * (vl * 64 + wide_d_doubles(W, nu, N)));
}

// FAM: the workgroup's LDS copy of the linear rows (behind everything else): nl | per row k: a_k[W] | b_k[W] | 1 / ||a_k||^2 [W] -- the family
// buffer's own layout (tinympc_handle.hip: refresh_families) with the RECIPROCAL of the norm, so that a sweep step neither waits for L2 nor divides
__host__ __device__ constexpr size_t wide_fam_lin_doubles(int W) { return (size_t)2 + (size_t)3 * MAX_LIN_ROWS * W; }

__device__ __forceinline__ unsigned lds_addr_w(const double *p) { return lds_address(p); }
template <int OFF>
__device__ __forceinline__ double lds_read_async_w(unsigned addr) {  // issued HERE, tracked by the compiler (tinympc_sweep.h); valid after the next lds_reads_landed()
    return lds_read_issued_here<OFF>(addr);
}
template <int OFF>
__device__ __forceinline__ void lds_write_async_w(unsigned addr, double v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write_masked_w(unsigned addr, double v, unsigned long long mask) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "ds_write_b64 %[a], %[v] offset:%[o]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved)
                 : [m] "s"(mask), [a] "v"(addr), [v] "v"(v), [o] "n"(OFF)
                 : "memory", "scc");
}
__device__ __forceinline__ void lds_wait_w() {
    lds_reads_landed();
    asm volatile("" ::: "memory");
}

// `bad` = ballot of lanes whose row already rules out convergence in this sweep, `live` = ballot of the lanes that
// are still iterating. True if some live instance (W lanes) has no bad lane.
template <int W>
__device__ __forceinline__ bool wave_may_converge_w(unsigned long long bad, unsigned long long live) {
    constexpr unsigned long long ones = (W == 64) ? ~0ull : ((1ull << (W % 64)) - 1ull);
    bool any = false;
#pragma unroll
    for (int j = 0; j < 64 / W; ++j) {
        const unsigned long long b = (bad >> (j * W)) & ones, l = (live >> (j * W)) & ones;
        any = any || (l != 0ull && b == 0ull);
    }
    return any;
}
}  // namespace

template <int W, int NX, int NU, int N, bool CT, int WPG, int VL, bool FAM = false>
__device__ __forceinline__ void k_admm_solve_wide_body(const SolveParams &p, double *smem) {
    constexpr int IPW = 64 / W, NXU = NX + NU, NS = N - 1, DS = IPW * NU, NVR = NS - VL;
    constexpr int KT = W;  // row stride of p.ops (choose_geometry)
    constexpr int TOFF = (N + 2) * W;
    static_assert((W == 32 || W == 64) && NS >= 3 && VL >= 0 && VL <= NS && NXU > W / 2 && NXU <= W, "wide layout D: N >= 4, W / 2 < nx+nu <= W");
    using Step = WideStep<W, NX, NU>;

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane / W, r = lane % W;
    const long grp = (long)blockIdx.x * WPG + wv;
    const bool grp_ok = grp < p.groups;
    const long inst = grp * IPW + j;
    const bool is_x = r < NX;
    const bool is_u = (r >= NX) && (r < NXU);
    const bool inst_ok = grp_ok && inst < p.batch;
    const int koff = is_x ? 1 : 0;  // slot s = knot s+1 on state lanes, knot s on input lanes

    double *sOps = smem;
    double *sT = smem + wide_ops_doubles(W);
    double *sV = sT + (CT ? 0 : wide_tab_doubles(W, N)) + (size_t)wv * (VL * 64 + wide_d_doubles(W, NU, N));
    double *sD = sV + VL * 64;

    // ---- workgroup-shared: the two sweep operators, transposed to [k][r] (conflict-free row reads), and the tables
    for (int i = threadIdx.x; i < wide_ops_doubles(W); i += 64 * WPG) {
        const int which = i / (W * W), k = (i / W) % W, rr = i % W;
        sOps[i] = p.ops[(size_t)which * W * KT + (size_t)rr * KT + k];
    }
    if constexpr (!CT)
        for (int i = threadIdx.x; i < wide_tab_doubles(W, N); i += 64 * WPG) sT[i] = p.tables[i];

    const size_t g0 = grp_ok ? (size_t)grp : 0;
    double *const gG = p.G + g0 * (N + 1) * 64 + lane;                 // row kn = knot kn
    double *const gD = p.D + g0 * (size_t)(NS * DS);
    double *const gV0 = p.V + (g0 * v_rows(N) + V_PAD) * 64 + lane;    // canonical v|z, knot 0
    double *const gV1u = p.V2 + (g0 * v_rows(N) + V_PAD) * 64;         // stale copy, knot 0 (wave-uniform: scalar base + 32-bit lane offset)
    const unsigned voff = (unsigned)(lane + koff * 64);
    double *const sVl = sV + lane;
    const bool cold = p.cold != 0;  // (uniform) the state is zero by contract and was never written to HBM (SolveParams::cold)
    if (grp_ok) {
        if (cold) {
            for (int i = lane; i < NS * DS; i += 64) sD[i] = 0.0;
            static_for_w<0, VL>([&](auto S) { sVl[S.value * 64] = 0.0; });
        } else {
            for (int i = lane; i < NS * DS; i += 64) sD[i] = gD[i];
            static_for_w<0, VL>([&](auto S) { sVl[S.value * 64] = gV0[(S.value + koff) * 64]; });
        }
    }
    __syncthreads();  // the only workgroup-wide barrier: from here on the waves are independent
    if (!grp_ok) return;

    // ---- register-resident state
    double G[NS], G0, Vr[NVR > 0 ? NVR : 1], V0;
    if (cold) {
        static_for_w<0, NS>([&](auto S) { G[S.value] = 0.0; });
        static_for_w<0, NVR>([&](auto S) { Vr[S.value] = 0.0; });
        G0 = 0.0;
        V0 = 0.0;
    } else {
        static_for_w<0, NS>([&](auto S) { G[S.value] = gG[(S.value + koff) * 64]; });
        static_for_w<0, NVR>([&](auto S) { Vr[S.value] = gV0[(VL + S.value + koff) * 64]; });
        G0 = gG[0];
        V0 = gV0[0];
    }

    const double cf = p.ops[(size_t)2 * W * KT + r];
    const double cb = p.ops[(size_t)2 * W * KT + W + r];
    const double pnref = p.tables[(size_t)3 * TOFF + r];
    const double rho = p.rho, nrho = -p.rho;
    const double lo_c = p.tables[W + r], hi_c = p.tables[(size_t)TOFF + W + r], lr_c = p.tables[(size_t)2 * TOFF + W + r];
    const double rhom = is_x ? nrho : 0.0;
    const double x0v = (inst_ok && is_x) ? p.x0[inst * NX + r] : 0.0;
    const int dIdx = j * NU + (is_u ? r - NX : 0);
    const double *const sDr = sD + dIdx;
    double *const sDw = sD + dIdx;
    const double *const sTl = sT + koff * W + r;  // (!CT) row of slot s: sTl[(s + 1) * W]
    const double *const sMf = sOps + r, *const sMb = sOps + W * W + r;
    const unsigned aV = lds_addr_w(sVl), aD = lds_addr_w(sDr), aT = lds_addr_w(sTl);
    const int ct = p.check_termination;
    // FAM: the families' evaluation (group reductions) and the HBM arrays of their duals / linear-cost term, slot s <-> row s + koff like V
    RedFamilies<W> fam;
    constexpr int FB = WIDE_FAM_BATCH;                                                                 // slots whose families are evaluated together
    constexpr int PD0 = (WIDE_FAM_AHEAD > FB ? WIDE_FAM_AHEAD : FB), PD = (PD0 < NS) ? PD0 : NS;      // ring depth (>= a batch: its duals are all in the ring)
    // (wave-uniform bases in SGPRs + ONE 32-bit per-lane offset, `voff` = lane + koff * 64, for every row of the three arrays: with per-lane
    // 64-bit pointers the compiler kept an address pair per (array, row) -- 170 VGPRs -- and the kernel spilled)
    double *const bGC = FAM ? p.GC + (g0 * v_rows(N) + V_PAD) * 64 : nullptr;
    double *const bGL = FAM ? p.GL + (g0 * v_rows(N) + V_PAD) * 64 : nullptr;
    double *const bLX = FAM ? p.LX + (g0 * v_rows(N) + V_PAD) * 64 : nullptr;
    if constexpr (FAM) {
        // (the rows' coefficients: staged into LDS once, reciprocal norms computed here; every wavefront of the workgroup stages the same
        // values -- no barrier needed beyond the wavefront's own program order... but other wavefronts read them too: written identically)
        double *const sLinW = smem + wide_lds_bytes(W, NU, N, CT, WPG, VL) / sizeof(double);
        const double *const gl = p.fam + 4 * W + (size_t)3 * W * KT;
        const int nl_ = (int)gl[0];
        for (int i = lane; i < 1 + 3 * nl_ * W; i += 64) {
            double v = gl[i];
            if (i >= 1 && ((i - 1) / W) % 3 == 2) v = 1.0 / v;  // 1 / ||a_k||^2
            sLinW[i] = v;
        }
        lds_wait_w();  // (this wavefront's LDS writes have landed before its first read)
        fam.init(p.fam, NXU, r, is_x, is_u, rho, sLinW);
    }
    const bool row_real = r < NXU;

    // Control: an instance that converges stops being `active` but its lanes keep iterating as a zombie (the sweeps are
    // unconditional for all 64 lanes -- no EXEC-masked region around the unrolled body). Its state is written back at
    // the top of the next round, before the next forward sweep touches G and V; the backward sweep in between leaves
    // G and V alone and skips a zombie's d. Instances that hit max_iter are written back by the same code in round
    // `max_iter`, which does nothing else.
    bool active = inst_ok;
    bool pending = false;  // converged in the previous round: state not yet written back
    int it_done = 0;
    int status = 11;  // TINY_UNSOLVED (admm.cpp:114)
    bool res_valid = false;
    double snap_pri = 0.0, snap_dua = 0.0;

    auto load_ops = [&](const double *src, double (&m)[W]) {
        static_for_w<0, W>([&](auto K) { m[K.value] = src[(K.value < NXU ? K.value : 0) * W]; });
    };
    auto vget = [&](auto S) -> double {
        if constexpr (decltype(S)::value >= VL) return Vr[decltype(S)::value - VL];
        else return sVl[decltype(S)::value * 64];
    };

    const int simd_slot = simd_slot_id();
    const int max_iter = p.max_iter;
    for (int it = 0; max_iter > 0; ++it) {  // admm.cpp:129
        // (readfirstlane: keeps the loop counter and everything derived from it in SGPRs, so that the branches below
        // are scalar branches and not EXEC-masked regions)
        const int it0 = __builtin_amdgcn_readfirstlane(it);
        const bool final_round = it0 >= max_iter;
        fair_share_priority<(W / 16) * NS>(it0, simd_slot);  // (tinympc_sweep.h: the two wavefronts of a SIMD finish together)
        // ---- write-back: G, D and the canonical v|z (not converged: v = vnew, admm.cpp:196-197; converged: the solve
        // returned before v <- vnew, so the canonical copy is the stale one in V2); solution = vnew / znew (:187-188, 204-205)
        const bool wb = pending || (final_round && active);
        if (__ballot(wb) != 0ull) {
            // Rare path (once per instance and solve), kept small in registers rather than fast: addresses are rebuilt
            // here from the kernel arguments (the opaque copy of `lane` keeps the compiler from hoisting them out of
            // the iteration loop, where they would occupy registers the unrolled sweeps need).
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int r_o = lane_o % W, j_o = lane_o / W;
            const bool x_o = r_o < NX;
            if (wb && r_o < NXU) {
                const int ko = x_o ? 1 : 0;
                const size_t inst_o = (size_t)grp * IPW + j_o;
                double *const wG = p.G + (size_t)grp * (N + 1) * 64 + lane_o + ko * 64;                // slot 0
                double *const wV = p.V + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane_o + ko * 64;   // slot 0
                double *const wS = x_o ? p.sol_x + (inst_o * N + 1) * NX + r_o : p.sol_u + inst_o * NS * NU + (r_o - NX);  // slot 0
                const int sst = x_o ? NX : NU;
                if (x_o) {  // knot 0
                    wG[-64] = G0;
                    wV[-64] = V0;
                    wS[-NX] = V0;
                }
                static_for_w<0, NS>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    const double vn = vget(S);
                    wG[s * 64] = G[s];
                    wV[s * 64] = vn;
                    wS[s * sst] = vn;
                });
                if (!x_o) {
                    double *const wD = p.D + (size_t)grp * (NS * DS) + j_o * NU + (r_o - NX);
                    for (int i = 0; i < NS; ++i) wD[i * DS] = sDw[i * DS];
                }
            }
            pending = false;
        }
        if (final_round || __ballot(active) == 0ull) break;
        const int it1 = it0 + 1;
        const bool check = __builtin_amdgcn_readfirstlane((int)((ct > 0) && ((it1 % ct) == 0))) != 0;  // admm.cpp:91 (iter already incremented, :143)

        double pri = 0.0, dua = 0.0;
        bool may = check;  // wave-uniform: can this sweep still end converged for some instance of the wave?
        double m[W];
        load_ops(sMf, m);
        // ---------------- knot 0, state lanes: x_0 is given (tiny_set_x0), no mat-vec
        {
            const double lo0 = CT ? lo_c : sT[W + r], hi0 = CT ? hi_c : sT[TOFF + W + r];
            const double s = x0v + G0;
            const double snew = fmin(hi0, fmax(lo0, s));
            G0 = s - snew;
            pri = is_x ? fabs(x0v - snew) : 0.0;
            dua = is_x ? fabs(V0 - snew) : 0.0;
            if (may) {  // first test on knot 0 alone (tinympc_solve_d.hip): forced iteration counts write no stale copy
                const bool bad = !((pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol));
                may = __builtin_amdgcn_readfirstlane((int)wave_may_converge_w<W>(__ballot(bad), __ballot(active))) != 0;
            }
            if (may && is_x && active) gV1u[(unsigned)lane] = V0;  // (active: as for the slots below)
            V0 = snew;
        }
        // FAM: the first WIDE_FAM_AHEAD slots' duals are requested now, knot 0 of the state rows is evaluated while they travel (its lx only
        // reaches p_0, which nothing reads; the duals persist)
        double gcr[FAM ? PD : 1], glr[FAM ? PD : 1], xel[FAM ? FB : 1];
        if constexpr (FAM) {
            const double gc0 = bGC[(unsigned)lane], gl0 = bGL[(unsigned)lane];
            static_for_w<0, PD>([&](auto Q) {
                gcr[Q.value] = (bGC + Q.value * 64)[voff];
                glr[Q.value] = (bGL + Q.value * 64)[voff];
            });
            double gcn, gln;
            (void)fam.eval(x0v, gc0, gl0, gcn, gln);
            if (is_x && active) {
                bGC[(unsigned)lane] = gcn;
                bGL[(unsigned)lane] = gln;
            }
        }
        // ---------------- forward sweep (F1) with S1 + D1 + R1 fused in
        // LDS operands of a step (its d, and vold of its slot if that lives in LDS) are requested right before the
        // PREVIOUS step's block and retired by the wait behind that block (lds_reads_landed, inside the Step functions).
        double xcur = x0v;
        double dcur = lds_read_async_w<0>(aD), vcur = 0.0;
        if constexpr (VL > 0) vcur = lds_read_async_w<0>(aV);
        // (!CT: bounds that vary over the horizon come from the workgroup's LDS copy of the tables, one step ahead like d)
        double locur = lo_c, hicur = hi_c;
        if constexpr (!CT) {
            locur = lds_read_async_w<W * 8>(aT);
            hicur = lds_read_async_w<(TOFF + W) * 8>(aT);
        }
        lds_wait_w();
        auto fstep = [&](auto S) {
            constexpr int q = decltype(S)::value;
            double dn = 0.0, vn = 0.0, lon = lo_c, hin = hi_c;
            if constexpr (q + 1 < NS) dn = lds_read_async_w<(q + 1) * DS * 8>(aD);
            if constexpr (q + 1 < VL) vn = lds_read_async_w<(q + 1) * 512>(aV);
            if constexpr (!CT && q + 1 < NS) {
                lon = lds_read_async_w<(q + 2) * W * 8>(aT);
                hin = lds_read_async_w<(TOFF + (q + 2) * W) * 8>(aT);
            }
            // operand vector [x_q; d_q]: one entry per lane, replicated across the instance's two DPP rows
            typename Step::Operand op;
            Step::replicate(is_x ? xcur : dcur, op);  // the operand vector on every DPP row of the instance
            double a = Step::fwd_head(op, m, cf);
            if constexpr (q >= VL) {
                Step::fwd_tail_reg(a, op, m, locur, hicur, G[q], Vr[q - VL], pri, dua);
            } else {
                double vnew;
                Step::fwd_tail_lds(a, op, m, locur, hicur, G[q], vcur, vnew, pri, dua);
                lds_write_async_w<q * 512>(aV, vnew);
            }
            xcur = a;
            if constexpr (FAM) {  // xcur: x_{q+1} on state lanes, u_q on input lanes -- this slot's element
                // The element waits in a small batch: the families of WIDE_FAM_BATCH slots are evaluated TOGETHER, as independent instruction
                // streams in one block -- a lone wavefront per SIMD (the variant's 470 registers) waits ~8 cycles for every dependent FP64
                // result and ~100 for every cross-row exchange of a reduction; with several elements in flight those waits overlap.
                xel[q % FB] = xcur;
                if constexpr ((q % FB) == FB - 1 || q == NS - 1) {
                    constexpr int q0 = q - (q % FB);
                    if constexpr (TINY_WIDE_FAM_FENCE != 0) __builtin_amdgcn_sched_barrier(0);
                    double gco[FB], glo[FB], gcn[FB], gln[FB], lxn[FB];
                    static_for_w<q0, q + 1>([&](auto T) {  // take the batch's duals out of the ring and request the next ones BEFORE the evaluation:
                        constexpr int t = decltype(T)::value;  // they travel while it runs
                        gco[t - q0] = gcr[t % PD];
                        glo[t - q0] = glr[t % PD];
                        if constexpr (t + PD < NS) {
                            gcr[t % PD] = (bGC + (t + PD) * 64)[voff];
                            glr[t % PD] = (bGL + (t + PD) * 64)[voff];
                        }
                    });
                    if constexpr (q - q0 + 1 == FB) {
                        double xb[FB];
                        static_for_w<0, FB>([&](auto T) { xb[T.value] = xel[(q0 + T.value) % FB]; });
                        fam.template eval_batch<FB>(xb, gco, glo, gcn, gln, lxn);
                    } else {  // (the sweep's last, shorter batch)
                        static_for_w<q0, q + 1>([&](auto T) {
                            constexpr int t = decltype(T)::value;
                            lxn[t - q0] = fam.eval(xel[t % FB], gco[t - q0], glo[t - q0], gcn[t - q0], gln[t - q0]);
                        });
                    }
                    static_for_w<q0, q + 1>([&](auto T) {
                        constexpr int t = decltype(T)::value;
                        if (active && row_real) {  // (a zombie's duals stay what they were when it converged)
                            (bGC + t * 64)[voff] = gcn[t - q0];
                            (bGL + t * 64)[voff] = gln[t - q0];
                            (bLX + t * 64)[voff] = lxn[t - q0];
                        }
                    });
                    if constexpr (TINY_WIDE_FAM_FENCE != 0) __builtin_amdgcn_sched_barrier(0);
                }
            }
            dcur = dn;
            vcur = vn;
            if constexpr (!CT) {
                locur = lon;
                hicur = hin;
            }
        };
        constexpr int WF = WIDE_FIRST < NS ? WIDE_FIRST : NS;
        constexpr int NG = 1 + (NS - WF + WIDE_GROUP - 1) / WIDE_GROUP;
        static_for_w<0, NG>([&](auto Gi) {
            constexpr int s0 = Gi.value == 0 ? 0 : WF + (Gi.value - 1) * WIDE_GROUP;
            constexpr int s1 = Gi.value == 0 ? WF : ((s0 + WIDE_GROUP < NS) ? s0 + WIDE_GROUP : NS);
            if (may) {
                // Stale copy of the group's slots (still holding the previous iterate) before the blocks overwrite them.
                // Rare path: the addresses are rebuilt from an opaque copy of the lane offset so that the compiler does
                // not keep one pointer per slot alive across the iteration loop.
                if constexpr (s0 > 0) {
                    const bool bad = !((pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol));
                    may = __builtin_amdgcn_readfirstlane((int)wave_may_converge_w<W>(__ballot(bad), __ballot(active))) != 0;
                }
                if (may) {
                    unsigned vo = voff;
                    double *base = gV1u;
                    asm volatile("" : "+v"(vo), "+s"(base));
                    // (only for instances that are still iterating: a converged instance's stale copy -- its canonical v|z,
                    // admm.cpp:181-197 -- must survive the later sweeps of the wavefront's other instance; tinympc_solve_d.hip)
                    if (active) static_for_w<s0, s1>([&](auto S) { (base + S.value * 64)[vo] = vget(S); });
                }
            }
            static_for_w<s0, s1>([&](auto S) { fstep(S); });
        });
        if (active) it_done = it1;  // admm.cpp:143

        // ---------------- R1: termination (admm.cpp:93-101), decided element-wise: one ballot, no reductions
        if (check) {
            const bool below = (pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol);
            constexpr unsigned long long ones = (W == 64) ? ~0ull : ((1ull << (W % 64)) - 1ull);
            const bool conv = ((__ballot(below) >> (j * W)) & ones) == ones;
            if (active) {
                snap_pri = pri;
                snap_dua = dua;
                res_valid = true;
                if (conv) {
                    status = 1;  // TINY_SOLVED: this instance stops before the backward pass (admm.cpp:181-192)
                    active = false;
                    pending = true;
                }
            }
        }

        // ---------------- backward sweep (B1, admm.cpp:13-20); linear cost (L1, :77-82) recomputed from V, G
        {
            const unsigned long long wr_d = __ballot(is_u && active);  // a zombie keeps the d of its last real iteration
            load_ops(sMb, m);
            // FAM: the families' term of a slot's linear cost, read back from HBM in the order the sweep consumes it -- slots NS-1, NS-2 (the
            // head), then NS-3 ... 0 -- through a ring requested WIDE_FAM_AHEAD slots ahead (this wavefront stored them in its forward sweep:
            // same lane, same address, program order)
            double lxr[FAM ? PD : 1];
            if constexpr (FAM) static_for_w<0, PD>([&](auto Q) { lxr[Q.value] = (bLX + (NS - 1 - Q.value) * 64)[voff]; });
            auto lx_of = [&](auto S) -> double {  // (every slot exactly once, descending; slot 0 a second time for the last block, whose tail is unused)
                constexpr int sl = decltype(S)::value, idx = (NS - 1 - sl) % PD;
                const double v = lxr[idx];
                if constexpr (sl - PD >= 0) lxr[idx] = (bLX + (sl - PD) * 64)[voff];
                return v;
            };
            auto lr_of = [&](auto S) -> double {  // linref of slot S (its knot differs by lane type) (+ the families' term)
                double base;
                if constexpr (CT) base = lr_c;
                else base = sTl[2 * TOFF + (S.value + 1) * W];
                return base;
            };
            double px, rcur, rnext, acc;
            {   // p_{N-1} (state lanes, admm.cpp:81-82) | r_{N-2} (input lanes) share slot NS-1; then slot NS-2
                double lrT = is_x ? pnref : lr_of(std::integral_constant<int, NS - 1>{});
                double lr2 = lr_of(std::integral_constant<int, NS - 2>{});
                if constexpr (FAM) {
                    lrT += lx_of(std::integral_constant<int, NS - 1>{});
                    lr2 += lx_of(std::integral_constant<int, NS - 2>{});
                }
                const double lrmc2 = is_x ? lr2 + cb : cb;
                const double v1 = vget(std::integral_constant<int, NS - 1>{}), v2 = vget(std::integral_constant<int, NS - 2>{});
                double t;
                asm("v_add_f64 %[t], %[v1], -%[g1]\n\t"
                    "v_fma_f64 %[px], %[nrho], %[t], %[lrT]\n\t"
                    "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                    "v_fma_f64 %[acc], %[rhom], %[t], %[lrmc]\n\t"
                    "v_fma_f64 %[rn], %[nrho], %[t], %[lr]"
                    : [t] "=&v"(t), [px] "=&v"(px), [acc] "=&v"(acc), [rn] "=&v"(rnext)
                    : [v1] "v"(v1), [g1] "v"(G[NS - 1]), [v2] "v"(v2), [g2] "v"(G[NS - 2]), [nrho] "s"(nrho), [lrT] "v"(lrT),
                      [rhom] "v"(rhom), [lrmc] "v"(lrmc2), [lr] "v"(lr2));
                rcur = px;
            }
            // slack operand of a block's tail: a register, or an LDS read issued one block ahead
            auto vreq = [&](auto S) -> double {
                if constexpr (decltype(S)::value >= VL) return Vr[decltype(S)::value - VL];
                else return lds_read_async_w<decltype(S)::value * 512>(aV);
            };
            double v2cur = vreq(std::integral_constant<int, (NS >= 3 ? NS - 3 : 0)>{});
            lds_wait_w();
            static_for_w<0, NS - 1>([&](auto I) {
                constexpr int s = NS - 1 - I.value;           // NS-1 .. 1
                constexpr int s2 = s >= 2 ? s - 2 : 0;        // slot feeding the tail (s = 1: any finite t will do)
                constexpr int s3 = s >= 3 ? s - 3 : 0;        // ... of the next block
                // (an asynchronous read MUST be consumed after its wait: the destination of a dead one would be handed to
                // the block's outputs while the read is still in flight)
                double v2n = 0.0;
                if constexpr (s >= 2) v2n = vreq(std::integral_constant<int, s3>{});
                double lr2 = lr_of(std::integral_constant<int, s2>{});
                if constexpr (FAM && s >= 2 && s <= NS - 1 && (s - 2) <= NS - 3) lr2 += lx_of(std::integral_constant<int, s2>{});  // (s = 1: the tail it feeds is unused)
                const double lrmc2 = is_x ? lr2 + cb : cb;
                double a = acc, an, rn;
                typename Step::Operand op;
                Step::replicate(is_x ? px : rcur, op);  // [p_{s+1}; r_s]
                Step::bwd(a, op, m, v2cur, G[s2], rhom, lrmc2, nrho, lr2, an, rn);
                lds_write_masked_w<s * DS * 8>(aD, a, wr_d);  // d_s
                px = a;
                rcur = rnext;
                rnext = rn;
                acc = an;
                v2cur = v2n;
            });
            {
                double a = acc;
                typename Step::Operand op;
                Step::replicate(is_x ? px : rcur, op);
                Step::bwd_last(a, op, m);
                lds_write_masked_w<0>(aD, a, wr_d);  // d_0
            }
        }
    }
    lds_wait_w();

    // A converged solve returned before v <- vnew (admm.cpp:181-197): its canonical v|z is the previous iterate, i.e. the
    // stale copy. (The write-back above stored vnew there; this wave wrote both, in program order.)
    if (inst_ok && status == 1 && r < NXU) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const int rows = is_x ? N : NS;
        double *const wV = p.V + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane;
        const double *const wV2 = p.V2 + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane;
        for (int kn = 0; kn < rows; ++kn) wV[kn * 64] = wV2[kn * 64];
    }

    const double res_px = group_max<W>(is_x ? snap_pri : 0.0), res_pu = group_max<W>(is_u ? snap_pri : 0.0);
    const double res_dx = group_max<W>(is_x ? snap_dua : 0.0) * rho, res_du = group_max<W>(is_u ? snap_dua : 0.0) * rho;

    if (inst_ok && r == 0) {
        p.istats[inst * 2 + 0] = it_done;
        p.istats[inst * 2 + 1] = status;
        if (res_valid) {
            p.dstats[inst * 4 + 0] = res_px;
            p.dstats[inst * 4 + 1] = res_dx;
            p.dstats[inst * 4 + 2] = res_pu;
            p.dstats[inst * 4 + 3] = res_du;
        }
    }
}

}  // namespace tinympc
