// tinympc_solve_dw.hip -- k_admm_solve_dw: layout D for WIDE systems (16 < nx+nu <= 32, 32 lanes per instance, two
// instances per wavefront), gfx950 FP64.
//
// Same plan as tinympc_solve_d.hip -- horizon a compile-time constant, both sweeps fully unrolled, the duals g|y in registers
// (2*(N-1)+2 VGPRs), the slack v|z split between registers and LDS, LDS otherwise only for the feed-forward d and one copy
// of the two sweep operators per workgroup, <= 256 VGPRs -> two wavefronts per SIMD -- and the same reference semantics
// (per-instance termination, `iter % check_termination`, solution = vnew / znew, stale v|z after a converged solve:
// admm.cpp:109-207). What differs is the mat-vec: an instance spans two DPP rows, so each step first replicates the operand
// vector across them (v_permlane16_swap on two copies) and then runs the fused DPP chain in two halves of 16 columns
// (tinympc_solve_dw_chain.h). Wide systems ran one wavefront per SIMD on layout A before (LDS: G + V = 33 KB per wave at
// N = 30; profiles/r02_wide_sweep.txt).
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {
template <int NX, int NU>
struct DWStep;  // specialised per (nx, nu) by tinympc_solve_dw_chain.h
}  // namespace tinympc

// The (nx, nu) pairs compiled into the library (see TINY_DW_SHAPES below).
#ifdef TINY_JIT  // run-time specialisation (tinympc_jit.hip): exactly one (nx, nu, N), from -D options
#define DW_NX TINY_JIT_NX
#define DW_NU TINY_JIT_NU
#include "tinympc_solve_dw_chain.h"
#else
#define DW_NX 24
#define DW_NU 8
#include "tinympc_solve_dw_chain.h"
#define DW_NX 20
#define DW_NU 4
#include "tinympc_solve_dw_chain.h"
#endif

namespace tinympc {

namespace {
template <int I, int E, class F>
__device__ __forceinline__ void static_for_w(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for_w<I + 1, E>(f);
    }
}

// ---- LDS plan per workgroup, in doubles: operators [2][32 k][32 r] | per wave: V[VL][64], D[(N-1)*2*nu]
constexpr int DW_OPS_DOUBLES = 2 * 32 * 32;
constexpr int DW_GROUP = 8;        // forward steps between two "can this sweep still converge" tests
#ifdef TINY_JIT_VREG
constexpr int DW_VREG_MAX = TINY_JIT_VREG;  // chosen by the host from its register estimate
#else
constexpr int DW_VREG_MAX = 20;
#endif    // slack knots kept in registers (the rest goes to LDS)
constexpr int DW_LDS_PER_CU = 160 * 1024;
__host__ __device__ constexpr int dw_d_doubles(int nu, int N) { return ((N - 1) * 2 * nu + 1) & ~1; }
// number of slack slots in LDS; -1 if the shape does not fit the plan (8 waves per CU)
__host__ __device__ constexpr int dw_tab_doubles(int N) { return 3 * (N + 2) * 32 + 32; }  // the workgroup's copy of the per-knot tables (!CT)
__host__ __device__ constexpr int dw_vl(int nu, int N, bool ct, int wpg, int cu_waves = 8) {  // cu_waves 4: one wavefront per SIMD, 512 registers
    const int ns = N - 1;
    const int wg_doubles = DW_LDS_PER_CU / 8 * wpg / cu_waves - DW_OPS_DOUBLES - (ct ? 0 : dw_tab_doubles(N));
    const int wave_doubles = wg_doubles / wpg - dw_d_doubles(nu, N);
    if (wave_doubles < 0) return -1;
    const int vlmax = wave_doubles / 64;
    const int want = ns > DW_VREG_MAX ? ns - DW_VREG_MAX : 0;
    return want <= vlmax ? want : -1;
}
__host__ __device__ constexpr size_t dw_lds_bytes(int nu, int N, bool ct, int wpg, int vl) {
    return sizeof(double) * ((size_t)DW_OPS_DOUBLES + (ct ? 0 : dw_tab_doubles(N)) + (size_t)wpg * (vl * 64 + dw_d_doubles(nu, N)));
}

typedef __attribute__((address_space(3))) double lds_double_w;
__device__ __forceinline__ unsigned lds_addr_w(const double *p) { return (unsigned)(size_t)(const lds_double_w *)p; }
template <int OFF>
__device__ __forceinline__ double lds_read_async_w(unsigned addr) {  // the value is valid after the next s_waitcnt lgkmcnt(0)
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
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
__device__ __forceinline__ void lds_wait_w() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// `bad` = ballot of lanes whose row already rules out convergence in this sweep, `live` = ballot of the lanes that
// are still iterating. True if some live instance (32 lanes) has no bad lane.
__device__ __forceinline__ bool wave_may_converge_w(unsigned long long bad, unsigned long long live) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned long long b = (bad >> (j * 32)) & 0xffffffffull, l = (live >> (j * 32)) & 0xffffffffull;
        any = any || (l != 0ull && b == 0ull);
    }
    return any;
}
}  // namespace

template <int NX, int NU, int N, bool CT, int WPG, int VL>
__device__ __forceinline__ void k_admm_solve_dw_body(const SolveParams &p, double *smem) {
    constexpr int W = 32, IPW = 2, NXU = NX + NU, NS = N - 1, DS = IPW * NU, NVR = NS - VL;
    constexpr int KT = 32;  // row stride of p.ops (choose_geometry)
    constexpr int TOFF = (N + 2) * W;
    static_assert(NS >= 3 && VL >= 0 && VL <= NS && NXU > 16 && NXU <= 32, "wide layout D: N >= 4, 16 < nx+nu <= 32");
    using Step = DWStep<NX, NU>;

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane >> 5, r = lane & 31;
    const long grp = (long)blockIdx.x * WPG + wv;
    const bool grp_ok = grp < p.groups;
    const long inst = grp * IPW + j;
    const bool is_x = r < NX;
    const bool is_u = (r >= NX) && (r < NXU);
    const bool inst_ok = grp_ok && inst < p.batch;
    const int koff = is_x ? 1 : 0;  // slot s = knot s+1 on state lanes, knot s on input lanes

    double *sOps = smem;
    double *sT = smem + DW_OPS_DOUBLES;
    double *sV = sT + (CT ? 0 : dw_tab_doubles(N)) + (size_t)wv * (VL * 64 + dw_d_doubles(NU, N));
    double *sD = sV + VL * 64;

    // ---- workgroup-shared: the two sweep operators, transposed to [k][r] (conflict-free row reads), and the tables
    for (int i = threadIdx.x; i < DW_OPS_DOUBLES; i += 64 * WPG) {
        const int which = i >> 10, k = (i >> 5) & 31, rr = i & 31;
        sOps[i] = p.ops[(size_t)which * W * KT + (size_t)rr * KT + k];
    }
    if constexpr (!CT)
        for (int i = threadIdx.x; i < dw_tab_doubles(N); i += 64 * WPG) sT[i] = p.tables[i];

    const size_t g0 = grp_ok ? (size_t)grp : 0;
    double *const gG = p.G + g0 * (N + 1) * 64 + lane;                 // row kn = knot kn
    double *const gD = p.D + g0 * (size_t)(NS * DS);
    double *const gV0 = p.V + (g0 * v_rows(N) + V_PAD) * 64 + lane;    // canonical v|z, knot 0
    double *const gV1u = p.V2 + (g0 * v_rows(N) + V_PAD) * 64;         // stale copy, knot 0 (wave-uniform: scalar base + 32-bit lane offset)
    const unsigned voff = (unsigned)(lane + koff * 64);
    double *const sVl = sV + lane;
    if (grp_ok) {
        for (int i = lane; i < NS * DS; i += 64) sD[i] = gD[i];
        static_for_w<0, VL>([&](auto S) { sVl[S.value * 64] = gV0[(S.value + koff) * 64]; });
    }
    __syncthreads();  // the only workgroup-wide barrier: from here on the waves are independent
    if (!grp_ok) return;

    // ---- register-resident state
    double G[NS], G0, Vr[NVR > 0 ? NVR : 1], V0;
    static_for_w<0, NS>([&](auto S) { G[S.value] = gG[(S.value + koff) * 64]; });
    static_for_w<0, NVR>([&](auto S) { Vr[S.value] = gV0[(VL + S.value + koff) * 64]; });
    G0 = gG[0];
    V0 = gV0[0];

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
    const double *const sMf = sOps + r, *const sMb = sOps + 1024 + r;
    const unsigned aV = lds_addr_w(sVl), aD = lds_addr_w(sDr), aT = lds_addr_w(sTl);
    const int ct = p.check_termination;

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

    auto load_ops = [&](const double *src, double (&m)[32]) {
        static_for_w<0, 32>([&](auto K) { m[K.value] = src[(K.value < NXU ? K.value : 0) * 32]; });
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
        fair_share_priority<2 * NS>(it0, simd_slot);  // (tinympc_sweep.h: the two wavefronts of a SIMD finish together)
        // ---- write-back: G, D and the canonical v|z (not converged: v = vnew, admm.cpp:196-197; converged: the solve
        // returned before v <- vnew, so the canonical copy is the stale one in V2); solution = vnew / znew (:187-188, 204-205)
        const bool wb = pending || (final_round && active);
        if (__ballot(wb) != 0ull) {
            // Rare path (once per instance and solve), kept small in registers rather than fast: addresses are rebuilt
            // here from the kernel arguments (the opaque copy of `lane` keeps the compiler from hoisting them out of
            // the iteration loop, where they would occupy registers the unrolled sweeps need).
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int r_o = lane_o & 31, j_o = lane_o >> 5;
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
        double m[32];
        load_ops(sMf, m);
        // ---------------- knot 0, state lanes: x_0 is given (tiny_set_x0), no mat-vec
        {
            const double lo0 = CT ? lo_c : sT[W + r], hi0 = CT ? hi_c : sT[TOFF + W + r];
            if (may && is_x) gV1u[(unsigned)lane] = V0;
            const double s = x0v + G0;
            const double snew = fmin(hi0, fmax(lo0, s));
            G0 = s - snew;
            pri = is_x ? fabs(x0v - snew) : 0.0;
            dua = is_x ? fabs(V0 - snew) : 0.0;
            V0 = snew;
        }
        // ---------------- forward sweep (F1) with S1 + D1 + R1 fused in
        // LDS operands of a step (its d, and vold of its slot if that lives in LDS) are requested right before the
        // PREVIOUS step's block and retired by that block's trailing s_waitcnt.
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
            double e, o;
            cross_row_pair<1>(is_x ? xcur : dcur, e, o);
            double a = Step::lo_fwd(e, m, cf);
            if constexpr (q >= VL) {
                Step::hi_fwd_reg(a, o, m, locur, hicur, G[q], Vr[q - VL], pri, dua);
            } else {
                double vnew;
                Step::hi_fwd_lds(a, o, m, locur, hicur, G[q], vcur, vnew, pri, dua);
                lds_write_async_w<q * 512>(aV, vnew);
            }
            xcur = a;
            dcur = dn;
            vcur = vn;
            if constexpr (!CT) {
                locur = lon;
                hicur = hin;
            }
        };
        constexpr int NG = (NS + DW_GROUP - 1) / DW_GROUP;
        static_for_w<0, NG>([&](auto Gi) {
            constexpr int s0 = Gi.value * DW_GROUP, s1 = (s0 + DW_GROUP < NS) ? s0 + DW_GROUP : NS;
            if (may) {
                // Stale copy of the group's slots (still holding the previous iterate) before the blocks overwrite them.
                // Rare path: the addresses are rebuilt from an opaque copy of the lane offset so that the compiler does
                // not keep one pointer per slot alive across the iteration loop.
                if constexpr (s0 > 0) {
                    const bool bad = !((pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol));
                    may = __builtin_amdgcn_readfirstlane((int)wave_may_converge_w(__ballot(bad), __ballot(active))) != 0;
                }
                if (may) {
                    unsigned vo = voff;
                    double *base = gV1u;
                    asm volatile("" : "+v"(vo), "+s"(base));
                    static_for_w<s0, s1>([&](auto S) { (base + S.value * 64)[vo] = vget(S); });
                }
            }
            static_for_w<s0, s1>([&](auto S) { fstep(S); });
        });
        if (active) it_done = it1;  // admm.cpp:143

        // ---------------- R1: termination (admm.cpp:93-101), decided element-wise: one ballot, no reductions
        if (check) {
            const bool below = (pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol);
            const bool conv = ((__ballot(below) >> (j * W)) & 0xffffffffull) == 0xffffffffull;
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
            auto lr_of = [&](auto S) -> double {  // linref of slot S (its knot differs by lane type)
                if constexpr (CT) return lr_c;
                else return sTl[2 * TOFF + (S.value + 1) * W];
            };
            double px, rcur, rnext, acc;
            {   // p_{N-1} (state lanes, admm.cpp:81-82) | r_{N-2} (input lanes) share slot NS-1; then slot NS-2
                const double lrT = is_x ? pnref : lr_of(std::integral_constant<int, NS - 1>{});
                const double lr2 = lr_of(std::integral_constant<int, NS - 2>{});
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
                const double lr2 = lr_of(std::integral_constant<int, s2>{});
                const double lrmc2 = is_x ? lr2 + cb : cb;
                double a = acc, an, rn, e, o;
                cross_row_pair<1>(is_x ? px : rcur, e, o);  // [p_{s+1}; r_s]
                Step::lo_bwd(a, e, m);
                Step::hi_bwd(a, o, m, v2cur, G[s2], rhom, lrmc2, nrho, lr2, an, rn);
                lds_write_masked_w<s * DS * 8>(aD, a, wr_d);  // d_s
                px = a;
                rcur = rnext;
                rnext = rn;
                acc = an;
                v2cur = v2n;
            });
            {
                double a = acc, e, o;
                cross_row_pair<1>(is_x ? px : rcur, e, o);
                Step::lo_bwd(a, e, m);
                Step::hi_bwd_last(a, o, m);
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

#ifndef TINY_JIT
template <int NX, int NU, int N, int WPG, int VL>
__global__ void __launch_bounds__(64 * WPG) __attribute__((amdgpu_waves_per_eu(2, 2))) k_admm_solve_dw(const SolveParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    k_admm_solve_dw_body<NX, NU, N, true, WPG, VL>(p, smem);  // (compiled in: time-invariant tables; the other form is specialised at run time)
}
#endif

#ifdef TINY_JIT
}  // namespace tinympc
// The one kernel of a run-time specialised module: a fixed C name, static LDS (its size is known here).
#ifndef TINY_JIT_WPS
#define TINY_JIT_WPS 2  // wavefronts per SIMD: 2 (256 registers each), or 1 (512) for horizons whose duals need them
#endif
#ifndef TINY_JIT_WPG
#define TINY_JIT_WPG (4 * TINY_JIT_WPS)
#endif
extern "C" __global__ void __launch_bounds__(64 * TINY_JIT_WPG) __attribute__((amdgpu_waves_per_eu(TINY_JIT_WPS, TINY_JIT_WPS)))
tinympc_jit_solve(const tinympc::SolveParams p) {
#ifndef TINY_JIT_CT
#define TINY_JIT_CT 1
#endif
    constexpr bool CTJ = TINY_JIT_CT != 0;  // bounds / references constant over the horizon
    constexpr int WPGJ = TINY_JIT_WPG;  // wavefronts per workgroup (4, or 8 where only that LDS plan fits)
    constexpr int VLJ = tinympc::dw_vl(TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, 4 * TINY_JIT_WPS);
    static_assert(VLJ >= 0, "shape does not fit the layout-D plan");
    __shared__ __attribute__((aligned(16))) double smem_jit[tinympc::dw_lds_bytes(TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ) / sizeof(double)];
    tinympc::k_admm_solve_dw_body<TINY_JIT_NX, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ>(p, smem_jit);
}
namespace tinympc {
#else
// ------------------------------------------------------------------------------------------------------------
// Host side: the instantiation table. A shape runs on the wide layout D only if it was compiled in.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int dw_wpg(int nu, int N) { return dw_vl(nu, N, true, 4) >= 0 ? 4 : 8; }  // (see d_wpg in tinympc_solve_d.hip)

template <int NX, int NU, int N>
static hipError_t launch_dw_one(const SolveParams &p, hipStream_t stream) {
    constexpr int WPG = dw_wpg(NU, N);
    constexpr int VL = dw_vl(NU, N, true, WPG);
    if constexpr (VL < 0) {
        return hipErrorInvalidValue;
    } else {
        constexpr size_t lds = dw_lds_bytes(NU, N, true, WPG, VL);
        static size_t lds_set[16] = {0};
        auto fn = &k_admm_solve_dw<NX, NU, N, WPG, VL>;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), lds, lds_set);
        if (e != hipSuccess) return e;
        const int wgs = (p.groups + WPG - 1) / WPG;
        hipLaunchKernelGGL(fn, dim3(wgs), dim3(64 * WPG), lds, stream, p);
        return hipGetLastError();
    }
}

#define TINY_DW_SHAPES(X) \
    X(24, 8, 30)          \
    X(20, 4, 30)

bool solve_dw_supported(int nx, int nu, int N, bool const_tables) {
    if (!const_tables) return false;
#define X(NX_, NU_, N_) \
    if (nx == NX_ && nu == NU_ && N == N_) return dw_vl(NU_, N_, true, dw_wpg(NU_, N_)) >= 0;
    TINY_DW_SHAPES(X)
#undef X
    return false;
}

size_t solve_dw_lds_bytes(int nu, int N) {
    const int wpg = dw_wpg(nu, N);
    return dw_lds_bytes(nu, N, true, wpg, dw_vl(nu, N, true, wpg));
}

int solve_dw_workgroups(int nu, int N, int groups) {
    const int wpg = dw_wpg(nu, N);
    return (groups + wpg - 1) / wpg;
}

hipError_t launch_solve_dw(const SolveParams &p, hipStream_t stream) {
    if (!p.const_tables) return hipErrorInvalidValue;
#define X(NX_, NU_, N_) \
    if (p.nx == NX_ && p.nu == NU_ && p.N == N_) return launch_dw_one<NX_, NU_, N_>(p, stream);
    TINY_DW_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

#endif  // TINY_JIT

}  // namespace tinympc
