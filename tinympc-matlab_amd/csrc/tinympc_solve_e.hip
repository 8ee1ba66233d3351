// tinympc_solve_e.hip -- k_admm_solve_e ("layout E"): layout D's register-resident sweeps with the HORIZON CUT ACROSS THE
// WAVEFRONTS OF A WORKGROUP. The throughput kernel for what no single wavefront can hold on chip: long horizons, and above all
// the cone / linear-inequality families at long horizons (BASELINE config 4: rocket landing N=100 -- five arrays per knot).
//
// Layout D gives a wavefront (4 instances x 16 lanes) the whole horizon: 2 register pairs per knot for the box path, 5 with the
// families -- N=100 with families is 990 VGPRs, and layout D ends at N ~ 30 there. Both sweeps are linear time-invariant
// recurrences (the observation behind the latency kernel, tinympc_solve_c.hip), so the horizon can be cut:
//     workgroup = 4 instances = WPG wavefronts; wavefront w owns the S consecutive slots [w S, (w+1) S) of all four instances
//     (the last wavefront: the S_LAST that are left), every array of its slots in REGISTERS (or its own LDS region) for the
//     whole solve, two wavefronts per SIMD as in layout D.
//   forward    pass 1: the chunk's S steps from a ZERO incoming state (wavefront 0: from x_0), bare mat-vec chains, only the
//              end value e_w is kept -> LDS -> ONE barrier -> every wavefront forms the true state entering its chunk by
//              Horner over the chunks before it, X_w = Phi^S X_(w-1) + e_(w-1) (at most WPG-1 short mat-vecs; Phi^S from
//              k_build_chunk_tables) -> pass 2: the real sweep with the row-local phases fused in, exactly layout D's step.
//   backward   the same from the top. The cut is placed so that no wavefront needs a neighbour's linear-cost entry: the chain of
//              chunk w starts from q~ + c_in, q~ = the q (p_{N-1} for the last chunk) of its OWN last state slot, and leaves out
//              the q of its first knot, which the chunk below adds from its own registers. The termination ballots cross with the
//              backward carries: TWO barriers per ADMM iteration.
// Given exact carries pass 2 IS the sequential sweep; results differ from layouts A/B/D through the rounding of the carries
// (~1e-14 relative), iteration counts match the restatement in every test.
//
// The families (PARITY UNPINNED upstream semantics, see tinympc_solve_fam.hip) are specialised on their STRUCTURE, which is a
// compile-time input here (the kernel exists only as a run-time specialisation, tinympc_jit.hip): the cone list (in list order,
// grouped into rounds of pairwise-disjoint cones; overlapping cones land in successive rounds = upstream's one-after-the-other
// projection) and the number of linear rows per side. Cross-lane sums then need no mask rows (72 VGPRs in the other kernels):
// a cone's ||w||^2 and t are gathered by dim DPP instructions under an EXEC mask of the cone's lanes, a linear row's dot
// product by one DPP instruction per row of its side -- 6 + 9 instead of 36 instructions per knot for the rocket.
//
// Reference semantics that need care are layout D's (tinympc_solve_d.hip): per-instance termination, zombies, the stale copy
// for converged solves (admm.cpp:181-197).
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

#if !defined(TINY_JIT) && !defined(TINY_BUILTIN)  // stand-alone instance (ISA lint, "does it compile"): the rocket landing of BASELINE config 4
#define TINY_CHAIN_NOP 1  // the chain blocks as the run-time specialisations get them (tinympc_solve_d_chain.h)
#define TINY_JIT_NX 6
#define TINY_JIT_NU 3
#define TINY_JIT_N 100
#define TINY_JIT_CT 0
#define TINY_JIT_FAM 1
#define TINY_JIT_E_WPG 8
#define TINY_JIT_E_S 13
#define TINY_JIT_E_NROUND 1
#define TINY_JIT_E_NCONE 2
#define TINY_JIT_E_CONES {0, 0, 2}, {0, 6, 8}
#define TINY_JIT_E_NLX 1
#define TINY_JIT_E_NLU 0
#define TINY_JIT_E_GC_LDS 0
#define TINY_JIT_E_GL_LDS 0
#define TINY_JIT_E_LX_LDS 0
#define TINY_JIT_E_KFAM 1
#define TINY_JIT_E_DREG 1
#endif
#ifndef TINY_JIT_E_DREG
#define TINY_JIT_E_DREG 0
#endif
#ifndef TINY_JIT_E_KFAM
#define TINY_JIT_E_KFAM 0
#endif

namespace tinympc {
template <int NX, int NU>
struct DStep;  // tinympc_solve_d_chain.h
}  // namespace tinympc
#define D_NX TINY_JIT_NX
#define D_NU TINY_JIT_NU
#include "tinympc_solve_d_chain.h"
#include "tinympc_solve_e_common.h"

// Timing experiments (tools/e_breakdown.py, through TINYMPC_JIT_DEFS="-DTINY_E_EXP=k"; results are WRONG, only the clock counts):
//   1 the families' row-local evaluation returns at once (its LDS traffic stays)   2 no families' work in the forward step at all
//   3 no pass 1 (forward and backward)   4 no workgroup barriers   5 = 2 + 3 + 4
#ifndef TINY_E_EXP
#define TINY_E_EXP 0
#endif

// 9: the shader clock at the phase boundaries of iteration 10, per wavefront, left in sol_x of the workgroup's first instance
// (tools/e_breakdown.py --stamps prints them)
#if TINY_E_EXP == 9
#define E_STAMP(k) do { if (it0 == 10) stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define E_STAMP(k) do { } while (0)
#endif

// Issue priority between the two wavefronts of a SIMD (experiments, TINY_E_PRIO): 0 none, 1 the wavefront in the odd slot leads for
// the whole kernel, 2 wall-clock slices of 2^TINY_E_PRIO_SHIFT x 10 ns sampled at every sweep step, 3 feedback from the barriers:
// whoever waited less than TINY_E_PRIO_TH cycles at the last barrier (i.e. came late) leads until the next one, 4 a RELAY between
// the two barriers: the wavefront in the even slot leads (priority 2 against 1) through the first part of the interval, then drops
// to 0 while its partner rises to 3 (TINY_E_RELAY_A / _B: where, 1 = early, 2 = late)
#ifndef TINY_E_PRIO
#define TINY_E_PRIO 0
#endif
// Woven sweep steps (tinympc_solve_d_chain.h: fwd_reg_woven / bwd_woven): the row-local block of slot q-1 rides between the chain
// instructions of slot q, the backward tail between those of its own step. A wavefront issues in order and every FP64 result takes
// ~8 cycles to come back: two dependent sequences interleaved fill each other's gaps (-DTINY_E_WOVEN=0: the blocks back to back).
#ifndef TINY_E_WOVEN
#define TINY_E_WOVEN 1
#endif
#ifndef TINY_E_PRIO_SHIFT
#define TINY_E_PRIO_SHIFT 4
#endif
#ifndef TINY_E_PRIO_TH
#define TINY_E_PRIO_TH 300
#endif
#ifndef TINY_E_RELAY_A
#define TINY_E_RELAY_A 1
#endif
#ifndef TINY_E_RELAY_B
#define TINY_E_RELAY_B 2
#endif

namespace tinympc {

#ifdef TINY_E_FIRST
constexpr int E_FIRST = TINY_E_FIRST;  // (experiments)
#else
constexpr int E_FIRST = 1;  // forward steps before the first "can this sweep still converge" test (a wavefront's chunk is
                            // a handful of slots: with one group all of them went to the stale copy in every checked sweep)
#endif
constexpr int E_GROUP = 8;  // ... and between two later ones

// KFAM: the families evaluated one KNOT per lane, once per iteration (KFamilies, tinympc_solve_e_common.h) instead of one (row,
// knot) element per lane in every slot of the sweep; the placement flags GC_LDS / GL_LDS / LX_LDS belong to the element form.
// DREG: the feed-forward d of the wavefront's slots in registers too (2 S VGPRs) instead of its LDS region: the backward chain
// leaves d_s on the input lanes, exactly where the forward chain's DPP columns read it.
// WPG == 1 (round 4): the horizon is NOT cut -- one wavefront sweeps all of it, no pass 1, no carries, no workgroup barrier inside the
// iteration: layout D's data flow with this kernel's knot-per-lane families, for the families at SHORT horizons (the rocket landing of
// the reference's own example, N = 10: layout D's families variant pays the element-per-lane evaluation in every slot). A workgroup
// is then GPW independent groups of four instances, one wavefront each, sharing the LDS copies of the operators and tables.
template <int NX, int NU, int N, bool CT, int WPG, int S, bool FAM, bool GC_LDS, bool GL_LDS, bool LX_LDS, bool KFAM, bool DREG, int GPW = 1>
__device__ __forceinline__ void k_admm_solve_e_body(const SolveParams &p, double *smem) {
    constexpr int W = 16, IPW = 4, NXU = NX + NU, NS = N - 1, DS = IPW * NU;
    constexpr int KT = NXU <= 8 ? 8 : NXU <= 12 ? 12 : 16;  // row stride of p.ops (choose_geometry)
    constexpr int KS = NX <= 8 ? 8 : NX <= 12 ? 12 : 16;    // row stride of the carry matrices (chunk_ks)
    constexpr int TOFF = (N + 2) * W;
    constexpr int S_LAST = NS - (WPG - 1) * S;              // slots of the last wavefront
    static_assert(WPG >= 1 && S >= 3 && S_LAST >= 3 && S_LAST <= S, "layout E: every chunk holds at least three slots");
    static_assert(GPW == 1 || WPG == 1, "several groups per workgroup: only where a group is one wavefront");
    constexpr bool CUT = WPG > 1;                            // the horizon is cut across wavefronts: pass 1, carries, barriers
    constexpr int NTHREADS = 64 * WPG * GPW;
    constexpr int PG = 2 * WPG * 64 + 16 + WPG * 16 + 6 * 64;  // LDS doubles per group: carries | flags | residual partials | knot 0
    constexpr bool KF = FAM && KFAM;
    constexpr bool GCL = FAM && !KF && GC_LDS, GLL = FAM && !KF && GL_LDS, LXL = FAM && !KF && LX_LDS;
    constexpr int NLDS = KF ? -1 : (GCL ? 1 : 0) + (GLL ? 1 : 0) + (LXL ? 1 : 0);
    constexpr int ES = kfam_es(NXU), NPASS = kfam_passes(S);  // (KF) doubles per entry of the exchange buffer; passes of 16 entries per instance
    constexpr int RS = e_fam_row(NXU);  // doubles per slot of a families' array in LDS: the four instances' real rows, packed
    using Step = DStep<NX, NU>;

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wavefront in the workgroup
    const int wv = CUT ? wib : 0;   // wavefront of its group (its chunk of the horizon)
    const int gi = CUT ? 0 : wib;   // group of the workgroup
    const int j = lane >> 4, r = lane & 15;
    const long grp_raw = (long)blockIdx.x * GPW + gi;  // one group = four instances
    const long grp = grp_raw < p.groups ? grp_raw : (long)p.groups - 1;  // (a workgroup's last groups may not exist: they load a real group's
                                                                          // state and store nothing -- every store is gated by inst_ok)
    const long inst = grp_raw * IPW + j;
    const bool is_x = r < NX;
    const bool is_u = (r >= NX) && (r < NXU);
    const bool inst_ok = inst < p.batch;
    const int koff = is_x ? 1 : 0;  // slot s = knot s+1 on state lanes, knot s on input lanes
    const bool top = wv == WPG - 1;
    const bool bottom = wv == 0;
    const int s0 = wv * S;          // first slot of this wavefront

    // ---- LDS
    double *sOps = smem;                                          // [2][16 k][16 r]
    double *sT = sOps + 512;                                      // tables (!CT)
    double *sLin = sT + (CT ? 0 : 3 * (N + 2) * 16 + 16);         // [E_NL][3][16]  a_k | b_k | 1/||a_k||^2 (FAM)
    double *sMu = sLin + (FAM ? 3 * E_NL * 16 : 0);               // [2][E_NCONE] the cones' slopes | their reciprocals (FAM)
    double *sPow = sMu + (FAM ? ((2 * E_NCONE + 1) & ~1) : 0);    // Phi^S | Psi^S, [16 k][16 r] each
    double *sE = sPow + 512 + (size_t)gi * PG;                    // per group: [WPG][64] forward carries
    double *sB = sE + WPG * 64;                                   // [WPG][64] backward carries
    int *sFlag = reinterpret_cast<int *>(sB + WPG * 64);          // [WPG] per-instance "below tolerance" bits (16 doubles)
    double *sRes = sB + WPG * 64 + 16;                            // [WPG][4 instances][4]
    double *sK0 = sRes + WPG * 16;                                // [6][64] knot 0 of the state rows (g, v, gc, gl) and x0: the bottom wavefront's
    double *sWave = sPow + 512 + (size_t)GPW * PG + (size_t)(gi * WPG + wv) * e_wave_doubles(NXU, NU, S, NLDS);
    double *sGC = sWave;                                          // [S][RS] each, where the plan puts them into LDS
    double *sGL = sGC + (GCL ? S * RS : 0);
    double *sLX = sGL + (GLL ? S * RS : 0);
    double *sD = KF ? sWave + kfam_doubles(NXU, S) : sLX + (LXL ? S * RS : 0);
    double *sKX = sWave;                                          // (KF) the exchange buffer: x | u up, the families' linear-cost term down
    // this lane's entry of a packed families' row; lanes beyond the system's rows read their instance's last real entry and
    // never write (EXEC mask of the real lanes)
    const int famIdx = j * NXU + (r < NXU ? r : NXU - 1);
    const unsigned long long mask_real = __ballot(r < NXU);
    // (KF) this lane's two roles in the exchange buffer: as row r of instance j (slot q = entry q+1: kxRow + (q+1) ES) and as
    // entry t = r (+ 16 per pass) of instance j (kxT + 16 pass ES .. + nx+nu)
    const int kxRow = j * (S + 1) * ES + (r < NXU ? r : NXU - 1);
    const int kxT = (j * (S + 1) + r) * ES;
    const int s_real = top ? S_LAST : S;  // slots this wavefront really owns

    for (int i = threadIdx.x; i < 512; i += NTHREADS) {
        const int which = i >> 8, k = (i >> 4) & 15, rr = i & 15;
        sOps[i] = (k < KT) ? p.ops[(size_t)which * W * KT + (size_t)rr * KT + k] : 0.0;
        if constexpr (CUT) sPow[i] = (k < NX && rr < NX) ? p.ctab[(size_t)which * W * KS + (size_t)rr * KS + k] : 0.0;
    }
    if constexpr (!CT)
        for (int i = threadIdx.x; i < 3 * (N + 2) * 16 + 16; i += NTHREADS) sT[i] = p.tables[i];
    if constexpr (FAM) EFamilies<NX, NU>::stage_linear_rows(p.fam, KT, sLin, (int)threadIdx.x, NTHREADS);
    if constexpr (FAM) KFamilies<NX, NU>::stage_cone_slopes(p.fam, KT, sMu, (int)threadIdx.x);

    // canonical HBM layout, shared with every other kernel
    const size_t vbase = ((size_t)grp * v_rows(N) + V_PAD) * 64;
    double *const gG = p.G + (size_t)grp * (N + 1) * 64 + lane;  // row kn = knot kn
    double *const gD = p.D + (size_t)grp * (size_t)(NS * DS);
    double *const gV0 = p.V + vbase + lane;                       // canonical v|z, knot 0
    double *const gV1u = p.V2 + vbase;                            // stale copy, knot 0 (wave-uniform)
    const unsigned voff = (unsigned)(lane + koff * 64);
    // slot i of this wavefront is real (compile-time i; the last wavefront owns S_LAST slots)
    auto real = [&](int i) -> bool { return (i < S_LAST) || !top; };
    // rows of the persistent arrays: knot s0 + i + koff; slots that do not exist read the per-lane dummy rows
    auto g_row = [&](int i) -> int { return real(i) ? (s0 + i + koff) : N; };
    auto v_row = [&](int i) -> int { return real(i) ? (s0 + i + koff) : N; };

    if constexpr (!DREG)
        for (int i = lane; i < S * DS; i += 64) {
            const int row = i / DS;
            sD[i] = (s0 + row < NS) ? gD[(size_t)(s0 + row) * DS + i % DS] : 0.0;
        }
    if constexpr (FAM) {
        const double *const gGC = p.GC + vbase + lane, *const gGL = p.GL + vbase + lane;
        if (r < NXU) {
            e_static_for<0, S>([&](auto I) {
                if constexpr (GCL) sGC[I.value * RS + famIdx] = gGC[(size_t)v_row(I.value) * 64];
                if constexpr (GLL) sGL[I.value * RS + famIdx] = gGL[(size_t)v_row(I.value) * 64];
                if constexpr (LXL) sLX[I.value * RS + famIdx] = 0.0;
            });
        }
    }
    if constexpr (KF) {
        for (int i = lane; i < kfam_doubles(NXU, S); i += 64) sKX[i] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (bottom && r < NXU) sKX[kxRow] = (inst_ok && is_x) ? p.x0[inst * NX + r] : 0.0;  // entry 0: x_0 (constant over the solve)
    }
    if (bottom) {  // knot 0 of the state rows and x0
        sK0[lane] = gG[0];
        sK0[64 + lane] = gV0[0];
        sK0[2 * 64 + lane] = FAM ? (p.GC + vbase)[lane] : 0.0;
        sK0[3 * 64 + lane] = FAM ? (p.GL + vbase)[lane] : 0.0;
        sK0[4 * 64 + lane] = (inst_ok && is_x) ? p.x0[inst * NX + r] : 0.0;
    }
    __syncthreads();  // (the only barrier that also waits for global loads)

    // ---- register-resident state of this wavefront's slots
    double G[S], V[S], DR[DREG ? S : 1];
    e_static_for<0, S>([&](auto I) {
        G[I.value] = gG[(size_t)g_row(I.value) * 64];
        V[I.value] = gV0[(size_t)v_row(I.value) * 64];
        if constexpr (DREG) DR[I.value] = (is_u && s0 + I.value < NS) ? gD[(size_t)(s0 + I.value) * DS + j * NU + (r - NX)] : 0.0;
    });
    double GC[(FAM && !KF && !GC_LDS) ? S : 1], GLr[(FAM && !KF && !GL_LDS) ? S : 1], LX[(FAM && !KF && !LX_LDS) ? S : 1];
    if constexpr (FAM && !KF) {
        const double *const gGC = p.GC + vbase + lane, *const gGL = p.GL + vbase + lane;
        e_static_for<0, S>([&](auto I) {
            if constexpr (!GC_LDS) GC[I.value] = gGC[(size_t)v_row(I.value) * 64];
            if constexpr (!GL_LDS) GLr[I.value] = gGL[(size_t)v_row(I.value) * 64];
            if constexpr (!LX_LDS) LX[I.value] = 0.0;
        });
    }
    // (KF) the families' duals of this lane's knots, one KFamilies per pass of 16 entries; canonical HBM layout as everywhere
    KFamilies<NX, NU> kf[KF ? NPASS : 1];
    unsigned long long kmask[KF ? NPASS : 1];  // lanes whose entry is a slot of this wavefront (they hand a linear-cost term back)
    if constexpr (KF) {
        const double *const bGC = p.GC + vbase + j * 16, *const bGL = p.GL + vbase + j * 16;
        e_static_for<0, NPASS>([&](auto Pp) {
            constexpr int pp = decltype(Pp)::value;
            const int t = r + 16 * pp;
            kmask[pp] = __ballot(t >= 1 && t <= s_real);
            kf[pp].init(p.fam, KT, sLin, sMu, p.rho);
            const bool ent = (t >= 1 && t <= s_real) || (t == 0 && bottom);
            e_static_for<0, NXU>([&](auto R) {
                constexpr int rr = decltype(R)::value;
                const int kn = rr < NX ? s0 + t : s0 + t - 1;  // state rows: knot s0+t; input rows: knot s0+t-1 (none for entry 0)
                const bool ok = ent && (rr < NX || t >= 1);
                kf[pp].gc[rr] = ok ? bGC[(size_t)kn * 64 + rr] : 0.0;
                kf[pp].gl[rr] = ok ? bGL[(size_t)kn * 64 + rr] : 0.0;
            });
        });
    }
    // the families, specialised on their structure (tinympc_solve_e_common.h)
    EFamilies<NX, NU> fam_eval;
    if constexpr (FAM && !KF) fam_eval.init(p.fam, KT, sLin, r, p.rho);
    auto families = [&](double val, double gc_old, double gl_old, double &gc_new, double &gl_new) -> double {
#if TINY_E_EXP == 1
        gc_new = gc_old;
        gl_new = gl_old;
        return val;
#else
        return fam_eval.eval(val, gc_old, gl_old, gc_new, gl_new);
#endif
    };

    const double cf = p.ops[(size_t)2 * W * KT + r];
    const double cb = p.ops[(size_t)2 * W * KT + W + r];
    const double pnref = p.tables[(size_t)3 * TOFF + r];
    const double nrho = -p.rho;
    const double rhom = is_x ? nrho : 0.0;
    const double lo_c = p.tables[W + r], hi_c = p.tables[(size_t)TOFF + W + r], lr_c = p.tables[(size_t)2 * TOFF + W + r];
    const int dIdx = j * NU + (is_u ? r - NX : 0);
    const double *const sTl = sT + (size_t)(s0 + koff) * W + r;  // (!CT) row of local slot i: sTl[(i + 1) * W]
    const double *const sMf = sOps + r, *const sMb = sOps + 256 + r;
    const unsigned aD = e_lds_addr(sD + dIdx), aT = e_lds_addr(sTl);
    const unsigned aGC = e_lds_addr(sGC + famIdx), aGL = e_lds_addr(sGL + famIdx), aLX = e_lds_addr(sLX + famIdx);
    const unsigned aKX = e_lds_addr(sKX + kxRow), aKT = e_lds_addr(sKX + kxT);
    const int ct = p.check_termination;

    bool active = inst_ok;
    bool pending = false;  // converged in the previous round: state not yet written back
    int it_done = 0;
    int status = 11;       // TINY_UNSOLVED (admm.cpp:114)
    bool res_valid = false;
    double snap_pri = 0.0, snap_dua = 0.0;

    auto load_ops = [&](const double *src, double (&m)[16]) {
        e_static_for<0, 16>([&](auto K) { m[K.value] = src[(K.value < NXU ? K.value : 0) * 16]; });
    };
    auto load_pow = [&](const double *src, double (&m)[16]) {  // a carry matrix row: state columns only
        e_static_for<0, 16>([&](auto K) { m[K.value] = (K.value < NX) ? src[K.value * 16] : 0.0; });
    };
    auto lr_of = [&](auto I) -> double {  // linref of local slot I (+ the families' term)
        double base;
        if constexpr (CT) base = lr_c;
        else base = sTl[2 * TOFF + (I.value + 1) * W];
        if constexpr (KF) base += sKX[(I.value + 1) * ES + kxRow];
        else if constexpr (FAM) {
            if constexpr (LX_LDS) base += sLX[I.value * RS + famIdx];
            else base += LX[I.value];
        }
        return base;
    };

#if TINY_E_EXP == 9
    unsigned long long stamp[10] = {};
#endif
    const unsigned simd_slot = (unsigned)simd_slot_id();
    if (TINY_E_PRIO == 1 && (simd_slot & 1u)) __builtin_amdgcn_s_setprio(3);
    auto prio_tick = [&]() {
        if constexpr (TINY_E_PRIO == 2) {
            const unsigned sl = (unsigned)(__builtin_amdgcn_s_memrealtime() >> TINY_E_PRIO_SHIFT);
            if ((sl ^ simd_slot) & 1u) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        }
    };
    // TINY_E_PRIO == 4: start of an interval between two barriers / the hand-over inside it
    const bool relay_first = (simd_slot & 1u) == 0u;
    auto relay_start = [&]() {
        if constexpr (TINY_E_PRIO == 4) {
            if (relay_first) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        }
    };
    auto relay_handover = [&]() {
        if constexpr (TINY_E_PRIO == 4) {
            if (relay_first) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(3);
        }
    };
    auto barrier_fb = [&]() {  // the workgroup barrier, with the priority feedback of TINY_E_PRIO == 3
        if constexpr (TINY_E_PRIO == 3) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            e_barrier();
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (t1 - t0 < (unsigned long long)TINY_E_PRIO_TH) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        } else {
            e_barrier();
        }
    };
    const int max_iter = p.max_iter;
    for (int it = 0; max_iter > 0; ++it) {  // admm.cpp:129
        const int it0 = __builtin_amdgcn_readfirstlane(it);
        const bool final_round = it0 >= max_iter;
        // ---- write-back of this wavefront's slots: G, D, the canonical v|z, the solution (see tinympc_solve_d.hip)
        const bool wb = pending || (final_round && active);
        if (__ballot(wb) != 0ull) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int r_o = lane_o & 15, j_o = lane_o >> 4;
            const bool x_o = r_o < NX;
            if (wb && r_o < NXU) {
                const int ko = x_o ? 1 : 0;
                const size_t inst_o = (size_t)grp * IPW + j_o;
                double *const wG = p.G + (size_t)grp * (N + 1) * 64 + lane_o;
                double *const wV = p.V + vbase + lane_o;
                double *const wS = x_o ? p.sol_x + inst_o * N * NX + r_o : p.sol_u + inst_o * NS * NU + (r_o - NX);
                const int sst = x_o ? NX : NU;
                if (x_o && bottom) {  // knot 0
                    wG[0] = sK0[lane_o];
                    wV[0] = sK0[64 + lane_o];
                    wS[0] = sK0[64 + lane_o];
                }
                e_static_for<0, S>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if (real(i)) {
                        const size_t kn = (size_t)(s0 + i + ko);
                        wG[kn * 64] = G[i];
                        wV[kn * 64] = V[i];
                        wS[kn * sst] = V[i];
                    }
                });
                if constexpr (FAM && !KF) {
                    double *const wGC = p.GC + vbase + lane_o, *const wGL = p.GL + vbase + lane_o;
                    e_static_for<0, S>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        if (real(i)) {
                            const size_t kn = (size_t)(s0 + i + ko);
                            if constexpr (GC_LDS) wGC[kn * 64] = sGC[i * RS + famIdx];
                            else wGC[kn * 64] = GC[i];
                            if constexpr (GL_LDS) wGL[kn * 64] = sGL[i * RS + famIdx];
                            else wGL[kn * 64] = GLr[i];
                        }
                    });
                    if (x_o && bottom) {
                        wGC[0] = sK0[2 * 64 + lane_o];
                        wGL[0] = sK0[3 * 64 + lane_o];
                    }
                }
                if (!x_o) {
                    double *const wD = p.D + (size_t)grp * (NS * DS) + j_o * NU + (r_o - NX);
                    if constexpr (DREG) {
                        e_static_for<0, S>([&](auto I) {
                            if (s0 + I.value < NS) wD[(size_t)(s0 + I.value) * DS] = DR[I.value];
                        });
                    } else {
                        for (int i = 0; i < S; ++i)
                            if (s0 + i < NS) wD[(size_t)(s0 + i) * DS] = sD[i * DS + dIdx];
                    }
                }
            }
            if constexpr (KF) {  // the families' duals leave from the knot-per-lane layout (wb is uniform over an instance's 16 lanes)
                if (wb) {
                    double *const wGC = p.GC + vbase + j_o * 16, *const wGL = p.GL + vbase + j_o * 16;
                    e_static_for<0, NPASS>([&](auto Pp) {
                        constexpr int pp = decltype(Pp)::value;
                        const int t = r_o + 16 * pp;
                        if ((t >= 1 && t <= s_real) || (t == 0 && bottom)) {
                            e_static_for<0, NXU>([&](auto R) {
                                constexpr int rr = decltype(R)::value;
                                if (rr < NX || t >= 1) {
                                    const size_t kn = (size_t)(rr < NX ? s0 + t : s0 + t - 1);
                                    wGC[kn * 64 + rr] = kf[pp].gc[rr];
                                    wGL[kn * 64 + rr] = kf[pp].gl[rr];
                                }
                            });
                        }
                    });
                }
            }
            pending = false;
        }
        if (final_round || __ballot(active) == 0ull) break;  // (uniform over the workgroup: termination is decided jointly)
        const int it1 = it0 + 1;
        E_STAMP(0);
        const bool check = __builtin_amdgcn_readfirstlane((int)((ct > 0) && ((it1 % ct) == 0))) != 0;  // admm.cpp:91

        // One ADMM iteration of this wavefront's chunk: S slots, of which the last wavefront owns only the first S_LAST -- ONE copy of
        // the code, the steps beyond S_LAST behind a scalar branch (two instantiations, one per chunk length, cost ~60 VGPRs).
        {
            double m[16];
            load_ops(sMf, m);
            // ================= forward, pass 1: the chunk's end state from a zero incoming state =================
            if constexpr (CUT) {
                double xt = bottom ? sK0[4 * 64 + lane] : 0.0;
                // (values of e_lds_read_async are only ever handed to the chain blocks, never copied: a two-entry ring indexed by the
                // slot's parity instead of `cur = next` -- the compiler may place such a copy in front of the block's s_waitcnt)
                double dring[2] = {0.0, 0.0};
                if constexpr (!DREG) {
                    dring[0] = e_lds_read_async<0>(aD);
                    e_lds_wait();
                }
#if TINY_E_EXP != 3 && TINY_E_EXP != 5
                e_static_for<0, S>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (!DREG && i + 1 < S) dring[(i + 1) & 1] = e_lds_read_async<(i + 1) * DS * 8>(aD);
                    prio_tick();
                    if constexpr (i < S_LAST) xt = Step::fwd_plain(xt, DREG ? DR[i] : dring[i & 1], m, cf);
                    else if (!top) xt = Step::fwd_plain(xt, DREG ? DR[i] : dring[i & 1], m, cf);
                });
#endif
                sE[wv * 64 + lane] = xt;
            }
            E_STAMP(1);
            if constexpr (CUT) barrier_fb();
            relay_start();
            E_STAMP(2);
            // the true state entering the chunk: X_w = Phi^S X_(w-1) + e_(w-1), X_1 = e_0
            double xin = 0.0;
            if (CUT && !bottom) {
                // (requesting all carries at once, ahead of the recurrence, measured no faster and costs 14 VGPRs -- with them the
                // rocket's all-in-registers plan spilled)
                double ph[16];
                load_pow(sPow + r, ph);
                xin = sE[lane];
#pragma unroll 1
                for (int q = 1; q < wv; ++q) xin = Step::fwd_plain(xin, 0.0, ph, sE[q * 64 + lane]);
            }
            // ================= forward, pass 2: the real sweep (F1) with S1 + D1 + R1 fused in =================
            double pri = 0.0, dua = 0.0;
            bool may = check;  // wave-uniform: can this sweep still end converged for some instance of the wave?
            if (bottom) {      // knot 0, state lanes: x_0 is given (tiny_set_x0), no mat-vec
                const double lo0 = CT ? lo_c : sT[W + r], hi0 = CT ? hi_c : sT[TOFF + W + r];
                const double x0v = sK0[4 * 64 + lane], G0 = sK0[lane], V0 = sK0[64 + lane];
                if (may && is_x && active) gV1u[(unsigned)lane] = V0;  // (active: as for the slots below)
                const double s = x0v + G0;
                const double snew = fmin(hi0, fmax(lo0, s));
                sK0[lane] = s - snew;
                pri = is_x ? fabs(x0v - snew) : 0.0;
                dua = is_x ? fabs(V0 - snew) : 0.0;
                sK0[64 + lane] = snew;
                if constexpr (FAM && !KF) {  // (its lx only reaches p_0, which nothing reads; the duals persist)
                    double gcn, gln;
                    (void)families(x0v, sK0[2 * 64 + lane], sK0[3 * 64 + lane], gcn, gln);
                    if (is_x) {
                        sK0[2 * 64 + lane] = gcn;
                        sK0[3 * 64 + lane] = gln;
                    }
                }
                xin = x0v;
            }
            E_STAMP(3);
            double xcur = xin;
            {
                // (rings instead of cur / next / prev variables: what e_lds_read_async returns goes into the chain blocks as it is -- no
                // register copy that the compiler could place in front of a block's s_waitcnt. Bounds: slot q's live in entry q % 3 --
                // the woven step also needs slot q-1's while slot q+1's are on their way --, d and the element form's duals in q % 2)
                double dring[2] = {0.0, 0.0}, gcring[2] = {0.0, 0.0}, glring[2] = {0.0, 0.0};
                double loring[3] = {lo_c, lo_c, lo_c}, hiring[3] = {hi_c, hi_c, hi_c};
                double xprev = 0.0;  // (woven steps: the previous slot's element)
                if constexpr (!DREG) dring[0] = e_lds_read_async<0>(aD);
                if constexpr (!CT) {
                    loring[0] = e_lds_read_async<W * 8>(aT);
                    hiring[0] = e_lds_read_async<(TOFF + W) * 8>(aT);
                }
                if constexpr (GCL) gcring[0] = e_lds_read_async<0>(aGC);
                if constexpr (GLL) glring[0] = e_lds_read_async<0>(aGL);
                e_lds_wait();
                auto fstep = [&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    constexpr int c3 = q % 3, n3 = (q + 1) % 3, p3 = (q + 2) % 3, c2 = q & 1, n2 = (q + 1) & 1;
                    if constexpr (!DREG && q + 1 < S) dring[n2] = e_lds_read_async<(q + 1) * DS * 8>(aD);
                    if constexpr (!CT && q + 1 < S) {
                        loring[n3] = e_lds_read_async<(q + 2) * W * 8>(aT);
                        hiring[n3] = e_lds_read_async<(TOFF + (q + 2) * W) * 8>(aT);
                    }
                    if constexpr (GCL && q + 1 < S) gcring[n2] = e_lds_read_async<(q + 1) * RS * 8>(aGC);
                    if constexpr (GLL && q + 1 < S) glring[n2] = e_lds_read_async<(q + 1) * RS * 8>(aGL);
                    prio_tick();
                    if constexpr (!TINY_E_WOVEN || q == 0) {
                        xcur = Step::fwd_reg(xcur, DREG ? DR[q] : dring[c2], m, cf, loring[c3], hiring[c3], G[q], V[q], pri, dua);  // (slot 0 complete: the first "can it still converge" test reads it)
                    } else if constexpr (q == 1) {
                        xprev = xcur = Step::fwd_plain(xcur, DREG ? DR[q] : dring[c2], m, cf);  // its row-local block rides on step 2
                    } else {
                        const double xp = xprev;  // slot q-1's element = this step's state operand on state lanes, u_(q-1) on input lanes
                        xprev = xcur = Step::fwd_reg_woven(xcur, DREG ? DR[q] : dring[c2], m, cf, xp, loring[p3], hiring[p3], G[q - 1], V[q - 1], pri, dua);
                    }
                    // (KF) this slot's element -- x_{q+1} on state lanes, u_q on input lanes -- goes up to its knot's lane
                    if constexpr (KF && TINY_E_EXP != 2 && TINY_E_EXP != 5) e_lds_write_masked<(q + 1) * ES * 8>(aKX, xcur, mask_real);
                    if constexpr (FAM && !KF && TINY_E_EXP != 2 && TINY_E_EXP != 5) {  // xcur: x_{q+1} on state lanes, u_q on input lanes -- this slot's element
                        double gcn, gln;
                        double gl_old, gc_old;
                        if constexpr (GC_LDS) gc_old = gcring[c2];
                        else gc_old = GC[q];
                        if constexpr (GL_LDS) gl_old = glring[c2];
                        else gl_old = GLr[q];
                        const double l = families(xcur, gc_old, gl_old, gcn, gln);
                        if constexpr (GC_LDS) e_lds_write_masked<q * RS * 8>(aGC, gcn, mask_real);
                        else GC[q] = gcn;
                        if constexpr (GL_LDS) e_lds_write_masked<q * RS * 8>(aGL, gln, mask_real);
                        else GLr[q] = gln;
                        if constexpr (LX_LDS) e_lds_write_masked<q * RS * 8>(aLX, l, mask_real);
                        else LX[q] = l;
                    }
                };
                constexpr int EF = E_FIRST < S ? E_FIRST : S;
                constexpr int NG = 1 + (S - EF + E_GROUP - 1) / E_GROUP;
                e_static_for<0, NG>([&](auto Gi) {
                    constexpr int g0 = Gi.value == 0 ? 0 : EF + (Gi.value - 1) * E_GROUP;
                    constexpr int g1 = Gi.value == 0 ? EF : ((g0 + E_GROUP < S) ? g0 + E_GROUP : S);
                    if (may) {
                        // stale copy of the group's slots (still the previous iterate) before the blocks overwrite them -- only
                        // while some instance of this wavefront can still converge in this sweep (exact: the maxima only grow)
                        if constexpr (g0 > 0) {
                            const bool bad = !((pri < p.abs_pri_tol) && (dua * p.rho < p.abs_dua_tol));
                            may = __builtin_amdgcn_readfirstlane((int)e_wave_may_converge(__ballot(bad), __ballot(active))) != 0;
                        }
                        if (may) {
                            unsigned vo = voff + (unsigned)(s0 * 64);
                            double *base = gV1u;
                            asm volatile("" : "+v"(vo), "+s"(base));
                            e_static_for<g0, g1>([&](auto Q) {
                                // (active: a converged instance's stale copy must survive its neighbours' later sweeps, tinympc_solve_d.hip)
                                if (real(Q.value) && active) (base + Q.value * 64)[vo] = V[Q.value];
                            });
                        }
                    }
                    e_static_for<g0, g1>([&](auto Q) {
                        if constexpr (Q.value < S_LAST) fstep(Q);
                        else if (!top) fstep(Q);
                    });
                });
                if constexpr (TINY_E_WOVEN != 0) {  // the row-local block of the sweep's last slot (every chunk has >= 3 slots)
                    if constexpr (S_LAST == S) {
                        Step::project_prev(xprev, loring[(S - 1) % 3], hiring[(S - 1) % 3], G[S - 1], V[S - 1], pri, dua);
                    } else {
                        if (top) Step::project_prev(xprev, loring[(S_LAST - 1) % 3], hiring[(S_LAST - 1) % 3], G[S_LAST - 1], V[S_LAST - 1], pri, dua);
                        else Step::project_prev(xprev, loring[(S - 1) % 3], hiring[(S - 1) % 3], G[S - 1], V[S - 1], pri, dua);
                    }
                }
            }
            if (active) it_done = it1;  // admm.cpp:143
            if constexpr (TINY_E_RELAY_A == 1) relay_handover();
            E_STAMP(4);

            // ---- (KF) the families of all slots of this wavefront at once, one knot per lane: rows in, projections, duals, the
            // linear-cost term out through the same entry (read again by both backward passes; entry 0 keeps x_0)
            if constexpr (KF && TINY_E_EXP != 2 && TINY_E_EXP != 5) {
                e_lds_wait();
                e_static_for<0, NPASS>([&](auto Pp) {
                    constexpr int pp = decltype(Pp)::value;
                    double val[NXU], lxo[NXU];
                    e_static_for<0, NXU>([&](auto R) { val[R.value] = e_lds_read_async<(16 * pp * ES + R.value) * 8>(aKT); });
                    e_lds_wait_for(val);  // (the values are consumed by plain arithmetic: see e_lds_wait_for)
#if TINY_E_EXP == 1
                    e_static_for<0, NXU>([&](auto R) { lxo[R.value] = val[R.value]; });
#else
                    kf[pp].eval(val, lxo);
#endif
                    e_static_for<0, NXU>([&](auto R) { e_lds_write_masked<(16 * pp * ES + R.value) * 8>(aKT, lxo[R.value], kmask[pp]); });
                });
            }

            // ---- R1 (admm.cpp:93-101), this wavefront's share: which of its four instances have every lane below tolerance
            if (check) {
                const bool below = (pri < p.abs_pri_tol) && (dua * p.rho < p.abs_dua_tol);
                const unsigned long long b = __ballot(below);
                int bits = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) bits |= (((b >> (q * 16)) & 0xffffull) == 0xffffull) ? (1 << q) : 0;
                if (lane == 0) sFlag[wv] = bits;
            }

            // ================= backward (B1, admm.cpp:13-20); linear cost (L1, :77-82) recomputed from V, G =================
            // chain of a chunk of n slots: P <- q~_(n-1) [+ c_in];  for i = n-1 .. 0:  a = [q_(i-1) (i >= 1) + cb | cb] + Mb [P; r_i];
            // d_i = a (input lanes);  P = a (state lanes).  What comes out (state lanes) is p of the chunk's first knot MINUS its q,
            // which the chunk below owns.
            if constexpr (TINY_E_RELAY_A == 2) relay_handover();
            E_STAMP(5);
            load_ops(sMb, m);
            auto bwd_chain = [&](double cin, auto STORE) -> double {
                constexpr bool store = decltype(STORE)::value;
                const unsigned long long wr_d = __ballot(is_u && active);  // a zombie keeps the d of its last real iteration
                double px, rcur, rnext, acc;
                // head of the chain, from the chunk's last slot T: P and r_T, the accumulator start of step T, r_(T-1)
                auto head = [&](auto T, bool terminal) {
                    constexpr int tt = decltype(T)::value;
                    // state lanes: q~ of the last slot (the last wavefront: p_{N-1}, admm.cpp:81-82); input lanes: r of the last slot
                    const double lr1 = lr_of(T);
                    double lrT = lr1;
                    if (terminal) {
                        double pT = pnref;
                        if constexpr (KF) pT += sKX[(tt + 1) * ES + kxRow];
                        else if constexpr (FAM) {
                            if constexpr (LX_LDS) pT += sLX[tt * RS + famIdx];
                            else pT += LX[tt];
                        }
                        lrT = is_x ? pT : lr1;
                    }
                    const double lr2 = lr_of(std::integral_constant<int, tt - 1>{});
                    const double lrmc2 = is_x ? lr2 + cb : cb;
                    double t;
                    asm("v_add_f64 %[t], %[v1], -%[g1]\n\t"
                        "v_fma_f64 %[px], %[nrho], %[t], %[lrT]\n\t"
                        "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                        "v_fma_f64 %[acc], %[rhom], %[t], %[lrmc]\n\t"
                        "v_fma_f64 %[rn], %[nrho], %[t], %[lr]"
                        : [t] "=&v"(t), [px] "=&v"(px), [acc] "=&v"(acc), [rn] "=&v"(rnext)
                        : [v1] "v"(V[tt]), [g1] "v"(G[tt]), [v2] "v"(V[tt - 1]), [g2] "v"(G[tt - 1]), [nrho] "s"(nrho), [lrT] "v"(lrT),
                          [rhom] "v"(rhom), [lrmc] "v"(lrmc2), [lr] "v"(lr2));
                    rcur = px;                   // (input lanes: r_T)
                    px = is_x ? px + cin : px;   // (state lanes: + the carry entering from above)
                };
                auto block = [&](auto Sl) {
                    constexpr int s = decltype(Sl)::value;
                    constexpr int s2 = s >= 2 ? s - 2 : 0;  // slot feeding the tail
                    const double lr2 = lr_of(std::integral_constant<int, s2>{});
                    // tail: accumulator start of step s-1 (state lanes: q_(s-2 slot) + cb; step 0 takes NO q: its knot belongs to the
                    // chunk below) and the input-row operand of step s-2
                    const double lrmc2 = (s >= 2) ? (is_x ? lr2 + cb : cb) : cb;
                    const double rh = (s >= 2) ? rhom : 0.0;
                    double a = acc, an, rn;
                    prio_tick();
                    if constexpr (TINY_E_WOVEN != 0) Step::bwd_woven(a, px, rcur, m, V[s2], G[s2], rh, lrmc2, nrho, lr2, an, rn);
                    else Step::bwd(a, px, rcur, m, V[s2], G[s2], rh, lrmc2, nrho, lr2, an, rn);
                    if constexpr (store) {  // d_s
                        if constexpr (DREG) e_reg_write_masked(DR[s], a, wr_d);
                        else e_lds_write_masked<s * DS * 8>(aD, a, wr_d);
                    }
                    px = a;
                    rcur = rnext;
                    rnext = rn;
                    acc = an;
                };
                if constexpr (S_LAST == S) {
                    head(std::integral_constant<int, S - 1>{}, top);
                } else {
                    if (top) {
                        head(std::integral_constant<int, S_LAST - 1>{}, true);
                    } else {  // the S - S_LAST steps only a full chunk has, then the common part
                        head(std::integral_constant<int, S - 1>{}, false);
                        e_static_for<0, S - S_LAST>([&](auto I) { block(std::integral_constant<int, S - 1 - I.value>{}); });
                    }
                }
                e_static_for<0, S_LAST - 1>([&](auto I) { block(std::integral_constant<int, S_LAST - 1 - I.value>{}); });  // S_LAST-1 .. 1
                double a = acc;
                Step::bwd_last(a, px, rcur, m);
                if constexpr (store) {  // d of the chunk's first slot
                    if constexpr (DREG) e_reg_write_masked(DR[0], a, wr_d);
                    else e_lds_write_masked<0>(aD, a, wr_d);
                }
                return a;
            };
            // pass 1: from q~ alone (speculative: runs before the termination verdict, writes nothing)
            if constexpr (CUT) {
#if TINY_E_EXP != 3 && TINY_E_EXP != 5
                const double e2 = bwd_chain(0.0, std::false_type{});
#else
                const double e2 = V[0];
#endif
                sB[wv * 64 + lane] = e2;
            }
            E_STAMP(6);
            if constexpr (CUT) barrier_fb();
            else e_lds_wait();  // (one wavefront: its own flag write is all there is to wait for)
            relay_start();
            E_STAMP(7);
            // ---- termination, decided jointly: an instance converged iff every wavefront saw all of its lanes below tolerance
            if (check) {
                int all = 0xf;
#pragma unroll
                for (int q = 0; q < WPG; ++q) all &= sFlag[q];
                const bool conv = ((all >> j) & 1) != 0;
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
            // the carry entering from above: c_w = e''_(w+1) + Psi^S c_(w+1), c of the last wavefront = 0
            double cin = 0.0;
            if (CUT && !top) {
                double ps[16];
                load_pow(sPow + 256 + r, ps);
                cin = sB[(WPG - 1) * 64 + lane];
#pragma unroll 1
                for (int q = WPG - 2; q > wv; --q) cin = Step::fwd_plain(cin, 0.0, ps, sB[q * 64 + lane]);
                load_ops(sMb, m);
            }
            // pass 2: the real sweep; only d is kept
            if constexpr (TINY_E_RELAY_B == 1) relay_handover();
            E_STAMP(8);
            (void)bwd_chain(is_x ? cin : 0.0, std::true_type{});
            if constexpr (TINY_E_RELAY_B == 2) relay_handover();
            E_STAMP(9);
        }
    }
    e_lds_wait();

    // A converged solve returned before v <- vnew (admm.cpp:181-197): its canonical v|z is the previous iterate, i.e. the
    // stale copy. (The write-back above stored vnew there; this wave wrote both, in program order.)
    if (inst_ok && status == 1 && r < NXU) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        double *const wV = p.V + vbase + lane;
        const double *const wV2 = p.V2 + vbase + lane;
        if (bottom && is_x) wV[0] = wV2[0];
        for (int i = 0; i < S; ++i) {
            const int kn = s0 + i + koff;
            if (s0 + i < NS) wV[(size_t)kn * 64] = wV2[(size_t)kn * 64];
        }
    }

#if TINY_E_EXP == 9
    if (lane < 10 && grp * IPW < p.batch) p.sol_x[(size_t)grp * IPW * N * NX + wv * 10 + lane] = (double)(stamp[lane] - stamp[0]);
#endif
    // the four residual norms of the last check: rows, then wavefronts through LDS
    const double gpx = group_max<W>(is_x ? snap_pri : 0.0), gpu = group_max<W>(is_u ? snap_pri : 0.0);
    const double gdx = group_max<W>(is_x ? snap_dua : 0.0), gdu = group_max<W>(is_u ? snap_dua : 0.0);
    if constexpr (CUT) e_barrier();  // (sRes shares nothing, but every wavefront has left the iteration loop's LDS traffic behind)
    if (r < 4) sRes[(wv * 4 + j) * 4 + r] = (r == 0) ? gpx : (r == 1) ? gdx : (r == 2) ? gpu : gdu;
    if constexpr (CUT) e_barrier();
    else e_lds_wait();
    if (bottom && inst_ok && r == 0) {
        double res[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < WPG; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) res[c] = fmax(res[c], sRes[(q * 4 + j) * 4 + c]);
        p.istats[inst * 2 + 0] = it_done;
        p.istats[inst * 2 + 1] = status;
        if (res_valid) {
            p.dstats[inst * 4 + 0] = res[0];
            p.dstats[inst * 4 + 1] = res[1] * p.rho;
            p.dstats[inst * 4 + 2] = res[2];
            p.dstats[inst * 4 + 3] = res[3] * p.rho;
        }
    }
}

}  // namespace tinympc

#ifndef TINY_JIT_E_WPS
#define TINY_JIT_E_WPS (TINY_JIT_E_WPG / 4)  // wavefronts per SIMD: a workgroup of 8 shares a CU two by two (256 registers each)
#endif
#ifndef TINY_JIT_E_GPW
#define TINY_JIT_E_GPW 1  // groups of four instances per workgroup (more than one only where a group is ONE wavefront, TINY_JIT_E_WPG = 1)
#endif
// the entry point: `tinympc_jit_solve` as a run-time specialisation (tinympc_jit.hip looks it up by that name), the name the build
// gives it as a compiled-in one (TINY_BUILTIN: __graft_entry__.HIP_BUILTINS)
#ifdef TINY_BUILTIN
#define TINY_KERNEL_NAME TINY_BUILTIN_NAME
#else
#define TINY_KERNEL_NAME tinympc_jit_solve
#endif
extern "C" __global__ void __launch_bounds__(64 * TINY_JIT_E_WPG * TINY_JIT_E_GPW) __attribute__((amdgpu_waves_per_eu(TINY_JIT_E_WPS, TINY_JIT_E_WPS)))
TINY_KERNEL_NAME(const tinympc::SolveParams p) {
    constexpr bool CTJ = TINY_JIT_CT != 0, FAMJ = TINY_JIT_FAM != 0;
    constexpr bool GCJ = TINY_JIT_E_GC_LDS != 0, GLJ = TINY_JIT_E_GL_LDS != 0, LXJ = TINY_JIT_E_LX_LDS != 0, KFJ = TINY_JIT_E_KFAM != 0, DRJ = TINY_JIT_E_DREG != 0;
    constexpr int nlds = !FAMJ ? 0 : KFJ ? -1 : (GCJ ? 1 : 0) + (GLJ ? 1 : 0) + (LXJ ? 1 : 0);
    constexpr size_t bytes = tinympc::e_lds_bytes(TINY_JIT_NX + TINY_JIT_NU, TINY_JIT_NU, TINY_JIT_N, CTJ, TINY_JIT_E_WPG, TINY_JIT_E_S, FAMJ, tinympc::E_NL, nlds, tinympc::E_NCONE, TINY_JIT_E_GPW);
    static_assert(bytes <= 160 * 1024, "layout E: the workgroup's LDS plan exceeds a CU");
    __shared__ __attribute__((aligned(16))) double smem_e[bytes / sizeof(double)];
    tinympc::k_admm_solve_e_body<TINY_JIT_NX, TINY_JIT_NU, TINY_JIT_N, CTJ, TINY_JIT_E_WPG, TINY_JIT_E_S, FAMJ, GCJ, GLJ, LXJ, KFJ, DRJ, TINY_JIT_E_GPW>(p, smem_e);
}
