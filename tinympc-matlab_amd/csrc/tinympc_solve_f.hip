// tinympc_solve_f.hip -- layout F: the LATENCY kernel as a run-time specialisation. One MPC instance per workgroup, the horizon
// cut into up to 32 chunks that the DPP rows of up to eight wavefronts sweep concurrently -- layout C's idea (tinympc_solve_c.hip:
// both sweeps are linear time-invariant recurrences, so a chunk can be swept from a zero incoming state, the true incoming states
// come from a carry scan, and a second pass IS the sequential sweep) with layout E's means:
//   * the shape (nx, nu, N), the chunk plan and the STRUCTURE of the cone / linear families are compile-time constants
//     (tinympc_jit.hip), so a sweep step is layout D's single asm block with exactly nx+nu columns, and the families' cross-lane
//     sums are a handful of DPP instructions under an EXEC mask instead of three masked 12-column mat-vecs and their 72 mask
//     registers (tinympc_solve_e_common.h);
//   * 150-odd VGPRs instead of 418, so the workgroup's wavefronts sit TWO per SIMD and fill each other's barrier and LDS waits,
//     where layout C's lone wavefront per SIMD issues VALU 55 % of the time (profiles/r03_rocket_instance_pmc.json);
//   * chunks of 3-4 slots instead of 7-8 (32 rows instead of 16): half the serial work per sweep;
//   * no linear-cost exchange between neighbours (layout E's cut: a chunk's backward chain starts from the q~ of its OWN last state
//     slot and leaves out the q of its first knot, which the chunk below adds from its own registers), the termination ballots ride
//     on the backward scan's barrier: two barriers per ADMM iteration instead of three.
// Chunk c = 4 w + j is DPP row j of wavefront w and owns slots [c S, (c+1) S); the last chunk in use owns what is left, rows
// beyond it idle (their steps sit behind lane predicates -- whole DPP rows, so no DPP read crosses the mask).
// Carry scan per sweep (as in layout C): inside a wavefront the four rows' prefix by cross-row swaps (P^S, P^2S), the
// wavefronts' totals through LDS behind ONE barrier and a Horner recurrence with P^4S, a last mat-vec (P^jS) per row.
// Given exact carries pass 2 IS the sequential sweep; results differ from the other layouts through the rounding of the carries
// (~1e-14 relative), iteration counts match the restatement in every test. Same persistent HBM state as every other kernel;
// single-instance handles' pinned-host paths (x0 in, solution / statistics / completion stamp out) as in layout C. No adaptive rho.
// SESSION (round 4, TINY_JIT_F_SESSION): the resident closed-loop variant -- layout C's mailbox protocol (tinympc_session.hip,
// tinympc_solve_c.hip) around this kernel's iteration, so that the rocket landing's closed loop (rocket_landing_constraints.m:86-121:
// N = 100, new references every tick) runs without a launch per tick; the latency kernel's own session ends at N = 65 with families.
// The ADMM state stays in registers from tick to tick; references live in the LDS copy of the tables (always per-knot tables here),
// refreshed from pinned memory on request or shifted by one knot with the new last column that came with the command.
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

#if !defined(TINY_JIT) && !defined(TINY_BUILTIN)  // stand-alone instance (ISA lint, "does it compile"): the rocket landing of BASELINE config 4
#define TINY_CHAIN_NOP 1
#define TINY_JIT_NX 6
#define TINY_JIT_NU 3
#define TINY_JIT_N 100
#define TINY_JIT_CT 0
#define TINY_JIT_FAM 1
#define TINY_JIT_F_WPG 7
#define TINY_JIT_F_S 4
#define TINY_JIT_E_NROUND 1
#define TINY_JIT_E_NCONE 2
#define TINY_JIT_E_CONES {0, 0, 2}, {0, 6, 8}
#define TINY_JIT_E_NLX 1
#define TINY_JIT_E_NLU 0
#endif
#ifndef TINY_JIT_F_KFAM
#define TINY_JIT_F_KFAM 0  // 1: the families one KNOT per lane (KFamilies, tinympc_solve_e_common.h) instead of one element per lane
#endif
// Timing experiment (tools/f_breakdown.py, through TINYMPC_JIT_DEFS=-DTINY_F_STAMP=1; the states of the solution are overwritten):
// the shader clock at the phase boundaries of iteration 10, per wavefront
#ifndef TINY_F_STAMP
#define TINY_F_STAMP 0
#endif
#if TINY_F_STAMP
#define F_STAMP(k) do { if (it0 == 10) stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define F_STAMP(k) do { } while (0)
#endif
#ifndef TINY_F_OPS_RESIDENT
#define TINY_F_OPS_RESIDENT 1  // 0 (experiments): the sweep operators' rows are read from LDS at the top of every sweep
#endif
#ifndef TINY_F_TSUM
#define TINY_F_TSUM 1  // 0 (experiments): the first forward pass sweeps every chunk from a zero incoming state, as in rounds 3-4a
#endif
#ifndef TINY_JIT_F_SESSION
#define TINY_JIT_F_SESSION 0
#endif

namespace tinympc {
template <int NX, int NU>
struct DStep;  // tinympc_solve_d_chain.h
}  // namespace tinympc
#define D_NX TINY_JIT_NX
#define D_NU TINY_JIT_NU
#include "tinympc_solve_d_chain.h"
#include "tinympc_solve_e_common.h"

namespace tinympc {

// (a reference to the 16-entry one of two arrays, whichever it is)
template <int A, int B>
__device__ __forceinline__ double (&ops_pick(double (&a)[A], double (&b)[B]))[16] {
    if constexpr (A == 16) return a;
    else return b;
}

// KFAM (round 4): layout E's knot-per-lane families here. The four DPP rows of a wavefront are four CHUNKS of the one instance; lane
// (row j, entry t) takes knot t of chunk 4 wv + j: after the forward sweep has left the wavefront's slots in an LDS exchange buffer it
// picks up all rows of its knot, evaluates every cone and linear row once per iteration, keeps the knot's duals gc | gl in its
// registers for the whole solve, and hands the linear-cost term back through the buffer -- the S evaluations of ~70 dependent FP64
// instructions per iteration that sat in every wavefront's sweep become one.
template <int NX, int NU, int N, bool CT, int WPG, int S, bool FAM, bool SESSION, bool KFAM = false>
__device__ __forceinline__ void k_admm_solve_f_body(const SolveParams &p, double *smem) {
    static_assert(!SESSION || !CT, "layout F: the session kernel keeps its references in the LDS copy of the per-knot tables");
    constexpr int W = 16, NXU = NX + NU, NS = N - 1, DS = 4 * NU;
    constexpr int KT = NXU <= 8 ? 8 : NXU <= 12 ? 12 : 16;  // row stride of p.ops (choose_geometry)
    constexpr int KS = NX <= 8 ? 8 : NX <= 12 ? 12 : 16;    // row stride of the carry matrices (chunk_ks)
    constexpr int TOFF = (N + 2) * W;
    constexpr int NCH = (NS + S - 1) / S;                    // chunks in use
    constexpr int S_LAST = NS - (NCH - 1) * S;               // slots of the last chunk in use, 1 .. S
    static_assert(S >= 2 && NCH >= 2 && NCH <= 4 * WPG && NCH > 4 * (WPG - 1), "layout F: chunk plan");
    using Step = DStep<NX, NU>;
    constexpr bool KF = FAM && KFAM;
    constexpr bool OPSR = TINY_F_OPS_RESIDENT != 0 && WPG <= 8 && S <= 8;
    // TSUM (round 4): no first forward pass. A chunk's end state from a zero incoming state is linear in its inputs,
    //     e = sum_s Phi^(S-1-s) (-B d_s + cf) = sum_s T_s d_s + aff   (+ Phi^S x_0 for chunk 0),
    // and the d_s are what the backward sweep's second pass has just produced: it accumulates T_s d_s on the way (NU FMAs per slot
    // instead of a whole sweep step), the matrices T_s and aff come from k_build_f_input_tables (p.ftab).
    // Measured (one instance, microseconds per iteration): quadrotor N=50 2.65 -> 2.58, cartpole N=20 1.54 -> 1.49, rocket N=44 / 20 (element
    // form) 2.97 -> 2.86 / 2.45 -> 2.35; with the knot-per-lane families' long chunks (rocket N=100, S=7) 3.99 -> 4.07: not there.
    constexpr bool TSUM = TINY_F_TSUM != 0 && !KF;
    constexpr int ES = kfam_es(NXU);
    static_assert(!KF || S + 1 <= 16, "layout F, knot-per-lane families: one pass of 16 entries per chunk");

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane >> 4, r = lane & 15;  // j: DPP row = chunk within the wavefront
    const int c = 4 * wv + j;                // chunk
    const long inst = blockIdx.x;
    const bool is_x = r < NX;
    const bool is_u = (r >= NX) && (r < NXU);
    const bool row_ok = r < NXU;
    const int koff = is_x ? 1 : 0;           // slot s = knot s+1 on state lanes, knot s on input lanes
    const bool topc = c == NCH - 1, bottomc = c == 0;
    const int nsl = c < NCH - 1 ? S : (topc ? S_LAST : 0);  // real slots of this row
    const int s0 = c * S;

    // ---- LDS
    double *sOps = smem;                                    // [2][16 k][16 r]
    double *sT = sOps + 512;                                // tables (!CT)
    double *sLin = sT + (CT ? 0 : 3 * (N + 2) * 16 + 16);   // [E_NL][3][16] (FAM)
    double *sMu = sLin + (FAM ? 3 * E_NL * 16 : 0);         // (KF) [2][E_NCONE] the cones' slopes | their reciprocals
    double *sPow = sMu + (KF ? ((2 * E_NCONE + 1) & ~1) : 0);  // [2 Phi|Psi][4 levels: powers S, 2S, 3S, 4S][16 k][16 r]
    double *sY = sPow + 2 * 4 * 256;                        // [2][WPG][16] the wavefronts' totals of a scan, double-buffered
    int *sFlag = reinterpret_cast<int *>(sY + 2 * WPG * 16);  // [2][WPG] "every lane below tolerance" per wavefront (16 doubles)
    double *sRes = sY + 2 * WPG * 16 + 16;                  // [4 WPG][4] residual maxima per chunk
    double *sMail = sRes + 4 * WPG * 4;                     // [64] the session's mailbox as last polled (+ the poller's verdict)
    double *sD = sMail + 64 + (size_t)wv * ((S * DS + 1) & ~1);  // per wavefront: d[S][4 rows x nu]
    double *sKX = sMail + 64 + (size_t)WPG * ((S * DS + 1) & ~1) + (size_t)wv * kfam_doubles(NXU, S);  // (KF) per wavefront: the exchange buffer
    double *sTd = sMail + 64 + (size_t)WPG * ((S * DS + 1) & ~1) + (KF ? (size_t)WPG * kfam_doubles(NXU, S) : 0);  // [S][NU][16] T_s | aff[16]
    // (KF) this lane's two roles in the exchange buffer: as row r of chunk-row j (slot q = entry q+1: kxRow + (q+1) ES) and as entry
    // t = r of chunk-row j (kxT .. + nx+nu)
    const int kxRow = j * (S + 1) * ES + (r < NXU ? r : NXU - 1);
    const int kxT = (j * (S + 1) + r) * ES;

    // references left in pinned host memory by set_x_ref / set_u_ref (single-instance handles; the closed loop with per-tick references,
    // rocket_landing_constraints.m:86-121): this workgroup rebuilds the table rows they determine before anything reads them -- the tick
    // stays ONE launch, as on layout C. (The resident kernel has its own path: flags 2 / 4 / 8 of a command.)
    if constexpr (!SESSION) refresh_reference_tables(p, W, KT);
    for (int i = threadIdx.x; i < 512; i += 64 * WPG) {
        const int which = i >> 8, k = (i >> 4) & 15, rr = i & 15;
        sOps[i] = (k < KT) ? p.ops[(size_t)which * W * KT + (size_t)rr * KT + k] : 0.0;
    }
    for (int i = threadIdx.x; i < 2 * 4 * 256; i += 64 * WPG) {
        const int mat = i >> 8, k = (i >> 4) & 15, rr = i & 15;  // mat = which * 4 + level, as k_build_chunk_tables lays them out
        sPow[i] = (k < NX && rr < NX) ? p.ctab[(size_t)mat * W * KS + (size_t)rr * KS + k] : 0.0;
    }
    if constexpr (!CT)
        for (int i = threadIdx.x; i < 3 * (N + 2) * 16 + 16; i += 64 * WPG) sT[i] = p.tables[i];
    for (int i = threadIdx.x; i < S * NU * 16 + 16; i += 64 * WPG) sTd[i] = p.ftab[i];
    if constexpr (FAM) EFamilies<NX, NU>::stage_linear_rows(p.fam, KT, sLin, (int)threadIdx.x, 64 * WPG);
    if constexpr (KF) {
        KFamilies<NX, NU>::stage_cone_slopes(p.fam, KT, sMu, (int)threadIdx.x);
        for (int i = lane; i < kfam_doubles(NXU, S); i += 64) sKX[i] = 0.0;
    }

    // canonical HBM layout shared with the other kernels (instance = lane group inst % 4 of wave group inst / 4)
    const long wg = inst >> 2;
    const int jj = (int)(inst & 3);
    const size_t vbase = ((size_t)wg * v_rows(N) + V_PAD) * 64 + jj * 16 + r;
    double *const gG = p.G + (size_t)wg * (N + 1) * 64 + jj * 16 + r;  // row kn = knot kn
    double *const gV = p.V + vbase;
    double *const gD = p.D + (size_t)wg * (size_t)(NS * DS) + jj * NU + (is_u ? r - NX : 0);
    const int dIdx = j * NU + (is_u ? r - NX : 0);

    // ---- this lane's elements, in registers for the whole solve: slot i <-> step s0 + i
    double G[S], V[S], Vp[S], GC[(FAM && !KF) ? S : 1], GL[(FAM && !KF) ? S : 1], LX[(FAM && !KF) ? S : 1];
    e_static_for<0, S>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const bool on = (i < nsl) && row_ok;
        const size_t kn = on ? (size_t)(s0 + i + koff) : 0;
        G[i] = on ? gG[kn * 64] : 0.0;
        V[i] = on ? gV[kn * 64] : 0.0;
        Vp[i] = V[i];
        if constexpr (FAM && !KF) {
            GC[i] = on ? (p.GC + vbase)[kn * 64] : 0.0;
            GL[i] = on ? (p.GL + vbase)[kn * 64] : 0.0;
            LX[i] = 0.0;
        }
        if ((i < nsl) && is_u) sD[i * DS + dIdx] = gD[(size_t)(s0 + i) * DS];
    });
    // knot 0 of the state rows: chunk 0, state lanes
    const bool k0 = bottomc && is_x;
    double G0 = k0 ? gG[0] : 0.0, V0 = k0 ? gV[0] : 0.0, V0p = V0;
    double GC0 = (FAM && !KF && k0) ? (p.GC + vbase)[0] : 0.0, GL0 = (FAM && !KF && k0) ? (p.GL + vbase)[0] : 0.0;
    double x0v = k0 ? p.x0[inst * NX + r] : 0.0;
    if (p.x0_mirror && k0) p.x0_mirror[inst * NX + r] = x0v;  // x0 came from pinned host memory: keep the device copy current
    // (KF) the duals of this lane's knot -- entry t = r of chunk-row j: state rows of knot s0 + t, input rows of knot s0 + t - 1; entry
    // 0 exists in chunk 0 only (knot 0 of the state rows) -- from the canonical HBM layout every kernel shares
    KFamilies<NX, NU> kf;
    const bool kent = KF && ((r >= 1 && r <= nsl) || (r == 0 && bottomc));  // this lane holds an entry
    const unsigned long long kmask = __ballot(KF && r >= 1 && r <= nsl);     // ... that is a slot: it hands a linear-cost term back
    if constexpr (KF) {
        const double *const bGC = p.GC + (vbase - r), *const bGL = p.GL + (vbase - r);
        e_static_for<0, NXU>([&](auto R) {
            constexpr int rr = decltype(R)::value;
            const int kn = rr < NX ? s0 + r : s0 + r - 1;
            const bool ok = kent && (rr < NX || r >= 1);
            kf.gc[rr] = ok ? bGC[(size_t)kn * 64 + rr] : 0.0;
            kf.gl[rr] = ok ? bGL[(size_t)kn * 64 + rr] : 0.0;
        });
    }
    __syncthreads();  // (the only barrier that also waits for global loads)
    if constexpr (KF) {
        kf.init(p.fam, KT, sLin, sMu, p.rho);
        if (bottomc && row_ok) sKX[kxRow] = is_x ? x0v : 0.0;  // entry 0 of chunk 0: x_0
    }

    EFamilies<NX, NU> fam_eval;
    if constexpr (FAM && !KF) fam_eval.init(p.fam, KT, sLin, r, p.rho);

    const double cf = p.ops[(size_t)2 * W * KT + r];
    const double cb = p.ops[(size_t)2 * W * KT + W + r];
    double pnref = p.tables[(size_t)3 * TOFF + r];
    const double nrho = -p.rho;
    const double rhom = is_x ? nrho : 0.0;
    const double lo_c = p.tables[W + r], hi_c = p.tables[(size_t)TOFF + W + r], lr_c = p.tables[(size_t)2 * TOFF + W + r];
    const double *const sTl = sT + (size_t)(s0 + koff) * W + r;  // (!CT) row of local slot i: sTl[(i + 1) * W]  (idle rows: never read)
    const double *const sMf = sOps + r, *const sMb = sOps + 256 + r;
    const int ct = p.check_termination;

    auto load_ops = [&](const double *src, double (&m)[16]) {
        e_static_for<0, 16>([&](auto K) { m[K.value] = src[(K.value < NXU ? K.value : 0) * 16]; });
    };
    auto load_pow = [&](const double *mat, double (&m)[KS]) {  // a carry matrix row (state lanes; zero elsewhere)
        e_static_for<0, KS>([&](auto K) { m[K.value] = (K.value < NX && is_x) ? mat[K.value * 16 + r] : 0.0; });
    };
    auto lr_of = [&](auto I) -> double {  // linref of local slot I (+ the families' term)
        double base;
        if constexpr (CT) base = lr_c;
        else base = (decltype(I)::value < nsl) ? sTl[2 * TOFF + (I.value + 1) * W] : 0.0;
        if constexpr (KF) base += (decltype(I)::value < nsl) ? sKX[(I.value + 1) * ES + kxRow] : 0.0;
        else if constexpr (FAM) base += LX[I.value];
        return base;
    };

    // ---- carry scan over the chunks (layout C's, for up to eight wavefronts). In: the chunk's pass-1 end value (state lanes; 0
    // elsewhere and in idle rows). Out: the true value ENTERING the chunk from its neighbour c + dir,
    //   I_c = sum_{q>=1} Pm^((q-1) S) end_(c + q dir).
    //   A  inside the wavefront, rows only (cross-row swaps):  t_j = e_j + P1 e_(j-1);  L_j = t_j + P2 t_(j-2)
    //   B  the wavefront's total goes to LDS; ONE barrier; Horner over the wavefronts before it with P4
    //   C  I_j = L_(j-1) + P_j Cin for the rows behind the first; the first row's incoming value is Cin itself.
    int cur = 0;
    auto rows_shift1 = [&](int dir, double x) -> double {  // row j <- row j + dir (the wavefront's edge row <- 0)
        double e, o, lo2, hi2;
        cross_row_pair<1>(x, e, o);  // e = [r0 r0 r2 r2], o = [r1 r1 r3 r3]
        if (dir < 0) {
            cross_row_pair<0>(o, lo2, hi2);  // lo2 = [r1 r1 r1 r1]
            return j == 0 ? 0.0 : (j == 2 ? lo2 : e);
        } else {
            cross_row_pair<0>(e, lo2, hi2);  // hi2 = [r2 r2 r2 r2]
            return j == 3 ? 0.0 : (j == 1 ? hi2 : o);
        }
    };
    auto rows_shift2 = [&](int dir, double x) -> double {  // row j <- row j + 2 dir
        double lo2, hi2;
        cross_row_pair<0>(x, lo2, hi2);  // lo2 = [r0 r1 r0 r1], hi2 = [r2 r3 r2 r3]
        return dir < 0 ? (j >= 2 ? lo2 : 0.0) : (j < 2 ? hi2 : 0.0);
    };
    auto carry_scan = [&](int dir, const double *Pm, double end_val) -> double {
        const int jr = dir < 0 ? j : 3 - j;  // rows counted from the side the carry comes from
        double L;
        {
            double m1[KS], m2[KS];
            load_pow(Pm, m1);
            load_pow(Pm + 256, m2);
            const double t = end_val + group_matvec<W, KS>(m1, rows_shift1(dir, end_val), 0.0);
            L = t + group_matvec<W, KS>(m2, rows_shift2(dir, t), 0.0);
        }
        if constexpr (WPG == 1) {
            // ONE wavefront (up to four chunks: short horizons, round 4): what enters a row is the prefix of the rows before it -- no
            // totals to exchange, no barrier in the whole iteration
            const double Lprev1 = rows_shift1(dir, L);
            return is_x ? (jr == 0 ? 0.0 : Lprev1) : 0.0;
        }
        double m4[KS], mj[KS];
        load_pow(Pm + 3 * 256, m4);
        load_pow(Pm + (size_t)(jr >= 1 ? jr - 1 : 0) * 256, mj);
        if (jr == 3 && is_x) sY[(cur * WPG + wv) * 16 + r] = L;
        const double Lprev = rows_shift1(dir, L);
        e_barrier();
        // totals of the wavefronts the carry comes from, nearest last: cin = T_far; cin = T_next + P4 cin; ...
        double tv[WPG > 1 ? WPG - 1 : 1];
        e_static_for<0, WPG - 1>([&](auto Q) {
            const int v = dir < 0 ? Q.value : WPG - 1 - Q.value;  // forward: 0, 1, ..; backward: WPG-1, WPG-2, ..
            tv[Q.value] = is_x ? sY[(cur * WPG + v) * 16 + r] : 0.0;
        });
        const int nbefore = dir < 0 ? wv : WPG - 1 - wv;  // wavefronts on the side the carry comes from (uniform)
        double cin = nbefore > 0 ? tv[0] : 0.0;
        e_static_for<1, WPG - 1>([&](auto Q) {
            if (Q.value < nbefore) cin = tv[Q.value] + group_matvec<W, KS>(m4, cin, 0.0);
        });
        const double far = Lprev + group_matvec<W, KS>(mj, cin, 0.0);
        cur ^= 1;
        return is_x ? (jr == 0 ? cin : far) : 0.0;
    };

    // TSUM: e += T_s d_s (d_s on the input lanes of this row, T_s's row r from LDS)
    double eacc = 0.0;
    const double aff_r = (TSUM && is_x) ? sTd[S * NU * 16 + r] : 0.0;
    double c0 = 0.0;  // Phi^S x_0 (chunk 0; set where x_0 is known: below, and per tick in a session)
    auto acc_T = [&](auto Sl, double dval) {
        constexpr int sl = decltype(Sl)::value;
        double tm[16];
        e_static_for<0, 16>([&](auto K) {
            constexpr int kk = decltype(K)::value;
            tm[kk] = (kk >= NX && kk < NXU) ? sTd[(sl * NU + (kk - NX)) * 16 + r] : 0.0;
        });
        eacc = Step::acc_inputs(eacc, dval, tm);
    };
    auto set_c0 = [&]() {
        if constexpr (TSUM) {
            double mp[KS];
            load_pow(sPow, mp);  // Phi^S (the forward scan's first level)
            const double v = group_matvec<W, KS>(mp, k0 ? x0v : 0.0, 0.0);
            c0 = (bottomc && is_x) ? v : 0.0;
        }
    };
    if constexpr (TSUM) {  // the inputs this solve (or session) starts from: once, from the d it loaded -- in the backward sweep's own
        // order, last slot first, so that a launched tick starts from the very bits a resident kernel carries over from its last sweep
        e_static_for<0, S>([&](auto I) {
            constexpr int i = S - 1 - decltype(I)::value;
            if (i < nsl) acc_T(std::integral_constant<int, i>{}, sD[i * DS + dIdx]);
        });
        set_c0();
    }
    int it_done = 0, status = 11;  // TINY_UNSOLVED (admm.cpp:114)
    bool res_valid = false, converged = false;
    double snap_pri = 0.0, snap_dua = 0.0;

    // ---- write-back of the persistent ADMM state (one-shot: after the solve; session: when the kernel leaves). A converged solve
    // returned before v <- vnew (admm.cpp:181-197): its canonical slack is the previous iterate.
    auto write_state = [&](bool conv) {
        e_static_for<0, S>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if ((i < nsl) && row_ok) {
                const size_t k = (size_t)(s0 + i), kn = k + koff;
                gG[kn * 64] = G[i];
                gV[kn * 64] = conv ? Vp[i] : V[i];
                if constexpr (FAM && !KF) {
                    (p.GC + vbase)[kn * 64] = GC[i];
                    (p.GL + vbase)[kn * 64] = GL[i];
                }
                if (is_u) gD[k * DS] = sD[i * DS + dIdx];
            }
        });
        if (k0) {
            gG[0] = G0;
            gV[0] = conv ? V0p : V0;
            if constexpr (FAM && !KF) {
                (p.GC + vbase)[0] = GC0;
                (p.GL + vbase)[0] = GL0;
            }
        }
        if constexpr (KF) {  // the families' duals leave from the knot-per-lane layout
            if (kent) {
                double *const wGC = p.GC + (vbase - r), *const wGL = p.GL + (vbase - r);
                e_static_for<0, NXU>([&](auto R) {
                    constexpr int rr = decltype(R)::value;
                    if (rr < NX || r >= 1) {
                        const size_t kn = (size_t)(rr < NX ? s0 + r : s0 + r - 1);
                        wGC[kn * 64 + rr] = kf.gc[rr];
                        wGL[kn * 64 + rr] = kf.gl[rr];
                    }
                });
            }
        }
    };

#if TINY_F_STAMP
    unsigned long long stamp[8] = {};
#endif
    // The two sweep operators' rows: in registers for the whole solve where the plan leaves room (one or two wavefronts per SIMD, chunks
    // of up to eight slots: 32 VGPRs on top of <= 210) -- quadrotor N=50 2.73 -> 2.65 us per iteration, N=20 2.12 -> 2.01 --, else
    // re-read from LDS at the top of every sweep.
    double mF[OPSR ? 16 : 1], mB[OPSR ? 16 : 1];
    if constexpr (OPSR) {
        load_ops(sMf, mF);
        load_ops(sMb, mB);
    }
    const int max_iter = p.max_iter;
    const int tid = (int)threadIdx.x;
    double expect = p.session_expect;
    unsigned long long t_prev_out = 0ull, t_seen = 0ull;  // (SESSION diagnostics: 100 MHz stamps, see the completion stamp below)
    for (;;) {  // ---- SESSION: every pass is one closed-loop tick (one pass otherwise)
    if constexpr (SESSION) {
        // Poll the mailbox (layout: SolveParams::mail; protocol and checksum as in tinympc_solve_c.hip): lanes 0..55 fetch its lines
        // in one load, everybody takes the same decision from LDS. The poller's clock ends the session after p.session_idle
        // ticks without a command -- the exit every wavefront reaches even if the host process is gone.
        const unsigned long long t_idle0 = __builtin_amdgcn_s_memrealtime();
        bool go = false, quit = false;
        while (!go && !quit) {
            if (tid < 64) {  // wavefront 0 fetches the seven lines in one load and decides in registers (mail_lines_ok, tinympc_device.h)
                const double w = tid < 56 ? __hip_atomic_load(p.mail + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
                const unsigned ok = mail_lines_ok(w, tid, expect);
                const unsigned long long w0 = (unsigned long long)__builtin_bit_cast(long long, w);
                const double flags_word = __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w0 >> 32)) << 32) |
                                                                                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)w0)));
                const int f0 = (ok & 1u) ? (int)flags_word : 0;  // line 0 carries the flags, which say how many lines the command uses
                const int npay = 1 + NX + ((f0 & 4) ? NX : 0) + ((f0 & 8) ? NU : 0), nlines = (npay + 6) / 7;
                const unsigned need = (1u << nlines) - 1u;
                if (tid < 56) sMail[tid] = w;
                if (tid == 0) {
                    sMail[56] = (__builtin_amdgcn_s_memrealtime() - t_idle0 > p.session_idle) ? 1.0 : 0.0;
                    sMail[57] = ((ok & need) == need) ? 1.0 : 0.0;
                }
            }
            __syncthreads();
            go = sMail[57] != 0.0;
            quit = !go && sMail[56] != 0.0;
            __syncthreads();  // (the next poll overwrites sMail)
        }
        t_seen = __builtin_amdgcn_s_memrealtime();
        const int flags = go ? (int)sMail[0] : 1;
        if (quit || (flags & 1)) {  // stop requested, or nobody is talking to this kernel any more
            write_state(false);     // (a converged tick already rolled its slack back, see the end of the loop)
            break;
        }
        auto payload = [&](int q) -> double { return sMail[8 * (q / 7) + q % 7]; };
        if (k0) {
            x0v = payload(1 + r);
            if (p.x0_mirror) p.x0_mirror[inst * NX + r] = x0v;
        }
        if constexpr (KF) {
            if (bottomc && row_ok) sKX[kxRow] = is_x ? x0v : 0.0;  // entry 0 of chunk 0: this tick's x_0
        }
        set_c0();
        double *const tab = const_cast<double *>(p.tables);  // (mirror: the table rows the other kernels read)
        if (flags & 2) {
            // the references changed: fetch them again from the pinned copies (tinympc_set_x_ref / _u_ref filled them), derive the
            // table rows -(Xref o Q), -(Uref o R) (the expressions of k_build_tables) and pNref. Acquire: the stamp was observed,
            // what the host wrote before it must be too.
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            const double *const dg = p.ops + (size_t)2 * W * KT + 2 * W;
            for (int i = tid; i < NX * N; i += 64 * WPG) {
                const double x = p.href_x[i];
                const int kn = i / NX, rr = i % NX;
                p.dXref[i] = x;
                const double v = -(x * dg[rr]);
                sT[2 * TOFF + (kn + 1) * W + rr] = v;
                tab[(size_t)2 * TOFF + (size_t)(kn + 1) * W + rr] = v;
            }
            for (int i = tid; i < NU * NS; i += 64 * WPG) {
                const double u = p.href_u[i];
                const int kn = i / NU, rr = NX + i % NU;
                p.dUref[i] = u;
                const double v = -(u * dg[rr]);
                sT[2 * TOFF + (kn + 1) * W + rr] = v;
                tab[(size_t)2 * TOFF + (size_t)(kn + 1) * W + rr] = v;
            }
            double acc = 0.0;  // pNref = -(Xref_{N-1}' Pinf)' (admm.cpp:81), the sum term by term in the order of k_build_tables
            if (is_x) {
#pragma unroll
                for (int q = 0; q < NX; ++q) acc += p.href_x[q + (size_t)(N - 1) * NX] * p.Pinf[q + (size_t)r * NX];
                acc = -acc;
            }
            pnref = acc;
            if (wv == 0 && j == 0) tab[(size_t)3 * TOFF + r] = pnref;
        } else if (flags & 12) {
            // Receding horizon (rocket_landing_constraints.m:96-101): the new reference is the previous one moved up by one knot plus
            // ONE new last column, which came with the command -- no 7 KB PCIe read, a shift of the LDS table's rows. The host
            // checked, bit for bit, that it IS a shift; the device copies / global tables lag until the session ends (the host knows).
            const double *const dg = p.ops + (size_t)2 * W * KT + 2 * W;
            constexpr int NT = 64 * WPG, NEL = (N - 1) * W, NQ = (NEL + NT - 1) / NT;
            double tmp[NQ];
            e_static_for<0, NQ>([&](auto Q) {
                const int idx = tid + Q.value * NT, kn = idx >> 4, rr = idx & 15;
                const bool mine = rr < NX ? ((flags & 4) != 0 && kn <= N - 2) : (rr < NXU ? ((flags & 8) != 0 && kn <= N - 3) : false);
                tmp[Q.value] = (idx < NEL && mine) ? sT[2 * TOFF + (kn + 2) * W + rr] : 0.0;
            });
            __syncthreads();
            e_static_for<0, NQ>([&](auto Q) {
                const int idx = tid + Q.value * NT, kn = idx >> 4, rr = idx & 15;
                const bool mine = rr < NX ? ((flags & 4) != 0 && kn <= N - 2) : (rr < NXU ? ((flags & 8) != 0 && kn <= N - 3) : false);
                if (idx < NEL && mine) sT[2 * TOFF + (kn + 1) * W + rr] = tmp[Q.value];
            });
            if (tid < NXU) {
                const int rr = tid;
                if (rr < NX && (flags & 4)) sT[2 * TOFF + N * W + rr] = -(payload(1 + NX + rr) * dg[rr]);                                        // knot N-1
                if (rr >= NX && (flags & 8)) sT[2 * TOFF + (N - 1) * W + rr] = -(payload(1 + NX + ((flags & 4) ? NX : 0) + (rr - NX)) * dg[rr]);  // knot N-2
            }
            if (flags & 4) {  // pNref from the new last column
                double acc = 0.0;
                if (is_x) {
#pragma unroll
                    for (int q = 0; q < NX; ++q) acc += payload(1 + NX + q) * p.Pinf[q + (size_t)r * NX];
                    acc = -acc;
                }
                pnref = acc;
            }
        }
        __syncthreads();  // sMail has been read by everyone; the table rows are in place
        it_done = 0;
        status = 11;
        res_valid = false;
        converged = false;
        snap_pri = snap_dua = 0.0;
    }
    for (int it = 0; it < max_iter; ++it) {  // admm.cpp:129
        const int it0 = __builtin_amdgcn_readfirstlane(it);
        const bool check = __builtin_amdgcn_readfirstlane((int)((ct > 0) && (((it0 + 1) % ct) == 0))) != 0;  // admm.cpp:91
        F_STAMP(0);
        double mloc[OPSR ? 1 : 16];  // (OPSR: both operators' rows live in registers)
        double (&m)[16] = ops_pick(mF, mloc);
        if constexpr (!OPSR) load_ops(sMf, m);
        // ================= forward, pass 1: the chunk's end state from a zero incoming state (chunk 0: from x_0) =================
        double xt;
        if constexpr (TSUM) {
            xt = (eacc + aff_r) + c0;  // (rows that are not a full chunk: never used, see carry_scan)
        } else {
            xt = bottomc ? x0v : 0.0;
            e_static_for<0, S>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (i < nsl) {
                    const double di = sD[i * DS + dIdx];
                    xt = Step::fwd_plain(xt, di, m, cf);
                }
            });
        }
        F_STAMP(1);
        const double xin = carry_scan(-1, sPow, (is_x && c < NCH) ? xt : 0.0);
        F_STAMP(2);
        // ================= forward, pass 2: the real sweep (F1) with S1 + D1 + R1 fused in =================
        double pri = 0.0, dua = 0.0;
        if (wv == 0) {  // knot 0, state lanes of chunk 0: x_0 is given (tiny_set_x0), no mat-vec
            const double lo0 = CT ? lo_c : sT[W + r], hi0 = CT ? hi_c : sT[TOFF + W + r];
            const double s = x0v + G0;
            const double snew = fmin(hi0, fmax(lo0, s));
            if (k0) {
                V0p = V0;
                G0 = s - snew;
                pri = fabs(x0v - snew);
                dua = fabs(V0 - snew);
                V0 = snew;
            }
            if constexpr (FAM && !KF) {  // (its lx only reaches p_0, which nothing reads; the duals persist)
                double gcn, gln;
                (void)fam_eval.eval(x0v, GC0, GL0, gcn, gln);
                if (k0) {
                    GC0 = gcn;
                    GL0 = gln;
                }
            }
        }
        double xcur = bottomc ? x0v : xin;
        e_static_for<0, S>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if (q < nsl) {
                const double dq = sD[q * DS + dIdx];
                double loq = lo_c, hiq = hi_c;
                if constexpr (!CT) {
                    loq = sTl[(q + 1) * W];
                    hiq = sTl[TOFF + (q + 1) * W];
                }
                Vp[q] = V[q];
                xcur = Step::fwd_reg(xcur, dq, m, cf, loq, hiq, G[q], V[q], pri, dua);
                if constexpr (KF) {  // this slot's element goes up to its knot's lane
                    if (row_ok) sKX[(q + 1) * ES + kxRow] = xcur;
                } else if constexpr (FAM) {  // xcur: x_{q+1} on state lanes, u_q on input lanes -- this slot's element
                    double gcn, gln;
                    LX[q] = fam_eval.eval(xcur, GC[q], GL[q], gcn, gln);
                    GC[q] = gcn;
                    GL[q] = gln;
                }
            }
        });
        it_done = it0 + 1;  // admm.cpp:143
        // ---- (KF) the families of all slots of this wavefront at once, one knot per lane: rows in, projections, duals, the linear-cost
        // term out through the same entry (read again by both backward passes; entry 0 keeps x_0). Same wavefront on both sides of the
        // buffer: LDS executes a wavefront's accesses in order, the fences keep the compiler from moving them.
        if constexpr (KF) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            double val[NXU], lxo[NXU];
            e_static_for<0, NXU>([&](auto R) { val[R.value] = sKX[kxT + R.value]; });
            kf.eval(val, lxo);
            e_static_for<0, NXU>([&](auto R) { e_lds_write_masked<R.value * 8>(e_lds_addr(sKX + kxT), lxo[R.value], kmask); });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }

        F_STAMP(3);
        // ---- R1 (admm.cpp:93-101): one ballot per wavefront; the flags cross with the backward scan's barrier
        if (check) {
            snap_pri = pri;
            snap_dua = dua;
            res_valid = true;
            const bool below = (pri < p.abs_pri_tol) && (dua * p.rho < p.abs_dua_tol);
            const bool wave_ok = __ballot(!below) == 0ull;
            if (lane == 0) sFlag[(it0 & 1) * WPG + wv] = wave_ok ? 1 : 0;
        }

        // ================= backward (B1, admm.cpp:13-20); linear cost (L1, :77-82) recomputed from V, G =================
        // chain of a chunk of n slots: P <- q~_(n-1) [+ c_in];  for i = n-1 .. 0:  a = [q_(i-1) (i >= 1) + cb | cb] + Mb [P; r_i];
        // d_i = a (input lanes);  P = a (state lanes).  What comes out (state lanes) is p of the chunk's first knot MINUS its q,
        // which the chunk below owns.
        F_STAMP(4);
        double (&mb_)[16] = ops_pick(mB, mloc);
        if constexpr (!OPSR) load_ops(sMb, mb_);
        auto bwd_chain = [&](double cin, auto STORE) -> double {
            constexpr bool store = decltype(STORE)::value;
            double px = 0.0, rcur = 0.0, rnext = 0.0, acc = cb;
            // head of the chain, from the chunk's last slot T: P and r_T, the accumulator start of step T, r_(T-1)
            auto head = [&](auto T, bool terminal) {
                constexpr int tt = decltype(T)::value;
                const double lr1 = lr_of(T);
                double lrT = lr1;
                if (terminal) {  // the last chunk: p_{N-1} (admm.cpp:81-82)
                    double pT = pnref;
                    if constexpr (KF) pT += sKX[(tt + 1) * ES + kxRow];
                    else if constexpr (FAM) pT += LX[tt];
                    lrT = is_x ? pT : lr1;
                }
                px = nrho * (V[tt] - G[tt]) + lrT;
                if constexpr (tt >= 1) {
                    const double lr2 = lr_of(std::integral_constant<int, tt - 1>{});
                    const double t2 = V[tt - 1] - G[tt - 1];
                    acc = rhom * t2 + (is_x ? lr2 + cb : cb);
                    rnext = nrho * t2 + lr2;
                } else {
                    acc = cb;  // (a one-slot chunk: its only step is the chunk's first, which takes no q)
                }
                rcur = px;                  // (input lanes: r_T)
                px = is_x ? px + cin : px;  // (state lanes: + the carry entering from above)
#if defined(TINY_BUILTIN) && !defined(TINY_CHAIN_NOP)
                // (compiled in, bare chain blocks: this select is the one place where the compiler puts a VALU write of a chain's DPP
                // operand right in front of the chain -- the build's lint refused two of the six compiled-in kernels for it, and guarding
                // EVERY chain of a kernel costs a lone wavefront ~7 cycles apiece, 4 % of the iteration; the two wait states, here only)
                asm volatile("s_nop 1" : "+v"(px));
#endif
            };
            auto block = [&](auto Sl) {
                constexpr int s = decltype(Sl)::value;
                constexpr int s2 = s >= 2 ? s - 2 : 0;  // slot feeding the tail
                const double lr2 = lr_of(std::integral_constant<int, s2>{});
                const double lrmc2 = (s >= 2) ? (is_x ? lr2 + cb : cb) : cb;
                const double rh = (s >= 2) ? rhom : 0.0;
                double a = acc, an, rn;
                Step::bwd(a, px, rcur, mb_, V[s2], G[s2], rh, lrmc2, nrho, lr2, an, rn);
                if constexpr (store) {
                    if (is_u) sD[s * DS + dIdx] = a;  // d_s
                    if constexpr (TSUM) acc_T(std::integral_constant<int, s>{}, a);
                }
                px = a;
                rcur = rnext;
                rnext = rn;
                acc = an;
            };
            if constexpr (S_LAST == S) {
                if (nsl > 0) head(std::integral_constant<int, S - 1>{}, topc);
            } else {
                if (topc) {
                    head(std::integral_constant<int, S_LAST - 1>{}, true);
                } else if (nsl > 0) {  // the S - S_LAST steps only a full chunk has, then the common part
                    head(std::integral_constant<int, S - 1>{}, false);
                    e_static_for<0, S - S_LAST>([&](auto I) { block(std::integral_constant<int, S - 1 - I.value>{}); });
                }
            }
            double a = 0.0;
            if (nsl > 0) {
                e_static_for<0, S_LAST - 1>([&](auto I) { block(std::integral_constant<int, S_LAST - 1 - I.value>{}); });  // S_LAST-1 .. 1
                a = acc;
                Step::bwd_last(a, px, rcur, mb_);
                if constexpr (store) {
                    if (is_u) sD[dIdx] = a;  // d of the chunk's first slot
                    if constexpr (TSUM) acc_T(std::integral_constant<int, 0>{}, a);
                }
            }
            return a;
        };
        // pass 1: from q~ alone (speculative: runs before the termination verdict, writes nothing)
        const double e2 = bwd_chain(0.0, std::false_type{});
        F_STAMP(5);
        const double pin = carry_scan(+1, sPow + 4 * 256, (is_x && c < NCH) ? e2 : 0.0);
        F_STAMP(6);
        if (check) {  // (behind the scan's barrier)
            int all = 1;
#pragma unroll
            for (int q = 0; q < WPG; ++q) all &= sFlag[(it0 & 1) * WPG + q];
            if (all != 0) {  // uniform over the workgroup: one instance. TINY_SOLVED: the solve returns BEFORE the backward pass and
                status = 1;  // before v <- vnew (admm.cpp:181-197)
                converged = true;
                break;
            }
        }
        // pass 2: the real sweep, from the true p entering the chunk; only d is kept (TSUM: ... and what the next forward scan needs of it)
        if constexpr (TSUM) eacc = 0.0;
        (void)bwd_chain(is_x ? pin : 0.0, std::true_type{});
        F_STAMP(7);
    }

    const unsigned long long t_iter_end = SESSION ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // ---- the four residual norms of the last check: rows, then chunks through LDS
    const double gpx = group_max<W>(is_x ? snap_pri : 0.0), gpu = group_max<W>(is_u ? snap_pri : 0.0);
    const double gdx = group_max<W>(is_x ? snap_dua : 0.0), gdu = group_max<W>(is_u ? snap_dua : 0.0);
    e_barrier();
    if (r < 4) sRes[c * 4 + r] = (r == 0) ? gpx : (r == 1) ? gdx : (r == 2) ? gpu : gdu;
    if constexpr (SESSION) {
        // the tick's first controls, for the early answer below (sMail's command has been consumed: its first words are free)
        if (max_iter > 0) {
            e_static_for<0, S>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if ((i < nsl) && row_ok && !is_x && ((size_t)(s0 + i) + koff) == 0) sMail[r - NX] = V[i];
            });
        }
    }
    e_barrier();
    if constexpr (SESSION) {
        // EARLY ANSWER (SolveParams::host_ans): lines [7 controls | mail_stamp(sequence number, controls)], each written by ONE store
        // instruction of wavefront 0 -- the host accepts a line whose stamp fits the payload it read with it, so nothing has to be
        // fenced or waited for here; the solution's write-out below happens while the host already steps its plant.
        constexpr int NLA = (NU + 6) / 7;
        if (p.host_ans && max_iter > 0 && tid < 8 * NLA) {
            const int line = tid >> 3, slot = tid & 7, idx = line * 7 + slot;  // (whole groups of eight lanes: mail_xor8)
            const double mine = (slot < 7 && idx < NU) ? sMail[idx] : 0.0;
            const unsigned h = mail_xor8(slot < 7 ? mail_term((unsigned long long)__builtin_bit_cast(long long, mine), slot) : 0u);
            host_store(p.host_ans + tid, slot == 7 ? mail_stamp(expect, h) : mine);
        }
    }

    // ---- write-back: solution (device + pinned host); one-shot launches also leave the ADMM state for the next launch
    if (max_iter > 0) {
        e_static_for<0, S>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if ((i < nsl) && row_ok) {
                const size_t kn = (size_t)(s0 + i) + koff;
                if (is_x) {
                    p.sol_x[((size_t)inst * N + kn) * NX + r] = V[i];
                    if (p.host_sol) host_store(&p.host_sol[kn * NX + r], V[i]);
                } else {
                    p.sol_u[((size_t)inst * NS + kn) * NU + (r - NX)] = V[i];
                    if (kn == 0 && p.u0_host) host_store(&p.u0_host[(size_t)inst * NU + (r - NX)], V[i]);  // first controls straight to the host
                    if (p.host_sol) host_store(&p.host_sol[(size_t)N * NX + kn * NU + (r - NX)], V[i]);
                }
            }
        });
        if (k0) {
            p.sol_x[(size_t)inst * N * NX + r] = V0;
            if (p.host_sol) host_store(&p.host_sol[r], V0);
        }
        if constexpr (!SESSION) write_state(converged);
    }
    if (threadIdx.x < 4) {  // lane k: the k-th residual norm over the chunks (one thread walking all four cost a closed-loop tick ~0.1 us)
        const int k = (int)threadIdx.x;
        double res = 0.0;
        for (int q = 0; q < NCH; ++q) res = fmax(res, sRes[q * 4 + k]);
        const double out = (k & 1) ? res * p.rho : res;  // pri_x, dua_x (scaled by rho), pri_u, dua_u
        if (k == 0) {
            p.istats[inst * 2 + 0] = it_done;
            p.istats[inst * 2 + 1] = status;
        }
        if (res_valid) p.dstats[inst * 4 + k] = out;
        if (p.host_sol) {
            double *hs = p.host_sol + (size_t)N * NX + (size_t)NS * NU;
            if (k == 0) {
                host_store(&hs[4], (double)it_done);
                host_store(&hs[5], (double)status);
            }
            if (res_valid) host_store(&hs[k], out);
        }
    }
#if TINY_F_STAMP
    e_barrier();
    if (lane < 8) {
        p.sol_x[(size_t)inst * N * NX + wv * 8 + lane] = (double)(stamp[lane] - stamp[0]);
        if (p.host_sol) host_store(&p.host_sol[wv * 8 + lane], (double)(stamp[lane] - stamp[0]));
    }
#endif
    if (p.host_sol && (SESSION || p.host_seq != 0.0)) {  // (uniform) everything above is in pinned memory: raise the completion stamp
        // Everything the host reads after the stamp was stored with host_store() (system scope: written through, nothing of it stays dirty in
        // the L2): once every thread's stores have been handed over (s_waitcnt vmcnt(0) -- a workgroup-scope release) the stamp may follow
        // them. Rounds 2-4 had plain stores and a system-scope fence here (__threadfence_system + a release store), which wrote back the
        // L2's dirty lines -- the solution just stored to HBM included -- and INVALIDATED the caches, twice per tick: the next tick then
        // fetched its operators and tables from HBM again.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        // (a system-scope atomic store: written through at once -- a plain store might be combined or delayed)
        if (threadIdx.x == 0) {
            double *const hs = p.host_sol + (size_t)N * NX + (size_t)NS * NU;
            if constexpr (SESSION) {
                // diagnostics in the spare slot behind the stamp (tinympc_debug_tick_timing): how long the kernel waited for this command
                // since its last answer, its ADMM iterations, its write-out -- 16 bits each, in ticks of 10 ns
                const unsigned long long t_out = __builtin_amdgcn_s_memrealtime();
                auto clip = [](unsigned long long d) -> unsigned long long { return d > 65535ull ? 65535ull : d; };
                const unsigned long long packed = clip(t_prev_out ? t_seen - t_prev_out : 0ull) | (clip(t_iter_end - t_seen) << 16) | (clip(t_out - t_iter_end) << 32);
                host_store(hs + 7, (double)packed);
                t_prev_out = t_out;
            }
            __hip_atomic_store(hs + 6, SESSION ? expect : p.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if constexpr (!SESSION) break;
    // The next tick warm-starts from the registers. A converged solve returns before v <- vnew (admm.cpp:181-197): its canonical
    // slack is the previous iterate -- what an ordinary launch would have written back and read again.
    if (converged) {
        e_static_for<0, S>([&](auto I) { V[I.value] = Vp[I.value]; });
        V0 = V0p;
    }
    expect += 1.0;
    }  // ticks
}

}  // namespace tinympc

// (two wavefronts per SIMD wherever the workgroup has more than four)
// the entry point: `tinympc_jit_solve` as a run-time specialisation (tinympc_jit.hip looks it up by that name), the name the build
// gives it as a compiled-in one (TINY_BUILTIN: __graft_entry__.HIP_BUILTINS)
#ifdef TINY_BUILTIN
#define TINY_KERNEL_NAME TINY_BUILTIN_NAME
#else
#define TINY_KERNEL_NAME tinympc_jit_solve
#endif
extern "C" __global__ void __launch_bounds__(64 * TINY_JIT_F_WPG) __attribute__((amdgpu_waves_per_eu(1, (TINY_JIT_F_WPG + 3) / 4 > 2 ? (TINY_JIT_F_WPG + 3) / 4 : 2)))
TINY_KERNEL_NAME(const tinympc::SolveParams p) {
    constexpr bool CTJ = TINY_JIT_CT != 0, FAMJ = TINY_JIT_FAM != 0;
    constexpr bool SESJ = TINY_JIT_F_SESSION != 0, KFJ = FAMJ && TINY_JIT_F_KFAM != 0;
    constexpr size_t bytes = tinympc::f_lds_bytes(TINY_JIT_NU, TINY_JIT_N, CTJ, TINY_JIT_F_WPG, TINY_JIT_F_S, FAMJ, tinympc::E_NL,
                                                  KFJ ? TINY_JIT_NX + TINY_JIT_NU : 0, tinympc::E_NCONE);
    static_assert(bytes <= 160 * 1024, "layout F: the workgroup's LDS plan exceeds a CU");
    __shared__ __attribute__((aligned(16))) double smem_f[bytes / sizeof(double)];
    tinympc::k_admm_solve_f_body<TINY_JIT_NX, TINY_JIT_NU, TINY_JIT_N, CTJ, TINY_JIT_F_WPG, TINY_JIT_F_S, FAMJ, SESJ, KFJ>(p, smem_f);
}
