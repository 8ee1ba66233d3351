// tinympc_solve_d.hip -- k_admm_solve_d: the solve kernel in "layout D" (register-resident throughput layout), gfx950 FP64.
//
// Same algorithm and lane layout as the other solve kernels (tinympc_solve.hip has the reference citations:
// M1 solve admm.cpp:109-207 = F1 :25-35, S1 :43-59, D1 :65-69, L1 :75-83, R1 :89-107, C1 :196-197, B1 :13-20).
// What changes is where the ADMM state lives and, with it, how many wavefronts a SIMD runs:
//
//   layout B   G + D in LDS (33 KB per wave), V in HBM/L2        -> LDS caps a CU at 4 waves = ONE per SIMD; a lone wave
//              issues one VALU instruction per ~8 cycles on the dependent FP64 chain; 104 of 512 VGPRs in use.
//   layout D   the horizon is a COMPILE-TIME constant and both sweeps are fully unrolled, so every knot's dual g|y
//              has its own register pair (2*(N-1)+2 VGPRs) and the slack v|z is split between registers and LDS; LDS
//              holds only the feed-forward d (compact), part of v|z and one copy of the sweep operators per workgroup.
//              <= 256 VGPRs and <= 20 KB of LDS per wave -> 8 waves per CU = TWO per SIMD, whose FP64 chains
//              interleave. HBM is touched at entry and exit only (plus the conditional stale copy, below).
//              Horizons whose duals outgrow 256 registers run ONE wavefront per SIMD with all 512 (run-time specialised,
//              tinympc_jit.hip); a workgroup is four wavefronts where the LDS plan allows it, so that mid-size batches
//              spread over the CUs wavefront by wavefront (d_wpg below).
//
// Per sweep step the instruction stream is one asm block (tinympc_solve_d_chain.h): mov + (nx+nu) fused DPP FMAs +
// the 8-instruction row-local block going forward, (nx+nu) FMAs + 3 going backward; no address arithmetic (LDS
// immediate offsets), no selects:
//   * the chain's columns k < nx read the state operand register, columns k >= nx the input-row operand register
//     (two different DPP sources), so [x_i; d_i] / [p_{i+1}; r_i] are never merged into one register;
//   * going backward every lane uses the SAME slot: slot s holds knot s+1 on state lanes and knot s on input lanes,
//     q_s is folded into the accumulator's start value (state lanes) instead of being added after the mat-vec;
//   * a converged instance keeps iterating as a zombie (no EXEC-masked region around the unrolled body); its state was
//     written back before the sweeps touch it again (see the control comment in the body).
// Run-time specialised variants (C++ between the asm blocks): FAM -- the cone / linear-inequality families --, and ADAPT --
// adaptive rho with rho per lane; see k_admm_solve_d_body.
// The sweep operators' rows (16 doubles per lane each) are re-read from LDS at the start of every sweep: holding both
// for the whole solve would cost 32 more VGPRs than the budget has.
//
// Reference semantics that need care are those of layout B (tinympc_solve_b.hip): per-instance termination,
// `iter % check_termination` with iter already incremented, the solution is vnew/znew, and a converged solve leaves
// the PREVIOUS iterate in v/z (admm.cpp:181-197) -- the stale copy goes to p.V2, written only in sweeps that can
// still end converged (decided on knot 0, after D_FIRST steps and then every D_GROUP steps; exact because the
// residual maxima only grow).
#ifndef TINY_JIT
#include <atomic>  // (host side only: the run-time compiler has no use for it)
#endif
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

// Slot refill is a TEXTUAL variant of this file (TINY_REFILL: tinympc_solve_dr.hip includes it; -DTINY_JIT_REFILL=1 for the
// run-time specialisations), not a template parameter of the plain kernels: as one, the non-refill instantiation kept its
// semantics but not its code, and lost 2.6 % (profiles/r03_dgroup_ab.txt). What it had lost was found afterwards -- its sweep
// chains had moved off the 8-byte grid by one 4-byte instruction, see D_AL in tinympc_solve_d_chain.h, which now puts every
// chain on the grid by construction; the variant stays textual all the same: the plain translation unit preprocesses to
// exactly the text it had before the variant existed.
#ifndef TINY_REFILL
#if defined(TINY_JIT_REFILL) && TINY_JIT_REFILL
#define TINY_REFILL 1
#else
#define TINY_REFILL 0
#endif
#endif

#define TINY_STR2(x) #x
#define TINY_STR(x) TINY_STR2(x)
namespace tinympc {
template <int NX, int NU>
struct DStep;  // specialised per (nx, nu) by tinympc_solve_d_chain.h
}  // namespace tinympc

// The (nx, nu) pairs compiled into the library: quadrotor, cartpole, rocket landing (BASELINE.json configs 2-5).
#ifdef TINY_JIT  // run-time specialisation (tinympc_jit.hip): exactly one (nx, nu, N), from -D options
#define D_NX TINY_JIT_NX
#define D_NU TINY_JIT_NU
#include "tinympc_solve_d_chain.h"
#else
#define D_NX 12
#define D_NU 4
#include "tinympc_solve_d_chain.h"
#define D_NX 4
#define D_NU 1
#include "tinympc_solve_d_chain.h"
#endif

namespace tinympc {

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// ---- LDS plan per workgroup, in doubles: operators [2][16 k][16 r] | tables (!CT) | per wave: V[VL][64], D[(N-1)*4*nu]
constexpr int D_OPS_DOUBLES = 2 * 16 * 16;
#ifdef TINY_D_GROUP
constexpr int D_GROUP = TINY_D_GROUP;  // (experiments)
constexpr int D_FIRST = TINY_D_FIRST;
#else
// Forward steps between two "can this sweep still converge" tests: one test on knot 0 alone, the next after D_FIRST
// steps (a sweep that cannot converge has then copied D_FIRST slots of stale iterate at most), then every D_GROUP steps.
// Each test is a scheduling boundary of the unrolled sweep; 4 | 23 measured best of the splits tried on the headline
// (profiles/r03_dgroup_ab.txt), forced iterations and converging batch alike.
#if TINY_REFILL
// (the slot-refill variant runs converging batches: finer groups cut a sweep's stale copies sooner -- 4 | 12 measured best of
// eight splits on the 65,536-instance converging batch, 10.7 against 10.85 ms with 4 | 23; profiles/r03_refill_probe.txt)
constexpr int D_GROUP = 12;
constexpr int D_FIRST = 4;
#else
constexpr int D_GROUP = 23;
constexpr int D_FIRST = 4;
#endif
#endif
#ifdef TINY_JIT_VREG
constexpr int D_VREG_MAX = TINY_JIT_VREG;  // chosen by the host from its register estimate
#else
constexpr int D_VREG_MAX = 24;
#endif    // slack knots kept in registers (the rest goes to LDS)
constexpr int D_LDS_PER_CU = 160 * 1024;
__host__ __device__ constexpr int d_d_doubles(int nu, int N) { return ((N - 1) * 4 * nu + 1) & ~1; }
__host__ __device__ constexpr int d_tab_doubles(int N) { return 3 * (N + 2) * 16 + 16; }
constexpr int D_FAM_DOUBLES = 3 * MAX_LIN_ROWS * 16;  // FAM: a_k | b_k | 1/||a_k||^2 of the linear rows, per lane row
constexpr int D_ADAPT_DOUBLES = 5 * 16 * 16;         // ADAPT: dMf | dMb | [A'; B'] | Pinf | dPinf, transposed like the operators
// number of slack slots in LDS; -1 if the shape does not fit the plan (cu_waves wavefronts per CU: 8, or 4 for the
// long-horizon plan with one wavefront per SIMD and 512 registers)
__host__ __device__ constexpr int d_vl(int nu, int N, bool ct, int wpg, int cu_waves = 8, bool fam = false, bool adapt = false) {
    const int ns = N - 1;
    const int wg_doubles = D_LDS_PER_CU / 8 * wpg / cu_waves - D_OPS_DOUBLES - (ct ? 0 : d_tab_doubles(N)) - (fam ? D_FAM_DOUBLES : 0) - (adapt ? D_ADAPT_DOUBLES : 0);
    const int wave_doubles = wg_doubles / wpg - d_d_doubles(nu, N);
    if (wave_doubles < 0) return -1;
    const int vlmax = wave_doubles / 64;
    const int want = ns > D_VREG_MAX ? ns - D_VREG_MAX : 0;
    return want <= vlmax ? want : -1;
}
__host__ __device__ constexpr size_t d_lds_bytes(int nu, int N, bool ct, int wpg, int vl, bool fam = false, bool adapt = false) {
    return sizeof(double) * ((size_t)D_OPS_DOUBLES + (ct ? 0 : d_tab_doubles(N)) + (fam ? D_FAM_DOUBLES : 0) + (adapt ? D_ADAPT_DOUBLES : 0) +
                             (size_t)wpg * (vl * 64 + d_d_doubles(nu, N)));
}

// LDS traffic of the sweeps: WHERE a read is issued is this file's decision, not the scheduler's (see tinympc_solve_d_chain.h) -- reads
// are issued one block ahead and retired by the wait behind the block; both in a form the compiler tracks (tinympc_sweep.h:
// lds_read_issued_here / lds_reads_landed). Writes carry no result and stay plain asm.
__device__ __forceinline__ unsigned lds_addr(const double *p) { return lds_address(p); }
template <int OFF>
__device__ __forceinline__ double lds_read_async(unsigned addr) {  // the value is valid after the next lds_reads_landed()
    return lds_read_issued_here<OFF>(addr);
}
template <int OFF>
__device__ __forceinline__ void lds_write_async(unsigned addr, double v) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// Store for the lanes of `mask` only, without a branch: EXEC is narrowed around the one instruction by two scalar
// instructions (a branch around the store would put a label between two sweep blocks -- one more taken-or-not decision per
// step, and a branch target the ISA lint could not see through).
template <int OFF>
__device__ __forceinline__ void lds_write_masked(unsigned addr, double v, unsigned long long mask) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "ds_write_b64 %[a], %[v] offset:%[o]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved)
                 : [m] "s"(mask), [a] "v"(addr), [v] "v"(v), [o] "n"(OFF)
                 : "memory", "scc");
}
__device__ __forceinline__ void lds_wait() {
    lds_reads_landed();
    asm volatile("" ::: "memory");
}
// ... and the point where the compiler retires ITS OWN reads of the sweep's operator rows: left alone it waits for them in
// front of the first block that uses them, i.e. right behind the asynchronous reads issued for the second block -- a full
// LDS latency in every sweep. (The empty asm makes the rows operands of THIS point.)
__device__ __forceinline__ void lds_wait_ops(double (&m)[16]) {
    lds_reads_landed();
#ifdef TINY_D_NO_OPS_WAIT
    asm volatile("" ::: "memory");
#else
    asm volatile(""
                 : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]),
                   "+v"(m[10]), "+v"(m[11]), "+v"(m[12]), "+v"(m[13]), "+v"(m[14]), "+v"(m[15])
                 :
                 : "memory");
#endif
}

// `bad` = ballot of lanes whose row already rules out convergence in this sweep, `live` = ballot of the lanes that
// are still iterating. True if some live instance (16-lane row) has no bad lane.
// (no short-circuit evaluation: four scalar compares instead of a chain of branches in every sweep)
__device__ __forceinline__ bool wave_may_converge_d(unsigned long long bad, unsigned long long live) {
    int any = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned b = (unsigned)(bad >> (j * 16)) & 0xffffu, l = (unsigned)(live >> (j * 16)) & 0xffffu;
        any |= (int)(l != 0u) & (int)(b == 0u);
    }
    return any != 0;
}
// Branches that are almost never taken (write-back, stale copies): the compiler moves their bodies out of the sweep's
// straight line. A TAKEN branch costs a wavefront ~110 cycles of instruction fetch (tools/microbench_fp64_step.hip, loops
// against straight-line code), so the common path should be the fall-through.
#ifdef TINY_D_NO_RARE
#define TINY_RARE(c) (c)
#else
#define TINY_RARE(c) __builtin_expect(!!(c), 0)
#endif

// FAM: the second-order-cone and linear-inequality slack families of k_admm_solve_fam (PARITY UNPINNED, see there) ride on
// the forward step exactly as in the latency kernel (tinympc_solve_c.hip): every knot carries two more duals gc|yc, gl|yl
// (persistent: the HBM arrays GC / GL) and the families' linear-cost term lx (forward -> backward), all in registers.
// Run-time specialised only (tinympc_jit.hip, -DTINY_JIT_FAM=1): 10 VGPRs per knot instead of 4, so short horizons.
// ADAPT: adaptive rho (admm.cpp:117-174, rho_benchmark.cpp) exactly as in k_admm_solve_adapt -- rho is PER INSTANCE (a lane
// variable, persistent in p.rho_inst), the lane's operator rows are (base row) + (rho - rho0) * (derivative row) rebuilt from
// LDS at every sweep, and every fifth iteration the four norms of the reference's dense KKT system ride on the forward sweep
// (one extra mat-vec per step). Run-time specialised only (-DTINY_JIT_ADAPT=1), not together with the families.
// HOSTX: the batched zero-copy tick (x0 read from pinned host memory and mirrored, first controls written to pinned host memory). A variant
// of its own: as run-time branches in the rare paths of the one kernel the two stores cost the sweeps 3.6 % (1.72 -> 1.78 ms: the
// pointers' scalar registers, live across the unrolled iteration loop).
#if TINY_REFILL
// REFILL: slot refill (SolveParams::refill_next). The launch is one resident set of wavefronts; every 16-lane row counts its OWN
// iterations, and a row whose instance has finished (converged, or max_iter) is written back -- state, solution, statistics -- and
// loaded with the next instance of the batch, cold or warm, while the other three rows of the wavefront keep iterating. The
// arithmetic of an instance is that of the plain kernel (bit-identical results: tests/test_hip_parity.py); what changes is that
// a wavefront's time is the sum of what its rows worked, not four times its slowest instance.
template <int NX, int NU, int N, bool CT, int WPG, int VL, bool FAM = false, bool ADAPT = false, bool TWO_PER_SIMD = true, bool HOSTX = false,
          bool REFILL = false>
#else
template <int NX, int NU, int N, bool CT, int WPG, int VL, bool FAM = false, bool ADAPT = false, bool TWO_PER_SIMD = true, bool HOSTX = false>
#endif
__device__ __forceinline__ void k_admm_solve_d_body(const SolveParams &p, double *smem) {
    static_assert(!(FAM && ADAPT), "adaptive rho and the constraint families exclude each other (as in the C ABI)");
#if TINY_REFILL
    static_assert(!REFILL || (!FAM && !ADAPT && !HOSTX), "slot refill: box-constrained path only");
#endif
    // this wavefront's slot on its SIMD (HW_REG_HW_ID bits 3:0): the two wavefronts of a SIMD sit in different slots
    const int simd_slot = TWO_PER_SIMD ? simd_slot_id() : 0;
    constexpr int W = 16, IPW = 4, NXU = NX + NU, NS = N - 1, DS = IPW * NU, NVR = NS - VL;
    constexpr int KT = NXU <= 8 ? 8 : NXU <= 12 ? 12 : 16;  // row stride of p.ops (choose_geometry)
    constexpr int TOFF = (N + 2) * W;
    static_assert(NS >= 3 && VL >= 0 && VL <= NS, "layout D: N >= 4");
    using Step = DStep<NX, NU>;

#ifdef TINY_CLOCK_STAMP
    const unsigned long long ck_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane >> 4, r = lane & 15;
    const long grp = (long)blockIdx.x * WPG + wv;
    const bool grp_ok = grp < p.groups;
    const long inst = grp * IPW + j;
    const bool is_x = r < NX;
    const bool is_u = (r >= NX) && (r < NXU);
    const bool inst_ok = grp_ok && inst < p.batch;
    const int koff = is_x ? 1 : 0;  // slot s = knot s+1 on state lanes, knot s on input lanes

    double *sOps = smem;
    double *sT = smem + D_OPS_DOUBLES;
    double *sLin = sT + (CT ? 0 : d_tab_doubles(N));  // FAM
    double *sAd = sLin + (FAM ? D_FAM_DOUBLES : 0);   // ADAPT: [5][16 k][16 r]
    double *sV = sAd + (ADAPT ? D_ADAPT_DOUBLES : 0) + (size_t)wv * (VL * 64 + d_d_doubles(NU, N));
    double *sD = sV + VL * 64;

    // ---- workgroup-shared: the two sweep operators, transposed to [k][r] (conflict-free row reads), and the tables
    for (int i = threadIdx.x; i < D_OPS_DOUBLES; i += 64 * WPG) {
        const int which = i >> 8, k = (i >> 4) & 15, rr = i & 15;
        sOps[i] = (k < KT) ? p.ops[(size_t)which * W * KT + (size_t)rr * KT + k] : 0.0;
    }
    if constexpr (!CT)
        for (int i = threadIdx.x; i < d_tab_doubles(N); i += 64 * WPG) sT[i] = p.tables[i];
    if constexpr (ADAPT) {  // k_build_adapt's tables: mt | pinf | dpinf | dmf | dmb, each [W][KT]; here: dmf | dmb | mt | pinf | dpinf
        for (int i = threadIdx.x; i < D_ADAPT_DOUBLES; i += 64 * WPG) {
            const int which = i >> 8, k = (i >> 4) & 15, rr = i & 15;
            const int tbl = which == 0 ? 3 : which == 1 ? 4 : which - 2;
            sAd[i] = (k < KT) ? p.adapt[(size_t)tbl * W * KT + (size_t)rr * KT + k] : 0.0;
        }
    }
    if constexpr (FAM) {  // layout of fam_doubles(): ... | nl | per linear row k: a_k[W] b_k[W] ||a_k||^2[W]
        const double *lin_rows = p.fam + 4 * W + (size_t)3 * W * KT;
        for (int i = threadIdx.x; i < D_FAM_DOUBLES; i += 64 * WPG) {
            const int k3 = i / W, rr = i % W;
            const double v = lin_rows[1 + (size_t)k3 * W + rr];
            sLin[i] = (k3 % 3 == 2) ? 1.0 / v : v;  // 1 / ||a_k||^2
        }
    }

    const size_t g0 = grp_ok ? (size_t)grp : 0;
    double *const gG = p.G + g0 * (N + 1) * 64 + lane;                 // row kn = knot kn
    double *const gD = p.D + g0 * (size_t)(NS * DS);
    double *const gV0 = p.V + (g0 * v_rows(N) + V_PAD) * 64 + lane;    // canonical v|z, knot 0
#if TINY_REFILL
    // stale copy, knot 0 (wave-uniform: scalar base + 32-bit lane offset; REFILL: the rows of a wavefront belong to different
    // groups of four, so the base is the array's and the offset carries the group -- below 2^32 doubles for any batch that fits HBM)
    double *const gV1u = p.V2 + ((REFILL ? 0 : g0) * v_rows(N) + V_PAD) * 64;
    int cur = (int)inst;  // REFILL: the instance this row works on
    auto row_offset = [&](int i) -> unsigned { return (unsigned)(i >> 2) * (unsigned)(v_rows(N) * 64) + (unsigned)((i & 3) * 16 + r); };
    // (REFILL: rebuilt from `cur` where it is needed -- rare paths -- instead of living in a register across the sweeps)
#else
    double *const gV1u = p.V2 + (g0 * v_rows(N) + V_PAD) * 64;         // stale copy, knot 0 (wave-uniform: scalar base + 32-bit lane offset)
#endif
    const unsigned voff = (unsigned)(lane + koff * 64);
    double *const sVl = sV + lane;
    const bool cold = p.cold != 0;  // (uniform) the state is zero by contract and was never written to HBM: nothing to load
    if (grp_ok) {
        if (cold) {
            for (int i = lane; i < NS * DS; i += 64) sD[i] = 0.0;
            static_for<0, VL>([&](auto S) { sVl[S.value * 64] = 0.0; });
        } else {
            for (int i = lane; i < NS * DS; i += 64) sD[i] = gD[i];
            static_for<0, VL>([&](auto S) { sVl[S.value * 64] = gV0[(S.value + koff) * 64]; });
        }
    }
    __syncthreads();  // the only workgroup-wide barrier: from here on the waves are independent
    if (!grp_ok) return;
#ifdef TINY_CLOCK_STAMP
    const unsigned long long ck_barrier = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- register-resident state
    double G[NS], G0, Vr[NVR > 0 ? NVR : 1], V0;
    if (cold) {
        static_for<0, NS>([&](auto S) { G[S.value] = 0.0; });
        static_for<0, NVR>([&](auto S) { Vr[S.value] = 0.0; });
        G0 = 0.0;
        V0 = 0.0;
    } else {
        static_for<0, NS>([&](auto S) { G[S.value] = gG[(S.value + koff) * 64]; });
        static_for<0, NVR>([&](auto S) { Vr[S.value] = gV0[(VL + S.value + koff) * 64]; });
        G0 = gG[0];
        V0 = gV0[0];
    }
    // FAM state: slot s <-> knot s + koff, like G / V (the arrays GC / GL have V's shape)
    double GCr[FAM ? NS : 1], GLr[FAM ? NS : 1], LX[FAM ? NS : 1], GC0 = 0.0, GL0 = 0.0;
    double cn[KT], ct_[KT], ty[KT];
    int role = 0, nl = 0;
    double mu = 0.0, inv_mu = 0.0;
    bool famc = false, faml = false, any_cone = false, any_lin = false;
    if constexpr (FAM) {
        const double *const gGC = p.GC + (g0 * v_rows(N) + V_PAD) * 64 + lane, *const gGL = p.GL + (g0 * v_rows(N) + V_PAD) * 64 + lane;
        static_for<0, NS>([&](auto S) {
            GCr[S.value] = gGC[(S.value + koff) * 64];
            GLr[S.value] = gGL[(S.value + koff) * 64];
            LX[S.value] = 0.0;
        });
        GC0 = gGC[0];
        GL0 = gGL[0];
        const double *Cn = p.fam + 4 * W + (size_t)r * KT, *Ct = Cn + (size_t)W * KT, *Ty = Ct + (size_t)W * KT;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            cn[k] = Cn[k];
            ct_[k] = Ct[k];
            ty[k] = Ty[k];
        }
        role = (int)p.fam[r];
        mu = p.fam[W + r];
        inv_mu = (mu != 0.0) ? 1.0 / mu : 0.0;  // (mu = 0: row in no cone)
        famc = p.fam[2 * W + r] != 0.0;
        faml = p.fam[3 * W + r] != 0.0;
        nl = (int)p.fam[4 * W + (size_t)3 * W * KT];
        any_cone = __ballot(famc) != 0ull;  // the same in every 16-lane row
        any_lin = __ballot(faml) != 0ull;
    }
    // One (row, knot) element of the two extra families, exactly as in k_admm_solve_fam / k_admm_solve_c: returns the element's
    // contribution to the linear cost and the new duals. All 16 lanes of the instance take part (no EXEC masking in this kernel).
    auto families = [&](double val, double gc_old, double gl_old, double &gc_new, double &gl_new) -> double {
        double lxv = 0.0;
        gc_new = gc_old;
        gl_new = gl_old;
        if (any_cone && any_lin && nl == 1) {
            // Both families and a single linear row (the rocket-landing case): the same operations as below, in ONE basic block, so
            // that the scheduler interleaves the three independent mat-vec chains and the two projection sequences -- a lone
            // wavefront issues a dependent FP64 instruction every ~7 cycles, an independent one every ~5.
            const double svc = val + gc_old, s0 = val + gl_old;
            const double a_k = sLin[r], b_k = sLin[W + r], in_k = sLin[2 * W + r];
            const double a2 = group_matvec<W, KT>(cn, svc * svc, 0.0);
            const double t = group_matvec<W, KT>(ct_, svc, 0.0);
            const double dot = group_matvec<W, KT>(ty, a_k * s0, 0.0);
            const double vc = soc_project_element(svc, a2, t, mu, inv_mu, role);
            const double svl = halfspace_project_element(s0, dot, a_k, b_k, in_k);
            const double gcn = svc - vc, gln = s0 - svl;
            if (famc) {
                gc_new = gcn;
                lxv -= p.rho * (vc - gcn);
            }
            if (faml) {
                gl_new = gln;
                lxv -= p.rho * (svl - gln);
            }
            return lxv;
        }
        if (any_cone) {
            const double sv = val + gc_old;
            const double a2 = group_matvec<W, KT>(cn, sv * sv, 0.0);
            const double t = group_matvec<W, KT>(ct_, sv, 0.0);
            const double vc = soc_project_element(sv, a2, t, mu, inv_mu, role);
            const double gcn = sv - vc;
            if (famc) {
                gc_new = gcn;
                lxv -= p.rho * (vc - gcn);
            }
        }
        if (any_lin) {
            const double s0 = val + gl_old;
            double sv = s0;
#pragma unroll 1
            for (int k = 0; k < nl; ++k) {  // (uniform trip count)
                const double a_k = sLin[(3 * k + 0) * W + r], b_k = sLin[(3 * k + 1) * W + r], in_k = sLin[(3 * k + 2) * W + r];
                const double dot = group_matvec<W, KT>(ty, a_k * sv, 0.0);
                sv = halfspace_project_element(sv, dot, a_k, b_k, in_k);
            }
            const double gln = s0 - sv;
            if (faml) {
                gl_new = gln;
                lxv -= p.rho * (sv - gln);
            }
        }
        return lxv;
    };
    const bool row_ok = r < NXU;

    const double cf = p.ops[(size_t)2 * W * KT + r];
    const double cb = p.ops[(size_t)2 * W * KT + W + r];
    const double pnref0 = p.tables[(size_t)3 * TOFF + r];
    // (ADAPT: rho, its negative and pNref are lane variables that change every fifth iteration; otherwise they never change and
    // stay in scalar registers)
    const double rho0 = p.rho;
    double rho = p.rho, pnref = pnref0;
    double dpnref = 0.0, dgr = 0.0;
    if constexpr (ADAPT) {
        rho = inst_ok ? p.rho_inst[inst] : rho0;  // persists across solves like cache->rho
        dpnref = p.adapt[(size_t)5 * W * KT + r];
        dgr = p.ops[(size_t)2 * W * KT + 2 * W + r];  // Q + rho0 / R + rho0 diagonal of this row (tiny_api.cpp:90-91)
        pnref = fma(rho - rho0, dpnref, pnref0);
    }
    double nrho = -rho;
    const double lo_c = p.tables[W + r], hi_c = p.tables[(size_t)TOFF + W + r], lr_c = p.tables[(size_t)2 * TOFF + W + r];
    double rhom = is_x ? nrho : 0.0;
#if TINY_REFILL
    double x0v = (inst_ok && is_x) ? p.x0[inst * NX + r] : 0.0;
#else
    const double x0v = (inst_ok && is_x) ? p.x0[inst * NX + r] : 0.0;
#endif
    if constexpr (HOSTX) {  // zero-copy tick: x0 came from pinned host memory
        if (p.x0_mirror && inst_ok && is_x) p.x0_mirror[inst * NX + r] = x0v;
    }
    const int dIdx = j * NU + (is_u ? r - NX : 0);
    const double *const sDr = sD + dIdx;
    double *const sDw = sD + dIdx;
    const double *const sTl = sT + koff * W + r;  // (!CT) row of slot s: sTl[(s + 1) * W]
    const double *const sMf = sOps + r, *const sMb = sOps + 256 + r;
    const unsigned aV = lds_addr(sVl), aD = lds_addr(sDr), aT = lds_addr(sTl);
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
    double snap_pri = 0.0, snap_dua = 0.0, snap_rho = p.rho;

    auto load_ops = [&](const double *src, double (&m)[16]) {  // (src: sMf / sMb; ADAPT: the derivative rows sit 512 doubles behind)
        if constexpr (ADAPT) {
            const double delta = rho - rho0;
            static_for<0, 16>([&](auto K) {
                constexpr int o = (K.value < NXU ? K.value : 0) * 16;
                m[K.value] = fma(delta, src[o + (sAd - sOps)], src[o]);
            });
        } else {
            static_for<0, 16>([&](auto K) { m[K.value] = src[(K.value < NXU ? K.value : 0) * 16]; });
        }
    };
    auto vget = [&](auto S) -> double {
        if constexpr (decltype(S)::value >= VL) return Vr[decltype(S)::value - VL];
        else return sVl[decltype(S)::value * 64];
    };

#ifdef TINY_CLOCK_STAMP  // diagnostic build only (tools/clock_check.py): the shader clock held under this kernel
    const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef TINY_D_PAD  // (experiment: TINY_D_PAD s_nop in front of the iteration loop -- shifts the loop's code by 4 bytes each)
    asm volatile(".rept " TINY_STR(TINY_D_PAD) "\n\ts_nop 0\n\t.endr" ::: "memory");
#endif
#ifdef TINY_D_LOOP_ALIGN  // (experiment: the iteration loop's code starts on a 2^TINY_D_LOOP_ALIGN byte boundary)
    asm volatile(".p2align " TINY_STR(TINY_D_LOOP_ALIGN) ::: "memory");
#endif
    const int max_iter = p.max_iter;
    for (int it = 0; max_iter > 0; ++it) {  // admm.cpp:129
        // (readfirstlane: keeps the loop counter and everything derived from it in SGPRs, so that the branches below
        // are scalar branches and not EXEC-masked regions)
        const int it0 = __builtin_amdgcn_readfirstlane(it);
#if TINY_REFILL
        const bool final_round = REFILL ? false : it0 >= max_iter;  // (REFILL: every row has its own count, `fin` below)
#else
        const bool final_round = it0 >= max_iter;
#endif
        if constexpr (TWO_PER_SIMD) fair_share_priority<NS>(it0, simd_slot);  // (tinympc_sweep.h: the two wavefronts of a SIMD finish together)
        // ---- write-back: G, D and the canonical v|z (not converged: v = vnew, admm.cpp:196-197; converged: the solve
        // returned before v <- vnew, so the canonical copy is the stale one in V2); solution = vnew / znew (:187-188, 204-205)
#if TINY_REFILL
        const bool fin = REFILL ? it_done >= max_iter : final_round;
        const bool wb = pending || (fin && active);
#else
        const bool wb = pending || (final_round && active);
#endif
        if (TINY_RARE(__ballot(wb) != 0ull)) {
            // Rare path (once per instance and solve), kept small in registers rather than fast: addresses are rebuilt
            // here from the kernel arguments (the opaque copy of `lane` keeps the compiler from hoisting them out of
            // the iteration loop, where they would occupy registers the unrolled sweeps need).
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int r_o = lane_o & 15, j_o = lane_o >> 4;
            const bool x_o = r_o < NX;
#if TINY_REFILL
            // (REFILL: this row's instance; its place in the group-of-four layout of the state arrays)
            int cur_o = cur;
            if constexpr (REFILL) asm volatile("" : "+v"(cur_o));
            const size_t grp_w = REFILL ? (size_t)(cur_o >> 2) : (size_t)grp;
            const int lane_w = REFILL ? (cur_o & 3) * 16 + r_o : lane_o;
#endif
            if (wb && r_o < NXU) {
                const int ko = x_o ? 1 : 0;
#if TINY_REFILL
                const size_t inst_o = REFILL ? (size_t)cur_o : (size_t)grp * IPW + j_o;
                double *const wG = p.G + grp_w * (N + 1) * 64 + lane_w + ko * 64;                // slot 0
                double *const wV = p.V + (grp_w * v_rows(N) + V_PAD) * 64 + lane_w + ko * 64;   // slot 0
#else
                const size_t inst_o = (size_t)grp * IPW + j_o;
                double *const wG = p.G + (size_t)grp * (N + 1) * 64 + lane_o + ko * 64;                // slot 0
                double *const wV = p.V + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane_o + ko * 64;   // slot 0
#endif
                double *const wS = x_o ? p.sol_x + (inst_o * N + 1) * NX + r_o : p.sol_u + inst_o * NS * NU + (r_o - NX);  // slot 0
                const int sst = x_o ? NX : NU;
                if (x_o) {  // knot 0
                    wG[-64] = G0;
                    wV[-64] = V0;
                    wS[-NX] = V0;
                }
                static_for<0, NS>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    const double vn = vget(S);
                    wG[s * 64] = G[s];
                    wV[s * 64] = vn;
                    wS[s * sst] = vn;
                    if constexpr (HOSTX && s == 0) {  // zero-copy tick: the first controls also go straight into pinned host memory
                        if (p.u0_host && !x_o) p.u0_host[inst_o * NU + (r_o - NX)] = vn;
                    }
                });
                if constexpr (FAM) {
                    double *const wGC = p.GC + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane_o + ko * 64;
                    double *const wGL = p.GL + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane_o + ko * 64;
                    static_for<0, NS>([&](auto S) {
                        wGC[S.value * 64] = GCr[S.value];
                        wGL[S.value * 64] = GLr[S.value];
                    });
                    if (x_o) {
                        wGC[-64] = GC0;
                        wGL[-64] = GL0;
                    }
                }
                if (!x_o) {
#if TINY_REFILL
                    double *const wD = p.D + grp_w * (NS * DS) + (REFILL ? (cur_o & 3) : j_o) * NU + (r_o - NX);
#else
                    double *const wD = p.D + (size_t)grp * (NS * DS) + j_o * NU + (r_o - NX);
#endif
                    for (int i = 0; i < NS; ++i) wD[i * DS] = sDw[i * DS];
#if TINY_REFILL
                }
            }
            if constexpr (REFILL) {
                // ---- what the plain kernel does behind its loop, here for the rows that are being written back
                // A converged solve returned before v <- vnew (admm.cpp:181-197): its canonical v|z is the stale copy.
                if (wb && status == 1 && r_o < NXU) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    const int rows = x_o ? N : NS;
                    double *const cV = p.V + (grp_w * v_rows(N) + V_PAD) * 64 + lane_w;
                    const double *const cV2 = p.V2 + (grp_w * v_rows(N) + V_PAD) * 64 + lane_w;
                    for (int kn = 0; kn < rows; ++kn) cV[kn * 64] = cV2[kn * 64];
                }
                const double s_px = group_max<W>(is_x ? snap_pri : 0.0), s_pu = group_max<W>(is_u ? snap_pri : 0.0);
                const double s_dx = group_max<W>(is_x ? snap_dua : 0.0) * rho, s_du = group_max<W>(is_u ? snap_dua : 0.0) * rho;
                if (wb && r_o == 0) {
                    p.istats[(size_t)cur_o * 2 + 0] = it_done;
                    p.istats[(size_t)cur_o * 2 + 1] = status;
                    if (res_valid) {
                        p.dstats[(size_t)cur_o * 4 + 0] = s_px;
                        p.dstats[(size_t)cur_o * 4 + 1] = s_dx;
                        p.dstats[(size_t)cur_o * 4 + 2] = s_pu;
                        p.dstats[(size_t)cur_o * 4 + 3] = s_du;
                    }
                }
                // ---- the next instance of the batch for this row (or none: the row idles as a zombie until the wavefront is done)
                int nxt = 0x7fffffff;
                if (wb && r_o == 0) nxt = (int)(gridDim.x * WPG * IPW) + atomicAdd(p.refill_next, 1);
                nxt = __shfl(nxt, lane_o & 48);
                const bool take = wb && nxt < p.batch;
                if (wb) {
                    active = take;
                    status = 11;
                    res_valid = false;
                    snap_pri = 0.0;
                    snap_dua = 0.0;
                    it_done = 0;
                }
                if (__ballot(take) != 0ull) {
                    if (take) cur = nxt;
                    const size_t grp_n = (size_t)(nxt >> 2);
                    const int lane_n = (nxt & 3) * 16 + r_o;
                    if (take && x_o) x0v = p.x0[(size_t)nxt * NX + r_o];
                    if (cold) {
                        static_for<0, NS>([&](auto S) { G[S.value] = take ? 0.0 : G[S.value]; });
                        static_for<0, NVR>([&](auto S) { Vr[S.value] = take ? 0.0 : Vr[S.value]; });
                        G0 = take ? 0.0 : G0;
                        V0 = take ? 0.0 : V0;
                        if (take) {
                            static_for<0, VL>([&](auto S) { sVl[S.value * 64] = 0.0; });
                            if (!x_o && r_o < NXU)
                                for (int i = 0; i < NS; ++i) sDw[i * DS] = 0.0;
                        }
                    } else if (take) {
                        const double *const nG = p.G + grp_n * (N + 1) * 64 + lane_n;
                        const double *const nV = p.V + (grp_n * v_rows(N) + V_PAD) * 64 + lane_n;
                        static_for<0, NS>([&](auto S) { G[S.value] = nG[(S.value + koff) * 64]; });
                        static_for<0, NVR>([&](auto S) { Vr[S.value] = nV[(VL + S.value + koff) * 64]; });
                        static_for<0, VL>([&](auto S) { sVl[S.value * 64] = nV[(S.value + koff) * 64]; });
                        G0 = nG[0];
                        V0 = nV[0];
                        if (!x_o && r_o < NXU) {
                            const double *const nD = p.D + grp_n * (NS * DS) + (nxt & 3) * NU + (r_o - NX);
                            for (int i = 0; i < NS; ++i) sDw[i * DS] = nD[i * DS];
                        }
                    }
#endif
                }
            }
            pending = false;
        }
        if (final_round || __ballot(active) == 0ull) break;
        const int it1 = it0 + 1;
#if TINY_REFILL
        // admm.cpp:91 (iter already incremented, :143). REFILL: every row checks by its own count; `check` = some live row does.
        bool chk = true, check_r = false;
        if constexpr (REFILL) {
            chk = active && (ct > 0) && (((it_done + 1) % ct) == 0);
            check_r = __ballot(chk) != 0ull;
        }
#define TINY_CHECK_PLAIN (__builtin_amdgcn_readfirstlane((int)((ct > 0) && ((it1 % ct) == 0))) != 0)
        const bool check = REFILL ? check_r : TINY_CHECK_PLAIN;
#undef TINY_CHECK_PLAIN
#else
        const bool check = __builtin_amdgcn_readfirstlane((int)((ct > 0) && ((it1 % ct) == 0))) != 0;  // admm.cpp:91 (iter already incremented, :143)
#endif

        double pri = 0.0, dua = 0.0;
        bool may = check;  // wave-uniform: can this sweep still end converged for some instance of the wave?
        double m[16];
        load_ops(sMf, m);
        // ADAPT: every fifth iteration (admm.cpp:155) the norms of the reference's KKT residuals ride on the forward sweep
        const bool adapt = ADAPT && __builtin_amdgcn_readfirstlane((int)((it0 > 0) && (it0 % 5 == 0))) != 0;
        double a_pr = 0.0, a_pn = 0.0, a_dr = 0.0, a_dn = 0.0;  // primal residual / norm, dual residual / norm
        double gprev = 0.0;  // g_i of the state rows; the x_0 column has no -g_0 term (y_vector starts at g_1)
        double mtr[16];      // this lane's row of [A'; B']
        if constexpr (ADAPT) {
            if (adapt) static_for<0, 16>([&](auto K) { mtr[K.value] = K.value < NXU ? sAd[2 * 256 + K.value * 16 + r] : 0.0; });
        }
        // ---------------- knot 0, state lanes: x_0 is given (tiny_set_x0), no mat-vec
        {
            const double lo0 = CT ? lo_c : sT[W + r], hi0 = CT ? hi_c : sT[TOFF + W + r];
            const double s = x0v + G0;
            const double snew = fmin(hi0, fmax(lo0, s));
            G0 = s - snew;
            pri = is_x ? fabs(x0v - snew) : 0.0;
            dua = is_x ? fabs(V0 - snew) : 0.0;
            // First "can this sweep still converge" test, on knot 0's residuals alone (exact like the later ones: the
            // maxima only grow). With tolerances nothing can meet -- forced iteration counts -- the sweep writes no stale copy.
            if (may && !adapt) {
                const bool bad = !((pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol));
#if TINY_REFILL
                may = __builtin_amdgcn_readfirstlane((int)wave_may_converge_d(__ballot(bad), __ballot(REFILL ? chk : active))) != 0;
#else
                may = __builtin_amdgcn_readfirstlane((int)wave_may_converge_d(__ballot(bad), __ballot(active))) != 0;
#endif
            }
#if TINY_REFILL
            if (TINY_RARE(may) && is_x && active) gV1u[REFILL ? row_offset(cur) : (unsigned)lane] = V0;
#else
            if (TINY_RARE(may) && is_x && active) gV1u[(unsigned)lane] = V0;  // (active: see the stale copy of the slots below)
#endif
            V0 = snew;
        }
        if constexpr (FAM) {  // knot 0 of the state rows (its lx only reaches p_0, which nothing reads; the duals persist)
            double gcn, gln;
            (void)families(x0v, GC0, GL0, gcn, gln);
            if (is_x) {
                GC0 = gcn;
                GL0 = gln;
            }
        }
        // ---------------- forward sweep (F1) with S1 + D1 + R1 fused in
        // LDS operands of a step (its d, and vold of its slot if that lives in LDS) are requested right before the
        // PREVIOUS step's block and retired by the wait behind that block (lds_reads_landed, inside the Step functions).
        double xcur = x0v;
        double dcur = lds_read_async<0>(aD), vcur = 0.0;
        if constexpr (VL > 0) vcur = lds_read_async<0>(aV);
        // (!CT: bounds that vary over the horizon come from the workgroup's LDS copy of the tables, one step ahead like d)
        double locur = lo_c, hicur = hi_c;
        if constexpr (!CT) {
            locur = lds_read_async<W * 8>(aT);
            hicur = lds_read_async<(TOFF + W) * 8>(aT);
        }
        lds_wait_ops(m);
        auto fstep = [&](auto S) {
            constexpr int q = decltype(S)::value;
            double dn = 0.0, vn = 0.0, lon = lo_c, hin = hi_c;
            if constexpr (q + 1 < NS) dn = lds_read_async<(q + 1) * DS * 8>(aD);
            if constexpr (q + 1 < VL) vn = lds_read_async<(q + 1) * 512>(aV);
            if constexpr (!CT && q + 1 < NS) {
                lon = lds_read_async<(q + 2) * W * 8>(aT);
                hin = lds_read_async<(TOFF + (q + 2) * W) * 8>(aT);
            }
            const double xprev = xcur;
            double snew_q;
            if constexpr (q >= VL) {
                xcur = Step::fwd_reg(xcur, dcur, m, cf, locur, hicur, G[q], Vr[q - VL], pri, dua);
                snew_q = Vr[q - VL];
            } else {
                double vnew;
                xcur = Step::fwd_lds(xcur, dcur, m, cf, locur, hicur, G[q], vcur, vnew, pri, dua);
                lds_write_async<q * 512>(aV, vnew);
                snew_q = vnew;
            }
            if constexpr (ADAPT) {
                if (adapt) {
                    // state lanes: column x_q (xprev, gprev) and row vnew_{q+1} (snew); input lanes: column / row u_q
                    const double gnew = G[q], out = xcur;
                    const double t = group_matvec<W, 16>(mtr, is_x ? gnew : 0.0, 0.0);  // [A'; B'] g_{q+1}
                    const double dgx = dgr * (is_x ? xprev : out);                       // Q.*x_q | R.*u_q
                    const double aty = is_x ? (t - gprev) : (gnew + t);
                    a_dn = fmax(fmax(a_dn, fabs(dgx)), fabs(aty));
                    a_dr = fmax(a_dr, fabs(2.0 * dgx + aty));
                    a_pr = fmax(a_pr, fabs(is_x ? snew_q : (out - snew_q)));
                    a_pn = fmax(fmax(a_pn, fabs(snew_q)), is_x ? 0.0 : fabs(out));
                    gprev = gnew;
                }
            }
            if constexpr (FAM) {  // xcur: x_{q+1} on state lanes, u_q on input lanes -- this slot's element
                double gcn, gln;
                const double l = families(xcur, GCr[q], GLr[q], gcn, gln);
                if (row_ok) {
                    GCr[q] = gcn;
                    GLr[q] = gln;
                    LX[q] = l;
                }
            }
            dcur = dn;
            vcur = vn;
            if constexpr (!CT) {
                locur = lon;
                hicur = hin;
            }
        };
        constexpr int DF = D_FIRST < NS ? D_FIRST : NS;
        constexpr int NG = 1 + (NS - DF + D_GROUP - 1) / D_GROUP;
        static_for<0, NG>([&](auto Gi) {
            constexpr int s0 = Gi.value == 0 ? 0 : DF + (Gi.value - 1) * D_GROUP;
            constexpr int s1 = Gi.value == 0 ? DF : ((s0 + D_GROUP < NS) ? s0 + D_GROUP : NS);
            if (TINY_RARE(may)) {
                // Stale copy of the group's slots (still holding the previous iterate) before the blocks overwrite them.
                // Rare path: the addresses are rebuilt from an opaque copy of the lane offset so that the compiler does
                // not keep one pointer per slot alive across the iteration loop.
                // (ADAPT: the check after an adaptation scales the dual residual with the NEW rho, which this sweep does not
                // know yet -- no early "cannot converge" verdict in those iterations)
                if constexpr (s0 > 0) {
                    if (!adapt) {
                        const bool bad = !((pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol));
#if TINY_REFILL
                        may = __builtin_amdgcn_readfirstlane((int)wave_may_converge_d(__ballot(bad), __ballot(REFILL ? chk : active))) != 0;
#else
                        may = __builtin_amdgcn_readfirstlane((int)wave_may_converge_d(__ballot(bad), __ballot(active))) != 0;
#endif
                    }
                }
                if (TINY_RARE(may)) {
#if TINY_REFILL
                    unsigned vo = REFILL ? row_offset(cur) + (unsigned)(koff * 64) : voff;
#else
                    unsigned vo = voff;
#endif
                    double *base = gV1u;
                    asm volatile("" : "+v"(vo), "+s"(base));
                    // Only for instances that are still iterating: a zombie's slots hold iterates it computed AFTER it converged,
                    // and its own stale copy -- the previous iterate of the sweep in which it converged, its canonical v|z
                    // (admm.cpp:181-197) -- must survive the later sweeps of its wavefront's other instances.
                    if (active) static_for<s0, s1>([&](auto S) { (base + S.value * 64)[vo] = vget(S); });
                }
            }
            static_for<s0, s1>([&](auto S) { fstep(S); });
        });
#if TINY_REFILL
        if constexpr (REFILL) {
            if (active) it_done += 1;
        } else {
            if (active) it_done = it1;  // admm.cpp:143
        }
#else
        if (active) it_done = it1;  // admm.cpp:143
#endif

        // ---------------- adaptive rho (admm.cpp:147-174), as in k_admm_solve_adapt
        const double nrho_lin = nrho, rhom_lin = rhom, pnref_lin = pnref;  // what update_linear_cost used this iteration
        if constexpr (ADAPT) {
            if (adapt) {
                const double x_last = xcur, g_last = gprev;  // x_{N-1}, g_{N-1} on state lanes
                // column block x_{N-1}: Pinf x + Q.*x - g_{N-1} with the CURRENT (adapted) Pinf (rho_benchmark.cpp:112)
                double pr[16];
                const double delta = rho - rho0;
                static_for<0, 16>([&](auto K) {
                    constexpr int o = K.value * 16;
                    pr[K.value] = K.value < NXU ? fma(delta, sAd[4 * 256 + o + r], sAd[3 * 256 + o + r]) : 0.0;
                });
                const double pxl = group_matvec<W, 16>(pr, is_x ? x_last : 0.0, 0.0);
                if (is_x) {
                    const double qv = dgr * x_last;
                    a_dn = fmax(fmax(fmax(a_dn, fabs(pxl)), fabs(qv)), fabs(g_last));
                    a_dr = fmax(a_dr, fabs(pxl + qv - g_last));
                }
                const double pri_res = group_max<W>(a_pr), pri_norm = group_max<W>(a_pn);
                const double dual_res = group_max<W>(a_dr), dual_norm = group_max<W>(a_dn);
                const double eps = 1e-10;  // rho_benchmark.cpp:190-197
                const double normalized_pri = pri_res / (pri_norm + eps);
                const double normalized_dual = dual_res / (dual_norm + eps);
                const double ratio = normalized_pri / (normalized_dual + eps);
                double new_rho = rho * sqrt(ratio);
                if (p.rho_clip) new_rho = fmin(fmax(new_rho, p.rho_min), p.rho_max);
                if (active) {  // (rho is unchanged for instances that are no longer iterating)
                    rho = new_rho;
                    nrho = -new_rho;
                    rhom = is_x ? nrho : 0.0;
                    pnref = fma(rho - rho0, dpnref, pnref0);
                }
            }
        }

        // ---------------- R1: termination (admm.cpp:93-101), decided element-wise: one ballot, no reductions
        if (check) {
            const bool below = (pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol);
            const bool conv = ((__ballot(below) >> (j * W)) & 0xffffull) == 0xffffull;
#if TINY_REFILL
            if (REFILL ? chk : active) {
#else
            if (active) {
#endif
                snap_pri = pri;
                snap_dua = dua;
                if constexpr (ADAPT) snap_rho = rho;  // cache->rho AFTER the adaptation (admm.cpp:95-96)
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
            auto lr_of = [&](auto S) -> double {  // linref of slot S (its knot differs by lane type) (+ the families' term)
                double base;
                if constexpr (CT) base = lr_c;
                else base = sTl[2 * TOFF + (S.value + 1) * W];
                if constexpr (FAM) base += LX[S.value];
                return base;
            };
            double px, rcur, rnext, acc;
            {   // p_{N-1} (state lanes, admm.cpp:81-82) | r_{N-2} (input lanes) share slot NS-1; then slot NS-2
                // (ADAPT: the linear cost of this iteration was formed BEFORE the adaptation -- rho / pNref of update_linear_cost --,
                // the operators loaded above carry the new Kinf)
                double lrT = is_x ? pnref_lin : lr_of(std::integral_constant<int, NS - 1>{});
                if constexpr (FAM) lrT = is_x ? pnref_lin + LX[NS - 1] : lrT;
                const double lr2 = lr_of(std::integral_constant<int, NS - 2>{});
                const double lrmc2 = is_x ? lr2 + cb : cb;
                const double v1 = vget(std::integral_constant<int, NS - 1>{}), v2 = vget(std::integral_constant<int, NS - 2>{});
                double t;
                if constexpr (ADAPT) {  // (rho is a lane variable here: vector operand)
                    asm("v_add_f64 %[t], %[v1], -%[g1]\n\t"
                        "v_fma_f64 %[px], %[nrho], %[t], %[lrT]\n\t"
                        "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                        "v_fma_f64 %[acc], %[rhom], %[t], %[lrmc]\n\t"
                        "v_fma_f64 %[rn], %[nrho], %[t], %[lr]"
                        : [t] "=&v"(t), [px] "=&v"(px), [acc] "=&v"(acc), [rn] "=&v"(rnext)
                        : [v1] "v"(v1), [g1] "v"(G[NS - 1]), [v2] "v"(v2), [g2] "v"(G[NS - 2]), [nrho] "v"(nrho_lin), [lrT] "v"(lrT),
                          [rhom] "v"(rhom_lin), [lrmc] "v"(lrmc2), [lr] "v"(lr2));
                } else {
                    asm("v_add_f64 %[t], %[v1], -%[g1]\n\t"
                        "v_fma_f64 %[px], %[nrho], %[t], %[lrT]\n\t"
                        "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                        "v_fma_f64 %[acc], %[rhom], %[t], %[lrmc]\n\t"
                        "v_fma_f64 %[rn], %[nrho], %[t], %[lr]"
                        : [t] "=&v"(t), [px] "=&v"(px), [acc] "=&v"(acc), [rn] "=&v"(rnext)
                        : [v1] "v"(v1), [g1] "v"(G[NS - 1]), [v2] "v"(v2), [g2] "v"(G[NS - 2]), [nrho] "s"(nrho), [lrT] "v"(lrT),
                          [rhom] "v"(rhom), [lrmc] "v"(lrmc2), [lr] "v"(lr2));
                }
                rcur = px;
            }
            // slack operand of a block's tail: a register, or an LDS read issued one block ahead
            auto vreq = [&](auto S) -> double {
                if constexpr (decltype(S)::value >= VL) return Vr[decltype(S)::value - VL];
                else return lds_read_async<decltype(S)::value * 512>(aV);
            };
            double v2cur = vreq(std::integral_constant<int, (NS >= 3 ? NS - 3 : 0)>{});
            lds_wait_ops(m);
            static_for<0, NS - 1>([&](auto I) {
                constexpr int s = NS - 1 - I.value;           // NS-1 .. 1
                constexpr int s2 = s >= 2 ? s - 2 : 0;        // slot feeding the tail (s = 1: any finite t will do)
                constexpr int s3 = s >= 3 ? s - 3 : 0;        // ... of the next block
                // (an asynchronous read MUST be consumed after its wait: the destination of a dead one would be handed to
                // the block's outputs while the read is still in flight)
                double v2n = 0.0;
                if constexpr (s >= 2) v2n = vreq(std::integral_constant<int, s3>{});
                const double lr2 = lr_of(std::integral_constant<int, s2>{});
                const double lrmc2 = is_x ? lr2 + cb : cb;
                double a = acc, an, rn;
                if constexpr (ADAPT) Step::bwd_v(a, px, rcur, m, v2cur, G[s2], rhom_lin, lrmc2, nrho_lin, lr2, an, rn);
                else Step::bwd(a, px, rcur, m, v2cur, G[s2], rhom, lrmc2, nrho, lr2, an, rn);
                lds_write_masked<s * DS * 8>(aD, a, wr_d);  // d_s
                px = a;
                rcur = rnext;
                rnext = rn;
                acc = an;
                v2cur = v2n;
            });
            {
                double a = acc;
                Step::bwd_last(a, px, rcur, m);
                lds_write_masked<0>(aD, a, wr_d);  // d_0
            }
        }
    }
    lds_wait();
#ifdef TINY_CLOCK_STAMP
    // Delta s_memtime (shader cycles) and Delta s_memrealtime (a constant 100 MHz) around the iteration loop, one pair per
    // wavefront, into a buffer of their own that nothing else reads (SolveParams::scratch, unused by this layout otherwise)
    const unsigned long long ck_t1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
#endif

    // A converged solve returned before v <- vnew (admm.cpp:181-197): its canonical v|z is the previous iterate, i.e. the
    // stale copy. (The write-back above stored vnew there; this wave wrote both, in program order.)
#if TINY_REFILL
    if constexpr (REFILL) return;  // (every row was finished inside the loop)
#endif
    if (inst_ok && status == 1 && r < NXU) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const int rows = is_x ? N : NS;
        double *const wV = p.V + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane;
        const double *const wV2 = p.V2 + ((size_t)grp * v_rows(N) + V_PAD) * 64 + lane;
        for (int kn = 0; kn < rows; ++kn) wV[kn * 64] = wV2[kn * 64];
    }

    const double res_px = group_max<W>(is_x ? snap_pri : 0.0), res_pu = group_max<W>(is_u ? snap_pri : 0.0);
    const double rho_res = ADAPT ? snap_rho : rho;
    const double res_dx = group_max<W>(is_x ? snap_dua : 0.0) * rho_res, res_du = group_max<W>(is_u ? snap_dua : 0.0) * rho_res;

#ifdef TINY_CLOCK_STAMP
    // One record per wavefront, into a buffer of its own that nothing else reads (SolveParams::scratch, unused by this layout
    // otherwise): shader cycles and 100 MHz ticks of the iteration loop (the final round's write-back included), and the ticks of
    // the phases around it -- entry -> the prologue's barrier (operators, d and the LDS part of the slack in), barrier -> loop
    // (the register part of the state in), loop end -> here (residual reductions), absolute entry / exit times.
    if (p.scratch && lane == 0) {
        unsigned long long *ck = reinterpret_cast<unsigned long long *>(p.scratch) + 8 * (size_t)grp;
        ck[0] = ck_t1 - ck_t0;
        ck[1] = ck_r1 - ck_r0;
        ck[2] = ck_barrier - ck_entry;
        ck[3] = ck_r0 - ck_barrier;
        ck[4] = __builtin_amdgcn_s_memrealtime() - ck_r1;
        ck[5] = ck_entry;
        ck[6] = __builtin_amdgcn_s_memrealtime();
        ck[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);  // HW_ID, XCC_ID
    }
#endif
    if (inst_ok && r == 0) {
        if constexpr (ADAPT) p.rho_inst[inst] = rho;
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
#if TINY_REFILL
template <int NX, int NU, int N, bool CT, int WPG, int VL>
__global__ void __launch_bounds__(64 * WPG) __attribute__((amdgpu_waves_per_eu(2, 2))) k_admm_solve_d_refill(const SolveParams p) {
#else
template <int NX, int NU, int N, bool CT, int WPG, int VL, bool HOSTX = false>
__global__ void __launch_bounds__(64 * WPG) __attribute__((amdgpu_waves_per_eu(2, 2))) k_admm_solve_d(const SolveParams p) {
#endif
    extern __shared__ __attribute__((aligned(16))) double smem[];
#if TINY_REFILL
    k_admm_solve_d_body<NX, NU, N, CT, WPG, VL, false, false, true, false, true>(p, smem);
#else
    k_admm_solve_d_body<NX, NU, N, CT, WPG, VL, false, false, true, HOSTX>(p, smem);
#endif
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
#ifndef TINY_JIT_FAM
#define TINY_JIT_FAM 0
#endif
    constexpr bool FAMJ = TINY_JIT_FAM != 0;  // cone / linear-inequality families
#ifndef TINY_JIT_ADAPT
#define TINY_JIT_ADAPT 0
#endif
    constexpr bool ADJ = TINY_JIT_ADAPT != 0;  // adaptive rho
    constexpr int VLJ = tinympc::d_vl(TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, 4 * TINY_JIT_WPS, FAMJ, ADJ);
    static_assert(VLJ >= 0, "shape does not fit the layout-D plan");
    __shared__ __attribute__((aligned(16))) double smem_jit[tinympc::d_lds_bytes(TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ, FAMJ, ADJ) / sizeof(double)];
#if TINY_REFILL
#ifndef TINY_JIT_REFILL
#define TINY_JIT_REFILL 0
#endif
    tinympc::k_admm_solve_d_body<TINY_JIT_NX, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ, FAMJ, ADJ, TINY_JIT_WPS == 2, false, TINY_JIT_REFILL != 0>(p, smem_jit);
#else
    tinympc::k_admm_solve_d_body<TINY_JIT_NX, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ, FAMJ, ADJ, TINY_JIT_WPS == 2>(p, smem_jit);
#endif
}
namespace tinympc {
#else
// ------------------------------------------------------------------------------------------------------------
// Host side: the instantiation table. A shape runs on layout D only if it was compiled in.
// ------------------------------------------------------------------------------------------------------------
// Wavefronts per workgroup: four where the LDS plan allows it, else eight. Small workgroups spread a batch over the CUs
// wavefront by wavefront: 4,096 quadrotor instances are 1,024 wavefronts = 256 workgroups of four = ONE wavefront per SIMD on
// every CU (6.4 us per iteration, 634 M iterations/s), where 128 workgroups of eight filled half the chip with two wavefronts
// per SIMD (10.1 us, 406 M); full batches gain a few percent from the finer tail (profiles/r02_layout_sweep.txt).
__host__ __device__ constexpr int d_wpg(int nu, int N, bool ct) { return d_vl(nu, N, ct, 4) >= 0 ? 4 : 8; }

#if !TINY_REFILL
// Slot refill launches one resident set: 2 wavefronts per SIMD x 4 SIMDs x the device's CUs, in workgroups of wpg.
int solve_d_resident_workgroups(int wpg) {
    // compute units of the CURRENT device, cached per device id (devices of one node may differ in partition mode); the slots
    // are written at most once each with the same value, so a relaxed atomic is all the synchronisation they need
    static std::atomic<int> cus_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = cus_of[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus_of[dev].store(cus = n, std::memory_order_relaxed);
    }
    return cus * 8 / wpg;
}
int solve_d_wavefronts_per_workgroup(int nu, int N, bool const_tables) { return d_wpg(nu, N, const_tables); }

#endif
#if TINY_REFILL
// (the slot-refill translation unit, tinympc_solve_dr.hip: one resident set of wavefronts, tinympc_plan.hip decides when)
template <int NX, int NU, int N, bool CT>
static hipError_t launch_d_one(const SolveParams &p, hipStream_t stream) {
    constexpr int WPG = d_wpg(NU, N, CT);
    constexpr int VL = d_vl(NU, N, CT, WPG);
    if constexpr (VL < 0) {
        return hipErrorInvalidValue;
    } else {
        constexpr size_t lds = d_lds_bytes(NU, N, CT, WPG, VL);
        static size_t lds_set_r[16] = {0};
        const int wgs = (p.groups + WPG - 1) / WPG;
        auto fn = &k_admm_solve_d_refill<NX, NU, N, CT, WPG, VL>;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), lds, lds_set_r);
        if (e != hipSuccess) return e;
        const int resident = solve_d_resident_workgroups(WPG);
        hipLaunchKernelGGL(fn, dim3(wgs < resident ? wgs : resident), dim3(64 * WPG), lds, stream, p);
        return hipGetLastError();
    }
}
#else
template <int NX, int NU, int N, bool CT>
static hipError_t launch_d_one(const SolveParams &p, hipStream_t stream) {
    constexpr int WPG = d_wpg(NU, N, CT);
    constexpr int VL = d_vl(NU, N, CT, WPG);
    if constexpr (VL < 0) {
        return hipErrorInvalidValue;
    } else {
        constexpr size_t lds = d_lds_bytes(NU, N, CT, WPG, VL);
        static size_t lds_set[16] = {0}, lds_set_x[16] = {0};
        const int wgs = (p.groups + WPG - 1) / WPG;
        if (p.x0_mirror || p.u0_host) {  // (the batched zero-copy tick)
            auto fn = &k_admm_solve_d<NX, NU, N, CT, WPG, VL, true>;
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), lds, lds_set_x);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(fn, dim3(wgs), dim3(64 * WPG), lds, stream, p);
        } else {
            auto fn = &k_admm_solve_d<NX, NU, N, CT, WPG, VL>;
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), lds, lds_set);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(fn, dim3(wgs), dim3(64 * WPG), lds, stream, p);
        }
        return hipGetLastError();
    }
}

#endif

#define TINY_D_SHAPES(X) \
    X(12, 4, 50)         \
    X(4, 1, 20)          \
    X(4, 1, 10)

#if TINY_REFILL
hipError_t launch_solve_d_refill(const SolveParams &p, hipStream_t stream) {
#define X(NX_, NU_, N_)                                       \
    if (p.nx == NX_ && p.nu == NU_ && p.N == N_)              \
        return p.const_tables ? launch_d_one<NX_, NU_, N_, true>(p, stream) : launch_d_one<NX_, NU_, N_, false>(p, stream);
    TINY_D_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}
#else
bool solve_d_supported(int nx, int nu, int N, bool const_tables) {
#define X(NX_, NU_, N_) \
    if (nx == NX_ && nu == NU_ && N == N_) return d_vl(NU_, N_, const_tables, d_wpg(NU_, N_, const_tables)) >= 0;
    TINY_D_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_solve_d(const SolveParams &p, hipStream_t stream) {
    if (p.refill_next) return launch_solve_d_refill(p, stream);  // (tinympc_solve_dr.hip)
#define X(NX_, NU_, N_)                                       \
    if (p.nx == NX_ && p.nu == NU_ && p.N == N_)              \
        return p.const_tables ? launch_d_one<NX_, NU_, N_, true>(p, stream) : launch_d_one<NX_, NU_, N_, false>(p, stream);
    TINY_D_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

size_t solve_d_lds_bytes(int nu, int N, bool const_tables) {
    const int wpg = d_wpg(nu, N, const_tables);
    return d_lds_bytes(nu, N, const_tables, wpg, d_vl(nu, N, const_tables, wpg));
}

int solve_d_workgroups(int nu, int N, bool const_tables, int groups) {
    const int wpg = d_wpg(nu, N, const_tables);
    return (groups + wpg - 1) / wpg;
}

#endif  // TINY_REFILL

#endif  // TINY_JIT

}  // namespace tinympc
