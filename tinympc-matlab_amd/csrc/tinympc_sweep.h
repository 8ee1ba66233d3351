// tinympc_sweep.h -- device helpers shared by the two solve kernels (layout A: tinympc_solve.hip,
// layout B: tinympc_solve_b.hip): the fused DPP mat-vec chain, the row-local projection block, and the
// group-wide max reduction.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "tinympc_device.h"

namespace tinympc {

// ------------------------------------------------------------------------------------------------
// LDS reads whose ISSUE POINT the kernel chooses and whose arrival the COMPILER tracks (layouts D, E and the wide forms).
//
// The register-resident sweeps request a step's LDS operands one block ahead, in the shadow of the previous step's ~25 FP64
// instructions, and retire them with one s_waitcnt behind that block. Through round 5 the requests were `asm volatile("ds_read_b64")`
// and the wait a line of the block's own asm text: the compiler knew of neither, believed the destination register valid at once,
// and under register pressure (the 512-register plans, whose overflow lives in AGPRs) it was free to COPY the register, or hand it
// to other code, between the request and the wait -- tools/fuzz_wide_families.py found a shape (nx 48, nu 14, N 6, per-knot bounds)
// where it parked the still-empty bound in an AGPR and reused the register for the operand vector (profiles/r05_inflight_bug.txt).
// Now: the request is a VOLATILE load (volatile pins its place among the asm volatile blocks; for LDS the memory legalizer adds
// no wait of its own) and the wait is __builtin_amdgcn_s_waitcnt -- both in the compiler's scoreboard. The instruction stream is the
// one the hand schedule wants; should register allocation ever touch an in-flight value early, the compiler now inserts the wait
// it needs instead of reading garbage. tools/inflight_lint.py checks the old form's assembly; the new form cannot violate it.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) double lds_double_as3;
__device__ __forceinline__ unsigned lds_address(const double *p) { return (unsigned)(size_t)(const lds_double_as3 *)p; }
template <int OFF>
__device__ __forceinline__ double lds_read_issued_here(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds offset field is 16 bits");
    return *((const volatile lds_double_as3 *)(size_t)addr + OFF / 8);
}
// s_waitcnt lgkmcnt(0) (vmcnt / expcnt untouched) that the compiler's wait-count pass sees: every LDS read issued so far has landed
__device__ __forceinline__ void lds_reads_landed() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// ------------------------------------------------------------------------------------------------
// References handed over in pinned host memory (single-instance handles; SolveParams::href_x). The ONE workgroup of
// the launch recomputes what k_build_tables derives from them -- the linref rows -(Xref .* Q), -(Uref .* R)
// (admm.cpp:77, 79) and pNref = -(Xref_{N-1}' Pinf)' (admm.cpp:81), same expressions, same order of operations, so the
// tables are bit-identical to a k_build_tables launch -- and mirrors the references into their device copies. Must be
// called by all threads of the workgroup before anything reads p.tables; ends with a barrier.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void refresh_reference_tables(const SolveParams &p, int W, int KT) {
    if (!p.href_x) return;  // kernel argument: uniform
    const int nx = p.nx, nu = p.nu, N = p.N, nxu = nx + nu, nthreads = (int)blockDim.x;
    // Phase 1: pinned host -> device copies, every thread's reads in flight at once (a host read is a PCIe round trip
    // of a microsecond or two; anything that waits for them one after the other -- the pNref dot products did, in a
    // first version -- costs that per element).
    const int X = nx * N, U = nu * (N - 1);
    for (int i = threadIdx.x; i < X; i += nthreads) p.dXref[i] = p.href_x[i];
    for (int i = threadIdx.x; i < U; i += nthreads) p.dUref[i] = p.href_u[i];
    __threadfence_block();
    __syncthreads();
    // Phase 2: the table rows, from the device copies (L2)
    const size_t TR = (size_t)(N + 2) * W;
    double *tab = const_cast<double *>(p.tables);
    double *lr = tab + 2 * TR, *pn = lr + TR;
    const double *dg = p.ops + (size_t)2 * W * KT + 2 * W;
    const double *Xr = p.dXref, *Ur = p.dUref;
    for (int idx = threadIdx.x; idx < N * W; idx += nthreads) {
        const int kn = idx / W, r = idx % W;
        double ref = 0.0;
        if (r < nx) ref = -(Xr[r + (size_t)kn * nx] * dg[r]);
        else if (r < nxu && kn < N - 1) ref = -(Ur[(r - nx) + (size_t)kn * nu] * dg[r]);
        lr[(size_t)(kn + 1) * W + r] = ref;
    }
    for (int c = threadIdx.x; c < W; c += nthreads) {
        double acc = 0.0;
        if (c < nx) {
            // same sum, same order as k_build_tables; the operands are fetched eight at a time so that the loop does
            // not pay one L2 round trip per term (products beyond nx are exact zeros)
            for (int k0 = 0; k0 < nx; k0 += 8) {
                double a[8], b[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool in = k0 + q < nx;
                    a[q] = in ? Xr[(k0 + q) + (size_t)(N - 1) * nx] : 0.0;
                    b[q] = in ? p.Pinf[(k0 + q) + (size_t)c * nx] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k0 + q < nx) acc += a[q] * b[q];
            }
            acc = -acc;
        }
        pn[c] = acc;
    }
    __threadfence_block();
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Wide systems (16 < nx+nu <= 64): an instance spans 2 (W = 32) or 4 (W = 64) DPP rows of 16 lanes. The operand
// vector of a sweep step is spread over those rows; `row_newbcast` only reaches inside a row, so the operand is first
// replicated ACROSS the rows of the instance with the gfx950 cross-row swaps -- v_permlane16_swap (odd rows of the
// first operand <-> even rows of the second) and v_permlane32_swap (upper half of the first <-> lower half of the
// second), applied to two copies of the same register (tools/microbench_mfma_f64.hip prints their lane maps):
//   W = 32   permlane16_swap(w, w)              -> E = [r0 r0 r2 r2], O = [r1 r1 r3 r3]
//   W = 64   permlane32_swap(w, w)              -> P = [r0 r1 r0 r1], Q = [r2 r3 r2 r3]
//            permlane16_swap(P, P), (Q, Q)      -> R0 .. R3 = [rj rj rj rj]
// after which the mat-vec is the same fused chain as for W = 16, 16 columns per replicated operand:
// acc += m[16 j + k] * Rj(row_newbcast:k). Per step: 2 (6) swap instructions on 32-bit halves + a few copies + KT FMAs,
// where the first version of this path issued, per column, two ds_bpermute (W = 32) or two v_readlane (W = 64) in
// front of the FMA -- on the serial x_i -> x_{i+1} chain.
// Hazards: the swaps write VGPRs that the chain reads through DPP (2 wait states, `s_nop 1` opens every operand's
// first block; checked by tools/isa_lint.py, which knows that the swaps write both of their operands).
// ------------------------------------------------------------------------------------------------
typedef unsigned tiny_uint2 __attribute__((ext_vector_type(2)));

template <int MODE16>
__device__ __forceinline__ void cross_row_pair(double w, double &lo_rows, double &hi_rows) {
    const unsigned l = (unsigned)__double2loint(w), h = (unsigned)__double2hiint(w);
    tiny_uint2 a, b;
    if constexpr (MODE16) {
        a = __builtin_amdgcn_permlane16_swap(l, l, false, false);
        b = __builtin_amdgcn_permlane16_swap(h, h, false, false);
    } else {
        a = __builtin_amdgcn_permlane32_swap(l, l, false, false);
        b = __builtin_amdgcn_permlane32_swap(h, h, false, false);
    }
    lo_rows = __hiloint2double((int)b[0], (int)a[0]);
    hi_rows = __hiloint2double((int)b[1], (int)a[1]);
}

// acc += sum_{k<16} m[k] * (lane k of the DPP row of `w`), as four schedulable blocks of four fused DPP FMAs
#define TINY_FMQ(i) "v_fmac_f64_dpp %[a], %[w], %[m" #i "] row_newbcast:%[b" #i "] row_mask:0xf bank_mask:0xf\n\t"
template <int B0>
__device__ __forceinline__ void dpp_block4(double &a, double w, const double *m4, bool first) {
    if (first)
        asm("s_nop 1\n\t" TINY_FMQ(0) TINY_FMQ(1) TINY_FMQ(2) TINY_FMQ(3)
            : [a] "+&v"(a)
            : [w] "v"(w), [m0] "v"(m4[0]), [m1] "v"(m4[1]), [m2] "v"(m4[2]), [m3] "v"(m4[3]), [b0] "n"(B0), [b1] "n"(B0 + 1), [b2] "n"(B0 + 2), [b3] "n"(B0 + 3));
    else
        asm(TINY_FMQ(0) TINY_FMQ(1) TINY_FMQ(2) TINY_FMQ(3)
            : [a] "+&v"(a)
            : [w] "v"(w), [m0] "v"(m4[0]), [m1] "v"(m4[1]), [m2] "v"(m4[2]), [m3] "v"(m4[3]), [b0] "n"(B0), [b1] "n"(B0 + 1), [b2] "n"(B0 + 2), [b3] "n"(B0 + 3));
}
template <int NCOLS>
__device__ __forceinline__ void dpp_row16(double &a, double w, const double *m16) {
    static_assert(NCOLS % 4 == 0 && NCOLS >= 4 && NCOLS <= 16, "columns per replicated operand: 4, 8, 12 or 16");
    dpp_block4<0>(a, w, m16, true);
    if constexpr (NCOLS > 4) dpp_block4<4>(a, w, m16 + 4, false);
    if constexpr (NCOLS > 8) dpp_block4<8>(a, w, m16 + 8, false);
    if constexpr (NCOLS > 12) dpp_block4<12>(a, w, m16 + 12, false);
}

#ifdef TINYMPC_WIDE_SHFL  // dev switch (tools/build_variants.sh): the first version of the wide path, for A/B runs
template <int W, int K>
__device__ __forceinline__ double group_bcast(double w) {
    if constexpr (W == 64) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(w), K);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(w), K);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl(w, K, W);
    }
}

template <int W, int KT, int K = 0>
__device__ __forceinline__ void matvec_accumulate(const double (&m)[KT], double w, double (&acc)[2]) {
    if constexpr (K < KT) {
        acc[K & 1] = fma(m[K], group_bcast<W, K>(w), acc[K & 1]);
        matvec_accumulate<W, KT, K + 1>(m, w, acc);
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// W = 16: the mat-vec as ONE chain of fused VOP2+DPP instructions, acc += m_k * (w of lane k of the
// 16-lane DPP row = of the instance). hipcc lowers the builtin form to v_mov_b64_dpp + v_fma_f64 pairs
// (its DPP combiner does not fold 64-bit moves); the fused form halves the instruction count:
// 64 vs 89 ns per 16x16 step on MI355X, bit-identical (tools/microbench_matvec.hip). A single
// accumulator is fastest (11.35 vs 11.96 / 12.21 ms per launch for 1 / 2 / 4 partial sums): the wave is
// issue-bound, not latency-bound, so extra partial sums only add moves and adds.
// Hazard: a VALU-written VGPR read through DPP needs 2 wait states, which hipcc does not insert
// inside inline asm -> `s_nop 1` opens the chain (w was just produced by a v_cndmask).
// The accumulator is an early-clobber operand ("+&v"): with plain "+v" the register allocator may give `a` and `w` ONE
// register when both hold the same value at the call (e.g. a mat-vec of a constant zero vector onto a zero start), and the
// chain would then overwrite its own operand (found by the ISA lint in layout F's carry recurrence).
// The chain is emitted as blocks of FOUR instructions, not volatile: the scheduler then drops the step's other
// work (LDS / global accesses, address arithmetic, selects) into the gaps between the blocks, where it issues in the
// shadow of the dependent FP64 chain instead of after it -- 4.79 -> 4.29 ms per launch on layout B (8,192 quadrotor
// instances), bit-identical. Only the first block carries the s_nop: `w` is an input of all four and was
// produced before the first. Should the register allocator ever copy `w` between two blocks, the copy would sit
// right in front of a DPP read; tests/test_isa_hazards.py compiles the kernels and checks every
// v_fmac_f64_dpp of the generated code for that (a nop in every block costs 7 %).
// ------------------------------------------------------------------------------------------------
#define TINY_FM(i) "v_fmac_f64_dpp %[a], %[w], %[m" #i "] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"
#define TINY_M8 [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [m4] "v"(m[4]), [m5] "v"(m[5]), [m6] "v"(m[6]), [m7] "v"(m[7])
#define TINY_M12 TINY_M8, [m8] "v"(m[8]), [m9] "v"(m[9]), [m10] "v"(m[10]), [m11] "v"(m[11])
#define TINY_M16 TINY_M12, [m12] "v"(m[12]), [m13] "v"(m[13]), [m14] "v"(m[14]), [m15] "v"(m[15])
#define TINY_FM8 TINY_FM(0) TINY_FM(1) TINY_FM(2) TINY_FM(3) TINY_FM(4) TINY_FM(5) TINY_FM(6) TINY_FM(7)
#define TINY_FM12 TINY_FM8 TINY_FM(8) TINY_FM(9) TINY_FM(10) TINY_FM(11)
#define TINY_FM16 TINY_FM12 TINY_FM(12) TINY_FM(13) TINY_FM(14) TINY_FM(15)

// out = c + sum_k m[k] * w_k, with w_k the operand held by lane k of the group.
template <int W, int KT>
__device__ __forceinline__ double group_matvec(const double (&m)[KT], double w, double c) {
    if constexpr (W == 16) {
        static_assert(KT == 8 || KT == 12 || KT == 16, "W=16 supports KT 8, 12, 16");
        double a = c;
        asm("s_nop 1\n\t" TINY_FM(0) TINY_FM(1) TINY_FM(2) TINY_FM(3) : [a] "+&v"(a) : [w] "v"(w), TINY_M8);
        asm(TINY_FM(4) TINY_FM(5) TINY_FM(6) TINY_FM(7) : [a] "+&v"(a) : [w] "v"(w), TINY_M8);
        if constexpr (KT >= 12) asm(TINY_FM(8) TINY_FM(9) TINY_FM(10) TINY_FM(11) : [a] "+&v"(a) : [w] "v"(w), TINY_M12);
        if constexpr (KT == 16) asm(TINY_FM(12) TINY_FM(13) TINY_FM(14) TINY_FM(15) : [a] "+&v"(a) : [w] "v"(w), TINY_M16);
        return a;
    } else {
#ifdef TINYMPC_WIDE_SHFL
        double acc[2] = {c, 0.0};
        matvec_accumulate<W, KT>(m, w, acc);
        return acc[0] + acc[1];
#else
        double a = c;
        if constexpr (W == 32) {
            static_assert(KT > 16 && KT <= 32 && KT % 4 == 0, "W=32: 16 < KT <= 32, a multiple of 4");
            double e, o;
            cross_row_pair<1>(w, e, o);
            dpp_row16<16>(a, e, m);
            dpp_row16<KT - 16>(a, o, m + 16);
        } else {
            static_assert(W == 64 && KT > 32 && KT <= 64 && KT % 4 == 0, "W=64: 32 < KT <= 64, a multiple of 4");
            double pq0, pq1, r0, r1, r2, r3;
            cross_row_pair<0>(w, pq0, pq1);
            cross_row_pair<1>(pq0, r0, r1);
            cross_row_pair<1>(pq1, r2, r3);
            dpp_row16<16>(a, r0, m);
            dpp_row16<16>(a, r1, m + 16);
            dpp_row16<(KT - 32 > 16 ? 16 : KT - 32)>(a, r2, m + 32);
            if constexpr (KT > 48) dpp_row16<KT - 48>(a, r3, m + 48);
        }
        return a;
#endif
    }
}

// S1 + D1 + R1 for one (row, knot) element (admm.cpp:45-58, 67-68, 93-96), 8 FP64 instructions:
//   s = val + g ; snew = min(hi, max(lo, s)) ; gnew = s - snew ;
//   pri = max(pri, |val - snew|) ; dua = max(dua, |vold - snew|)
// Written as one asm block: fmin()/fmax() would make hipcc add a canonicalising v_max_f64 x,x,x per
// operand (sNaN quieting) and every separate asm statement costs a boundary s_nop.
__device__ __forceinline__ void project_element(double val, double g, double lo, double hi, double vold, double &gnew,
                                                double &snew, double &pri, double &dua) {
    double s, t;
    asm("v_add_f64 %[s], %[val], %[g]\n\t"
        "v_max_f64 %[sn], %[lo], %[s]\n\t"
        "v_min_f64 %[sn], %[hi], %[sn]\n\t"
        "v_add_f64 %[gn], %[s], -%[sn]\n\t"
        "v_add_f64 %[t], %[val], -%[sn]\n\t"
        "v_max_f64 %[pri], %[pri], |%[t]|\n\t"
        "v_add_f64 %[t], %[vold], -%[sn]\n\t"
        "v_max_f64 %[dua], %[dua], |%[t]|"
        : [s] "=&v"(s), [sn] "=&v"(snew), [gn] "=&v"(gnew), [t] "=&v"(t), [pri] "+v"(pri), [dua] "+v"(dua)
        : [val] "v"(val), [g] "v"(g), [lo] "v"(lo), [hi] "v"(hi), [vold] "v"(vold));
}

// ------------------------------------------------------------------------------------------------
// Row-local math of the cone / linear-inequality families (PARITY UNPINNED upstream semantics: Euclidean projection onto
// the second-order cone ||w|| <= mu t, and onto half-spaces one after another), written for issue slots, not for the
// textbook: the second-order-cone
// projection needs sqrt(a2) AND u0 / sqrt(a2) -- one v_rsq_f64 seed and two coupled Goldschmidt steps give both
// (g -> sqrt(a2) with a final residual correction, 2h -> 1/sqrt(a2)), where sqrt() followed by a division costs two
// such sequences plus the division's scaling / fix-up instructions (~70 -> ~20 instructions per element); divisions by
// per-lane constants (the cone's slope, ||a_k||^2) are multiplications by reciprocals computed once per launch; the
// three-way case distinction is selects, not branches. Results differ from the literal formulas by rounding
// only (<= 2 ulp per element; the tests assert 1e-9 over whole solves).
// ------------------------------------------------------------------------------------------------
// role: 0 = row in no cone (returned unchanged), 1 = norm member, 2 = the cone's last ("t") row.
// sv: the row's own entry; a2: sum of squares of the cone's norm members; t: the cone's last entry.
__device__ __forceinline__ double soc_project_element(double sv, double a2, double t, double mu, double inv_mu, int role) {
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
    const double a = fma(d, h, g);        // sqrt(a2)
    const double inv_a = h + h;           // 1 / sqrt(a2)
    const double scale = 0.5 * (1.0 + u0 * inv_a);
    const double proj = (role == 1) ? scale * sv : scale * (a * inv_mu);
    const bool apex = a <= -u0, inside = a <= u0;
    const double vc = apex ? 0.0 : (inside ? sv : proj);
    return role != 0 ? vc : sv;
}
// One half-space a_k' s <= b_k: sv <- sv - ((dot - b_k) / ||a_k||^2) a_k  if violated. inv_nk = 1 / ||a_k||^2.
__device__ __forceinline__ double halfspace_project_element(double sv, double dot, double ak, double bk, double inv_nk) {
    const double dist = (dot - bk) * inv_nk;
    return dot > bk ? fma(-dist, ak, sv) : sv;
}

// ------------------------------------------------------------------------------------------------
// Fair share of a SIMD between the two wavefronts of the register-resident kernels (layout D and its wide forms). The issue
// arbiter prefers the older wavefront: left alone, the wavefront in slot 0 of every SIMD finished its solve at 1.18 ms and its
// partner at 1.92 ms of the same launch, the partner running ALONE -- every step's LDS wait exposed -- for the last third of the
// kernel (in-kernel stamps per wavefront with HW_REG_HW_ID, tools/clock_check.py --hwid). The user priority therefore
// alternates in WALL-CLOCK slices (s_memrealtime: 100 MHz, the same for every wavefront): in even slices of 2^14 ticks
// (164 us) the wavefront in the even slot leads, in odd slices the other one. At any moment the two hold opposite priorities,
// each leads half of the time, both advance at the same average rate: 1.68 / 1.71 ms, kernel 1.945 -> 1.80 ms. (Slices shorter
// than a few iterations do not work -- the wavefronts sample the clock at different moments and compute the same priority half
// of the time --, and a scheme keyed to the iteration count has no restoring force: one iteration apart both compute the same
// priority. A/B on the bench workload, stamped builds: off 2.03 ms, slices of 41 / 164 / 655 us 1.855 / 1.845 / 1.874 ms, 164 us
// re-evaluated only every eighth iteration 1.90 ms.) Re-evaluated every iteration where an iteration is long (WORK = sweep steps
// x DPP rows per instance >= 32: ~100 cycles of ~18,000), every 2nd / 4th / 8th for shorter ones; harmless where a wavefront has its
// SIMD to itself.
// ------------------------------------------------------------------------------------------------
#ifndef TINY_PRIO_SHIFT
#define TINY_PRIO_SHIFT 14
#endif
__device__ __forceinline__ int simd_slot_id() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | 4); }  // HW_REG_HW_ID bits 3:0: the wave's slot on its SIMD
template <int WORK>
__device__ __forceinline__ void fair_share_priority(int it0, int simd_slot) {
#ifdef TINY_PRIO_OFF  // (A/B builds, tools/clock_check.py)
    return;
#endif
#ifdef TINY_PRIO_EVERY
    constexpr int EVERY = TINY_PRIO_EVERY;
#else
    constexpr int EVERY = WORK >= 32 ? 1 : WORK >= 16 ? 2 : WORK >= 8 ? 4 : 8;
#endif
    if ((it0 & (EVERY - 1)) != 0) return;  // (it0: wave-uniform iteration counter)
#ifndef TINY_PRIO_HI
#define TINY_PRIO_HI 3
#endif
    const unsigned slice = (unsigned)(__builtin_amdgcn_s_memrealtime() >> TINY_PRIO_SHIFT);
    if ((slice ^ (unsigned)simd_slot) & 1u) __builtin_amdgcn_s_setprio(TINY_PRIO_HI);
    else __builtin_amdgcn_s_setprio(0);
}

// Sum over the W lanes of a group, the same bit pattern in every lane (a symmetric butterfly: both partners of an exchange add
// the same two numbers): four DPP exchanges inside a 16-lane row (64-bit DPP knows row_newbcast only, so the halves move
// separately), then one / two cross-row exchanges.
template <int CTRL>
__device__ __forceinline__ double dpp_exchange(double v) {
    // (compiler builtins only: this header is also compiled by hiprtc, whose built-in headers lack __double2loint and friends)
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)bits, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(bits >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
template <int W>
__device__ __forceinline__ double group_sum(double v) {
    static_assert(W == 16 || W == 32 || W == 64, "groups of 16, 32 or 64 lanes");
    v += dpp_exchange<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_exchange<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_exchange<0x141>(v);  // row_half_mirror: the other quad of the half row
    v += dpp_exchange<0x140>(v);  // row_mirror: the other half of the row
    if constexpr (W >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (W == 64) v += __shfl_xor(v, 32, 64);
    return v;
}

template <int W>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int m = 1; m < W; m <<= 1) v = fmax(v, __shfl_xor(v, m, W));
    return v;
}

}  // namespace tinympc
