// tinympc_solve_d_chain.h -- X-macro body: the sweep-step asm blocks of layout D for ONE (nx, nu) pair.
//
// Include with D_NX and D_NU defined (tinympc_solve_d.hip does, once per supported pair). Defines the explicit
// specialisation DStep<D_NX, D_NU>. The split of the fused DPP mat-vec chain into "columns fed by the state
// operand" (k < nx) and "columns fed by the input-row operand" (nx <= k < nx+nu) has to be spelled in the asm
// text itself, hence the preprocessor: column k reads %[x] or %[d] and columns >= nx+nu are not emitted at all
// (rocket landing, nx+nu = 9: 9 FMAs per step where the run-time-nx kernels issue 12).
//
// What bounds a wavefront (tools/microbench_fp64_occupancy.hip, gpurun_out/mb_occ_r02.txt): in straight-line code a
// DEPENDENT chain of v_fmac_f64_dpp issues every 4.1 cycles from a single wave -- the FP64 pipe's own rate (16 lanes
// per clock), so the mat-vec needs neither partial sums nor interleaving. What does stall a wave is waiting for LDS:
// hipcc sinks ds_reads next to their use and waits lgkmcnt(0) right behind them, exposing ~100+ cycles per step.
// Hence every block is `asm volatile` and is followed by a wait: the caller issues the NEXT step's LDS reads right before a
// block, they complete in the shadow of the block's ~25 FP64 instructions, and the wait behind the block retires them before
// any later code can touch them. Reads and wait are in the form the compiler TRACKS (tinympc_sweep.h: lds_read_issued_here,
// a volatile load, and lds_reads_landed, __builtin_amdgcn_s_waitcnt -- through round 5 both were asm text, invisible to the
// register allocator, which in one wide specialisation reused a register that was still being filled).
//
// DPP hazard (a VGPR written by VALU must not be read through DPP within 2 wait states; nothing guards it inside
// inline asm): `x` is the `a` the previous block's chain wrote; forward blocks end with >= 8 non-DPP instructions
// after that write, backward blocks with 3. tests/test_isa_hazards.py checks the generated code.
#if !defined(D_NX) || !defined(D_NU)
#error "define D_NX and D_NU before including tinympc_solve_d_chain.h"
#endif
#if D_NX < 1 || D_NU < 1 || D_NX + D_NU > 16
#error "layout D: 1 <= nx, 1 <= nu, nx + nu <= 16"
#endif

// Run-time specialisations (TINY_JIT; tinympc_jit.hip) are compiled by hiprtc ON THE GPU BOX, where nothing can lint the
// generated code: there every chain block opens with its own `s_nop 1`, whatever the compiler may have placed in front of it
// (a register copy or v_accvgpr_read of `x` / `d` on the 512-register plan). The compiled-in instantiations keep the bare
// blocks and are linted by the build (tools/isa_lint.py; the nop costs 1-2 % there).
// Every chain starts on an 8-byte boundary. Its instructions are 8 bytes each, and whether they sit on the 8-byte grid or
// straddle it decided, alone, between the two speeds every build of these kernels had shown (+-2.5 %, profiles/r03_dgroup_ab.txt):
// four bytes of s_nop in front of the iteration loop switch a fast build to the slow one and back; with the alignment both are
// fast (1.669 / 1.669 against 1.670 / 1.712 ms on the headline). At most one 4-byte s_nop per block.
#ifdef TINY_CHAIN_NOALIGN  // (experiments)
#define D_AL ""
#define D_MOV64 "v_mov_b64"
#else
#define D_AL ".p2align 3\n\t"
#ifdef TINY_CHAIN_MOV32  // (experiments)
#define D_MOV64 "v_mov_b64"
#else
#define D_MOV64 "v_mov_b64_e64"
#endif
#endif
#if defined(TINY_JIT) || defined(TINY_CHAIN_NOP)
#define D_HAZ "s_nop 1\n\t" D_AL
#else
#define D_HAZ D_AL
#endif
#define D_FM_(src, i) "v_fmac_f64_dpp %[a], " src ", %[m" #i "] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"
// column i of the chain: state operand, input operand, or nothing.
// (The preprocessor cannot index, so each column is resolved by its own #if ladder.)
#if 0 < D_NX
#define D_C0 D_FM_("%[x]", 0)
#else
#define D_C0 D_FM_("%[d]", 0)
#endif
#if 1 < D_NX
#define D_C1 D_FM_("%[x]", 1)
#elif 1 < D_NX + D_NU
#define D_C1 D_FM_("%[d]", 1)
#else
#define D_C1 ""
#endif
#if 2 < D_NX
#define D_C2 D_FM_("%[x]", 2)
#elif 2 < D_NX + D_NU
#define D_C2 D_FM_("%[d]", 2)
#else
#define D_C2 ""
#endif
#if 3 < D_NX
#define D_C3 D_FM_("%[x]", 3)
#elif 3 < D_NX + D_NU
#define D_C3 D_FM_("%[d]", 3)
#else
#define D_C3 ""
#endif
#if 4 < D_NX
#define D_C4 D_FM_("%[x]", 4)
#elif 4 < D_NX + D_NU
#define D_C4 D_FM_("%[d]", 4)
#else
#define D_C4 ""
#endif
#if 5 < D_NX
#define D_C5 D_FM_("%[x]", 5)
#elif 5 < D_NX + D_NU
#define D_C5 D_FM_("%[d]", 5)
#else
#define D_C5 ""
#endif
#if 6 < D_NX
#define D_C6 D_FM_("%[x]", 6)
#elif 6 < D_NX + D_NU
#define D_C6 D_FM_("%[d]", 6)
#else
#define D_C6 ""
#endif
#if 7 < D_NX
#define D_C7 D_FM_("%[x]", 7)
#elif 7 < D_NX + D_NU
#define D_C7 D_FM_("%[d]", 7)
#else
#define D_C7 ""
#endif
#if 8 < D_NX
#define D_C8 D_FM_("%[x]", 8)
#elif 8 < D_NX + D_NU
#define D_C8 D_FM_("%[d]", 8)
#else
#define D_C8 ""
#endif
#if 9 < D_NX
#define D_C9 D_FM_("%[x]", 9)
#elif 9 < D_NX + D_NU
#define D_C9 D_FM_("%[d]", 9)
#else
#define D_C9 ""
#endif
#if 10 < D_NX
#define D_C10 D_FM_("%[x]", 10)
#elif 10 < D_NX + D_NU
#define D_C10 D_FM_("%[d]", 10)
#else
#define D_C10 ""
#endif
#if 11 < D_NX
#define D_C11 D_FM_("%[x]", 11)
#elif 11 < D_NX + D_NU
#define D_C11 D_FM_("%[d]", 11)
#else
#define D_C11 ""
#endif
#if 12 < D_NX
#define D_C12 D_FM_("%[x]", 12)
#elif 12 < D_NX + D_NU
#define D_C12 D_FM_("%[d]", 12)
#else
#define D_C12 ""
#endif
#if 13 < D_NX
#define D_C13 D_FM_("%[x]", 13)
#elif 13 < D_NX + D_NU
#define D_C13 D_FM_("%[d]", 13)
#else
#define D_C13 ""
#endif
#if 14 < D_NX
#define D_C14 D_FM_("%[x]", 14)
#elif 14 < D_NX + D_NU
#define D_C14 D_FM_("%[d]", 14)
#else
#define D_C14 ""
#endif
#if 15 < D_NX
#define D_C15 D_FM_("%[x]", 15)
#elif 15 < D_NX + D_NU
#define D_C15 D_FM_("%[d]", 15)
#else
#define D_C15 ""
#endif

// the input columns only (layout F, round 4: e += T d, see DStep::acc_inputs)
#if 0 >= D_NX && 0 < D_NX + D_NU
#define D_I0 D_FM_("%[d]", 0)
#else
#define D_I0 ""
#endif
#if 1 >= D_NX && 1 < D_NX + D_NU
#define D_I1 D_FM_("%[d]", 1)
#else
#define D_I1 ""
#endif
#if 2 >= D_NX && 2 < D_NX + D_NU
#define D_I2 D_FM_("%[d]", 2)
#else
#define D_I2 ""
#endif
#if 3 >= D_NX && 3 < D_NX + D_NU
#define D_I3 D_FM_("%[d]", 3)
#else
#define D_I3 ""
#endif
#if 4 >= D_NX && 4 < D_NX + D_NU
#define D_I4 D_FM_("%[d]", 4)
#else
#define D_I4 ""
#endif
#if 5 >= D_NX && 5 < D_NX + D_NU
#define D_I5 D_FM_("%[d]", 5)
#else
#define D_I5 ""
#endif
#if 6 >= D_NX && 6 < D_NX + D_NU
#define D_I6 D_FM_("%[d]", 6)
#else
#define D_I6 ""
#endif
#if 7 >= D_NX && 7 < D_NX + D_NU
#define D_I7 D_FM_("%[d]", 7)
#else
#define D_I7 ""
#endif
#if 8 >= D_NX && 8 < D_NX + D_NU
#define D_I8 D_FM_("%[d]", 8)
#else
#define D_I8 ""
#endif
#if 9 >= D_NX && 9 < D_NX + D_NU
#define D_I9 D_FM_("%[d]", 9)
#else
#define D_I9 ""
#endif
#if 10 >= D_NX && 10 < D_NX + D_NU
#define D_I10 D_FM_("%[d]", 10)
#else
#define D_I10 ""
#endif
#if 11 >= D_NX && 11 < D_NX + D_NU
#define D_I11 D_FM_("%[d]", 11)
#else
#define D_I11 ""
#endif
#if 12 >= D_NX && 12 < D_NX + D_NU
#define D_I12 D_FM_("%[d]", 12)
#else
#define D_I12 ""
#endif
#if 13 >= D_NX && 13 < D_NX + D_NU
#define D_I13 D_FM_("%[d]", 13)
#else
#define D_I13 ""
#endif
#if 14 >= D_NX && 14 < D_NX + D_NU
#define D_I14 D_FM_("%[d]", 14)
#else
#define D_I14 ""
#endif
#if 15 >= D_NX && 15 < D_NX + D_NU
#define D_I15 D_FM_("%[d]", 15)
#else
#define D_I15 ""
#endif
#define D_CHAIN_IN D_I0 D_I1 D_I2 D_I3 D_I4 D_I5 D_I6 D_I7 D_I8 D_I9 D_I10 D_I11 D_I12 D_I13 D_I14 D_I15
#define D_CHAIN D_C0 D_C1 D_C2 D_C3 D_C4 D_C5 D_C6 D_C7 D_C8 D_C9 D_C10 D_C11 D_C12 D_C13 D_C14 D_C15
#define D_MOPS                                                                                                      \
    [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [m4] "v"(m[4]), [m5] "v"(m[5]), [m6] "v"(m[6]), \
        [m7] "v"(m[7]), [m8] "v"(m[8]), [m9] "v"(m[9]), [m10] "v"(m[10]), [m11] "v"(m[11]), [m12] "v"(m[12]),       \
        [m13] "v"(m[13]), [m14] "v"(m[14]), [m15] "v"(m[15])

// S1 + D1 + R1 for the element the chain just produced (admm.cpp:45-58, 67-68, 93-96); g is updated in place.
#define D_PROJECT                                  \
    "v_add_f64 %[s], %[a], %[g]\n\t"               \
    "v_max_f64 %[sn], %[lo], %[s]\n\t"             \
    "v_min_f64 %[sn], %[hi], %[sn]\n\t"            \
    "v_add_f64 %[g], %[s], -%[sn]\n\t"             \
    "v_add_f64 %[t], %[a], -%[sn]\n\t"             \
    "v_max_f64 %[pri], %[pri], |%[t]|\n\t"         \
    "v_add_f64 %[t], %[v], -%[sn]\n\t"             \
    "v_max_f64 %[dua], %[dua], |%[t]|\n\t"
// The same row-local block for the element of the PREVIOUS step (%[ap], its slot's g / v / bounds), woven into the chain of the
// current step one instruction per column (layout E, round 4): the block is a dependent sequence of its own, the chain another, and a
// wavefront issues in order -- back to back each instruction waits for its predecessor's result (~8 cycles for FP64), interleaved
// the two sequences fill each other's gaps. Columns the system does not have emit nothing; the block's instructions keep their order.
#define D_P1 "v_add_f64 %[s], %[ap], %[g]\n\t"
#define D_P2 "v_max_f64 %[sn], %[lo], %[s]\n\t"
#define D_P3 "v_min_f64 %[sn], %[hi], %[sn]\n\t"
#define D_P4 "v_add_f64 %[g], %[s], -%[sn]\n\t"
#define D_P5 "v_add_f64 %[t], %[ap], -%[sn]\n\t"
#define D_P6 "v_max_f64 %[pri], %[pri], |%[t]|\n\t"
#define D_P7 "v_add_f64 %[t], %[v], -%[sn]\n\t"
#define D_P8 "v_max_f64 %[dua], %[dua], |%[t]|\n\t"
#define D_CHAIN_WOVEN D_C0 D_P1 D_C1 D_P2 D_C2 D_P3 D_C3 D_P4 D_C4 D_P5 D_C5 D_P6 D_C6 D_P7 D_C7 D_P8 D_C8 D_C9 D_C10 D_C11 D_C12 D_C13 D_C14 D_C15
#define D_PROJECT_PREV D_P1 D_P2 D_P3 D_P4 D_P5 D_P6 D_P7 D_P8
// ... and the backward step's tail (the next steps' linear-cost terms, independent of this step's chain) woven in the same way
#define D_T1 "v_add_f64 %[t], %[v2], -%[g2]\n\t"
#define D_T2 "v_fma_f64 %[an], %[rhom], %[t], %[lrmc]\n\t"
#define D_T3 "v_fma_f64 %[rn], %[nrho], %[t], %[lr]\n\t"
#define D_CHAIN_BWD_WOVEN D_C0 D_T1 D_C1 D_C2 D_T2 D_C3 D_T3 D_C4 D_C5 D_C6 D_C7 D_C8 D_C9 D_C10 D_C11 D_C12 D_C13 D_C14 D_C15

namespace tinympc {

template <>
struct DStep<D_NX, D_NU> {
    // Forward step, slack kept in a REGISTER: a = cf + Mf * [x; d], then the row-local block; v (vold in, vnew out)
    // and g are updated in place so that no register rotates across the iteration loop's back edge.
    static __device__ __forceinline__ double fwd_reg(double x, double d, const double (&m)[16], double cf, double lo, double hi,
                                                     double &g, double &v, double &pri, double &dua) {
        double a, s, t, sn;
        asm volatile("v_mov_b64 %[a], %[cf]\n\t" D_HAZ D_CHAIN D_PROJECT "v_mov_b64 %[v], %[sn]\n\t"
                     : [a] "=&v"(a), [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(sn), [g] "+v"(g), [v] "+v"(v), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [x] "v"(x), [d] "v"(d), [cf] "v"(cf), [lo] "v"(lo), [hi] "v"(hi), D_MOPS);
        lds_reads_landed();
        return a;
    }
    // Forward step, slack kept in LDS: vold comes in, vnew goes out (the caller loads / stores them).
    static __device__ __forceinline__ double fwd_lds(double x, double d, const double (&m)[16], double cf, double lo, double hi,
                                                     double &g, double v, double &vnew, double &pri, double &dua) {
        double a, s, t;
        // (v_mov_b64 in its 8-byte encoding: with the 4-byte one this block and what the caller puts around it -- s_waitcnt, a
        // hazard s_nop, the slot's LDS write and the next step's two reads -- come to 4 bytes more than a multiple of 8, every
        // second LDS step's chain would straddle the 8-byte grid and D_AL would pad each of them with an s_nop)
        asm volatile(D_MOV64 " %[a], %[cf]\n\t" D_HAZ D_CHAIN D_PROJECT
                     : [a] "=&v"(a), [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(vnew), [g] "+v"(g), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [x] "v"(x), [d] "v"(d), [cf] "v"(cf), [lo] "v"(lo), [hi] "v"(hi), [v] "v"(v), D_MOPS);
        lds_reads_landed();
        return a;
    }
    // Forward step with the row-local block of the PREVIOUS step woven into its chain (layout E): returns a = cf + Mf * [x; d];
    // (ap, g, v, lo, hi) belong to the previous step's slot: ap its element (the `a` that step returned -- usually `x` itself, but
    // the caller decides), g and v are updated in place.
    static __device__ __forceinline__ double fwd_reg_woven(double x, double d, const double (&m)[16], double cf, double ap, double lo, double hi,
                                                           double &g, double &v, double &pri, double &dua) {
        double a, s, t, sn;
        asm volatile(D_MOV64 " %[a], %[cf]\n\t" D_HAZ D_CHAIN_WOVEN D_MOV64 " %[v], %[sn]\n\t"
                     : [a] "=&v"(a), [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(sn), [g] "+v"(g), [v] "+v"(v), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [x] "v"(x), [d] "v"(d), [cf] "v"(cf), [ap] "v"(ap), [lo] "v"(lo), [hi] "v"(hi), D_MOPS);
        lds_reads_landed();
        return a;
    }
    // ... and the row-local block alone, for the last step of a sweep
    static __device__ __forceinline__ void project_prev(double ap, double lo, double hi, double &g, double &v, double &pri, double &dua) {
        double s, t, sn;
        asm volatile(D_PROJECT_PREV "v_mov_b64 %[v], %[sn]"
                     : [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(sn), [g] "+v"(g), [v] "+v"(v), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [ap] "v"(ap), [lo] "v"(lo), [hi] "v"(hi));
    }
    // The bare sweep step a = c + M * [x; d] -- no row-local block behind it (layout E: pass 1 of a chunk and the carry
    // recurrences, tinympc_solve_e.hip). Back to back, `x` is the `a` the previous block's last FMA wrote: the block's own
    // s_waitcnt, the accumulator's start and an `s_nop 1` are the wait states in front of the first DPP read.
    static __device__ __forceinline__ double fwd_plain(double x, double d, const double (&m)[16], double c) {
        double a;
        asm volatile("v_mov_b64 %[a], %[c]\n\t"
                     "s_nop 1\n\t" D_AL D_CHAIN
                     : [a] "=&v"(a)
                     : [x] "v"(x), [d] "v"(d), [c] "v"(c), D_MOPS);
        lds_reads_landed();
        return a;
    }
    // Backward step for slot s: a (in: accumulator start = q_s + cb on state lanes, cb on input lanes; out: p_s | d_s)
    // += Mb * [x; d] with x = p_{s+1} (state lanes) and d = r_s (input lanes). The tail prepares, from slot s-2's
    // (v2, g2), the accumulator start of step s-1 (`an`) and the input-row operand of step s-2 (`rn`)   (L1, admm.cpp:77-80):
    //     t = v2 - g2 ;  an = rhom * t + lrmc ;  rn = nrho * t + lr
    // with rhom = -rho / 0 and lrmc = lr + cb / cb on state / input lanes.
    static __device__ __forceinline__ void bwd(double &a, double x, double d, const double (&m)[16], double v2, double g2,
                                               double rhom, double lrmc, double nrho, double lr, double &an, double &rn) {
        double t;
        asm volatile(D_HAZ D_CHAIN
                     "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                     "v_fma_f64 %[an], %[rhom], %[t], %[lrmc]\n\t"
                     "v_fma_f64 %[rn], %[nrho], %[t], %[lr]\n\t"
                     : [a] "+v"(a), [an] "=&v"(an), [rn] "=&v"(rn), [t] "=&v"(t)
                     : [x] "v"(x), [d] "v"(d), [v2] "v"(v2), [g2] "v"(g2), [rhom] "v"(rhom), [lrmc] "v"(lrmc), [nrho] "s"(nrho), [lr] "v"(lr), D_MOPS);
        lds_reads_landed();
    }
    // ... the same step with its tail woven into the chain (layout E)
    static __device__ __forceinline__ void bwd_woven(double &a, double x, double d, const double (&m)[16], double v2, double g2,
                                                     double rhom, double lrmc, double nrho, double lr, double &an, double &rn) {
        double t;
        asm volatile(D_HAZ D_CHAIN_BWD_WOVEN
                     : [a] "+v"(a), [an] "=&v"(an), [rn] "=&v"(rn), [t] "=&v"(t)
                     : [x] "v"(x), [d] "v"(d), [v2] "v"(v2), [g2] "v"(g2), [rhom] "v"(rhom), [lrmc] "v"(lrmc), [nrho] "s"(nrho), [lr] "v"(lr), D_MOPS);
        lds_reads_landed();
    }
    // ... the same with -rho as a vector operand (adaptive rho: rho is per instance)
    static __device__ __forceinline__ void bwd_v(double &a, double x, double d, const double (&m)[16], double v2, double g2,
                                                 double rhom, double lrmc, double nrho, double lr, double &an, double &rn) {
        double t;
        asm volatile(D_HAZ D_CHAIN
                     "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                     "v_fma_f64 %[an], %[rhom], %[t], %[lrmc]\n\t"
                     "v_fma_f64 %[rn], %[nrho], %[t], %[lr]\n\t"
                     : [a] "+v"(a), [an] "=&v"(an), [rn] "=&v"(rn), [t] "=&v"(t)
                     : [x] "v"(x), [d] "v"(d), [v2] "v"(v2), [g2] "v"(g2), [rhom] "v"(rhom), [lrmc] "v"(lrmc), [nrho] "v"(nrho), [lr] "v"(lr), D_MOPS);
        lds_reads_landed();
    }
    // a += T * d over the INPUT columns only (T_k in m[NX + k]; the other entries of m are not read): layout F accumulates a chunk's end
    // state from a zero incoming state, sum_s Phi^(S-1-s) (-B) d_s, while the backward sweep produces the d_s (tinympc_solve_f.hip).
    static __device__ __forceinline__ double acc_inputs(double a, double d, const double (&m)[16]) {
        // (its operand d is ALWAYS the result of the chain right in front of it: the two wait states of the DPP hazard are spelled out
        // here in every build, so that a compiled-in kernel's other chain blocks can stay bare)
        asm volatile("s_nop 1\n\t" D_AL D_CHAIN_IN : [a] "+v"(a) : [d] "v"(d), D_MOPS); lds_reads_landed();
        return a;
    }
    // Last backward step (slot 0): nothing left to prepare.
    static __device__ __forceinline__ void bwd_last(double &a, double x, double d, const double (&m)[16]) {
        asm volatile(D_HAZ D_CHAIN : [a] "+v"(a) : [x] "v"(x), [d] "v"(d), D_MOPS); lds_reads_landed();
    }
};

}  // namespace tinympc

#undef D_FM_
#undef D_HAZ
#undef D_AL
#undef D_MOV64
#undef D_C0
#undef D_C1
#undef D_C2
#undef D_C3
#undef D_C4
#undef D_C5
#undef D_C6
#undef D_C7
#undef D_C8
#undef D_C9
#undef D_C10
#undef D_C11
#undef D_C12
#undef D_C13
#undef D_C14
#undef D_C15
#undef D_CHAIN
#undef D_I0
#undef D_I1
#undef D_I2
#undef D_I3
#undef D_I4
#undef D_I5
#undef D_I6
#undef D_I7
#undef D_I8
#undef D_I9
#undef D_I10
#undef D_I11
#undef D_I12
#undef D_I13
#undef D_I14
#undef D_I15
#undef D_CHAIN_IN
#undef D_MOPS
#undef D_PROJECT
#undef D_P1
#undef D_P2
#undef D_P3
#undef D_P4
#undef D_P5
#undef D_P6
#undef D_P7
#undef D_P8
#undef D_CHAIN_WOVEN
#undef D_PROJECT_PREV
#undef D_T1
#undef D_T2
#undef D_T3
#undef D_CHAIN_BWD_WOVEN
#undef D_NX
#undef D_NU
