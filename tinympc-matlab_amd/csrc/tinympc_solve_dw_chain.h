// tinympc_solve_dw_chain.h -- X-macro body: the sweep-step asm blocks of the WIDE layout D (32 lanes per instance) for ONE
// (nx, nu) pair with 16 < nx + nu <= 32. Include with DW_NX and DW_NU defined; defines DWStep<DW_NX, DW_NU>.
//
// The operand vector of a step, [x_i; d_i] or [p_{i+1}; r_i], is spread over the two DPP rows of the instance. It is
// replicated across them by the caller (v_permlane16_swap on two copies: `e` = the lower row's 16 entries in both rows, `o` =
// the upper row's), after which the mat-vec is the fused chain of tinympc_solve_d_chain.h with 16 columns per copy:
//     a += m[k] * e(row_newbcast:k)   k < 16        a += m[16 + k] * o(row_newbcast:k)   16 + k < nx + nu
// An asm statement takes at most 30 operands, so a step is two `asm volatile` blocks (columns 0-15 | columns 16.. + the
// row-local instructions; the wait that retires the caller's LDS reads follows the block: lds_reads_landed). Hazards: `e` and `o` are written by the swaps (VALU) right before
// the first block, which therefore opens with `s_nop 1`; tools/isa_lint.py checks the generated code.
#if !defined(DW_NX) || !defined(DW_NU)
#error "define DW_NX and DW_NU before including tinympc_solve_dw_chain.h"
#endif
#if DW_NX < 1 || DW_NU < 1 || DW_NX + DW_NU <= 16 || DW_NX + DW_NU > 32
#error "wide layout D: 16 < nx + nu <= 32"
#endif
#define DW_NXU (DW_NX + DW_NU)
// (run-time specialisations cannot be linted where they are compiled: there the second block of a step is guarded too,
// see tinympc_solve_d_chain.h)
#ifdef TINY_JIT
#define DW_HAZ "s_nop 1\n\t"
#else
#define DW_HAZ ""
#endif
// (every 16-column block of the chain starts on an 8-byte boundary: tinympc_solve_d_chain.h)
#ifdef TINY_CHAIN_NOALIGN
#define TINY_CHAIN_AL ""
#else
#define TINY_CHAIN_AL ".p2align 3\n\t"
#endif
#define DW_FE_(i) "v_fmac_f64_dpp %[a], %[e], %[m" #i "] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"
#define DW_FO_(i, b) "v_fmac_f64_dpp %[a], %[o], %[m" #i "] row_newbcast:" #b " row_mask:0xf bank_mask:0xf\n\t"
#define DW_LO TINY_CHAIN_AL DW_FE_(0) DW_FE_(1) DW_FE_(2) DW_FE_(3) DW_FE_(4) DW_FE_(5) DW_FE_(6) DW_FE_(7) DW_FE_(8) DW_FE_(9) DW_FE_(10) DW_FE_(11) DW_FE_(12) DW_FE_(13) DW_FE_(14) DW_FE_(15)
#if DW_NXU > 16
#define DW_H16 DW_FO_(16, 0)
#else
#define DW_H16 ""
#endif
#if DW_NXU > 17
#define DW_H17 DW_FO_(17, 1)
#else
#define DW_H17 ""
#endif
#if DW_NXU > 18
#define DW_H18 DW_FO_(18, 2)
#else
#define DW_H18 ""
#endif
#if DW_NXU > 19
#define DW_H19 DW_FO_(19, 3)
#else
#define DW_H19 ""
#endif
#if DW_NXU > 20
#define DW_H20 DW_FO_(20, 4)
#else
#define DW_H20 ""
#endif
#if DW_NXU > 21
#define DW_H21 DW_FO_(21, 5)
#else
#define DW_H21 ""
#endif
#if DW_NXU > 22
#define DW_H22 DW_FO_(22, 6)
#else
#define DW_H22 ""
#endif
#if DW_NXU > 23
#define DW_H23 DW_FO_(23, 7)
#else
#define DW_H23 ""
#endif
#if DW_NXU > 24
#define DW_H24 DW_FO_(24, 8)
#else
#define DW_H24 ""
#endif
#if DW_NXU > 25
#define DW_H25 DW_FO_(25, 9)
#else
#define DW_H25 ""
#endif
#if DW_NXU > 26
#define DW_H26 DW_FO_(26, 10)
#else
#define DW_H26 ""
#endif
#if DW_NXU > 27
#define DW_H27 DW_FO_(27, 11)
#else
#define DW_H27 ""
#endif
#if DW_NXU > 28
#define DW_H28 DW_FO_(28, 12)
#else
#define DW_H28 ""
#endif
#if DW_NXU > 29
#define DW_H29 DW_FO_(29, 13)
#else
#define DW_H29 ""
#endif
#if DW_NXU > 30
#define DW_H30 DW_FO_(30, 14)
#else
#define DW_H30 ""
#endif
#if DW_NXU > 31
#define DW_H31 DW_FO_(31, 15)
#else
#define DW_H31 ""
#endif
#define DW_HI TINY_CHAIN_AL DW_H16 DW_H17 DW_H18 DW_H19 DW_H20 DW_H21 DW_H22 DW_H23 DW_H24 DW_H25 DW_H26 DW_H27 DW_H28 DW_H29 DW_H30 DW_H31
#define DW_MLO [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [m4] "v"(m[4]), [m5] "v"(m[5]), [m6] "v"(m[6]), [m7] "v"(m[7]), [m8] "v"(m[8]), [m9] "v"(m[9]), [m10] "v"(m[10]), [m11] "v"(m[11]), [m12] "v"(m[12]), [m13] "v"(m[13]), [m14] "v"(m[14]), [m15] "v"(m[15])
#define DW_MHI [m16] "v"(m[16]), [m17] "v"(m[17]), [m18] "v"(m[18]), [m19] "v"(m[19]), [m20] "v"(m[20]), [m21] "v"(m[21]), [m22] "v"(m[22]), [m23] "v"(m[23]), [m24] "v"(m[24]), [m25] "v"(m[25]), [m26] "v"(m[26]), [m27] "v"(m[27]), [m28] "v"(m[28]), [m29] "v"(m[29]), [m30] "v"(m[30]), [m31] "v"(m[31])
// S1 + D1 + R1 for the element the chain just produced (admm.cpp:45-58, 67-68, 93-96); g is updated in place.
#define DW_PROJECT                                 \
    "v_add_f64 %[s], %[a], %[g]\n\t"               \
    "v_max_f64 %[sn], %[lo], %[s]\n\t"             \
    "v_min_f64 %[sn], %[hi], %[sn]\n\t"            \
    "v_add_f64 %[g], %[s], -%[sn]\n\t"             \
    "v_add_f64 %[t], %[a], -%[sn]\n\t"             \
    "v_max_f64 %[pri], %[pri], |%[t]|\n\t"         \
    "v_add_f64 %[t], %[v], -%[sn]\n\t"             \
    "v_max_f64 %[dua], %[dua], |%[t]|\n\t"

namespace tinympc {

template <>
struct DWStep<DW_NX, DW_NU> {
    // columns 0..15: a = start + sum_k m[k] * e_k   (forward: start = cf; backward: the accumulator's start value)
    static __device__ __forceinline__ double lo_fwd(double e, const double (&m)[32], double cf) {
        double a;
        asm volatile("s_nop 1\n\tv_mov_b64 %[a], %[cf]\n\t" DW_LO : [a] "=&v"(a) : [e] "v"(e), [cf] "v"(cf), DW_MLO);
        return a;
    }
    static __device__ __forceinline__ void lo_bwd(double &a, double e, const double (&m)[32]) {
        asm volatile("s_nop 1\n\t" DW_LO : [a] "+v"(a) : [e] "v"(e), DW_MLO);
    }
    // columns 16..: forward, slack in a REGISTER (vold in, vnew out, in place)
    static __device__ __forceinline__ void hi_fwd_reg(double &a, double o, const double (&m)[32], double lo, double hi, double &g, double &v,
                                                      double &pri, double &dua) {
        double s, t, sn;
        asm volatile(DW_HAZ DW_HI DW_PROJECT "v_mov_b64 %[v], %[sn]\n\t"
                     : [a] "+v"(a), [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(sn), [g] "+v"(g), [v] "+v"(v), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [o] "v"(o), [lo] "v"(lo), [hi] "v"(hi), DW_MHI);
        lds_reads_landed();
    }
    // ... slack in LDS: vold comes in, vnew goes out (the caller loads / stores them)
    static __device__ __forceinline__ void hi_fwd_lds(double &a, double o, const double (&m)[32], double lo, double hi, double &g, double v,
                                                      double &vnew, double &pri, double &dua) {
        double s, t;
        asm volatile(DW_HAZ DW_HI DW_PROJECT
                     : [a] "+v"(a), [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(vnew), [g] "+v"(g), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [o] "v"(o), [lo] "v"(lo), [hi] "v"(hi), [v] "v"(v), DW_MHI);
        lds_reads_landed();
    }
    // columns 16.. going backward + the tail of tinympc_solve_d_chain.h (an: accumulator start of the next step, rn: its
    // input-row operand):  t = v2 - g2 ;  an = rhom * t + lrmc ;  rn = nrho * t + lr
    static __device__ __forceinline__ void hi_bwd(double &a, double o, const double (&m)[32], double v2, double g2, double rhom, double lrmc,
                                                  double nrho, double lr, double &an, double &rn) {
        double t;
        asm volatile(DW_HAZ DW_HI
                     "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                     "v_fma_f64 %[an], %[rhom], %[t], %[lrmc]\n\t"
                     "v_fma_f64 %[rn], %[nrho], %[t], %[lr]\n\t"
                     : [a] "+v"(a), [an] "=&v"(an), [rn] "=&v"(rn), [t] "=&v"(t)
                     : [o] "v"(o), [v2] "v"(v2), [g2] "v"(g2), [rhom] "v"(rhom), [lrmc] "v"(lrmc), [nrho] "s"(nrho), [lr] "v"(lr), DW_MHI);
        lds_reads_landed();
    }
    static __device__ __forceinline__ void hi_bwd_last(double &a, double o, const double (&m)[32]) {
        asm volatile(DW_HAZ DW_HI : [a] "+v"(a) : [o] "v"(o), DW_MHI); lds_reads_landed();
    }
};

}  // namespace tinympc

#undef DW_FE_
#undef DW_FO_
#undef TINY_CHAIN_AL
#undef DW_LO
#undef DW_HI
#undef DW_MLO
#undef DW_MHI
#undef DW_PROJECT
#undef DW_NXU
#undef DW_H16
#undef DW_H17
#undef DW_H18
#undef DW_H19
#undef DW_H20
#undef DW_H21
#undef DW_H22
#undef DW_H23
#undef DW_H24
#undef DW_H25
#undef DW_H26
#undef DW_H27
#undef DW_H28
#undef DW_H29
#undef DW_H30
#undef DW_H31
#undef DW_NX
#undef DW_NU
