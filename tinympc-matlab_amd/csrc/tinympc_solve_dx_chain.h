// tinympc_solve_dx_chain.h -- X-macro body: the sweep-step asm blocks of the 64-LANE layout D (one instance per wavefront) for
// ONE (nx, nu) pair with 32 < nx + nu <= 64. Include with DX_NX and DX_NU defined; defines DXStep<DX_NX, DX_NU>.
//
// As tinympc_solve_dw_chain.h, with the operand vector replicated across the FOUR DPP rows of the instance (caller:
// v_permlane32_swap, then v_permlane16_swap on both results -> r0..r3 = row j's 16 entries in every row) and the chain in
// four blocks of 16 columns: a += m[16 j + k] * rj(row_newbcast:k). Every block opens with `s_nop 1`: the scheduler places the
// swap that produces a block's operand right in front of it (the build's ISA lint caught exactly that in the first version).
#if !defined(DX_NX) || !defined(DX_NU)
#error "define DX_NX and DX_NU before including tinympc_solve_dx_chain.h"
#endif
#if DX_NX < 1 || DX_NU < 1 || DX_NX + DX_NU <= 32 || DX_NX + DX_NU > 64
#error "64-lane layout D: 32 < nx + nu <= 64"
#endif
#define DX_NXU (DX_NX + DX_NU)
// (every 16-column block of the chain starts on an 8-byte boundary: tinympc_solve_d_chain.h)
#ifdef TINY_CHAIN_NOALIGN
#define TINY_CHAIN_AL ""
#else
#define TINY_CHAIN_AL ".p2align 3\n\t"
#endif
#define DX_F_(i, b) "v_fmac_f64_dpp %[a], %[w], %[m" #i "] row_newbcast:" #b " row_mask:0xf bank_mask:0xf\n\t"
#define DX_C0 DX_F_(0, 0)
#define DX_C1 DX_F_(1, 1)
#define DX_C2 DX_F_(2, 2)
#define DX_C3 DX_F_(3, 3)
#define DX_C4 DX_F_(4, 4)
#define DX_C5 DX_F_(5, 5)
#define DX_C6 DX_F_(6, 6)
#define DX_C7 DX_F_(7, 7)
#define DX_C8 DX_F_(8, 8)
#define DX_C9 DX_F_(9, 9)
#define DX_C10 DX_F_(10, 10)
#define DX_C11 DX_F_(11, 11)
#define DX_C12 DX_F_(12, 12)
#define DX_C13 DX_F_(13, 13)
#define DX_C14 DX_F_(14, 14)
#define DX_C15 DX_F_(15, 15)
#define DX_C16 DX_F_(16, 0)
#define DX_C17 DX_F_(17, 1)
#define DX_C18 DX_F_(18, 2)
#define DX_C19 DX_F_(19, 3)
#define DX_C20 DX_F_(20, 4)
#define DX_C21 DX_F_(21, 5)
#define DX_C22 DX_F_(22, 6)
#define DX_C23 DX_F_(23, 7)
#define DX_C24 DX_F_(24, 8)
#define DX_C25 DX_F_(25, 9)
#define DX_C26 DX_F_(26, 10)
#define DX_C27 DX_F_(27, 11)
#define DX_C28 DX_F_(28, 12)
#define DX_C29 DX_F_(29, 13)
#define DX_C30 DX_F_(30, 14)
#define DX_C31 DX_F_(31, 15)
#if DX_NXU > 32
#define DX_C32 DX_F_(32, 0)
#else
#define DX_C32 ""
#endif
#if DX_NXU > 33
#define DX_C33 DX_F_(33, 1)
#else
#define DX_C33 ""
#endif
#if DX_NXU > 34
#define DX_C34 DX_F_(34, 2)
#else
#define DX_C34 ""
#endif
#if DX_NXU > 35
#define DX_C35 DX_F_(35, 3)
#else
#define DX_C35 ""
#endif
#if DX_NXU > 36
#define DX_C36 DX_F_(36, 4)
#else
#define DX_C36 ""
#endif
#if DX_NXU > 37
#define DX_C37 DX_F_(37, 5)
#else
#define DX_C37 ""
#endif
#if DX_NXU > 38
#define DX_C38 DX_F_(38, 6)
#else
#define DX_C38 ""
#endif
#if DX_NXU > 39
#define DX_C39 DX_F_(39, 7)
#else
#define DX_C39 ""
#endif
#if DX_NXU > 40
#define DX_C40 DX_F_(40, 8)
#else
#define DX_C40 ""
#endif
#if DX_NXU > 41
#define DX_C41 DX_F_(41, 9)
#else
#define DX_C41 ""
#endif
#if DX_NXU > 42
#define DX_C42 DX_F_(42, 10)
#else
#define DX_C42 ""
#endif
#if DX_NXU > 43
#define DX_C43 DX_F_(43, 11)
#else
#define DX_C43 ""
#endif
#if DX_NXU > 44
#define DX_C44 DX_F_(44, 12)
#else
#define DX_C44 ""
#endif
#if DX_NXU > 45
#define DX_C45 DX_F_(45, 13)
#else
#define DX_C45 ""
#endif
#if DX_NXU > 46
#define DX_C46 DX_F_(46, 14)
#else
#define DX_C46 ""
#endif
#if DX_NXU > 47
#define DX_C47 DX_F_(47, 15)
#else
#define DX_C47 ""
#endif
#if DX_NXU > 48
#define DX_C48 DX_F_(48, 0)
#else
#define DX_C48 ""
#endif
#if DX_NXU > 49
#define DX_C49 DX_F_(49, 1)
#else
#define DX_C49 ""
#endif
#if DX_NXU > 50
#define DX_C50 DX_F_(50, 2)
#else
#define DX_C50 ""
#endif
#if DX_NXU > 51
#define DX_C51 DX_F_(51, 3)
#else
#define DX_C51 ""
#endif
#if DX_NXU > 52
#define DX_C52 DX_F_(52, 4)
#else
#define DX_C52 ""
#endif
#if DX_NXU > 53
#define DX_C53 DX_F_(53, 5)
#else
#define DX_C53 ""
#endif
#if DX_NXU > 54
#define DX_C54 DX_F_(54, 6)
#else
#define DX_C54 ""
#endif
#if DX_NXU > 55
#define DX_C55 DX_F_(55, 7)
#else
#define DX_C55 ""
#endif
#if DX_NXU > 56
#define DX_C56 DX_F_(56, 8)
#else
#define DX_C56 ""
#endif
#if DX_NXU > 57
#define DX_C57 DX_F_(57, 9)
#else
#define DX_C57 ""
#endif
#if DX_NXU > 58
#define DX_C58 DX_F_(58, 10)
#else
#define DX_C58 ""
#endif
#if DX_NXU > 59
#define DX_C59 DX_F_(59, 11)
#else
#define DX_C59 ""
#endif
#if DX_NXU > 60
#define DX_C60 DX_F_(60, 12)
#else
#define DX_C60 ""
#endif
#if DX_NXU > 61
#define DX_C61 DX_F_(61, 13)
#else
#define DX_C61 ""
#endif
#if DX_NXU > 62
#define DX_C62 DX_F_(62, 14)
#else
#define DX_C62 ""
#endif
#if DX_NXU > 63
#define DX_C63 DX_F_(63, 15)
#else
#define DX_C63 ""
#endif
#define DX_Q0 TINY_CHAIN_AL DX_C0 DX_C1 DX_C2 DX_C3 DX_C4 DX_C5 DX_C6 DX_C7 DX_C8 DX_C9 DX_C10 DX_C11 DX_C12 DX_C13 DX_C14 DX_C15
#define DX_M0 [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [m4] "v"(m[4]), [m5] "v"(m[5]), [m6] "v"(m[6]), [m7] "v"(m[7]), [m8] "v"(m[8]), [m9] "v"(m[9]), [m10] "v"(m[10]), [m11] "v"(m[11]), [m12] "v"(m[12]), [m13] "v"(m[13]), [m14] "v"(m[14]), [m15] "v"(m[15])
#define DX_Q1 TINY_CHAIN_AL DX_C16 DX_C17 DX_C18 DX_C19 DX_C20 DX_C21 DX_C22 DX_C23 DX_C24 DX_C25 DX_C26 DX_C27 DX_C28 DX_C29 DX_C30 DX_C31
#define DX_M1 [m16] "v"(m[16]), [m17] "v"(m[17]), [m18] "v"(m[18]), [m19] "v"(m[19]), [m20] "v"(m[20]), [m21] "v"(m[21]), [m22] "v"(m[22]), [m23] "v"(m[23]), [m24] "v"(m[24]), [m25] "v"(m[25]), [m26] "v"(m[26]), [m27] "v"(m[27]), [m28] "v"(m[28]), [m29] "v"(m[29]), [m30] "v"(m[30]), [m31] "v"(m[31])
#define DX_Q2 TINY_CHAIN_AL DX_C32 DX_C33 DX_C34 DX_C35 DX_C36 DX_C37 DX_C38 DX_C39 DX_C40 DX_C41 DX_C42 DX_C43 DX_C44 DX_C45 DX_C46 DX_C47
#define DX_M2 [m32] "v"(m[32]), [m33] "v"(m[33]), [m34] "v"(m[34]), [m35] "v"(m[35]), [m36] "v"(m[36]), [m37] "v"(m[37]), [m38] "v"(m[38]), [m39] "v"(m[39]), [m40] "v"(m[40]), [m41] "v"(m[41]), [m42] "v"(m[42]), [m43] "v"(m[43]), [m44] "v"(m[44]), [m45] "v"(m[45]), [m46] "v"(m[46]), [m47] "v"(m[47])
#define DX_Q3 TINY_CHAIN_AL DX_C48 DX_C49 DX_C50 DX_C51 DX_C52 DX_C53 DX_C54 DX_C55 DX_C56 DX_C57 DX_C58 DX_C59 DX_C60 DX_C61 DX_C62 DX_C63
#define DX_M3 [m48] "v"(m[48]), [m49] "v"(m[49]), [m50] "v"(m[50]), [m51] "v"(m[51]), [m52] "v"(m[52]), [m53] "v"(m[53]), [m54] "v"(m[54]), [m55] "v"(m[55]), [m56] "v"(m[56]), [m57] "v"(m[57]), [m58] "v"(m[58]), [m59] "v"(m[59]), [m60] "v"(m[60]), [m61] "v"(m[61]), [m62] "v"(m[62]), [m63] "v"(m[63])
#define DX_PROJECT                                 \
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
struct DXStep<DX_NX, DX_NU> {
    // block 0 (columns 0..15), forward: a = cf + ...; `w` was just written by the swaps -> s_nop 1
    static __device__ __forceinline__ double q0_fwd(double w, const double (&m)[64], double cf) {
        double a;
        asm volatile("s_nop 1\n\tv_mov_b64 %[a], %[cf]\n\t" DX_Q0 : [a] "=&v"(a) : [w] "v"(w), [cf] "v"(cf), DX_M0);
        return a;
    }
    static __device__ __forceinline__ void q0_bwd(double &a, double w, const double (&m)[64]) {
        asm volatile("s_nop 1\n\t" DX_Q0 : [a] "+v"(a) : [w] "v"(w), DX_M0);
    }
    // blocks 1 and 2 (columns 16..31, 32..47)
    static __device__ __forceinline__ void q1(double &a, double w, const double (&m)[64]) { asm volatile("s_nop 1\n\t" DX_Q1 : [a] "+v"(a) : [w] "v"(w), DX_M1); }
    static __device__ __forceinline__ void q2(double &a, double w, const double (&m)[64]) { asm volatile("s_nop 1\n\t" DX_Q2 : [a] "+v"(a) : [w] "v"(w), DX_M2); }
    // block 3 (columns 48..) + the row-local instructions + the wait; forward, slack in a register / in LDS
    static __device__ __forceinline__ void q3_fwd_reg(double &a, double w, const double (&m)[64], double lo, double hi, double &g, double &v,
                                                      double &pri, double &dua) {
        double s, t, sn;
        asm volatile("s_nop 1\n\t" DX_Q3 DX_PROJECT "v_mov_b64 %[v], %[sn]\n\t"
                     : [a] "+v"(a), [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(sn), [g] "+v"(g), [v] "+v"(v), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [w] "v"(w), [lo] "v"(lo), [hi] "v"(hi), DX_M3);
        lds_reads_landed();
    }
    static __device__ __forceinline__ void q3_fwd_lds(double &a, double w, const double (&m)[64], double lo, double hi, double &g, double v,
                                                      double &vnew, double &pri, double &dua) {
        double s, t;
        asm volatile("s_nop 1\n\t" DX_Q3 DX_PROJECT
                     : [a] "+v"(a), [s] "=&v"(s), [t] "=&v"(t), [sn] "=&v"(vnew), [g] "+v"(g), [pri] "+v"(pri), [dua] "+v"(dua)
                     : [w] "v"(w), [lo] "v"(lo), [hi] "v"(hi), [v] "v"(v), DX_M3);
        lds_reads_landed();
    }
    // block 3 going backward + the tail (see tinympc_solve_d_chain.h)
    static __device__ __forceinline__ void q3_bwd(double &a, double w, const double (&m)[64], double v2, double g2, double rhom, double lrmc,
                                                  double nrho, double lr, double &an, double &rn) {
        double t;
        asm volatile("s_nop 1\n\t" DX_Q3
                     "v_add_f64 %[t], %[v2], -%[g2]\n\t"
                     "v_fma_f64 %[an], %[rhom], %[t], %[lrmc]\n\t"
                     "v_fma_f64 %[rn], %[nrho], %[t], %[lr]\n\t"
                     : [a] "+v"(a), [an] "=&v"(an), [rn] "=&v"(rn), [t] "=&v"(t)
                     : [w] "v"(w), [v2] "v"(v2), [g2] "v"(g2), [rhom] "v"(rhom), [lrmc] "v"(lrmc), [nrho] "s"(nrho), [lr] "v"(lr), DX_M3);
        lds_reads_landed();
    }
    static __device__ __forceinline__ void q3_bwd_last(double &a, double w, const double (&m)[64]) {
        asm volatile("s_nop 1\n\t" DX_Q3 : [a] "+v"(a) : [w] "v"(w), DX_M3); lds_reads_landed();
    }
};

}  // namespace tinympc

#undef DX_F_
#undef DX_PROJECT
#undef DX_NXU
#undef DX_C0
#undef DX_C1
#undef DX_C2
#undef DX_C3
#undef DX_C4
#undef DX_C5
#undef DX_C6
#undef DX_C7
#undef DX_C8
#undef DX_C9
#undef DX_C10
#undef DX_C11
#undef DX_C12
#undef DX_C13
#undef DX_C14
#undef DX_C15
#undef DX_C16
#undef DX_C17
#undef DX_C18
#undef DX_C19
#undef DX_C20
#undef DX_C21
#undef DX_C22
#undef DX_C23
#undef DX_C24
#undef DX_C25
#undef DX_C26
#undef DX_C27
#undef DX_C28
#undef DX_C29
#undef DX_C30
#undef DX_C31
#undef DX_C32
#undef DX_C33
#undef DX_C34
#undef DX_C35
#undef DX_C36
#undef DX_C37
#undef DX_C38
#undef DX_C39
#undef DX_C40
#undef DX_C41
#undef DX_C42
#undef DX_C43
#undef DX_C44
#undef DX_C45
#undef DX_C46
#undef DX_C47
#undef DX_C48
#undef DX_C49
#undef DX_C50
#undef DX_C51
#undef DX_C52
#undef DX_C53
#undef DX_C54
#undef DX_C55
#undef DX_C56
#undef DX_C57
#undef DX_C58
#undef DX_C59
#undef DX_C60
#undef DX_C61
#undef DX_C62
#undef DX_C63
#undef TINY_CHAIN_AL
#undef DX_Q0
#undef DX_M0
#undef DX_Q1
#undef DX_M1
#undef DX_Q2
#undef DX_M2
#undef DX_Q3
#undef DX_M3
#undef DX_NX
#undef DX_NU
