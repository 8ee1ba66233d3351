// tinympc_sweep.h -- device helpers shared by the two solve kernels (layout A: tinympc_solve.hip,
// layout B: tinympc_solve_b.hip): the fused DPP mat-vec chain, the row-local projection block, and the
// group-wide max reduction.
#pragma once
#include <hip/hip_runtime.h>

namespace tinympc {

// ------------------------------------------------------------------------------------------------
// Generic-width helpers (W = 32, 64): broadcast lane K of every W-lane group to the whole group.
// ------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------
// W = 16: the mat-vec as ONE chain of fused VOP2+DPP instructions, acc += m_k * (w of lane k of the
// 16-lane DPP row = of the instance). hipcc lowers the builtin form to v_mov_b64_dpp + v_fma_f64 pairs
// (its DPP combiner does not fold 64-bit moves); the fused form halves the instruction count:
// 64 vs 89 ns per 16x16 step on MI355X, bit-identical (tools/microbench_matvec.hip). A single
// accumulator is fastest (11.35 vs 11.96 / 12.21 ms per launch for 1 / 2 / 4 partial sums): the wave is
// issue-bound, not latency-bound, so extra partial sums only add moves and adds.
// Hazard: a VALU-written VGPR read through DPP needs 2 wait states, which hipcc does not insert
// inside inline asm -> `s_nop 1` opens the chain (w was just produced by a v_cndmask).
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
        asm("s_nop 1\n\t" TINY_FM(0) TINY_FM(1) TINY_FM(2) TINY_FM(3) : [a] "+v"(a) : [w] "v"(w), TINY_M8);
        asm(TINY_FM(4) TINY_FM(5) TINY_FM(6) TINY_FM(7) : [a] "+v"(a) : [w] "v"(w), TINY_M8);
        if constexpr (KT >= 12) asm(TINY_FM(8) TINY_FM(9) TINY_FM(10) TINY_FM(11) : [a] "+v"(a) : [w] "v"(w), TINY_M12);
        if constexpr (KT == 16) asm(TINY_FM(12) TINY_FM(13) TINY_FM(14) TINY_FM(15) : [a] "+v"(a) : [w] "v"(w), TINY_M16);
        return a;
    } else {
        double acc[2] = {c, 0.0};
        matvec_accumulate<W, KT>(m, w, acc);
        return acc[0] + acc[1];
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

template <int W>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int m = 1; m < W; m <<= 1) v = fmax(v, __shfl_xor(v, m, W));
    return v;
}

}  // namespace tinympc
