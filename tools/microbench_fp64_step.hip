// Dev microbenchmark for the forward step of the sweeps (16 x v_fmac_f64_dpp acc, x, m[k] row_newbcast:k, then the 8
// row-local FP64 instructions of S1 + D1 + R1), explicit registers, m[k] = v[2k:2k+1] as in the kernels:
//   1. does the issue rate depend on WHICH registers hold acc and x (VGPR banks)?
//   2. what does the dependent 8-instruction tail cost, and what if it is interleaved with the NEXT step's chain
//      (which only needs the tail's input, not its results)?
// Full grid (256 CUs x 1 or 2 workgroups of 4 wavefronts = 1 or 2 wavefronts per SIMD, the occupancy of layout D's headline).
// Prints cycles per VALU instruction and SIMD at the clock measured by s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define FM(A, X, M, K) "v_fmac_f64_dpp " A ", " X ", " M " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define CHAIN(A, X)                                                                                                      \
    FM(A, X, "v[0:1]", 0) FM(A, X, "v[2:3]", 1) FM(A, X, "v[4:5]", 2) FM(A, X, "v[6:7]", 3) FM(A, X, "v[8:9]", 4)        \
    FM(A, X, "v[10:11]", 5) FM(A, X, "v[12:13]", 6) FM(A, X, "v[14:15]", 7) FM(A, X, "v[16:17]", 8) FM(A, X, "v[18:19]", 9) \
    FM(A, X, "v[20:21]", 10) FM(A, X, "v[22:23]", 11) FM(A, X, "v[24:25]", 12) FM(A, X, "v[26:27]", 13)                  \
    FM(A, X, "v[28:29]", 14) FM(A, X, "v[30:31]", 15)
// the row-local block on scratch registers: s v[48:49], sn v[52:53], g v[50:51], t v[56:57], pri v[54:55], dua v[58:59], v v[60:61]
#define T1(A) "v_add_f64 v[48:49], " A ", v[50:51]\n\t"
#define T2(A) "v_max_f64 v[52:53], v[62:63], v[48:49]\n\t"
#define T3(A) "v_min_f64 v[52:53], v[46:47], v[52:53]\n\t"
#define T4(A) "v_add_f64 v[50:51], v[48:49], -v[52:53]\n\t"
#define T5(A) "v_add_f64 v[56:57], " A ", -v[52:53]\n\t"
#define T6(A) "v_max_f64 v[54:55], v[54:55], |v[56:57]|\n\t"
#define T7(A) "v_add_f64 v[56:57], v[60:61], -v[52:53]\n\t"
#define T8(A) "v_max_f64 v[58:59], v[58:59], |v[56:57]|\n\t"
#define TAIL(A) T1(A) T2(A) T3(A) T4(A) T5(A) T6(A) T7(A) T8(A)
// chain of the step that reads X (= the previous step's result) interleaved with the previous step's row-local block on X:
// two of its instructions first (X was just written by VALU: 2 wait states before the first DPP read), the rest between FMAs
#define MIXED(A, X)                                                                                                      \
    T1(X) T2(X) FM(A, X, "v[0:1]", 0) T3(X) FM(A, X, "v[2:3]", 1) T4(X) FM(A, X, "v[4:5]", 2) T5(X) FM(A, X, "v[6:7]", 3) \
    T6(X) FM(A, X, "v[8:9]", 4) T7(X) FM(A, X, "v[10:11]", 5) T8(X) FM(A, X, "v[12:13]", 6) FM(A, X, "v[14:15]", 7)      \
    FM(A, X, "v[16:17]", 8) FM(A, X, "v[18:19]", 9) FM(A, X, "v[20:21]", 10) FM(A, X, "v[22:23]", 11)                   \
    FM(A, X, "v[24:25]", 12) FM(A, X, "v[26:27]", 13) FM(A, X, "v[28:29]", 14) FM(A, X, "v[30:31]", 15)
#define CLOB                                                                                                                   \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", \
        "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", \
        "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", \
        "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#define R16(X) X X X X X X X X X X X X X X X X
#define Z(R) "v_mov_b32 v" #R ", 0\n\t"
#define Z8(R) Z(R##0) Z(R##1) Z(R##2) Z(R##3) Z(R##4) Z(R##5) Z(R##6) Z(R##7)
#define KERNEL(NAME, BODY)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *out, int iters) {                            \
        asm volatile(Z(0) Z(1) Z(2) Z(3) Z(4) Z(5) Z(6) Z(7) Z(8) Z(9) Z(10) Z(11) Z(12) Z(13) Z(14) Z(15) Z(16) \
                     Z(17) Z(18) Z(19) Z(20) Z(21) Z(22) Z(23) Z(24) Z(25) Z(26) Z(27) Z(28) Z(29) Z(30) Z(31)  \
                     Z(32) Z(33) Z(34) Z(35) Z(36) Z(37) Z(38) Z(39) Z(40) Z(41) Z(42) Z(43) Z(44) Z(45) Z(46)  \
                     Z(47) Z(48) Z(49) Z(50) Z(51) Z(52) Z(53) Z(54) Z(55) Z(56) Z(57) Z(58) Z(59) Z(60) Z(61)  \
                     Z(62) Z(63)::: CLOB);                                                                       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();       \
        for (int i = 0; i < iters; ++i) asm volatile(R16(BODY)::: CLOB); /* 16 x: straight-line code, the loop's taken branch amortised */                                              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();       \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                                               \
            out[0] = t1 - t0;                                                                                    \
            out[1] = r1 - r0;                                                                                    \
        }                                                                                                        \
    }
// two steps per loop trip: the result of one is the operand of the next, like xcur in the sweep.
// Banks of an (even-aligned) 64-bit pair: its first register modulo 4, i.e. 0 or 2 (m[k] alternates 0 / 2).
#define A0 "v[32:33]"
#define B0 "v[36:37]"
#define A2 "v[34:35]"
#define B2 "v[38:39]"
KERNEL(k_00, CHAIN(A0, B0) TAIL(A0) CHAIN(B0, A0) TAIL(B0))  // acc bank 0, x bank 0
KERNEL(k_02, CHAIN(A0, A2) TAIL(A0) CHAIN(A2, A0) TAIL(A2))  // banks 0 / 2
KERNEL(k_22, CHAIN(A2, B2) TAIL(A2) CHAIN(B2, A2) TAIL(B2))  // banks 2 / 2
KERNEL(k_chain, CHAIN(A0, A2) "s_nop 1\n\t" CHAIN(A2, A0) "s_nop 1\n\t")  // chains alone (the s_nop: DPP hazard)
KERNEL(k_mixed, MIXED(A0, A2) MIXED(A2, A0))                // row-local block of step q inside the chain of step q+1
// the same stream 4 bytes off the 8-byte grid (every instruction of the body is 8 bytes long: one 4-byte s_nop in front of the
// loop decides for all of them) -- what the two speeds of the layout-D kernels' builds came from (tinympc_solve_d_chain.h, D_AL)
#define KERNEL_OFF(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *out, int iters) {                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();       \
        for (int i = 0; i < iters; ++i) asm volatile(".p2align 3\n\ts_nop 0\n\t" R16(BODY)::: CLOB);            \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();       \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                                               \
            out[0] = t1 - t0;                                                                                    \
            out[1] = r1 - r0;                                                                                    \
        }                                                                                                        \
    }
#define KERNEL_ON(NAME, BODY)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *out, int iters) {                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();       \
        for (int i = 0; i < iters; ++i) asm volatile(".p2align 3\n\t" R16(BODY)::: CLOB);                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();       \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                                               \
            out[0] = t1 - t0;                                                                                    \
            out[1] = r1 - r0;                                                                                    \
        }                                                                                                        \
    }
KERNEL_ON(k_on_grid, CHAIN(A0, A2) TAIL(A0) CHAIN(A2, A0) TAIL(A2))
KERNEL_OFF(k_off_grid, CHAIN(A0, A2) TAIL(A0) CHAIN(A2, A0) TAIL(A2))

typedef void (*kern_t)(unsigned long long *, int);
int main() {
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 16);
    const int iters = 2000;
    struct { const char *name; kern_t f; int valu; } ks[] = {
        {"chain + block, acc bank 0, x bank 0", k_00, 48}, {"chain + block, acc bank 0, x bank 2", k_02, 48}, {"chain + block, acc bank 2, x bank 2", k_22, 48},
        {"chains alone", k_chain, 32}, {"block of step q inside chain q+1", k_mixed, 48},
        {"chain + block, on the 8-byte grid", k_on_grid, 48}, {"chain + block, 4 bytes off the grid", k_off_grid, 48}};
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps) {
        for (auto &k : ks) {
            hipLaunchKernelGGL(k.f, dim3(256 * wps), dim3(256), 0, 0, d, 500);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.f, dim3(256 * wps), dim3(256), 0, 0, d, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            const double mhz = (double)h[0] / ((double)h[1] / 100e6) / 1e6;
            const double instr = (double)iters * 16 * k.valu * wps;  // per SIMD
            printf("%d wavefront(s) per SIMD  %-38s %8.3f ms  clock %.0f MHz  %.3f cycles per VALU instruction and SIMD\n", wps, k.name, ms, mhz,
                   ms * 1e-3 * mhz * 1e6 / instr);
        }
    }
    return 0;
}
