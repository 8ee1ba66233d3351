// tools/microbench_matvec.hip -- development microbenchmark (not part of the product).
// Times the per-step cost of the 16x16 FP64 "row-per-lane" mat-vec chain x <- M x used by
// k_admm_solve, in several instruction formulations, and checks they agree.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_matvec.hip && /tmp/mb
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                      \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

template <int K>
__device__ __forceinline__ double bc(double w) {
    return __builtin_amdgcn_update_dpp(w, w, 0x150 + K, 0xf, 0xf, true);
}

// V0: what the compiler makes of fma(m[k], bcast_k(w), acc[k&3])
__device__ __forceinline__ double mv_v0(const double (&m)[16], double w) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    a0 = fma(m[0], bc<0>(w), a0); a1 = fma(m[1], bc<1>(w), a1); a2 = fma(m[2], bc<2>(w), a2); a3 = fma(m[3], bc<3>(w), a3);
    a0 = fma(m[4], bc<4>(w), a0); a1 = fma(m[5], bc<5>(w), a1); a2 = fma(m[6], bc<6>(w), a2); a3 = fma(m[7], bc<7>(w), a3);
    a0 = fma(m[8], bc<8>(w), a0); a1 = fma(m[9], bc<9>(w), a1); a2 = fma(m[10], bc<10>(w), a2); a3 = fma(m[11], bc<11>(w), a3);
    a0 = fma(m[12], bc<12>(w), a0); a1 = fma(m[13], bc<13>(w), a1); a2 = fma(m[14], bc<14>(w), a2); a3 = fma(m[15], bc<15>(w), a3);
    return (a0 + a1) + (a2 + a3);
}

// V1: all 16 broadcasts first (independent), then the FMAs
__device__ __forceinline__ double mv_v1(const double (&m)[16], double w) {
    double b[16];
    b[0] = bc<0>(w); b[1] = bc<1>(w); b[2] = bc<2>(w); b[3] = bc<3>(w); b[4] = bc<4>(w); b[5] = bc<5>(w); b[6] = bc<6>(w); b[7] = bc<7>(w);
    b[8] = bc<8>(w); b[9] = bc<9>(w); b[10] = bc<10>(w); b[11] = bc<11>(w); b[12] = bc<12>(w); b[13] = bc<13>(w); b[14] = bc<14>(w); b[15] = bc<15>(w);
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(b[k]));  // keep the order: movs, then FMAs
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    a0 = fma(m[0], b[0], a0); a1 = fma(m[1], b[1], a1); a2 = fma(m[2], b[2], a2); a3 = fma(m[3], b[3], a3);
    a0 = fma(m[4], b[4], a0); a1 = fma(m[5], b[5], a1); a2 = fma(m[6], b[6], a2); a3 = fma(m[7], b[7], a3);
    a0 = fma(m[8], b[8], a0); a1 = fma(m[9], b[9], a1); a2 = fma(m[10], b[10], a2); a3 = fma(m[11], b[11], a3);
    a0 = fma(m[12], b[12], a0); a1 = fma(m[13], b[13], a1); a2 = fma(m[14], b[14], a2); a3 = fma(m[15], b[15], a3);
    return (a0 + a1) + (a2 + a3);
}

// V2: fused v_fmac_f64_dpp (VOP2 + DPP row_newbcast), one asm statement; NOPS wait states up front
#define FM(acc, k) "v_fmac_f64_dpp %" #acc ", %4, %" #k " row_newbcast:"
template <int NOPS>
__device__ __forceinline__ double mv_v2(const double (&m)[16], double w) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    asm volatile(
        "s_nop %21\n\t"
        "v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(w), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]),
          "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]), "n"(NOPS));
    return (a0 + a1) + (a2 + a3);
}

// V3: 8 partial sums (shorter dependency chains), compiler-scheduled
__device__ __forceinline__ double mv_v3(const double (&m)[16], double w) {
    double a0 = m[0] * bc<0>(w), a1 = m[1] * bc<1>(w), a2 = m[2] * bc<2>(w), a3 = m[3] * bc<3>(w);
    double a4 = m[4] * bc<4>(w), a5 = m[5] * bc<5>(w), a6 = m[6] * bc<6>(w), a7 = m[7] * bc<7>(w);
    a0 = fma(m[8], bc<8>(w), a0); a1 = fma(m[9], bc<9>(w), a1); a2 = fma(m[10], bc<10>(w), a2); a3 = fma(m[11], bc<11>(w), a3);
    a4 = fma(m[12], bc<12>(w), a4); a5 = fma(m[13], bc<13>(w), a5); a6 = fma(m[14], bc<14>(w), a6); a7 = fma(m[15], bc<15>(w), a7);
    return ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

template <int V>
__global__ void __launch_bounds__(64) chain(const double *M, const double *x0, double *out, int steps) {
    const int lane = threadIdx.x, r = lane & 15;
    double m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = M[r * 16 + k];
    double x = x0[lane];
    for (int i = 0; i < steps; ++i) {
        if (V == 0) x = mv_v0(m, x);
        if (V == 1) x = mv_v1(m, x);
        if (V == 2) x = mv_v2<1>(m, x);
        if (V == 3) x = mv_v3(m, x);
        if (V == 4) x = mv_v2<4>(m, x);
        if (V == 5) x = mv_v2<0>(m, x);
    }
    out[blockIdx.x * 64 + lane] = x;
}

template <int V>
int run(const char *name, const double *dM, const double *dx, double *dout, int steps, int blocks, std::vector<double> &res) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain<V>, dim3(blocks), dim3(64), 0, 0, dM, dx, dout, steps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chain<V>, dim3(blocks), dim3(64), 0, 0, dM, dx, dout, steps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    res.resize(64);
    CK(hipMemcpy(res.data(), dout, 64 * sizeof(double), hipMemcpyDeviceToHost));
    printf("%-28s blocks=%5d  %8.3f ms  %7.1f ns/step  (%6.1f cycles @2.4GHz)  x[0]=%.15e x[17]=%.15e\n", name, blocks, ms,
           1e6 * ms / steps, 2.4e3 * ms / steps * 1e3 / 1e3, res[0], res[17]);
    return 0;
}

int main() {
    const int steps = 20000;
    std::vector<double> M(256), x(64);
    // orthogonal-ish contraction so the chain stays finite: M = 0.999 * (block rotations) + small coupling
    for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 16; ++k) M[r * 16 + k] = 0.02 * std::sin(1.0 + r * 3.1 + k * 1.7);
    for (int r = 0; r < 16; r += 2) {
        double c = std::cos(0.3 + r), s = std::sin(0.3 + r);
        M[r * 16 + r] += 0.9 * c; M[r * 16 + r + 1] += -0.9 * s; M[(r + 1) * 16 + r] += 0.9 * s; M[(r + 1) * 16 + r + 1] += 0.9 * c;
    }
    for (int l = 0; l < 64; ++l) x[l] = 1.0 + 0.01 * l;
    double *dM, *dx, *dout;
    CK(hipMalloc(&dM, 256 * 8));
    CK(hipMalloc(&dx, 64 * 8));
    CK(hipMalloc(&dout, 4096 * 64 * 8));
    CK(hipMemcpy(dM, M.data(), 256 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), 64 * 8, hipMemcpyHostToDevice));
    std::vector<double> r0, r1, r2, r3, r4, r5;
    for (int blocks : {1, 512, 1024, 2048}) {
        run<0>("V0 mov_dpp+fma interleaved", dM, dx, dout, steps, blocks, r0);
        run<1>("V1 movs first", dM, dx, dout, steps, blocks, r1);
        run<2>("V2 fused fmac_dpp nop1", dM, dx, dout, steps, blocks, r2);
        run<4>("V2 fused fmac_dpp nop4", dM, dx, dout, steps, blocks, r4);
        run<5>("V2 fused fmac_dpp nop0", dM, dx, dout, steps, blocks, r5);
        run<3>("V3 8 partial sums", dM, dx, dout, steps, blocks, r3);
    }
    double d1 = 0, d2 = 0, d4 = 0, d5 = 0;
    for (int l = 0; l < 64; ++l) {
        d1 = fmax(d1, fabs(r1[l] - r0[l])); d2 = fmax(d2, fabs(r2[l] - r0[l]));
        d4 = fmax(d4, fabs(r4[l] - r0[l])); d5 = fmax(d5, fabs(r5[l] - r0[l]));
    }
    printf("max |V1-V0| = %.3e   |V2(nop1)-V0| = %.3e   |V2(nop4)-V0| = %.3e   |V2(nop0)-V0| = %.3e\n", d1, d2, d4, d5);
    return 0;
}
