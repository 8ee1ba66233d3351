// Dev microbenchmark: issue cost of FP64 FMAs on one lone wavefront (gfx950) -- dependent vs independent chains,
// with and without DPP row_newbcast on the multiplicand. Prints ns and cycles (at the measured shader clock) per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ void k(double *out, int iters) {
    double a = threadIdx.x, b = 1.0 + 1e-9 * threadIdx.x, w = 1.0000001, m = 0.9999999;
    int c = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP16("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a), "+v"(b) : "v"(w), "v"(m));
        if (MODE == 1) asm volatile(REP16("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a), "+v"(b) : "v"(w), "v"(m));
        if (MODE == 2) asm volatile(REP16("v_fmac_f64 %0, %2, %3\n\t") : "+v"(a), "+v"(b) : "v"(w), "v"(m));
        if (MODE == 3) asm volatile(REP16("v_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %3\n\t") : "+v"(a), "+v"(b) : "v"(w), "v"(m));
        if (MODE == 4) asm volatile(REP16("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_add_f64 %1, %1, %3\n\t") : "+v"(a), "+v"(b) : "v"(w), "v"(m));
        if (MODE == 5) asm volatile(REP16("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_add_u32 %4, %4, %4\n\t") : "+v"(a), "+v"(b) : "v"(w), "v"(m), "v"(c));
    }
    out[threadIdx.x] = a + b + c;
}
template <int MODE>
float run(double *d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, 100);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *d; (void)hipMalloc(&d, 64 * 8);
    const int iters = 200000;
    const char *names[] = {"16 dependent fmac_dpp", "16x2 alternating fmac_dpp (2 accumulators)", "16 dependent fmac", "16x2 alternating fmac", "16x (fmac_dpp + independent v_add_f64)", "16x (fmac_dpp + independent v_add_u32)"};
    float t[6] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters)};
    int n[6] = {16, 32, 16, 32, 32, 32};
    for (int i = 0; i < 6; ++i) printf("%-48s %7.3f ns/instr  (%.2f cycles at 2.4 GHz)  block %.1f ns\n", names[i], 1e6 * t[i] / iters / n[i], 2.4e3 * t[i] / iters / n[i], 1e6 * t[i] / iters);
    return 0;
}
