// Dev microbenchmark: FP64 issue rate of a FULL chip with 1, 2 or 4 wavefronts per SIMD (gfx950), for the instruction
// patterns of the solve kernels' sweep step. Reports the shader clock held under that load (s_memtime / s_memrealtime,
// the latter a constant 100 MHz) and the aggregate cycles per instruction per SIMD -- the FP64 pipe's floor is 4.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_fp64_occupancy.hip -o tools/bin/mb_occ && tools/bin/mb_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP16(X) X X X X X X X X X X X X X X X X
#define DPP " row_mask:0xf bank_mask:0xf\n\t"
// MODE 0: 16 dependent fused DPP FMAs            (the mat-vec chain as is)
// MODE 1: 2 x 8 alternating accumulators         (two partial sums)
// MODE 2: chain alternating with independent FP64 adds (row-local block of the previous step interleaved)
// MODE 3: 16 independent plain FMAs (4 accumulators)  -- the pipe's own rate
template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, unsigned long long *clk, int iters) {
    double a = threadIdx.x, b = 1.0 + 1e-9 * threadIdx.x, c = 0.5, d = 0.25, w = 1.0000001, m = 0.9999999;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3" DPP) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(m));
        if (MODE == 1) asm volatile(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3" DPP "v_fmac_f64_dpp %1, %4, %5 row_newbcast:5" DPP) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(m));
        if (MODE == 2) asm volatile(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3" DPP "v_add_f64 %1, %1, %5\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(m));
        if (MODE == 3) asm volatile(REP16("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(m));
        // MODE 4 / 5: as 0 / 1 but 4096 instructions of straight-line code per trip (32 KB: instruction fetch from the
        // I-cache instead of the wave's instruction buffer), i.e. what a fully unrolled sweep looks like to the front end
        if (MODE == 4) asm volatile(REP16(REP16(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3" DPP))) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(m));
        if (MODE == 5) asm volatile(REP16(REP16(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3" DPP "v_fmac_f64_dpp %1, %4, %5 row_newbcast:5" DPP))) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(m));
        if (MODE == 6) asm volatile(REP16(REP16(REP16("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\t"))) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w), "v"(m));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int MODE>
void run(double *d, unsigned long long *clk, int cus, int waves_per_simd, int iters, int n_instr, const char *name) {
    const int threads = 256 * waves_per_simd;  // one workgroup per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, d, clk, iters);  // warm up the clocks
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, d, clk, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * cus);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost);
    std::vector<double> ghz(cus);
    for (int i = 0; i < cus; ++i) ghz[i] = 0.1 * (double)h[2 * i] / (double)h[2 * i + 1];
    std::sort(ghz.begin(), ghz.end());
    const double clock = ghz[cus / 2];
    const double ns_per_instr_simd = 1e6 * ms / ((double)iters * n_instr * waves_per_simd);
    printf("%-44s %d waves/SIMD  %7.3f ms  clock %.3f GHz  %.3f ns = %.2f cycles per instruction per SIMD\n", name, waves_per_simd, ms, clock,
           ns_per_instr_simd, ns_per_instr_simd * clock);
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *d; unsigned long long *clk;
    (void)hipMalloc(&d, sizeof(double) * cus * 1024); (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * cus);
    printf("%s, %d CUs\n", prop.name, cus);
    const int iters = 40000;
    for (int w : {1, 2, 4}) {
        run<0>(d, clk, cus, w, iters, 16, "16 dependent fmac_dpp");
        run<1>(d, clk, cus, w, iters, 32, "2 accumulators alternating fmac_dpp");
        run<2>(d, clk, cus, w, iters, 32, "fmac_dpp chain + independent v_add_f64");
        run<3>(d, clk, cus, w, iters, 64, "4 independent plain fmac");
        run<4>(d, clk, cus, w, iters / 256, 4096, "dependent fmac_dpp, 32 KB straight-line");
        run<5>(d, clk, cus, w, iters / 256, 8192, "2 accumulators fmac_dpp, 64 KB straight-line");
        run<6>(d, clk, cus, w, iters / 256, 8192, "2 accumulators plain fmac (VOP2 4 B), 32 KB");
    }
    return 0;
}
