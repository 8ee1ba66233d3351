// Dev microbenchmark (gfx950): can the FP64 matrix pipe add to the FP64 vector pipe?
//
// MI355X lists the same peak for FP64 MFMA and FP64 VALU (78.6 TFLOP/s). The solve kernels are bound by VALU FP64 issue;
// if v_mfma_f64 ran on its own pipe, a second wavefront on the same SIMD could do mat-vec work there while the first
// one keeps the VALU busy. This measures, per SIMD, the instruction rate of
//   V : a dependent chain of v_fmac_f64_dpp                     (the sweeps' mat-vec)
//   M : v_mfma_f64_16x16x4_f64, four independent accumulators   (2048 flop per instruction)
//   B : v_mfma_f64_4x4x4_4b_f64, eight independent accumulators (512 flop per instruction)
// alone (one wavefront per SIMD) and side by side (two wavefronts per SIMD: waves 0-3 of the workgroup run V, waves 4-7
// run M or B), with the shader clock held under each load (s_memtime against the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma_f64.hip -o tools/bin/mb_mfma && tools/bin/mb_mfma
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define REP16(X) X X X X X X X X X X X X X X X X
#define DPP " row_mask:0xf bank_mask:0xf\n\t"

__device__ __forceinline__ double run_valu(int iters, double seed) {
    double a = seed, w = 1.0000001, m = 0.9999999;
    for (int i = 0; i < iters; ++i) asm volatile(REP16("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3" DPP) : "+v"(a) : "v"(w), "v"(m));
    return a;
}
__device__ __forceinline__ double run_mfma16(int iters, double seed) {
    double4_t c0 = {seed, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0000001, b = 0.9999999 + 1e-9 * seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
}
__device__ __forceinline__ double run_mfma4(int iters, double seed) {
    double c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = seed + k;
    const double a = 1.0000001, b = 0.9999999 + 1e-9 * seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[k], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += c[k];
    return s;
}

// lo: what waves 0-3 of the workgroup run, hi: what waves 4-7 run (0 = nothing / not launched, 'V', 'M', 'B')
template <char LO, char HI>
__global__ void __launch_bounds__(512) k(double *out, unsigned long long *clk, int iters) {
    const int wave = threadIdx.x >> 6;
    const char what = wave < 4 ? LO : HI;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double res = 0.0;
    if (what == 'V') res = run_valu(iters, (double)threadIdx.x);
    if (what == 'M') res = run_mfma16(iters, (double)threadIdx.x);
    if (what == 'B') res = run_mfma4(iters, (double)threadIdx.x);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) {
        clk[((size_t)blockIdx.x * 8 + wave) * 2] = t1 - t0;
        clk[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
}

template <char LO, char HI>
void run(double *d, unsigned long long *clk, int cus, int iters, const char *name) {
    const int waves = HI ? 8 : 4;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<LO, HI>), dim3(cus), dim3(64 * waves), 0, 0, d, clk, iters);
    hipLaunchKernelGGL((k<LO, HI>), dim3(cus), dim3(64 * waves), 0, 0, d, clk, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)cus * 16);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    printf("%-34s", name);
    for (int half = 0; half < (HI ? 2 : 1); ++half) {
        std::vector<double> cyc, ghz;
        for (int b = 0; b < cus; ++b)
            for (int w = half * 4; w < half * 4 + 4; ++w) {
                cyc.push_back((double)h[((size_t)b * 8 + w) * 2]);
                ghz.push_back(0.1 * (double)h[((size_t)b * 8 + w) * 2] / (double)h[((size_t)b * 8 + w) * 2 + 1]);
            }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ghz.begin(), ghz.end());
        const char what = half ? HI : LO;
        const double per_iter = what == 'V' ? 16.0 : 16.0;  // instructions per loop trip (all three: 16)
        const double flop = what == 'V' ? 128.0 : what == 'M' ? 2048.0 : 512.0;
        const double cpi = cyc[cyc.size() / 2] / ((double)iters * per_iter);
        printf("  | %c: %6.2f cycles/instr = %5.1f flop/cycle/SIMD, clock %.2f GHz", what, cpi, flop / cpi, ghz[ghz.size() / 2]);
    }
    printf("\n");
}

// ---- semantics probe of the gfx950 cross-row swaps (what the wide lane layouts are built on)
__global__ void k_swap(int *out) {
    const int lane = threadIdx.x;
    int a = lane, b = 100 + lane;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[lane] = a;
    out[64 + lane] = b;
    int c = lane, e = 100 + lane;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(e));
    out[128 + lane] = c;
    out[192 + lane] = e;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *d;
    unsigned long long *clk;
    int *sw;
    (void)hipMalloc(&d, sizeof(double) * cus * 512);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * 16 * cus);
    (void)hipMalloc(&sw, sizeof(int) * 256);
    printf("%s, %d CUs\n", prop.name, cus);
    const int iters = 20000;
    run<'V', 0>(d, clk, cus, iters, "V alone (1 wave/SIMD)");
    run<'M', 0>(d, clk, cus, iters, "M alone (1 wave/SIMD)");
    run<'B', 0>(d, clk, cus, iters, "B alone (1 wave/SIMD)");
    run<'V', 'V'>(d, clk, cus, iters, "V + V");
    run<'M', 'M'>(d, clk, cus, iters, "M + M");
    run<'V', 'M'>(d, clk, cus, iters, "V + M");
    run<'V', 'B'>(d, clk, cus, iters, "V + B");
    hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, sw);
    int h[256];
    (void)hipMemcpy(h, sw, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap vdst (was lane)", "permlane16_swap src0 (was 100+lane)", "permlane32_swap vdst (was lane)",
                            "permlane32_swap src0 (was 100+lane)"};
    for (int q = 0; q < 4; ++q) {
        printf("%s: rows", names[q]);
        for (int row = 0; row < 4; ++row) printf(" [%d..%d]", h[q * 64 + row * 16], h[q * 64 + row * 16 + 15]);
        printf("\n");
    }
    return 0;
}
