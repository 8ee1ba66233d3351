// tools/microbench_dpp_masks.hip -- do row_mask / bank_mask work on the 64-bit DPP FMA of gfx950 (v_fmac_f64_dpp ... row_newbcast:k)?
// The ISA text takes the fields; what the hardware does with them for DP operations is not in the guides. One wavefront: acc = 1000 + lane,
// x = lane, acc += lane_k(x) * 1.0 with k = 3 under (row_mask, bank_mask) = (0xf, 0xf) | (0xf, 0x3) | (0xf, 0xc) | (0x5, 0xf) | (0xa, 0x3);
// prints, per variant, which lanes changed and by how much (expected: += 16 * row + 3 on the enabled lanes).
// Why: two instances per 16-lane DPP row (systems with nx+nu <= 8, e.g. the cartpole) need the two halves of a row to read DIFFERENT lanes.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_dpp_masks.hip -o tools/bin/microbench_dpp_masks && tools/bin/microbench_dpp_masks
#include <hip/hip_runtime.h>
#include <cstdio>

#define VARIANT(name, rm, bm)                                                                                                      \
    __global__ void name(double *out) {                                                                                            \
        const int lane = threadIdx.x;                                                                                              \
        double acc = 1000.0 + lane, x = (double)lane, one = 1.0;                                                                   \
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %[a], %[x], %[o] row_newbcast:3 row_mask:" rm " bank_mask:" bm "\n\ts_nop 1"        \
                     : [a] "+v"(acc) : [x] "v"(x), [o] "v"(one));                                                                  \
        out[lane] = acc;                                                                                                           \
    }
VARIANT(k_ff, "0xf", "0xf")
VARIANT(k_f3, "0xf", "0x3")
VARIANT(k_fc, "0xf", "0xc")
VARIANT(k_5f, "0x5", "0xf")
VARIANT(k_a3, "0xa", "0x3")

int main() {
    double *d, h[64];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    struct { const char *what; void (*k)(double *); } v[] = {{"row_mask 0xf bank_mask 0xf", k_ff}, {"row_mask 0xf bank_mask 0x3", k_f3}, {"row_mask 0xf bank_mask 0xc", k_fc},
                                                              {"row_mask 0x5 bank_mask 0xf", k_5f}, {"row_mask 0xa bank_mask 0x3", k_a3}};
    for (auto &e : v) {
        hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, d);
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        unsigned long long changed = 0, right = 0;
        for (int l = 0; l < 64; ++l) {
            const double delta = h[l] - (1000.0 + l);
            if (delta != 0.0) changed |= 1ull << l;
            if (delta == 16.0 * (l / 16) + 3.0) right |= 1ull << l;
        }
        printf("%s: lanes changed %016llx, of them with the expected increment %016llx\n", e.what, changed, right & changed);
    }
    return 0;
}
