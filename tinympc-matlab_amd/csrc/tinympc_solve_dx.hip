// tinympc_solve_dx.hip -- k_admm_solve_dx: layout D with 64 lanes per instance (32 < nx+nu <= 64, ONE instance per wavefront), gfx950 FP64.
// The body is shared with the other wide form (tinympc_solve_dwide.h, one template on the lane width); this file holds what is
// specific to 64 lanes: the chain blocks (tinympc_solve_dx_chain.h), their adapter WideStep<64, nx, nu>, the kernel entry points and
// the table of compiled-in shapes. Every other shape that fits the plan is specialised at run time (tinympc_jit.hip).
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {
template <int NX, int NU>
struct DXStep;  // specialised per (nx, nu) by tinympc_solve_dx_chain.h
}  // namespace tinympc

#ifdef TINY_JIT  // run-time specialisation: exactly one (nx, nu, N), from -D options
#define DX_NX TINY_JIT_NX
#define DX_NU TINY_JIT_NU
#include "tinympc_solve_dx_chain.h"
#else
#define DX_NX 48
#define DX_NU 16
#include "tinympc_solve_dx_chain.h"
#define DX_NX 40
#define DX_NU 8
#include "tinympc_solve_dx_chain.h"
#endif
#include "tinympc_solve_dwide.h"

namespace tinympc {

// WideStep<64, nx, nu>: the operand vector on all four DPP rows of the wavefront (v_permlane32_swap, then v_permlane16_swap on both
// results), the chain in four blocks of 16 columns
template <int NX, int NU>
struct WideStep<64, NX, NU> {
    using Blocks = DXStep<NX, NU>;
    struct Operand {
        double r0, r1, r2, r3;  // rj = the 16 entries that DPP row j holds of the operand vector, in every row
    };
    static __device__ __forceinline__ void replicate(double w, Operand &op) {
        double pq0, pq1;
        cross_row_pair<0>(w, pq0, pq1);  // [r0 r1 r0 r1], [r2 r3 r2 r3]
        cross_row_pair<1>(pq0, op.r0, op.r1);
        cross_row_pair<1>(pq1, op.r2, op.r3);
    }
    static __device__ __forceinline__ double fwd_head(const Operand &op, const double (&m)[64], double cf) {
        double a = Blocks::q0_fwd(op.r0, m, cf);
        Blocks::q1(a, op.r1, m);
        Blocks::q2(a, op.r2, m);
        return a;
    }
    static __device__ __forceinline__ void fwd_tail_reg(double &a, const Operand &op, const double (&m)[64], double lo, double hi, double &g, double &v,
                                                        double &pri, double &dua) {
        Blocks::q3_fwd_reg(a, op.r3, m, lo, hi, g, v, pri, dua);
    }
    static __device__ __forceinline__ void fwd_tail_lds(double &a, const Operand &op, const double (&m)[64], double lo, double hi, double &g, double v,
                                                        double &vnew, double &pri, double &dua) {
        Blocks::q3_fwd_lds(a, op.r3, m, lo, hi, g, v, vnew, pri, dua);
    }
    static __device__ __forceinline__ void bwd(double &a, const Operand &op, const double (&m)[64], double v2, double g2, double rhom, double lrmc,
                                               double nrho, double lr, double &an, double &rn) {
        Blocks::q0_bwd(a, op.r0, m);
        Blocks::q1(a, op.r1, m);
        Blocks::q2(a, op.r2, m);
        Blocks::q3_bwd(a, op.r3, m, v2, g2, rhom, lrmc, nrho, lr, an, rn);
    }
    static __device__ __forceinline__ void bwd_last(double &a, const Operand &op, const double (&m)[64]) {
        Blocks::q0_bwd(a, op.r0, m);
        Blocks::q1(a, op.r1, m);
        Blocks::q2(a, op.r2, m);
        Blocks::q3_bwd_last(a, op.r3, m);
    }
};

#ifdef TINY_JIT_VREG
constexpr int DX_VREG_MAX = TINY_JIT_VREG;  // slack knots kept in registers: chosen by the host from its register estimate
#else
constexpr int DX_VREG_MAX = 12;
#endif

#ifdef TINY_JIT
}  // namespace tinympc
// The one kernel of a run-time specialised module: a fixed C name, static LDS (its size is known here).
#ifndef TINY_JIT_WPS
#define TINY_JIT_WPS 2  // wavefronts per SIMD: 2 (256 registers each), or 1 (512) for horizons whose duals need them
#endif
#ifndef TINY_JIT_WPG
#define TINY_JIT_WPG (4 * TINY_JIT_WPS)
#endif
#ifndef TINY_JIT_CT
#define TINY_JIT_CT 1
#endif
extern "C" __global__ void __launch_bounds__(64 * TINY_JIT_WPG) __attribute__((amdgpu_waves_per_eu(TINY_JIT_WPS, TINY_JIT_WPS)))
tinympc_jit_solve(const tinympc::SolveParams p) {
    constexpr bool CTJ = TINY_JIT_CT != 0;  // bounds / references constant over the horizon
    constexpr int WPGJ = TINY_JIT_WPG;      // wavefronts per workgroup (4, or 8 where only that LDS plan fits)
    constexpr int VLJ = tinympc::wide_vl(64, tinympc::DX_VREG_MAX, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, 4 * TINY_JIT_WPS);
    static_assert(VLJ >= 0, "shape does not fit the layout-D plan");
#ifndef TINY_JIT_FAM
#define TINY_JIT_FAM 0
#endif
    __shared__ __attribute__((aligned(16))) double smem_jit[tinympc::wide_lds_bytes(64, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ) / sizeof(double) +
                                                            (TINY_JIT_FAM != 0 ? tinympc::wide_fam_lin_doubles(64) : 0)];
#ifndef TINY_JIT_FAM
#define TINY_JIT_FAM 0
#endif
    tinympc::k_admm_solve_wide_body<64, TINY_JIT_NX, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ, TINY_JIT_FAM != 0>(p, smem_jit);  // (FAM: the streamed families, tinympc_solve_dwide.h)
}
#else
template <int NX, int NU, int N, int WPG, int VL>
__global__ void __launch_bounds__(64 * WPG) __attribute__((amdgpu_waves_per_eu(2, 2))) k_admm_solve_dx(const SolveParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    k_admm_solve_wide_body<64, NX, NU, N, true, WPG, VL>(p, smem);  // (compiled in: time-invariant tables; the other form is specialised at run time)
}

// ------------------------------------------------------------------------------------------------------------
// Host side: the instantiation table. A shape runs on this form only if it was compiled in (or specialised at run time).
// ------------------------------------------------------------------------------------------------------------
constexpr int dx_vl(int nu, int N, int wpg) { return wide_vl(64, DX_VREG_MAX, nu, N, true, wpg); }
constexpr int dx_wpg(int nu, int N) { return dx_vl(nu, N, 4) >= 0 ? 4 : 8; }  // (see d_wpg in tinympc_solve_d.hip)

template <int NX, int NU, int N>
static hipError_t launch_dx_one(const SolveParams &p, hipStream_t stream) {
    constexpr int WPG = dx_wpg(NU, N);
    constexpr int VL = dx_vl(NU, N, WPG);
    if constexpr (VL < 0) {
        return hipErrorInvalidValue;
    } else {
        constexpr size_t lds = wide_lds_bytes(64, NU, N, true, WPG, VL);
        static size_t lds_set[16] = {0};
        auto fn = &k_admm_solve_dx<NX, NU, N, WPG, VL>;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), lds, lds_set);
        if (e != hipSuccess) return e;
        const int wgs = (p.groups + WPG - 1) / WPG;
        hipLaunchKernelGGL(fn, dim3(wgs), dim3(64 * WPG), lds, stream, p);
        return hipGetLastError();
    }
}

#define TINY_DX_SHAPES(X) \
    X(48, 16, 20)         \
    X(40, 8, 20)

bool solve_dx_supported(int nx, int nu, int N, bool const_tables) {
    if (!const_tables) return false;
#define X(NX_, NU_, N_) \
    if (nx == NX_ && nu == NU_ && N == N_) return dx_vl(NU_, N_, dx_wpg(NU_, N_)) >= 0;
    TINY_DX_SHAPES(X)
#undef X
    return false;
}

size_t solve_dx_lds_bytes(int nu, int N) {
    const int wpg = dx_wpg(nu, N);
    return wide_lds_bytes(64, nu, N, true, wpg, dx_vl(nu, N, wpg));
}

int solve_dx_workgroups(int nu, int N, int groups) {
    const int wpg = dx_wpg(nu, N);
    return (groups + wpg - 1) / wpg;
}

hipError_t launch_solve_dx(const SolveParams &p, hipStream_t stream) {
    if (!p.const_tables) return hipErrorInvalidValue;
#define X(NX_, NU_, N_) \
    if (p.nx == NX_ && p.nu == NU_ && p.N == N_) return launch_dx_one<NX_, NU_, N_>(p, stream);
    TINY_DX_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace tinympc
#endif  // TINY_JIT
