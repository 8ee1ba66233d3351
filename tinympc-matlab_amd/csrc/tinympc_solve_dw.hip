// tinympc_solve_dw.hip -- k_admm_solve_dw: layout D for WIDE systems (16 < nx+nu <= 32, 32 lanes per instance, two instances per wavefront), gfx950 FP64.
// The body is shared with the other wide form (tinympc_solve_dwide.h, one template on the lane width); this file holds what is
// specific to 32 lanes: the chain blocks (tinympc_solve_dw_chain.h), their adapter WideStep<32, nx, nu>, the kernel entry points and
// the table of compiled-in shapes. Every other shape that fits the plan is specialised at run time (tinympc_jit.hip).
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {
template <int NX, int NU>
struct DWStep;  // specialised per (nx, nu) by tinympc_solve_dw_chain.h
}  // namespace tinympc

#ifdef TINY_JIT  // run-time specialisation: exactly one (nx, nu, N), from -D options
#define DW_NX TINY_JIT_NX
#define DW_NU TINY_JIT_NU
#include "tinympc_solve_dw_chain.h"
#else
#define DW_NX 24
#define DW_NU 8
#include "tinympc_solve_dw_chain.h"
#define DW_NX 20
#define DW_NU 4
#include "tinympc_solve_dw_chain.h"
#endif
#include "tinympc_solve_dwide.h"

namespace tinympc {

// WideStep<32, nx, nu>: the operand vector on both DPP rows of the instance (v_permlane16_swap), the chain in two blocks of 16 columns
template <int NX, int NU>
struct WideStep<32, NX, NU> {
    using Blocks = DWStep<NX, NU>;
    struct Operand {
        double e, o;  // even-row / odd-row copy of the operand vector
    };
    static __device__ __forceinline__ void replicate(double w, Operand &op) { cross_row_pair<1>(w, op.e, op.o); }
    static __device__ __forceinline__ double fwd_head(const Operand &op, const double (&m)[32], double cf) { return Blocks::lo_fwd(op.e, m, cf); }
    static __device__ __forceinline__ void fwd_tail_reg(double &a, const Operand &op, const double (&m)[32], double lo, double hi, double &g, double &v,
                                                        double &pri, double &dua) {
        Blocks::hi_fwd_reg(a, op.o, m, lo, hi, g, v, pri, dua);
    }
    static __device__ __forceinline__ void fwd_tail_lds(double &a, const Operand &op, const double (&m)[32], double lo, double hi, double &g, double v,
                                                        double &vnew, double &pri, double &dua) {
        Blocks::hi_fwd_lds(a, op.o, m, lo, hi, g, v, vnew, pri, dua);
    }
    static __device__ __forceinline__ void bwd(double &a, const Operand &op, const double (&m)[32], double v2, double g2, double rhom, double lrmc,
                                               double nrho, double lr, double &an, double &rn) {
        Blocks::lo_bwd(a, op.e, m);
        Blocks::hi_bwd(a, op.o, m, v2, g2, rhom, lrmc, nrho, lr, an, rn);
    }
    static __device__ __forceinline__ void bwd_last(double &a, const Operand &op, const double (&m)[32]) {
        Blocks::lo_bwd(a, op.e, m);
        Blocks::hi_bwd_last(a, op.o, m);
    }
};

#ifdef TINY_JIT_VREG
constexpr int DW_VREG_MAX = TINY_JIT_VREG;  // slack knots kept in registers: chosen by the host from its register estimate
#else
constexpr int DW_VREG_MAX = 20;
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
    constexpr int VLJ = tinympc::wide_vl(32, tinympc::DW_VREG_MAX, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, 4 * TINY_JIT_WPS);
    static_assert(VLJ >= 0, "shape does not fit the layout-D plan");
#ifndef TINY_JIT_FAM
#define TINY_JIT_FAM 0
#endif
    __shared__ __attribute__((aligned(16))) double smem_jit[tinympc::wide_lds_bytes(32, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ) / sizeof(double) +
                                                            (TINY_JIT_FAM != 0 ? tinympc::wide_fam_lin_doubles(32) : 0)];
#ifndef TINY_JIT_FAM
#define TINY_JIT_FAM 0
#endif
    tinympc::k_admm_solve_wide_body<32, TINY_JIT_NX, TINY_JIT_NU, TINY_JIT_N, CTJ, WPGJ, VLJ, TINY_JIT_FAM != 0>(p, smem_jit);  // (FAM: the streamed families, tinympc_solve_dwide.h)
}
#else
template <int NX, int NU, int N, int WPG, int VL>
__global__ void __launch_bounds__(64 * WPG) __attribute__((amdgpu_waves_per_eu(2, 2))) k_admm_solve_dw(const SolveParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    k_admm_solve_wide_body<32, NX, NU, N, true, WPG, VL>(p, smem);  // (compiled in: time-invariant tables; the other form is specialised at run time)
}

// ------------------------------------------------------------------------------------------------------------
// Host side: the instantiation table. A shape runs on this form only if it was compiled in (or specialised at run time).
// ------------------------------------------------------------------------------------------------------------
constexpr int dw_vl(int nu, int N, int wpg) { return wide_vl(32, DW_VREG_MAX, nu, N, true, wpg); }
constexpr int dw_wpg(int nu, int N) { return dw_vl(nu, N, 4) >= 0 ? 4 : 8; }  // (see d_wpg in tinympc_solve_d.hip)

template <int NX, int NU, int N>
static hipError_t launch_dw_one(const SolveParams &p, hipStream_t stream) {
    constexpr int WPG = dw_wpg(NU, N);
    constexpr int VL = dw_vl(NU, N, WPG);
    if constexpr (VL < 0) {
        return hipErrorInvalidValue;
    } else {
        constexpr size_t lds = wide_lds_bytes(32, NU, N, true, WPG, VL);
        static size_t lds_set[16] = {0};
        auto fn = &k_admm_solve_dw<NX, NU, N, WPG, VL>;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fn), lds, lds_set);
        if (e != hipSuccess) return e;
        const int wgs = (p.groups + WPG - 1) / WPG;
        hipLaunchKernelGGL(fn, dim3(wgs), dim3(64 * WPG), lds, stream, p);
        return hipGetLastError();
    }
}

#define TINY_DW_SHAPES(X) \
    X(24, 8, 30)          \
    X(20, 4, 30)

bool solve_dw_supported(int nx, int nu, int N, bool const_tables) {
    if (!const_tables) return false;
#define X(NX_, NU_, N_) \
    if (nx == NX_ && nu == NU_ && N == N_) return dw_vl(NU_, N_, dw_wpg(NU_, N_)) >= 0;
    TINY_DW_SHAPES(X)
#undef X
    return false;
}

size_t solve_dw_lds_bytes(int nu, int N) {
    const int wpg = dw_wpg(nu, N);
    return wide_lds_bytes(32, nu, N, true, wpg, dw_vl(nu, N, wpg));
}

int solve_dw_workgroups(int nu, int N, int groups) {
    const int wpg = dw_wpg(nu, N);
    return (groups + wpg - 1) / wpg;
}

hipError_t launch_solve_dw(const SolveParams &p, hipStream_t stream) {
    if (!p.const_tables) return hipErrorInvalidValue;
#define X(NX_, NU_, N_) \
    if (p.nx == NX_ && p.nu == NU_ && p.N == N_) return launch_dw_one<NX_, NU_, N_>(p, stream);
    TINY_DW_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace tinympc
#endif  // TINY_JIT
