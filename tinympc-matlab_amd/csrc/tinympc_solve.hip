// tinympc_solve.hip -- k_admm_solve: the whole TinyMPC solve() as ONE persistent kernel (gfx950, FP64).
//
//   M1 solve                 admm.cpp:109-207      F1 forward_pass          admm.cpp:25-35
//   S1 update_slack          admm.cpp:43-59        D1 update_dual           admm.cpp:65-69
//   L1 update_linear_cost    admm.cpp:75-83        R1 termination_condition admm.cpp:89-107
//   C1 v/z copies            admm.cpp:196-197      B1 backward_pass_grad    admm.cpp:13-20
//
// Design (DESIGN.md has the full rationale and the measurements behind each choice):
//   * one wavefront = 64/W MPC instances, W lanes per instance, one lane per state/input row;
//   * the ADMM state that survives an iteration (duals g|y, slack v|z, feed-forward d) lives in LDS
//     for the whole solve: HBM is touched once at entry and once at exit (plus the stale-v stream, below);
//   * each sweep step is ONE (nx+nu)x(nx+nu) mat-vec per instance: the operand vector is spread over
//     the W lanes of the instance and broadcast with DPP row_newbcast fused into v_fmac_f64 (W=16),
//     so a step is KT FP64 FMAs per lane and no LDS round trip sits on the dependency chain;
//   * slack projection, dual ascent, linear-cost refresh and the four inf-norm residuals are row-local,
//     so they are fused into the forward sweep, lane by lane, right after the lane's row of
//     x_{i+1} / u_i has been produced; their LDS operands are prefetched one step ahead;
//   * a lone wavefront issues at most one VALU instruction every ~4 cycles (8 for FP64), so the kernel
//     is instruction-issue bound. The sweep bodies are therefore branch-free (lanes that must not store
//     write to a per-lane dummy row instead of being masked off), address arithmetic is reduced to
//     pointer increments (arrays carry a padding row at each end so the prefetch needs no clamping),
//     the sweeps are unrolled by two so that the prefetch registers ping-pong without moves, and the
//     FMA chain and the row-local math are single asm blocks (no per-statement hazard padding).
#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {

// LDS plan per workgroup (= one wavefront), in doubles:
//   G[N+2][64]  V[N+2][64]   row k+1 = knot k. Row 0 and row N+1 are padding touched by the one-step-
//                            ahead prefetch at the ends of a sweep; row N+1 doubles as the per-lane
//                            dummy slot that lanes which must not store (converged instance, padding
//                            lanes) write to, so the sweeps need no exec masking.
//   D[(N-1)*IPW*nu + 64]     feed-forward term, compact; the last 64 are dummy / overshoot slots
//   tables (optional)        lo | hi | linref [N+2][W], pNref[W]
size_t solve_lds_bytes(int nx, int nu, int N, int W, bool tables_in_lds) {
    const int ipw = 64 / W;
    size_t d = (size_t)2 * (N + 2) * 64 + (size_t)(N - 1) * ipw * nu + 64;
    d = (d + 1) & ~(size_t)1;
    if (tables_in_lds) d += tables_doubles(W, N);
    (void)nx;
    return d * sizeof(double);
}

bool choose_geometry(int nx, int nu, int *W, int *KT) {
    const int nxu = nx + nu;
    if (nx < 1 || nu < 1 || nxu > 64) return false;
    if (nxu <= 8) { *W = 16; *KT = 8; }
    else if (nxu <= 12) { *W = 16; *KT = 12; }
    else if (nxu <= 16) { *W = 16; *KT = 16; }
    else if (nxu <= 32) { *W = 32; *KT = 32; }
    else { *W = 64; *KT = 64; }
    return true;
}

struct FwdOperands { double g, vold, lo, hi, dv; };
struct BwdOperands { double bg, bv, blr; };

// GMEM: the working copy of the state lives in p.scratch (HBM) instead of LDS -- the fallback for horizons
// that do not fit 160 KB of LDS. Same code, same results; the row-local operands then come from L2.
template <int W, int KT, bool TLDS, bool GMEM = false>
__global__ void __launch_bounds__(64) k_admm_solve(const SolveParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    refresh_reference_tables(p, W, KT);  // references handed over in pinned host memory (single-instance handles)
    constexpr int IPW = 64 / W;
    const int lane = threadIdx.x;
    const int j = lane / W, r = lane % W;
    const int nx = p.nx, nu = p.nu, N = p.N, nxu = nx + nu;
    const long grp = blockIdx.x;
    const long inst = grp * IPW + j;
    const bool is_x = r < nx;
    const bool is_u = (r >= nx) && (r < nxu);
    const bool inst_ok = inst < p.batch;
    const bool row_ok = inst_ok && (r < nxu);
    const int dstride = IPW * nu;
    const int dsize = (N - 1) * dstride;
    const int VOFF = (N + 2) * 64;             // sV[k] - sG[k]
    const int TOFF = (int)table_rows(N) * W;   // hi[k] - lo[k]
    const int ldummy = (N + 1) * 64 + lane;    // this lane's dummy slot (LDS row N+1)
    const int gdummy = N * 64 + lane;          // same in the HBM layout (row N)

    double *sG = GMEM ? (p.scratch + (size_t)blockIdx.x * p.scratch_stride) : smem;
    double *sV = sG + VOFF;
    double *sD = sV + VOFF;
    double *sT = sD + ((dsize + 64 + 1) & ~1);
    const double *tab = TLDS ? sT : p.tables;
    const double *t_lo = tab, *t_lr = tab + 2 * TOFF;

    double *gG = p.G + (size_t)grp * (N + 1) * 64;
    double *gV = p.V + ((size_t)grp * v_rows(N) + V_PAD) * 64;  // knot 0
    double *gD = p.D + (size_t)grp * dsize;

    // ---- one coalesced pass HBM -> LDS (512-byte lines); knot k lands in LDS row k+1
    for (int kn = 0; kn < N; ++kn) {
        sG[(kn + 1) * 64 + lane] = gG[kn * 64 + lane];
        sV[(kn + 1) * 64 + lane] = gV[kn * 64 + lane];
    }
    sG[lane] = 0.0;
    sV[lane] = 0.0;
    sG[ldummy] = 0.0;
    sV[ldummy] = 0.0;
    for (int i = lane; i < dsize; i += 64) sD[i] = gD[i];
    sD[dsize + lane] = 0.0;
    if (TLDS) {
        const int tn = (int)tables_doubles(W, N);
        for (int i = lane; i < tn; i += 64) sT[i] = p.tables[i];
    }

    // ---- per-lane operator rows and constants (registers for the whole solve)
    double mf[KT], mb[KT];
    {
        const double *Mf = p.ops + (size_t)r * KT, *Mb = p.ops + (size_t)W * KT + (size_t)r * KT;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            mf[k] = Mf[k];
            mb[k] = Mb[k];
        }
    }
    const double cf = p.ops[(size_t)2 * W * KT + r];
    const double cb = p.ops[(size_t)2 * W * KT + W + r];
    const double pnref = p.tables[(size_t)3 * TOFF + r];
    const double rho = p.rho;
    const double x0v = (inst_ok && is_x) ? p.x0[inst * nx + r] : 0.0;
    if (p.x0_mirror && inst_ok && is_x) p.x0_mirror[inst * nx + r] = x0v;  // zero-copy tick: x0 came from host memory
    const int dIdx = is_u ? (j * nu + (r - nx)) : 0;
    const int koff = is_x ? 1 : 0;  // at step i a state lane finishes knot i+1, an input lane knot i
    const int ct = p.check_termination;
    __syncthreads();

    bool active = inst_ok;
    int it_done = 0;
    int status = 11;  // TINY_UNSOLVED (admm.cpp:114)
    bool res_valid = false;
    double snap_pri = 0.0, snap_dua = 0.0;  // this lane's residual maxima at its instance's last termination check

    for (int it = 0; it < p.max_iter; ++it) {  // admm.cpp:129
        if (__ballot(active) == 0ull) break;
        const bool check = (ct > 0) && (((it + 1) % ct) == 0);  // admm.cpp:91 (iter already incremented, :143)
        const bool st = active && row_ok;
        double pri, dua;

        // ---------------- forward sweep (F1) with the row-local phases S1+D1+R1 fused in.
        // The reference returns from a converged solve BEFORE v <- vnew (admm.cpp:181-197), so its
        // workspace keeps the previous iteration's v/z: on check iterations the old value is streamed to
        // HBM while it is still in a register; on convergence that copy is exactly the reference's v/z.
        {   // knot 0, state lanes only: x_0 is given (tiny_set_x0), no mat-vec
            const bool on = st && is_x;
            const double g = sG[64 + lane], vold = sV[64 + lane];
            const double s = x0v + g;
            const double snew = fmin(t_lo[TOFF + W + r], fmax(t_lo[W + r], s));
            pri = is_x ? fabs(x0v - snew) : 0.0;
            dua = is_x ? fabs(vold - snew) : 0.0;
            if (check) gV[on ? lane : gdummy] = vold;
            sG[on ? 64 + lane : ldummy] = s - snew;
            sV[on ? 64 + lane : ldummy] = snew;
        }
        {
            const double *pg = sG + (1 + koff) * 64 + lane;  // this lane's operands of step 0
            const double *pt = t_lo + (1 + koff) * W + r;
            const double *pd = sD + dIdx;
            double *ps = sG + (st ? (1 + koff) * 64 + lane : ldummy);
            double *pgv = gV + (st ? koff * 64 + lane : gdummy);
            const int inc = st ? 64 : 0;
            double xcur = x0v;
            FwdOperands A{pg[0], pg[VOFF], pt[0], pt[TOFF], pd[0]}, B;
            auto fstep = [&](const FwdOperands &cur, FwdOperands &nxt) {
                const double w = is_x ? xcur : cur.dv;
                pg += 64;  // operands of the next step, fetched while this step's mat-vec runs
                pt += W;
                pd += dstride;
                nxt.g = pg[0]; nxt.vold = pg[VOFF]; nxt.lo = pt[0]; nxt.hi = pt[TOFF]; nxt.dv = pd[0];
                const double out = group_matvec<W, KT>(mf, w, cf);  // state lanes: x_{i+1}; input lanes: u_i
                double gnew, snew;
                project_element(out, cur.g, cur.lo, cur.hi, cur.vold, gnew, snew, pri, dua);
                if (check) *pgv = cur.vold;
                ps[0] = gnew;
                ps[VOFF] = snew;
                ps += inc;
                pgv += inc;
                xcur = out;
            };
            int i = 0;
            for (; i + 2 <= N - 1; i += 2) {
                fstep(A, B);
                fstep(B, A);
            }
            if (i < N - 1) fstep(A, B);
        }
        if (active) it_done = it + 1;  // admm.cpp:143

        // ---------------- R1: termination test (admm.cpp:93-101)
        if (check) {
            // decided element-wise with one ballot (max_i a_i < tol iff every a_i < tol; rho > 0): see tinympc_solve_b.hip
            const bool below = (pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol);
            constexpr unsigned long long ones = (W == 64) ? ~0ull : ((1ull << (W % 64)) - 1ull);
            const bool conv = ((__ballot(below) >> (j * W)) & ones) == ones;
            if (active) {
                snap_pri = pri;
                snap_dua = dua;
                res_valid = true;
                if (conv) {
                    status = 1;  // TINY_SOLVED: stop this instance before the backward pass (admm.cpp:181-192)
                    active = false;
                }
            }
        }

        // ---------------- backward sweep (B1, admm.cpp:13-20); linear cost (L1, :77-82) recomputed from V,G
        {
            const bool stb = active && row_ok && is_u;
            const double *pb = sG + N * 64 + lane;  // knot N-1
            double pcur = pnref - rho * (pb[VOFF] - pb[0]);  // p_{N-1}, admm.cpp:81-82 (state lanes)
            pb -= 64;                                        // knot N-2
            const double *pl = t_lr + (N - 1) * W + r;
            double *pdst = sD + (stb ? (N - 2) * dstride + dIdx : dsize + lane);
            const int ddec = stb ? dstride : 0;
            BwdOperands A{pb[0], pb[VOFF], pl[0]}, B;
            auto bstep = [&](const BwdOperands &cur, BwdOperands &nxt) {
                const double lin = cur.blr - rho * (cur.bv - cur.bg);  // q_i (state lanes) / r_i (input lanes), admm.cpp:77-80
                const double w = is_x ? pcur : lin;
                pb -= 64;
                pl -= W;
                nxt.bg = pb[0]; nxt.bv = pb[VOFF]; nxt.blr = pl[0];
                const double out = group_matvec<W, KT>(mb, w, cb);
                *pdst = out;  // d_i (input lanes)
                pdst -= ddec;
                pcur = lin + out;  // p_i (state lanes)
            };
            int i = N - 2;
            for (; i >= 1; i -= 2) {
                bstep(A, B);
                bstep(B, A);
            }
            if (i == 0) bstep(A, B);
        }
    }

    // ---- write-back: state for the next (warm-started) solve, solution, stats
    // the four norms of the last check (for get_stats), reduced once
    const double res_px = group_max<W>(is_x ? snap_pri : 0.0), res_pu = group_max<W>(is_u ? snap_pri : 0.0);
    const double res_dx = group_max<W>(is_x ? snap_dua : 0.0) * rho, res_du = group_max<W>(is_u ? snap_dua : 0.0) * rho;

    if (p.max_iter > 0 && inst_ok) {
        for (int kn = 0; kn < N; ++kn) {
            const int e = (kn + 1) * 64 + lane;
            gG[kn * 64 + lane] = sG[e];
            if (status != 1) gV[kn * 64 + lane] = sV[e];  // converged: HBM already holds the reference's stale v/z
            const double sol = sV[e];                       // solution = vnew / znew (admm.cpp:187-188, 204-205)
            if (is_x) p.sol_x[((size_t)inst * N + kn) * nx + r] = sol;
            if (is_u && kn < N - 1) p.sol_u[((size_t)inst * (N - 1) + kn) * nu + (r - nx)] = sol;
            if (is_u && kn == 0 && p.u0_host) p.u0_host[(size_t)inst * nu + (r - nx)] = sol;  // first controls straight to the host
            if (p.host_sol) {  // single-instance handle: the solution also goes straight into pinned host memory
                if (is_x) p.host_sol[(size_t)kn * nx + r] = sol;
                if (is_u && kn < N - 1) p.host_sol[(size_t)N * nx + (size_t)kn * nu + (r - nx)] = sol;
            }
        }
        if (is_u)
            for (int i = 0; i < N - 1; ++i) gD[i * dstride + dIdx] = sD[i * dstride + dIdx];
    }
    if (inst_ok && r == 0) {
        p.istats[inst * 2 + 0] = it_done;
        p.istats[inst * 2 + 1] = status;
        if (p.host_sol) {
            double *hs = p.host_sol + (size_t)N * nx + (size_t)(N - 1) * nu;
            hs[4] = (double)it_done;
            hs[5] = (double)status;
            if (res_valid) { hs[0] = res_px; hs[1] = res_dx; hs[2] = res_pu; hs[3] = res_du; }
        }
        if (res_valid) {
            p.dstats[inst * 4 + 0] = res_px;
            p.dstats[inst * 4 + 1] = res_dx;
            p.dstats[inst * 4 + 2] = res_pu;
            p.dstats[inst * 4 + 3] = res_du;
        }
    }
}

template <int W, int KT>
static hipError_t launch_solve_t(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    constexpr int IPW = 64 / W;
    const int groups = (p.batch + IPW - 1) / IPW;
    static size_t lds_set_t[16] = {0}, lds_set_f[16] = {0};
    hipError_t e;
    if (p.scratch) {  // state in HBM scratch, tables from global memory, no dynamic LDS at all
        hipLaunchKernelGGL((k_admm_solve<W, KT, false, true>), dim3(groups), dim3(64), 0, stream, p);
    } else if (p.tables_in_lds) {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve<W, KT, true>), lds_bytes, lds_set_t);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_admm_solve<W, KT, true>), dim3(groups), dim3(64), lds_bytes, stream, p);
    } else {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve<W, KT, false>), lds_bytes, lds_set_f);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_admm_solve<W, KT, false>), dim3(groups), dim3(64), lds_bytes, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_solve(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream) {
    if (W == 16 && KT == 8) return launch_solve_t<16, 8>(p, lds_bytes, stream);
    if (W == 16 && KT == 12) return launch_solve_t<16, 12>(p, lds_bytes, stream);
    if (W == 16 && KT == 16) return launch_solve_t<16, 16>(p, lds_bytes, stream);
    if (W == 32 && KT == 32) return launch_solve_t<32, 32>(p, lds_bytes, stream);
    if (W == 64 && KT == 64) return launch_solve_t<64, 64>(p, lds_bytes, stream);
    return hipErrorInvalidValue;
}

}  // namespace tinympc
