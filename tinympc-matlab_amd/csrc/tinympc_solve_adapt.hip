// tinympc_solve_adapt.hip -- k_admm_solve_adapt: the solve loop with adaptive rho.
//
//   admm.cpp:117-174            every 5th iteration (i > 0 && i % 5 == 0), after update_linear_cost and before
//                               the termination test: benchmark_rho_adaptation + Taylor update of the cache
//   rho_benchmark.cpp:44-150    format_matrices   -- the reference assembles a dense KKT-style system
//   rho_benchmark.cpp:152-180   compute_residuals -- four inf-norms of dense mat-vecs with it
//   rho_benchmark.cpp:182-198   predict_rho       -- rho * sqrt(normalised primal / normalised dual), clipped
//   rho_benchmark.cpp:200-216   update_matrices_with_derivatives -- Kinf, Pinf (C1, C2) += d_rho * sensitivities
//
// Same lane layout and LDS plan as layout A (tinympc_solve.hip); what is added:
//   * rho is PER INSTANCE (each instance of a batch adapts on its own) and persists across solves in p.rho_inst,
//     like the reference's cache->rho. The reference's cache is then K0 + (rho - rho0) dK, P0 + (rho - rho0) dP, so
//     the lane's rows of the two sweep operators are rebuilt from (base row) + (rho - rho0) * (derivative row) after
//     every adaptation -- no per-instance matrices are stored anywhere;
//   * the dense system is never formed. With x_decision = [x_0; u_0; x_1; ...] its rows/columns are knots, so the
//     four norms are row-local maxima that ride on the forward sweep of an adaptation iteration, plus ONE extra
//     mat-vec per step, [A'; B'] g_{i+1} (the A_matrix' * y_vector term), and one Pinf * x_{N-1} at the end:
//        primal rows     u_i - znew_i                    |  (A x_i + B u_i - x_{i+1}) - vnew_{i+1}
//        dual cols x_i   2 Q.*x_i + A' g_{i+1} - g_i     |  x_0: no -g_0 ; x_{N-1}: Pinf x + Q.*x - g_{N-1}
//        dual cols u_i   2 R.*u_i + y_i + B' g_{i+1}
//     The dynamics defect A x_i + B u_i - x_{i+1} is taken as exactly 0 (x_{i+1} was just computed as that sum;
//     the reference's dense product differs from it by rounding only, ~1e-16 relative to the norms it enters).
//   * the linear-cost terms of the iteration were formed with the OLD rho and Pinf (update_linear_cost runs before
//     the adaptation), the termination test and the backward pass use the NEW rho and Kinf: the backward sweep of
//     an adaptation iteration therefore takes rho / p_N's reference term from before the update.
// Not adapted: Quu_inv and AmBKt (the reference updates the copies C1 / C2, which no solve phase reads) and the
// affine-dynamics constants APf / BPf (not in the snapshot).
#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {

namespace {
struct FwdOps { double g, vold, lo, hi, dv; };
struct BwdOps { double bg, bv, blr; };
__device__ __forceinline__ double amax2(double m, double v) { return fmax(m, fabs(v)); }
}  // namespace

// Tables for the kernel below (adapt_doubles()): mt | pinf | dpinf | dmf | dmb [W][KT], then dpnref[W].
__global__ void __launch_bounds__(256) k_build_adapt(const AdaptTableParams p) {
    const int nx = p.nx, nu = p.nu, W = p.W, KT = p.KT, nxu = nx + nu;
    const size_t M = (size_t)W * KT;
    double *mt = p.out, *pinf = mt + M, *dpinf = pinf + M, *dmf = dpinf + M, *dmb = dmf + M, *dpn = dmb + M;
    for (int idx = threadIdx.x; idx < W * KT; idx += 256) {
        const int r = idx / KT, k = idx % KT;
        double vt = 0.0, vp = 0.0, vdp = 0.0, vf = 0.0, vb = 0.0;
        if (r < nx && k < nx) {
            vt = p.A[k + (size_t)r * nx];  // A'
            vp = p.Pinf[r + (size_t)k * nx];
            vdp = p.dP[r + (size_t)k * nx];
            for (int j = 0; j < nu; ++j) vf -= p.B[r + (size_t)j * nx] * p.dK[j + (size_t)k * nu];  // d(A - B K)
        } else if (r < nx && k < nxu) {
            vb = -p.dK[(k - nx) + (size_t)r * nu];  // d(-K')
        } else if (r < nxu && k < nx) {
            vt = p.B[k + (size_t)(r - nx) * nx];  // B'
            vf = -p.dK[(r - nx) + (size_t)k * nu];  // d(-K)
        }
        mt[idx] = vt; pinf[idx] = vp; dpinf[idx] = vdp; dmf[idx] = vf; dmb[idx] = vb;
    }
    for (int r = threadIdx.x; r < W; r += 256) {
        double v = 0.0;  // d/drho of -(Xref_{N-1}' Pinf)'  (admm.cpp:81)
        if (r < nx)
            for (int i = 0; i < nx; ++i) v -= p.Xref[i + (size_t)(p.N - 1) * nx] * p.dP[i + (size_t)r * nx];
        dpn[r] = v;
    }
}

hipError_t launch_build_adapt(const AdaptTableParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_adapt, dim3(1), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <int W, int KT, bool TLDS, bool GMEM>
__global__ void __launch_bounds__(64) k_admm_solve_adapt(const SolveParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
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
    const int VOFF = (N + 2) * 64;
    const int TOFF = (int)table_rows(N) * W;
    const int ldummy = (N + 1) * 64 + lane;
    const int gdummy = N * 64 + lane;

    double *sG = GMEM ? (p.scratch + (size_t)blockIdx.x * p.scratch_stride) : smem;
    double *sV = sG + VOFF;
    double *sD = sV + VOFF;
    double *sT = sD + ((dsize + 64 + 1) & ~1);
    const double *tab = TLDS ? sT : p.tables;
    const double *t_lo = tab, *t_lr = tab + 2 * TOFF;

    double *gG = p.G + (size_t)grp * (N + 1) * 64;
    double *gV = p.V + ((size_t)grp * v_rows(N) + V_PAD) * 64;
    double *gD = p.D + (size_t)grp * dsize;

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

    // base rows (rho0) and derivative rows of the two sweep operators; [A'; B'] rows for the dual residual
    const size_t M = (size_t)W * KT;
    const double *Mf0 = p.ops + (size_t)r * KT, *Mb0 = p.ops + M + (size_t)r * KT;
    const double *Mt = p.adapt + (size_t)r * KT, *Pi = p.adapt + M + (size_t)r * KT, *dPi = p.adapt + 2 * M + (size_t)r * KT;
    const double *dMf = p.adapt + 3 * M + (size_t)r * KT, *dMb = p.adapt + 4 * M + (size_t)r * KT;
    const double dpnref = p.adapt[5 * M + r];
    const double rho0 = p.rho;
    double rho = inst_ok ? p.rho_inst[inst] : rho0;  // persists across solves like cache->rho
    // 64 lanes per instance: three operator rows of 64 doubles are 384 VGPRs and, with the adaptation's Pinf row, more than a
    // wavefront has -- the kernel spilled 200 of them. A sweep needs ONE operator (Mf forward, Mb backward; [A'; B'] only in the
    // sweeps that adapt), so RELOAD keeps one row array and rebuilds it from L2 at the start of each sweep (layout D does the
    // same from LDS); the narrower forms keep all three resident.
    constexpr bool RELOAD = W == 64;
    double mf[KT], mb_store[RELOAD ? 1 : KT], mt[KT];
    double (&mb)[KT] = *reinterpret_cast<double (*)[KT]>(RELOAD ? &mf[0] : &mb_store[0]);  // (RELOAD: the one array, Mb during the backward sweep)
    auto load_forward = [&](double delta) {
#pragma unroll
        for (int k = 0; k < KT; ++k) mf[k] = fma(delta, dMf[k], Mf0[k]);
    };
    auto load_backward = [&](double delta) {
#pragma unroll
        for (int k = 0; k < KT; ++k) mb[k] = fma(delta, dMb[k], Mb0[k]);
    };
    auto load_operators = [&](double delta) {
        if constexpr (!RELOAD) {
            load_forward(delta);
            load_backward(delta);
        }
    };
    load_operators(rho - rho0);
    if constexpr (!RELOAD) {
#pragma unroll
        for (int k = 0; k < KT; ++k) mt[k] = Mt[k];
    }
    const double cf = p.ops[2 * M + r];
    const double cb = p.ops[2 * M + W + r];
    const double dgr = p.ops[2 * M + 2 * W + r];  // Q + rho0 / R + rho0 diagonal of this row (tiny_api.cpp:90-91)
    const double pnref0 = p.tables[(size_t)3 * TOFF + r];
    double pnref = fma(rho - rho0, dpnref, pnref0);
    const double x0v = (inst_ok && is_x) ? p.x0[inst * nx + r] : 0.0;
    if (p.x0_mirror && inst_ok && is_x) p.x0_mirror[inst * nx + r] = x0v;  // zero-copy tick: x0 came from host memory
    const int dIdx = is_u ? (j * nu + (r - nx)) : 0;
    const int koff = is_x ? 1 : 0;
    const int ct = p.check_termination;
    __syncthreads();

    bool active = inst_ok;
    int it_done = 0;
    int status = 11;
    bool res_valid = false;
    double res_px = 0.0, res_dx = 0.0, res_pu = 0.0, res_du = 0.0;

    for (int it = 0; it < p.max_iter; ++it) {
        if (__ballot(active) == 0ull) break;
        const bool check = (ct > 0) && (((it + 1) % ct) == 0);
        const bool adapt = (it > 0) && (it % 5 == 0);  // admm.cpp:155
        const bool st = active && row_ok;
        double pri, dua;
        double a_pr = 0.0, a_pn = 0.0, a_dr = 0.0, a_dn = 0.0;  // adaptation: primal res / norm, dual res / norm
        double x_last = x0v, g_last = 0.0;

        if constexpr (RELOAD) {
            load_forward(rho - rho0);
            if (adapt) {  // (uniform)
#pragma unroll
                for (int k = 0; k < KT; ++k) mt[k] = Mt[k];
            }
        }
        {   // knot 0, state lanes
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
            const double *pg = sG + (1 + koff) * 64 + lane;
            const double *pt = t_lo + (1 + koff) * W + r;
            const double *pd = sD + dIdx;
            double *ps = sG + (st ? (1 + koff) * 64 + lane : ldummy);
            double *pgv = gV + (st ? koff * 64 + lane : gdummy);
            const int inc = st ? 64 : 0;
            double xcur = x0v;
            double gprev = 0.0;  // g_i of the state rows; the x_0 column has no -g_0 term (y_vector starts at g_1)
            FwdOps cur{pg[0], pg[VOFF], pt[0], pt[TOFF], pd[0]};
            for (int i = 0; i < N - 1; ++i) {
                const double w = is_x ? xcur : cur.dv;
                pg += 64;
                pt += W;
                pd += dstride;
                const FwdOps nxt{pg[0], pg[VOFF], pt[0], pt[TOFF], pd[0]};
                const double out = group_matvec<W, KT>(mf, w, cf);
                double gnew, snew;
                project_element(out, cur.g, cur.lo, cur.hi, cur.vold, gnew, snew, pri, dua);
                if (check) *pgv = cur.vold;
                ps[0] = gnew;
                ps[VOFF] = snew;
                ps += inc;
                pgv += inc;
                if (adapt) {
                    // state lanes: column x_i (xcur, gprev) and row vnew_{i+1} (snew); input lanes: column / row u_i
                    const double t = group_matvec<W, KT>(mt, is_x ? gnew : 0.0, 0.0);  // [A'; B'] g_{i+1}
                    const double dgx = dgr * (is_x ? xcur : out);                       // Q.*x_i | R.*u_i  (= P x and q entries)
                    const double aty = is_x ? (t - gprev) : (gnew + t);
                    a_dn = fmax(amax2(a_dn, dgx), fabs(aty));
                    a_dr = amax2(a_dr, 2.0 * dgx + aty);
                    a_pr = amax2(a_pr, is_x ? snew : (out - snew));
                    a_pn = fmax(amax2(a_pn, snew), is_x ? 0.0 : fabs(out));
                }
                gprev = gnew;
                xcur = out;
                cur = nxt;
            }
            x_last = xcur;   // x_{N-1} on state lanes
            g_last = gprev;  // g_{N-1}
        }
        if (active) it_done = it + 1;

        // ---------------- adaptive rho (admm.cpp:147-174)
        const double rho_lin = rho, pnref_lin = pnref;  // what update_linear_cost used this iteration
        if (adapt) {
            // column block x_{N-1}: Pinf x + Q.*x - g_{N-1} with the CURRENT (adapted) Pinf (rho_benchmark.cpp:112)
            double pr[KT];
            const double delta = rho - rho0;
#pragma unroll
            for (int k = 0; k < KT; ++k) pr[k] = fma(delta, dPi[k], Pi[k]);
            const double px = group_matvec<W, KT>(pr, is_x ? x_last : 0.0, 0.0);
            if (is_x) {
                const double qv = dgr * x_last;
                a_dn = fmax(fmax(amax2(a_dn, px), fabs(qv)), fabs(g_last));
                a_dr = amax2(a_dr, px + qv - g_last);
            }
            const double pri_res = group_max<W>(a_pr), pri_norm = group_max<W>(a_pn);
            const double dual_res = group_max<W>(a_dr), dual_norm = group_max<W>(a_dn);
            const double eps = 1e-10;  // rho_benchmark.cpp:190-197
            const double normalized_pri = pri_res / (pri_norm + eps);
            const double normalized_dual = dual_res / (dual_norm + eps);
            const double ratio = normalized_pri / (normalized_dual + eps);
            double new_rho = rho * sqrt(ratio);
            if (p.rho_clip) new_rho = fmin(fmax(new_rho, p.rho_min), p.rho_max);
            if (active) {
                rho = new_rho;
                pnref = fma(rho - rho0, dpnref, pnref0);
            }
            load_operators(rho - rho0);  // rho is unchanged for instances that are no longer active
        }

        if (check) {
            const double px = group_max<W>(is_x ? pri : 0.0);
            const double pu = group_max<W>(is_u ? pri : 0.0);
            const double dx = group_max<W>(is_x ? dua : 0.0) * rho;  // cache->rho AFTER the adaptation (admm.cpp:95-96)
            const double du = group_max<W>(is_u ? dua : 0.0) * rho;
            if (active) {
                res_px = px; res_dx = dx; res_pu = pu; res_du = du;
                res_valid = true;
                if (px < p.abs_pri_tol && pu < p.abs_pri_tol && dx < p.abs_dua_tol && du < p.abs_dua_tol) {
                    status = 1;
                    active = false;
                }
            }
        }

        {   // backward sweep: linear cost with the rho / Pinf of update_linear_cost, operators with the new Kinf
            if constexpr (RELOAD) load_backward(rho - rho0);
            const bool stb = active && row_ok && is_u;
            const double *pb = sG + N * 64 + lane;
            double pcur = pnref_lin - rho_lin * (pb[VOFF] - pb[0]);
            pb -= 64;
            const double *pl = t_lr + (N - 1) * W + r;
            double *pdst = sD + (stb ? (N - 2) * dstride + dIdx : dsize + lane);
            const int ddec = stb ? dstride : 0;
            BwdOps cur{pb[0], pb[VOFF], pl[0]};
            for (int i = N - 2; i >= 0; --i) {
                const double lin = cur.blr - rho_lin * (cur.bv - cur.bg);
                const double w = is_x ? pcur : lin;
                pb -= 64;
                pl -= W;
                const BwdOps nxt{pb[0], pb[VOFF], pl[0]};
                const double out = group_matvec<W, KT>(mb, w, cb);
                *pdst = out;
                pdst -= ddec;
                pcur = lin + out;
                cur = nxt;
            }
        }
    }

    if (p.max_iter > 0 && inst_ok) {
        for (int kn = 0; kn < N; ++kn) {
            const int e = (kn + 1) * 64 + lane;
            gG[kn * 64 + lane] = sG[e];
            if (status != 1) gV[kn * 64 + lane] = sV[e];
            const double sol = sV[e];
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
        p.rho_inst[inst] = rho;
        if (res_valid) {
            p.dstats[inst * 4 + 0] = res_px;
            p.dstats[inst * 4 + 1] = res_dx;
            p.dstats[inst * 4 + 2] = res_pu;
            p.dstats[inst * 4 + 3] = res_du;
        }
    }
}

template <int W, int KT>
static hipError_t launch_adapt_t(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    constexpr int IPW = 64 / W;
    const int groups = (p.batch + IPW - 1) / IPW;
    static size_t lds_set_t[16] = {0}, lds_set_f[16] = {0};
    hipError_t e;
    if (p.scratch) {
        hipLaunchKernelGGL((k_admm_solve_adapt<W, KT, false, true>), dim3(groups), dim3(64), 0, stream, p);
    } else if (p.tables_in_lds) {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve_adapt<W, KT, true, false>), lds_bytes, lds_set_t);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_admm_solve_adapt<W, KT, true, false>), dim3(groups), dim3(64), lds_bytes, stream, p);
    } else {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve_adapt<W, KT, false, false>), lds_bytes, lds_set_f);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_admm_solve_adapt<W, KT, false, false>), dim3(groups), dim3(64), lds_bytes, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_solve_adapt(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream) {
    if (!p.adapt || !p.rho_inst) return hipErrorInvalidValue;
    if (W == 16 && KT == 8) return launch_adapt_t<16, 8>(p, lds_bytes, stream);
    if (W == 16 && KT == 12) return launch_adapt_t<16, 12>(p, lds_bytes, stream);
    if (W == 16 && KT == 16) return launch_adapt_t<16, 16>(p, lds_bytes, stream);
    if (W == 32 && KT == 32) return launch_adapt_t<32, 32>(p, lds_bytes, stream);
    if (W == 64 && KT == 64) return launch_adapt_t<64, 64>(p, lds_bytes, stream);
    return hipErrorInvalidValue;
}

}  // namespace tinympc
