// tinympc_solve_fam.hip -- k_admm_solve_fam: layout-A solve kernel + the second-order-cone and
// linear-inequality slack families (BASELINE config 4).
//
// PARITY UNPINNED. The reference tree has no source for these families (they live in the newer
// TinyMPC/TinyMPC core behind the calls at /root/reference/src/bindings.cpp:408-478); this kernel
// implements the same restatement of the upstream algorithm as the CPU checker under oracle/ and is
// tested against it and against first principles (cone / half-space feasibility).
//
// On top of the box family (g|y, v|z -- identical to k_admm_solve, in LDS) every row carries
//   cone family    dual gc|yc, slack vcnew|zcnew = projection of (x|u) + (gc|yc) onto the row's cone
//                  { (w, t) : ||w||_2 <= mu * t } (rows in no cone are left as they are);
//   linear family  dual gl|yl, slack vlnew|zlnew = (x|u) + (gl|yl) pushed through the half-spaces
//                  a_k' s <= b_k one after another;
// the slacks are transient, the duals persist (HBM, prefetched one step ahead), and the linear cost gets
//   q_i (r_i) -= rho * (vcnew - gc) + rho * (vlnew - gl)      ("Lx", stored in HBM, forward -> backward).
// Termination still looks at the box family only, as upstream does.
//
// Cross-row quantities reuse the fused DPP mat-vec with 0/1 mask rows instead of a shuffle butterfly:
//   ||w||^2 = Cn_row . s.^2     t = Ct_row . s     a_k' s = Ty_row . (a_k .* s)
// so any number of pairwise-disjoint cones costs two mat-vecs per step, and each linear row one.
#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {

struct FamFwd { double g, vold, lo, hi, dv, gc, gl; };
struct FamBwd { double bg, bv, blr, lx; };

// GMEM: the working copy of the state lives in p.scratch (HBM) instead of LDS -- the fallback for horizons
// that do not fit 160 KB of LDS. Same code, same results; the row-local operands then come from L2.
template <int W, int KT, bool TLDS, bool GMEM = false>
__global__ void __launch_bounds__(64) k_admm_solve_fam(const SolveParams p) {
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
    const size_t vbase = ((size_t)grp * v_rows(N) + V_PAD) * 64;  // knot 0 in the padded HBM layout
    double *gV = p.V + vbase, *gGC = p.GC + vbase, *gGL = p.GL + vbase, *gLX = p.LX + vbase;
    double *gD = p.D + (size_t)grp * dsize;

    for (int kn = 0; kn < N; ++kn) {
        sG[(kn + 1) * 64 + lane] = gG[kn * 64 + lane];
        sV[(kn + 1) * 64 + lane] = gV[kn * 64 + lane];
    }
    sG[lane] = 0.0; sV[lane] = 0.0; sG[ldummy] = 0.0; sV[ldummy] = 0.0;
    for (int i = lane; i < dsize; i += 64) sD[i] = gD[i];
    sD[dsize + lane] = 0.0;
    if (TLDS) {
        const int tn = (int)tables_doubles(W, N);
        for (int i = lane; i < tn; i += 64) sT[i] = p.tables[i];
    }

    // RED (wide systems, 32 / 64 lanes per instance): the cross-row quantities come from group REDUCTIONS instead of mat-vecs with
    // 0/1 mask rows. Three mask rows of KT doubles per lane next to the two operator rows are 640 VGPRs at 64 lanes: the kernel
    // lived in scratch (nx=48, nu=16 with one cone and two linear rows: 2.0 M iterations/s against 131 M on the box path). What
    // the masks encode is small: per round a lane's cone is known by its LAST row (the t entry, Ct's one column), its role says
    // whether it belongs to the tail; a cone's ||w||^2 is one group sum over its tail rows, t one lane read, a linear row's a'x and
    // a'u two group sums.
    constexpr bool RED = W > 16;
    double mf[KT], mb[KT], cn[RED ? 1 : KT], ct[RED ? 1 : KT], ty[RED ? 1 : KT];
    {
        const double *Mf = p.ops + (size_t)r * KT, *Mb = p.ops + (size_t)W * KT + (size_t)r * KT;
        const double *Cn = p.fam + 4 * W + (size_t)r * KT, *Ct = Cn + (size_t)W * KT, *Ty = Ct + (size_t)W * KT;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            mf[k] = Mf[k]; mb[k] = Mb[k];
            if constexpr (!RED) { cn[k] = Cn[k]; ct[k] = Ct[k]; ty[k] = Ty[k]; }
        }
    }
    // RED: per round q -- this lane's role / slope / last row of its cone (-1: in no cone of the round), and the set of last rows
    // (= of cones) of the round as a wave-uniform bit mask of group-relative lane numbers
    int r_role[MAX_ROUNDS], r_head[MAX_ROUNDS];
    double r_mu[MAX_ROUNDS], r_imu[MAX_ROUNDS];
    unsigned long long r_cones[MAX_ROUNDS];
    if constexpr (RED) {
        const int nrounds = (int)p.fam[fam_nround_offset(W, KT)];
#pragma unroll
        for (int q = 0; q < MAX_ROUNDS; ++q) {
            r_role[q] = 0; r_head[q] = -1; r_mu[q] = 0.0; r_imu[q] = 0.0; r_cones[q] = 0ull;
            if (q < nrounds) {  // (uniform)
                const double *rd = q == 0 ? p.fam : p.fam + fam_round_offset(W, KT, q);
                const double *ctq = q == 0 ? p.fam + 4 * W + (size_t)W * KT : rd + 2 * W + (size_t)W * KT;
                r_role[q] = (int)rd[r];
                r_mu[q] = rd[W + r];
                r_imu[q] = (r_mu[q] != 0.0) ? 1.0 / r_mu[q] : 0.0;
                for (int k = 0; k < nxu; ++k)
                    if (ctq[(size_t)r * KT + k] != 0.0) r_head[q] = k;
                const unsigned long long heads = __ballot(r_head[q] == r);
                r_cones[q] = (W == 64) ? heads : (heads & ((1ull << (W % 64)) - 1ull));  // (every instance of the wave has the same cones)
            }
        }
    }
    const int role = (int)p.fam[r];
    const double mu = p.fam[W + r];
    const double inv_mu = (mu != 0.0) ? 1.0 / mu : 0.0;  // (mu = 0: row in no cone)
    const bool famc = p.fam[2 * W + r] != 0.0, faml = p.fam[3 * W + r] != 0.0;
    const double *lin = p.fam + 4 * W + (size_t)3 * W * KT;
    const int nl = (int)lin[0];
    // the first FAM_REG_ROWS rows' coefficients in registers; problems with more rows (an equality constraint of five rows is
    // ten) read the rest from the family buffer (L2) where they are used
    double ak[FAM_REG_ROWS], bk[FAM_REG_ROWS], ink[FAM_REG_ROWS];  // ink = 1 / ||a_k||^2
#pragma unroll
    for (int k = 0; k < (RED ? 0 : FAM_REG_ROWS); ++k) {
        ak[k] = lin[1 + (size_t)(3 * k + 0) * W + r];
        bk[k] = lin[1 + (size_t)(3 * k + 1) * W + r];
        ink[k] = 1.0 / lin[1 + (size_t)(3 * k + 2) * W + r];
    }
    // rounds of the cone list beyond the first (cones that share rows are projected one after another, as upstream does)
    const int nround = (int)p.fam[fam_nround_offset(W, KT)];
    // wave-uniform switches: is either family in use at all?
    const bool any_cone = __ballot(famc) != 0ull, any_lin = __ballot(faml) != 0ull;

    const double cf = p.ops[(size_t)2 * W * KT + r];
    const double cb = p.ops[(size_t)2 * W * KT + W + r];
    const double pnref = p.tables[(size_t)3 * TOFF + r];
    const double rho = p.rho;
    const double x0v = (inst_ok && is_x) ? p.x0[inst * nx + r] : 0.0;
    if (p.x0_mirror && inst_ok && is_x) p.x0_mirror[inst * nx + r] = x0v;  // zero-copy tick: x0 came from host memory
    const int dIdx = is_u ? (j * nu + (r - nx)) : 0;
    const int koff = is_x ? 1 : 0;
    const int ct_ = p.check_termination;
    __syncthreads();

    // The two extra families for one (row, knot) element with rollout value `val`: returns the row's
    // contribution to the linear cost and the new duals.
    auto families = [&](double val, double gc_old, double gl_old, double &gc_new, double &gl_new) -> double {
        double lx = 0.0;
        gc_new = gc_old;
        gl_new = gl_old;
        if constexpr (RED) {
            if (any_cone) {
                const double sv = val + gc_old;  // vcnew = x + gc (all rows of an enabled side)
                double vc = sv;
#pragma unroll
                for (int q = 0; q < MAX_ROUNDS; ++q) {
                    if (q < nround) {  // (uniform)
                        double a2 = 0.0;
                        for (unsigned long long m = r_cones[q]; m != 0ull; m &= m - 1ull) {  // one cone of the round after the other (uniform)
                            const int hc = __builtin_ctzll(m);
                            const bool mine = r_head[q] == hc;
                            const double tail2 = group_sum<W>((mine && r_role[q] == 1) ? vc * vc : 0.0);  // ||w||^2 of that cone
                            a2 = mine ? tail2 : a2;
                        }
                        const double t = __shfl(vc, r_head[q] >= 0 ? r_head[q] : r, W);  // last entry of the row's cone
                        vc = soc_project_element(vc, a2, t, r_mu[q], r_imu[q], r_role[q]);
                    }
                }
                const double gcn = sv - vc;  // gc + x - vcnew
                if (famc) {
                    gc_new = gcn;
                    lx -= rho * (vc - gcn);
                }
            }
            if (any_lin) {
                const double s0 = val + gl_old;
                double sv = s0;
#pragma unroll 1
                for (int k = 0; k < nl; ++k) {  // (uniform trip count; the rows' coefficients from the family buffer in L2)
                    const double a_k = lin[1 + (size_t)(3 * k + 0) * W + r], b_k = lin[1 + (size_t)(3 * k + 1) * W + r];
                    const double in_k = 1.0 / lin[1 + (size_t)(3 * k + 2) * W + r];
                    const double prod = a_k * sv;
                    const double dx = group_sum<W>(is_x ? prod : 0.0), du = group_sum<W>(is_u ? prod : 0.0);  // a_k' x | a_k' u
                    sv = halfspace_project_element(sv, is_x ? dx : du, a_k, b_k, in_k);
                }
                const double gln = s0 - sv;
                if (faml) {
                    gl_new = gln;
                    lx -= rho * (sv - gln);
                }
            }
            return lx;
        } else {
        if (any_cone) {
            const double sv = val + gc_old;                              // vcnew = x + gc (all rows of an enabled side)
            const double a2 = group_matvec<W, KT>(cn, sv * sv, 0.0);     // ||w||^2 of the row's cone
            const double t = group_matvec<W, KT>(ct, sv, 0.0);           // last entry of the row's cone
            double vc = soc_project_element(sv, a2, t, mu, inv_mu, role);
#pragma unroll 1
            for (int q = 1; q < nround; ++q) {                           // (uniform trip count; the masks of round q from L2)
                const double *rd = p.fam + fam_round_offset(W, KT, q);
                const int role_q = (int)rd[r];
                const double mu_q = rd[W + r];
                double cq[KT], tq[KT];
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    cq[k] = rd[2 * W + (size_t)r * KT + k];
                    tq[k] = rd[2 * W + (size_t)W * KT + (size_t)r * KT + k];
                }
                const double a2q = group_matvec<W, KT>(cq, vc * vc, 0.0);
                const double tq_ = group_matvec<W, KT>(tq, vc, 0.0);
                vc = soc_project_element(vc, a2q, tq_, mu_q, (mu_q != 0.0) ? 1.0 / mu_q : 0.0, role_q);
            }
            const double gcn = sv - vc;                                  // gc + x - vcnew
            if (famc) {
                gc_new = gcn;
                lx -= rho * (vc - gcn);
            }
        }
        if (any_lin) {
            const double s0 = val + gl_old;
            double sv = s0;
#pragma unroll
            for (int k = 0; k < FAM_REG_ROWS; ++k) {
                if (k < nl) {                                            // wave-uniform
                    const double dot = group_matvec<W, KT>(ty, ak[k] * sv, 0.0);
                    sv = halfspace_project_element(sv, dot, ak[k], bk[k], ink[k]);
                }
            }
#pragma unroll 1
            for (int k = FAM_REG_ROWS; k < nl; ++k) {                    // (uniform trip count)
                const double a_k = lin[1 + (size_t)(3 * k + 0) * W + r], b_k = lin[1 + (size_t)(3 * k + 1) * W + r];
                const double in_k = 1.0 / lin[1 + (size_t)(3 * k + 2) * W + r];
                const double dot = group_matvec<W, KT>(ty, a_k * sv, 0.0);
                sv = halfspace_project_element(sv, dot, a_k, b_k, in_k);
            }
            const double gln = s0 - sv;
            if (faml) {
                gl_new = gln;
                lx -= rho * (sv - gln);
            }
        }
        return lx;
        }
    };

    bool active = inst_ok;
    int it_done = 0;
    int status = 11;
    bool res_valid = false;
    double snap_pri = 0.0, snap_dua = 0.0;  // this lane's residual maxima at its instance's last termination check

    for (int it = 0; it < p.max_iter; ++it) {
        if (__ballot(active) == 0ull) break;
        const bool check = (ct_ > 0) && (((it + 1) % ct_) == 0);
        const bool st = active && row_ok;
        double pri, dua;

        // ---------------- forward sweep
        {   // knot 0, state lanes
            const bool on = st && is_x;
            const double g = sG[64 + lane], vold = sV[64 + lane];
            const double s = x0v + g;
            const double snew = fmin(t_lo[TOFF + W + r], fmax(t_lo[W + r], s));
            pri = is_x ? fabs(x0v - snew) : 0.0;
            dua = is_x ? fabs(vold - snew) : 0.0;
            double gcn, gln;
            const double lx = families(x0v, gGC[lane], gGL[lane], gcn, gln);
            if (check) gV[on ? lane : gdummy] = vold;
            sG[on ? 64 + lane : ldummy] = s - snew;
            sV[on ? 64 + lane : ldummy] = snew;
            gGC[on ? lane : gdummy] = gcn;
            gGL[on ? lane : gdummy] = gln;
            gLX[on ? lane : gdummy] = lx;
        }
        {
            const double *pg = sG + (1 + koff) * 64 + lane;
            const double *pt = t_lo + (1 + koff) * W + r;
            const double *pd = sD + dIdx;
            const double *pgc = gGC + koff * 64 + lane, *pgl = gGL + koff * 64 + lane;
            double *ps = sG + (st ? (1 + koff) * 64 + lane : ldummy);
            const int goff0 = st ? koff * 64 + lane : gdummy;
            double *pgv = gV + goff0, *pwc = gGC + goff0, *pwl = gGL + goff0, *pwx = gLX + goff0;
            const int inc = st ? 64 : 0;
            double xcur = x0v;
            FamFwd A{pg[0], pg[VOFF], pt[0], pt[TOFF], pd[0], pgc[0], pgl[0]}, B;
            auto fstep = [&](const FamFwd &cur, FamFwd &nxt) {
                const double w = is_x ? xcur : cur.dv;
                pg += 64; pt += W; pd += dstride; pgc += 64; pgl += 64;
                nxt.g = pg[0]; nxt.vold = pg[VOFF]; nxt.lo = pt[0]; nxt.hi = pt[TOFF]; nxt.dv = pd[0];
                nxt.gc = pgc[0]; nxt.gl = pgl[0];
                const double out = group_matvec<W, KT>(mf, w, cf);
                double gnew, snew, gcn, gln;
                project_element(out, cur.g, cur.lo, cur.hi, cur.vold, gnew, snew, pri, dua);
                const double lx = families(out, cur.gc, cur.gl, gcn, gln);
                if (check) *pgv = cur.vold;
                ps[0] = gnew;
                ps[VOFF] = snew;
                *pwc = gcn; *pwl = gln; *pwx = lx;
                ps += inc; pgv += inc; pwc += inc; pwl += inc; pwx += inc;
                xcur = out;
            };
            int i = 0;
            for (; i + 2 <= N - 1; i += 2) {
                fstep(A, B);
                fstep(B, A);
            }
            if (i < N - 1) fstep(A, B);
        }
        if (active) it_done = it + 1;

        // ---------------- residuals (box family only, as upstream)
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
                    status = 1;
                    active = false;
                }
            }
        }

        // ---------------- backward sweep
        {
            const bool stb = active && row_ok && is_u;
            const double *pb = sG + N * 64 + lane;
            const double *px_ = gLX + (N - 1) * 64 + lane;
            double pcur = pnref - rho * (pb[VOFF] - pb[0]) + px_[0];  // p_{N-1} incl. the family terms
            pb -= 64;
            px_ -= 64;
            const double *pl = t_lr + (N - 1) * W + r;
            double *pdst = sD + (stb ? (N - 2) * dstride + dIdx : dsize + lane);
            const int ddec = stb ? dstride : 0;
            FamBwd A{pb[0], pb[VOFF], pl[0], px_[0]}, B;
            auto bstep = [&](const FamBwd &cur, FamBwd &nxt) {
                const double lin_ = cur.blr - rho * (cur.bv - cur.bg) + cur.lx;
                const double w = is_x ? pcur : lin_;
                pb -= 64; pl -= W; px_ -= 64;
                nxt.bg = pb[0]; nxt.bv = pb[VOFF]; nxt.blr = pl[0]; nxt.lx = px_[0];
                const double out = group_matvec<W, KT>(mb, w, cb);
                *pdst = out;
                pdst -= ddec;
                pcur = lin_ + out;
            };
            int i = N - 2;
            for (; i >= 1; i -= 2) {
                bstep(A, B);
                bstep(B, A);
            }
            if (i == 0) bstep(A, B);
        }
    }

    // the four norms of the last check (for get_stats), reduced once
    const double res_px = group_max<W>(is_x ? snap_pri : 0.0), res_pu = group_max<W>(is_u ? snap_pri : 0.0);
    const double res_dx = group_max<W>(is_x ? snap_dua : 0.0) * rho, res_du = group_max<W>(is_u ? snap_dua : 0.0) * rho;

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
        if (res_valid) {
            p.dstats[inst * 4 + 0] = res_px;
            p.dstats[inst * 4 + 1] = res_dx;
            p.dstats[inst * 4 + 2] = res_pu;
            p.dstats[inst * 4 + 3] = res_du;
        }
    }
}

template <int W, int KT>
static hipError_t launch_fam_t(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    constexpr int IPW = 64 / W;
    const int groups = (p.batch + IPW - 1) / IPW;
    static size_t lds_set_t[16] = {0}, lds_set_f[16] = {0};
    hipError_t e;
    if (p.scratch) {  // state in HBM scratch, tables from global memory, no dynamic LDS at all
        hipLaunchKernelGGL((k_admm_solve_fam<W, KT, false, true>), dim3(groups), dim3(64), 0, stream, p);
    } else if (p.tables_in_lds) {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve_fam<W, KT, true>), lds_bytes, lds_set_t);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_admm_solve_fam<W, KT, true>), dim3(groups), dim3(64), lds_bytes, stream, p);
    } else {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve_fam<W, KT, false>), lds_bytes, lds_set_f);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_admm_solve_fam<W, KT, false>), dim3(groups), dim3(64), lds_bytes, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_solve_fam(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream) {
    if (W == 16 && KT == 8) return launch_fam_t<16, 8>(p, lds_bytes, stream);
    if (W == 16 && KT == 12) return launch_fam_t<16, 12>(p, lds_bytes, stream);
    if (W == 16 && KT == 16) return launch_fam_t<16, 16>(p, lds_bytes, stream);
    if (W == 32 && KT == 32) return launch_fam_t<32, 32>(p, lds_bytes, stream);
    if (W == 64 && KT == 64) return launch_fam_t<64, 64>(p, lds_bytes, stream);
    return hipErrorInvalidValue;
}

}  // namespace tinympc
