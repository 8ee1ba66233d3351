// tinympc_solve_b.hip -- k_admm_solve_b: the solve kernel in "layout B" (throughput layout), gfx950 FP64.
//
// Same algorithm, lane layout and arithmetic as k_admm_solve (tinympc_solve.hip, layout A; see there for
// the reference citations). What changes is WHERE the slack array V (v|z) lives, to double the number of
// resident wavefronts per CU:
//
//   layout A   1 wave / workgroup, G + V + D + tables in LDS: 80 KB  -> 2 waves per CU (2 of 4 SIMDs idle)
//   layout B   4 waves / workgroup, per wave G + D in LDS (33 KB), the tables once per workgroup (20 KB):
//              154 KB -> 4 waves per CU, one per SIMD. V lives in HBM as a ping-pong PAIR that stays
//              L2 / Infinity-Cache resident: sweep k reads V[(k-1)&1] (= vold) and writes V[k&1] (= vnew).
//
// The pair also gives the reference's warm-start semantics for free: a converged solve returns before
// v <- vnew (admm.cpp:181-197), i.e. its workspace keeps the PREVIOUS iteration's v/z -- which is simply
// the other buffer. (Layout A has to stream the old value out on every check iteration for this.)
//
// V traffic is off the serial chain and is prefetched four sweep steps ahead into a 4-register ring
// (the sweeps are unrolled by four so the ring rotates without moves); the arrays carry V_PAD rows at
// both ends so the prefetch never needs clamping. The values a sweep needs right at its start are
// never waited for: the backward sweep takes the last forward steps' vnew straight from registers, and
// the operands of the next forward sweep's first steps are requested at the start of the backward sweep.
#include <type_traits>

#include "tinympc_device.h"
#include "tinympc_sweep.h"

namespace tinympc {

constexpr int WPG = WAVES_PER_GROUP_B;

// LDS plan per workgroup, in doubles: tables (shared) | per wave: G[N+2][64], D[(N-1)*IPW*nu + 64]
static __host__ __device__ inline size_t b_tables_doubles(int W, int N) { return (tables_doubles(W, N) + 1) & ~(size_t)1; }
static __host__ __device__ inline size_t b_wave_doubles(int nu, int N, int W) {
    return (size_t)(N + 2) * 64 + ((((size_t)(N - 1) * (64 / W) * nu + 64) + 1) & ~(size_t)1);
}
size_t solve_b_lds_bytes(int nx, int nu, int N, int W) {
    (void)nx;
    return (b_tables_doubles(W, N) + WPG * b_wave_doubles(nu, N, W)) * sizeof(double);
}

#ifndef TINY_B_INPLACE
#define TINY_B_INPLACE 1  // 1: V updated in place + conditional stale copy; 0: unconditional ping-pong pair
#endif

// `bad` = ballot of lanes whose row already rules out convergence in this sweep, `live` = ballot of the
// lanes that belong to a still-active instance. True if some live instance has no bad lane.
template <int W>
__device__ __forceinline__ bool wave_may_converge(unsigned long long bad, unsigned long long live) {
    constexpr unsigned long long ones = (W == 64) ? ~0ull : ((1ull << (W % 64)) - 1ull);
    bool any = false;
#pragma unroll
    for (int j = 0; j < 64 / W; ++j) {
        const unsigned long long b = (bad >> (j * W)) & ones, l = (live >> (j * W)) & ones;
        any = any || (l != 0ull && b == 0ull);
    }
    return any;
}

struct FwdLds { double g, lo, hi, dv; };
struct BwdLds { double bg, blr; };

// R4 = (N-1) & 3 is a template parameter so that the rotation of the prefetch rings is resolved at
// compile time: with a run-time rotation hipcc demotes the ring to scratch memory (pointer table).
// CT ("constant tables"): bounds and references do not vary over the horizon (p.const_tables, decided on the host from
// what the verbs received) -- the per-knot table entries lo / hi / linref of a lane are then the same at every
// knot and live in three registers instead of being fetched from LDS at every sweep step.
template <int W, int KT, int R4, bool CT>
__global__ void __launch_bounds__(64 * WPG) k_admm_solve_b(const SolveParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    refresh_reference_tables(p, W, KT);  // references handed over in pinned host memory (single-instance handles)
    constexpr int IPW = 64 / W;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane / W, r = lane % W;
    const int nx = p.nx, nu = p.nu, N = p.N, nxu = nx + nu;
    const long grp = (long)blockIdx.x * WPG + wv;
    const bool grp_ok = grp < p.groups;
    const long inst = grp * IPW + j;
    const bool is_x = r < nx;
    const bool is_u = (r >= nx) && (r < nxu);
    const bool inst_ok = grp_ok && inst < p.batch;
    const bool row_ok = inst_ok && (r < nxu);
    const int dstride = IPW * nu;
    const int dsize = (N - 1) * dstride;
    const int TOFF = (int)table_rows(N) * W;
    const int ldummy = (N + 1) * 64 + lane;  // G dummy slot (LDS row N+1)
    const int vdummy = N * 64 + lane;        // V dummy slot, relative to knot 0 (HBM row N + V_PAD)
    const int nsteps = N - 1;

    double *sT = smem;
    double *sG = smem + b_tables_doubles(W, N) + (size_t)wv * b_wave_doubles(nu, N, W);
    double *sD = sG + (size_t)(N + 2) * 64;
    const double *t_lo = sT, *t_lr = sT + 2 * TOFF;

    // ---- tables: once per workgroup, all four waves
    {
        const int tn = (int)tables_doubles(W, N);
        for (int i = threadIdx.x; i < tn; i += 64 * WPG) sT[i] = p.tables[i];
    }
    double *gG = p.G + (size_t)(grp_ok ? grp : 0) * (N + 1) * 64;
    double *gD = p.D + (size_t)(grp_ok ? grp : 0) * dsize;
    double *const gV0 = p.V + ((size_t)(grp_ok ? grp : 0) * v_rows(N) + V_PAD) * 64;   // knot 0 of buffer 0 (canonical)
    double *const gV1 = p.V2 + ((size_t)(grp_ok ? grp : 0) * v_rows(N) + V_PAD) * 64;  // knot 0 of buffer 1
    if (grp_ok) {
        for (int kn = 0; kn < N; ++kn) sG[(kn + 1) * 64 + lane] = gG[kn * 64 + lane];
        sG[lane] = 0.0;
        sG[ldummy] = 0.0;
        for (int i = lane; i < dsize; i += 64) sD[i] = gD[i];
        sD[dsize + lane] = 0.0;
    }
    __syncthreads();  // the only workgroup-wide barrier: from here on the four waves are independent
    if (!grp_ok) return;

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
    const double lo_c = p.tables[W + r], hi_c = p.tables[(size_t)TOFF + W + r], lr_c = p.tables[(size_t)2 * TOFF + W + r];  // knot 0 (CT)
    // The forward sweep runs its R4 = nsteps % 4 odd steps FIRST, then whole groups of four.

    bool active = inst_ok;
    int it_done = 0;
    int status = 11;  // TINY_UNSOLVED (admm.cpp:114)
    bool res_valid = false;
    double snap_pri = 0.0, snap_dua = 0.0;  // this lane's residual maxima at its instance's last termination check

    // vold operands of the next forward sweep's first steps: knot 0 (state lanes) and the 4-slot ring.
    // Slot of forward step i is (i - R4) & 3, so that the sweep always ENDS on slot 3.
    double vk0, v0, v1, v2, v3;
    auto request_forward_head = [&](const double *X) {  // X = knot 0 of the buffer the next sweep reads
        const double *q = X + koff * 64 + lane;
        vk0 = X[lane];
        if constexpr (R4 == 0) { v0 = q[0]; v1 = q[64]; v2 = q[128]; v3 = q[192]; }
        else if constexpr (R4 == 1) { v3 = q[0]; v0 = q[64]; v1 = q[128]; v2 = q[192]; }
        else if constexpr (R4 == 2) { v2 = q[0]; v3 = q[64]; v0 = q[128]; v1 = q[192]; }
        else { v1 = q[0]; v2 = q[64]; v3 = q[128]; v0 = q[192]; }
    };
    request_forward_head(gV0);

    for (int it = 0; it < p.max_iter; ++it) {  // admm.cpp:129
        if (__ballot(active) == 0ull) break;
        const bool check = (ct > 0) && (((it + 1) % ct) == 0);  // admm.cpp:91 (iter already incremented, :143)
        const bool st = active && row_ok;
#if TINY_B_INPLACE
        double *const Vr = gV0, *const Vw = gV0;  // V is updated in place; gV1 receives the stale copy (below)
#else
        double *const Vr = (it & 1) ? gV1 : gV0;  // sweep k = it+1 reads V[(k-1)&1] ...
        double *const Vw = (it & 1) ? gV0 : gV1;  // ... and writes V[k&1]
#endif
        double pri, dua;
        double w0, w1, w2, w3;  // vnew of the last four forward steps (w3 = last), consumed by the backward sweep

        // ---------------- forward sweep (F1) with S1+D1+R1 fused in
        {   // knot 0, state lanes only: x_0 is given (tiny_set_x0), no mat-vec
            const bool on = st && is_x;
            const double g = sG[64 + lane];
            const double s = x0v + g;
            const double snew = fmin(t_lo[TOFF + W + r], fmax(t_lo[W + r], s));
            pri = is_x ? fabs(x0v - snew) : 0.0;
            dua = is_x ? fabs(vk0 - snew) : 0.0;
            sG[on ? 64 + lane : ldummy] = s - snew;
#if TINY_B_INPLACE
            if (check) gV1[on ? lane : vdummy] = vk0;
#endif
            Vw[on ? lane : vdummy] = snew;
        }
        {
#if TINY_B_INPLACE
            // Stale copy (the reference's v/z after a converged solve = the PREVIOUS iterate): needed only
            // if this very sweep ends converged, i.e. only while some instance of the wave still has all
            // its rows' running residual maxima below the tolerances (maxima only grow, so the test is exact).
            // In iterations that cannot converge the copy stops after the first group of four steps. The
            // decision is taken once per group of four steps and selects one of two copies of the group
            // body, so the common no-copy path carries no per-step overhead.
            const long sdelta = gV1 - gV0;  // wave-uniform distance (in doubles) from V to its stale copy
            bool may = check;
#else
            constexpr bool may = false;
#endif
            const double *pg = sG + (1 + koff) * 64 + lane;
            const double *pt = t_lo + (1 + koff) * W + r;
            const double *pd = sD + dIdx;
            const double *pvr = Vr + koff * 64 + lane;  // this lane's row of the current step in the read buffer
            double *ps = sG + (st ? (1 + koff) * 64 + lane : ldummy);
            double *pvw = Vw + (st ? koff * 64 + lane : vdummy);
            const int inc = st ? 64 : 0;
            double xcur = x0v;
            FwdLds A, B;
            auto fstep = [&](auto stale, const FwdLds &cur, FwdLds &nxt, double &vslot, double &wslot) {
                const double w = is_x ? xcur : cur.dv;
                pg += 64;
                pt += W;
                pd += dstride;
                nxt.g = pg[0]; nxt.dv = pd[0];
                if constexpr (!CT) { nxt.lo = pt[0]; nxt.hi = pt[TOFF]; }
                const double vold = vslot;
                vslot = pvr[4 * 64];  // vold of the step four ahead (same ring slot)
                pvr += 64;
                const double out = group_matvec<W, KT>(mf, w, cf);  // state lanes: x_{i+1}; input lanes: u_i
                double gnew;
                project_element(out, cur.g, CT ? lo_c : cur.lo, CT ? hi_c : cur.hi, vold, gnew, wslot, pri, dua);
                ps[0] = gnew;
                *pvw = wslot;
#if TINY_B_INPLACE
                if constexpr (decltype(stale)::value) pvw[sdelta] = vold;
#endif
                ps += inc;
                pvw += inc;
                xcur = out;
            };
            constexpr std::true_type with_copy{};
            constexpr std::false_type no_copy{};
            // Head: the R4 odd steps, on the ring slots that make the sweep end on slot 3.
            A.g = pg[0]; A.lo = pt[0]; A.hi = pt[TOFF]; A.dv = pd[0];  // (lo / hi unused under CT)
            if constexpr (R4 & 1) B = A;  // an odd head starts with B as the current operand set
            if (may) {
                if constexpr (R4 == 3) { fstep(with_copy, B, A, v1, w1); fstep(with_copy, A, B, v2, w2); fstep(with_copy, B, A, v3, w3); }
                if constexpr (R4 == 2) { fstep(with_copy, A, B, v2, w2); fstep(with_copy, B, A, v3, w3); }
                if constexpr (R4 == 1) { fstep(with_copy, B, A, v3, w3); }
            } else {
                if constexpr (R4 == 3) { fstep(no_copy, B, A, v1, w1); fstep(no_copy, A, B, v2, w2); fstep(no_copy, B, A, v3, w3); }
                if constexpr (R4 == 2) { fstep(no_copy, A, B, v2, w2); fstep(no_copy, B, A, v3, w3); }
                if constexpr (R4 == 1) { fstep(no_copy, B, A, v3, w3); }
            }
            // Two loops rather than one loop with a branch in it: a branch inside the body makes hipcc drain
            // every outstanding memory operation (vmcnt(0)) at the loop latch, which exposes the full L2
            // latency of the prefetch ring once per group. Loop 1 copies and re-tests; loop 2 finishes.
            int i = R4;
#if TINY_B_INPLACE
            while (may && i < nsteps) {
                fstep(with_copy, A, B, v0, w0);
                fstep(with_copy, B, A, v1, w1);
                fstep(with_copy, A, B, v2, w2);
                fstep(with_copy, B, A, v3, w3);
                i += 4;
                const bool bad = st && !((pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol));
                may = wave_may_converge<W>(__ballot(bad), __ballot(st));
            }
#endif
            for (; i < nsteps; i += 4) {
                fstep(no_copy, A, B, v0, w0);
                fstep(no_copy, B, A, v1, w1);
                fstep(no_copy, A, B, v2, w2);
                fstep(no_copy, B, A, v3, w3);
            }
        }
        if (active) it_done = it + 1;  // admm.cpp:143

        // ---------------- backward-sweep operands that are not in registers: request them now
        // (vnew of knots N-5..N-8 from the buffer just written) together with the head of the NEXT forward sweep.
        double b0, b1, b2, b3;
        {
            const double *q = Vw + (N - 5) * 64 + lane;
            b0 = q[0]; b1 = q[-64]; b2 = q[-128]; b3 = q[-192];
        }
        request_forward_head(Vw);

        // ---------------- R1: termination (admm.cpp:93-101). "All four inf-norms below tolerance" is decided element-
        // wise (max_i a_i < tol iff every a_i < tol; scaling by rho > 0 is monotone): one ballot instead of four
        // shuffle butterflies per iteration. The norms themselves (for get_stats) are reduced once, after the loop,
        // from the snapshot each lane takes at its instance's last check.
        if (check) {
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
            const double *pb = sG + N * 64 + lane;        // G of knot N-1
            double pcur = pnref - rho * (w3 - pb[0]);     // p_{N-1}, admm.cpp:81-82 (state lanes: w3 = vnew_{N-1})
            pb -= 64;                                      // knot N-2
            const double *pl = t_lr + (N - 1) * W + r;
            const double *pvb = Vw + (N - 5) * 64 + lane;  // row of the first ring-served knot
            double *pdst = sD + (stb ? (N - 2) * dstride + dIdx : dsize + lane);
            const int ddec = stb ? dstride : 0;
            BwdLds A{pb[0], pl[0]}, B;
            auto bcore = [&](const BwdLds &cur, BwdLds &nxt, double bv) {
                const double lin = (CT ? lr_c : cur.blr) - rho * (bv - cur.bg);  // q_i (state lanes) / r_i (input lanes), admm.cpp:77-80
                const double w = is_x ? pcur : lin;
                pb -= 64;
                pl -= W;
                nxt.bg = pb[0];
                if constexpr (!CT) nxt.blr = pl[0];
                const double out = group_matvec<W, KT>(mb, w, cb);
                *pdst = out;  // d_i (input lanes)
                pdst -= ddec;
                pcur = lin + out;  // p_i (state lanes)
            };
            auto bstep = [&](const BwdLds &cur, BwdLds &nxt, double &slot) {
                const double bv = slot;
                slot = pvb[-4 * 64];  // vnew of the knot four further down (same ring slot)
                pvb -= 64;
                bcore(cur, nxt, bv);
            };
            // knots N-2, N-3, N-4: vnew still in registers from the forward sweep. A state lane finished knot k
            // at forward step k-1, an input lane at step k, hence the per-lane-type select.
            bcore(A, B, is_x ? w2 : w3);
            bcore(B, A, is_x ? w1 : w2);
            bcore(A, B, is_x ? w0 : w1);
            // knots N-5 .. 0 from the ring
            int i = N - 5;
            for (; i >= 3; i -= 4) {
                bstep(B, A, b0);
                bstep(A, B, b1);
                bstep(B, A, b2);
                bstep(A, B, b3);
            }
            constexpr int RB = (R4 + 1) & 3;  // (N - 4) % 4 ring-served knots left over
            if constexpr (RB >= 1) bstep(B, A, b0);
            if constexpr (RB >= 2) bstep(A, B, b1);
            if constexpr (RB >= 3) bstep(B, A, b2);
            (void)i;
        }
    }

    const double res_px = group_max<W>(is_x ? snap_pri : 0.0), res_pu = group_max<W>(is_u ? snap_pri : 0.0);
    const double res_dx = group_max<W>(is_x ? snap_dua : 0.0) * rho, res_du = group_max<W>(is_u ? snap_dua : 0.0) * rho;

    // ---- write-back. k = it_done sweeps ran for this instance: vnew is in V[k&1], the previous iterate in
    // V[(k-1)&1]. The canonical buffer (0) must end up holding the reference's workspace v/z:
    //   not converged (max_iter hit): v = vnew (admm.cpp:196-197)  -> copy if vnew sits in buffer 1
    //   converged: the solve returned before v <- vnew              -> copy if the previous iterate sits in buffer 1
    // (In-place variant: vnew is always in buffer 0; a converged instance's previous iterate is the stale
    //  copy in buffer 1, which then becomes the canonical v/z.)
    if (p.max_iter > 0 && inst_ok && it_done > 0) {
#if TINY_B_INPLACE
        const double *last = gV0;
        const bool copy_1_to_0 = (status == 1);
#else
        const bool last_is_1 = (it_done & 1) != 0;
        const double *last = last_is_1 ? gV1 : gV0;
        const bool copy_1_to_0 = (status == 1) ? !last_is_1 : last_is_1;
#endif
        for (int kn = 0; kn < N; ++kn) {
            const int e = kn * 64 + lane;
            gG[e] = sG[e + 64];
            const double sol = last[e];  // solution = vnew / znew (admm.cpp:187-188, 204-205)
            if (copy_1_to_0) gV0[e] = gV1[e];
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

template <int W, int KT, int R4, bool CT>
static hipError_t launch_b_c(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    const int wgs = (p.groups + WPG - 1) / WPG;
    static size_t lds_set[16] = {0};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve_b<W, KT, R4, CT>), lds_bytes, lds_set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_admm_solve_b<W, KT, R4, CT>), dim3(wgs), dim3(64 * WPG), lds_bytes, stream, p);
    return hipGetLastError();
}

template <int W, int KT, int R4>
static hipError_t launch_b_r(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    return p.const_tables ? launch_b_c<W, KT, R4, true>(p, lds_bytes, stream) : launch_b_c<W, KT, R4, false>(p, lds_bytes, stream);
}

template <int W, int KT>
static hipError_t launch_b_t(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    switch ((p.N - 1) & 3) {
        case 0: return launch_b_r<W, KT, 0>(p, lds_bytes, stream);
        case 1: return launch_b_r<W, KT, 1>(p, lds_bytes, stream);
        case 2: return launch_b_r<W, KT, 2>(p, lds_bytes, stream);
        default: return launch_b_r<W, KT, 3>(p, lds_bytes, stream);
    }
}

hipError_t launch_solve_b(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream) {
    if (p.N < 8) return hipErrorInvalidValue;
    if (W == 16 && KT == 8) return launch_b_t<16, 8>(p, lds_bytes, stream);
    if (W == 16 && KT == 12) return launch_b_t<16, 12>(p, lds_bytes, stream);
    if (W == 16 && KT == 16) return launch_b_t<16, 16>(p, lds_bytes, stream);
    return hipErrorInvalidValue;
}

}  // namespace tinympc
