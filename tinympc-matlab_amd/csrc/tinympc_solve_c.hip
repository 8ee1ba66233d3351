// tinympc_solve_c.hip -- k_admm_solve_c ("layout C"): ONE MPC instance per workgroup, horizon cut into chunks
// that 16 lane groups sweep concurrently. The latency kernel: small batches and the single-instance solve.
//
// Layouts A/B give an instance 16 lanes and walk the N-1 steps of each Riccati-style sweep one after the other:
// 98 dependent mat-vecs per ADMM iteration, ~100 ns each, whatever the batch size. Both sweeps are linear
// time-invariant recurrences, so they can be cut in time:
//     forward   [x_{k+1}; u_k] = Mf [x_k; d_k] + cf                  (admm.cpp:25-35,  Phi = A - B*Kinf)
//     backward  [p_k;    d_k] = Mb [p_{k+1}; r_k] + [q_k;0] + cb     (admm.cpp:13-20,  Psi = AmBKt)
//   pass 1  every chunk c (S consecutive steps, one 16-lane group) runs its S steps from a ZERO incoming state
//           (chunk 0: from x_0; the last chunk of the backward sweep: from p_{N-1}) -- only the end value is kept;
//   carry   the true state after chunk c is  X_c = end_c + Phi^S X_{c-1}: a two-level scan over the C <= 16 chunks with
//           the precomputed powers Phi^S .. Phi^(4S). Inside a wavefront (4 chunks = its 4 DPP rows) the prefix is
//           formed with cross-row swaps -- no LDS, no barrier; the four wavefronts' totals cross through LDS behind ONE
//           barrier; a last mat-vec brings the incoming carry to every row. 5 mat-vecs and 1 barrier per sweep (the
//           first version: a 4-level Hillis-Steele scan over all 16 chunks, 4 mat-vecs but 4 barriers + 4 LDS exchanges
//           on the dependent path -- half of the iteration);
//   pass 2  the chunk is swept again from its true incoming state X_{c-1}: these are the sweep's real values.
// Depth per sweep: 2*S + log2(C) mat-vecs instead of N-1 (quadrotor N=50: 12 instead of 49). Passes 1 and 2 use
// the lane's rows of Mf / Mb, resident in registers; only the carry matrices come from LDS. (A variant that
// replaced pass 2 by S independent "fix-up" mat-vecs with precomputed [Phi^(i+1); -Kinf Phi^i] was slower: four
// wavefronts fetching 2 KB matrix rows per mat-vec saturate the CU's single 128 B/clk LDS port.)
// Everything row-local (slack projection, dual update, linear cost, residual maxima) happens where the row lives,
// exactly as in the other layouts; each lane keeps its <= SMAX (knot,row) elements of g|y, v|z, d and the matching
// table entries in REGISTERS for the whole solve -- HBM is read once and written once.
// Given exact carries pass 2 IS the sequential sweep, so results differ from layouts A/B only through the rounding
// of the carries (~1e-14 relative); iteration counts match the reference in every test.
#include "tinympc_device.h"
#include "tinympc_sweep.h"

#ifndef TINY_EXP
#define TINY_EXP 0
#endif

namespace tinympc {

namespace {
constexpr int CW = 16;          // lanes per group
constexpr int CGROUPS = 16;     // groups per workgroup (256 threads)
constexpr int CTHREADS = CW * CGROUPS;

// Rows of the LDS-resident matrices are KT + 2 doubles apart: with a stride of KT the 16 lanes of a group would
// hit 2 (KT = 16) or 4 (KT = 8) bank groups with their 16-byte reads; +2 spreads them over all 64 banks.
template <int KT>
__device__ __forceinline__ void load_row(const double *M, int r, double (&m)[KT]) {
    const double *row = M + r * (KT + 2);
#pragma unroll
    for (int k = 0; k < KT; ++k) m[k] = row[k];
}

// Max over the 16 lanes of a DPP row with rotations inside the row (no LDS crossbar round trips). 64-bit DPP moves
// only exist for row_newbcast on this target, so the value travels as two 32-bit halves; the max is asm because
// fmax() adds a canonicalising v_max_f64 per operand. Nothing is volatile: callers reduce four quantities back to
// back and the scheduler interleaves them.
__device__ __forceinline__ double row_max(double v) {
#define TINY_STAGE(n)                                                                                        \
    {                                                                                                        \
        const int tlo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + (n), 0xf, 0xf, false);     \
        const int thi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + (n), 0xf, 0xf, false);     \
        const double t = __hiloint2double(thi, tlo);                                                         \
        asm("v_max_f64 %[o], %[a], %[b]" : [o] "=v"(v) : [a] "v"(v), [b] "v"(t));                            \
    }
    TINY_STAGE(8) TINY_STAGE(4) TINY_STAGE(2) TINY_STAGE(1)
#undef TINY_STAGE
    return v;
}
}  // namespace

// Columns of a carry matrix: the state dimension rounded up to what the fused DPP chain supports.
__host__ __device__ inline int chunk_ks_impl(int nx) { return nx <= 8 ? 8 : nx <= 12 ? 12 : 16; }
int chunk_ks(int nx) { return chunk_ks_impl(nx); }

// ---- carry matrices PhiS_l = Phi^(S (l+1)) | PsiS_l = Psi^(S (l+1))  (l < Lc = 4), each [16][KS] row-major (KS = nx rounded
//      up to 8 / 12 / 16: their operand is a state vector), zero outside the nx x nx state block; Phi, Psi are the
//      state blocks of the fused operators of k_build_operators
__global__ void __launch_bounds__(256) k_build_chunk_tables(const ChunkTableParams p) {
    __shared__ double Base[256], Cur[256], Tmp[256], Pw[256];
    const int nx = p.nx, KT = p.KT, KS = chunk_ks_impl(nx), S = p.S, Lc = p.Lc, tid = threadIdx.x;
    const int r = tid / 16, k = tid % 16;  // 16 x 16 working matrices
    const size_t M = (size_t)CW * KT, MS = (size_t)CW * KS;
    const bool in = (r < nx) && (k < nx);
    for (size_t i = tid; i < (size_t)2 * Lc * MS; i += 256) p.out[i] = 0.0;
    __syncthreads();
    auto mul = [&](double *Cm, const double *Am, const double *Bm) {  // Cm = Am * Bm (16 x 16, LDS)
        double acc = 0.0;
        for (int j = 0; j < nx; ++j) acc += Am[r * 16 + j] * Bm[j * 16 + k];
        __syncthreads();
        Cm[tid] = in ? acc : 0.0;
        __syncthreads();
    };
    for (int which = 0; which < 2; ++which) {
        const double *Op = p.ops + (size_t)which * M;  // Mf, then Mb
        double *dst = p.out + (size_t)which * Lc * MS;
        Base[tid] = in ? Op[r * KT + k] : 0.0;
        Cur[tid] = (r == k && r < nx) ? 1.0 : 0.0;
        __syncthreads();
        for (int i = 0; i < S; ++i) {  // Cur = Base^S
            mul(Tmp, Base, Cur);
            Cur[tid] = Tmp[tid];
            __syncthreads();
        }
        Pw[tid] = Cur[tid];  // Base^S
        __syncthreads();
        for (int l = 0; l < Lc; ++l) {  // Base^(S (l+1))
            if (in) dst[(size_t)l * MS + r * KS + k] = Cur[tid];
            mul(Tmp, Pw, Cur);
            Cur[tid] = Tmp[tid];
            __syncthreads();
        }
    }
}

hipError_t launch_build_chunk_tables(const ChunkTableParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_chunk_tables, dim3(1), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---- layout F (round 4): what a chunk's end state owes to its own inputs. One forward step is x+ = Phi x + Bd d + cf (the state
//      rows of the fused operator Mf: Phi = columns 0 .. nx-1, Bd = -B = columns nx .. nx+nu-1; cf = fdyn), so S steps from a zero
//      incoming state end in   sum_s Phi^(S-1-s) (Bd d_s + cf)   -- layout F accumulates the first part while its backward sweep
//      produces the d_s instead of sweeping every chunk a second time. Out: T_s = Phi^(S-1-s) Bd as [s][k][16 rows] (s < S, k < nu),
//      then aff[16] = sum_s Phi^s cf.
__global__ void __launch_bounds__(256) k_build_f_input_tables(const ChunkTableParams p) {
    __shared__ double Phi[256], T[256], Tn[256], aff[16], affn[16];
    const int nx = p.nx, nu = p.nu, KT = p.KT, S = p.S, tid = threadIdx.x;
    const int r = tid / 16, k = tid % 16;
    const double *Mf = p.ops;
    const double *cf = p.ops + (size_t)2 * CW * KT;
    Phi[tid] = (r < nx && k < nx) ? Mf[r * KT + k] : 0.0;
    T[tid] = (r < nx && k < nu) ? Mf[r * KT + nx + k] : 0.0;  // T_(S-1) = Bd
    if (tid < 16) aff[tid] = (tid < nx) ? cf[tid] : 0.0;
    __syncthreads();
    for (int s = S - 1; s >= 0; --s) {
        if (k < nu) p.out[((size_t)s * nu + k) * 16 + r] = T[tid];
        double acc = 0.0;
        for (int j = 0; j < nx; ++j) acc += Phi[r * 16 + j] * T[j * 16 + k];
        Tn[tid] = (r < nx && k < nu) ? acc : 0.0;
        __syncthreads();
        T[tid] = Tn[tid];
        __syncthreads();
    }
    for (int s = 1; s < S; ++s) {  // aff <- Phi aff + cf
        if (tid < 16) {
            double acc = (tid < nx) ? cf[tid] : 0.0;
            for (int j = 0; j < nx; ++j) acc += Phi[tid * 16 + j] * aff[j];
            affn[tid] = (tid < nx) ? acc : 0.0;
        }
        __syncthreads();
        if (tid < 16) aff[tid] = affn[tid];
        __syncthreads();
    }
    if (tid < 16) p.out[(size_t)S * nu * 16 + tid] = aff[tid];
}
size_t f_input_table_doubles(int nu, int S) { return (size_t)S * nu * 16 + 16; }
hipError_t launch_build_f_input_tables(const ChunkTableParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_f_input_tables, dim3(1), dim3(256), 0, stream, p);
    return hipGetLastError();
}

void chunk_plan(int N, int *S, int *C, int *Lc) {
    const int T = N - 1;
    *S = (T + CGROUPS - 1) / CGROUPS;
    *C = (T + *S - 1) / *S;
    *Lc = 4;  // carry matrices: powers S, 2S, 3S, 4S
}

size_t chunk_table_doubles(int nx, int Lc) { return (size_t)2 * Lc * CW * chunk_ks(nx); }

size_t solve_c_lds_bytes(int nx, int Lc) {
    // carry matrices + wave totals ping-pong Y[2][256] (64 used each) + boundary q Q[256] + residual partials R[16][4]
    return sizeof(double) * ((size_t)2 * Lc * CW * (chunk_ks(nx) + 2) + 2 * 256 + 256 + 64 + 64 + 3 * MAX_LIN_ROWS * CW);
}
// ... + the staged references (nx x N | nu x (N-1)) of a single-instance handle, see the kernel's prologue
size_t solve_c_lds_bytes_refs(int nx, int nu, int N, int Lc) {
    return solve_c_lds_bytes(nx, Lc) + sizeof(double) * ((size_t)nx * N + (size_t)nu * (N - 1));
}

// FAM: the second-order-cone and linear-inequality slack families of k_admm_solve_fam (PARITY UNPINNED, see there)
// ride on the row-local phase: every slot carries the extra duals gc|yc, gl|yl (persistent, HBM arrays GC / GL) and
// the extra linear-cost term lx (forward -> backward, registers).
// SESSION: the resident closed-loop variant (p.mail != NULL), a separate instantiation so that the one-shot kernel keeps
// its register budget (wrapped in the tick loop at run time it grew from 244 to 336 VGPRs and lost its second workgroup
// per CU).
template <int KT, int KS, int SMAX, bool FAM, bool SESSION>
__global__ void __launch_bounds__(CTHREADS) k_admm_solve_c(const SolveParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, c = tid >> 4, r = tid & 15;
    const int nx = p.nx, nu = p.nu, N = p.N, T = N - 1, nxu = nx + nu;
    const int S = p.chunk_len, C = p.chunk_count, Lc = p.chunk_levels;
    const long inst = blockIdx.x;
    const bool is_x = r < nx;
    const bool is_u = (r >= nx) && (r < nxu);
    const bool row_ok = r < nxu;
    const size_t M = (size_t)CW * KT;         // one fused operator / mask matrix in the global tables
    const size_t MS = (size_t)CW * KS;        // one carry matrix in the global tables
    const size_t ML = (size_t)CW * (KS + 2);  // and in LDS (padded rows)
    double *sTab = smem;
    double *sY = sTab + (size_t)2 * Lc * ML;  // [2][256]
    double *sQ = sY + 512;                    // [256]
    double *sR = sQ + 256;                    // [16][4]
    double *sMail = sR + 64;                  // [64] the session mailbox as last polled (+ the poller's time-out verdict)
    double *sLin = sMail + 64;                // [MAX_LIN_ROWS][3][16]  a_k | b_k | 1/||a_k||^2 per lane row (FAM)
    double *sRef = sLin + 3 * MAX_LIN_ROWS * CW;  // [nx*N | nu*(N-1)] references staged from pinned host memory (href only)
    const double *PH = sTab, *PS = PH + (size_t)Lc * ML;
    for (int i = tid; i < 2 * Lc * (int)MS; i += CTHREADS) {
        const int mat = i / (int)MS, rem = i % (int)MS;
        sTab[(size_t)mat * ML + (rem / KS) * (KS + 2) + rem % KS] = p.ctab[i];
    }

    // canonical HBM layout shared with the other kernels (instance = lane group inst%4 of wave group inst/4)
    const long wg = inst >> 2;
    const int jj = (int)(inst & 3);
    const int dstride = 4 * nu;
    double *gG = p.G + (size_t)wg * (N + 1) * 64 + jj * 16 + r;
    double *gV = p.V + ((size_t)wg * v_rows(N) + V_PAD) * 64 + jj * 16 + r;
    double *gD = p.D + (size_t)wg * (N - 1) * dstride + jj * nu + (is_u ? r - nx : 0);
    const int TOFF = (int)table_rows(N) * CW;
    const size_t vrow0 = ((size_t)wg * v_rows(N) + V_PAD) * 64 + jj * 16 + r;  // knot 0 in the V-shaped arrays
    double *gGC = p.GC + (FAM ? vrow0 : 0), *gGL = p.GL + (FAM ? vrow0 : 0);

    // Single-instance handles hand the references over in pinned host memory (SolveParams::href_x). A read of host
    // memory is a PCIe round trip and the link keeps only a few dozen of them in flight, so the references are fetched
    // ONCE, fully coalesced (about a hundred 64-byte requests for the quadrotor), into LDS -- and mirrored into their
    // device copies on the way; every lane then derives its own linear-cost entries from LDS behind the prologue's
    // barrier (a first version let every lane fetch its own entries from the host: ~6 us of a 25 us tick).
    constexpr bool session = SESSION;
    const bool href = p.href_x != nullptr;
    const double dg_r = (href || session) ? p.ops[2 * M + 2 * CW + r] : 0.0;
    const int Xn = nx * N, Un = nu * T;
    auto stage_refs = [&]() {  // pinned host -> LDS (+ device copies); callers put a barrier behind it
        for (int i = tid; i < Xn; i += CTHREADS) {
            const double x = p.href_x[i];
            sRef[i] = x;
            p.dXref[i] = x;
        }
        for (int i = tid; i < Un; i += CTHREADS) {
            const double u = p.href_u[i];
            sRef[Xn + i] = u;
            p.dUref[i] = u;
        }
    };
    if (href) stage_refs();
    // ---- this lane's elements: slot i <-> step k = c*S + i; state lanes own knot k+1, input lanes knot k
    double g[SMAX], v[SMAX], lo[SMAX], hi[SMAX], lr[SMAX], dd[SMAX];
    double gc[SMAX], gl[SMAX], lx[SMAX];  // FAM only
    bool ok[SMAX];     // slot holds a real element of this lane
    bool step[SMAX];   // slot is a real step of this group (uniform over the group)
    const int koff = is_x ? 1 : 0;
#pragma unroll
    for (int i = 0; i < SMAX; ++i) {
        const int k = c * S + i;
        step[i] = (i < S) && (k < T);
        ok[i] = step[i] && row_ok;
        const int kn = k + koff;
        g[i] = ok[i] ? gG[(size_t)kn * 64] : 0.0;
        v[i] = ok[i] ? gV[(size_t)kn * 64] : 0.0;
        dd[i] = (step[i] && is_u) ? gD[(size_t)k * dstride] : 0.0;
        lo[i] = ok[i] ? p.tables[(size_t)(kn + 1) * CW + r] : 0.0;
        hi[i] = ok[i] ? p.tables[(size_t)TOFF + (size_t)(kn + 1) * CW + r] : 0.0;
        lr[i] = (ok[i] && !href) ? p.tables[(size_t)2 * TOFF + (size_t)(kn + 1) * CW + r] : 0.0;  // (href: after the barrier)
        gc[i] = (FAM && ok[i]) ? gGC[(size_t)kn * 64] : 0.0;
        gl[i] = (FAM && ok[i]) ? gGL[(size_t)kn * 64] : 0.0;
        lx[i] = 0.0;
    }
    // knot 0 of the state rows: group 0, state lanes
    const bool k0 = (c == 0) && is_x;
    double g0 = k0 ? gG[0] : 0.0, v0 = k0 ? gV[0] : 0.0;
    const double lo0 = k0 ? p.tables[CW + r] : 0.0, hi0 = k0 ? p.tables[(size_t)TOFF + CW + r] : 0.0;
    double x0v = k0 ? p.x0[inst * nx + r] : 0.0;
    if (p.x0_mirror && k0) p.x0_mirror[inst * nx + r] = x0v;  // zero-copy tick: x0 came from host memory
    double gc0 = (FAM && k0) ? gGC[0] : 0.0, gl0 = (FAM && k0) ? gGL[0] : 0.0;

    double mf[KT], mb[KT];
    {
        const double *Mf = p.ops + (size_t)r * KT, *Mb = p.ops + M + (size_t)r * KT;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            mf[k] = Mf[k];
            mb[k] = Mb[k];
        }
    }
    const double cf = p.ops[2 * M + r];
    const double cb = p.ops[2 * M + CW + r];
    // ---- families: mask rows and coefficients (layout of fam_doubles(), built on the host from the verbs' data)
    double cn[KT], ct_[KT], ty[KT];
    // (the linear rows' coefficients live in LDS and the rows are walked by a run-time loop: unrolled over MAX_LIN_ROWS
    // with the coefficients in registers, every slot carried eight copies of the mat-vec behind scalar branches -- the
    // iteration's code no longer fitted the instruction cache and 48 VGPRs were pinned for rows that mostly do not exist)
    int role = 0, nl = 0;
    double mu = 0.0, inv_mu = 0.0;
    bool famc = false, faml = false, any_cone = false, any_lin = false;
    if (FAM) {
        const double *Cn = p.fam + 4 * CW + (size_t)r * KT, *Ct = Cn + M, *Ty = Ct + M;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            cn[k] = Cn[k];
            ct_[k] = Ct[k];
            ty[k] = Ty[k];
        }
        role = (int)p.fam[r];
        mu = p.fam[CW + r];
        inv_mu = (mu != 0.0) ? 1.0 / mu : 0.0;  // (mu = 0: row in no cone)
        famc = p.fam[2 * CW + r] != 0.0;
        faml = p.fam[3 * CW + r] != 0.0;
        const double *lin_rows = p.fam + 4 * CW + 3 * M;
        nl = (int)lin_rows[0];
        if (c == 0)
            for (int k = 0; k < MAX_LIN_ROWS; ++k) {
                sLin[(3 * k + 0) * CW + r] = lin_rows[1 + (size_t)(3 * k + 0) * CW + r];
                sLin[(3 * k + 1) * CW + r] = lin_rows[1 + (size_t)(3 * k + 1) * CW + r];
                sLin[(3 * k + 2) * CW + r] = 1.0 / lin_rows[1 + (size_t)(3 * k + 2) * CW + r];  // 1 / ||a_k||^2
            }
        any_cone = __ballot(famc) != 0ull;  // the same in every wavefront: rows repeat per group
        any_lin = __ballot(faml) != 0ull;
    }
    // One (row, knot) element of the two extra families, exactly as in k_admm_solve_fam: returns the element's
    // contribution to the linear cost and the new duals. Must run with all 16 lanes of the group enabled.
    auto families = [&](double val, double gc_old, double gl_old, double &gc_new, double &gl_new) -> double {
        double lxv = 0.0;
        gc_new = gc_old;
        gl_new = gl_old;
#if TINY_EXP == 1  // timing experiments only (tools/build_variants.sh): no family work at all
        return 0.0;
#endif
        if (any_cone) {
            const double sv = val + gc_old;
#if TINY_EXP == 2  // ... the cone's projection math without its two mat-vecs
            const double a2 = sv * sv, t = sv;
#else
            const double a2 = group_matvec<CW, KT>(cn, sv * sv, 0.0);
            const double t = group_matvec<CW, KT>(ct_, sv, 0.0);
#endif
#if TINY_EXP == 3  // ... the mat-vecs without the projection math
            const double vc = a2 + t;
#else
            const double vc = soc_project_element(sv, a2, t, mu, inv_mu, role);
#endif
            const double gcn = sv - vc;
            if (famc) {
                gc_new = gcn;
                lxv -= p.rho * (vc - gcn);
            }
        }
        if (any_lin) {
            const double s0 = val + gl_old;
            double sv = s0;
#pragma unroll 1
            for (int k = 0; k < nl; ++k) {  // (uniform trip count)
                const double a_k = sLin[(3 * k + 0) * CW + r], b_k = sLin[(3 * k + 1) * CW + r], in_k = sLin[(3 * k + 2) * CW + r];
                const double dot = group_matvec<CW, KT>(ty, a_k * sv, 0.0);
                sv = halfspace_project_element(sv, dot, a_k, b_k, in_k);
            }
            const double gln = s0 - sv;
            if (faml) {
                gl_new = gln;
                lxv -= p.rho * (sv - gln);
            }
        }
        return lxv;
    };
    double pnref = p.tables[(size_t)3 * TOFF + r];
    const double rho = p.rho;
    const int ct = p.check_termination;
    const int i_last = (T - 1) - (C - 1) * S;  // slot of the last step, in group C-1
    __syncthreads();
    auto apply_refs = [&]() {  // behind the barrier that follows stage_refs()
        // the expressions of k_build_tables, on the staged copy: -(Xref .* Q) / -(Uref .* R) (admm.cpp:77-79) and
        // pNref = -(Xref_{N-1}' Pinf)' (admm.cpp:81), the sum term by term in the same order
        double *tab = const_cast<double *>(p.tables);
#pragma unroll
        for (int i = 0; i < SMAX; ++i) {
            const int kn = c * S + i + koff;
            if (ok[i]) {
                const double ref = is_x ? sRef[r + kn * nx] : sRef[Xn + (r - nx) + kn * nu];
                lr[i] = -(ref * dg_r);
                tab[(size_t)2 * TOFF + (size_t)(kn + 1) * CW + r] = lr[i];  // mirror: the table row the other kernels read
            }
        }
        double acc = 0.0;
        if constexpr (SESSION) {  // (the resident variants have no registers to spare; a full refresh is their rare path)
            if (is_x) {
#pragma unroll
                for (int q = 0; q < CW; ++q)
                    if (q < nx) acc += sRef[q + (N - 1) * nx] * p.Pinf[q + (size_t)r * nx];
                acc = -acc;
            }
        } else {
            // the row's Pinf entries are requested all at once, clamped instead of guarded: with the loads inside `if (q < nx)`
            // they went out one by one, each behind an s_waitcnt for the previous -- nx serial L2 round trips per tick
            double pin[CW];
            const int rr = is_x ? r : 0;
#pragma unroll
            for (int q = 0; q < CW; ++q) pin[q] = p.Pinf[(q < nx ? q : 0) + (size_t)rr * nx];
#pragma unroll
            for (int q = 0; q < CW; ++q)
                if (q < nx) acc += sRef[q + (N - 1) * nx] * pin[q];
            acc = is_x ? -acc : 0.0;
        }
        pnref = acc;
        if (k0) tab[(size_t)2 * TOFF + (size_t)CW + r] = -(sRef[r] * dg_r);
        if (c == 0) tab[(size_t)3 * TOFF + r] = pnref;
    };
    if (href) apply_refs();

    int it_done = 0, status = 11;
    bool res_valid = false, converged = false;
    double snap_pri = 0.0, snap_dua = 0.0;  // this lane's residual maxima at the last termination check
    double vprev[SMAX], v0prev = v0;
#pragma unroll
    for (int i = 0; i < SMAX; ++i) vprev[i] = v[i];
    int cur = 0;  // carry ping-pong buffer

    // Carry scan over the chunks. In: the chunk's pass-1 end value (state lanes; 0 elsewhere and in idle groups). Out:
    // the true value ENTERING the chunk from its neighbour c + dir,  I_c = sum_{j>=1} Pm^(j-1) end_(c + j*dir).
    // Chunk c = 4 w + j is row j of wavefront w.
    //   A  inside the wavefront, rows only (cross-row swaps):  t_j = e_j + P1 e_(j-1);  L_j = t_j + P2 t_(j-2)
    //      -> L_j = the prefix over the wavefront's own rows up to j (P_n = Pm^(n S), "j-1" meaning the row towards -dir)
    //   B  the wavefront's total L_last goes to LDS; ONE barrier; Horner over the three wavefronts before it:
    //      Cin = T_(w-3);  Cin = T_(w-2) + P4 Cin;  Cin = T_(w-1) + P4 Cin   (absent wavefronts count as zero)
    //   C  I_j = L_(j-1) + P_j Cin for the rows behind the first; the first row's incoming value is Cin itself.
    // The matrix rows of all three stages are requested up front, so the LDS reads overlap stage A.
    const int wv = tid >> 6, jrow = (tid >> 4) & 3;
    auto rows_shift1 = [&](int dir, double x) -> double {  // row j <- row j + dir (the wavefront's edge row <- 0)
        double e, o, lo2, hi2;
        cross_row_pair<1>(x, e, o);  // e = [r0 r0 r2 r2], o = [r1 r1 r3 r3]
        if (dir < 0) {               // [0 r0 r1 r2]
            cross_row_pair<0>(o, lo2, hi2);  // lo2 = [r1 r1 r1 r1]
            return jrow == 0 ? 0.0 : (jrow == 2 ? lo2 : e);
        } else {                     // [r1 r2 r3 0]
            cross_row_pair<0>(e, lo2, hi2);  // hi2 = [r2 r2 r2 r2]
            return jrow == 3 ? 0.0 : (jrow == 1 ? hi2 : o);
        }
    };
    auto rows_shift2 = [&](int dir, double x) -> double {  // row j <- row j + 2 dir
        double lo2, hi2;
        cross_row_pair<0>(x, lo2, hi2);  // lo2 = [r0 r1 r0 r1], hi2 = [r2 r3 r2 r3]
        return dir < 0 ? (jrow >= 2 ? lo2 : 0.0) : (jrow < 2 ? hi2 : 0.0);
    };
    auto carry_scan = [&](int dir, const double *Pm, double end_val) -> double {
#if TINY_EXP == 4  // timing experiment: no carry scan at all
        return end_val;
#endif
        const int jr = dir < 0 ? jrow : 3 - jrow;  // rows counted from the side the carry comes from
        // (two matrix rows live at a time: the rows of stages B and C are requested after stage A's mat-vecs, still
        // ahead of the barrier that hides their latency -- all four up front cost 48 VGPRs and a workgroup slot per CU)
        double L;
        {
            double m1[KS], m2[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) m1[k] = m2[k] = 0.0;
            if (is_x) {
                load_row<KS>(Pm, r, m1);
                load_row<KS>(Pm + ML, r, m2);
            }
            // A
            const double t = end_val + group_matvec<CW, KS>(m1, rows_shift1(dir, end_val), 0.0);
            L = t + group_matvec<CW, KS>(m2, rows_shift2(dir, t), 0.0);
        }
        double m4[KS], mj[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) m4[k] = mj[k] = 0.0;
        if (is_x) {
            load_row<KS>(Pm + 3 * ML, r, m4);
            load_row<KS>(Pm + (size_t)(jr >= 1 ? jr - 1 : 0) * ML, r, mj);
        }
        // B
        if (jr == 3 && is_x) sY[cur * 256 + wv * 16 + r] = L;
        const double Lprev = rows_shift1(dir, L);
        __syncthreads();
        const int v1 = wv + dir, v2 = wv + 2 * dir, v3 = wv + 3 * dir;
        const double T1 = (v1 >= 0 && v1 < 4 && is_x) ? sY[cur * 256 + v1 * 16 + r] : 0.0;
        const double T2 = (v2 >= 0 && v2 < 4 && is_x) ? sY[cur * 256 + v2 * 16 + r] : 0.0;
        const double T3 = (v3 >= 0 && v3 < 4 && is_x) ? sY[cur * 256 + v3 * 16 + r] : 0.0;
        double cin = T2 + group_matvec<CW, KS>(m4, T3, 0.0);
        cin = T1 + group_matvec<CW, KS>(m4, cin, 0.0);
        // C
        const double far = Lprev + group_matvec<CW, KS>(mj, cin, 0.0);
        cur ^= 1;
        return is_x ? (jr == 0 ? cin : far) : 0.0;
    };

    // The ADMM state in its canonical HBM layout. `conv`: the solve converged, i.e. the reference returned before v <- vnew
    // (admm.cpp:181-197) and the canonical slack is the previous iterate.
    auto write_state = [&](bool conv) {
#pragma unroll
        for (int i = 0; i < SMAX; ++i) {
            if (ok[i]) {
                const int kn = c * S + i + koff;
                gG[(size_t)kn * 64] = g[i];
                gV[(size_t)kn * 64] = conv ? vprev[i] : v[i];
                if (FAM) {
                    gGC[(size_t)kn * 64] = gc[i];
                    gGL[(size_t)kn * 64] = gl[i];
                }
            }
            if (step[i] && is_u) gD[(size_t)(c * S + i) * dstride] = dd[i];
        }
        if (k0) {
            gG[0] = g0;
            gV[0] = conv ? v0prev : v0;
            if (FAM) {
                gGC[0] = gc0;
                gGL[0] = gl0;
            }
        }
    };

    // ---- SESSION: the kernel stays resident; every pass of this loop is one closed-loop tick (one pass otherwise).
    double expect = p.session_expect;
    for (;;) {
    if constexpr (SESSION) {
        // Poll the mailbox: lanes 0..23 fetch its three lines in one load, everybody sees them through LDS and takes
        // the same decision. A command is complete when the stamp of every line it uses matches `expect`. The poller's
        // clock ends the session after p.session_idle ticks without a command -- the exit every wavefront reaches
        // even if the host process is gone.
        const unsigned long long t_idle0 = __builtin_amdgcn_s_memrealtime();
        // payload: 0 flags | 1.. x0 (nx) | new last column of x_ref (nx, flag 4) | new last column of u_ref (nu, flag 8)
        bool go = false, quit = false;
        while (!go && !quit) {
            // Wavefront 0 fetches the mailbox -- lanes 0..55: the seven lines in ONE load instruction -- and decides in registers: a line
            // is accepted when its stamp is `expect` plus the checksum of the seven payload words READ WITH IT (mail_lines_ok,
            // tinympc_device.h): a poll that caught a line half-written -- the 64 bytes come in one load, but nothing in PCIe
            // promises they are one snapshot -- fails the test and is simply repeated. Line 0 carries the flags, which say how many
            // lines the command uses. Words, the time-out's verdict and the decision reach the other wavefronts through sMail.
            if (tid < 64) {
                const double w = tid < 56 ? __hip_atomic_load(p.mail + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
                const unsigned ok = mail_lines_ok(w, tid, expect);
                const unsigned long long w0 = (unsigned long long)__double_as_longlong(w);
                const double flags_word = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w0 >> 32)) << 32) |
                                                                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)w0)));
                const int f0 = (ok & 1u) ? (int)flags_word : 0;
                const int npay = 1 + nx + ((f0 & 4) ? nx : 0) + ((f0 & 8) ? nu : 0), nlines = (npay + 6) / 7;
                const unsigned need = (1u << nlines) - 1u;
                if (tid < 56) sMail[tid] = w;
                if (tid == 0) {
                    sMail[56] = (__builtin_amdgcn_s_memrealtime() - t_idle0 > p.session_idle) ? 1.0 : 0.0;
                    sMail[57] = ((ok & need) == need) ? 1.0 : 0.0;
                }
            }
            __syncthreads();
            go = sMail[57] != 0.0;
            quit = !go && sMail[56] != 0.0;
            __syncthreads();  // (the next poll overwrites sMail)
        }

        const int flags = go ? (int)sMail[0] : 1;
        if (quit || (flags & 1)) {  // stop requested, or nobody is talking to this kernel any more
            write_state(false);     // (a converged tick already rolled its slack back, see the end of the loop)
            break;
        }
        if (k0) {
            const int q = 1 + r;  // payload index of x0[r]
            x0v = sMail[8 * (q / 7) + q % 7];
            if (p.x0_mirror) p.x0_mirror[inst * nx + r] = x0v;
        }
        if (flags & 2) {  // the references changed: fetch them again (tinympc_set_x_ref / _u_ref filled the pinned copies)
            // acquire: the stamp was observed, what the host wrote before it must be too -- without the fence these plain
            // loads can be served from L2 lines that an earlier read of the same buffer (the prologue's) left there
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            stage_refs();
            __syncthreads();
            apply_refs();
        } else if (flags & 12) {
            // Receding horizon (rocket_landing_constraints.m:96-101): the new reference is the previous one moved up by one
            // knot plus ONE new last column, which came with the command -- no 6 KB PCIe read (~5 us), just a shift of the
            // linear-cost entries through the chunk boundaries. The host checked, bit for bit, that it IS a shift.
            const bool mine = is_x ? (flags & 4) != 0 : (flags & 8) != 0;
            const int qn = is_x ? 1 + nx + r : 1 + nx + ((flags & 4) ? nx : 0) + (r - nx);  // payload index of this row's new entry
            const double fresh = row_ok ? sMail[8 * (qn / 7) + qn % 7] : 0.0;
            double first = 0.0;
#pragma unroll
            for (int i = 0; i < SMAX; ++i) first = (i == 0) ? lr[i] : first;
            sQ[tid] = first;
            __syncthreads();
            const double next_first = (c + 1 < CGROUPS) ? sQ[(c + 1) * 16 + r] : 0.0;
            if (mine) {
#pragma unroll
                for (int i = 0; i < SMAX; ++i) {
                    const int k = c * S + i;
                    if (ok[i]) lr[i] = (k + 1 >= T) ? -(fresh * dg_r) : ((i + 1 < S && i + 1 < SMAX) ? lr[i + 1 < SMAX ? i + 1 : i] : next_first);
                }
            }
            if (flags & 4) {  // pNref = -(Xref_{N-1}' Pinf)' from the new last column, the sum of k_build_tables
                double acc = 0.0;
                if (is_x) {
#pragma unroll
                    for (int q = 0; q < CW; ++q)
                        if (q < nx) acc += sMail[8 * ((1 + nx + q) / 7) + (1 + nx + q) % 7] * p.Pinf[q + (size_t)r * nx];
                    acc = -acc;
                }
                pnref = acc;
            }
        }

        __syncthreads();  // sMail has been read by everyone
        it_done = 0;
        status = 11;
        res_valid = false;
        converged = false;
#pragma unroll
        for (int i = 0; i < SMAX; ++i) vprev[i] = v[i];
        v0prev = v0;
    }
    for (int it = 0; it < p.max_iter; ++it) {
        const bool check = (ct > 0) && (((it + 1) % ct) == 0);
#if TINY_EXP == 9  // timing experiment: cycle stamps of one iteration's phases (tools/c_breakdown.py prints them)
        unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool stamp = it == 7 && blockIdx.x == 0;
#define TSTAMP(n, dep) if (stamp) { asm volatile("s_nop 0" :: "v"(dep)); ts[n] = __builtin_readcyclecounter(); }
#else
#define TSTAMP(n, dep)
#endif
        TSTAMP(0, x0v)
        // ================= forward sweep =================
        double out[SMAX];
        {   // pass 1: end state of the chunk from a zero incoming state (chunk 0: from x_0)
            double xt = x0v;
#if TINY_EXP != 5  // (5: timing experiment without pass 1)
#pragma unroll
            for (int i = 0; i < SMAX; ++i)
                if (i < S) {
                    const double o = group_matvec<CW, KT>(mf, is_x ? xt : dd[i], cf);
                    xt = step[i] ? o : xt;
                }
#endif
            TSTAMP(1, xt)
            const double xin = carry_scan(-1, PH, (is_x && c < C) ? xt : 0.0);
            TSTAMP(2, xin)
            // pass 2: the real sweep, from the true state entering the chunk
            xt = (c >= 1) ? xin : x0v;
#pragma unroll
            for (int i = 0; i < SMAX; ++i) {
                out[i] = 0.0;
                if (i < S) {
                    out[i] = group_matvec<CW, KT>(mf, is_x ? xt : dd[i], cf);  // state lanes x_{k+1}, input lanes u_k
                    xt = step[i] ? out[i] : xt;
                }
            }
        }

        TSTAMP(3, out[0])
        // ================= row-local phases (S1, D1, R1) =================
        double pri = 0.0, dua = 0.0;
        if (k0) {
            const double s = x0v + g0;
            const double snew = fmin(hi0, fmax(lo0, s));
            pri = fabs(x0v - snew);
            dua = fabs(v0 - snew);
            v0prev = v0;
            g0 = s - snew;
            v0 = snew;
        }
        if (FAM) {  // knot 0 of the state rows (its lx only reaches p_0, which nothing reads; the duals persist)
            double gcn, gln;
            (void)families(x0v, gc0, gl0, gcn, gln);
            if (k0) {
                gc0 = gcn;
                gl0 = gln;
            }
        }
#pragma unroll
        for (int i = 0; i < SMAX; ++i) {
            double gnew, snew, tp = 0.0, td = 0.0;
            project_element(out[i], g[i], lo[i], hi[i], v[i], gnew, snew, tp, td);
            vprev[i] = v[i];
            if (ok[i]) {
                g[i] = gnew;
                v[i] = snew;
                pri = fmax(pri, tp);
                dua = fmax(dua, td);
            }
            if (FAM && i < S) {
                double gcn, gln;
                const double l = families(out[i], gc[i], gl[i], gcn, gln);
                if (ok[i]) {
                    gc[i] = gcn;
                    gl[i] = gln;
                    lx[i] = l;
                }
            }
        }
        it_done = it + 1;
        // linear cost of the backward sweep; its chunk-boundary exchange shares the barrier of the residual exchange
        double lin[SMAX];
#pragma unroll
        for (int i = 0; i < SMAX; ++i) lin[i] = lr[i] - rho * (v[i] - g[i]) + lx[i];  // q_{k+1} (state lanes) | r_k (input lanes)
        {   // q_k of a chunk's first step lives in the previous group (its last state slot)
            double qlast = 0.0;
#pragma unroll
            for (int i = 0; i < SMAX; ++i) qlast = (i == S - 1) ? lin[i] : qlast;
            sQ[tid] = qlast;
        }
        // R1 (admm.cpp:93-101). "All four inf-norms below their tolerances" is decided element-wise: max_i a_i < tol
        // iff every a_i < tol, and scaling by rho > 0 is monotone, so one ballot per wavefront and four flags through
        // LDS replace the eight row reductions of the norms themselves; the norms (for get_stats) are reduced once,
        // after the loop, from the snapshot taken at the last check.
        if (check) {
            snap_pri = pri;
            snap_dua = dua;
            res_valid = true;
            const bool below = (pri < p.abs_pri_tol) && (dua * rho < p.abs_dua_tol);
            const bool wave_ok = __ballot(!below) == 0ull;
            // (flag slots alternate with the iteration's parity: with no carry-scan level -- N == 2 -- this barrier is
            // the only one of an iteration, and a fast wavefront would otherwise overwrite its flag for iteration it+1
            // before a slow one has read the flags of iteration it)
            if ((tid & 63) == 0) reinterpret_cast<int *>(sR)[(it & 1) * 4 + (tid >> 6)] = wave_ok ? 1 : 0;
        }
        __syncthreads();
        if (check) {
            const int *flags = reinterpret_cast<const int *>(sR) + (it & 1) * 4;
            if ((flags[0] & flags[1] & flags[2] & flags[3]) != 0) {
                status = 1;  // uniform over the workgroup: one instance
                converged = true;
                break;
            }
        }

        // ================= backward sweep =================
        TSTAMP(4, pri)
        {
            double pterm = 0.0;
#pragma unroll
            for (int i = 0; i < SMAX; ++i) pterm = (i == i_last) ? (pnref - rho * (v[i] - g[i]) + lx[i]) : pterm;  // p_{N-1}, admm.cpp:81-82
            const double qin = (c >= 1 && is_x) ? sQ[(c - 1) * 16 + r] : 0.0;
            const double pend = (c == C - 1 && is_x) ? pterm : 0.0;
            // pass 1: p at the chunk's first knot from a zero incoming p (last chunk: from p_{N-1})
            double pcur = pend;
#if TINY_EXP != 5
#pragma unroll
            for (int i = SMAX - 1; i >= 0; --i)
                if (i < S) {
                    const double qk = (i >= 1) ? lin[i >= 1 ? i - 1 : 0] : qin;
                    const double o = group_matvec<CW, KT>(mb, is_x ? pcur : lin[i], cb);
                    pcur = step[i] ? (qk + o) : pcur;
                }
#endif
            TSTAMP(5, pcur)
            const double pin = carry_scan(+1, PS, (is_x && c < C) ? pcur : 0.0);
            TSTAMP(6, pin)
            // pass 2: the real sweep, from the true p entering the chunk; only d_k is kept
            pcur = (c < C - 1) ? pin : pend;
#pragma unroll
            for (int i = SMAX - 1; i >= 0; --i)
                if (i < S) {
                    const double qk = (i >= 1) ? lin[i >= 1 ? i - 1 : 0] : qin;
                    const double o = group_matvec<CW, KT>(mb, is_x ? pcur : lin[i], cb);
                    dd[i] = (step[i] && is_u) ? o : dd[i];
                    pcur = step[i] ? (qk + o) : pcur;
                }
            TSTAMP(7, pcur)
        }
#if TINY_EXP == 9
        if (stamp && (tid & 63) == 0)
            printf("wave %d: fwd pass 1 %llu | scan %llu | pass 2 %llu | row-local + linear cost + flags barrier %llu | bwd pass 1 %llu | scan %llu | pass 2 %llu | total %llu cycles\n",
                   tid >> 6, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6], ts[7] - ts[0]);
#endif
    }


    if constexpr (SESSION) {
        // EARLY ANSWER (SolveParams::host_ans, as in layout F's resident kernel): the tick's first controls go to the host at once, in lines
        // [7 controls | mail_stamp(sequence number, controls)] written by one store instruction each -- accepted by the host when the stamp
        // fits the payload, so nothing is fenced or waited for; residual norms and the solution's write-out follow below.
        if (p.host_ans && p.max_iter > 0) {  // (uniform)
#pragma unroll
            for (int i = 0; i < SMAX; ++i)
                if (ok[i] && !is_x && (c * S + i + koff) == 0) sMail[r - nx] = v[i];  // (the command in sMail has been consumed)
            __syncthreads();
            const int nla = (nu + 6) / 7;
            if (tid < 8 * nla) {  // (whole groups of eight lanes: mail_xor8)
                const int line = tid >> 3, slot = tid & 7, idx = line * 7 + slot;
                const double mine = (slot < 7 && idx < nu) ? sMail[idx] : 0.0;
                const unsigned h = mail_xor8(slot < 7 ? mail_term((unsigned long long)__double_as_longlong(mine), slot) : 0u);
                host_store(p.host_ans + tid, slot == 7 ? mail_stamp(expect, h) : mine);
            }
        }
    }

    // ---- the four residual norms of the last check, for get_stats (two-stage max: rows, then groups through LDS)
    double res_px = 0.0, res_dx = 0.0, res_pu = 0.0, res_du = 0.0;
    if (res_valid) {
        __syncthreads();  // the flags in sR have been read by everyone
        const double gpx = row_max(is_x ? snap_pri : 0.0), gpu_ = row_max(is_u ? snap_pri : 0.0);
        const double gdx = row_max(is_x ? snap_dua : 0.0), gdu = row_max(is_u ? snap_dua : 0.0);
        if (r < 4) sR[c * 4 + r] = (r == 0) ? gpx : (r == 1) ? gpu_ : (r == 2) ? gdx : gdu;
        __syncthreads();
        res_px = row_max(sR[r * 4 + 0]);
        res_pu = row_max(sR[r * 4 + 1]);
        res_dx = row_max(sR[r * 4 + 2]) * rho;
        res_du = row_max(sR[r * 4 + 3]) * rho;
    }

    // ---- write-back: solution (device + pinned host) every pass; the ADMM state for the next launch only when this
    // kernel is about to end (a session keeps it in registers from tick to tick and stores it once, on its way out)
    if (p.max_iter > 0) {
#pragma unroll
        for (int i = 0; i < SMAX; ++i) {
            if (ok[i]) {
                const int k = c * S + i, kn = k + koff;
                if (is_x) {
                    p.sol_x[((size_t)inst * N + kn) * nx + r] = v[i];
                } else {
                    p.sol_u[((size_t)inst * (N - 1) + kn) * nu + (r - nx)] = v[i];
                    if (kn == 0 && p.u0_host) host_store(&p.u0_host[(size_t)inst * nu + (r - nx)], v[i]);  // first controls straight to the host
                }
                if (p.host_sol) {  // single-instance handle: the solution also goes straight into pinned host memory
                    if (is_x) host_store(&p.host_sol[(size_t)kn * nx + r], v[i]);
                    else host_store(&p.host_sol[(size_t)N * nx + (size_t)kn * nu + (r - nx)], v[i]);
                }
            }
        }
        if (k0) {
            p.sol_x[(size_t)inst * N * nx + r] = v0;
            if (p.host_sol) host_store(&p.host_sol[r], v0);
        }
        if constexpr (!SESSION) write_state(converged);
    }
    if (tid == 0) {
        p.istats[inst * 2 + 0] = it_done;
        p.istats[inst * 2 + 1] = status;
        if (p.host_sol) {
            double *hs = p.host_sol + (size_t)N * nx + (size_t)(N - 1) * nu;
            host_store(&hs[4], (double)it_done);
            host_store(&hs[5], (double)status);
            if (res_valid) { host_store(&hs[0], res_px); host_store(&hs[1], res_dx); host_store(&hs[2], res_pu); host_store(&hs[3], res_du); }
        }
        if (res_valid) {
            p.dstats[inst * 4 + 0] = res_px;
            p.dstats[inst * 4 + 1] = res_dx;
            p.dstats[inst * 4 + 2] = res_pu;
            p.dstats[inst * 4 + 3] = res_du;
        }
    }

    if (p.host_sol && (session || p.host_seq != 0.0)) {  // (uniform) everything above is in pinned memory: raise the flag
        // Everything the host reads after the stamp was stored with host_store() (system scope: written through, nothing of it stays dirty in
        // the L2): once every thread's stores have been handed over (s_waitcnt vmcnt(0) -- a workgroup-scope release) the stamp may follow
        // them. Rounds 2-4 had plain stores and a system-scope fence here (__threadfence_system + a release store), which wrote back the
        // L2's dirty lines -- the solution just stored to HBM included -- and INVALIDATED the caches, twice per tick: the next tick then
        // fetched its operators and tables from HBM again.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        // (a system-scope atomic store: written through at once -- a plain store might be combined or delayed)
        if (tid == 0)
            __hip_atomic_store(p.host_sol + (size_t)N * nx + (size_t)(N - 1) * nu + 6, session ? expect : p.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if constexpr (!SESSION) break;
    // The next tick warm-starts from the registers. A converged solve returns before v <- vnew (admm.cpp:181-197): its
    // canonical slack is the previous iterate -- what the write-back above stored, and what an ordinary launch would
    // read back.
    if (converged) {
#pragma unroll
        for (int i = 0; i < SMAX; ++i) v[i] = vprev[i];
        v0 = v0prev;
    }
    expect += 1.0;
    }  // ticks
}

template <int KT, int KS, int SMAX, bool FAM, bool SESSION>
static hipError_t launch_c_s(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    static size_t lds_set[16] = {0};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_admm_solve_c<KT, KS, SMAX, FAM, SESSION>), lds_bytes, lds_set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_admm_solve_c<KT, KS, SMAX, FAM, SESSION>), dim3(p.batch), dim3(CTHREADS), lds_bytes, stream, p);
    return hipGetLastError();
}
template <int KT, int KS, int SMAX, bool FAM>
static hipError_t launch_c_f(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    if (p.mail) {
        // (families + more than 4 slots per lane: the resident variant would not fit the register file -- the host
        // layer refuses such sessions; ticks of that size are far above the launch overhead a session saves anyway)
        if constexpr (FAM && SMAX > 4) return hipErrorInvalidValue;
        else return p.batch == 1 ? launch_c_s<KT, KS, SMAX, FAM, true>(p, lds_bytes, stream) : hipErrorInvalidValue;
    }
    return launch_c_s<KT, KS, SMAX, FAM, false>(p, lds_bytes, stream);
}

template <int KT, int KS>
static hipError_t launch_c_t(const SolveParams &p, size_t lds_bytes, hipStream_t stream) {
    if constexpr (KS > KT) {
        return hipErrorInvalidValue;
    } else {
        const bool small = p.chunk_len <= 4;
        if (p.families) {
            if (!p.fam || !p.GC || !p.GL) return hipErrorInvalidValue;
            return small ? launch_c_f<KT, KS, 4, true>(p, lds_bytes, stream) : launch_c_f<KT, KS, 8, true>(p, lds_bytes, stream);
        }
        return small ? launch_c_f<KT, KS, 4, false>(p, lds_bytes, stream) : launch_c_f<KT, KS, 8, false>(p, lds_bytes, stream);
    }
}

// One instance per workgroup; W must be 16 and the chunk length at most 8 (N <= 129).
hipError_t launch_solve_c(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream) {
    if (W != 16 || !p.ctab || p.chunk_len < 1 || p.chunk_len > 8 || p.chunk_count > CGROUPS) return hipErrorInvalidValue;
    const int KS = chunk_ks(p.nx);
    if (KT == 8 && KS == 8) return launch_c_t<8, 8>(p, lds_bytes, stream);
    if (KT == 12 && KS == 8) return launch_c_t<12, 8>(p, lds_bytes, stream);
    if (KT == 12 && KS == 12) return launch_c_t<12, 12>(p, lds_bytes, stream);
    if (KT == 16 && KS == 8) return launch_c_t<16, 8>(p, lds_bytes, stream);
    if (KT == 16 && KS == 12) return launch_c_t<16, 12>(p, lds_bytes, stream);
    if (KT == 16 && KS == 16) return launch_c_t<16, 16>(p, lds_bytes, stream);
    return hipErrorInvalidValue;
}

}  // namespace tinympc
