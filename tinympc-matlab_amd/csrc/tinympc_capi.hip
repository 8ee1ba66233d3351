// tinympc_capi.hip -- the extern "C" boundary declared in include/tinympc_hip.h.
//
// Host-side bookkeeping only: argument validation with the reference's error behaviour, device
// buffer ownership, lazy rebuild of the fused operators / per-knot tables when their inputs change,
// and kernel launches. Every number the solver produces is computed by the kernels in
// tinympc_kernels.hip; there is no CPU fallback anywhere in this file.
#include "tinympc_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "tinympc_device.h"
#include "tinympc_host.h"

namespace tinympc {

std::string &last_error_slot() {
    thread_local std::string slot;
    return slot;
}

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

}  // namespace tinympc

using namespace tinympc;

namespace {

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail(TINYMPC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

struct Settings {  // TinySettings (types.hpp:61-74) + the newer flags (bindings.cpp:583-586)
    double abs_pri_tol, abs_dua_tol;
    int max_iter, check_termination;
    int en_state_bound, en_input_bound;
    int en_state_soc, en_input_soc, en_state_linear, en_input_linear;
    int adaptive_rho;
    double adaptive_rho_min, adaptive_rho_max;
    int adaptive_rho_enable_clipping;
};

constexpr double kBoundInf = 1e17;  // TinyMPC.m:261-264
// Every pinned buffer the kernels and the host exchange data through WHILE a kernel runs (completion flags, the session
// mailbox, references re-read by a resident kernel) is allocated hipHostMallocCoherent: with the default flags the
// GPU may keep host lines in its L2 until the kernel ends, and a resident kernel then polls a stale copy forever.
constexpr int kZeroCopyTickMax = 256;   // mpc_step: up to this many instances exchange x0 / u0 through pinned host memory
constexpr int kLayoutEBatchMin = 260;   // families at horizons layout D cannot hold: from here on layout E (4 instances per CU, the whole
                                        // state on chip) passes the latency kernel (1 instance per CU); measured, profiles/r03_rocket_sweep.txt
constexpr int kLayoutCBatchMax = 768;  // above this the batch-oriented layouts win (profiles/r02_layout_sweep.txt: layout D with four
                                       // wavefronts per workgroup passes the latency kernel between 512 and 1,024 instances)

}  // namespace

struct tinympc_solver {
    int nx = 0, nu = 0, N = 0, batch = 0, device = 0;
    int W = 0, KT = 0, IPW = 0, groups = 0;
    double rho = 0.0;
    Settings st{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // problem + cache
    double *dA = nullptr, *dB = nullptr, *dfdyn = nullptr, *dQd = nullptr, *dRd = nullptr;
    double *dKinf = nullptr, *dPinf = nullptr, *dQuu = nullptr, *dAmBKt = nullptr, *dAPf = nullptr, *dBPf = nullptr;
    double *dscratch = nullptr;
    int *dinfo = nullptr;
    // the .m class's own Riccati helpers (compute_cache_terms / solve_lqr / compute_sensitivity_autograd)
    double *dQfull = nullptr, *dRfull = nullptr, *dlqr_scratch = nullptr, *dlqr_out = nullptr;  // dlqr_out: 3 x cache_doubles()
    // adaptive rho: sensitivities dKinf/drho, dPinf/drho (zeros until set), kernel tables, per-instance rho
    double *ddK = nullptr, *ddP = nullptr, *dadapt = nullptr, *drho_inst = nullptr;
    size_t cache_doubles() const { return (size_t)nu * nx + (size_t)2 * nx * nx + (size_t)nu * nu; }  // K | P | C1 | C2
    // user-layout bounds / refs
    double *dxmin = nullptr, *dxmax = nullptr, *dumin = nullptr, *dumax = nullptr, *dXref = nullptr, *dUref = nullptr;
    // derived
    double *dops = nullptr, *dtables = nullptr;
    bool ops_dirty = true, tables_dirty = true;
    // per-instance state
    double *dx0 = nullptr, *dG = nullptr, *dV = nullptr, *dD = nullptr, *dsolx = nullptr, *dsolu = nullptr;
    int *distats = nullptr;
    double *ddstats = nullptr;
    size_t lds_bytes = 0;
    bool tables_in_lds = false;
    bool layout_b = false;  // 4-wave workgroups, V as an HBM ping-pong pair (tinympc_solve_b.hip)
    bool layout_c = false;  // one instance per workgroup, horizon swept in concurrent chunks (tinympc_solve_c.hip)
    // Horizon unrolled at compile time, state in registers, two waves per SIMD (tinympc_solve_d.hip). Wanted for large
    // batches of a shape that is compiled in; each launch still checks that bounds / references are time-invariant and
    // that no family / adaptive rho is active, and otherwise runs the layout-B (or A) kernel on the same HBM state.
    bool layout_d = false;
    bool d_jit = false;     // ... as a run-time specialisation (tinympc_jit.hip) rather than a compiled-in instantiation
    bool d_jit_asked = false;  // the specialiser was asked for the box kernel at setup (its answer may have been a refusal)
    // Large systems, 64 < nx+nu <= 128: tiles of 16 instances on the FP64 matrix cores, state streamed from HBM in the tile's
    // own layout (tinympc_solve_m.hip). The only kernel for these sizes: box path, batched or single, no families /
    // adaptive rho / session.
    bool layout_m = false;
    bool d_varying_jit = false;  // ... and that kernel is a run-time specialisation even if the constant-table one is compiled in
    int d_adapt = -1;       // ... and with adaptive rho
    int d_fam = -1;         // layout D with the cone / linear families (run-time specialised, short horizons): -1 not asked yet, 0 no, 1 yes
    int d_varying = -1;     // layout D with bounds / references that vary over the horizon: -1 not asked yet, 0 no, 1 yes
    // Layout E (tinympc_solve_e.hip, run-time specialised on the families' STRUCTURE): the horizon cut across the wavefronts of a
    // workgroup -- the throughput kernel for the families at horizons layout D cannot hold. Decided per launch
    // (decide_layout_variants): `e_sig` is the structure / table kind the answer `e_ok` belongs to.
    FamilyStructure fs;
    std::string e_sig;
    bool e_ok = false;
    int e_chunk_len = 0, e_wpg = 0;
    size_t e_lds = 0;
    // Layout F (tinympc_solve_f.hip): the latency kernel as a run-time specialisation (shape, chunk plan and the families'
    // structure compiled in); decided per launch like layout E
    std::string f_sig;
    bool f_ok = false;
    FamilyStructure f_fs;
    int f_chunk_len = 0, f_chunks = 0, f_wpg = 0;
    size_t f_lds = 0;
    double *dctab_f = nullptr;   // Phi^(S..4S) | Psi^(S..4S) for layout F's chunk length
    int dctab_f_len = 0;
    double *dclock = nullptr;    // (diagnostic build TINY_CLOCK_STAMP only) per-wavefront clock stamps of the last launch
    double *dctab_e = nullptr;   // Phi^S | Psi^S for layout E's chunk length
    int dctab_e_len = 0;         // ... the chunk length it was built for (0: not built)
    // every row of the bounds / references is the same at all knots (what the verbs last received; defaults are)
    bool xmin_const = true, xmax_const = true, umin_const = true, umax_const = true, xref_const = true, uref_const = true;
    bool tables_const() const { return xmin_const && xmax_const && umin_const && umax_const && xref_const && uref_const; }
    bool zero_copy_tick = false;  // this launch reads x0 from / writes u0 to pinned host memory (mpc_step, small batches)
    bool c_tables = false;  // the chunk tables exist (layout C is possible for this shape and not excluded)
    bool fam_c = false;     // the cone / linear families run in the latency kernel's FAM variant
    int chunk_len = 0, chunk_count = 0, chunk_levels = 0;
    size_t lds_bytes_c = 0;
    double *dctab = nullptr;
    double *dV2 = nullptr;
    int n_cone_x = 0, n_cone_u = 0, n_lin_x = 0, n_lin_u = 0;
    // cone / linear families (host copies of what the verbs received; k_admm_solve_fam consumes `dfam`)
    std::vector<int> Acx, qcx, Acu, qcu;
    std::vector<double> cx, cu, Alin_x, blin_x, Alin_u, blin_u;
    double *dfam = nullptr, *dGC = nullptr, *dGL = nullptr, *dLX = nullptr;
    double *h_x0 = nullptr, *h_u0 = nullptr;  // pinned staging for tinympc_mpc_step_batch (and x0 of single-instance handles)
    // Single-instance handles (batch == 1, what the MEX shim creates): set_x0 only fills the pinned h_x0, the next
    // launch reads it from there (and mirrors it into dx0), and the kernels also write solution + statistics into the
    // pinned h_sol -- the reference's per-tick sequence set_x0 / solve / get_solution then costs ONE launch and ONE
    // synchronisation instead of three synchronous copies around the launch.
    double *h_sol = nullptr;           // [X | U | 4 residuals | iter, status | completion flag]
    // Closed-loop session (tinympc_session_begin / _step / _end): the latency kernel stays resident and takes its ticks
    // from this mailbox in pinned memory (layout: SolveParams::mail).
    double *h_mail = nullptr;          // [64]
    bool session_active = false;
    // references re-sent inside a session that turned out to be the previous ones moved up by one knot (receding horizon):
    // only the new last column travels, with the command (flags 4 / 8); two shifts without a step in between, or any other
    // change, fall back to the full re-read (refs_on_host)
    bool xref_shift = false, uref_shift = false;
    bool session_refs_shifted = false;  // the device copies / tables lag behind the pinned references
    // ONE counter stamps session commands and flag-raising launches alike (both complete by writing their stamp into the
    // same slot of h_sol: a launch after a session must not find its number already there)
    unsigned long long session_seq = 0;  // stamp of the last session command / flag-raising launch
    bool flag_pending = false;
    // ... and set_x_ref / set_u_ref only fill these pinned copies; the next launch's workgroup rebuilds the
    // reference-dependent table rows from them (refresh_reference_tables): a tick with per-tick references
    // (rocket_landing_constraints.m:86-121) is still one launch and one synchronisation.
    double *h_xref = nullptr, *h_uref = nullptr;
    bool refs_on_host = false;         // the pinned references are newer than dXref / dUref and the tables
    bool x0_on_host = false;           // h_x0 is newer than dx0
    int host_sol_state = 0;            // 0: not valid, 1: a launch that writes it is in flight, 2: valid
    bool host_path() const { return batch == 1 && h_sol != nullptr && !layout_d && !layout_m; }  // (layout D writes to device memory only)
    bool state_in_global = false;             // horizon too long for LDS: layout-A kernels work on dscratch
    double *dscratch_state = nullptr;
    bool fam_dirty = true;
    size_t lds_bytes_a = 0;       // layout-A LDS plan (the families kernel always uses layout A)
    bool tables_in_lds_a = false;

    bool use_layout_d() const {
        return layout_d && (tables_const() || d_varying == 1) && (!families_active() || d_fam == 1) && (!st.adaptive_rho || d_adapt == 1);
    }
    bool use_layout_e() const { return e_ok && !st.adaptive_rho && !use_layout_d(); }
    bool use_layout_f() const { return f_ok && !st.adaptive_rho && !use_layout_d() && !use_layout_e(); }
    bool families_active() const {
        return (st.en_state_soc && n_cone_x > 0) || (st.en_input_soc && n_cone_u > 0) ||
               (st.en_state_linear && n_lin_x > 0) || (st.en_input_linear && n_lin_u > 0);
    }
    std::vector<void *> allocs;

    size_t X() const { return (size_t)nx * N; }
    size_t U() const { return (size_t)nu * (N - 1); }
    size_t state_doubles() const { return layout_m ? solve_m_state_doubles(nx, nu, N, groups) : (size_t)groups * (N + 1) * 64; }  // G; row N: per-lane dummy slot
    size_t v_doubles() const { return layout_m ? solve_m_state_doubles(nx, nu, N, groups) : (size_t)groups * v_rows(N) * 64; }     // V (and V2)
    size_t d_doubles() const { return layout_m ? solve_m_state_doubles(nx, nu, N, groups) : (size_t)groups * (N - 1) * IPW * nu; }
};

namespace {

template <typename T>
int dalloc(tinympc_solver *s, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) return fail(TINYMPC_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", sizeof(T) * count, hipGetErrorString(e));
    s->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return TINYMPC_OK;
}

int end_session(tinympc_solver *s);

// Every verb that touches the device passes through here first: a resident session kernel would make it wait forever
// on the handle's stream, so the session is ended (its state is in HBM after every tick) before anything else happens.
int bind_device(tinympc_solver *s) {
    HIP_TRY(hipSetDevice(s->device));
    if (s->session_active) return end_session(s);
    return TINYMPC_OK;
}

// true if every row of the column-major rows x cols matrix holds one value (bit-wise; inf == inf)
bool rows_constant(const double *m, int rows, int cols) {
    for (int c = 1; c < cols; ++c)
        for (int r = 0; r < rows; ++r)
            if (!(m[r + (size_t)c * rows] == m[r])) return false;
    return true;
}

int upload(tinympc_solver *s, double *dst, const double *src, size_t count) {
    HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * count, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));  // the caller keeps ownership of src: copy completes inside the call
    return TINYMPC_OK;
}

int download(tinympc_solver *s, void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TINYMPC_OK;
}

int fill_host_upload(tinympc_solver *s, double *dst, size_t count, double value) {
    std::vector<double> h(count, value);
    return upload(s, dst, h.data(), count);
}

int check_handle(const tinympc_solver *s) {
    if (!s) return fail(TINYMPC_ERR_NOT_INITIALIZED, "Solver not initialized");
    return TINYMPC_OK;
}

int run_precompute(tinympc_solver *s) {
    PrecomputeParams p{};
    p.nx = s->nx; p.nu = s->nu; p.rho = s->rho;
    p.A = s->dA; p.B = s->dB; p.fdyn = s->dfdyn; p.Qd = s->dQd; p.Rd = s->dRd;
    p.Kinf = s->dKinf; p.Pinf = s->dPinf; p.Quu_inv = s->dQuu; p.AmBKt = s->dAmBKt; p.APf = s->dAPf; p.BPf = s->dBPf;
    p.info = s->dinfo; p.scratch = s->dscratch;
    p.use_lds = precompute_scratch_doubles(s->nx, s->nu) <= 6500 ? 1 : 0;
    HIP_TRY(launch_precompute(p, s->stream));
    s->ops_dirty = true;
    s->tables_dirty = true;
    return TINYMPC_OK;
}

// References left in pinned host memory by set_x_ref / set_u_ref -> device copies, the ordinary way.
int flush_host_refs(tinympc_solver *s) {
    int rc;
    if ((rc = upload(s, s->dXref, s->h_xref, s->X()))) return rc;
    if (s->U() && (rc = upload(s, s->dUref, s->h_uref, s->U()))) return rc;
    s->refs_on_host = false;
    s->tables_dirty = true;
    return TINYMPC_OK;
}

int refresh_derived(tinympc_solver *s) {
    if (s->ops_dirty) {
        OperatorParams p{};
        p.nx = s->nx; p.nu = s->nu; p.W = s->W; p.KT = s->KT;
        p.A = s->dA; p.B = s->dB; p.fdyn = s->dfdyn; p.Qd = s->dQd; p.Rd = s->dRd;
        p.Kinf = s->dKinf; p.Quu_inv = s->dQuu; p.AmBKt = s->dAmBKt; p.APf = s->dAPf; p.BPf = s->dBPf;
        p.ops = s->dops;
        HIP_TRY(launch_build_operators(p, s->stream));
        if (s->c_tables) {  // powers of the sweep operators for the chunked kernel
            ChunkTableParams c{};
            c.nx = s->nx; c.nu = s->nu; c.KT = s->KT; c.S = s->chunk_len; c.Lc = s->chunk_levels;
            c.ops = s->dops; c.out = s->dctab;
            HIP_TRY(launch_build_chunk_tables(c, s->stream));
        }
        s->dctab_e_len = 0;  // (layout E's and F's carry matrices are rebuilt on demand, below)
        s->dctab_f_len = 0;
        s->ops_dirty = false;
        s->tables_dirty = true;
    }
    if (s->e_ok && s->dctab_e && s->dctab_e_len != s->e_chunk_len) {  // Phi^S, Psi^S for layout E's chunk length
        ChunkTableParams c{};
        c.nx = s->nx; c.nu = s->nu; c.KT = s->KT; c.S = s->e_chunk_len; c.Lc = 1;
        c.ops = s->dops; c.out = s->dctab_e;
        HIP_TRY(launch_build_chunk_tables(c, s->stream));
        s->dctab_e_len = s->e_chunk_len;
    }
    if (s->f_ok && s->dctab_f && s->dctab_f_len != s->f_chunk_len) {  // powers S .. 4S for layout F's chunk length
        ChunkTableParams c{};
        c.nx = s->nx; c.nu = s->nu; c.KT = s->KT; c.S = s->f_chunk_len; c.Lc = 4;
        c.ops = s->dops; c.out = s->dctab_f;
        HIP_TRY(launch_build_chunk_tables(c, s->stream));
        s->dctab_f_len = s->f_chunk_len;
    }
    if (s->tables_dirty) {
        TableParams p{};
        p.nx = s->nx; p.nu = s->nu; p.N = s->N; p.W = s->W; p.KT = s->KT;
        p.en_state_bound = s->st.en_state_bound; p.en_input_bound = s->st.en_input_bound;
        p.x_min = s->dxmin; p.x_max = s->dxmax; p.u_min = s->dumin; p.u_max = s->dumax;
        p.Xref = s->dXref; p.Uref = s->dUref; p.Pinf = s->dPinf; p.ops = s->dops; p.tables = s->dtables;
        HIP_TRY(launch_build_tables(p, s->stream));
        s->tables_dirty = false;
    }
    return TINYMPC_OK;
}

// The ACTIVE cones in list order (state cones, then input cones) with their rounds, and the linear rows per side: what layout E
// is specialised on (FamilyStructure, tinympc_device.h). `mu` receives the slopes in the same order.
FamilyStructure family_structure(const tinympc_solver *s, double *mu = nullptr) {
    FamilyStructure fs;
    const bool cone_x = s->st.en_state_soc && s->n_cone_x > 0, cone_u = s->st.en_input_soc && s->n_cone_u > 0;
    unsigned long long used = 0;  // lanes taken by the cones of the current round
    auto add = [&](bool on, const std::vector<int> &Ac, const std::vector<int> &qc, const std::vector<double> &c, int base) {
        if (!on) return;
        for (size_t k = 0; k < Ac.size() && fs.ncone < MAX_CONES; ++k) {
            const int first = base + Ac[k], last = first + qc[k] - 1;
            unsigned long long lanes = 0;
            for (int r = first; r <= last; ++r) lanes |= 1ull << (r & 63);
            if (fs.ncone == 0) fs.nround = 1;
            if (lanes & used) {  // overlaps an earlier cone of this round: upstream projects one after the other
                fs.nround += 1;
                used = 0;
            }
            used |= lanes;
            fs.cone[fs.ncone][0] = fs.nround - 1;
            fs.cone[fs.ncone][1] = first;
            fs.cone[fs.ncone][2] = last;
            if (mu) mu[fs.ncone] = c[k];
            fs.ncone += 1;
        }
    };
    add(cone_x, s->Acx, s->qcx, s->cx, 0);
    add(cone_u, s->Acu, s->qcu, s->cu, s->nx);
    fs.nlx = (s->st.en_state_linear && s->n_lin_x > 0) ? s->n_lin_x : 0;
    fs.nlu = (s->st.en_input_linear && s->n_lin_u > 0) ? s->n_lin_u : 0;
    return fs;
}

// Per-lane description of the cone / linear families for k_admm_solve_fam (layout: fam_doubles()).
// Masks and user coefficients only -- no solver arithmetic happens here.
int refresh_families(tinympc_solver *s) {
    const int W = s->W, KT = s->KT, nx = s->nx, nu = s->nu, nxu = nx + nu;
    if (!s->dfam) {
        int rc;
        if ((rc = dalloc(s, &s->dfam, fam_doubles(W, KT)))) return rc;
        if ((rc = dalloc(s, &s->dGC, s->v_doubles()))) return rc;
        if ((rc = dalloc(s, &s->dGL, s->v_doubles()))) return rc;
        if ((rc = dalloc(s, &s->dLX, s->v_doubles()))) return rc;
        HIP_TRY(hipMemsetAsync(s->dGC, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dGL, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dLX, 0, sizeof(double) * s->v_doubles(), s->stream));
        s->fam_dirty = true;
    }
    if (!s->fam_dirty) return TINYMPC_OK;
    std::vector<double> f(fam_doubles(W, KT), 0.0);
    double *role = f.data(), *mu = role + W, *famc = mu + W, *faml = famc + W;
    double *Cn = faml + W, *Ct = Cn + (size_t)W * KT, *Ty = Ct + (size_t)W * KT, *lin = Ty + (size_t)W * KT;
    const bool cone_x = s->st.en_state_soc && s->n_cone_x > 0, cone_u = s->st.en_input_soc && s->n_cone_u > 0;
    const bool lin_x = s->st.en_state_linear && s->n_lin_x > 0, lin_u = s->st.en_input_linear && s->n_lin_u > 0;
    for (int r = 0; r < nxu; ++r) {
        const bool is_x = r < nx;
        famc[r] = (is_x ? cone_x : cone_u) ? 1.0 : 0.0;
        faml[r] = (is_x ? lin_x : lin_u) ? 1.0 : 0.0;
        for (int k = 0; k < nxu; ++k)
            if ((k < nx) == is_x) Ty[(size_t)r * KT + k] = 1.0;
    }
    // cones, round by round (family_structure: cones of one round are pairwise disjoint): round 0 into the arrays every kernel
    // reads, later rounds -- they exist only where cones share rows -- behind them for the kernels that walk rounds
    {
        std::vector<double> cmu(MAX_CONES, 0.0);
        const FamilyStructure fs = family_structure(s, cmu.data());
        f[fam_nround_offset(W, KT)] = (double)(fs.nround > 0 ? fs.nround : 1);
        for (int c = 0; c < fs.ncone; ++c) {
            const int q = fs.cone[c][0], first = fs.cone[c][1], last = fs.cone[c][2];
            double *rl = role, *m = mu, *cn = Cn, *ct = Ct;
            if (q >= 1) {
                rl = f.data() + fam_round_offset(W, KT, q);
                m = rl + W;
                cn = m + W;
                ct = cn + (size_t)W * KT;
            }
            for (int r = first; r <= last; ++r) {
                rl[r] = (r == last) ? 2.0 : 1.0;
                m[r] = cmu[c];
                for (int k = first; k < last; ++k) cn[(size_t)r * KT + k] = 1.0;
                ct[(size_t)r * KT + last] = 1.0;
            }
        }
    }
    (void)cone_x;
    (void)cone_u;
    const int nlx = lin_x ? s->n_lin_x : 0, nlu = lin_u ? s->n_lin_u : 0;
    const int nl = nlx > nlu ? nlx : nlu;
    lin[0] = (double)nl;
    const double inf = std::numeric_limits<double>::infinity();
    for (int k = 0; k < MAX_LIN_ROWS; ++k) {
        double *ak = lin + 1 + (size_t)(3 * k + 0) * W, *bk = ak + W, *nk = bk + W;
        double nrm_x = 0.0, nrm_u = 0.0;
        if (k < nlx) for (int c = 0; c < nx; ++c) { const double a = s->Alin_x[k + (size_t)c * s->n_lin_x]; nrm_x += a * a; }
        if (k < nlu) for (int c = 0; c < nu; ++c) { const double a = s->Alin_u[k + (size_t)c * s->n_lin_u]; nrm_u += a * a; }
        for (int r = 0; r < W; ++r) {
            ak[r] = 0.0; bk[r] = inf; nk[r] = 1.0;
            if (r < nx && k < nlx) { ak[r] = s->Alin_x[k + (size_t)r * s->n_lin_x]; bk[r] = s->blin_x[k]; nk[r] = nrm_x; }
            if (r >= nx && r < nxu && k < nlu) { ak[r] = s->Alin_u[k + (size_t)(r - nx) * s->n_lin_u]; bk[r] = s->blin_u[k]; nk[r] = nrm_u; }
        }
    }
    (void)family_structure(s, f.data() + fam_cone_mu_offset(W, KT));  // slopes of the active cones, in list order (layout E)
    int rc = upload(s, s->dfam, f.data(), f.size());
    if (rc) return rc;
    s->fam_dirty = false;
    return TINYMPC_OK;
}

// Single-instance launches of the latency kernel end by writing a sequence number behind the solution in pinned host
// memory; tinympc_synchronize polls it (a few hundred nanoseconds after the kernel's last store) instead of sleeping in
// hipStreamSynchronize (whose wake-up costs several microseconds of a ~25 us tick).
void arm_completion_flag(tinympc_solver *s, SolveParams &p) {
    if (!p.host_sol) return;
    s->session_seq += 1;
    p.host_seq = (double)s->session_seq;
    s->flag_pending = true;
}

// Layout D's variants beyond the constant-table box path are decided when first needed (they may have to be specialised,
// which takes seconds): time-varying tables and the cone / linear families. Called by everything that asks use_layout_d()
// before a launch, so that the answer does not change between that question and the launch itself.
void decide_layout_d_variants(tinympc_solver *s) {
    if (!s->layout_d) return;
    if (s->d_varying < 0 && !s->tables_const()) {
        // is there a kernel for per-knot tables (compiled in -- 16-lane form only -- or specialised now)? Otherwise these launches
        // run on layout B / A, as before.
        if (s->W == 16 && !s->d_jit && solve_d_supported(s->nx, s->nu, s->N, false)) {
            s->d_varying = 1;
        } else {
            s->d_varying_jit = solve_jit_supported(s->W, s->nx, s->nu, s->N, false);
            s->d_varying = s->d_varying_jit ? 1 : 0;
        }
    }
    if (s->st.adaptive_rho && !s->families_active())
        s->d_adapt = (s->W == 16 && solve_jit_supported(s->W, s->nx, s->nu, s->N, s->tables_const(), false, true)) ? 1 : 0;
    if (s->families_active() && !s->st.adaptive_rho) {
        // families: a run-time specialisation (16-lane form, horizons whose five register pairs per knot fit)? Asked every
        // time -- the answer is cached inside -- because it also depends on the tables' kind.
        // (cones that share rows need the round-by-round projection, which layout D's families variant and the latency kernel
        // do not have: layout E or k_admm_solve_fam run those)
        s->d_fam = (s->W == 16 && family_structure(s).nround <= 1 && solve_jit_supported(s->W, s->nx, s->nu, s->N, s->tables_const(), true)) ? 1 : 0;
    }
}

// Layout E for the families where layout D has no kernel (long horizons): asked whenever the structure of the families or the
// kind of the tables changed (the kernel is specialised on both; compiling takes seconds the first time, the answer is cached
// inside tinympc_jit.hip). TINYMPC_LAYOUT=E forces it at any batch size (tests), any other value excludes it.
int decide_layout_e(tinympc_solver *s) {
    const bool fam = s->families_active();
    // (single-instance handles exchange x0 / the solution through pinned host memory, which only the latency kernel serves)
    const bool possible = s->W == 16 && !s->layout_m && !s->st.adaptive_rho && s->batch > 1;
    // families: wherever layout D has no kernel; box path: horizons for which the specialiser has no layout-D kernel at all (no plan,
    // or a plan whose code object spilled) -- long horizons, where layouts B / A are left with one or two wavefronts per CU
    bool want = possible && !s->use_layout_d() &&
                (fam ? s->batch >= kLayoutEBatchMin : (s->d_jit_asked && !s->layout_d && s->batch > kLayoutCBatchMax && s->N >= 26));
    if (const char *env = getenv("TINYMPC_LAYOUT")) want = (env[0] == 'E' || env[0] == 'e') && possible;
    if (!want) {
        s->e_ok = false;
        s->e_sig.clear();
        return TINYMPC_OK;
    }
    const FamilyStructure fs = fam ? family_structure(s) : FamilyStructure();
    std::string sig = s->tables_const() ? "ct|" : "var|";
    for (int c = 0; c < fs.ncone; ++c) sig += std::to_string(fs.cone[c][0]) + "," + std::to_string(fs.cone[c][1]) + "," + std::to_string(fs.cone[c][2]) + ";";
    sig += "|" + std::to_string(fs.nlx) + "," + std::to_string(fs.nlu) + (fam ? "|fam" : "|box");
    if (sig == s->e_sig) return TINYMPC_OK;
    s->e_sig = sig;
    s->fs = fs;
    s->e_ok = solve_e_supported(s->nx, s->nu, s->N, s->tables_const(), fam, fs) &&
              solve_e_plan(s->nx, s->nu, s->N, s->tables_const(), fam, fs, &s->e_chunk_len, &s->e_wpg, &s->e_lds);
    if (s->e_ok && !s->dctab_e) {
        int rc = dalloc(s, &s->dctab_e, chunk_table_doubles(s->nx, 1));
        if (rc) return rc;
    }
    return TINYMPC_OK;
}

// Layout F for what the latency kernel (layout C) serves: single solves and small batches. The kernel is specialised on the
// shape, the kind of the tables and the structure of the families, so it is asked whenever one of them changed (seconds the
// first time; cached inside tinympc_jit.hip). TINYMPC_LAYOUT=F forces it at any batch size (tests), any other value excludes it.
int decide_layout_f(tinympc_solver *s) {
    const bool fam = s->families_active();
    const bool possible = s->W == 16 && !s->layout_m && !s->st.adaptive_rho && !s->session_active && s->N >= 6;
    // Default: the families at small batches -- rocket landing N=100, one instance: 4.5 us per iteration against 6.55 on the
    // round-1 latency kernel. The box path stays on layout C (quadrotor N=50: 2.9 us against 3.4 here: chunks of two slots leave the
    // carry scans most of the iteration).
    bool want = possible && fam && !s->use_layout_d() && !s->use_layout_e() && s->batch < kLayoutEBatchMin && s->fam_c;
    if (const char *env = getenv("TINYMPC_LAYOUT")) want = (env[0] == 'F' || env[0] == 'f') && possible;
    if (!want) {
        s->f_ok = false;
        s->f_sig.clear();
        return TINYMPC_OK;
    }
    const FamilyStructure fs = fam ? family_structure(s) : FamilyStructure();
    std::string sig = s->tables_const() ? "ct|" : "var|";
    for (int c = 0; c < fs.ncone; ++c) sig += std::to_string(fs.cone[c][0]) + "," + std::to_string(fs.cone[c][1]) + "," + std::to_string(fs.cone[c][2]) + ";";
    sig += "|" + std::to_string(fs.nlx) + "," + std::to_string(fs.nlu) + (fam ? "|fam" : "|box");
    if (sig == s->f_sig) return TINYMPC_OK;
    s->f_sig = sig;
    s->f_fs = fs;
    s->f_ok = solve_f_supported(s->nx, s->nu, s->N, s->tables_const(), fam, fs) &&
              solve_f_plan(s->nx, s->nu, s->N, s->tables_const(), fam, fs, &s->f_chunk_len, &s->f_chunks, &s->f_wpg, &s->f_lds);
    if (s->f_ok && !s->dctab_f) {
        int rc = dalloc(s, &s->dctab_f, chunk_table_doubles(s->nx, 4));
        if (rc) return rc;
    }
    return TINYMPC_OK;
}

int launch(tinympc_solver *s, bool timed) {
    int rc;
    s->flag_pending = false;
    decide_layout_d_variants(s);
    if ((rc = decide_layout_e(s))) return rc;
    if ((rc = decide_layout_f(s))) return rc;
    const bool fam = s->families_active();
    const bool adaptive = s->st.adaptive_rho != 0;
    // k_build_adapt reads the device copy of the references before the solve kernel starts; layout F has no in-kernel staging of
    // references left in pinned host memory (layout C does): bring the device copies and the tables up to date the ordinary way
    if (s->refs_on_host && (adaptive || s->use_layout_f())) {
        if ((rc = flush_host_refs(s))) return rc;
    }
    if ((rc = refresh_derived(s))) return rc;
    if (adaptive && fam)
        return fail(TINYMPC_ERR_UNSUPPORTED, "adaptive_rho together with cone / linear constraint families is not supported");
    if (s->layout_m && (adaptive || fam))
        return fail(TINYMPC_ERR_UNSUPPORTED, "systems with nx+nu > 64 support box constraints only (no cone / linear families, no adaptive_rho)");
    if (adaptive) {  // tiny tables from the current cache, sensitivities and Xref; rebuilt per launch (a few microseconds)
        AdaptTableParams a{};
        a.nx = s->nx; a.nu = s->nu; a.N = s->N; a.W = s->W; a.KT = s->KT;
        a.A = s->dA; a.B = s->dB; a.Pinf = s->dPinf; a.dK = s->ddK; a.dP = s->ddP; a.Xref = s->dXref; a.out = s->dadapt;
        HIP_TRY(launch_build_adapt(a, s->stream));
    }
    if (fam && (rc = refresh_families(s))) return rc;
    SolveParams p{};
    p.nx = s->nx; p.nu = s->nu; p.N = s->N; p.batch = s->batch;
    p.max_iter = s->st.max_iter; p.check_termination = s->st.check_termination;
    p.rho = s->rho; p.abs_pri_tol = s->st.abs_pri_tol; p.abs_dua_tol = s->st.abs_dua_tol;
    p.ops = s->dops; p.tables = s->dtables; p.x0 = s->dx0;
    p.groups = s->groups;
    p.G = s->dG; p.V = s->dV; p.V2 = s->dV2; p.D = s->dD; p.sol_x = s->dsolx; p.sol_u = s->dsolu;
    p.istats = s->distats; p.dstats = s->ddstats;
    p.tables_in_lds = s->tables_in_lds ? 1 : 0;
    p.fam = s->dfam; p.GC = s->dGC; p.GL = s->dGL; p.LX = s->dLX;
    p.scratch = s->state_in_global ? s->dscratch_state : nullptr;
#ifdef TINY_CLOCK_STAMP  // diagnostic build (tools/clock_check.py): layout D stamps its iteration loop into this buffer
    if (!s->state_in_global) {
        if (!s->dclock && (rc = dalloc(s, &s->dclock, (size_t)8 * s->groups))) return rc;
        p.scratch = s->dclock;
    }
#endif
    p.scratch_stride = state_scratch_doubles(s->nu, s->N, s->W);
    p.const_tables = s->tables_const() ? 1 : 0;
    if (s->zero_copy_tick && !s->use_layout_d() && !s->use_layout_e() && !s->layout_m) {  // set by tinympc_mpc_step_batch for the duration of one launch
        p.x0 = s->h_x0;
        p.x0_mirror = s->dx0;
        p.u0_host = s->h_u0;
    }
    if (s->host_path()) {
        if (s->x0_on_host) {
            p.x0 = s->h_x0;
            p.x0_mirror = s->dx0;
            s->x0_on_host = false;  // the kernel mirrors it into dx0
        }
        if (s->st.max_iter > 0) {   // (a 0-iteration solve writes nothing anywhere)
            p.host_sol = s->h_sol;
            s->host_sol_state = 1;
        }
    }
    if (s->refs_on_host) {  // (host_path() handles only: batch == 1, one workgroup)
        p.href_x = s->h_xref; p.href_u = s->h_uref;
        p.dXref = s->dXref; p.dUref = s->dUref; p.Pinf = s->dPinf;
        s->refs_on_host = false;  // the kernel brings the tables and the device copies up to date
    }
    p.adapt = s->dadapt; p.rho_inst = s->drho_inst;
    p.rho_min = s->st.adaptive_rho_min; p.rho_max = s->st.adaptive_rho_max; p.rho_clip = s->st.adaptive_rho_enable_clipping;
    if (timed) HIP_TRY(hipEventRecord(s->ev0, s->stream));
    if (s->layout_m) {
        HIP_TRY(launch_solve_m(p, s->stream));
    } else if (adaptive && s->use_layout_d()) {
        p.adaptive = 1;
        HIP_TRY(launch_solve_jit(p, s->W, s->stream));
    } else if (adaptive) {
        // layout A's LDS plan; shares the persistent state (G, canonical V, D) with the other kernels
        p.tables_in_lds = s->tables_in_lds_a ? 1 : 0;
        HIP_TRY(launch_solve_adapt(p, s->W, s->KT, s->lds_bytes_a, s->stream));
    } else if (fam && s->use_layout_d()) {
        p.families = 1;
        HIP_TRY(launch_solve_jit(p, s->W, s->stream));
    } else if (s->use_layout_f()) {  // the specialised latency kernel (families or box path alike)
        p.families = fam ? 1 : 0;
        p.ctab = s->dctab_f; p.chunk_len = s->f_chunk_len; p.chunk_count = s->f_chunks; p.chunk_levels = 4;
        arm_completion_flag(s, p);
        HIP_TRY(launch_solve_f(p, s->f_fs, s->stream));
    } else if (fam && s->use_layout_e()) {
        p.families = 1;
        p.ctab = s->dctab_e; p.chunk_len = s->e_chunk_len; p.chunk_count = s->e_wpg; p.chunk_levels = 1;
        HIP_TRY(launch_solve_e(p, s->fs, s->stream));
    } else if (fam && s->fam_c && family_structure(s).nround <= 1) {
        // the latency kernel carries the families itself (same HBM state as k_admm_solve_fam)
        p.ctab = s->dctab; p.chunk_len = s->chunk_len; p.chunk_count = s->chunk_count; p.chunk_levels = s->chunk_levels;
        p.families = 1;
        arm_completion_flag(s, p);
        HIP_TRY(launch_solve_c(p, s->W, s->KT, s->lds_bytes_c, s->stream));
    } else if (fam) {
        // The families kernel shares the persistent state (G, canonical V, D) with layouts A and B, so a
        // handle can switch between them from one solve to the next.
        p.tables_in_lds = s->tables_in_lds_a ? 1 : 0;
        HIP_TRY(launch_solve_fam(p, s->W, s->KT, s->lds_bytes_a, s->stream));
    } else if (s->use_layout_e()) {  // (box path: horizons beyond layout D's plans)
        p.ctab = s->dctab_e; p.chunk_len = s->e_chunk_len; p.chunk_count = s->e_wpg; p.chunk_levels = 1;
        HIP_TRY(launch_solve_e(p, s->fs, s->stream));
    } else if (s->use_layout_d()) {
        HIP_TRY((s->d_jit || (!p.const_tables && s->d_varying_jit)) ? launch_solve_jit(p, s->W, s->stream)
                         : s->W == 64 ? launch_solve_dx(p, s->stream) : s->W == 32 ? launch_solve_dw(p, s->stream) : launch_solve_d(p, s->stream));
    } else if (s->layout_c) {
        p.ctab = s->dctab; p.chunk_len = s->chunk_len; p.chunk_count = s->chunk_count; p.chunk_levels = s->chunk_levels;
        arm_completion_flag(s, p);
        HIP_TRY(launch_solve_c(p, s->W, s->KT, s->lds_bytes_c, s->stream));
    } else if (s->layout_b) {
        HIP_TRY(launch_solve_b(p, s->W, s->KT, s->lds_bytes, s->stream));
    } else {
        HIP_TRY(launch_solve(p, s->W, s->KT, s->lds_bytes, s->stream));
    }
    if (timed) HIP_TRY(hipEventRecord(s->ev1, s->stream));
    return TINYMPC_OK;
}

// ---- closed-loop session ------------------------------------------------------------------------------------
constexpr double kSessionIdleSeconds = 2.0;  // the resident kernel leaves on its own after this long without a command

void write_command(tinympc_solver *s, int flags, const double *x0) {
    // payload: 0 flags | x0 (nx) | new last column of x_ref (flag 4) | new last column of u_ref (flag 8); line l = [7 payload |
    // stamp]. Payload before stamp, line by line (x86 keeps the order of stores; the fences keep the compiler from
    // reordering them).
    double pay[56];
    int npay = 0;
    pay[npay++] = (double)flags;
    for (int i = 0; i < s->nx; ++i) pay[npay++] = x0 ? x0[i] : 0.0;
    if (flags & 4) for (int i = 0; i < s->nx; ++i) pay[npay++] = s->h_xref[(size_t)(s->N - 1) * s->nx + i];
    if (flags & 8) for (int i = 0; i < s->nu; ++i) pay[npay++] = s->h_uref[(size_t)(s->N - 2) * s->nu + i];
    volatile double *m = s->h_mail;
    const double stamp = (double)(++s->session_seq);
    const int nlines = (npay + 6) / 7;
    for (int l = 0; l < nlines; ++l) {
        for (int q = 7 * l; q < 7 * l + 7 && q < npay; ++q) m[8 * l + q % 7] = pay[q];
        std::atomic_thread_fence(std::memory_order_release);
        m[8 * l + 7] = stamp;
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
}

int launch_session_kernel(tinympc_solver *s) {
    int rc = refresh_derived(s);
    if (rc) return rc;
    const bool fam = s->families_active();
    if (fam && (rc = refresh_families(s))) return rc;
    SolveParams p{};
    p.nx = s->nx; p.nu = s->nu; p.N = s->N; p.batch = 1;
    p.max_iter = s->st.max_iter; p.check_termination = s->st.check_termination;
    p.rho = s->rho; p.abs_pri_tol = s->st.abs_pri_tol; p.abs_dua_tol = s->st.abs_dua_tol;
    p.ops = s->dops; p.tables = s->dtables; p.x0 = s->dx0; p.x0_mirror = s->dx0;
    p.groups = s->groups;
    p.G = s->dG; p.V = s->dV; p.V2 = s->dV2; p.D = s->dD; p.sol_x = s->dsolx; p.sol_u = s->dsolu;
    p.istats = s->distats; p.dstats = s->ddstats;
    p.fam = s->dfam; p.GC = s->dGC; p.GL = s->dGL; p.LX = s->dLX;
    p.const_tables = s->tables_const() ? 1 : 0;
    p.host_sol = s->h_sol;
    p.href_x = s->h_xref; p.href_u = s->h_uref; p.dXref = s->dXref; p.dUref = s->dUref; p.Pinf = s->dPinf;  // (re-read on request)
    s->refs_on_host = false;  // the prologue stages them
    s->xref_shift = s->uref_shift = false;
    p.ctab = s->dctab; p.chunk_len = s->chunk_len; p.chunk_count = s->chunk_count; p.chunk_levels = s->chunk_levels;
    p.families = fam ? 1 : 0;
    p.mail = s->h_mail;
    p.session_expect = (double)(s->session_seq + 1);
    p.session_idle = (unsigned long long)(kSessionIdleSeconds * 1e8);
    HIP_TRY(launch_solve_c(p, s->W, s->KT, s->lds_bytes_c, s->stream));
    return TINYMPC_OK;
}

int end_session(tinympc_solver *s) {
    if (!s->session_active) return TINYMPC_OK;
    write_command(s, 1, nullptr);  // stop
    s->session_active = false;     // (before anything that could come back here)
    if (s->session_refs_shifted || s->xref_shift || s->uref_shift) s->refs_on_host = true;  // device copies / tables lag: restage
    s->session_refs_shifted = s->xref_shift = s->uref_shift = false;
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TINYMPC_OK;
}

void destroy(tinympc_solver *s) {
    if (!s) return;
    // teardown is best effort: errors here have nowhere useful to go
    (void)hipSetDevice(s->device);
    if (s->session_active) (void)end_session(s);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (void *q : s->allocs) (void)hipFree(q);
    if (s->h_sol) (void)hipHostFree(s->h_sol);
    if (s->h_mail) (void)hipHostFree(s->h_mail);
    if (s->h_xref) (void)hipHostFree(s->h_xref);
    if (s->h_uref) (void)hipHostFree(s->h_uref);
    if (s->h_x0) (void)hipHostFree(s->h_x0);
    if (s->h_u0) (void)hipHostFree(s->h_u0);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

}  // namespace

extern "C" {

const char *tinympc_last_error(void) { return last_error_slot().c_str(); }
int tinympc_abi_version(void) { return TINYMPC_ABI_VERSION; }

int tinympc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tinympc_setup_batch(tinympc_solver **out, const double *A, const double *B, const double *fdyn,
                        const double *Q, const double *R, double rho, int nx, int nu, int N, int batch,
                        int device, int verbose) {
    if (!out) return fail(TINYMPC_ERR_INVALID_INPUT, "setup: output handle pointer is NULL");
    *out = nullptr;
    if (!A || !B || !Q || !R) return fail(TINYMPC_ERR_INVALID_INPUT, "setup requires A, B, Q, R");
    if (nx < 1 || nu < 1) return fail(TINYMPC_ERR_INVALID_INPUT, "setup: nx and nu must be >= 1 (got %d, %d)", nx, nu);
    if (N < 2) return fail(TINYMPC_ERR_INVALID_INPUT, "setup: N must be >= 2 (TinyMPC.m:52), got %d", N);
    if (batch < 1) return fail(TINYMPC_ERR_INVALID_INPUT, "setup: batch must be >= 1, got %d", batch);
    int W = 0, KT = 0;
    const bool large = solve_m_supported(nx, nu);
    if (large) {
        W = KT = 128;  // geometry of the operators / tables; the kernel works on 16-instance tiles
    } else if (!choose_geometry(nx, nu, &W, &KT)) {
        return fail(TINYMPC_ERR_UNSUPPORTED, "nx+nu = %d: systems beyond 128 rows are not supported by this build", nx + nu);
    }
    int ndev = tinympc_device_count();
    if (ndev < 1) return fail(TINYMPC_ERR_NO_DEVICE, "no HIP device visible: the HIP path has no CPU fallback");
    if (device >= ndev) return fail(TINYMPC_ERR_INVALID_INPUT, "device %d out of range (have %d)", device, ndev);

    tinympc_solver *s = new (std::nothrow) tinympc_solver();
    if (!s) return fail(TINYMPC_ERR_ALLOC, "out of host memory");
    if (device < 0) {
        if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
    } else {
        s->device = device;
    }
    s->nx = nx; s->nu = nu; s->N = N; s->batch = batch; s->rho = rho;
    s->W = W; s->KT = KT;
    s->layout_m = large;
    s->IPW = large ? 16 : 64 / W;
    s->groups = (batch + s->IPW - 1) / s->IPW;  // wave groups; tiles of 16 instances for the large-system kernel
    // tiny_set_default_settings (tiny_api.cpp:213-231, tiny_api_constants.hpp:5-10)
    s->st = Settings{1e-3, 1e-3, 1000, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, 100.0, 1};

    int rc = TINYMPC_OK;
#define TRY(x)            \
    do {                  \
        rc = (x);         \
        if (rc) {         \
            destroy(s);   \
            return rc;    \
        }                 \
    } while (0)
#define HIP_TRY_S(expr)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            destroy(s);                                                                              \
            return fail(TINYMPC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));            \
        }                                                                                            \
    } while (0)

    HIP_TRY_S(hipSetDevice(s->device));
    HIP_TRY_S(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY_S(hipEventCreate(&s->ev0));
    HIP_TRY_S(hipEventCreate(&s->ev1));

    // LDS plan: ADMM state always; the per-knot tables too when the total fits the 160 KB of a CU.
    // (the large-system kernel plans its own LDS: everything below describes layouts A - D and stays unused for it)
    const int Wp = large ? 64 : W;
    const size_t with_tables = solve_lds_bytes(nx, nu, N, Wp, true);
    const size_t without = solve_lds_bytes(nx, nu, N, Wp, false);
    constexpr size_t kLdsMax = 160 * 1024;
    // Horizons whose state does not fit a CU's LDS run the same kernels on an HBM working copy (GMEM variant).
    s->state_in_global = !large && without > kLdsMax;
    // Two workgroups per CU need <= 80 KB each; prefer LDS tables whenever they do not cost a workgroup slot.
    const size_t slots_without = s->state_in_global ? 0 : kLdsMax / without;
    const size_t slots_with = kLdsMax / with_tables;
    // (up to four: one wavefront per SIMD. Wide systems -- 2 or 1 instances per wavefront -- have short state arrays and
    // long tables; keeping the tables in L2 doubles their wavefronts per CU, profiles/r02_wide_sweep.txt)
    s->tables_in_lds = !s->state_in_global && (with_tables <= kLdsMax) && (slots_with >= (slots_without > 4 ? 4 : slots_without));
    s->lds_bytes = s->state_in_global ? 0 : (s->tables_in_lds ? with_tables : without);
    s->lds_bytes_a = s->lds_bytes;
    s->tables_in_lds_a = s->tables_in_lds;
    // Layout A keeps all ADMM state in LDS (2 wavefronts per CU at quadrotor size); layout B keeps V as an
    // L2-resident ping-pong pair in HBM and fits 4 wavefronts per CU. Measured on MI355X (quadrotor N=50):
    // B is 1.72x faster on a GPU-filling batch (5.56 vs 9.57 ms per 8192 x 200 iterations) and also
    // 13 % faster for a single instance (fewer LDS instructions per step), so B is used whenever it fits.
    // TINYMPC_LAYOUT=A|B overrides the choice (kernel A/B experiments, and the tests cover both).
    {
        const size_t b_bytes = solve_b_lds_bytes(nx, nu, N, W);
        const bool b_possible = (W == 16) && (N >= 8) && (b_bytes <= kLdsMax);
        bool want_b = b_possible;
        if (const char *env = getenv("TINYMPC_LAYOUT")) {
            if (env[0] == 'A' || env[0] == 'a') want_b = false;
            if ((env[0] == 'B' || env[0] == 'b') && b_possible) want_b = true;
        }
        if (want_b) {
            s->layout_b = true;
            s->tables_in_lds = true;
            s->lds_bytes = b_bytes;
        }
        // Layout C gives every instance a whole workgroup and cuts the horizon into 16 concurrent chunks: a
        // 2-3x shorter iteration for one instance, lower throughput once the batch fills the chip's wave slots.
        // Default for small batches; TINYMPC_LAYOUT=C forces it, =A / =B exclude it.
        chunk_plan(N, &s->chunk_len, &s->chunk_count, &s->chunk_levels);
        // (single-instance handles stage their references in LDS, see k_admm_solve_c's prologue)
        s->lds_bytes_c = batch == 1 ? solve_c_lds_bytes_refs(nx, nu, N, s->chunk_levels) : solve_c_lds_bytes(nx, s->chunk_levels);
        const bool c_possible = (W == 16) && (s->chunk_len <= 8) && (s->lds_bytes_c <= kLdsMax);
        bool want_c = c_possible && batch <= kLayoutCBatchMax;
        if (const char *env = getenv("TINYMPC_LAYOUT")) {
            if (env[0] == 'C' || env[0] == 'c') want_c = c_possible;
            else if (env[0] == 'A' || env[0] == 'a' || env[0] == 'B' || env[0] == 'b' || env[0] == 'D' || env[0] == 'd') want_c = false;
        }
        s->layout_c = want_c;
        // Layout D (compile-time shape, duals in registers, two wavefronts per SIMD): default above the latency kernel's
        // range -- for wide systems, which have no latency kernel, from 16 instances up --; TINYMPC_LAYOUT=D forces it at any
        // batch size (tests), =A / =B / =C exclude it. The shapes compiled into the library run as they are; every other
        // shape that fits the plan is specialised at run time (tinympc_jit.hip: hiprtc, cached), now, so that a failure
        // simply leaves the handle on layout B / A.
        const bool d_compiled = (W == 16 && solve_d_supported(nx, nu, N, true)) || (W == 32 && solve_dw_supported(nx, nu, N, true)) ||
                                (W == 64 && solve_dx_supported(nx, nu, N, true));
        bool want_d = (W == 16) ? !want_c : (batch >= 16 && !large);
        if (const char *env = getenv("TINYMPC_LAYOUT")) want_d = (env[0] == 'D' || env[0] == 'd');
        if (want_d && !d_compiled) {
            HIP_TRY_S(hipSetDevice(s->device));
            s->d_jit_asked = true;
            s->d_jit = solve_jit_supported(W, nx, nu, N, true);
            want_d = s->d_jit;
        }
        s->layout_d = want_d;
        if (want_d) s->layout_c = false;
        // The families: k_admm_solve_fam keeps the whole ADMM state in LDS (layout A), so a long horizon leaves it
        // one wavefront = 4 instances per CU; the latency kernel then wins at EVERY batch size (rocket N=100:
        // 30 vs 15.6 M iterations/s at 4096 instances, profiles/r01d_rocket_sweep.txt). Otherwise as for the box path.
        bool c_excluded = false;
        if (const char *env = getenv("TINYMPC_LAYOUT")) c_excluded = (env[0] == 'A' || env[0] == 'a' || env[0] == 'B' || env[0] == 'b' || env[0] == 'D' || env[0] == 'd');
        const size_t fam_a_waves_per_cu = s->state_in_global ? 1 : kLdsMax / (s->lds_bytes_a ? s->lds_bytes_a : kLdsMax);
        s->fam_c = c_possible && !c_excluded && (want_c || fam_a_waves_per_cu <= 1);
        s->c_tables = s->layout_c || s->fam_c;
    }

    const size_t X = s->X(), U = s->U();
    TRY(dalloc(s, &s->dA, (size_t)nx * nx)); TRY(dalloc(s, &s->dB, (size_t)nx * nu)); TRY(dalloc(s, &s->dfdyn, nx));
    TRY(dalloc(s, &s->dQd, nx)); TRY(dalloc(s, &s->dRd, nu));
    TRY(dalloc(s, &s->dKinf, (size_t)nu * nx)); TRY(dalloc(s, &s->dPinf, (size_t)nx * nx));
    TRY(dalloc(s, &s->dQuu, (size_t)nu * nu)); TRY(dalloc(s, &s->dAmBKt, (size_t)nx * nx));
    TRY(dalloc(s, &s->dAPf, nx)); TRY(dalloc(s, &s->dBPf, nu)); TRY(dalloc(s, &s->dinfo, 4));
    TRY(dalloc(s, &s->dscratch, precompute_scratch_doubles(nx, nu) + 8));
    TRY(dalloc(s, &s->dQfull, (size_t)nx * nx)); TRY(dalloc(s, &s->dRfull, (size_t)nu * nu));
    TRY(dalloc(s, &s->ddK, (size_t)nu * nx)); TRY(dalloc(s, &s->ddP, (size_t)nx * nx));
    TRY(dalloc(s, &s->dadapt, adapt_doubles(W, KT))); TRY(dalloc(s, &s->drho_inst, batch));
    if (s->c_tables) TRY(dalloc(s, &s->dctab, chunk_table_doubles(nx, s->chunk_levels)));
    TRY(dalloc(s, &s->dlqr_scratch, lqr_scratch_doubles(nx, nu) + 8)); TRY(dalloc(s, &s->dlqr_out, 3 * s->cache_doubles()));
    TRY(dalloc(s, &s->dxmin, X)); TRY(dalloc(s, &s->dxmax, X)); TRY(dalloc(s, &s->dumin, U)); TRY(dalloc(s, &s->dumax, U));
    TRY(dalloc(s, &s->dXref, X)); TRY(dalloc(s, &s->dUref, U));
    TRY(dalloc(s, &s->dops, ops_doubles(W, KT))); TRY(dalloc(s, &s->dtables, tables_doubles(W, N)));
    TRY(dalloc(s, &s->dx0, (size_t)batch * nx));
    TRY(dalloc(s, &s->dG, s->state_doubles())); TRY(dalloc(s, &s->dV, s->v_doubles())); TRY(dalloc(s, &s->dV2, s->v_doubles())); TRY(dalloc(s, &s->dD, s->d_doubles()));
    if (s->state_in_global) TRY(dalloc(s, &s->dscratch_state, (size_t)s->groups * state_scratch_doubles(nu, N, W)));
    TRY(dalloc(s, &s->dsolx, X * batch)); TRY(dalloc(s, &s->dsolu, U * batch));
    TRY(dalloc(s, &s->distats, (size_t)batch * 2)); TRY(dalloc(s, &s->ddstats, (size_t)batch * 4));

    // Problem data. Only the diagonals of Q and R are kept, each + rho (tiny_api.cpp:90-91).
    std::vector<double> qd(nx), rd(nu), fz(nx, 0.0);
    for (int i = 0; i < nx; ++i) qd[i] = Q[i + (size_t)i * nx] + rho;
    for (int i = 0; i < nu; ++i) rd[i] = R[i + (size_t)i * nu] + rho;
    TRY(upload(s, s->dA, A, (size_t)nx * nx)); TRY(upload(s, s->dB, B, (size_t)nx * nu));
    TRY(upload(s, s->dfdyn, fdyn ? fdyn : fz.data(), nx));
    TRY(upload(s, s->dQd, qd.data(), nx)); TRY(upload(s, s->dRd, rd.data(), nu));
    TRY(upload(s, s->dQfull, Q, (size_t)nx * nx)); TRY(upload(s, s->dRfull, R, (size_t)nu * nu));
    TRY(fill_host_upload(s, s->dxmin, X, -kBoundInf)); TRY(fill_host_upload(s, s->dxmax, X, kBoundInf));
    TRY(fill_host_upload(s, s->dumin, U, -kBoundInf)); TRY(fill_host_upload(s, s->dumax, U, kBoundInf));
    // Everything tiny_setup zeroes (tiny_api.cpp:41-44, 73-88, 100-111)
    HIP_TRY_S(hipMemsetAsync(s->ddK, 0, sizeof(double) * nu * nx, s->stream));
    HIP_TRY_S(hipMemsetAsync(s->ddP, 0, sizeof(double) * nx * nx, s->stream));
    TRY(fill_host_upload(s, s->drho_inst, batch, rho));
    HIP_TRY_S(hipMemsetAsync(s->dXref, 0, sizeof(double) * X, s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dUref, 0, sizeof(double) * U, s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dx0, 0, sizeof(double) * batch * nx, s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dG, 0, sizeof(double) * s->state_doubles(), s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dV, 0, sizeof(double) * s->v_doubles(), s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dV2, 0, sizeof(double) * s->v_doubles(), s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dD, 0, sizeof(double) * s->d_doubles(), s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dsolx, 0, sizeof(double) * X * batch, s->stream));
    HIP_TRY_S(hipMemsetAsync(s->dsolu, 0, sizeof(double) * U * batch, s->stream));
    HIP_TRY_S(hipMemsetAsync(s->distats, 0, sizeof(int) * batch * 2, s->stream));
    HIP_TRY_S(hipMemsetAsync(s->ddstats, 0, sizeof(double) * batch * 4, s->stream));

    if (batch == 1) {
        HIP_TRY_S(hipHostMalloc((void **)&s->h_sol, sizeof(double) * (X + U + 8), hipHostMallocCoherent));
        HIP_TRY_S(hipHostMalloc((void **)&s->h_x0, sizeof(double) * nx, hipHostMallocCoherent));
        HIP_TRY_S(hipHostMalloc((void **)&s->h_u0, sizeof(double) * nu, hipHostMallocCoherent));
        std::memset(s->h_sol, 0, sizeof(double) * (X + U + 8));
        HIP_TRY_S(hipHostMalloc((void **)&s->h_xref, sizeof(double) * X, hipHostMallocCoherent));
        HIP_TRY_S(hipHostMalloc((void **)&s->h_uref, sizeof(double) * (U ? U : 1), hipHostMallocCoherent));
        std::memset(s->h_xref, 0, sizeof(double) * X);  // tiny_setup zeroes the references (tiny_api.cpp:83-84)
        std::memset(s->h_uref, 0, sizeof(double) * (U ? U : 1));
    }
    TRY(run_precompute(s));  // tiny_api.cpp:113
    HIP_TRY_S(hipStreamSynchronize(s->stream));
#undef TRY
#undef HIP_TRY_S
    if (verbose) {
        int steps = 0;
        download(s, &steps, s->dinfo, sizeof(int));
        printf("TinyMPC-HIP setup: nx=%d nu=%d N=%d rho=%g batch=%d device=%d | lanes/instance=%d LDS=%zu B tables_in_lds=%d | Kinf converged after %d iterations\n",
               nx, nu, N, rho, batch, s->device, s->W, s->lds_bytes, (int)s->tables_in_lds, steps);
    }
    *out = s;
    return TINYMPC_OK;
}

int tinympc_setup(tinympc_solver **out, const double *A, const double *B, const double *fdyn,
                  const double *Q, const double *R, double rho, int nx, int nu, int N, int verbose) {
    return tinympc_setup_batch(out, A, B, fdyn, Q, R, rho, nx, nu, N, 1, -1, verbose);
}

int tinympc_set_x0(tinympc_solver *s, const double *x0, int len, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!x0) return fail(TINYMPC_ERR_INVALID_INPUT, "set_x0: x0 is NULL");
    // The reference only perror()s on a wrong length and still assigns (tiny_api.cpp:238-241); an
    // Eigen column assignment of the wrong length is undefined behaviour, so the C ABI rejects it.
    if (len != s->nx) return fail(TINYMPC_ERR_INVALID_INPUT, "set_x0: x0 has %d entries, expected %d", len, s->nx);
    if (s->host_path()) {  // no device call at all: the next launch picks x0 up from pinned host memory
        // (a launch of tinympc_solve_async may still be reading the buffer: wait for it first)
        if (s->host_sol_state == 1 && (rc = tinympc_synchronize(s))) return rc;
        std::memcpy(s->h_x0, x0, sizeof(double) * s->nx);
        s->x0_on_host = true;
        if (verbose) printf("Initial state set\n");
        return TINYMPC_OK;
    }
    if ((rc = bind_device(s))) return rc;
    rc = upload(s, s->dx0, x0, s->nx);
    if (!rc && verbose) printf("Initial state set\n");
    return rc;
}

int tinympc_set_x_ref(tinympc_solver *s, const double *Xref, int rows, int cols, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!Xref) return fail(TINYMPC_ERR_INVALID_INPUT, "set_x_ref: Xref is NULL");
    // The reference prints the mismatch and assigns anyway (tiny_api.cpp:250-254), which would change
    // the workspace shape under the solver; rejected here.
    if (rows != s->nx || cols != s->N)
        return fail(TINYMPC_ERR_INVALID_INPUT, "State reference trajectory (x_ref) is %d x %d. Expected %d x %d.", rows, cols, s->nx, s->N);
    s->xref_const = rows_constant(Xref, s->nx, s->N);
    if (s->host_path()) {  // no device call: the next launch reads the pinned copy and rebuilds the table rows itself
        if (s->host_sol_state == 1 && (rc = tinympc_synchronize(s))) return rc;  // a launch in flight may be reading it
        const size_t col = sizeof(double) * s->nx;
        if (std::memcmp(s->h_xref, Xref, col * s->N) == 0) {
            // the same reference again (tracking loops re-send it every tick): nothing to do
        } else if (s->session_active && !s->refs_on_host && !s->xref_shift && std::memcmp(s->h_xref + s->nx, Xref, col * (s->N - 1)) == 0) {
            s->xref_shift = true;  // moved up by one knot: only the new last column has to travel
            std::memcpy(s->h_xref, Xref, col * s->N);
        } else {
            std::memcpy(s->h_xref, Xref, col * s->N);
            s->refs_on_host = true;
        }
        if (verbose) printf("State reference set\n");
        return TINYMPC_OK;
    }
    if ((rc = bind_device(s))) return rc;
    rc = upload(s, s->dXref, Xref, s->X());
    s->tables_dirty = true;
    if (!rc && verbose) printf("State reference set\n");
    return rc;
}

int tinympc_set_u_ref(tinympc_solver *s, const double *Uref, int rows, int cols, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!Uref) return fail(TINYMPC_ERR_INVALID_INPUT, "set_u_ref: Uref is NULL");
    if (rows != s->nu || cols != s->N - 1)
        return fail(TINYMPC_ERR_INVALID_INPUT, "Control/input reference trajectory (u_ref) is %d x %d. Expected %d x %d.", rows, cols, s->nu, s->N - 1);
    s->uref_const = rows_constant(Uref, s->nu, s->N - 1);
    if (s->host_path()) {
        if (s->host_sol_state == 1 && (rc = tinympc_synchronize(s))) return rc;
        const size_t col = sizeof(double) * s->nu;
        if (std::memcmp(s->h_uref, Uref, col * (s->N - 1)) == 0) {
            // unchanged
        } else if (s->session_active && !s->refs_on_host && !s->uref_shift && s->N > 2 &&
                   std::memcmp(s->h_uref + s->nu, Uref, col * (s->N - 2)) == 0) {
            s->uref_shift = true;
            std::memcpy(s->h_uref, Uref, col * (s->N - 1));
        } else {
            std::memcpy(s->h_uref, Uref, col * (s->N - 1));
            s->refs_on_host = true;
        }
        if (verbose) printf("Input reference set\n");
        return TINYMPC_OK;
    }
    if ((rc = bind_device(s))) return rc;
    rc = upload(s, s->dUref, Uref, s->U());
    s->tables_dirty = true;
    if (!rc && verbose) printf("Input reference set\n");
    return rc;
}

int tinympc_set_bound_constraints(tinympc_solver *s, const double *x_min, const double *x_max,
                                  const double *u_min, const double *u_max, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!x_min || !x_max || !u_min || !u_max)
        return fail(TINYMPC_ERR_INVALID_INPUT, "set_bound_constraints requires x_min, x_max, u_min, u_max (expanded to nx x N / nu x (N-1))");
    if ((rc = bind_device(s))) return rc;
    if ((rc = upload(s, s->dxmin, x_min, s->X()))) return rc;
    if ((rc = upload(s, s->dxmax, x_max, s->X()))) return rc;
    if ((rc = upload(s, s->dumin, u_min, s->U()))) return rc;
    if ((rc = upload(s, s->dumax, u_max, s->U()))) return rc;
    s->xmin_const = rows_constant(x_min, s->nx, s->N); s->xmax_const = rows_constant(x_max, s->nx, s->N);
    s->umin_const = rows_constant(u_min, s->nu, s->N - 1); s->umax_const = rows_constant(u_max, s->nu, s->N - 1);
    s->st.en_state_bound = 1;  // bindings.cpp:206-207
    s->st.en_input_bound = 1;
    s->tables_dirty = true;
    if (verbose) printf("Bound constraints set\n");
    return TINYMPC_OK;
}

int tinympc_solve_async(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    return launch(s, false);
}

int tinympc_synchronize(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (s->session_active) return TINYMPC_OK;  // session steps are synchronous; the resident kernel never "finishes"
    if (s->flag_pending && s->host_sol_state == 1) {
        // The launch in flight raises a flag in pinned memory after its last store: poll it. Bounded: a kernel that takes
        // longer than the polling budget (long solves) is waited for the ordinary way.
        const volatile double *flag = s->h_sol + s->X() + s->U() + 6;
        const double want = (double)s->session_seq;
        for (int spin = 0; spin < 400000; ++spin) {
            if (*flag == want) {
                std::atomic_thread_fence(std::memory_order_acquire);
                s->flag_pending = false;
                s->host_sol_state = 2;
                return TINYMPC_OK;
            }
            __builtin_ia32_pause();
        }
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->flag_pending = false;
    if (s->host_sol_state == 1) s->host_sol_state = 2;
    return TINYMPC_OK;
}

int tinympc_solve(tinympc_solver *s, int verbose) {
    int rc = tinympc_solve_async(s);
    if (rc) return rc;
    if ((rc = tinympc_synchronize(s))) return rc;
    if (verbose) {
        int is[2] = {0, 0};
        if ((rc = download(s, is, s->distats, sizeof(is)))) return rc;
        if (is[1] == TINYMPC_STATUS_SOLVED) printf("Solver converged in %d iterations\n", is[0]);  // admm.cpp:190
        printf("Solve completed with status: %d\n", is[1] == TINYMPC_STATUS_SOLVED ? 0 : 1);
    }
    return TINYMPC_OK;
}

int tinympc_solve_timed(tinympc_solver *s, float *kernel_ms) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    if ((rc = refresh_derived(s))) return rc;  // keep table rebuilds out of the timed region
    if ((rc = launch(s, true))) return rc;
    HIP_TRY(hipEventSynchronize(s->ev1));
    if (s->host_sol_state == 1) s->host_sol_state = 2;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    if (kernel_ms) *kernel_ms = ms;
    return TINYMPC_OK;
}

int tinympc_mpc_step_batch(tinympc_solver *s, const double *x0s, double *u0_out) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!x0s || !u0_out) return fail(TINYMPC_ERR_INVALID_INPUT, "mpc_step: x0s and u0_out are required");
    if ((rc = bind_device(s))) return rc;
    const size_t nx0 = (size_t)s->batch * s->nx, nu0 = (size_t)s->batch * s->nu;
    s->x0_on_host = false;  // this call brings its own x0
    if (!s->h_x0) {  // pinned staging, so that the small copies are true async DMA and need no extra sync
        HIP_TRY(hipHostMalloc((void **)&s->h_x0, sizeof(double) * nx0, hipHostMallocCoherent));
        HIP_TRY(hipHostMalloc((void **)&s->h_u0, sizeof(double) * nu0, hipHostMallocCoherent));
    }
    if (s->host_sol_state == 1) {  // a launch of tinympc_solve_async may still be reading h_x0
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->host_sol_state = 2;
    }
    std::memcpy(s->h_x0, x0s, sizeof(double) * nx0);
    decide_layout_d_variants(s);
    if ((rc = decide_layout_e(s))) return rc;
    if ((rc = decide_layout_f(s))) return rc;
    if (s->batch <= kZeroCopyTickMax && s->st.max_iter > 0 && !s->use_layout_d() && !s->use_layout_e() && !s->layout_m) {
        // Small batches: no copy engine at all. The kernel reads x0 from the pinned host buffer (and mirrors it into
        // the device copy the other verbs use) and writes the first controls into the pinned host buffer; both
        // are device-visible host allocations, and the stream synchronisation makes the writes visible here.
        s->zero_copy_tick = true;
        rc = launch(s, false);
        s->zero_copy_tick = false;
        if (rc) return rc;
        if ((rc = tinympc_synchronize(s))) return rc;  // (single instance: polls the completion flag in pinned memory)
    } else {
        HIP_TRY(hipMemcpyAsync(s->dx0, s->h_x0, sizeof(double) * nx0, hipMemcpyHostToDevice, s->stream));
        if ((rc = launch(s, false))) return rc;
        HIP_TRY(hipMemcpy2DAsync(s->h_u0, sizeof(double) * s->nu, s->dsolu, sizeof(double) * s->U(), sizeof(double) * s->nu,
                                 s->batch, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    std::memcpy(u0_out, s->h_u0, sizeof(double) * nu0);
    return TINYMPC_OK;
}

int tinympc_session_begin(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;  // (ends a session that is still open)
    if (!s->host_path() || !(s->layout_c || (s->families_active() && s->fam_c)))
        return fail(TINYMPC_ERR_UNSUPPORTED, "session: single-instance handles on the latency kernel only (batch 1, nx+nu <= 16, N <= 129)");
    if (s->st.adaptive_rho) return fail(TINYMPC_ERR_UNSUPPORTED, "session: adaptive_rho is not supported");
    if (s->families_active() && s->chunk_len > 4)
        return fail(TINYMPC_ERR_UNSUPPORTED, "session: cone / linear families are supported for horizons up to N = 65 (got %d)", s->N);
    if (s->families_active() && family_structure(s).nround > 1)
        return fail(TINYMPC_ERR_UNSUPPORTED, "session: cones that share rows are not supported by the resident kernel");
    if (s->st.max_iter < 1) return fail(TINYMPC_ERR_INVALID_INPUT, "session: max_iter must be >= 1");
    if (!s->h_mail) {
        HIP_TRY(hipHostMalloc((void **)&s->h_mail, sizeof(double) * 64, hipHostMallocCoherent));
        std::memset(s->h_mail, 0, sizeof(double) * 64);
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    if ((rc = launch_session_kernel(s))) return rc;
    s->session_active = true;
    s->flag_pending = false;
    return TINYMPC_OK;
}

int tinympc_session_step(tinympc_solver *s, const double *x0, double *u0_out) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!x0 || !u0_out) return fail(TINYMPC_ERR_INVALID_INPUT, "session_step: x0 and u0_out are required");
    if (!s->session_active) return fail(TINYMPC_ERR_NOT_INITIALIZED, "session_step: no session is open (tinympc_session_begin)");
    HIP_TRY(hipSetDevice(s->device));  // (the restart path below launches; a multi-GPU caller may have another device current)
    int flags = 0;
    if (s->refs_on_host) flags = 2;  // full re-read (covers any pending shift: the pinned copies are current)
    else flags = (s->xref_shift ? 4 : 0) | (s->uref_shift ? 8 : 0);
    if (flags & 12) s->session_refs_shifted = true;
    s->refs_on_host = s->xref_shift = s->uref_shift = false;
    write_command(s, flags, x0);
    const volatile double *done = s->h_sol + s->X() + s->U() + 6;
    double want = (double)s->session_seq;
    const auto t_start = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        if (*done == want) break;
        __builtin_ia32_pause();
        if ((spin & 0xffff) == 0xffff) {
            // Nothing for a while: has the kernel left (idle time-out)? Then start it again; it waits for exactly the
            // command that is pending. A stream error or 30 s without an answer end the session with an error.
            const hipError_t q = hipStreamQuery(s->stream);
            if (q == hipSuccess) {
                // The new kernel stages the (current) pinned references in its prologue, so the command is issued again
                // under a NEW stamp and without reference flags -- the old one, still in the mailbox, must not be taken.
                rc = launch_session_kernel(s);  // waits for session_seq + 1
                if (rc) { s->session_active = false; return rc; }
                write_command(s, 0, x0);
                want = (double)s->session_seq;
            } else if (q != hipErrorNotReady) {
                s->session_active = false;
                return fail(TINYMPC_ERR_HIP, "session_step: the handle's stream reports %s", hipGetErrorString(q));
            }
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(30)) {
                (void)end_session(s);
                return fail(TINYMPC_ERR_HIP, "session_step: no answer from the resident kernel within 30 s");
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    s->host_sol_state = 2;
    std::memcpy(u0_out, s->h_sol + s->X(), sizeof(double) * s->nu);
    return TINYMPC_OK;
}

int tinympc_session_end(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    return end_session(s);
}

int tinympc_get_solution_batch(tinympc_solver *s, double *x_out, double *u_out, int first, int count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (first < 0 || count < 0 || first + count > s->batch)
        return fail(TINYMPC_ERR_INVALID_INPUT, "instance range [%d, %d) outside batch of %d", first, first + count, s->batch);
    if (s->host_path() && count == 1 && s->host_sol_state != 0) {
        if (s->host_sol_state == 1 && (rc = tinympc_synchronize(s))) return rc;
        if (x_out) std::memcpy(x_out, s->h_sol, sizeof(double) * s->X());
        if (u_out) std::memcpy(u_out, s->h_sol + s->X(), sizeof(double) * s->U());
        return TINYMPC_OK;
    }
    if ((rc = bind_device(s))) return rc;
    if (x_out && (rc = download(s, x_out, s->dsolx + (size_t)first * s->X(), sizeof(double) * s->X() * count))) return rc;
    if (u_out && (rc = download(s, u_out, s->dsolu + (size_t)first * s->U(), sizeof(double) * s->U() * count))) return rc;
    return TINYMPC_OK;
}

int tinympc_get_solution(tinympc_solver *s, double *x_out, double *u_out, int verbose) {
    int rc = tinympc_get_solution_batch(s, x_out, u_out, 0, 1);
    if (!rc && verbose) printf("Solution retrieved\n");
    return rc;
}

int tinympc_get_first_controls_batch(tinympc_solver *s, double *u0_out, int first, int count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!u0_out) return fail(TINYMPC_ERR_INVALID_INPUT, "u0_out is NULL");
    if (first < 0 || count < 0 || first + count > s->batch)
        return fail(TINYMPC_ERR_INVALID_INPUT, "instance range [%d, %d) outside batch of %d", first, first + count, s->batch);
    if ((rc = bind_device(s))) return rc;
    HIP_TRY(hipMemcpy2DAsync(u0_out, sizeof(double) * s->nu, s->dsolu + (size_t)first * s->U(), sizeof(double) * s->U(),
                             sizeof(double) * s->nu, count, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TINYMPC_OK;
}

int tinympc_get_stats_batch(tinympc_solver *s, int *iters, int *status, double *residuals, int first, int count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (first < 0 || count < 0 || first + count > s->batch)
        return fail(TINYMPC_ERR_INVALID_INPUT, "instance range [%d, %d) outside batch of %d", first, first + count, s->batch);
    if (s->host_path() && count == 1 && s->host_sol_state != 0) {
        if (s->host_sol_state == 1 && (rc = tinympc_synchronize(s))) return rc;
        const double *hs = s->h_sol + s->X() + s->U();
        if (iters) iters[0] = (int)hs[4];
        if (status) status[0] = (int)hs[5];
        if (residuals) std::memcpy(residuals, hs, sizeof(double) * 4);
        return TINYMPC_OK;
    }
    if ((rc = bind_device(s))) return rc;
    if (iters || status) {
        std::vector<int> is((size_t)count * 2);
        if ((rc = download(s, is.data(), s->distats + (size_t)first * 2, sizeof(int) * 2 * count))) return rc;
        for (int b = 0; b < count; ++b) {
            if (iters) iters[b] = is[2 * (size_t)b];
            if (status) status[b] = is[2 * (size_t)b + 1];
        }
    }
    if (residuals && (rc = download(s, residuals, s->ddstats + (size_t)first * 4, sizeof(double) * 4 * count))) return rc;
    return TINYMPC_OK;
}

int tinympc_get_stats(tinympc_solver *s, int *iter, int *status, double *pri_res_state, double *pri_res_input, int verbose) {
    double res[4];
    int it = 0, st = 0;
    int rc = tinympc_get_stats_batch(s, &it, &st, res, 0, 1);
    if (rc) return rc;
    if (iter) *iter = it;
    if (status) *status = st;
    if (pri_res_state) *pri_res_state = res[0];
    if (pri_res_input) *pri_res_input = res[2];
    if (verbose) printf("Statistics retrieved: iter=%d, status=%d\n", it, st);
    return TINYMPC_OK;
}

int tinympc_get_residuals(tinympc_solver *s, double residuals[4]) {
    return tinympc_get_stats_batch(s, nullptr, nullptr, residuals, 0, 1);
}

namespace {
// Gather what the emitter needs from the device (the cache the kernels computed or the caller installed, the
// rho-augmented cost diagonals, dynamics, bounds) and write the embedded project's data files.
int codegen_from_handle(tinympc_solver *s, const char *output_dir, const double *dK, const double *dP, const double *dC1,
                        const double *dC2, int verbose) {
    int rc = bind_device(s);
    if (rc) return rc;
    const size_t nx = s->nx, nu = s->nu, X = s->X(), U = s->U();
    std::vector<double> K(nu * nx), P(nx * nx), Qi(nu * nu), Am(nx * nx), qd(nx), rd(nu), A(nx * nx), B(nx * nu), xmin(X), xmax(X), umin(U), umax(U);
    const struct { void *dst; const void *src; size_t n; } pulls[] = {
        {K.data(), s->dKinf, K.size()}, {P.data(), s->dPinf, P.size()}, {Qi.data(), s->dQuu, Qi.size()}, {Am.data(), s->dAmBKt, Am.size()},
        {qd.data(), s->dQd, nx}, {rd.data(), s->dRd, nu}, {A.data(), s->dA, A.size()}, {B.data(), s->dB, B.size()},
        {xmin.data(), s->dxmin, X}, {xmax.data(), s->dxmax, X}, {umin.data(), s->dumin, U}, {umax.data(), s->dumax, U}};
    for (const auto &p : pulls)
        if ((rc = download(s, p.dst, p.src, sizeof(double) * p.n))) return rc;
    int is[2] = {0, 0};  // iter, status of instance 0
    if ((rc = download(s, is, s->distats, sizeof(is)))) return rc;
    tinympc_codegen_data d{};
    d.nx = s->nx; d.nu = s->nu; d.N = s->N; d.rho = s->rho;
    d.iter = is[0]; d.solved = is[1] == TINYMPC_STATUS_SOLVED ? 1 : 0;
    d.Kinf = K.data(); d.Pinf = P.data(); d.Quu_inv = Qi.data(); d.AmBKt = Am.data();
    d.dKinf_drho = dK; d.dPinf_drho = dP; d.dC1_drho = dC1; d.dC2_drho = dC2;
    d.abs_pri_tol = s->st.abs_pri_tol; d.abs_dua_tol = s->st.abs_dua_tol; d.max_iter = s->st.max_iter;
    d.check_termination = s->st.check_termination; d.en_state_bound = s->st.en_state_bound; d.en_input_bound = s->st.en_input_bound;
    d.adaptive_rho = s->st.adaptive_rho;
    d.Q = qd.data(); d.R = rd.data(); d.Adyn = A.data(); d.Bdyn = B.data();
    d.x_min = xmin.data(); d.x_max = xmax.data(); d.u_min = umin.data(); d.u_max = umax.data();
    return tinympc_codegen_emit(&d, output_dir, verbose);
}
}  // namespace

int tinympc_codegen(tinympc_solver *s, const char *output_dir, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    return codegen_from_handle(s, output_dir, nullptr, nullptr, nullptr, nullptr, verbose);
}

int tinympc_codegen_with_sensitivity(tinympc_solver *s, const char *output_dir, const double *dK, const double *dP, const double *dC1,
                                     const double *dC2, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!dK || !dP || !dC1 || !dC2) return fail(TINYMPC_ERR_INVALID_INPUT, "codegen_with_sensitivity requires dK, dP, dC1, dC2");
    // the four matrices reach the cache (and the generated file) only while adaptive_rho is enabled (codegen.cpp:79-86, 237-252)
    if (s->st.adaptive_rho) {
        if ((rc = bind_device(s))) return rc;
        if ((rc = upload(s, s->ddK, dK, (size_t)s->nu * s->nx))) return rc;
        if ((rc = upload(s, s->ddP, dP, (size_t)s->nx * s->nx))) return rc;
    }
    return codegen_from_handle(s, output_dir, dK, dP, dC1, dC2, verbose);
}

int tinympc_set_sensitivity_matrices(tinympc_solver *s, const double *dK, const double *dP, const double *dC1, const double *dC2, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!dK || !dP || !dC1 || !dC2) return fail(TINYMPC_ERR_INVALID_INPUT, "set_sensitivity_matrices requires 4 matrices");
    // bindings.cpp:338-352 only prints their norms; the old core's adaptive-rho update reads them from the cache
    // (rho_benchmark.cpp:205-208), which is where they go here: dKinf/drho and dPinf/drho feed k_admm_solve_adapt
    // (dC1/dC2 would update C1/C2, which no solve phase reads).
    if ((rc = bind_device(s))) return rc;
    if ((rc = upload(s, s->ddK, dK, (size_t)s->nu * s->nx))) return rc;
    if ((rc = upload(s, s->ddP, dP, (size_t)s->nx * s->nx))) return rc;
    if (verbose) printf("Sensitivity matrices set\n");
    return TINYMPC_OK;
}

int tinympc_set_cache_terms(tinympc_solver *s, const double *Kinf, const double *Pinf, const double *Quu_inv, const double *AmBKt, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!Kinf || !Pinf || !Quu_inv || !AmBKt) return fail(TINYMPC_ERR_INVALID_INPUT, "set_cache_terms requires Kinf, Pinf, Quu_inv, AmBKt");
    if ((rc = bind_device(s))) return rc;
    if ((rc = upload(s, s->dKinf, Kinf, (size_t)s->nu * s->nx))) return rc;
    if ((rc = upload(s, s->dPinf, Pinf, (size_t)s->nx * s->nx))) return rc;
    if ((rc = upload(s, s->dQuu, Quu_inv, (size_t)s->nu * s->nu))) return rc;
    if ((rc = upload(s, s->dAmBKt, AmBKt, (size_t)s->nx * s->nx))) return rc;
    s->ops_dirty = true;
    s->tables_dirty = true;
    if (verbose) printf("Cache terms set\n");
    return TINYMPC_OK;
}

int tinympc_set_linear_constraints(tinympc_solver *s, const double *Alin_x, const double *blin_x, int nlx,
                                   const double *Alin_u, const double *blin_u, int nlu) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (nlx < 0 || nlu < 0) return fail(TINYMPC_ERR_INVALID_INPUT, "negative constraint count");
    if ((nlx > 0 && (!Alin_x || !blin_x)) || (nlu > 0 && (!Alin_u || !blin_u)))
        return fail(TINYMPC_ERR_INVALID_INPUT, "set_linear_constraints: NULL matrix for a non-empty side");
    if (s->session_active && (rc = bind_device(s))) return rc;  // the resident kernel was started with the old families
    if (nlx > MAX_LIN_ROWS || nlu > MAX_LIN_ROWS)
        return fail(TINYMPC_ERR_UNSUPPORTED, "at most %d linear rows per side are supported by the HIP kernel (got %d state, %d input)",
                    MAX_LIN_ROWS, nlx, nlu);
    auto zero_row = [](const double *A, int rows, int cols, int k) {
        for (int c = 0; c < cols; ++c)
            if (A[k + (size_t)c * rows] != 0.0) return false;
        return true;
    };
    for (int k = 0; k < nlx; ++k)
        if (zero_row(Alin_x, nlx, s->nx, k)) return fail(TINYMPC_ERR_INVALID_INPUT, "Alin_x row %d is all zero", k);
    for (int k = 0; k < nlu; ++k)
        if (zero_row(Alin_u, nlu, s->nu, k)) return fail(TINYMPC_ERR_INVALID_INPUT, "Alin_u row %d is all zero", k);
    s->n_lin_x = nlx;
    s->n_lin_u = nlu;
    s->Alin_x.assign(Alin_x, Alin_x + (size_t)nlx * s->nx);
    s->blin_x.assign(blin_x, blin_x + nlx);
    s->Alin_u.assign(Alin_u, Alin_u + (size_t)nlu * s->nu);
    s->blin_u.assign(blin_u, blin_u + nlu);
    if (nlx > 0) s->st.en_state_linear = 1;  // bindings.cpp:422-429
    if (nlu > 0) s->st.en_input_linear = 1;
    s->fam_dirty = true;
    return TINYMPC_OK;
}

int tinympc_set_cone_constraints(tinympc_solver *s, const int *Acx, const int *qcx, const double *cx, int ncx,
                                 const int *Acu, const int *qcu, const double *cu, int ncu) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (ncx < 0 || ncu < 0) return fail(TINYMPC_ERR_INVALID_INPUT, "negative cone count");
    if ((ncx > 0 && (!Acx || !qcx || !cx)) || (ncu > 0 && (!Acu || !qcu || !cu)))
        return fail(TINYMPC_ERR_INVALID_INPUT, "set_cone_constraints: NULL array for a non-empty side");
    if (s->session_active && (rc = bind_device(s))) return rc;  // the resident kernel was started with the old families
    // Each cone must lie inside its vector. Cones of one side MAY share rows: upstream projects the cones of a knot one after
    // another (bindings.cpp:433-478 hands the list over in order), which the kernels reproduce by grouping the list into rounds of
    // pairwise-disjoint cones (family_structure); at most MAX_CONES cones in MAX_ROUNDS rounds.
    auto check_side = [&](const int *Ac, const int *qc, const double *c, int n, int dim, const char *side) -> int {
        for (int k = 0; k < n; ++k) {
            if (qc[k] < 1 || Ac[k] < 0 || Ac[k] + qc[k] > dim)
                return fail(TINYMPC_ERR_INVALID_INPUT, "%s cone %d (start %d, dimension %d) does not fit a vector of %d rows", side, k, Ac[k], qc[k], dim);
            if (!(c[k] > 0.0)) return fail(TINYMPC_ERR_INVALID_INPUT, "%s cone %d has non-positive slope %g", side, k, c[k]);
        }
        return TINYMPC_OK;
    };
    if ((rc = check_side(Acx, qcx, cx, ncx, s->nx, "state"))) return rc;
    if ((rc = check_side(Acu, qcu, cu, ncu, s->nu, "input"))) return rc;
    if (ncx + ncu > MAX_CONES)
        return fail(TINYMPC_ERR_UNSUPPORTED, "at most %d cones are supported by the HIP kernels (got %d state + %d input)", MAX_CONES, ncx, ncu);
    {   // rounds of the whole list, both sides enabled (the worst case of what a launch can see)
        int rounds = 0;
        unsigned long long used = 0;
        auto walk = [&](const int *Ac, const int *qc, int n, int base) {
            for (int k = 0; k < n; ++k) {
                unsigned long long lanes = 0;
                for (int r = base + Ac[k]; r < base + Ac[k] + qc[k]; ++r) lanes |= 1ull << (r & 63);
                if (rounds == 0) rounds = 1;
                if (lanes & used) {
                    rounds += 1;
                    used = 0;
                }
                used |= lanes;
            }
        };
        if (s->nx + s->nu <= 64) {  // (larger systems have no families kernel at all: refused at launch)
            walk(Acx, qcx, ncx, 0);
            walk(Acu, qcu, ncu, s->nx);
        }
        if (rounds > MAX_ROUNDS)
            return fail(TINYMPC_ERR_UNSUPPORTED, "the cone list needs %d rounds of pairwise-disjoint cones; the HIP kernels support %d", rounds, MAX_ROUNDS);
    }
    s->n_cone_x = ncx;
    s->n_cone_u = ncu;
    s->Acx.assign(Acx, Acx + ncx); s->qcx.assign(qcx, qcx + ncx); s->cx.assign(cx, cx + ncx);
    s->Acu.assign(Acu, Acu + ncu); s->qcu.assign(qcu, qcu + ncu); s->cu.assign(cu, cu + ncu);
    if (ncx > 0) s->st.en_state_soc = 1;  // bindings.cpp:468-476
    if (ncu > 0) s->st.en_input_soc = 1;
    s->fam_dirty = true;
    return TINYMPC_OK;
}

int tinympc_reset(tinympc_solver **s, int verbose) {
    if (!s || !*s) return TINYMPC_OK;  // bindings.cpp:539: silently nothing to do
    destroy(*s);
    *s = nullptr;
    if (verbose) printf("Solver reset\n");
    return TINYMPC_OK;
}

int tinympc_update_settings(tinympc_solver *s, double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                            int en_state_bound, int en_input_bound, int en_state_soc, int en_input_soc,
                            int en_state_linear, int en_input_linear, int adaptive_rho, double adaptive_rho_min,
                            double adaptive_rho_max, int adaptive_rho_enable_clipping, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (max_iter < 0) return fail(TINYMPC_ERR_INVALID_INPUT, "max_iter must be >= 0");
    // A resident session kernel carries the settings it was launched with (max_iter, tolerances, family flags): end it, so
    // that nothing switches behaviour in the middle of a session or bypasses the checks of tinympc_session_begin on a
    // restart after the idle time-out. The next tinympc_session_begin starts one with the new settings.
    if (s->session_active && (rc = bind_device(s))) return rc;
    const bool bounds_changed = (s->st.en_state_bound != en_state_bound) || (s->st.en_input_bound != en_input_bound);
    s->st.abs_pri_tol = abs_pri_tol; s->st.abs_dua_tol = abs_dua_tol;
    s->st.max_iter = max_iter; s->st.check_termination = check_termination;
    s->st.en_state_bound = en_state_bound; s->st.en_input_bound = en_input_bound;
    s->st.en_state_soc = en_state_soc; s->st.en_input_soc = en_input_soc;
    s->st.en_state_linear = en_state_linear; s->st.en_input_linear = en_input_linear;
    s->st.adaptive_rho = adaptive_rho; s->st.adaptive_rho_min = adaptive_rho_min;
    s->st.adaptive_rho_max = adaptive_rho_max; s->st.adaptive_rho_enable_clipping = adaptive_rho_enable_clipping;
    if (bounds_changed) s->tables_dirty = true;
    s->fam_dirty = true;  // the family enable flags are folded into the per-lane family description
    if (verbose) printf("Settings updated successfully\n");
    return TINYMPC_OK;
}

int tinympc_print_problem_data(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    int is[2] = {0, 0};
    if ((rc = bind_device(s))) return rc;
    if ((rc = download(s, is, s->distats, sizeof(is)))) return rc;
    printf("solution iter: %d\nsolution solved: %d\n", is[0], is[1] == TINYMPC_STATUS_SOLVED);
    printf("\n\ncache rho: %f\n", s->rho);
    printf("\n\nabs_pri_tol: %f\nabs_dua_tol: %f\nmax_iter: %d\ncheck_termination: %d\nen_state_bound: %d\nen_input_bound: %d\n",
           s->st.abs_pri_tol, s->st.abs_dua_tol, s->st.max_iter, s->st.check_termination, s->st.en_state_bound, s->st.en_input_bound);
    printf("\n\nnx: %d\nnu: %d\niter: %d\nstatus: %d\n", s->nx, s->nu, is[0], is[1]);
    return TINYMPC_OK;
}

int tinympc_get_cache(tinympc_solver *s, double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt, int *riccati_iters) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    if (Kinf && (rc = download(s, Kinf, s->dKinf, sizeof(double) * s->nu * s->nx))) return rc;
    if (Pinf && (rc = download(s, Pinf, s->dPinf, sizeof(double) * s->nx * s->nx))) return rc;
    if (Quu_inv && (rc = download(s, Quu_inv, s->dQuu, sizeof(double) * s->nu * s->nu))) return rc;
    if (AmBKt && (rc = download(s, AmBKt, s->dAmBKt, sizeof(double) * s->nx * s->nx))) return rc;
    if (riccati_iters && (rc = download(s, riccati_iters, s->dinfo, sizeof(int)))) return rc;
    return TINYMPC_OK;
}

namespace {
// One Riccati recursion of the .m class on the device; results land in slot `slot` of dlqr_out.
int run_lqr(tinympc_solver *s, int slot, double rho, double reg, double tol, int norm_kind, int max_iter, int min_iter, int p0_augmented) {
    LqrParams p{};
    p.nx = s->nx; p.nu = s->nu; p.rho = rho; p.reg = reg; p.tol = tol; p.norm_kind = norm_kind;
    p.max_iter = max_iter; p.min_iter = min_iter; p.p0_augmented = p0_augmented;
    p.A = s->dA; p.B = s->dB; p.Q = s->dQfull; p.R = s->dRfull;
    double *o = s->dlqr_out + (size_t)slot * s->cache_doubles();
    p.K = o; p.P = p.K + (size_t)s->nu * s->nx; p.C1 = p.P + (size_t)s->nx * s->nx; p.C2 = p.C1 + (size_t)s->nu * s->nu;
    p.info = s->dinfo + 1 + slot; p.scratch = s->dlqr_scratch;
    p.use_lds = lqr_scratch_doubles(s->nx, s->nu) <= 6500 ? 1 : 0;
    HIP_TRY(launch_lqr(p, s->stream));
    return TINYMPC_OK;
}

int download_cache_slot(tinympc_solver *s, int slot, double *K, double *P, double *C1, double *C2) {
    const double *o = s->dlqr_out + (size_t)slot * s->cache_doubles();
    const size_t nK = (size_t)s->nu * s->nx, nP = (size_t)s->nx * s->nx, nC1 = (size_t)s->nu * s->nu;
    int rc;
    if (K && (rc = download(s, K, o, sizeof(double) * nK))) return rc;
    if (P && (rc = download(s, P, o + nK, sizeof(double) * nP))) return rc;
    if (C1 && (rc = download(s, C1, o + nK + nP, sizeof(double) * nC1))) return rc;
    if (C2 && (rc = download(s, C2, o + nK + nP + nC1, sizeof(double) * nP))) return rc;
    return TINYMPC_OK;
}
}  // namespace

// TinyMPC.m:194-221: P0 = Q, up to 5000 steps, 1e-8 regulariser in the gain solve, stop at norm(K-Kprev) < 1e-10.
int tinympc_compute_cache_terms(tinympc_solver *s, double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt, int *iters, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    if ((rc = run_lqr(s, 0, s->rho, 1e-8, 1e-10, 2, 5000, 1, 0))) return rc;
    if ((rc = download_cache_slot(s, 0, Kinf, Pinf, Quu_inv, AmBKt))) return rc;
    int it = 0;
    if ((rc = download(s, &it, s->dinfo + 1, sizeof(int)))) return rc;
    if (iters) *iters = it;
    if (verbose) printf("Cache terms computed on the device after %d Riccati steps\n", it);
    return TINYMPC_OK;
}

// TinyMPC.m:336-366: the stabilising solution of the DARE for Q + rho I, R + rho I (MATLAB: idare; here the same
// fixed-point recursion as the class's fallback branch, without regulariser, run until K stops changing).
int tinympc_solve_lqr(tinympc_solver *s, double rho_val, double *K, double *P, double *C1, double *C2, int *iters) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    if ((rc = run_lqr(s, 0, rho_val, 0.0, 1e-15, 0, 200000, 2, 1))) return rc;
    if ((rc = download_cache_slot(s, 0, K, P, C1, C2))) return rc;
    if (iters && (rc = download(s, iters, s->dinfo + 1, sizeof(int)))) return rc;
    return TINYMPC_OK;
}

// TinyMPC.m:223-241: forward differences of solve_lqr in rho with h = 1e-6.
int tinympc_compute_sensitivity(tinympc_solver *s, double *dK, double *dP, double *dC1, double *dC2, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    const double h = 1e-6;
    if ((rc = run_lqr(s, 0, s->rho, 0.0, 1e-15, 0, 200000, 2, 1))) return rc;
    if ((rc = run_lqr(s, 1, s->rho + h, 0.0, 1e-15, 0, 200000, 2, 1))) return rc;
    FiniteDiffParams f{};
    f.count = (int)s->cache_doubles(); f.h = h;
    f.lo = s->dlqr_out; f.hi = s->dlqr_out + s->cache_doubles(); f.out = s->dlqr_out + 2 * s->cache_doubles();
    HIP_TRY(launch_finite_diff(f, s->stream));
    if ((rc = download_cache_slot(s, 2, dK, dP, dC1, dC2))) return rc;
    if (verbose) printf("Sensitivity matrices computed on the device (forward differences, h = %g)\n", h);
    return TINYMPC_OK;
}

int tinympc_set_x0_batch(tinympc_solver *s, const double *x0s, int first, int count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!x0s) return fail(TINYMPC_ERR_INVALID_INPUT, "x0s is NULL");
    if (first < 0 || count < 0 || first + count > s->batch)
        return fail(TINYMPC_ERR_INVALID_INPUT, "instance range [%d, %d) outside batch of %d", first, first + count, s->batch);
    if ((rc = bind_device(s))) return rc;
    s->x0_on_host = false;
    return upload(s, s->dx0 + (size_t)first * s->nx, x0s, (size_t)count * s->nx);
}

int tinympc_set_x0_batch_device(tinympc_solver *s, const double *d_x0s, int first, int count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!d_x0s) return fail(TINYMPC_ERR_INVALID_INPUT, "d_x0s is NULL");
    if (first < 0 || count < 0 || first + count > s->batch)
        return fail(TINYMPC_ERR_INVALID_INPUT, "instance range [%d, %d) outside batch of %d", first, first + count, s->batch);
    if ((rc = bind_device(s))) return rc;
    s->x0_on_host = false;
    HIP_TRY(hipMemcpyAsync(s->dx0 + (size_t)first * s->nx, d_x0s, sizeof(double) * count * s->nx, hipMemcpyDeviceToDevice, s->stream));
    // The copy runs on the handle's own (non-blocking) stream: wait for it here, so that the caller may free or reuse
    // d_x0s as soon as the call returns -- the same ownership rule as every host-pointer verb. (Work that PRODUCES
    // d_x0s on another stream must have completed before the call; see the header.)
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TINYMPC_OK;
}

int tinympc_reset_workspace(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    HIP_TRY(hipMemsetAsync(s->dG, 0, sizeof(double) * s->state_doubles(), s->stream));
    HIP_TRY(hipMemsetAsync(s->dV, 0, sizeof(double) * s->v_doubles(), s->stream));
    // (V2, the stale-copy buffer of layout B, and LX, the families' forward -> backward term, are always written
    //  before they are read within a solve: nothing to reset)
    HIP_TRY(hipMemsetAsync(s->dD, 0, sizeof(double) * s->d_doubles(), s->stream));
    if (s->dGC) {
        HIP_TRY(hipMemsetAsync(s->dGC, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dGL, 0, sizeof(double) * s->v_doubles(), s->stream));
    }
    HIP_TRY(hipMemsetAsync(s->dsolx, 0, sizeof(double) * s->X() * s->batch, s->stream));
    HIP_TRY(hipMemsetAsync(s->dsolu, 0, sizeof(double) * s->U() * s->batch, s->stream));
    HIP_TRY(hipMemsetAsync(s->distats, 0, sizeof(int) * s->batch * 2, s->stream));
    HIP_TRY(hipMemsetAsync(s->ddstats, 0, sizeof(double) * s->batch * 4, s->stream));
    HIP_TRY(launch_fill(s->drho_inst, (size_t)s->batch, s->rho, s->stream));  // adapted rho back to the setup value
    s->host_sol_state = 0;  // the device solution / statistics were just zeroed: read them from there
    return TINYMPC_OK;
}

int tinympc_get_rho_batch(tinympc_solver *s, double *rho_out, int first, int count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!rho_out || first < 0 || count < 0 || first + count > s->batch)
        return fail(TINYMPC_ERR_INVALID_INPUT, "get_rho_batch: range [%d, %d) outside the batch of %d", first, first + count, s->batch);
    if ((rc = bind_device(s))) return rc;
    return download(s, rho_out, s->drho_inst + first, sizeof(double) * count);
}

int tinympc_get_solution_device_ptrs(tinympc_solver *s, const double **d_x, const double **d_u) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (d_x) *d_x = s->dsolx;
    if (d_u) *d_u = s->dsolu;
    return TINYMPC_OK;
}

int tinympc_get_launch_info(tinympc_solver *s, int *lanes_per_instance, int *instances_per_wave, int *workgroups,
                            int *lds_bytes, int *tables_in_lds) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (lanes_per_instance) *lanes_per_instance = s->W;
    if (instances_per_wave) *instances_per_wave = s->IPW;
    if (workgroups && s->use_layout_f()) *workgroups = s->batch;
    else if (workgroups && s->use_layout_e()) *workgroups = s->groups;
    else if (workgroups) *workgroups = s->use_layout_d() ? ((s->d_jit || s->families_active() || s->st.adaptive_rho || (!s->tables_const() && s->d_varying_jit)) ? solve_jit_workgroups(s->W, s->nx, s->nu, s->N, s->tables_const(), s->groups, s->families_active(), s->st.adaptive_rho != 0) : s->W == 64 ? solve_dx_workgroups(s->nu, s->N, s->groups) : s->W == 32 ? solve_dw_workgroups(s->nu, s->N, s->groups) : solve_d_workgroups(s->nu, s->N, s->tables_const(), s->groups)) : s->layout_c ? s->batch : s->layout_b ? (s->groups + WAVES_PER_GROUP_B - 1) / WAVES_PER_GROUP_B : s->groups;
    if (lds_bytes) {
        size_t l = s->layout_c ? s->lds_bytes_c : s->lds_bytes;
        if (s->layout_m) l = 0;  // (static LDS: see the kernel)
        else if (s->use_layout_f()) l = s->f_lds;
        else if (s->use_layout_e()) l = s->e_lds;
        else if (s->use_layout_d())
            l = (s->d_jit || s->families_active() || s->st.adaptive_rho || (!s->tables_const() && s->d_varying_jit))
                    ? solve_jit_lds_bytes(s->W, s->nx, s->nu, s->N, s->tables_const(), s->families_active(), s->st.adaptive_rho != 0)
                    : s->W == 64 ? solve_dx_lds_bytes(s->nu, s->N) : s->W == 32 ? solve_dw_lds_bytes(s->nu, s->N) : solve_d_lds_bytes(s->nu, s->N, s->tables_const());
        *lds_bytes = (int)l;
    }
    if (tables_in_lds) *tables_in_lds = (s->tables_in_lds && !s->layout_c) ? 1 : 0;  // layout C keeps its table entries in registers
    return TINYMPC_OK;
}

int tinympc_get_layout(tinympc_solver *s) {
    if (!s) return 0;
    if (s->layout_m) return 'M';
    if (s->use_layout_d()) return 'D';
    if (s->use_layout_e()) return 'E';
    if (s->use_layout_f()) return 'F';
    // the families and adaptive rho have kernels of their own on layout A's plan (the families also in the latency kernel)
    if (s->families_active()) return (s->fam_c && family_structure(s).nround <= 1) ? 'C' : 'A';
    if (s->st.adaptive_rho) return 'A';
    return s->layout_c ? 'C' : s->layout_b ? 'B' : 'A';
}

int tinympc_prepare(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    decide_layout_d_variants(s);
    if ((rc = decide_layout_e(s))) return rc;
    return decide_layout_f(s);
}

int tinympc_get_jit_info(tinympc_solver *s, char *buf, int len) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!buf || len < 1) return fail(TINYMPC_ERR_INVALID_INPUT, "get_jit_info: buffer required");
    buf[0] = '\0';
    const bool fam = s->families_active(), adaptive = s->st.adaptive_rho != 0;
    if (s->layout_m) snprintf(buf, (size_t)len, "compiled-in layout=M");
    else if (!s->f_sig.empty() && !adaptive && !s->use_layout_d() && !s->use_layout_e()) solve_f_describe(s->nx, s->nu, s->N, s->tables_const(), fam, s->f_fs, buf, (size_t)len);
    else if (!s->e_sig.empty() && !adaptive && !s->use_layout_d()) solve_e_describe(s->nx, s->nu, s->N, s->tables_const(), fam, s->fs, buf, (size_t)len);
    else if (s->use_layout_d() && !(s->d_jit || fam || adaptive || (!s->tables_const() && s->d_varying_jit))) snprintf(buf, (size_t)len, "compiled-in layout=D");
    else if (s->layout_d || s->d_jit || s->d_jit_asked) {
        if ((rc = bind_device(s))) return rc;
        solve_jit_describe(s->W, s->nx, s->nu, s->N, s->tables_const(), fam && !adaptive, adaptive && !fam, buf, (size_t)len);
    } else snprintf(buf, (size_t)len, "compiled-in layout=%c", (char)tinympc_get_layout(s));
    return TINYMPC_OK;
}

#ifdef TINY_CLOCK_STAMP
// Diagnostic build only (not part of the ABI): the stamp records of the last layout-D launch, 8 values per wavefront (see the
// kernel); returns the number of wavefronts copied into records[8 * capacity].
int tinympc_debug_clock_stamps(tinympc_solver *s, unsigned long long *records, int capacity) {
    if (!s || !s->dclock) return 0;
    if (hipSetDevice(s->device) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) return 0;
    const int n = s->groups < capacity ? s->groups : capacity;
    if (hipMemcpy(records, s->dclock, sizeof(unsigned long long) * 8 * n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}
#endif

void *tinympc_get_stream(tinympc_solver *s) { return s ? (void *)s->stream : nullptr; }

}  // extern "C"
