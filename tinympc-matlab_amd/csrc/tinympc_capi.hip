// tinympc_capi.hip -- the verbs of the extern "C" boundary declared in include/tinympc_hip.h: argument validation with the
// reference's error behaviour, data in and out. The handle and its helpers: tinympc_handle.h / .hip; which kernel a launch runs:
// tinympc_plan.hip; the resident session: tinympc_session.hip. Host-side bookkeeping only -- every number the solver produces is
// computed by the kernels; there is no CPU fallback anywhere.
#include "tinympc_handle.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <limits>
#include <new>

using namespace tinympc;
using namespace tinympc::host;

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
        W = KT = solve_m_geometry(nx, nu);  // geometry of the operators / tables (128, 256 or 512); the kernel works on 16-instance tiles
    } else if (!choose_geometry(nx, nu, &W, &KT)) {
        return fail(TINYMPC_ERR_UNSUPPORTED, "nx+nu = %d: systems beyond 512 rows are not supported by this build", nx + nu);
    }
    int ndev = tinympc_device_count();
    if (ndev < 1) return fail(TINYMPC_ERR_NO_DEVICE, "no HIP device visible: the HIP path has no CPU fallback");
    if (device >= ndev) return fail(TINYMPC_ERR_INVALID_INPUT, "device %d out of range (have %d)", device, ndev);

    tinympc_solver *s = new (std::nothrow) tinympc_solver();
    if (!s) return fail(TINYMPC_ERR_ALLOC, "out of host memory");
    const auto t_setup = std::chrono::steady_clock::now();
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
    park_sessions_on_device(s->device, s);  // (the allocations below synchronise the device)
    TRY(acquire_stream_kit(s));

    // LDS plan: ADMM state always; the per-knot tables too when the total fits the 160 KB of a CU.
    // (the large-system kernel plans its own LDS: everything below describes layouts A - D and stays unused for it)
    const int Wp = large ? 64 : W;
    const size_t with_tables = solve_lds_bytes(nx, nu, N, Wp, true);
    const size_t without = solve_lds_bytes(nx, nu, N, Wp, false);
    constexpr size_t kLdsMax = 160 * 1024;
    // Horizons whose state does not fit a CU's LDS run the same kernels on an HBM working copy (GMEM variant).
    s->state_in_global = !large && without > kLdsMax;
    if (const char *e = getenv("TINYMPC_STATE_GLOBAL")) s->state_in_global = !large && (e[0] == '1' || s->state_in_global);  // (experiments: the HBM working copy also where LDS would hold the state)
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
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); };
    s->setup_us[0] = us_since(t_setup);  // device, stream, events, the layout decisions above
    auto t_phase = clk::now();
    // ---- ONE block of device memory (ArenaPlan, tinympc_handle.h). Three spans:
    //   upload: the problem data, in the order of the pinned staging copy (one H2D copy below)
    ArenaPlan up;
    double *uA = nullptr, *uB = nullptr, *uf = nullptr, *uQd = nullptr, *uRd = nullptr, *uQ = nullptr, *uR = nullptr;  // staging slots
    up.add(&uA, (size_t)nx * nx); up.add(&uB, (size_t)nx * nu); up.add(&uf, nx); up.add(&uQd, nx); up.add(&uRd, nu);
    up.add(&uQ, (size_t)nx * nx); up.add(&uR, (size_t)nu * nu);
    const size_t upload_bytes = up.mark();
    ArenaPlan dev;
    dev.add(&s->dA, (size_t)nx * nx); dev.add(&s->dB, (size_t)nx * nu); dev.add(&s->dfdyn, nx); dev.add(&s->dQd, nx); dev.add(&s->dRd, nu);
    dev.add(&s->dQfull, (size_t)nx * nx); dev.add(&s->dRfull, (size_t)nu * nu);
    if (dev.mark() != upload_bytes) { destroy(s); return fail(TINYMPC_ERR_ALLOC, "setup: arena plans disagree"); }
    //   zeroed: everything tiny_setup zeroes (tiny_api.cpp:41-44, 73-88, 100-111) -- one memset
    const size_t zero_begin = dev.mark();
    dev.add(&s->ddK, (size_t)nu * nx); dev.add(&s->ddP, (size_t)nx * nx);
    dev.add(&s->dXref, X); dev.add(&s->dUref, U); dev.add(&s->dx0, (size_t)batch * nx);
    dev.add(&s->dG, s->state_doubles()); dev.add(&s->dV, s->v_doubles()); dev.add(&s->dV2, s->v_doubles()); dev.add(&s->dD, s->d_doubles());
    dev.add(&s->dsolx, X * batch); dev.add(&s->dsolu, U * batch);
    dev.add(&s->distats, (size_t)batch * 2); dev.add(&s->ddstats, (size_t)batch * 4);
    dev.add(&s->drefill, 1);
    const size_t zero_end = dev.mark();
    //   the rest: written by a kernel before anything reads it
    dev.add(&s->dKinf, (size_t)nu * nx); dev.add(&s->dPinf, (size_t)nx * nx);
    dev.add(&s->dQuu, (size_t)nu * nu); dev.add(&s->dAmBKt, (size_t)nx * nx);
    dev.add(&s->dAPf, nx); dev.add(&s->dBPf, nu); dev.add(&s->dinfo, 4);
    dev.add(&s->dscratch, (large ? precompute_large_scratch_doubles(nx, nu) : precompute_scratch_doubles(nx, nu)) + 8);
    dev.add(&s->dadapt, adapt_doubles(W, KT)); dev.add(&s->drho_inst, batch);
    if (s->c_tables) dev.add(&s->dctab, chunk_table_doubles(nx, s->chunk_levels));
    if (large && solve_m_tiled_ops_doubles(nx, nu)) dev.add(&s->dctab, solve_m_tiled_ops_doubles(nx, nu));  // layout M beyond 128 rows: tile-major operators
    dev.add(&s->dlqr_scratch, lqr_scratch_doubles(nx, nu) + 8); dev.add(&s->dlqr_out, 3 * s->cache_doubles());
    dev.add(&s->dxmin, X); dev.add(&s->dxmax, X); dev.add(&s->dumin, U); dev.add(&s->dumax, U);
    dev.add(&s->dops, ops_doubles(W, KT)); dev.add(&s->dtables, tables_doubles(W, N));
    if (s->state_in_global) dev.add(&s->dscratch_state, (size_t)s->groups * state_scratch_doubles(nu, N, W));
    if (!large && W == 16) {  // layouts E / F: the chunk operators of their plans (tinympc_plan.hip builds them at the launch that needs them)
        dev.add(&s->dctab_e, chunk_table_doubles(nx, 1)); dev.add(&s->dctab_f, chunk_table_doubles(nx, 4));
    }
    // ---- ... and ONE block of pinned host memory: the staging copy of the problem data and, for a single-instance handle, everything
    // the host and a running kernel exchange (coherent: see kBoundInf's neighbour comment in tinympc_handle.h)
    ArenaPlan pin;
    pin.add(&s->h_stage, upload_bytes / sizeof(double));
    if (batch == 1) {
        pin.add(&s->h_sol, X + U + 8); pin.add(&s->h_x0, nx); pin.add(&s->h_u0, nu);
        pin.add(&s->h_xref, X); pin.add(&s->h_uref, U ? U : 1); pin.add(&s->h_mail, 64); pin.add(&s->h_ans, 32);
    }
    TRY(acquire_arenas(s, dev.mark(), pin.mark(), batch == 1));
    dev.bind(s->arena_dev);
    s->setup_us[1] = us_since(t_phase); t_phase = clk::now();
    pin.bind(s->arena_pin);
    std::memset(s->arena_pin, 0, pin.total);  // h_sol, the references (tiny_api.cpp:83-84) and the mailbox start as zeros
    s->setup_us[2] = us_since(t_phase); t_phase = clk::now();
    // Problem data. Only the diagonals of Q and R are kept, each + rho (tiny_api.cpp:90-91).
    up.bind(s->h_stage);
    std::memcpy(uA, A, sizeof(double) * nx * nx); std::memcpy(uB, B, sizeof(double) * nx * nu);
    if (fdyn) std::memcpy(uf, fdyn, sizeof(double) * nx);
    for (int i = 0; i < nx; ++i) uQd[i] = Q[i + (size_t)i * nx] + rho;
    for (int i = 0; i < nu; ++i) uRd[i] = R[i + (size_t)i * nu] + rho;
    std::memcpy(uQ, Q, sizeof(double) * nx * nx); std::memcpy(uR, R, sizeof(double) * nu * nu);
    if (zero_end - zero_begin <= ((size_t)1 << 20)) {
        // a small handle: ONE launch does the upload (reading the pinned staging copy itself), the zeroing, the fills (k_setup_init)
        SetupInitParams ip{};
        ip.stage = s->h_stage; ip.upload_dst = s->dA; ip.upload_doubles = upload_bytes / sizeof(double);
        ip.zero = reinterpret_cast<double *>(static_cast<char *>(s->arena_dev) + zero_begin); ip.zero_doubles = (zero_end - zero_begin) / sizeof(double);
        ip.xmin = s->dxmin; ip.xmax = s->dxmax; ip.umin = s->dumin; ip.umax = s->dumax; ip.X = X; ip.U = U; ip.inf = kBoundInf;  // TinyMPC.m:261-264
        ip.rho_inst = s->drho_inst; ip.batch = batch; ip.rho = rho; ip.mail = s->d_mail;
        HIP_TRY_S(launch_setup_init(ip, s->stream));
    } else {
        HIP_TRY_S(hipMemcpyAsync(s->dA, s->h_stage, upload_bytes, hipMemcpyHostToDevice, s->stream));
        HIP_TRY_S(hipMemsetAsync(static_cast<char *>(s->arena_dev) + zero_begin, 0, zero_end - zero_begin, s->stream));
        if (s->d_mail) HIP_TRY_S(hipMemsetAsync(s->d_mail, 0, sizeof(double) * 64, s->stream));
        HIP_TRY_S(launch_fill_bounds(s->dxmin, s->dxmax, X, s->dumin, s->dumax, U, kBoundInf, s->stream));  // TinyMPC.m:261-264
        HIP_TRY_S(launch_reset_stats(s->distats, s->ddstats, s->drho_inst, batch, rho, s->stream));
    }
    s->setup_us[3] = us_since(t_phase); t_phase = clk::now();
    TRY(run_precompute(s));  // tiny_api.cpp:113
    s->setup_us[4] = us_since(t_phase); t_phase = clk::now();
    HIP_TRY_S(hipStreamSynchronize(s->stream));
    s->setup_us[5] = us_since(t_phase);
    s->setup_us[6] = us_since(t_setup);
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
                s->dbg_tick[2] = (double)spin;
                s->dbg_tick[3] = 0.0;
                return TINYMPC_OK;
            }
            __builtin_ia32_pause();
        }
        s->dbg_tick[3] = 1.0;  // the polling budget ran out
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->flag_pending = false;
    if (s->host_sol_state == 1) s->host_sol_state = 2;
    return TINYMPC_OK;
}

// Resident solves (round 5, an extension; off by default): the reference's per-tick sequence  set_x0 -> solve -> get_solution  served by the
// resident session kernel instead of a launch per solve -- what tinympc_session_step does, behind the reference's own verbs, so that a
// script written against the reference needs ONE extra line (solver.set_resident(true)) for a 2.5x shorter tick. Bit-identical to
// launched solves (the session's contract). Everything that ends a session (any verb that needs the device) ends this one too; the next
// solve opens it again. Where no resident kernel exists for the configuration (batched handles, adaptive rho, wide systems) solves are
// launched as before.
int tinympc_set_resident(tinympc_solver *s, int enable) {
    int rc = check_handle(s);
    if (rc) return rc;
    s->resident_solves = enable != 0;
    s->resident_refused = false;
    if (!enable && s->session_active) {
        HIP_TRY(hipSetDevice(s->device));
        return end_session(s);
    }
    return TINYMPC_OK;
}

static int print_solve_result(tinympc_solver *s) {
    int it = 0, st = 0;
    int rc = tinympc_get_stats(s, &it, &st, nullptr, nullptr, 0);
    if (rc) return rc;
    if (st == TINYMPC_STATUS_SOLVED) printf("Solver converged in %d iterations\n", it);  // admm.cpp:190
    printf("Solve completed with status: %d\n", st == TINYMPC_STATUS_SOLVED ? 0 : 1);
    return TINYMPC_OK;
}

int tinympc_solve(tinympc_solver *s, int verbose) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (s->resident_solves && !s->resident_refused && s->host_path()) {
        if (!s->session_active) {
            rc = tinympc_session_begin(s);
            if (rc == TINYMPC_ERR_UNSUPPORTED || rc == TINYMPC_ERR_INVALID_INPUT) s->resident_refused = true;  // (launched solves from here on)
            else if (rc) return rc;
        }
        if (s->session_active) {
            double u0[16];
            if ((rc = tinympc_session_step(s, s->h_x0, u0))) return rc;  // (x0: what tinympc_set_x0 left in the pinned buffer)
            s->x0_on_host = false;
            return verbose ? print_solve_result(s) : TINYMPC_OK;
        }
    }
    rc = tinympc_solve_async(s);
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

int tinympc_solve_queued(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    if (s->session_active || s->host_path()) return fail(TINYMPC_ERR_UNSUPPORTED, "solve_queued: batched handles outside a session only");
    if (s->ring_count >= 4096) return fail(TINYMPC_ERR_INVALID_INPUT, "solve_queued: 4096 launches queued, collect their times first");
    if ((rc = refresh_derived(s))) return rc;  // keep table rebuilds out of the timed region
    while ((int)s->ring_ev.size() < 2 * (s->ring_count + 1)) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        s->ring_ev.push_back(e);
    }
    // the launch records the handle's event pair: lend it this launch's slot of the ring
    hipEvent_t keep0 = s->ev0, keep1 = s->ev1;
    s->ev0 = s->ring_ev[2 * (size_t)s->ring_count];
    s->ev1 = s->ring_ev[2 * (size_t)s->ring_count + 1];
    rc = launch(s, true);
    s->ev0 = keep0;
    s->ev1 = keep1;
    if (rc) return rc;
    s->ring_count++;
    return TINYMPC_OK;
}

int tinympc_collect_kernel_ms(tinympc_solver *s, float *kernel_ms, int capacity, int *count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int n = s->ring_count;
    if (count) *count = n;
    if (n > capacity || (n > 0 && !kernel_ms)) return fail(TINYMPC_ERR_INVALID_INPUT, "collect_kernel_ms: %d launches queued, room for %d", n, capacity);
    for (int i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&kernel_ms[i], s->ring_ev[2 * (size_t)i], s->ring_ev[2 * (size_t)i + 1]));
    s->ring_count = 0;
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
        ArenaPlan pin;  // (one block for both; batched handles only -- a single-instance handle has them in its setup arena)
        pin.add(&s->h_x0, nx0); pin.add(&s->h_u0, nu0);
        void *base = nullptr;
        HIP_TRY(hipHostMalloc(&base, pin.mark(), hipHostMallocCoherent));
        s->host_allocs.push_back(base);
        pin.bind(base);
    }
    if (s->host_sol_state == 1) {  // a launch of tinympc_solve_async may still be reading h_x0
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->host_sol_state = 2;
    }
    std::memcpy(s->h_x0, x0s, sizeof(double) * nx0);
    if ((rc = resolve_plan(s))) return rc;
    int zc_max = current_plan(s).layout == 'D' ? kZeroCopyTickMaxD : kZeroCopyTickMax;
    if (const char *e = getenv("TINYMPC_ZERO_COPY_MAX")) zc_max = atoi(e);  // (experiments)
    if (s->batch <= zc_max && s->st.max_iter > 0 && current_plan(s).host_exchange) {
        // Small batches: no copy engine at all. The kernel reads x0 from the pinned host buffer (and mirrors it into
        // the device copy the other verbs use) and writes the first controls into the pinned host buffer; both
        // are device-visible host allocations, and the stream synchronisation makes the writes visible here.
        s->zero_copy_tick = true;
        const auto t_a = std::chrono::steady_clock::now();
        rc = launch(s, false);
        const auto t_b = std::chrono::steady_clock::now();
        s->zero_copy_tick = false;
        if (rc) return rc;
        if ((rc = tinympc_synchronize(s))) return rc;  // (single instance: polls the completion flag in pinned memory)
        const auto t_c = std::chrono::steady_clock::now();
        s->dbg_tick[0] = std::chrono::duration<double, std::micro>(t_b - t_a).count();  // launch
        s->dbg_tick[1] = std::chrono::duration<double, std::micro>(t_c - t_b).count();  // wait
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




// ---- diagnostics (include/tinympc_hip_bench.h; not part of the drop-in boundary)
int tinympc_debug_setup_timing(tinympc_solver *s, double *out10) {
    int rc = check_handle(s);
    if (rc) return rc;
    for (int i = 0; i < 7; ++i) out10[i] = s->setup_us[i];
    int info[4] = {0, 0, 0, 0};
    if ((rc = bind_device(s))) return rc;
    if ((rc = download(s, info, s->dinfo, sizeof(info)))) return rc;
    out10[7] = (double)info[1];         // shader clocks of the Riccati loop (k_precompute_rows; 0 from the other precompute kernels)
    out10[8] = 0.01 * (double)info[2];  // ... its duration in us (100 MHz counter)
    out10[9] = (double)info[0];         // Riccati steps
    return TINYMPC_OK;
}

double tinympc_debug_mail_stamp(double sequence_number, const double *words7) { return tinympc::mail_stamp_of(sequence_number, words7); }

int tinympc_debug_tick_timing(tinympc_solver *s, double *out4) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (s->session_active && s->session_on_f && s->h_sol) {
        // layout F's resident kernel (not layout C's) leaves its own split of the last tick in the spare slot behind the completion stamp: us it waited
        // for the command since its previous answer, us of ADMM iterations, us of write-out (16 bits each, ticks of 10 ns)
        if (s->host_sol_state == 3 && (rc = wait_session_solution(s))) return rc;
        const unsigned long long packed = (unsigned long long)s->h_sol[s->X() + s->U() + 7];
        out4[0] = 0.01 * (double)(packed & 0xffffull);
        out4[1] = 0.01 * (double)((packed >> 16) & 0xffffull);
        out4[2] = 0.01 * (double)((packed >> 32) & 0xffffull);
        out4[3] = 0.0;
        return TINYMPC_OK;
    }
    for (int i = 0; i < 4; ++i) out4[i] = s->dbg_tick[i];
    return TINYMPC_OK;
}

int tinympc_get_solution_batch(tinympc_solver *s, double *x_out, double *u_out, int first, int count) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (first < 0 || count < 0 || first + count > s->batch)
        return fail(TINYMPC_ERR_INVALID_INPUT, "instance range [%d, %d) outside batch of %d", first, first + count, s->batch);
    if (s->host_path() && count == 1 && s->host_sol_state != 0) {
        if (s->host_sol_state == 1 && (rc = tinympc_synchronize(s))) return rc;
        if (s->host_sol_state == 3 && (rc = wait_session_solution(s))) return rc;
        if (x_out) std::memcpy(x_out, s->h_sol, sizeof(double) * s->X());
        if (u_out) std::memcpy(u_out, s->h_sol + s->X(), sizeof(double) * s->U());
        return TINYMPC_OK;
    }
    if ((rc = bind_device(s))) return rc;
    if ((rc = materialize_zero_solution(s))) return rc;
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
    if ((rc = materialize_zero_solution(s))) return rc;
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
        if (s->host_sol_state == 3 && (rc = wait_session_solution(s))) return rc;
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
    // (up to MAX_LIN_ROWS rows per side every kernel holds; beyond, the structure-specialised kernels -- layouts E and F -- size
    // their copies from the count itself: checked at launch, where the kernel is known)
    if (nlx > HARD_MAX_LIN_ROWS || nlu > HARD_MAX_LIN_ROWS)
        return fail(TINYMPC_ERR_UNSUPPORTED, "at most %d linear rows per side are supported by the HIP kernels (got %d state, %d input)",
                    HARD_MAX_LIN_ROWS, nlx, nlu);
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
    // pairwise-disjoint cones (family_structure). Up to MAX_CONES cones in MAX_ROUNDS rounds every families kernel holds; up to
    // HARD_MAX_CONES in any number of rounds run on the structure-specialised kernels (layouts E and F).
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
    if (ncx + ncu > HARD_MAX_CONES)
        return fail(TINYMPC_ERR_UNSUPPORTED, "at most %d cones are supported by the HIP kernels (got %d state + %d input)", HARD_MAX_CONES, ncx, ncu);
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
        if (s->nx + s->nu <= 64) {  // (larger systems: layout M walks the cone list in order, it has no rounds)
            walk(Acx, qcx, ncx, 0);
            walk(Acu, qcu, ncu, s->nx);
        }
        // (more than MAX_ROUNDS rounds, like more than MAX_CONES cones: the structure-specialised kernels only -- checked at launch)
        (void)rounds;
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
    // G, V, D: zero by CONTRACT from here on, written to HBM only if the next kernel needs them there (materialize_cold_state):
    // the throughput kernel starts a cold solve from zero registers instead of loading 14 KB of zeros per instance that a
    // memset would have had to write first (119 MB per 8,192 quadrotor instances, ~35 us of a 1.86 ms step).
    // (V2, the stale-copy buffer of layout B, and LX, the families' forward -> backward term, are always written
    //  before they are read within a solve: nothing to reset)
    s->cold_state = true;
    if (s->dGC) {
        HIP_TRY(hipMemsetAsync(s->dGC, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dGL, 0, sizeof(double) * s->v_doubles(), s->stream));
    }
    s->sol_zero_pending = true;  // (sol_x / sol_u: zero by contract; the next solve overwrites them, a reader before that gets zeros written first)
    if (s->sol_ptrs_exported && (rc = materialize_zero_solution(s))) return rc;  // ... a reader that holds the device pointers comes through no verb
    HIP_TRY(launch_reset_stats(s->distats, s->ddstats, s->drho_inst, s->batch, s->rho, s->stream));  // (adapted rho back to the setup value)
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
    if ((rc = bind_device(s))) return rc;
    if ((rc = materialize_zero_solution(s))) return rc;
    s->sol_ptrs_exported = true;  // from here on tinympc_reset_workspace zeroes the buffers itself instead of deferring it
    if (d_x) *d_x = s->dsolx;
    if (d_u) *d_u = s->dsolu;
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
