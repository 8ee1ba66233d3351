// tinympc_handle.hip -- host-side helpers of the solver handle (tinympc_handle.h): device buffers and copies, the lazily rebuilt
// operators / per-knot tables, the per-lane description of the cone / linear families. No solver arithmetic happens here.
#include "tinympc_handle.h"

#include <cstdlib>

#include <atomic>
#include <cstdarg>
#include <cstring>
#include <limits>

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

namespace host {

// Streams and events are kept for the life of the process: hipStreamCreateWithFlags takes 1.4 ms on MI355X / ROCm 7.2 and
// hipStreamDestroy 1.3 ms (profiles/r05_setup_time.txt) -- more than the rest of tinympc_setup_batch and twenty times the
// reference's whole tiny_setup. A handle takes a stream + its event pair from the device's pool and puts them back when it is
// destroyed (idle: destroy() has waited for the stream); the MEX usage -- one global solver, `setup` replacing it again and again
// (bindings.cpp:17, 92) -- then pays for one stream once.
namespace {
struct StreamKit { hipStream_t stream; hipEvent_t ev0, ev1; };
std::mutex g_kits_mu;
std::vector<StreamKit> g_kits[64];  // [device]
constexpr size_t kKitsKept = 16;    // per device; beyond that a returned stream is destroyed
}  // namespace

// ... and so are the memory blocks of small handles: hipFree + hipHostFree of a single-instance handle's two arenas take 0.2 ms
// (three times the reference's whole tiny_setup), and a MEX user's `setup` destroys the previous solver first (bindings.cpp:92).
// Arenas of up to kArenaKeepBytes go back to a per-device pool when their handle is destroyed and serve the next setup that fits.
namespace {
struct ArenaKit { void *dev; size_t dev_bytes; void *pin; size_t pin_bytes; void *mail; };
std::vector<ArenaKit> g_arenas[64];             // [device]; under g_kits_mu
constexpr size_t kArenaKeepBytes = 4u << 20;    // device bytes of an arena worth keeping (a quadrotor N=50 instance: 0.1 MB)
constexpr size_t kArenasKept = 8;
}  // namespace

// HIP streams are multiplexed onto a few hardware queues -- four per process unless GPU_MAX_HW_QUEUES says otherwise --, and a hardware queue
// runs its packets in order: with more streams alive than queues, a launch can land BEHIND another handle's resident session kernel and
// wait until that kernel's idle time-out (2 s), tick after tick (tools/thread_sessions.py, round 5: three host threads with a session and
// a launched twin each stalled in five runs of six; none of twelve with 16 queues). The runtime reads the variable when it initialises:
// set a default when this library is loaded (never over the user's own value). A process that initialised HIP before loading the library
// has to set it itself -- INTEGRATION.md.
__attribute__((constructor(101))) static void tinympc_default_environment() { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); }

// Can the host store into device memory (the whole of it mapped through the PCIe BAR)? Asked once per device.
static bool device_is_large_bar(int device) {
    static std::atomic<int> cache[64];  // 0: not asked, 1: no, 2: yes
    if (device < 0 || device >= 64) return false;
    int v = cache[device].load(std::memory_order_relaxed);
    if (v == 0) {
        int flag = 0;
        v = (hipDeviceGetAttribute(&flag, hipDeviceAttributeIsLargeBar, device) == hipSuccess && flag) ? 2 : 1;
        cache[device].store(v, std::memory_order_relaxed);
    }
    return v == 2;
}

int acquire_arenas(tinympc_solver *s, size_t dev_bytes, size_t pin_bytes, bool want_mailbox) {
    // The session's mailbox on the DEVICE side of the bus (round 5): a line of fine-grained device memory the host stores into through
    // the BAR (posted writes) and the resident kernel polls in its own HBM -- a poll is 0.2 us instead of a 1.2 us PCIe read
    // (tools/mailbox_probe.hip, profiles/r05_mailbox_probe.txt). Where the device's memory is not mapped into the host's address space,
    // or the allocation is refused, the mailbox stays in pinned host memory (h_mail). TINYMPC_MAILBOX=host forces that (A/B runs).
    const char *env = getenv("TINYMPC_MAILBOX");
    const bool use_mail = want_mailbox && device_is_large_bar(s->device) && !(env && (env[0] == 'h' || env[0] == 'H'));
    s->arena_mail = nullptr;
    s->d_mail = nullptr;
    if (s->device >= 0 && s->device < 64) {
        std::lock_guard<std::mutex> lock(g_kits_mu);
        auto &pool = g_arenas[s->device];
        for (size_t i = 0; i < pool.size(); ++i) {
            if (pool[i].dev_bytes >= dev_bytes && pool[i].pin_bytes >= pin_bytes && (pool[i].mail != nullptr || !use_mail)) {
                s->arena_dev = pool[i].dev; s->arena_dev_bytes = pool[i].dev_bytes;
                s->arena_pin = pool[i].pin; s->arena_pin_bytes = pool[i].pin_bytes;
                s->arena_mail = pool[i].mail;  // (owned with the arenas; used only if this handle wants a device-side mailbox)
                s->d_mail = use_mail ? static_cast<double *>(pool[i].mail) : nullptr;
                pool.erase(pool.begin() + (long)i);
                return TINYMPC_OK;
            }
        }
    }
    if (use_mail) {
        void *m = nullptr;
        if (hipExtMallocWithFlags(&m, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
            s->arena_mail = m;
            s->d_mail = static_cast<double *>(m);
        } else {
            (void)hipGetLastError();
        }
    }
    hipError_t e = hipMalloc(&s->arena_dev, dev_bytes);
    if (e != hipSuccess) return fail(TINYMPC_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", dev_bytes, hipGetErrorString(e));
    s->arena_dev_bytes = dev_bytes;
    e = hipHostMalloc(&s->arena_pin, pin_bytes, hipHostMallocCoherent);
    if (e != hipSuccess) return fail(TINYMPC_ERR_ALLOC, "hipHostMalloc(%zu bytes) failed: %s", pin_bytes, hipGetErrorString(e));
    s->arena_pin_bytes = pin_bytes;
    return TINYMPC_OK;
}

static void release_arenas(tinympc_solver *s) {
    if (s->arena_dev && s->arena_pin && s->arena_dev_bytes <= kArenaKeepBytes && s->device >= 0 && s->device < 64) {
        std::lock_guard<std::mutex> lock(g_kits_mu);
        auto &pool = g_arenas[s->device];
        if (pool.size() < kArenasKept) {
            pool.push_back({s->arena_dev, s->arena_dev_bytes, s->arena_pin, s->arena_pin_bytes, s->arena_mail});
            s->arena_dev = s->arena_pin = s->arena_mail = nullptr;
            s->d_mail = nullptr;
            return;
        }
    }
    if (s->arena_dev) (void)hipFree(s->arena_dev);
    if (s->arena_pin) (void)hipHostFree(s->arena_pin);
    if (s->arena_mail) (void)hipFree(s->arena_mail);
    s->arena_dev = s->arena_pin = s->arena_mail = nullptr;
    s->d_mail = nullptr;
}

int acquire_stream_kit(tinympc_solver *s) {
    if (s->device >= 0 && s->device < 64) {
        std::lock_guard<std::mutex> lock(g_kits_mu);
        auto &pool = g_kits[s->device];
        if (!pool.empty()) {
            s->stream = pool.back().stream; s->ev0 = pool.back().ev0; s->ev1 = pool.back().ev1;
            pool.pop_back();
            return TINYMPC_OK;
        }
    }
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));
    return TINYMPC_OK;
}

static void release_stream_kit(tinympc_solver *s) {
    if (s->stream && s->ev0 && s->ev1 && s->device >= 0 && s->device < 64 && hipStreamQuery(s->stream) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_kits_mu);
        auto &pool = g_kits[s->device];
        if (pool.size() < kKitsKept) {
            pool.push_back({s->stream, s->ev0, s->ev1});
            s->stream = nullptr; s->ev0 = s->ev1 = nullptr;
            return;
        }
    }
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    s->stream = nullptr; s->ev0 = s->ev1 = nullptr;
}

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

int materialize_cold_state(tinympc_solver *s) {
    if (!s->cold_state) return TINYMPC_OK;
    HIP_TRY(hipMemsetAsync(s->dG, 0, sizeof(double) * s->state_doubles(), s->stream));
    HIP_TRY(hipMemsetAsync(s->dV, 0, sizeof(double) * s->v_doubles(), s->stream));
    HIP_TRY(hipMemsetAsync(s->dD, 0, sizeof(double) * s->d_doubles(), s->stream));
    s->cold_state = false;
    return TINYMPC_OK;
}

int materialize_zero_solution(tinympc_solver *s) {
    if (!s->sol_zero_pending) return TINYMPC_OK;
    HIP_TRY(hipMemsetAsync(s->dsolx, 0, sizeof(double) * s->X() * s->batch, s->stream));
    HIP_TRY(hipMemsetAsync(s->dsolu, 0, sizeof(double) * s->U() * s->batch, s->stream));
    s->sol_zero_pending = false;
    return TINYMPC_OK;
}

int run_precompute(tinympc_solver *s) {
    PrecomputeParams p{};
    p.nx = s->nx; p.nu = s->nu; p.rho = s->rho;
    p.A = s->dA; p.B = s->dB; p.fdyn = s->dfdyn; p.Qd = s->dQd; p.Rd = s->dRd;
    p.Kinf = s->dKinf; p.Pinf = s->dPinf; p.Quu_inv = s->dQuu; p.AmBKt = s->dAmBKt; p.APf = s->dAPf; p.BPf = s->dBPf;
    p.info = s->dinfo; p.scratch = s->dscratch;
    p.use_lds = precompute_scratch_doubles(s->nx, s->nu) <= 6500 ? 1 : 0;
    // TINYMPC_PRECOMPUTE=lds: the one-workgroup LDS kernel also where the register-resident one applies (A/B runs, tests)
    const char *env = getenv("TINYMPC_PRECOMPUTE");
    const bool rows = precompute_rows_supported(s->nx, s->nu) && !(env && (env[0] == 'l' || env[0] == 'L'));
    if (s->layout_m) HIP_TRY(launch_precompute_large(p, s->stream));  // (large systems: one launch per matrix product, on the matrix cores)
    else if (rows) HIP_TRY(launch_precompute_rows(p, s->stream));     // (nx <= 12, nu <= 4: registers of one wavefront)
    else HIP_TRY(launch_precompute(p, s->stream));
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
        if (s->layout_m && s->dctab) HIP_TRY(launch_tile_operators_m(s->dops, s->dctab, s->nx, s->nu, s->stream));  // (beyond 128 rows)
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
        if (!s->dftab || s->dftab_cap < s->f_chunk_len) {  // (a longer chunk than before: a new block, the old one stays on the free list)
            int rc = dalloc(s, &s->dftab, f_input_table_doubles(s->nu, s->f_chunk_len));
            if (rc) return rc;
            s->dftab_cap = s->f_chunk_len;
        }
        c.out = s->dftab;
        HIP_TRY(launch_build_f_input_tables(c, s->stream));
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
FamilyStructure family_structure(const tinympc_solver *s, double *mu) {
    FamilyStructure fs;
    const bool cone_x = s->st.en_state_soc && s->n_cone_x > 0, cone_u = s->st.en_input_soc && s->n_cone_u > 0;
    unsigned long long used = 0;  // lanes taken by the cones of the current round
    auto add = [&](bool on, const std::vector<int> &Ac, const std::vector<int> &qc, const std::vector<double> &c, int base) {
        if (!on) return;
        for (size_t k = 0; k < Ac.size() && fs.ncone < HARD_MAX_CONES; ++k) {
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
// Layout M (nx+nu > 64): the compact description of tinympc_solve_m.hip -- counts, the active cones in list order as first row /
// last row / slope, one coefficient vector per linear row k (state side row k and input side row k share it: disjoint rows).
static int refresh_families_m(tinympc_solver *s) {
    const int nx = s->nx, nu = s->nu, GW = solve_m_geometry(nx, nu);
    std::vector<double> cmu(HARD_MAX_CONES, 0.0);
    const FamilyStructure fs = family_structure(s, cmu.data());
    const int nl = fs.nlx > fs.nlu ? fs.nlx : fs.nlu;
    const size_t need = solve_m_fam_doubles(nx, nu, nl);
    int rc;
    if (!s->dfam || need > s->fam_alloc_doubles) {
        if ((rc = dalloc(s, &s->dfam, need))) return rc;
        s->fam_alloc_doubles = need;
        s->fam_dirty = true;
    }
    if (!s->dGC) {  // (allocated once, in the tile layout: v_doubles())
        if ((rc = dalloc(s, &s->dGC, s->v_doubles()))) return rc;
        if ((rc = dalloc(s, &s->dGL, s->v_doubles()))) return rc;
        if ((rc = dalloc(s, &s->dLX, s->v_doubles()))) return rc;
        HIP_TRY(hipMemsetAsync(s->dGC, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dGL, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dLX, 0, sizeof(double) * s->v_doubles(), s->stream));
        s->fam_dirty = true;
    }
    if (!s->fam_dirty) return TINYMPC_OK;
    std::vector<double> f(need, 0.0);
    int ncx = 0, ncu = 0;
    for (int c = 0; c < fs.ncone; ++c) {
        double *cd = f.data() + solve_m_fam_cone_offset() + (size_t)3 * c;
        cd[0] = (double)fs.cone[c][1];
        cd[1] = (double)fs.cone[c][2];
        cd[2] = cmu[c];
        (fs.cone[c][1] < nx ? ncx : ncu) += 1;
    }
    f[0] = (double)ncx; f[1] = (double)ncu; f[2] = (double)fs.nlx; f[3] = (double)fs.nlu;
    const double inf = std::numeric_limits<double>::infinity();
    for (int k = 0; k < nl; ++k) {
        double *ak = f.data() + solve_m_fam_lin_offset() + (size_t)k * (GW + 4);
        double nrm_x = 0.0, nrm_u = 0.0;
        ak[GW] = inf; ak[GW + 1] = inf; ak[GW + 2] = 1.0; ak[GW + 3] = 1.0;
        if (k < fs.nlx) {
            for (int c = 0; c < nx; ++c) { const double a = s->Alin_x[k + (size_t)c * s->n_lin_x]; ak[c] = a; nrm_x += a * a; }
            ak[GW] = s->blin_x[k];
            ak[GW + 2] = 1.0 / nrm_x;
        }
        if (k < fs.nlu) {
            for (int c = 0; c < nu; ++c) { const double a = s->Alin_u[k + (size_t)c * s->n_lin_u]; ak[nx + c] = a; nrm_u += a * a; }
            ak[GW + 1] = s->blin_u[k];
            ak[GW + 3] = 1.0 / nrm_u;
        }
    }
    if (nl <= solve_m_fam_fast_rows()) {  // the rows' Gram matrices (the one-pass form of the phase): [side][k][j] = a_k' a_j
        f[4] = 1.0;
        double *Gx = f.data() + solve_m_fam_lin_offset() + (size_t)nl * (GW + 4), *Gu = Gx + (size_t)nl * nl;
        for (int k = 0; k < nl; ++k)
            for (int j = 0; j < nl; ++j) {
                const double *ak = f.data() + solve_m_fam_lin_offset() + (size_t)k * (GW + 4), *aj = f.data() + solve_m_fam_lin_offset() + (size_t)j * (GW + 4);
                double gx = 0.0, gu = 0.0;
                for (int c = 0; c < nx; ++c) gx += ak[c] * aj[c];
                for (int c = nx; c < nx + nu; ++c) gu += ak[c] * aj[c];
                Gx[(size_t)k * nl + j] = gx;
                Gu[(size_t)k * nl + j] = gu;
            }
    }
    if (const char *env = getenv("TINYMPC_M_FAM_FAST")) f[4] = (env[0] == '0') ? 0.0 : f[4];  // (A/B: the general form of the phase)
    if ((rc = upload(s, s->dfam, f.data(), f.size()))) return rc;
    s->fam_dirty = false;
    return TINYMPC_OK;
}

int refresh_families(tinympc_solver *s) {
    if (s->layout_m) return refresh_families_m(s);
    const int W = s->W, KT = s->KT, nx = s->nx, nu = s->nu, nxu = nx + nu;
    // the buffer's capacities: the generic kernels' (the default layout every kernel reads), or this configuration's own counts
    // where it has more -- such a configuration runs on layouts E / F only (launch() checks)
    std::vector<double> cmu(HARD_MAX_CONES, 0.0);
    const FamilyStructure fs = family_structure(s, cmu.data());
    const int lin_cap = fam_lin_cap(fs.nlx > fs.nlu ? fs.nlx : fs.nlu), cone_cap = fam_cone_cap(fs.ncone), round_cap = fam_round_cap(fs.nround);
    // (the layout follows the CURRENT configuration's capacities exactly: the specialised kernels derive theirs from the counts
    // they were compiled for; a block that has become too small is replaced, the old one stays on the handle's free list)
    const size_t need = fam_doubles(W, KT, lin_cap, cone_cap, round_cap);
    if (!s->dfam || need > s->fam_alloc_doubles) {
        int rc;
        if ((rc = dalloc(s, &s->dfam, need))) return rc;
        s->fam_alloc_doubles = need;
        s->fam_dirty = true;
    }
    if (lin_cap != s->fam_lin_cap || cone_cap != s->fam_cone_cap || round_cap != s->fam_round_cap) {
        s->fam_lin_cap = lin_cap;
        s->fam_cone_cap = cone_cap;
        s->fam_round_cap = round_cap;
        s->fam_dirty = true;
    }
    if (!s->dGC) {
        int rc;
        if ((rc = dalloc(s, &s->dGC, s->v_doubles()))) return rc;
        if ((rc = dalloc(s, &s->dGL, s->v_doubles()))) return rc;
        if ((rc = dalloc(s, &s->dLX, s->v_doubles()))) return rc;
        HIP_TRY(hipMemsetAsync(s->dGC, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dGL, 0, sizeof(double) * s->v_doubles(), s->stream));
        HIP_TRY(hipMemsetAsync(s->dLX, 0, sizeof(double) * s->v_doubles(), s->stream));
        s->fam_dirty = true;
    }
    if (!s->fam_dirty) return TINYMPC_OK;
    const int LC = s->fam_lin_cap, CC = s->fam_cone_cap, RC = s->fam_round_cap;
    std::vector<double> f(fam_doubles(W, KT, LC, CC, RC), 0.0);
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
        f[fam_nround_offset(W, KT, LC, CC)] = (double)(fs.nround > 0 ? fs.nround : 1);
        for (int c = 0; c < fs.ncone; ++c) {
            const int q = fs.cone[c][0], first = fs.cone[c][1], last = fs.cone[c][2];
            double *rl = role, *m = mu, *cn = Cn, *ct = Ct;
            if (q >= 1) {
                rl = f.data() + fam_round_offset(W, KT, q, LC, CC);
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
    for (int k = 0; k < LC; ++k) {
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
    for (int c = 0; c < fs.ncone; ++c) f[fam_cone_mu_offset(W, KT, LC) + c] = cmu[c];  // slopes of the active cones, in list order (layouts E, F)
    int rc = upload(s, s->dfam, f.data(), f.size());
    if (rc) return rc;
    s->fam_dirty = false;
    return TINYMPC_OK;
}

void destroy(tinympc_solver *s) {
    if (!s) return;
    // teardown is best effort: errors here have nowhere useful to go
    (void)hipSetDevice(s->device);
    if (s->session_active) (void)end_session(s);
    park_sessions_on_device(s->device, s);  // (hipFree synchronises the device: no other handle's resident kernel may be spinning)
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (void *q : s->allocs) (void)hipFree(q);
    for (void *q : s->host_allocs) (void)hipHostFree(q);
    release_arenas(s);  // (small ones: back to the device's pool)
    for (hipEvent_t e : s->ring_ev) (void)hipEventDestroy(e);
    s->ring_ev.clear();
    release_stream_kit(s);  // (back to the device's pool; an errored stream is destroyed)
    delete s;
}

}  // namespace host
}  // namespace tinympc
