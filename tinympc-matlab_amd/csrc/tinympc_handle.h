// tinympc_handle.h -- the solver handle behind the C ABI (include/tinympc_hip.h) and the host-side helpers its translation units
// share:  tinympc_capi.hip (the verbs), tinympc_handle.hip (buffers, derived tables, the families' description),
//         tinympc_plan.hip (WHICH kernel a launch runs -- one LaunchPlan, chosen in one function -- and the launch itself),
//         tinympc_session.hip (the resident closed-loop session).
// Host-side bookkeeping only: every number the solver produces is computed by the kernels; there is no CPU fallback anywhere.
#pragma once
#include "tinympc_hip.h"
#include "tinympc_hip_bench.h"  // (the diagnostics the product library itself exports)

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "tinympc_device.h"
#include "tinympc_host.h"

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return tinympc::fail(TINYMPC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                 __FILE__, __LINE__);                                                     \
    } while (0)

namespace tinympc {

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
constexpr int kZeroCopyTickMaxD = 1 << 30;  // ... on layout D: any batch (measured: tools/tick_latency.py)
constexpr int kLayoutEBatchMin = 260;   // families at horizons layout D cannot hold: from here on layout E (4 instances per CU, the whole
                                        // state on chip) passes the latency kernel (1 instance per CU); measured, profiles/r03_rocket_sweep.txt
constexpr int kLayoutCBatchMax = 768;  // above this the batch-oriented layouts win (profiles/r02_layout_sweep.txt: layout D with four
                                       // wavefronts per workgroup passes the latency kernel between 512 and 1,024 instances)


}  // namespace tinympc

struct tinympc_solver {
    int nx = 0, nu = 0, N = 0, batch = 0, device = 0;
    int W = 0, KT = 0, IPW = 0, groups = 0;
    double rho = 0.0;
    tinympc::Settings st{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> ring_ev;  // tinympc_solve_queued: one event pair per queued launch (2 i, 2 i + 1)
    int ring_count = 0;               // pairs recorded since the last tinympc_collect_kernel_ms
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
    int *drefill = nullptr;  // slot refill (layout D): the next-instance counter of a launch (SolveParams::refill_next)
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
    double dbg_tick[4] = {0.0, 0.0, 0.0, 0.0};  // tinympc_debug_tick_timing
    bool sol_zero_pending = false;  // the device solution is zero by contract (reset_workspace), not yet zeroed: every solve of >= 1 iteration overwrites all of it
    bool sol_ptrs_exported = false;  // tinympc_get_solution_device_ptrs has handed the buffers out: a reset zeroes them eagerly (the caller reads them without a verb)
    bool cold_state = false;  // G, V, D are zero by contract (reset_workspace) but NOT yet zeroed in HBM: see SolveParams::cold
    bool d_varying_jit = false;  // ... and that kernel is a run-time specialisation even if the constant-table one is compiled in
    int d_adapt = -1;       // ... and with adaptive rho
    int d_fam = -1;         // layout D with the cone / linear families (run-time specialised, short horizons): -1 not asked yet, 0 no, 1 yes
    int d_varying = -1;     // layout D with bounds / references that vary over the horizon: -1 not asked yet, 0 no, 1 yes
    // Layout E (tinympc_solve_e.hip, run-time specialised on the families' STRUCTURE): the horizon cut across the wavefronts of a
    // workgroup -- the throughput kernel for the families at horizons layout D cannot hold. Decided per launch
    // (decide_layout_variants): `e_sig` is the structure / table kind the answer `e_ok` belongs to.
    tinympc::FamilyStructure fs;
    std::string e_sig;
    bool e_ok = false;
    int e_chunk_len = 0, e_wpg = 0, e_gpw = 1;
    size_t e_lds = 0;
    // Layout F (tinympc_solve_f.hip): the latency kernel as a run-time specialisation (shape, chunk plan and the families'
    // structure compiled in); decided per launch like layout E
    std::string f_sig;
    bool f_ok = false;
    tinympc::FamilyStructure f_fs;
    signed char f_box_builtin[2] = {-1, -1};  // is the box path's layout F kernel compiled in? [tables constant?]; -1: not asked yet
    unsigned f_box_key = 0;  // decide_layout_f, box path: what the current f_sig was decided for (0: nothing)
    bool specialise_asked = false;  // tinympc_prepare() was called: run-time specialisation is welcome wherever it is faster (decide_layout_f)
    int f_chunk_len = 0, f_chunks = 0, f_wpg = 0;
    size_t f_lds = 0;
    double *dctab_f = nullptr;   // Phi^(S..4S) | Psi^(S..4S) for layout F's chunk length
    int dctab_f_len = 0;
    double *dftab = nullptr;     // layout F: the chunks' input tables T_s | aff for f_chunk_len (k_build_f_input_tables)
    int dftab_cap = 0;           // (slots it was allocated for)
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
    int fam_lin_cap = 0, fam_cone_cap = 0, fam_round_cap = 0;  // capacities `dfam` is laid out for (fam_doubles(), tinympc_device.h)
    size_t fam_alloc_doubles = 0;
    double *h_x0 = nullptr, *h_u0 = nullptr;  // pinned staging for tinympc_mpc_step_batch (and x0 of single-instance handles)
    // Single-instance handles (batch == 1, what the MEX shim creates): set_x0 only fills the pinned h_x0, the next
    // launch reads it from there (and mirrors it into dx0), and the kernels also write solution + statistics into the
    // pinned h_sol -- the reference's per-tick sequence set_x0 / solve / get_solution then costs ONE launch and ONE
    // synchronisation instead of three synchronous copies around the launch.
    double *h_sol = nullptr;           // [X | U | 4 residuals | iter, status | completion flag]
    // Closed-loop session (tinympc_session_begin / _step / _end): the latency kernel stays resident and takes its ticks
    // from this mailbox in pinned memory (layout: SolveParams::mail).
    double *h_mail = nullptr;          // [64] in pinned host memory (the kernel polls across PCIe) ...
    double *h_ans = nullptr;           // [32] pinned: the early answer lines of layout F's resident kernel (SolveParams::host_ans)
    double *d_mail = nullptr;          // ... or, where the host can store into device memory (large BAR), in fine-grained device memory
    double *mailbox() const { return d_mail ? d_mail : h_mail; }  // what the session uses (tinympc_handle.hip: acquire_arenas)
    bool session_active = false;
    bool resident_solves = false;  // tinympc_set_resident: tinympc_solve goes through the resident session kernel where one exists
    bool resident_refused = false; // ... and none does for this configuration (asked once)
    bool session_on_f = false;  // ... and its resident kernel is layout F's (families beyond what the latency kernel's session holds)
    // Taken by everything that writes the mailbox or (re)starts the resident kernel: session_step (for the whole tick), end_session
    // and park_sessions_on_device -- the one place where a thread reaches into a handle it does not own. Lock order: the session
    // registry (tinympc_session.hip) first, then this.
    std::mutex session_mu;
    // references re-sent inside a session that turned out to be the previous ones moved up by one knot (receding horizon):
    // only the new last column travels, with the command (flags 4 / 8); two shifts without a step in between, or any other
    // change, fall back to the full re-read (refs_on_host)
    bool xref_shift = false, uref_shift = false;
    bool session_refs_shifted = false;  // the device copies / tables lag behind the pinned references
    // ONE counter stamps session commands and flag-raising launches alike (both complete by writing their stamp into the
    // same slot of h_sol: a launch after a session must not find its number already there)
    unsigned long long session_seq = 0;  // stamp of the last session command / flag-raising launch
    unsigned long long answered_seq = 0;  // ... of the session tick whose early answer was taken last (its write-out is what host_sol_state 3 waits for)
    bool flag_pending = false;
    // ... and set_x_ref / set_u_ref only fill these pinned copies; the next launch's workgroup rebuilds the
    // reference-dependent table rows from them (refresh_reference_tables): a tick with per-tick references
    // (rocket_landing_constraints.m:86-121) is still one launch and one synchronisation.
    double *h_xref = nullptr, *h_uref = nullptr;
    bool refs_on_host = false;         // the pinned references are newer than dXref / dUref and the tables
    bool x0_on_host = false;           // h_x0 is newer than dx0
    int host_sol_state = 0;            // 0: not valid, 1: a launch that writes it is in flight, 2: valid, 3: a session tick has answered with
                                       //    its first controls; solution + statistics are valid once the stamp behind them reads session_seq
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
    void *arena_mail = nullptr;                       // ... and the 4 KB of fine-grained device memory that travel with them (the session's mailbox, d_mail)
    void *arena_dev = nullptr, *arena_pin = nullptr;  // the setup arenas (ArenaPlan below; pooled per device, tinympc_handle.hip)
    size_t arena_dev_bytes = 0, arena_pin_bytes = 0;
    std::vector<void *> allocs;       // hipMalloc blocks dalloc() added after setup
    std::vector<void *> host_allocs;  // hipHostMalloc blocks added after setup (staging of mpc_step_batch on batched handles)
    double *h_stage = nullptr;        // pinned staging of the problem data (setup's one upload); layout = the device arena's upload block
    double setup_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // tinympc_debug_setup_timing: host microseconds of setup's phases

    size_t X() const { return (size_t)nx * N; }
    size_t U() const { return (size_t)nu * (N - 1); }
    size_t state_doubles() const { return layout_m ? tinympc::solve_m_state_doubles(nx, nu, N, groups) : (size_t)groups * (N + 1) * 64; }  // G; row N: per-lane dummy slot
    size_t v_doubles() const { return layout_m ? tinympc::solve_m_state_doubles(nx, nu, N, groups) : (size_t)groups * tinympc::v_rows(N) * 64; }     // V (and V2)
    size_t d_doubles() const { return layout_m ? tinympc::solve_m_state_doubles(nx, nu, N, groups) : (size_t)groups * (N - 1) * IPW * nu; }
};

namespace tinympc {
namespace host {

void park_sessions_on_device(int device, const tinympc_solver *except);  // (tinympc_session.hip; described below)

// One block of device memory and one block of pinned host memory per handle (round 5): every array whose size is known at setup is
// carved out of them -- tinympc_setup_batch used to make ~45 hipMalloc and 5 hipHostMalloc calls (5.9 ms for a quadrotor, twelve
// times the reference's whole tiny_setup). ArenaPlan collects (slot, offset) pairs in a first pass, the block is allocated once,
// bind() writes the pointers. 256-byte alignment: every array starts on its own pair of 128-byte lines.
struct ArenaPlan {
    struct Item { void **slot; size_t offset; };
    std::vector<Item> items;
    size_t total = 0;
    size_t mark() { total = (total + 255) & ~(size_t)255; return total; }
    template <typename T>
    void add(T **slot, size_t count) {
        mark();
        items.push_back({reinterpret_cast<void **>(slot), total});
        total += sizeof(T) * (count ? count : 1);
    }
    void bind(void *base) const {
        for (const Item &it : items) *it.slot = static_cast<char *>(base) + it.offset;
    }
};

// Arrays whose size depends on what the verbs receive AFTER setup (the families' description and duals, a longer layout-F chunk,
// the HBM working copy of a long horizon on layout M): their own hipMalloc, at the launch that first needs them. hipMalloc
// synchronises the device, so resident session kernels of OTHER handles are sent home first (as setup and destroy do) -- it would
// otherwise stall until their 2 s idle time-out. Not from inside this handle's own session tick (session_mu is held there and the
// registry lock comes first in the lock order): a session's kernel was launched once by session_begin, everything it needs exists.
template <typename T>
int dalloc(tinympc_solver *s, T **p, size_t count) {
    if (!s->session_active) park_sessions_on_device(s->device, s);
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) return fail(TINYMPC_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", sizeof(T) * count, hipGetErrorString(e));
    s->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return TINYMPC_OK;
}

// ---- tinympc_handle.hip
int acquire_arenas(tinympc_solver *s, size_t dev_bytes, size_t pin_bytes, bool want_mailbox);  // -> s->arena_dev / arena_pin (from the device's pool when one fits)
int acquire_stream_kit(tinympc_solver *s);  // stream + event pair from the device's pool (created when the pool is empty)
int bind_device(tinympc_solver *s);   // every verb that touches the device passes through here first (ends an open session)
bool rows_constant(const double *m, int rows, int cols);
int upload(tinympc_solver *s, double *dst, const double *src, size_t count);
int download(tinympc_solver *s, void *dst, const void *src, size_t bytes);
int fill_host_upload(tinympc_solver *s, double *dst, size_t count, double value);
int check_handle(const tinympc_solver *s);
int run_precompute(tinympc_solver *s);
int flush_host_refs(tinympc_solver *s);
int refresh_derived(tinympc_solver *s);
FamilyStructure family_structure(const tinympc_solver *s, double *mu = nullptr);
int refresh_families(tinympc_solver *s);
void destroy(tinympc_solver *s);

// ---- tinympc_plan.hip: the kernel of a launch, decided in ONE place
enum class KernelId { M, D_COMPILED, D_JIT, E, F, C, FAM_A, ADAPT_A, B, A };
struct LaunchPlan {
    KernelId kernel = KernelId::A;
    char layout = 'A';                       // what tinympc_get_layout reports
    bool families = false, adaptive = false;  // variant bits of the launch
    bool jit = false;                        // a run-time specialisation (tinympc_jit.hip) rather than a compiled-in kernel
    bool host_exchange = false;              // the kernel serves the pinned-host paths (x0 in, solution / completion stamp out)
    int workgroups = 0;
    size_t lds_bytes = 0;
    bool tables_in_lds = false;
};
LaunchPlan current_plan(const tinympc_solver *s);  // from what has been decided so far (compiles nothing)
int resolve_plan(tinympc_solver *s);               // decide (and, where needed, specialise) the variants of the current configuration
int launch(tinympc_solver *s, bool timed);

// ---- tinympc_session.hip
int end_session(tinympc_solver *s);
int wait_session_solution(tinympc_solver *s);  // host_sol_state 3 -> 2: spins on the completion stamp of the last session tick
// Resident session kernels of OTHER handles on `device` are sent home before anything that synchronises the device (hipMalloc /
// hipFree in setup and teardown): such a call would otherwise stall until the spinning kernel's idle time-out (2 s). Their
// sessions stay open: the next session_step finds the kernel gone and starts it again (its restart path).
void park_sessions_on_device(int device, const tinympc_solver *except);
// Writes the zeros of a pending cold start into G, V, D (for a kernel that loads its state from HBM whatever its value).
int materialize_cold_state(tinympc_solver *s);
// ... and of a pending zero solution into sol_x / sol_u (before anything reads them that no solve has written since the reset).
int materialize_zero_solution(tinympc_solver *s);

}  // namespace host
}  // namespace tinympc
