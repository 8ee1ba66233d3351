// tinympc_session.hip -- the closed-loop SESSION (tinympc_session_begin / _step / _end, include/tinympc_hip.h): a latency kernel -- layout
// F's resident variant where the handle's launches run on layout F, else layout C's -- stays resident and takes its ticks from a MAILBOX
// instead of being launched per tick. Host side only: the mailbox protocol, the (re)start of the resident kernel, the verbs.
//   command   host -> kernel: lines [7 payload doubles | stamp], stamp = mail_stamp(sequence number, payload) -- a line is taken when the stamp
//             fits the payload read with it. Round 5: the mailbox is a line of fine-grained DEVICE memory the host stores into through the
//             PCIe BAR (the kernel polls its own HBM: 0.2 us per poll instead of a 1.2 us PCIe read), pinned host memory where the device is
//             not large-BAR (tinympc_handle.hip: acquire_arenas).
//   answer    kernel -> host, pinned memory: FIRST the tick's first controls in self-checking lines [7 controls | stamp] (SolveParams::host_ans:
//             no fence, no wait for the rest), THEN solution + statistics behind a completion stamp (host_sol_state 3 until it is seen);
//             every host-bound store of the kernels is a write-through store (host_store, tinympc_device.h).
// tinympc_set_resident (tinympc_capi.hip) puts the reference's own verbs set_x0 / solve / get_solution on this path.
#include "tinympc_handle.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>
#include <chrono>
#include <cstring>
#include <immintrin.h>

using namespace tinympc;
using namespace tinympc::host;

namespace {

// ---- closed-loop session ------------------------------------------------------------------------------------
constexpr double kSessionIdleSeconds = 2.0;  // the resident kernel leaves on its own after this long without a command

void write_command(tinympc_solver *s, int flags, const double *x0) {
    // payload: 0 flags | x0 (nx) | new last column of x_ref (flag 4) | new last column of u_ref (flag 8); line l = [7 payload |
    // stamp]. Payload before stamp, line by line (x86 keeps the order of stores; the fences keep the compiler from
    // reordering them); the stamp also carries a checksum of its line's payload, so that a reader which caught the line
    // between two of these stores -- or torn in any other way -- rejects it and polls again.
    double pay[56];
    int npay = 0;
    pay[npay++] = (double)flags;
    for (int i = 0; i < s->nx; ++i) pay[npay++] = x0 ? x0[i] : 0.0;
    if (flags & 4) for (int i = 0; i < s->nx; ++i) pay[npay++] = s->h_xref[(size_t)(s->N - 1) * s->nx + i];
    if (flags & 8) for (int i = 0; i < s->nu; ++i) pay[npay++] = s->h_uref[(size_t)(s->N - 2) * s->nu + i];
    volatile double *m = s->mailbox();  // (device memory through the BAR: stores only -- the host never reads the mailbox)
    const double seq = (double)(++s->session_seq);
    const int nlines = (npay + 6) / 7;
    for (int q = npay; q < 7 * nlines; ++q) pay[q] = 0.0;  // (every word of a used line is written: the stamp covers all seven)
    for (int l = 0; l < nlines; ++l) {
        for (int q = 7 * l; q < 7 * l + 7; ++q) m[8 * l + q % 7] = pay[q];
        std::atomic_thread_fence(std::memory_order_release);
        m[8 * l + 7] = tinympc::mail_stamp_of(seq, pay + 7 * l);  // (sequence number + checksum of the line's payload, see tinympc_device.h)
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    _mm_sfence();  // (a mailbox behind the BAR is write-combining memory: push the lines out now, not when the buffer is evicted)
}

int launch_session_kernel(tinympc_solver *s) {
    int rc;
    // (layout F's kernel has no staging of references left in pinned memory in its prologue: the ordinary upload first)
    // (... also on a restart after the idle time-out: reference shifts of the session so far only reached the LDS tables of the
    // kernel that has left; the pinned copies are current)
    if (s->session_on_f && (s->refs_on_host || s->session_refs_shifted || s->xref_shift || s->uref_shift)) {
        if ((rc = flush_host_refs(s))) return rc;
        s->session_refs_shifted = false;
    }
    if ((rc = refresh_derived(s))) return rc;
    const bool fam = s->families_active();
    if (fam && (rc = refresh_families(s))) return rc;
    if ((rc = materialize_cold_state(s))) return rc;  // (the resident kernel loads its state from HBM)
    if ((rc = materialize_zero_solution(s))) return rc;
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
    p.mail = s->mailbox();
    p.host_ans = s->h_ans;  // (both resident kernels answer early, see SolveParams::host_ans)
    // (no line of an earlier session of this handle may look like an answer of this one -- its sequence numbers go on, so none would;
    // setup zeroes the pinned arena, pooled or new; this costs nothing and leaves nothing to those two arguments)
    if (s->h_ans) for (int i = 0; i < 32; ++i) s->h_ans[i] = -1.0;
    if (s->h_sol) s->h_sol[s->X() + s->U() + 6] = -1.0;  // (the completion stamp of solution + statistics, likewise)
    std::atomic_thread_fence(std::memory_order_seq_cst);
    p.session_expect = (double)(s->session_seq + 1);
    p.session_idle = (unsigned long long)(kSessionIdleSeconds * 1e8);
    if (s->session_on_f) {
        p.const_tables = 0;  // (the session kernel always carries per-knot tables: references may change from tick to tick)
        p.ctab = s->dctab_f; p.ftab = s->dftab; p.chunk_len = s->f_chunk_len; p.chunk_count = s->f_chunks; p.chunk_levels = 4;
        HIP_TRY(launch_solve_f_session(p, s->f_fs, s->tables_const(), s->stream));
        return TINYMPC_OK;
    }
    HIP_TRY(launch_solve_c(p, s->W, s->KT, s->lds_bytes_c, s->stream));
    return TINYMPC_OK;
}

}  // namespace

namespace {
std::mutex g_sessions_mu;
std::vector<tinympc_solver *> g_sessions;  // handles whose session is open
void session_registry(tinympc_solver *s, bool open) {
    std::lock_guard<std::mutex> lock(g_sessions_mu);
    auto it = std::find(g_sessions.begin(), g_sessions.end(), s);
    if (open && it == g_sessions.end()) g_sessions.push_back(s);
    if (!open && it != g_sessions.end()) g_sessions.erase(it);
}
}  // namespace

void tinympc::host::park_sessions_on_device(int device, const tinympc_solver *except) {
    // The registry lock is held for the whole walk: a handle leaves the registry (end_session, first thing; destroy() passes through
    // it) before anything of it is torn down, so none of the pointers can die under this loop. Each handle's own session mutex keeps
    // the stop command out of a tick its owner thread is in the middle of (the stamp and the mailbox lines of that tick would
    // be overwritten and the owner would spin on a stamp nobody answers, until the restart path or the idle time-out).
    std::lock_guard<std::mutex> lock(g_sessions_mu);
    for (tinympc_solver *o : g_sessions) {
        if (o == except || o->device != device) continue;
        std::lock_guard<std::mutex> tick(o->session_mu);
        if (!o->session_active) continue;
        write_command(o, 1, nullptr);                 // stop: the kernel writes its state back and leaves
        (void)hipStreamSynchronize(o->stream);        // (session_active stays set: session_step restarts the kernel)
        if (o->host_sol_state == 3) o->host_sol_state = 2;  // (the kernel has left: its last tick's write-out is complete)
    }
}

int tinympc::host::wait_session_solution(tinympc_solver *s) {
    if (s->host_sol_state != 3) return TINYMPC_OK;
    // Under the handle's session mutex: another thread's setup may be parking this kernel right now (park_sessions_on_device) -- its stop
    // command takes the next sequence number, and a reader that looked at session_seq instead of the answered tick's own number waited
    // for a stamp nobody writes (tools/thread_stress.py, round 5: "not completed ... within 5 s"). The park completes the state itself.
    std::lock_guard<std::mutex> tick(s->session_mu);
    if (s->host_sol_state != 3) return TINYMPC_OK;
    const volatile double *done = s->h_sol + s->X() + s->U() + 6;
    const double want = (double)s->answered_seq;
    const auto t_start = std::chrono::steady_clock::now();
    for (long spin = 0; *done != want; ++spin) {
        __builtin_ia32_pause();
        if ((spin & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(5))
            return fail(TINYMPC_ERR_HIP, "the resident kernel has not completed the write-out of its last tick within 5 s");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    s->host_sol_state = 2;
    return TINYMPC_OK;
}

int tinympc::host::end_session(tinympc_solver *s) {
    if (!s->session_active) return TINYMPC_OK;
    session_registry(s, false);  // (registry first, then the handle's mutex: the lock order of park_sessions_on_device)
    std::lock_guard<std::mutex> tick(s->session_mu);
    if (!s->session_active) return TINYMPC_OK;
    write_command(s, 1, nullptr);  // stop
    s->session_active = false;     // (before anything that could come back here)
    s->session_on_f = false;
    if (s->session_refs_shifted || s->xref_shift || s->uref_shift) s->refs_on_host = true;  // device copies / tables lag: restage
    s->session_refs_shifted = s->xref_shift = s->uref_shift = false;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->host_sol_state == 3) s->host_sol_state = 2;  // (the kernel has left: its last tick's write-out is complete)
    return TINYMPC_OK;
}

extern "C" {

int tinympc_session_begin(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;  // (ends a session that is still open)
    if (s->st.adaptive_rho) return fail(TINYMPC_ERR_UNSUPPORTED, "session: adaptive_rho is not supported");
    if (s->st.max_iter < 1) return fail(TINYMPC_ERR_INVALID_INPUT, "session: max_iter must be >= 1");
    // Which resident kernel: layout F's (round 4) where the handle's launches run on layout F -- then the session's ticks are bit-identical
    // to launched ticks --, else the latency kernel's SESSION variant (layout C: box path, families up to N = 65 with disjoint cones).
    const bool fam = s->families_active();
    const bool c_ok = s->host_path() && (s->layout_c || (fam && s->fam_c)) && !(fam && s->chunk_len > 4) &&
                      !(fam && (family_structure(s).nround > 1 || family_structure(s).beyond_generic()));
    s->session_on_f = false;
    // (round 4) where the handle's launches already run on layout F -- the families by default, the box path compiled in or asked for
    // with tinympc_prepare() -- its resident variant is the one whose ticks are bit-identical to those launches (and the faster one):
    // taken if it is there for the asking (the families specialise at run time anyway; the box path: compiled in / prepare())
    bool f_first = false;
    if (c_ok && s->host_path()) {
        if ((rc = resolve_plan(s))) return rc;
        f_first = current_plan(s).kernel == KernelId::F &&
                  (fam || s->specialise_asked || solve_f_builtin(s->nx, s->nu, s->N, false, false, FamilyStructure(), true)) &&
                  solve_f_session_supported(s->nx, s->nu, s->N, s->tables_const(), fam, s->f_fs);
    }
    if (f_first) s->session_on_f = true;
    else if (!c_ok) {
        if (!s->host_path()) return fail(TINYMPC_ERR_UNSUPPORTED, "session: single-instance handles only (batch 1, nx+nu <= 16)");
        if ((rc = resolve_plan(s))) return rc;
        if (current_plan(s).kernel != KernelId::F || !solve_f_session_supported(s->nx, s->nu, s->N, s->tables_const(), fam, s->f_fs))
            return fail(TINYMPC_ERR_UNSUPPORTED, "session: neither the latency kernel (box path; families up to N = 65, disjoint cones) nor layout F "
                        "(run-time specialised; TINYMPC_JIT not 0) has a resident kernel for this configuration (N = %d)", s->N);
        s->session_on_f = true;
    }
    // (the mailbox h_mail is part of a single-instance handle's pinned arena: tinympc_setup_batch)
    HIP_TRY(hipStreamSynchronize(s->stream));
    {
        std::lock_guard<std::mutex> tick(s->session_mu);
        // (likewise the mailbox: no stamp of an earlier session's last command survives into this one)
        volatile double *m = s->mailbox();
        for (int l = 0; l < 7; ++l) m[8 * l + 7] = -1.0;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        _mm_sfence();
        if ((rc = launch_session_kernel(s))) return rc;
        s->session_active = true;
        s->flag_pending = false;
    }
    session_registry(s, true);
    return TINYMPC_OK;
}

// The session is over without the kernel's orderly exit (restart failed, stream error, no answer): the handle goes back to ordinary
// launches, whose device tables may lag behind the pinned references -- after shifts that only reached the resident kernel's LDS
// tables, or a full re-read (`pending_flags` & 2) the kernel never consumed. Restage them at the next launch.
static void mark_session_dead(tinympc_solver *s, int pending_flags) {
    s->session_active = false;
    if (s->host_sol_state == 3) s->host_sol_state = 0;  // (whatever the kernel completed is in the device copies)
    s->session_on_f = false;
    if ((pending_flags & 14) || s->session_refs_shifted || s->xref_shift || s->uref_shift) s->refs_on_host = true;
    s->session_refs_shifted = s->xref_shift = s->uref_shift = false;
}

// One tick under the handle's session mutex. `*dead` is set when the session ended with an error: the caller then takes the
// handle out of the registry AFTER the mutex is released (lock order: registry before handle).
static int session_tick_locked(tinympc_solver *s, const double *x0, double *u0_out, bool *dead) {
    int rc;
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
    // The answer: the resident kernels send the first controls ahead, in lines [7 controls | mail_stamp(seq, controls)] that are complete
    // when the stamp fits the payload read with it (SolveParams::host_ans); solution and statistics follow under the completion stamp.
    const bool early = s->h_ans != nullptr;
    const int nlines = (s->nu + 6) / 7;
    double u0_lines[24];
    // Reading order: the STAMP first. The kernel stores a line with one 64-byte store; once the stamp shows this tick's sequence number
    // the payload read behind it is this tick's (x86 does not reorder loads). Payload first and stamp last -- the order of rounds 4-5 --
    // left a window of a few nanoseconds in which the answer could land between the two: old payload, new stamp, and only the checksum to
    // tell (which, as a folded XOR, four saturated controls shared with a line of zeros). The checksum and a second look at the stamp
    // stay, for a line that should ever arrive in pieces.
    auto answered = [&]() -> bool {
        if (!early) return *done == want;
        const volatile double *a = s->h_ans;
        for (int l = 0; l < nlines; ++l) {
            const double stamp = a[8 * l + 7];
            if (!(stamp >= want && stamp < want + 1.0)) return false;
            for (int q = 0; q < 7; ++q) u0_lines[7 * l + q] = a[8 * l + q];
            if (stamp != tinympc::mail_stamp_of(want, u0_lines + 7 * l) || a[8 * l + 7] != stamp) return false;
        }
        return true;
    };
    const auto t_start = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        if (answered()) break;
        __builtin_ia32_pause();
        if ((spin & 0xffff) == 0xffff) {
            // Nothing for a while: has the kernel left (idle time-out, or parked by another handle's setup)? Then start it again; it
            // waits for exactly the command that is pending. A stream error or 30 s without an answer end the session with an error.
            const hipError_t q = hipStreamQuery(s->stream);
            if (q == hipSuccess) {
                // The command is issued again under a NEW stamp and without reference flags -- the old one, still in the mailbox,
                // must not be taken. What it asked for is handed to the start of the new kernel instead: a full re-read the kernel
                // that left never consumed (flags 2) is pending again -- layout C's prologue stages the pinned references, layout
                // F's kernel has no staging, launch_session_kernel uploads them the ordinary way first -- and shifts (flags 4 / 8)
                // count as "the device copies lag" (session_refs_shifted, set above).
                if (flags & 2) s->refs_on_host = true;
                rc = launch_session_kernel(s);  // waits for session_seq + 1
                if (rc) { mark_session_dead(s, flags); *dead = true; return rc; }
                flags = 0;
                write_command(s, 0, x0);
                want = (double)s->session_seq;
            } else if (q != hipErrorNotReady) {
                mark_session_dead(s, flags);
                *dead = true;
                return fail(TINYMPC_ERR_HIP, "session_step: the handle's stream reports %s", hipGetErrorString(q));
            }
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(30)) {
                write_command(s, 1, nullptr);  // stop (should the kernel still be there); the caller waits for the stream
                mark_session_dead(s, flags);
                *dead = true;
                return fail(TINYMPC_ERR_HIP, "session_step: no answer from the resident kernel within 30 s");
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (early) {
        s->answered_seq = s->session_seq;
        s->host_sol_state = 3;  // (solution + statistics: valid once the stamp behind them reads answered_seq -- wait_session_solution)
        std::memcpy(u0_out, u0_lines, sizeof(double) * s->nu);
    } else {
        s->host_sol_state = 2;
        std::memcpy(u0_out, s->h_sol + s->X(), sizeof(double) * s->nu);
    }
    return TINYMPC_OK;
}

int tinympc_session_step(tinympc_solver *s, const double *x0, double *u0_out) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!x0 || !u0_out) return fail(TINYMPC_ERR_INVALID_INPUT, "session_step: x0 and u0_out are required");
    bool dead = false;
    {
        std::lock_guard<std::mutex> tick(s->session_mu);
        rc = session_tick_locked(s, x0, u0_out, &dead);
    }
    if (dead) {
        session_registry(s, false);
        (void)hipStreamSynchronize(s->stream);
    }
    return rc;
}

int tinympc_session_end(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    return end_session(s);
}

}  // extern "C"
