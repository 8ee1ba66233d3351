// tinympc_plan.hip -- WHICH kernel a launch runs, and the launch. Nine kernel families (tinympc_device.h) x (compiled in | run-time
// specialised) x (constant | per-knot tables) x (box path | cone / linear families | adaptive rho): the choice is made in ONE
// function, current_plan(), from the handle's capabilities (what setup found possible for the shape) and the variants decided
// for the CURRENT configuration (resolve_plan(): may specialise a kernel, seconds the first time). launch(), the launch geometry
// of tinympc_get_launch_info, tinympc_get_layout and tinympc_get_jit_info all read that one LaunchPlan.
#include <cstring>
#include "tinympc_handle.h"

using namespace tinympc;
using namespace tinympc::host;

namespace {

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
        const FamilyStructure fsd = family_structure(s);
        // Round 4: at these horizons layout E has an UNCUT form (one wavefront per group, the families one knot per lane) that does in
        // ~200 instructions per iteration what this variant pays in every slot -- it takes precedence where it builds (TINYMPC_LAYOUT=D
        // keeps the variant for the tests and A/B runs).
        bool e_uncut = false;
        const char *env = getenv("TINYMPC_LAYOUT");
        if (s->W == 16 && s->batch > 1 && s->N - 1 <= 31 && !(env && (env[0] == 'D' || env[0] == 'd'))) {
            int wpg = 0;
            e_uncut = solve_e_plan(s->nx, s->nu, s->N, s->tables_const(), true, fsd, nullptr, &wpg, nullptr, nullptr) && wpg == 1 &&
                      solve_e_supported(s->nx, s->nu, s->N, s->tables_const(), true, fsd);
        }
        if (s->W == 16)
            s->d_fam = (!e_uncut && fsd.nround <= 1 && !fsd.beyond_generic() && solve_jit_supported(s->W, s->nx, s->nu, s->N, s->tables_const(), true)) ? 1 : 0;
        else  // (round 5) wide systems: the families streamed from HBM next to the register-resident box sweeps; rounds are walked there
            s->d_fam = (!s->layout_m && !fsd.beyond_generic() && fsd.nround <= MAX_ROUNDS && solve_jit_supported(s->W, s->nx, s->nu, s->N, s->tables_const(), true)) ? 1 : 0;
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
    sig += solve_jit_enabled() ? "|jit" : "|nojit";  // (TINYMPC_JIT=0 switches the specialised kernels off from the next launch on)
    if (sig == s->e_sig) return TINYMPC_OK;
    s->e_sig = sig;
    s->fs = fs;
    s->e_ok = solve_e_supported(s->nx, s->nu, s->N, s->tables_const(), fam, fs) &&
              solve_e_plan(s->nx, s->nu, s->N, s->tables_const(), fam, fs, &s->e_chunk_len, &s->e_wpg, &s->e_lds, &s->e_gpw);
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
    const bool possible = s->W == 16 && !s->layout_m && !s->st.adaptive_rho && (!s->session_active || s->session_on_f) && s->N >= 6;
    // Default: the families at small batches -- rocket landing N=100, one instance: 4.5 us per iteration against 6.55 on the
    // round-1 latency kernel
    // (a configuration beyond what the generic kernels hold -- more than MAX_LIN_ROWS rows, MAX_CONES cones, MAX_ROUNDS rounds -- has
    // no other kernel at these batch sizes)
    bool want = possible && fam && !s->use_layout_d() && !s->use_layout_e() && s->batch < kLayoutEBatchMin &&
                (s->fam_c || family_structure(s).beyond_generic());
    // ... and (round 4) the box path wherever the latency kernel would run: layout F is ahead on every shape measured
    // (tools/single_cf_probe.py, microseconds per iteration, one instance: quadrotor N=10 / 20 / 50 / 100 1.81 / 2.24 / 2.73 / 4.07
    // against 2.45 / 2.63 / 2.90 / 4.33, cartpole N=20 1.60 against 2.13) and has had the resident session since this round. It is
    // a specialisation, though: a first setup of a new shape must not cost seconds behind the caller's back, so it is taken where it
    // costs nothing -- the configurations compiled into the library (BASELINE configs 2 and 3) -- or where the caller asked for the
    // specialised kernels with tinympc_prepare() (the real-time workflow of INTEGRATION.md); otherwise layout C, which needs none.
    bool box_f = possible && !fam && s->layout_c && !s->use_layout_d() && !s->use_layout_e() && solve_jit_enabled();
    if (box_f && !s->specialise_asked) {  // (asked once per handle and kind of tables: this runs in front of every launch)
        signed char &known = s->f_box_builtin[s->tables_const() ? 1 : 0];
        if (known < 0) known = solve_f_builtin(s->nx, s->nu, s->N, s->tables_const(), false, FamilyStructure(), false) ? 1 : 0;
        box_f = known == 1;
    }
    want = want || box_f;
    if (const char *env = getenv("TINYMPC_LAYOUT")) want = (env[0] == 'F' || env[0] == 'f') && possible;
    if (!want) {
        s->f_ok = false;
        s->f_sig.clear();
        s->f_box_key = 0;
        return TINYMPC_OK;
    }
    // (the box path decides in front of every closed-loop tick: a small integer instead of the signature string below)
    const unsigned box_key = fam ? 0u : (8u | (s->tables_const() ? 1u : 0u) | (solve_jit_enabled() ? 2u : 0u));
    if (!fam && box_key == s->f_box_key && !s->f_sig.empty()) return TINYMPC_OK;
    s->f_box_key = box_key;
    const FamilyStructure fs = fam ? family_structure(s) : FamilyStructure();
    std::string sig = s->tables_const() ? "ct|" : "var|";
    for (int c = 0; c < fs.ncone; ++c) sig += std::to_string(fs.cone[c][0]) + "," + std::to_string(fs.cone[c][1]) + "," + std::to_string(fs.cone[c][2]) + ";";
    sig += "|" + std::to_string(fs.nlx) + "," + std::to_string(fs.nlu) + (fam ? "|fam" : "|box");
    sig += solve_jit_enabled() ? "|jit" : "|nojit";
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

// (per-launch pieces of the plan's run-time specialised kernels)
bool d_is_jit(const tinympc_solver *s, bool fam, bool adaptive) {
    return s->d_jit || fam || adaptive || (!s->tables_const() && s->d_varying_jit);
}

}  // namespace

namespace tinympc {
namespace host {

// The one place where the kernel of a launch is chosen. Order of preference per variant:
//   large systems                      M
//   adaptive rho                       D (run-time specialised, rho per lane)  >  k_admm_solve_adapt on layout A's plan
//   cone / linear families             D (N <= 22)  >  F (small batches)  >  E (batches)  >  C<FAM> (disjoint cones only)  >  k_admm_solve_fam
//   box path                           E (only where D has no kernel)  >  D  >  F (on request)  >  C (small batches)  >  B  >  A
LaunchPlan current_plan(const tinympc_solver *s) {
    LaunchPlan pl;
    const bool fam = s->families_active(), adaptive = s->st.adaptive_rho != 0, d = s->use_layout_d();
    pl.families = fam;
    pl.adaptive = adaptive;
    if (s->layout_m) pl.kernel = KernelId::M;
    else if (adaptive) pl.kernel = d ? KernelId::D_JIT : KernelId::ADAPT_A;
    else if (fam)
        pl.kernel = d ? KernelId::D_JIT : s->use_layout_f() ? KernelId::F : s->use_layout_e() ? KernelId::E
                    : (s->fam_c && family_structure(s).nround <= 1) ? KernelId::C : KernelId::FAM_A;
    else
        pl.kernel = s->use_layout_e() ? KernelId::E : d ? (d_is_jit(s, false, false) ? KernelId::D_JIT : KernelId::D_COMPILED)
                    : s->use_layout_f() ? KernelId::F : s->layout_c ? KernelId::C : s->layout_b ? KernelId::B : KernelId::A;
    const bool ct = s->tables_const();
    switch (pl.kernel) {
        case KernelId::M:
            pl.layout = 'M';
            pl.workgroups = s->groups;
            pl.lds_bytes = 0;  // (static LDS: see the kernel)
            break;
        case KernelId::D_JIT:
            pl.layout = 'D';
            pl.jit = true;
            pl.workgroups = solve_jit_workgroups(s->W, s->nx, s->nu, s->N, ct, s->groups, fam, adaptive);
            pl.lds_bytes = solve_jit_lds_bytes(s->W, s->nx, s->nu, s->N, ct, fam, adaptive);
            break;
        case KernelId::D_COMPILED:
            pl.layout = 'D';
            pl.host_exchange = s->W == 16;  // (the compiled-in 16-lane shapes have a variant that reads x0 from / writes the first controls to pinned host memory)
            pl.workgroups = s->W == 64 ? solve_dx_workgroups(s->nu, s->N, s->groups) : s->W == 32 ? solve_dw_workgroups(s->nu, s->N, s->groups)
                                                                                                   : solve_d_workgroups(s->nu, s->N, ct, s->groups);
            pl.lds_bytes = s->W == 64 ? solve_dx_lds_bytes(s->nu, s->N) : s->W == 32 ? solve_dw_lds_bytes(s->nu, s->N) : solve_d_lds_bytes(s->nu, s->N, ct);
            break;
        case KernelId::E:
            pl.layout = 'E';
            pl.jit = true;
            pl.workgroups = (s->groups + s->e_gpw - 1) / s->e_gpw;
            pl.lds_bytes = s->e_lds;
            break;
        case KernelId::F:
            pl.layout = 'F';
            pl.jit = true;
            pl.host_exchange = true;
            pl.workgroups = s->batch;
            pl.lds_bytes = s->f_lds;
            break;
        case KernelId::C:
            pl.layout = 'C';
            pl.host_exchange = true;
            pl.workgroups = s->batch;
            pl.lds_bytes = s->lds_bytes_c;
            break;
        case KernelId::FAM_A:
        case KernelId::ADAPT_A:  // kernels of their own on layout A's plan
            pl.layout = 'A';
            pl.host_exchange = true;
            pl.workgroups = s->groups;
            pl.lds_bytes = s->lds_bytes_a;
            pl.tables_in_lds = s->tables_in_lds_a;
            break;
        case KernelId::B:
            pl.layout = 'B';
            pl.host_exchange = true;
            pl.workgroups = (s->groups + WAVES_PER_GROUP_B - 1) / WAVES_PER_GROUP_B;
            pl.lds_bytes = s->lds_bytes;
            pl.tables_in_lds = s->tables_in_lds;
            break;
        case KernelId::A:
            pl.layout = 'A';
            pl.host_exchange = true;
            pl.workgroups = s->groups;
            pl.lds_bytes = s->lds_bytes;
            pl.tables_in_lds = s->tables_in_lds;
            break;
    }
    return pl;
}

int resolve_plan(tinympc_solver *s) {
    int rc;
    decide_layout_d_variants(s);
    if ((rc = decide_layout_e(s))) return rc;
    return decide_layout_f(s);
}

// Slot refill (tinympc_solve_d.hip, REFILL): layout D's 16-lane kernels (box path), when the batch is more than the device holds at
// once and the tolerances can be met -- i.e. when instances finish at different times and there is something to refill a row
// with. TINYMPC_REFILL=0 switches it off, =1 takes it for any batch beyond one resident set whatever the tolerances.
static bool refill_applies(const tinympc_solver *s, const LaunchPlan &pl) {
    const bool jit = pl.kernel == KernelId::D_JIT;
    if ((pl.kernel != KernelId::D_COMPILED && !jit) || s->W != 16 || pl.adaptive || pl.families || s->zero_copy_tick || s->st.max_iter <= 0 ||
        s->st.check_termination <= 0)
        return false;
    const char *env = getenv("TINYMPC_REFILL");
    const int mode = env ? atoi(env) : -1;
    if (mode == 0) return false;
    const bool ct = s->tables_const();
    long resident;  // wavefronts = groups of four instances
    if (jit) resident = solve_jit_resident_wavefronts(s->W, s->nx, s->nu, s->N, ct);
    else {
        const int wpg = solve_d_wavefronts_per_workgroup(s->nu, s->N, ct);
        resident = (long)solve_d_resident_workgroups(wpg) * wpg;
    }
    // Tolerances nothing can meet (forced iteration counts): every instance runs max_iter, there is nothing to balance, and the
    // plain kernel's sweeps are the faster ones (65,536 x 200 iterations: 13.3 against 13.7 ms) -- keep it.
    const bool reachable = s->st.abs_pri_tol > 0.0 && s->st.abs_dua_tol > 0.0;
    if ((long)s->groups <= resident) return false;
    if (!reachable && mode != 1) return false;
    return !jit || solve_jit_refill_supported(s->W, s->nx, s->nu, s->N, ct);  // (run-time specialised shapes: built on first use)
}

int launch(tinympc_solver *s, bool timed) {
    int rc;
    s->flag_pending = false;
    if ((rc = resolve_plan(s))) return rc;
    const LaunchPlan pl = current_plan(s);
    const bool fam = pl.families, adaptive = pl.adaptive;
    // k_build_adapt reads the device copy of the references before the solve kernel starts: bring the device copies and the tables up
    // to date the ordinary way (every other kernel of a single-instance handle stages references left in pinned host memory itself)
    if (s->refs_on_host && adaptive) {
        if ((rc = flush_host_refs(s))) return rc;
    }
    if ((rc = refresh_derived(s))) return rc;
    if (adaptive && fam)
        return fail(TINYMPC_ERR_UNSUPPORTED, "adaptive_rho together with cone / linear constraint families is not supported");
    if (s->layout_m && adaptive)
        return fail(TINYMPC_ERR_UNSUPPORTED, "adaptive_rho is not supported for systems with nx+nu > 64");
    if (fam && pl.kernel != KernelId::E && pl.kernel != KernelId::F && pl.kernel != KernelId::M && family_structure(s).beyond_generic())
        return fail(TINYMPC_ERR_UNSUPPORTED, "more than %d linear rows per side, %d cones or %d rounds of overlapping cones run on the run-time "
                    "specialised kernels only (nx+nu <= 16, TINYMPC_JIT not 0): this configuration has none (%s)", MAX_LIN_ROWS, MAX_CONES, MAX_ROUNDS,
                    s->W != 16 ? "nx+nu > 16" : "the specialiser refused it, see tinympc_get_jit_info");
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
    p.tables_in_lds = pl.tables_in_lds ? 1 : 0;
    p.fam = s->dfam; p.GC = s->dGC; p.GL = s->dGL; p.LX = s->dLX;
    p.families = fam ? 1 : 0;
    p.adaptive = adaptive ? 1 : 0;
    p.scratch = s->state_in_global ? s->dscratch_state : nullptr;
    if (s->layout_m && fam) {  // layout M with families: the forward sweep leaves its rollout x | u here for the families' phase
        if (!s->dscratch_state && (rc = dalloc(s, &s->dscratch_state, s->v_doubles()))) return rc;
        p.scratch = s->dscratch_state;
    }
#ifdef TINY_CLOCK_STAMP  // diagnostic build (tools/clock_check.py): layout D stamps its iteration loop into this buffer
    if (!s->state_in_global) {
        if (!s->dclock && (rc = dalloc(s, &s->dclock, (size_t)8 * s->groups))) return rc;
        p.scratch = s->dclock;
    }
#endif
    p.scratch_stride = state_scratch_doubles(s->nu, s->N, s->W);
    p.const_tables = s->tables_const() ? 1 : 0;
    if (s->zero_copy_tick && pl.host_exchange) {  // set by tinympc_mpc_step_batch for the duration of one launch
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
    // a pending cold start: layout D's kernels start from zero registers; every other kernel loads its state from HBM
    // (a solve of zero iterations writes nothing back: the zeros must then be in HBM)
    const bool kernel_takes_cold = (pl.kernel == KernelId::D_COMPILED || pl.kernel == KernelId::D_JIT) && s->st.max_iter > 0;
    if (s->cold_state && !kernel_takes_cold && (rc = materialize_cold_state(s))) return rc;
    p.cold = s->cold_state ? 1 : 0;
    s->cold_state = false;  // (the launch below writes the state back)
    if (s->st.max_iter > 0) s->sol_zero_pending = false;  // (every instance writes its whole solution)
    else if ((rc = materialize_zero_solution(s))) return rc;
    if (refill_applies(s, pl) && !p.x0_mirror && !p.u0_host && !p.host_sol) {  // slot refill (see refill_applies)
        if (!s->drefill && (rc = dalloc(s, &s->drefill, 1))) return rc;
        HIP_TRY(hipMemsetAsync(s->drefill, 0, sizeof(int), s->stream));
        p.refill_next = s->drefill;
    }
    if (timed) HIP_TRY(hipEventRecord(s->ev0, s->stream));
    switch (pl.kernel) {
        case KernelId::M:
            p.ctab = s->dctab;  // (beyond 128 rows: the tile-major copy of the operators; NULL otherwise)
            HIP_TRY(launch_solve_m(p, s->stream));
            break;
        case KernelId::D_JIT:  // (box path, families with everything in registers, or adaptive rho per lane)
            HIP_TRY(launch_solve_jit(p, s->W, s->stream));
            break;
        case KernelId::D_COMPILED:
            HIP_TRY(s->W == 64 ? launch_solve_dx(p, s->stream) : s->W == 32 ? launch_solve_dw(p, s->stream) : launch_solve_d(p, s->stream));
            break;
        case KernelId::E:
            p.ctab = s->dctab_e; p.chunk_len = s->e_chunk_len; p.chunk_count = s->e_wpg; p.chunk_levels = 1;
            HIP_TRY(launch_solve_e(p, s->fs, s->stream));
            break;
        case KernelId::F:
            p.ctab = s->dctab_f; p.ftab = s->dftab; p.chunk_len = s->f_chunk_len; p.chunk_count = s->f_chunks; p.chunk_levels = 4;
            arm_completion_flag(s, p);
            HIP_TRY(launch_solve_f(p, s->f_fs, s->stream));
            break;
        case KernelId::C:  // (with the families: the latency kernel carries them itself, same HBM state as k_admm_solve_fam)
            p.ctab = s->dctab; p.chunk_len = s->chunk_len; p.chunk_count = s->chunk_count; p.chunk_levels = s->chunk_levels;
            arm_completion_flag(s, p);
            HIP_TRY(launch_solve_c(p, s->W, s->KT, s->lds_bytes_c, s->stream));
            break;
        case KernelId::FAM_A:  // shares the persistent state (G, canonical V, D) with every other kernel
            HIP_TRY(launch_solve_fam(p, s->W, s->KT, s->lds_bytes_a, s->stream));
            break;
        case KernelId::ADAPT_A:
            HIP_TRY(launch_solve_adapt(p, s->W, s->KT, s->lds_bytes_a, s->stream));
            break;
        case KernelId::B:
            HIP_TRY(launch_solve_b(p, s->W, s->KT, s->lds_bytes, s->stream));
            break;
        case KernelId::A:
            HIP_TRY(launch_solve(p, s->W, s->KT, s->lds_bytes, s->stream));
            break;
    }
    if (timed) HIP_TRY(hipEventRecord(s->ev1, s->stream));
    return TINYMPC_OK;
}

}  // namespace host
}  // namespace tinympc

extern "C" {

int tinympc_get_launch_info(tinympc_solver *s, int *lanes_per_instance, int *instances_per_wave, int *workgroups,
                            int *lds_bytes, int *tables_in_lds) {
    int rc = check_handle(s);
    if (rc) return rc;
    const LaunchPlan pl = current_plan(s);
    if (lanes_per_instance) *lanes_per_instance = s->W;
    if (instances_per_wave) *instances_per_wave = s->IPW;
    if (workgroups) *workgroups = pl.workgroups;
    if (lds_bytes) *lds_bytes = (int)pl.lds_bytes;
    if (tables_in_lds) *tables_in_lds = pl.tables_in_lds ? 1 : 0;  // (layouts C - F keep their table entries in registers or copy them themselves)
    return TINYMPC_OK;
}

int tinympc_get_layout(tinympc_solver *s) {
    if (!s) return 0;
    return current_plan(s).layout;
}

int tinympc_prepare(tinympc_solver *s) {
    int rc = check_handle(s);
    if (rc) return rc;
    if ((rc = bind_device(s))) return rc;
    s->specialise_asked = true;  // (the caller pays for specialised kernels now: layout F for the box path of small batches)
    if ((rc = resolve_plan(s))) return rc;
    (void)refill_applies(s, current_plan(s));  // (a run-time specialised shape builds its slot-refill variant here, not in the first solve)
    return TINYMPC_OK;
}

int tinympc_get_jit_info(tinympc_solver *s, char *buf, int len) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!buf || len < 1) return fail(TINYMPC_ERR_INVALID_INPUT, "get_jit_info: buffer required");
    buf[0] = '\0';
    const LaunchPlan pl = current_plan(s);
    const bool ct = s->tables_const();
    // a specialisation that was asked for and refused leaves the plan on a generic kernel: say why
    if (!pl.adaptive && !s->f_sig.empty() && !s->f_ok && pl.kernel != KernelId::E && pl.kernel != KernelId::D_JIT && pl.kernel != KernelId::D_COMPILED)
        solve_f_describe(s->nx, s->nu, s->N, ct, pl.families, s->f_fs, buf, (size_t)len);
    else if (!pl.adaptive && !s->e_sig.empty() && !s->e_ok && pl.kernel != KernelId::F && pl.kernel != KernelId::D_JIT && pl.kernel != KernelId::D_COMPILED)
        solve_e_describe(s->nx, s->nu, s->N, ct, pl.families, s->fs, buf, (size_t)len);
    else if (pl.kernel == KernelId::F) solve_f_describe(s->nx, s->nu, s->N, ct, pl.families, s->f_fs, buf, (size_t)len);
    else if (pl.kernel == KernelId::E) solve_e_describe(s->nx, s->nu, s->N, ct, pl.families, s->fs, buf, (size_t)len);
    else if (pl.kernel == KernelId::D_JIT || (s->d_jit_asked && !s->layout_d && !s->layout_m)) {
        if ((rc = bind_device(s))) return rc;
        solve_jit_describe(s->W, s->nx, s->nu, s->N, ct, pl.families && !pl.adaptive, pl.adaptive && !pl.families, buf, (size_t)len);
        if (pl.kernel == KernelId::D_JIT && refill_applies(s, pl)) strncat(buf, " slot-refill", (size_t)len - strlen(buf) - 1);
    } else snprintf(buf, (size_t)len, "compiled-in layout=%c%s", pl.layout, refill_applies(s, pl) ? " slot-refill" : "");
    return TINYMPC_OK;
}

}  // extern "C"
