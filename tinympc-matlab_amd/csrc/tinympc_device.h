// tinympc_device.h -- parameter blocks and launcher declarations shared by the kernels
// (tinympc_kernels.hip) and the C-ABI layer (tinympc_capi.hip).
//
// Lane layout of the solve kernel ("row-per-lane"): a wavefront of 64 lanes is cut into 64/W groups
// of W lanes (W = 16, 32 or 64, the smallest with nx+nu <= W); one group = one MPC instance; inside
// a group lane r < nx owns state row r and lane nx+j owns input row j. Every trajectory array of the
// reference (TinyWorkspace, types.hpp:79-136) that is indexed [row, knot] therefore becomes
// [knot][lane]: one 512-byte line per knot per wavefront, in LDS and in HBM alike.
#pragma once
#ifndef __HIPCC_RTC__  // (run-time compilation, tinympc_jit.hip: hiprtc brings its own runtime header)
#include <hip/hip_runtime.h>
#endif

namespace tinympc {

// Fixed-size operators and per-lane constants built by k_build_operators (doubles):
//   Mf[W*KT]   forward sweep rows    [ A-B*Kinf | -B ]  (state rows),  [ -Kinf | -I ]        (input rows)
//   Mb[W*KT]   backward sweep rows   [ AmBKt | -Kinf' ] (state rows),  [ Quu_inv*B' | Quu_inv ] (input rows)
//   cf[W]      forward constants     fdyn (state rows), 0 (input rows)
//   cb[W]      backward constants    APf (state rows), Quu_inv*BPf (input rows)
//   dg[W]      diagonal of Q+rho*I (state rows) / R+rho*I (input rows)   (tiny_api.cpp:90-91)
__host__ __device__ inline size_t ops_doubles(int W, int KT) { return (size_t)2 * W * KT + 3 * W; }

// Per-knot tables built by k_build_tables (doubles), all [N+2][W] (row k+1 = knot k; rows 0 and N+1
// are padding for the solve kernel's one-step-ahead prefetch):
//   lo, hi     clamp bounds (-inf/+inf where the bound flag is off)              (admm.cpp:49-58)
//   linref     -(Xref .* Q) / -(Uref .* R)                                       (admm.cpp:77, 79)
// followed by pNref[W] = -(Xref[:,N-1]' * Pinf)'                                  (admm.cpp:81)
__host__ __device__ inline size_t table_rows(int N) { return (size_t)N + 2; }

// HBM layout of the slack array V (v|z): [groups][N + 1 + 2*V_PAD][64]. Knot k is row k + V_PAD, row
// N + V_PAD holds the per-lane dummy slots, and V_PAD untouched rows at either end absorb the
// 4-steps-ahead prefetch of the layout-B kernel. Layout B uses two such buffers as a ping-pong pair.
constexpr int V_PAD = 4;
__host__ __device__ inline size_t v_rows(int N) { return (size_t)N + 1 + 2 * V_PAD; }
__host__ __device__ inline size_t tables_doubles(int W, int N) { return 3 * table_rows(N) * W + W; }

// A store into pinned host memory that must be visible to the HOST while the kernel is still running (the solution / statistics behind
// a completion stamp, SolveParams::host_sol): system scope, i.e. written through the device's caches at once. A plain store may stay
// dirty in the L2 until something writes the whole L2 back -- the system-scope release fence the stamp used to carry (round 2-4), which
// also cost every tick the L2's other dirty lines and, as an acquire, its cached operators and tables.
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
__device__ __forceinline__ void host_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#endif

struct PrecomputeParams {
    int nx, nu;
    double rho;
    const double *A, *B, *fdyn, *Qd, *Rd;  // Qd/Rd: diagonals already + rho (tiny_api.cpp:90-91)
    double *Kinf, *Pinf, *Quu_inv, *AmBKt, *APf, *BPf;
    int *info;        // info[0] = Riccati steps taken
    double *scratch;  // global scratch, used when the working set does not fit in LDS
    int use_lds;
};

// The .m class's own Riccati recursions (TinyMPC.m:194-221 compute_cache_terms, :336-366 solve_lqr): full Q
// and R, rho added ONCE, an optional regulariser inside the gain solve only, a matrix-norm stopping test.
struct LqrParams {
    int nx, nu;
    double rho;         // Q_rho = Q + rho*I, R_rho = R + rho*I                         (TinyMPC.m:199-200, 341-342)
    double reg;         // K = (R_rho + B'PB + reg*I) \ (B'PA)                          (:209, :354)
    double tol;         // stop when the norm below of K - Kprev is < tol               (:211, :356)
    int norm_kind;      // 2: spectral norm (MATLAB norm()); 0: max-abs relative to max(1, max|K|)
    int max_iter;       // 5000 in the .m class
    int min_iter;       // first iteration at which the test may stop the loop (solve_lqr: `iter > 1`)
    int p0_augmented;   // initial P: 0 = Q (:204), 1 = Q_rho (:350)
    const double *A, *B, *Q, *R;
    double *K, *P, *C1, *C2;  // Kinf, Pinf, Quu_inv = inv(R_rho + B'PB), AmBKt = (A - B K)'
    int *info;                // info[0] = iterations taken
    double *scratch;
    int use_lds;
};

// out[i] = (hi[i] - lo[i]) / h over the four cache matrices packed back to back (TinyMPC.m:237-240)
struct FiniteDiffParams {
    int count;
    double h;
    const double *lo, *hi;
    double *out;
};

struct OperatorParams {
    int nx, nu, W, KT;
    const double *A, *B, *fdyn, *Qd, *Rd, *Kinf, *Quu_inv, *AmBKt, *APf, *BPf;
    double *ops;
};

struct TableParams {
    int nx, nu, N, W;
    int en_state_bound, en_input_bound;
    const double *x_min, *x_max, *u_min, *u_max, *Xref, *Uref, *Pinf;
    const double *ops;  // for dg[]
    int KT;
    double *tables;
};

// Stamp of a session-mailbox line: the command's sequence number plus a 16-bit checksum of the line's seven payload words and of the
// sequence number itself, as a fraction -- exact in a double for every sequence number below 2^36. Host and kernel compute it the
// same way; a reader accepts a line only if the stamp fits the payload it read with it. The checksum is POSITION-DEPENDENT and
// multiplicative (mail_term per word): through most of round 5 it was the folded XOR of the words, under which equal words
// cancel -- four saturated controls (+-0.4, +-0.4) had the checksum of an all-zero line, and a host that read a line's payload just before
// the kernel's answer landed and its stamp just after took zeros for the answer (tools/fuzz_layout_f.py, 1 tick in ~3,000;
// profiles/r05_session_stamp_bug.txt). Readers also look at the stamp FIRST now (tinympc_session.hip).
// (the checksum: XOR over the seven words of one multiplicative term per word whose multipliers depend on the word's POSITION -- equal
// words at different places do not cancel, swapped words change it -- so that the seven terms can be computed by seven lanes at once and
// combined in three DPP steps: the sequential form of the first fix cost the resident kernels ~0.5 us per tick.)
__host__ __device__ inline unsigned mail_term(unsigned long long word_bits, int position) {
    // (three multiplies, the high half rotated before it enters: a double's sign bit is the TOP bit of its high half, and a multiplication
    // only carries upwards -- with the high half entering as it is, swapping +0.4 and -0.4 between two positions left the checksum unchanged)
    const unsigned c1 = 0x9E3779B1u + 0x3C6EF372u * (unsigned)position, c2 = 0x85EBCA77u + 0x27D4EB2Eu * (unsigned)position;  // (odd)
    const unsigned lo = (unsigned)word_bits, hi = (unsigned)(word_bits >> 32);
    unsigned t = (lo + 0x7F4A7C15u) * c1;
    t ^= t >> 16;
    unsigned u = (((hi << 11) | (hi >> 21)) + t) * c2;
    u ^= u >> 15;
    u *= c1;
    return u ^ (u >> 13);
}
__host__ __device__ inline double mail_stamp(double seq, unsigned payload_hash) {
    unsigned f = payload_hash ^ ((unsigned)(unsigned long long)seq * 0xC2B2AE3Du);
    f ^= f >> 16;
    return seq + (double)(f & 0xffffu) * (1.0 / 65536.0);
}
// host side / tests: the stamp of seven words
__host__ __device__ inline double mail_stamp_of(double seq, const double *words7) {
    unsigned h = 0u;
    for (int q = 0; q < 7; ++q) h ^= mail_term((unsigned long long)__builtin_bit_cast(long long, words7[q]), q);
    return mail_stamp(seq, h);
}
// XOR over each aligned group of eight lanes (a 64-byte line held one word per lane), the same value in all eight: two quad steps and
// the other quad of the half row, on the DPP crossbar. All lanes of the groups that take part must be active.
__device__ __forceinline__ unsigned mail_xor8(unsigned t) {
    t ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    t ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    t ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x141, 0xf, 0xf, false);  // row_half_mirror
    return t;
}
// The mailbox as one wavefront holds it after ONE load instruction -- lane t: word t & 7 of line t >> 3, lanes 56.. hold nothing --: bit l of
// the result says that line l's stamp is `expect` plus the checksum of the seven words that came with it. Wave-uniform; all 64 lanes active.
__device__ __forceinline__ unsigned mail_lines_ok(double word, int lane, double expect) {
    const int slot = lane & 7;
    const unsigned h = mail_xor8((lane < 56 && slot < 7) ? mail_term((unsigned long long)__builtin_bit_cast(long long, word), slot) : 0u);
    const unsigned long long fits = __ballot(lane < 56 && slot == 7 && word == mail_stamp(expect, h));
    unsigned ok = 0u;
#pragma unroll
    for (int l = 0; l < 7; ++l) ok |= (unsigned)((fits >> (8 * l + 7)) & 1ull) << l;
    return ok;
}

struct SolveParams {
    int nx, nu, N, batch;
    int max_iter, check_termination;
    double rho, abs_pri_tol, abs_dua_tol;
    const double *ops;
    const double *tables;
    const double *x0;  // [batch][nx]
    int groups;        // ceil(batch / (64/W))
    double *G;         // [groups][N+1][64]          duals (g|y), persistent across solves (row N: dummy slots)
    double *V, *V2;    // [groups][v_rows(N)][64]    slack (v|z); V is the canonical copy between solves,
                       //                            V2 the second half of layout B's ping-pong pair
    double *D;         // [groups][N-1][IPW*nu]  feed-forward term d, persistent across solves
    double *sol_x;     // [batch][N][nx]      == nx x N x batch column-major
    double *sol_u;     // [batch][N-1][nu]
    int *istats;       // [batch][2]  iter, status
    double *dstats;    // [batch][4]  pri_x, dua_x, pri_u, dua_u
    int tables_in_lds;
    // Long horizons whose ADMM state does not fit the 160 KB of LDS: the layout-A kernels then keep their
    // working copy of (G, V, D) in this per-group HBM scratch instead (same code path, `GMEM` variant).
    double *scratch;          // [groups][scratch_stride] or NULL
    size_t scratch_stride;    // doubles per group: state_scratch_doubles()
    // Cone / linear-inequality slack families (k_admm_solve_fam only; PARITY UNPINNED upstream semantics)
    const double *fam;  // per-lane family description, see fam_doubles()
    double *GC, *GL;    // [groups][v_rows(N)][64]  duals gc|yc and gl|yl, persistent across solves
    double *LX;         // [groups][v_rows(N)][64]  -rho*(vcnew-gc) - rho*(vlnew-gl), forward -> backward
    // Adaptive rho (k_admm_solve_adapt only)
    const double *adapt;  // tables of k_build_adapt, see adapt_doubles()
    double *rho_inst;     // [batch] current rho of every instance; persists across solves like the reference's cache->rho
    double rho_min, rho_max;
    int rho_clip;
    // Layout C (k_admm_solve_c): horizon cut into chunk_count chunks of chunk_len steps, see chunk_plan()
    const double *ctab;   // PhiS_l | PsiS_l (l < chunk_levels), each [16][chunk_ks(nx)]: powers S*2^l of the sweeps' state blocks
    int chunk_len, chunk_count, chunk_levels;
    int families;         // layout C / the specialised layout D: run the cone / linear families too (the other layouts use k_admm_solve_fam)
    int adaptive;         // the specialised layout D: adaptive rho (the other layouts use k_admm_solve_adapt)
    int const_tables;     // bounds and references are the same at every knot (layout B keeps them in registers)
    // Zero-copy closed-loop tick (tinympc_mpc_step_batch, small batches): x0 points into pinned host memory and is
    // mirrored into the device copy; the first controls are also written straight into pinned host memory.
    double *x0_mirror;    // [batch][nx] or NULL
    double *u0_host;      // [batch][nu] or NULL
    // Single-instance handles (the MEX use): solution and statistics are also written into pinned host memory, laid
    // out [nx*N | nu*(N-1) | pri_x, dua_x, pri_u, dua_u | iter, status (as doubles)], so that get_solution / get_stats
    // after a synchronous solve are host copies. NULL otherwise.
    double *host_sol;
    // Completion flag of a single-instance launch (layout C): after its last store to host_sol the workgroup writes this
    // (non-zero) sequence number behind it, at host_sol[nx*N + nu*(N-1) + 6]; the host can then poll pinned memory instead
    // of paying the wake-up latency of hipStreamSynchronize. 0 = no flag.
    double host_seq;
    // Closed-loop SESSION (tinympc_session_begin / _step / _end; layout C, single instance): the kernel stays resident
    // and polls a mailbox in pinned host memory for the next tick's command instead of being launched per tick. The
    // mailbox is an array of 64-byte lines [7 payload doubles | stamp]; payload 0 = flags (1: stop, 2: references changed),
    // payloads 1.. = x0. The host writes a line's payload before its stamp, and the stamp is mail_stamp(sequence number,
    // payload): a line is complete when its stamp fits the expected sequence number AND the payload read with it.
    // NULL = an ordinary one-shot launch.
    const double *mail;
    // The session's EARLY answer (round 5, both resident kernels): the tick's first controls travel ahead of everything else, in
    // 64-byte lines [7 controls | mail_stamp(sequence number, controls)] written by ONE store instruction each -- the host takes them as
    // soon as a line's stamp fits its payload (no fence in front of it, no wait for the solution's write-out: 1.2 us of a 7 us
    // quadrotor tick). Solution, statistics and the completion stamp behind host_sol follow; host readers of those wait for that
    // stamp. NULL: the completion stamp is the answer (one-shot launches).
    double *host_ans;
    double session_expect;             // stamp of the first command to wait for
    unsigned long long session_idle;   // exit after this many 100 MHz ticks without a command (the exit every wave reaches)
    // Single-instance handles, references left in pinned host memory by set_x_ref / set_u_ref (the closed-loop tick
    // with per-tick references, rocket_landing_constraints.m:86-121): the launch's one workgroup rebuilds the
    // reference-dependent table rows itself before anything reads them (refresh_reference_tables, tinympc_sweep.h),
    // so the tick stays ONE launch. NULL otherwise.
    const double *href_x, *href_u;  // nx x N, nu x (N-1), column-major, pinned host
    double *dXref, *dUref;          // their device copies (kept current for the other kernels)
    const double *Pinf;             // for pNref (admm.cpp:81)
    // Cold start without the traffic: the persistent state (G, V, D) is all zeros by CONTRACT (tinympc_reset_workspace, a fresh
    // handle) and has not been materialised in HBM -- the kernel starts from zero registers / LDS instead of loading 14 KB of zeros
    // per instance. Only set for kernels that honour it (layout D and its wide forms); the host zeroes the arrays for every other kernel.
    int cold;
    // Slot refill (layout D, 16 lanes per instance, batches larger than the chip holds at once, tolerances that can be met): the
    // launch is as many wavefronts as are resident together; a 16-lane row whose instance has finished writes it back and takes the
    // next instance index from this counter (instance = 4 x launched wavefronts + old value; zeroed by the host before the launch),
    // so that a wavefront does not idle three rows while its slowest instance iterates. NULL: one instance per row and launch.
    int *refill_next;
    // Layout F (round 4): T_s = Phi^(S-1-s) (-B) as [s][k][16 rows] | aff[16], for the chunk length of the launch (k_build_f_input_tables)
    const double *ftab;
};

struct ChunkTableParams {
    int nx, nu, KT, S, Lc;
    const double *ops;  // fused operators of k_build_operators
    double *out;        // chunk_table_doubles(KT, Lc)
};

// Tables of the adaptive-rho kernel, doubles: mt | pinf | dpinf | dmf | dmb, each [W][KT], then dpnref[W]
//   mt     [A'; B'] rows                       (the A_matrix' * y_vector term of the dual residual)
//   pinf   rows of the base Pinf, dpinf of dPinf/drho
//   dmf    d/drho of the forward operator rows  [ -B*dK | 0 ] / [ -dK | 0 ]
//   dmb    d/drho of the backward operator rows [ 0 | -dK' ] / 0
//   dpnref d/drho of -(Xref_{N-1}' Pinf)'
__host__ __device__ inline size_t adapt_doubles(int W, int KT) { return (size_t)5 * W * KT + W; }
struct AdaptTableParams {
    int nx, nu, N, W, KT;
    const double *A, *B, *Pinf, *dK, *dP, *Xref;
    double *out;
};

// The knot-per-lane form of the families (KFamilies, tinympc_solve_e_common.h; layouts E and F): ONE exchange buffer per wavefront --
// entry (j, t) = nx+nu doubles at (j (S+1) + t) ES, ES odd (conflict-free for the lanes that walk t), padded so that
// the lanes beyond the last entry read inside the buffer
__host__ __device__ constexpr int kfam_es(int nxu) { return nxu | 1; }  // (a stride of 17 -- no EXEC mask on the sweep's stores -- measured no faster: 2.604 against 2.614 ms, 30 KB more LDS)
__host__ __device__ constexpr int kfam_passes(int S) { return (S + 1 + 15) / 16; }
__host__ __device__ constexpr int kfam_doubles(int nxu, int S) { return ((3 * (S + 1) + 16 * kfam_passes(S)) * kfam_es(nxu) + nxu + 1) & ~1; }

// ---- layout F (tinympc_solve_f.hip): one instance per workgroup, up to 4 * wpg chunks of S slots on the DPP rows of `wpg`
// wavefronts. LDS per workgroup, in doubles: operators | tables (!ct) | linear rows (fam) | carry matrices [2][4][16][16] |
// wavefront totals [2][wpg][16] | flags [16] | residual partials [4 wpg][4] | per wavefront d[S * 4 nu]
// (kf_nxu > 0: the families one knot per lane -- the cones' slopes [2][ncone] and one exchange buffer per wavefront on top)
__host__ __device__ constexpr size_t f_lds_bytes(int nu, int N, bool ct, int wpg, int S, bool fam, int nl, int kf_nxu = 0, int ncone = 0) {
    return sizeof(double) * ((size_t)512 + (ct ? 0 : 3 * (N + 2) * 16 + 16) + (fam ? 3 * nl * 16 : 0) + (kf_nxu > 0 ? ((2 * ncone + 1) & ~1) : 0) +
                             2 * 4 * 256 + 2 * wpg * 16 + 16 + 16 * wpg + 64 + (size_t)wpg * ((S * 4 * nu + 1) & ~1) +  // (+ 64: the session's mailbox copy)
                             (kf_nxu > 0 ? (size_t)wpg * kfam_doubles(kf_nxu, S) : 0) + (size_t)S * nu * 16 + 16);       // (+ the chunks' input tables T_s | aff)
}

// Family description built on the host by the C-ABI layer (masks and coefficients only), doubles:
//   role[W]   0 = row in no cone, 1 = norm member, 2 = the cone's last ("t") row
//   mu[W]     slope of the row's cone
//   famc[W]   1 if the cone family is enabled for this row's side (state / input), else 0
//   faml[W]   same for the linear family
//   Cn[W][KT] 0/1: columns = norm-member rows of the row's cone      (a^2 = Cn * s.^2)
//   Ct[W][KT] 0/1: column  = the t row of the row's cone             (t   = Ct * s)
//   Ty[W][KT] 0/1: columns = rows of the same side                   (dot = Ty * (a_k .* s))
//   nl        number of linear rows in use (max over the two sides), then for k < MAX_LIN_ROWS:
//   ak[W], bk[W], nk[W]   coefficient of the row in constraint k, its bound (+inf if absent), ||a_k||^2
//   ... and, for layout E (which walks the cones one by one, in list order): cone_mu[MAX_CONES], the slope of cone c of the
//   active list (state cones first, then input cones)
//   ... then the ROUNDS of the cone list beyond the first: upstream projects the cones one after another, which differs from
//   projecting them at once only where two cones share a row. The cones are grouped into rounds of pairwise-disjoint cones (a
//   cone that overlaps an earlier cone of the current round opens the next round); role / mu / Cn / Ct above describe round
//   0, and behind cone_mu follow nround (1 double) and, for rounds 1 .. MAX_ROUNDS-1, role[W] | mu[W] | Cn[W][KT] | Ct[W][KT]:
//   fam_round_offset(). The generic kernels that walk rounds (k_admm_solve_fam) read them from there.
// MAX_LIN_ROWS / MAX_CONES / MAX_ROUNDS are what the GENERIC kernels hold (the LDS copy of the latency kernel and of layout D's
// families variant, the per-round registers of k_admm_solve_fam): the buffer's default capacities. A configuration beyond them
// (round 4: up to HARD_MAX_LIN_ROWS rows per side, HARD_MAX_CONES cones, any number of rounds) gets a buffer whose blocks are sized
// for IT -- `lin_cap`, `cone_cap`, `round_cap` of the offset functions, max(default, what the configuration has) -- and runs on the
// structure-specialised kernels (layouts E and F), which know the counts at compile time and derive the same capacities.
constexpr int MAX_LIN_ROWS = 32;  // per side
constexpr int FAM_REG_ROWS = 8;   // k_admm_solve_fam keeps this many rows' coefficients in registers, the rest is read per use
constexpr int MAX_CONES = 16;
constexpr int MAX_ROUNDS = 4;
constexpr int HARD_MAX_LIN_ROWS = 128, HARD_MAX_CONES = 64;
__host__ __device__ constexpr int fam_lin_cap(int nl) { return nl > MAX_LIN_ROWS ? nl : MAX_LIN_ROWS; }
__host__ __device__ constexpr int fam_cone_cap(int ncone) { return ncone > MAX_CONES ? ncone : MAX_CONES; }
__host__ __device__ constexpr int fam_round_cap(int nround) { return nround > MAX_ROUNDS ? nround : MAX_ROUNDS; }
__host__ __device__ inline size_t fam_cone_mu_offset(int W, int KT, int lin_cap = MAX_LIN_ROWS) { return (size_t)4 * W + (size_t)3 * W * KT + 1 + (size_t)3 * lin_cap * W; }
__host__ __device__ inline size_t fam_round_doubles(int W, int KT) { return (size_t)2 * W + (size_t)2 * W * KT; }
__host__ __device__ inline size_t fam_nround_offset(int W, int KT, int lin_cap = MAX_LIN_ROWS, int cone_cap = MAX_CONES) { return fam_cone_mu_offset(W, KT, lin_cap) + cone_cap; }
__host__ __device__ inline size_t fam_round_offset(int W, int KT, int round, int lin_cap = MAX_LIN_ROWS, int cone_cap = MAX_CONES) {
    return fam_nround_offset(W, KT, lin_cap, cone_cap) + 1 + (size_t)(round - 1) * fam_round_doubles(W, KT);
}
__host__ __device__ inline size_t fam_doubles(int W, int KT, int lin_cap = MAX_LIN_ROWS, int cone_cap = MAX_CONES, int round_cap = MAX_ROUNDS) {
    return fam_round_offset(W, KT, round_cap, lin_cap, cone_cap);
}

// ---- layout E (tinympc_solve_e.hip): the horizon cut across the `wpg` wavefronts of a workgroup, S slots each (the last
// wavefront: what is left). LDS plan per workgroup, in doubles:
//   operators [2][16 k][16 r] | tables (!ct) | linear rows (fam) | carry matrices Phi^S, Psi^S [2][16 k][16 r] |
//   carries fwd / bwd [2][wpg][64] | termination flags [16] | residual partials [wpg][16] |
//   knot-0 state of the bottom wavefront + per-lane scalars [6][64] |
//   per wavefront: the families' arrays that live in LDS -- gc, gl, lx, each [S][e_fam_row(nx+nu)], the four instances' real
//   rows packed (`lds_arrays` of them) -- | d[S * 4 nu]
__host__ __device__ constexpr int e_d_doubles(int nu, int S) { return (S * 4 * nu + 1) & ~1; }
__host__ __device__ constexpr int e_fam_row(int nxu) { return (4 * nxu + 1) & ~1; }
__host__ __device__ constexpr int e_shared_doubles(int N, bool ct, int wpg, bool fam, int nl, int ncone = 0, int gpw = 1) {
    // (... | linear rows | the cones' slopes and their reciprocals [2][ncone] | carry matrices | per group: carries, flags, partials, knot 0)
    return 512 + (ct ? 0 : 3 * (N + 2) * 16 + 16) + (fam ? 3 * nl * 16 + ((2 * ncone + 1) & ~1) : 0) + 512 + gpw * (2 * wpg * 64 + 16 + wpg * 16 + 6 * 64);
}
__host__ __device__ constexpr int e_wave_doubles(int nxu, int nu, int S, int lds_arrays) {
    return (lds_arrays < 0 ? kfam_doubles(nxu, S) : lds_arrays * S * e_fam_row(nxu)) + e_d_doubles(nu, S);
}
__host__ __device__ constexpr size_t e_lds_bytes(int nxu, int nu, int N, bool ct, int wpg, int S, bool fam, int nl, int lds_arrays, int ncone = 0, int gpw = 1) {
    return sizeof(double) * ((size_t)e_shared_doubles(N, ct, wpg, fam, nl, ncone, gpw) + (size_t)gpw * wpg * e_wave_doubles(nxu, nu, S, fam ? lds_arrays : 0));
}

#ifndef __HIPCC_RTC__  // host side only
// Structure of the cone / linear families, as layout E is specialised on it: the ACTIVE cones in list order (state cones, then
// input cones) as {round, first lane, last lane} -- lane = row for a state row, nx + row for an input row; cones of one round
// are pairwise disjoint, a cone that overlaps an earlier cone of the current round opens the next one -- and the number of
// linear rows per side.
struct FamilyStructure {
    int ncone = 0, nround = 0, nlx = 0, nlu = 0;
    int cone[HARD_MAX_CONES][3] = {};
    // beyond what the generic kernels hold (their LDS copies / per-round registers): only layouts E and F run such a configuration
    bool beyond_generic() const { return ncone > MAX_CONES || nround > MAX_ROUNDS || nlx > MAX_LIN_ROWS || nlu > MAX_LIN_ROWS; }
};
// Layout E (tinympc_solve_e.hip, run-time specialised only): the horizon cut across the wavefronts of a workgroup
bool solve_e_plan(int nx, int nu, int N, bool const_tables, bool families, const FamilyStructure &fs, int *chunk_len, int *wpg, size_t *lds_bytes, int *gpw);
bool solve_jit_enabled();  // TINYMPC_JIT is not 0
bool solve_e_supported(int nx, int nu, int N, bool const_tables, bool families, const FamilyStructure &fs);  // plans AND compiles
hipError_t launch_solve_e(const SolveParams &p, const FamilyStructure &fs, hipStream_t stream);
void solve_e_describe(int nx, int nu, int N, bool const_tables, bool families, const FamilyStructure &fs, char *buf, size_t len);
void solve_jit_describe(int W, int nx, int nu, int N, bool const_tables, bool families, bool adaptive, char *buf, size_t len);
// Layout F (tinympc_solve_f.hip, run-time specialised only): the latency kernel with compile-time shape and structure
bool solve_f_plan(int nx, int nu, int N, bool const_tables, bool families, const FamilyStructure &fs, int *chunk_len, int *chunks, int *wpg, size_t *lds_bytes);
bool solve_f_supported(int nx, int nu, int N, bool const_tables, bool families, const FamilyStructure &fs);  // plans AND compiles
hipError_t launch_solve_f(const SolveParams &p, const FamilyStructure &fs, hipStream_t stream);
bool solve_f_builtin(int nx, int nu, int N, bool const_tables, bool families, const FamilyStructure &fs, bool session);  // compiled in: costs nothing to ask for
bool solve_f_session_supported(int nx, int nu, int N, bool launch_const_tables, bool families, const FamilyStructure &fs);  // the resident closed-loop variant, in the form of the handle's launch kernel (compiles)
hipError_t launch_solve_f_session(const SolveParams &p, const FamilyStructure &fs, bool launch_const_tables, hipStream_t stream);
void solve_f_describe(int nx, int nu, int N, bool const_tables, bool families, const FamilyStructure &fs, char *buf, size_t len);
// Launchers (defined in tinympc_kernels.hip). All are asynchronous on `stream`.
hipError_t launch_precompute(const PrecomputeParams &p, hipStream_t stream);
// nx <= 12, nu <= 4: the Riccati fixed point in the registers of one wavefront (tinympc_precompute_rows.hip)
bool precompute_rows_supported(int nx, int nu);
hipError_t launch_precompute_rows(const PrecomputeParams &p, hipStream_t stream);
hipError_t launch_lqr(const LqrParams &p, hipStream_t stream);
hipError_t launch_finite_diff(const FiniteDiffParams &p, hipStream_t stream);
hipError_t launch_fill(double *dst, size_t count, double value, hipStream_t stream);  // asynchronous constant fill
struct SetupInitParams {  // k_setup_init: everything a small handle's setup puts into device memory before k_precompute
    const double *stage;  // pinned host: the problem data in the device arena's upload-span layout
    double *upload_dst;
    size_t upload_doubles;
    double *zero;         // the zero-initialised span (references, x0, G, V, V2, D, solutions, statistics ...)
    size_t zero_doubles;
    double *xmin, *xmax, *umin, *umax;
    size_t X, U;
    double inf;
    double *rho_inst;
    int batch;
    double rho;
    double *mail;         // the session's device-side mailbox (or NULL)
};
hipError_t launch_setup_init(const SetupInitParams &p, hipStream_t stream);
hipError_t launch_fill_bounds(double *xmin, double *xmax, size_t X, double *umin, double *umax, size_t U, double inf, hipStream_t stream);
hipError_t launch_reset_stats(int *istats, double *dstats, double *rho_inst, int batch, double rho, hipStream_t stream);
size_t lqr_scratch_doubles(int nx, int nu);
// The same precompute for large systems (nx+nu > 64; tinympc_precompute_large.hip): every matrix product of an iteration its own
// launch on the FP64 matrix cores; blocks on the stream while it looks at the data-dependent iteration count.
size_t precompute_large_scratch_doubles(int nx, int nu);
hipError_t launch_precompute_large(const PrecomputeParams &p, hipStream_t stream);
hipError_t launch_build_operators(const OperatorParams &p, hipStream_t stream);
hipError_t launch_build_tables(const TableParams &p, hipStream_t stream);
// Layout A: one wavefront per workgroup, all ADMM state in LDS (lowest latency, 2 waves per CU).
// Chooses the <W,KT> instantiation; returns hipErrorInvalidValue when none fits.
hipError_t launch_solve(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream);
// Layout B: four wavefronts per workgroup sharing the tables in LDS, G and D in LDS, V as an
// L2-resident ping-pong pair in HBM (4 waves per CU). Only W = 16, N >= 8.
hipError_t launch_solve_b(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream);
size_t solve_b_lds_bytes(int nx, int nu, int N, int W);
constexpr int WAVES_PER_GROUP_B = 4;
// Layout D: horizon unrolled at compile time, duals in registers, slack split between registers and LDS, two waves
// per SIMD. Only for the (nx, nu, N) shapes compiled into the library and time-invariant bounds / references.
bool solve_d_supported(int nx, int nu, int N, bool const_tables);
hipError_t launch_solve_d(const SolveParams &p, hipStream_t stream);
hipError_t launch_solve_d_refill(const SolveParams &p, hipStream_t stream);  // tinympc_solve_dr.hip (SolveParams::refill_next set)
int solve_d_workgroups(int nu, int N, bool const_tables, int groups);
size_t solve_d_lds_bytes(int nu, int N, bool const_tables);  // per workgroup
int solve_d_resident_workgroups(int wpg);  // workgroups of `wpg` wavefronts the device holds at two wavefronts per SIMD (slot refill)
int solve_d_wavefronts_per_workgroup(int nu, int N, bool const_tables);
// ... and its 32-lanes-per-instance form for wide systems (16 < nx+nu <= 32), tinympc_solve_dw.hip
bool solve_dw_supported(int nx, int nu, int N, bool const_tables);
hipError_t launch_solve_dw(const SolveParams &p, hipStream_t stream);
int solve_dw_workgroups(int nu, int N, int groups);
size_t solve_dw_lds_bytes(int nu, int N);
// ... and with 64 lanes per instance (32 < nx+nu <= 64), tinympc_solve_dx.hip
bool solve_dx_supported(int nx, int nu, int N, bool const_tables);
hipError_t launch_solve_dx(const SolveParams &p, hipStream_t stream);
int solve_dx_workgroups(int nu, int N, int groups);
size_t solve_dx_lds_bytes(int nu, int N);
// Layout A plus the cone / linear slack families (extra duals and the extra linear-cost term in HBM).
hipError_t launch_solve_fam(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream);
// Layout C: one instance per 256-thread workgroup, the horizon swept in 16 concurrent chunks (latency kernel
// for small batches). W = 16, N <= 129.
hipError_t launch_solve_c(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream);
hipError_t launch_build_chunk_tables(const ChunkTableParams &p, hipStream_t stream);
// layout F: T_s = Phi^(S-1-s) (-B) [s][k][16] | aff[16] (tinympc_solve_c.hip: k_build_f_input_tables); p.Lc unused
size_t f_input_table_doubles(int nu, int S);
hipError_t launch_build_f_input_tables(const ChunkTableParams &p, hipStream_t stream);
void chunk_plan(int N, int *S, int *C, int *Lc);
size_t solve_c_lds_bytes(int nx, int Lc);
size_t solve_c_lds_bytes_refs(int nx, int nu, int N, int Lc);
size_t chunk_table_doubles(int nx, int Lc);
int chunk_ks(int nx);
// Layout A plus adaptive rho (per-instance rho, Taylor-updated operators).
hipError_t launch_solve_adapt(const SolveParams &p, int W, int KT, size_t lds_bytes, hipStream_t stream);
hipError_t launch_build_adapt(const AdaptTableParams &p, hipStream_t stream);
// Raise a kernel's dynamic-LDS limit to `bytes` unless a previous launch on this device already did
// (the attribute is sticky per function and device; re-setting it costs ~5 us per launch, which matters
// for the per-tick latency of closed-loop callers). `cache` is one static array per kernel instantiation.
inline hipError_t ensure_dynamic_lds(const void *fn, size_t bytes, size_t (&cache)[16]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16 && cache[dev] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 16) cache[dev] = bytes;
    return e;
}

// Large systems (64 < nx+nu <= 128) on the FP64 matrix cores, 16 instances per tile (tinympc_solve_m.hip)
bool solve_m_supported(int nx, int nu);
int solve_m_geometry(int nx, int nu);
size_t solve_m_tiled_ops_doubles(int nx, int nu);  // beyond 128 rows: the tile-major copy of the operators the kernel streams (0 otherwise)
hipError_t launch_tile_operators_m(const double *ops, double *out, int nx, int nu, hipStream_t stream);  // W = KT of the operators / tables of a large system: 128, 256 (beyond 128 rows) or 512 (beyond 256)
size_t solve_m_state_doubles(int nx, int nu, int N, int tiles);
// the families' description for layout M (compact: tinympc_solve_m.hip, FAM): size for `nl` linear rows, and where its blocks start
size_t solve_m_fam_doubles(int nx, int nu, int nl);
size_t solve_m_fam_cone_offset();
size_t solve_m_fam_lin_offset();
int solve_m_fam_fast_rows();  // up to this many linear rows per side the description carries the rows' Gram matrices (behind the rows)
hipError_t launch_solve_m(const SolveParams &p, hipStream_t stream);
// Run-time specialisation of layout D (tinympc_jit.hip): any (nx, nu, N) that fits the register / LDS plan, compiled with
// hiprtc from the very sources of the compiled-in instantiations on first use and cached (memory + disk).
bool solve_jit_supported(int W, int nx, int nu, int N, bool const_tables, bool families = false, bool adaptive = false);
hipError_t launch_solve_jit(const SolveParams &p, int W, hipStream_t stream);
bool solve_jit_refill_supported(int W, int nx, int nu, int N, bool const_tables);   // (compiles the slot-refill variant on first use)
int solve_jit_resident_wavefronts(int W, int nx, int nu, int N, bool const_tables);  // wavefronts of the shape's plan the device holds at once
int solve_jit_workgroups(int W, int nx, int nu, int N, bool const_tables, int groups, bool families = false, bool adaptive = false);
size_t solve_jit_lds_bytes(int W, int nx, int nu, int N, bool const_tables, bool families = false, bool adaptive = false);  // per workgroup, from the plan  // 8 wavefronts per workgroup, 4 on the long-horizon plan
#endif  // !__HIPCC_RTC__

// Doubles of working state per group in layout A (G and V with N+2 rows, D with 64 dummy slots).
__host__ __device__ inline size_t state_scratch_doubles(int nu, int N, int W) {
    return (size_t)2 * (N + 2) * 64 + ((((size_t)(N - 1) * (64 / W) * nu + 64) + 1) & ~(size_t)1);
}

#ifndef __HIPCC_RTC__
// Geometry helpers shared with the host layer.
bool choose_geometry(int nx, int nu, int *W, int *KT);
size_t solve_lds_bytes(int nx, int nu, int N, int W, bool tables_in_lds);
size_t precompute_scratch_doubles(int nx, int nu);
#endif

}  // namespace tinympc
