/* tinympc_hip.h -- C ABI of libtinympc_hip.so: the MI355X (gfx950) implementation of TinyMPC's
 * ADMM hot path behind the reference's MEX verb surface.
 *
 * The reference exposes ONE MEX function, tinympc_matlab('<verb>', args...), dispatching 17 string
 * verbs onto a process-global solver (/root/reference/src/bindings.cpp:17, 641-692). This header
 * declares one extern "C" function per verb, taking an explicit handle instead of the global, raw
 * column-major `const double*` buffers with explicit sizes instead of mxArrays, and returning an
 * int status instead of long-jumping out through mexErrMsgIdAndTxt. A MEX shim that forwards the
 * 17 verbs to these functions is in tinympc-matlab_amd/matlab/tinympc_matlab_mex.cpp; INTEGRATION.md
 * shows the binding a maintainer would add.
 *
 * Conventions
 *  - All matrices are FP64, column-major, column = knot point (reference types.hpp:15-17,
 *    bindings.cpp:40-41). Inputs are copied during the call (the caller keeps ownership);
 *    outputs are written into caller-allocated buffers.
 *  - Every function returns TINYMPC_OK (0) or a negative TINYMPC_ERR_* code; the message is
 *    available from tinympc_last_error() (thread-local). Nothing throws or long-jumps.
 *  - Threading: a handle is externally synchronised -- one caller at a time per handle; DIFFERENT handles may be driven
 *    from different threads concurrently (one handle per thread, or one per GPU, is the intended use). The library keeps
 *    no other shared mutable state than (a) the run-time specialiser's kernel cache and (b) the registry of open closed-loop
 *    sessions, both behind their own mutexes. The one place where a call reaches into a handle it was not given --
 *    tinympc_setup / tinympc_setup_batch / tinympc_reset sending the resident session kernels of OTHER handles on the same
 *    device home before they allocate or free device memory -- takes that handle's session mutex, which
 *    tinympc_session_step holds for the duration of a tick: a tick in flight on another thread completes undisturbed, and
 *    its handle cannot be destroyed under the walk (it leaves the registry first).
 *  - A handle owns one HIP stream and all its device buffers, and is bound to one GPU. tinympc_solve is synchronous.
 *  - There is NO CPU fallback: without a GPU / without the gfx950 code object every compute verb
 *    fails with TINYMPC_ERR_NO_DEVICE or TINYMPC_ERR_HIP.
 */
#ifndef TINYMPC_HIP_H
#define TINYMPC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define TINYMPC_ABI_VERSION 1

#define TINYMPC_OK 0
#define TINYMPC_ERR_INVALID_INPUT (-1)   /* MEX id TinyMPC:InvalidInput   (bindings.cpp:22,50,...) */
#define TINYMPC_ERR_NOT_INITIALIZED (-2) /* MEX id TinyMPC:NotInitialized (bindings.cpp:113,...)   */
#define TINYMPC_ERR_HIP (-3)             /* a HIP runtime call failed (message has the hipError)   */
#define TINYMPC_ERR_UNSUPPORTED (-4)     /* problem shape outside what the kernels support         */
#define TINYMPC_ERR_NOT_IMPLEMENTED (-5) /* reserved: no verb returns it any more (every verb is implemented) */
#define TINYMPC_ERR_NO_DEVICE (-6)       /* no HIP device visible                                  */
#define TINYMPC_ERR_ALLOC (-7)

/* solver status values, as the reference core reports them (admm.cpp:114, 183) */
#define TINYMPC_STATUS_SOLVED 1
#define TINYMPC_STATUS_UNSOLVED 11

typedef struct tinympc_solver tinympc_solver; /* opaque */

const char *tinympc_last_error(void);
int tinympc_abi_version(void);
/* Number of visible HIP devices (0 when none); never fails. */
int tinympc_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * The 17 MEX verbs (bindings.cpp:641-692). `verbose` is the trailing scalar every verb takes.
 * ------------------------------------------------------------------------------------------- */

/* verb 'setup'  (bindings.cpp:47-104 -> tiny_setup, tiny_api.cpp:21-122).
 * A nx*nx, B nx*nu, fdyn nx (may be NULL = zeros), Q nx*nx, R nu*nu (only the diagonals of Q and R
 * are used, tiny_api.cpp:90-91). Allocates the handle, uploads the problem, runs the LQR-cache
 * precompute kernel (tiny_precompute_and_set_cache, tiny_api.cpp:124-190) on the device and installs
 * the core's default settings (tiny_api.cpp:213-231). Equivalent to
 * tinympc_setup_batch(out, ..., batch=1, device=-1). */
int tinympc_setup(tinympc_solver **out, const double *A, const double *B, const double *fdyn,
                  const double *Q, const double *R, double rho, int nx, int nu, int N, int verbose);

/* verb 'set_x0' (bindings.cpp:107-131 -> tiny_set_x0, tiny_api.cpp:233-243). len must be nx.
 * In a batched handle this sets instance 0. */
int tinympc_set_x0(tinympc_solver *s, const double *x0, int len, int verbose);

/* verb 'set_x_ref' (bindings.cpp:134-158 -> tiny_api.cpp:245-255). Xref is nx x N. */
int tinympc_set_x_ref(tinympc_solver *s, const double *Xref, int rows, int cols, int verbose);

/* verb 'set_u_ref' (bindings.cpp:161-185 -> tiny_api.cpp:257-267). Uref is nu x (N-1). */
int tinympc_set_u_ref(tinympc_solver *s, const double *Uref, int rows, int cols, int verbose);

/* verb 'set_bound_constraints' (bindings.cpp:188-209). x_min/x_max nx x N, u_min/u_max nu x (N-1),
 * already expanded by the caller (TinyMPC.m:256-264). Enables both bound flags (bindings.cpp:206-207). */
int tinympc_set_bound_constraints(tinympc_solver *s, const double *x_min, const double *x_max,
                                  const double *u_min, const double *u_max, int verbose);

/* verb 'solve' (bindings.cpp:212-232 -> tiny_solve -> solve(), admm.cpp:109-207). One kernel launch
 * runs the whole ADMM loop, in-kernel termination included, for every instance of the handle.
 * Returns TINYMPC_OK whether or not the solver converged (the MEX verb swallows the core status,
 * bindings.cpp:224-231); the core status is reported by tinympc_get_stats. */
int tinympc_solve(tinympc_solver *s, int verbose);

/* verb 'get_solution' (bindings.cpp:235-261): x_out nx*N, u_out nu*(N-1) = the projected slack
 * variables vnew/znew (admm.cpp:187-188, 204-205). Instance 0 of a batched handle. */
int tinympc_get_solution(tinympc_solver *s, double *x_out, double *u_out, int verbose);

/* verb 'get_stats' (bindings.cpp:264-285): iter, status, primal_residual_state, primal_residual_input. */
int tinympc_get_stats(tinympc_solver *s, int *iter, int *status, double *pri_res_state,
                      double *pri_res_input, int verbose);

/* verb 'codegen' (bindings.cpp:288-316 -> tiny_codegen, codegen.cpp:56-68): writes
 * <output_dir>/tinympc/tiny_data.hpp, <output_dir>/src/tiny_data.cpp and <output_dir>/src/tiny_main.cpp from
 * the cache on the device, the settings, the dynamics and the bounds. Copying the solver sources next to them
 * is the caller's job, as in the reference (TinyMPC.m:415-434). */
int tinympc_codegen(tinympc_solver *s, const char *output_dir, int verbose);

/* Everything the emitter writes, as plain host pointers (column-major). tinympc_codegen() fills this from a
 * handle; tinympc_codegen_emit() is host-only file writing and needs no device. */
typedef struct tinympc_codegen_data {
    int nx, nu, N;
    int iter, solved;                                              /* TinySolution.iter / .solved at the time of the call */
    double rho;
    const double *Kinf, *Pinf, *Quu_inv, *AmBKt;                   /* nu x nx, nx x nx, nu x nu, nx x nx */
    const double *dKinf_drho, *dPinf_drho, *dC1_drho, *dC2_drho;   /* emitted only when adaptive_rho != 0; may be NULL */
    double abs_pri_tol, abs_dua_tol;
    int max_iter, check_termination, en_state_bound, en_input_bound, adaptive_rho;
    const double *Q, *R;                                           /* cost diagonals INCLUDING rho: nx, nu (tiny_api.cpp:90-91) */
    const double *Adyn, *Bdyn;                                     /* nx x nx, nx x nu */
    const double *x_min, *x_max, *u_min, *u_max;                   /* nx x N, nx x N, nu x (N-1), nu x (N-1) */
} tinympc_codegen_data;
int tinympc_codegen_emit(const tinympc_codegen_data *data, const char *output_dir, int verbose);

/* verb 'set_sensitivity_matrices' (bindings.cpp:319-361; there it only prints the norms). dK (nu x nx) and
 * dP (nx x nx) become the cache's dKinf_drho / dPinf_drho that the adaptive-rho update reads
 * (rho_benchmark.cpp:205-206); dC1 / dC2 are validated and otherwise unused (they would update the copies
 * C1 / C2, which no solve phase reads). Zero until set. */
int tinympc_set_sensitivity_matrices(tinympc_solver *s, const double *dK, const double *dP,
                                     const double *dC1, const double *dC2, int verbose);

/* verb 'set_cache_terms' (bindings.cpp:364-405): overwrite Kinf (nu x nx), Pinf (nx x nx),
 * Quu_inv (nu x nu), AmBKt (nx x nx); C1/C2 alias the last two. */
int tinympc_set_cache_terms(tinympc_solver *s, const double *Kinf, const double *Pinf,
                            const double *Quu_inv, const double *AmBKt, int verbose);

/* verb 'set_linear_constraints' (bindings.cpp:408-431): rows of Alin_x (nlx x nx) * x <= blin_x and
 * Alin_u (nlu x nu) * u <= blin_u at every knot; a non-empty side is auto-enabled (:422-429).
 * PARITY UNPINNED (no core source in the reference tree). */
int tinympc_set_linear_constraints(tinympc_solver *s, const double *Alin_x, const double *blin_x,
                                   int nlx, const double *Alin_u, const double *blin_u, int nlu);

/* verb 'set_cone_constraints' (bindings.cpp:433-478): per cone (start row Ac, 0-based; dimension qc;
 * slope c), constraint ||s[Ac : Ac+qc-1)||_2 <= c * s[Ac+qc-1] at every knot. State cones first,
 * then input cones (the MATLAB-side order). PARITY UNPINNED. */
int tinympc_set_cone_constraints(tinympc_solver *s, const int *Acx, const int *qcx, const double *cx,
                                 int ncx, const int *Acu, const int *qcu, const double *cu, int ncu);

/* verb 'codegen_with_sensitivity' (bindings.cpp:481-529 -> codegen.cpp:70-90): as 'codegen'; the four
 * sensitivity matrices are written into the generated cache only while adaptive_rho is enabled. */
int tinympc_codegen_with_sensitivity(tinympc_solver *s, const char *output_dir, const double *dK,
                                     const double *dP, const double *dC1, const double *dC2,
                                     int verbose);

/* verb 'reset' (bindings.cpp:532-545): destroy the handle and free every host and device buffer.
 * *s is set to NULL. Passing NULL / a NULL handle is a no-op, as in the reference. */
int tinympc_reset(tinympc_solver **s, int verbose);

/* verb 'update_settings' (bindings.cpp:548-603), same argument order as the MEX verb. With
 * adaptive_rho != 0 solve runs the adaptive-rho loop of the old core (admm.cpp:117-174, rho_benchmark.cpp) on
 * the device, per instance; it cannot be combined with the cone / linear families (TINYMPC_ERR_UNSUPPORTED
 * at solve). */
int tinympc_update_settings(tinympc_solver *s, double abs_pri_tol, double abs_dua_tol, int max_iter,
                            int check_termination, int en_state_bound, int en_input_bound,
                            int en_state_soc, int en_input_soc, int en_state_linear,
                            int en_input_linear, int adaptive_rho, double adaptive_rho_min,
                            double adaptive_rho_max, int adaptive_rho_enable_clipping, int verbose);

/* verb 'print_problem_data' (bindings.cpp:606-638): prints the same scalars to stdout. */
int tinympc_print_problem_data(tinympc_solver *s);

/* ---------------------------------------------------------------------------------------------
 * Methods the reference implements in MATLAB inside the class (no MEX verb); here they run on the
 * device and the .m class / Python mirror forward to them. Any output pointer may be NULL.
 * ------------------------------------------------------------------------------------------- */

/* TinyMPC.compute_cache_terms (src/TinyMPC.m:194-221): Riccati recursion on the FULL user Q, R with rho
 * added once, P0 = Q, gain solve regularised by 1e-8, at most 5000 steps, stop at ||K - Kprev||_2 < 1e-10.
 * Outputs Kinf (nu x nx), Pinf (nx x nx), Quu_inv (nu x nu), AmBKt (nx x nx), column-major. */
int tinympc_compute_cache_terms(tinympc_solver *s, double *Kinf, double *Pinf, double *Quu_inv,
                                double *AmBKt, int *riccati_iters, int verbose);

/* TinyMPC.solve_lqr (src/TinyMPC.m:336-366): stabilising DARE solution for Q + rho_val*I, R + rho_val*I
 * (MATLAB calls idare; here the recursion runs until K is stationary to rounding). C1 = inv(R_rho + B'PB),
 * C2 = (A - BK)'. */
int tinympc_solve_lqr(tinympc_solver *s, double rho_val, double *K, double *P, double *C1, double *C2,
                      int *riccati_iters);

/* TinyMPC.compute_sensitivity_autograd (src/TinyMPC.m:223-241): forward differences of solve_lqr in rho
 * with h = 1e-6. dK (nu x nx), dP (nx x nx), dC1 (nu x nu), dC2 (nx x nx). */
int tinympc_compute_sensitivity(tinympc_solver *s, double *dK, double *dP, double *dC1, double *dC2,
                                int verbose);

/* ---------------------------------------------------------------------------------------------
 * Extensions (not in the reference): cache read-back, batched mode, timing.
 * ------------------------------------------------------------------------------------------- */

/* Read back what the precompute kernel produced (any pointer may be NULL). riccati_iters = number
 * of fixed-point steps taken before max|dK| < 1e-5 (tiny_api.cpp:157). */
int tinympc_get_cache(tinympc_solver *s, double *Kinf, double *Pinf, double *Quu_inv, double *AmBKt,
                      int *riccati_iters);

/* All four residuals of instance 0: primal state, dual state, primal input, dual input
 * (admm.cpp:93-96). */
int tinympc_get_residuals(tinympc_solver *s, double residuals[4]);

/* Batched setup: `batch` independent MPC instances sharing (A,B,fdyn,Q,R,rho,bounds,refs,settings),
 * each with its own x0 and its own persistent ADMM state. device < 0 keeps the current HIP device. */
int tinympc_setup_batch(tinympc_solver **out, const double *A, const double *B, const double *fdyn,
                        const double *Q, const double *R, double rho, int nx, int nu, int N,
                        int batch, int device, int verbose);

/* x0s is nx x count (column b = instance first+b), host memory. */
int tinympc_set_x0_batch(tinympc_solver *s, const double *x0s, int first, int count);
/* Same, from device memory on the handle's GPU (e.g. a torch tensor's data_ptr). Contract: whatever
 * PRODUCED d_x0s on another stream must have completed before the call (the copy runs on the handle's own
 * non-blocking stream, which is ordered against no other stream); the copy itself has completed when the
 * call returns, so the caller may free or overwrite d_x0s right away. */
int tinympc_set_x0_batch_device(tinympc_solver *s, const double *d_x0s, int first, int count);

/* Zero the persistent ADMM state (cold start) of every instance and put every instance's rho back to
 * the setup value; x0 is kept. */
int tinympc_reset_workspace(tinympc_solver *s);

/* Adaptive rho (settings adaptive_rho*, admm.cpp:117-174, rho_benchmark.cpp): every instance carries its own
 * rho, adapted every 5th iteration and kept across solves like the reference's cache->rho. Reads it back for
 * instances [first, first+count). Equals the setup rho while adaptive_rho has never been on. */
int tinympc_get_rho_batch(tinympc_solver *s, double *rho_out, int first, int count);

/* Solutions of instances [first, first+count): x_out nx*N*count, u_out nu*(N-1)*count. */
int tinympc_get_solution_batch(tinympc_solver *s, double *x_out, double *u_out, int first, int count);
/* First control of each instance, nu x count: what a closed-loop caller applies. */
int tinympc_get_first_controls_batch(tinympc_solver *s, double *u0_out, int first, int count);
/* iters[count], status[count], residuals 4 x count (any may be NULL). */
int tinympc_get_stats_batch(tinympc_solver *s, int *iters, int *status, double *residuals, int first,
                            int count);
/* Device pointers of the solution buffers (nx*N*batch and nu*(N-1)*batch doubles), valid until
 * tinympc_reset; lets a caller consume results without a D2H copy. The pointers are stable for the handle's lifetime and
 * need not be queried again after tinympc_reset_workspace: once they have been handed out, a reset zeroes the buffers on
 * the handle's stream itself (before, it only marks them zero and lets the next solve overwrite them). Reads must be
 * ordered behind the handle's stream (tinympc_get_stream) or a synchronous verb. */
int tinympc_get_solution_device_ptrs(tinympc_solver *s, const double **d_x, const double **d_u);

/* Closed-loop SESSION (single-instance handles): the reference's control loops call set_x0 -> solve -> get_solution
 * once per tick (examples/cartpole_example_mpc.m:36-44); every such tick pays a kernel launch and a stream
 * synchronisation (~10 us of a ~15 us tick). In a session the solve kernel is launched ONCE and stays resident: it polls
 * a mailbox for the next x0 (a line of device memory the host writes through the PCIe BAR; pinned host memory where the device's
 * memory is not host-visible), runs the warm-started solve with the ADMM state kept in registers, sends the first controls to
 * pinned host memory at once -- tinympc_session_step returns with them (a quadrotor N=50 tick: 6 us) -- and then the solution and
 * a completion stamp, which tinympc_get_solution / _get_stats wait for.
 *   tinympc_session_begin   settings, bounds, cache are frozen for the session; references may change between ticks
 *                           (tinympc_set_x_ref / _set_u_ref: picked up by the next step)
 *   tinympc_session_step    x0 in (nx), first controls out (nu); tinympc_get_solution / _get_stats work as usual
 *   tinympc_session_end     stops the kernel; the handle continues with ordinary solves from the same ADMM state.
 * Any other verb that needs the device ends the session implicitly, and so do tinympc_update_settings,
 * tinympc_set_cone_constraints and tinympc_set_linear_constraints (the resident kernel carries the settings and
 * families it was launched with): call tinympc_session_begin again afterwards. The resident kernel leaves on its own
 * after 2 s without a command (a crashed host does not leave it spinning); the next step restarts it transparently.
 * While the resident kernel spins, calls that synchronise the whole DEVICE wait for it -- up to the 2 s idle time-out.
 * This library's own such calls (tinympc_setup / tinympc_reset of ANOTHER handle on the device: hipMalloc / hipFree) first
 * send the resident kernels of the device home; their sessions stay open and the next tinympc_session_step starts the
 * kernel again (a few milliseconds, once). Calls from outside the library (hipDeviceSynchronize, torch.cuda.synchronize,
 * somebody else's hipMalloc) cannot be seen coming: end the session before them. The resident kernel is the variant of the
 * kernel the handle's launches run on (layout F where that is compiled in, was asked for with tinympc_prepare, or carries the cone /
 * linear families; layout C otherwise): its results are identical, bit for bit, to the same ticks issued as tinympc_mpc_step_batch
 * calls. (Only where layout F has no resident kernel for a configuration its launches run on does the session fall back to layout
 * C's, whose results differ from layout F's in the last bits: the carries of its chunks round differently.) */
int tinympc_session_begin(tinympc_solver *s);
int tinympc_session_step(tinympc_solver *s, const double *x0, double *u0_out);
int tinympc_session_end(tinympc_solver *s);
/* Resident solves (an extension, off by default): after tinympc_set_resident(h, 1) the reference's own per-tick sequence
 * tinympc_set_x0 -> tinympc_solve -> tinympc_get_solution / tinympc_get_stats runs on the resident session kernel instead of one launch per
 * solve (a quadrotor N=50 tick: 6 us instead of 15). Results are bit-identical to launched solves. Single-instance handles with nx+nu <= 16
 * and no adaptive rho; elsewhere solves are launched as before. Any verb that needs the device (new bounds, settings, ...) sends the
 * resident kernel home, the next solve starts it again; it also leaves by itself after 2 s without a solve. While it is resident it occupies
 * one compute unit and a DEVICE-wide synchronisation of other code in the process (hipMalloc, hipDeviceSynchronize) waits for it -- up to
 * that idle time-out: the reason this is not the default. enable = 0 ends it. */
int tinympc_set_resident(tinympc_solver *s, int enable);

/* Launch the solve without waiting (same kernel as tinympc_solve); pair with tinympc_synchronize. Every verb
 * that changes an input of the launch in flight (set_x0, mpc_step, ... -- on single-instance handles x0 lives in
 * pinned host memory that the kernel reads directly) waits for the launch first, so the sequence solve_async ->
 * set_x0 (next tick) -> synchronize is safe for every batch size. */
int tinympc_solve_async(tinympc_solver *s);
int tinympc_synchronize(tinympc_solver *s);
/* Synchronous solve that also reports the kernel's duration measured with HIP events recorded on
 * the handle's stream immediately around the launch. */
int tinympc_solve_timed(tinympc_solver *s, float *kernel_ms);

/* Throughput form of the timed solve: queue the launch on the handle's stream and return at once; every queued launch records its
 * own HIP event pair around the solve kernel. tinympc_collect_kernel_ms waits for the stream and returns the kernel durations of
 * the launches queued since the last collect, in order (*count; an error if they exceed `capacity`, at most 4,096). Between two
 * queued solves the caller may queue tinympc_reset_workspace (stream-ordered, returns at once). tinympc_set_x0_batch_device is
 * correct there too but NOT asynchronous: it waits for the handle's stream -- i.e. for every launch queued so far -- before it
 * returns, because the caller may free or reuse d_x0s as soon as it does; a pipeline that wants new x0 per queued solve
 * without draining the queue keeps them in one device array and switches by offset before queueing. Host readers
 * (get_solution ...) synchronise as always. Batched handles outside a session only. */
int tinympc_solve_queued(tinympc_solver *s);
int tinympc_collect_kernel_ms(tinympc_solver *s, float *kernel_ms, int capacity, int *count);

/* One closed-loop control tick for every instance of the handle (the loop of
 * examples/cartpole_example_mpc.m:36-44 / rocket_landing_constraints.m:86-121 as ONE call): upload the
 * measured states x0s (nx x batch), run the warm-started solve, download the first control of each
 * instance into u0_out (nu x batch). One stream submission and one synchronisation instead of three
 * synchronous verbs: up to 256 instances the kernel reads x0 from and writes the first controls to pinned
 * host memory itself (no copy engine involved), larger batches go through two async copies. Results are
 * identical to tinympc_set_x0_batch + tinympc_solve + tinympc_get_first_controls_batch.
 * (Single-instance handles get the same treatment for the plain verbs: tinympc_set_x0 only fills a pinned
 * buffer the next launch reads, and tinympc_get_solution / tinympc_get_stats after a solve are host copies.) */
int tinympc_mpc_step_batch(tinympc_solver *s, const double *x0s, double *u0_out);

/* Launch geometry of the solve kernel, for reports: lanes per instance, instances per wavefront,
 * workgroups in the grid, dynamic LDS bytes per workgroup, and whether the per-knot bound/reference
 * tables are LDS-resident. Any pointer may be NULL. */
int tinympc_get_launch_info(tinympc_solver *s, int *lanes_per_instance, int *instances_per_wave,
                            int *workgroups, int *lds_bytes, int *tables_in_lds);

/* Which solve kernel the handle's next launch uses: 'A' (all ADMM state in LDS, one wavefront per workgroup), 'B' (V
 * L2-resident in HBM, four wavefronts per workgroup), 'C' (one instance per workgroup, horizon swept in 16 concurrent
 * chunks; batches up to 768 and every single solve), 'D' (horizon unrolled at compile time, state in registers; large
 * batches), 'E' (as D with the horizon cut across the wavefronts of a workgroup: cone / linear families at long horizons),
 * 'F' (the latency kernel specialised at run time: up to 32 chunks on two wavefronts per SIMD; the families at small batches),
 * 'M' (64 < nx+nu <= 512 on the FP64 matrix cores). The environment variable TINYMPC_LAYOUT=A|B|C|D|E|F overrides the choice.
 * 0 for a NULL handle. */
int tinympc_get_layout(tinympc_solver *s);

/* Where the kernel of the handle's CURRENT configuration comes from, as text: "compiled-in layout=X", or for the run-time
 * specialised kernels (layouts D and E) "compiled ..." / "disk-cache ..." with the code object's register count, scratch and LDS
 * bytes, or "refused(<reason>)" when the specialiser declined (plan overflow, spills, compile error, TINYMPC_JIT=0) and the
 * launch falls back to a generic kernel. Refusals are also printed once to stderr (TINYMPC_JIT_QUIET=1 silences them).
 * " slot-refill" is appended when the launch uses layout D's slot-refill variant: a batch larger than the device holds at once
 * (8,192 instances of 16 lanes on MI355X) runs as ONE resident set of wavefronts, and a 16-lane row whose instance has finished
 * (converged, or max_iter) is written back and given the next instance of the batch while the rest of its wavefront keeps
 * iterating -- a wavefront then costs the sum of what its rows worked instead of four times its slowest instance. Results are
 * bit-identical to the plain kernel's. Taken whenever the tolerances can be met (with forced iteration counts there is nothing to
 * balance); TINYMPC_REFILL=0 switches it off, =1 takes it for any batch beyond one resident set. */
int tinympc_get_jit_info(tinympc_solver *s, char *buf, int len);

/* Decide (and, where needed, specialise -- seconds the first time) the solve kernel for the handle's CURRENT configuration:
 * bounds / references that vary over the horizon, cone / linear families, adaptive rho and slot refill select variants that are
 * otherwise built at the first launch that needs them. Call it once after the constraints and settings are in place to keep that
 * one-off cost out of the first real-time tick. It also tells the library that run-time specialisation is welcome on this handle:
 * a single instance / small batch of a shape that is not compiled in then runs on the structure-specialised latency kernel (layout
 * F: 5-25 % fewer microseconds per iteration than the generic latency kernel, resident session included) instead of staying on the
 * generic one, which needs no compiler. No reference counterpart (the reference has one code path). */
int tinympc_prepare(tinympc_solver *s);

/* The HIP stream of the handle as an opaque pointer (hipStream_t). */
void *tinympc_get_stream(tinympc_solver *s);

#ifdef __cplusplus
}
#endif
#endif /* TINYMPC_HIP_H */
