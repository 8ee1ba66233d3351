/* tinympc_hip_bench.h -- measurement helpers and diagnostics of libtinympc_hip. NOT part of the drop-in boundary
 * (include/tinympc_hip.h = the MEX verbs of /root/reference/src/bindings.cpp:641-692 + the batched extensions): nothing a user of the
 * solver needs is declared here. bench.py, tools/ and the tests that check the measurement itself are the only callers.
 *
 *   tinympc_bench_closed_loop      libtinympc_bench.so  (tinympc-matlab_amd/csrc/bench/tinympc_bench_loop.cpp: public verbs only)
 *   tinympc_debug_tick_timing      libtinympc_hip.so    (reads the handle's diagnostic counters)
 *   tinympc_debug_setup_timing     libtinympc_hip.so
 *   tinympc_debug_mail_stamp       libtinympc_hip.so    (the session mailbox's line stamp, for the tests of its checksum)
 */
#ifndef TINYMPC_HIP_BENCH_H
#define TINYMPC_HIP_BENCH_H

#include "tinympc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* `ticks` closed-loop ticks of a single-instance handle (of an nx-state, nu-input system) driven from C (x0 in, warm-started solve, first controls out, plant step
 * x+ = A x + B u0 + f; f may be NULL) through tinympc_mpc_step_batch (session == 0) or the session the caller has opened
 * (session == 1), or -- session == 2 -- the reference's own three verbs per tick, tinympc_set_x0 + tinympc_solve + tinympc_get_solution (whole
 * solution copied out; launched solves, or resident ones after tinympc_set_resident): what a caller written in C pays per tick, next to the reference core timed the same way (oracle/ref_shim.cpp:
 * ref_bench_closed_loop -- /root/reference/examples/cartpole_example_mpc.m:36-44 is the loop). Only the tick verb is timed; the first
 * `skip` ticks are not counted (0 <= skip < ticks); *seconds and *iterations are sums over the counted ticks, tick_us (may be NULL)
 * receives every tick's duration. x is advanced in place. */
int tinympc_bench_closed_loop(tinympc_solver *s, int nx, int nu, int N, const double *A, const double *B, const double *f, double *x, int ticks, int skip, int session,
                              double *seconds, long *iterations, double *tick_us);

/* Where the last zero-copy tick spent its time: launch us, wait us, polls, 1 if the polling budget ran out. */
int tinympc_debug_tick_timing(tinympc_solver *s, double *out4);

/* Host microseconds of the phases of the handle's tinympc_setup_batch: [0] device / stream / events / layout decisions, [1] the device
 * arena (one hipMalloc), [2] the pinned arena (one hipHostMalloc + clearing it), [3] staging the problem data and queueing the upload,
 * the memset and the fills, [4] queueing k_precompute, [5] waiting for the stream, [6] the whole call; then, read back from the device:
 * [7] shader clocks and [8] microseconds (100 MHz counter) of the Riccati loop inside k_precompute_rows (0 where another precompute
 * kernel ran), [9] the Riccati steps taken. */
int tinympc_debug_setup_timing(tinympc_solver *s, double *out10);

/* The stamp the session protocol puts behind a 64-byte line's seven payload words (tinympc_device.h: mail_stamp over mail_mix): sequence
 * number + a 16-bit order-dependent checksum of the words and of the sequence number. No handle, no GPU. */
double tinympc_debug_mail_stamp(double sequence_number, const double *words7);

#ifdef __cplusplus
}
#endif
#endif
