// tinympc_solve_dr.hip -- the slot-refill variant of layout D's compiled-in kernels (k_admm_solve_d_refill, launch_solve_d_refill):
// tinympc_solve_d.hip with TINY_REFILL set. A translation unit of its own so that the plain kernels' text -- and with it their
// code -- is exactly what it is without the variant (see the note at the top of tinympc_solve_d.hip).
#define TINY_REFILL 1
#include "tinympc_solve_d.hip"
