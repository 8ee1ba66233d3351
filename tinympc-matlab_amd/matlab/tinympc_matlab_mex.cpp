// tinympc_matlab_mex.cpp -- MEX shim: tinympc_matlab('<verb>', args...) over the C ABI of libtinympc_hip.so.
//
// Drop-in replacement for the reference's MEX function (/root/reference/src/bindings.cpp): same
// function name, same 17 verbs, same argument order per verb, same outputs, same error identifiers.
// It contains no numerics: every verb unpacks mxArrays (column-major doubles, exactly what the C ABI
// takes) and forwards to one tinympc_* call on a process-global handle (the reference keeps a
// process-global solver too, bindings.cpp:17).
//
// Build (where MATLAB exists):  mex -I<repo>/include tinympc_matlab_mex.cpp -L<libdir> -ltinympc_hip -output tinympc_matlab
// In this repository it is compile-checked against a minimal mock of mex.h (tests/mock_mex/).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"
#include "tinympc_hip.h"

namespace {

tinympc_solver *g_handle = nullptr;  // bindings.cpp:17
int g_nx = 0, g_nu = 0, g_N = 0;

const char *mex_id_for(int code) {
    switch (code) {
        case TINYMPC_ERR_INVALID_INPUT: return "TinyMPC:InvalidInput";
        case TINYMPC_ERR_NOT_INITIALIZED: return "TinyMPC:NotInitialized";
        case TINYMPC_ERR_NOT_IMPLEMENTED: return "TinyMPC:InvalidFunction";
        case TINYMPC_ERR_UNSUPPORTED:
        case TINYMPC_ERR_NO_DEVICE:
        case TINYMPC_ERR_ALLOC: return "TinyMPC:SetupFailed";
        default: return "TinyMPC:Exception";
    }
}

void check(int code, const char *override_id = nullptr) {
    if (code != TINYMPC_OK) mexErrMsgIdAndTxt(override_id ? override_id : mex_id_for(code), "%s", tinympc_last_error());
}

void need_args(int nrhs, int expected, const char *verb) {
    if (nrhs != expected)
        mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "%s requires %d input argument%s", verb, expected, expected == 1 ? "" : "s");
}

void need_solver() {
    if (!g_handle) mexErrMsgIdAndTxt("TinyMPC:NotInitialized", "Solver not initialized");
}

const double *real_doubles(const mxArray *a) {
    if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "Input must be a real double array");
    return mxGetPr(a);
}

int as_int(const mxArray *a) { return (int)mxGetScalar(a); }

std::vector<int> index_vector(const mxArray *a) {  // int32 or double index vectors (bindings.cpp:433-448)
    const size_t n = mxGetM(a) * mxGetN(a);
    std::vector<int> out(n);
    if (mxIsInt32(a)) {
        const int *p = static_cast<const int *>(mxGetData(a));
        for (size_t i = 0; i < n; ++i) out[i] = p[i];
    } else if (mxIsDouble(a)) {
        const double *p = mxGetPr(a);
        for (size_t i = 0; i < n; ++i) out[i] = (int)std::lround(p[i]);
    } else {
        mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "Input must be int32 or double array");
    }
    return out;
}

mxArray *matrix_out(int rows, int cols) { return mxCreateDoubleMatrix((mwSize)rows, (mwSize)cols, mxREAL); }

// ---- verbs ------------------------------------------------------------------------------------

void verb_setup(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {  // A,B,fdyn,Q,R,rho,nx,nu,N,verbose
    if (nrhs != 10)
        mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "setup requires 10 input arguments: A, B, fdyn, Q, R, rho, nx, nu, N, verbose");
    const double *A = real_doubles(prhs[0]), *B = real_doubles(prhs[1]), *f = real_doubles(prhs[2]);  // (f: NULL when empty, below)
    const double *Q = real_doubles(prhs[3]), *R = real_doubles(prhs[4]);
    const double rho = mxGetScalar(prhs[5]);
    const int nx = as_int(prhs[6]), nu = as_int(prhs[7]), N = as_int(prhs[8]), verbose = as_int(prhs[9]);
    // The C ABI reads nx*nx, nx*nu, ... doubles behind these pointers: the arrays must really have that shape
    // (the reference gets the same guarantee from its Eigen conversions and tiny_setup's dimension checks).
    const struct { const mxArray *a; int rows, cols; const char *name; } shapes[] = {
        {prhs[0], nx, nx, "A"}, {prhs[1], nx, nu, "B"}, {prhs[3], nx, nx, "Q"}, {prhs[4], nu, nu, "R"}};
    for (const auto &sh : shapes)
        if ((int)mxGetM(sh.a) != sh.rows || (int)mxGetN(sh.a) != sh.cols)
            mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "setup: %s is %dx%d, expected %dx%d", sh.name, (int)mxGetM(sh.a),
                              (int)mxGetN(sh.a), sh.rows, sh.cols);
    const size_t nf = mxGetM(prhs[2]) * mxGetN(prhs[2]);
    if (nf != 0 && nf != (size_t)nx)
        mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "setup: fdyn has %d entries, expected 0 or %d", (int)nf, nx);
    if (nf == 0) f = nullptr;
    if (g_handle) tinympc_reset(&g_handle, 0);  // setup replaces the global solver (bindings.cpp:92)
    check(tinympc_setup(&g_handle, A, B, f, Q, R, rho, nx, nu, N, verbose), "TinyMPC:SetupFailed");
    g_nx = nx; g_nu = nu; g_N = N;
    plhs[0] = mxCreateDoubleScalar(0);
}

void verb_set_x0(int, mxArray *[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 2, "set_x0");
    need_solver();
    const int len = (int)(mxGetM(prhs[0]) * mxGetN(prhs[0]));
    check(tinympc_set_x0(g_handle, real_doubles(prhs[0]), len, as_int(prhs[1])), "TinyMPC:SetX0Failed");
}

void verb_set_x_ref(int, mxArray *[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 2, "set_x_ref");
    need_solver();
    check(tinympc_set_x_ref(g_handle, real_doubles(prhs[0]), (int)mxGetM(prhs[0]), (int)mxGetN(prhs[0]), as_int(prhs[1])),
          "TinyMPC:SetXRefFailed");
}

void verb_set_u_ref(int, mxArray *[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 2, "set_u_ref");
    need_solver();
    check(tinympc_set_u_ref(g_handle, real_doubles(prhs[0]), (int)mxGetM(prhs[0]), (int)mxGetN(prhs[0]), as_int(prhs[1])),
          "TinyMPC:SetURefFailed");
}

void verb_set_bound_constraints(int, mxArray *[], int nrhs, const mxArray *prhs[]) {  // x_min,x_max,u_min,u_max,verbose
    need_solver();
    if (nrhs < 5) mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "set_bound_constraints requires 5 input arguments");
    const mxArray *const a[4] = {prhs[0], prhs[1], prhs[2], prhs[3]};
    const int rows[4] = {g_nx, g_nx, g_nu, g_nu}, cols[4] = {g_N, g_N, g_N - 1, g_N - 1};
    for (int i = 0; i < 4; ++i)
        if ((int)mxGetM(a[i]) != rows[i] || (int)mxGetN(a[i]) != cols[i])
            mexErrMsgIdAndTxt("TinyMPC:SetBoundConstraintsFailed", "bound array %d is %dx%d, expected %dx%d", i + 1,
                              (int)mxGetM(a[i]), (int)mxGetN(a[i]), rows[i], cols[i]);
    check(tinympc_set_bound_constraints(g_handle, real_doubles(a[0]), real_doubles(a[1]), real_doubles(a[2]), real_doubles(a[3]),
                                        as_int(prhs[4])),
          "TinyMPC:SetBoundConstraintsFailed");
}

void verb_solve(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 1, "solve");
    need_solver();
    check(tinympc_solve(g_handle, as_int(prhs[0])));
    plhs[0] = mxCreateDoubleScalar(0);  // the reference always returns 0 here (bindings.cpp:230)
}

// Extension (include/tinympc_hip.h): take the kernel-variant decision before the first solve.
void verb_prepare(int, mxArray *[], int, const mxArray *[]) {
    need_solver();
    check(tinympc_prepare(g_handle));
}

// Closed-loop session (extension, include/tinympc_hip.h): the solve kernel stays resident between ticks.

void verb_session_begin(int, mxArray *[], int, const mxArray *[]) {
    need_solver();
    check(tinympc_session_begin(g_handle));
}

void verb_session_step(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {  // x0 -> first controls (nu x 1)
    need_args(nrhs, 1, "session_step");
    need_solver();
    if ((int)(mxGetM(prhs[0]) * mxGetN(prhs[0])) != g_nx)
        mexErrMsgIdAndTxt("TinyMPC:SetX0Failed", "session_step: x0 has %d entries, expected %d", (int)(mxGetM(prhs[0]) * mxGetN(prhs[0])), g_nx);
    plhs[0] = matrix_out(g_nu, 1);
    check(tinympc_session_step(g_handle, real_doubles(prhs[0]), mxGetPr(plhs[0])));
}

void verb_session_end(int, mxArray *[], int, const mxArray *[]) {
    need_solver();
    check(tinympc_session_end(g_handle));
}

void verb_set_resident(int, mxArray *[], int nrhs, const mxArray *prhs[]) {  // on/off: solve() on the resident session kernel (extension)
    need_args(nrhs, 1, "set_resident");
    need_solver();
    check(tinympc_set_resident(g_handle, as_int(prhs[0])));
}

void verb_get_solution(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 1, "get_solution");
    need_solver();
    plhs[0] = matrix_out(g_nx, g_N);
    plhs[1] = matrix_out(g_nu, g_N - 1);
    check(tinympc_get_solution(g_handle, mxGetPr(plhs[0]), mxGetPr(plhs[1]), as_int(prhs[0])));
}

void verb_get_stats(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 1, "get_stats");
    need_solver();
    int iter = 0, status = 0;
    double ps = 0, pi = 0;
    check(tinympc_get_stats(g_handle, &iter, &status, &ps, &pi, as_int(prhs[0])));
    plhs[0] = mxCreateDoubleScalar(iter);
    plhs[1] = mxCreateDoubleScalar(status);
    plhs[2] = mxCreateDoubleScalar(ps);
    plhs[3] = mxCreateDoubleScalar(pi);
}

void verb_codegen(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 2, "codegen");
    need_solver();
    char *dir = mxArrayToString(prhs[0]);
    const int status = tinympc_codegen(g_handle, dir, as_int(prhs[1]));
    mxFree(dir);
    plhs[0] = mxCreateDoubleScalar(status);  // non-zero: TinyMPC.m raises TinyMPC:CodegenFailed
}

void verb_codegen_with_sensitivity(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 6, "codegen_with_sensitivity");
    need_solver();
    char *dir = mxArrayToString(prhs[0]);
    const int status = tinympc_codegen_with_sensitivity(g_handle, dir, real_doubles(prhs[1]), real_doubles(prhs[2]),
                                                        real_doubles(prhs[3]), real_doubles(prhs[4]), as_int(prhs[5]));
    mxFree(dir);
    plhs[0] = mxCreateDoubleScalar(status);
}

void verb_set_sensitivity_matrices(int, mxArray *[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 5, "set_sensitivity_matrices");
    need_solver();
    check(tinympc_set_sensitivity_matrices(g_handle, real_doubles(prhs[0]), real_doubles(prhs[1]), real_doubles(prhs[2]),
                                           real_doubles(prhs[3]), as_int(prhs[4])));
}

void verb_set_cache_terms(int, mxArray *[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 5, "set_cache_terms");
    need_solver();
    check(tinympc_set_cache_terms(g_handle, real_doubles(prhs[0]), real_doubles(prhs[1]), real_doubles(prhs[2]),
                                  real_doubles(prhs[3]), as_int(prhs[4])));
}

// The three verbs below have no counterpart in bindings.cpp: the reference runs these recursions in MATLAB
// inside TinyMPC.m (:194-241, :336-366); this build's TinyMPC.m forwards them to the device instead.
void cache_outputs(mxArray *plhs[]) {
    plhs[0] = matrix_out(g_nu, g_nx);
    plhs[1] = matrix_out(g_nx, g_nx);
    plhs[2] = matrix_out(g_nu, g_nu);
    plhs[3] = matrix_out(g_nx, g_nx);
}

void verb_compute_cache_terms(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {  // -> Kinf,Pinf,Quu_inv,AmBKt
    need_args(nrhs, 1, "compute_cache_terms");
    need_solver();
    cache_outputs(plhs);
    check(tinympc_compute_cache_terms(g_handle, mxGetPr(plhs[0]), mxGetPr(plhs[1]), mxGetPr(plhs[2]), mxGetPr(plhs[3]),
                                      nullptr, as_int(prhs[0])));
}

void verb_solve_lqr(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {  // rho_val -> K,P,C1,C2
    need_args(nrhs, 2, "solve_lqr");
    need_solver();
    cache_outputs(plhs);
    check(tinympc_solve_lqr(g_handle, mxGetScalar(prhs[0]), mxGetPr(plhs[0]), mxGetPr(plhs[1]), mxGetPr(plhs[2]),
                            mxGetPr(plhs[3]), nullptr));
}

void verb_compute_sensitivity(int, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {  // -> dK,dP,dC1,dC2
    need_args(nrhs, 1, "compute_sensitivity");
    need_solver();
    cache_outputs(plhs);
    check(tinympc_compute_sensitivity(g_handle, mxGetPr(plhs[0]), mxGetPr(plhs[1]), mxGetPr(plhs[2]), mxGetPr(plhs[3]),
                                      as_int(prhs[0])));
}

void verb_set_linear_constraints(int, mxArray *[], int nrhs, const mxArray *prhs[]) {  // Alin_x, blin_x, Alin_u, blin_u
    need_solver();
    if (nrhs < 4) mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "set_linear_constraints requires 4 input arguments");
    const int nlx = (int)(mxGetM(prhs[1]) * mxGetN(prhs[1])), nlu = (int)(mxGetM(prhs[3]) * mxGetN(prhs[3]));
    check(tinympc_set_linear_constraints(g_handle, nlx ? real_doubles(prhs[0]) : nullptr, nlx ? real_doubles(prhs[1]) : nullptr, nlx,
                                         nlu ? real_doubles(prhs[2]) : nullptr, nlu ? real_doubles(prhs[3]) : nullptr, nlu),
          "TinyMPC:SetLinearConstraintsFailed");
}

void verb_set_cone_constraints(int, mxArray *[], int nrhs, const mxArray *prhs[]) {  // Acx,qcx,cx,Acu,qcu,cu (state first)
    need_solver();
    if (nrhs < 6) mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "set_cone_constraints requires 6 input arguments");
    const std::vector<int> Acx = index_vector(prhs[0]), qcx = index_vector(prhs[1]);
    const std::vector<int> Acu = index_vector(prhs[3]), qcu = index_vector(prhs[4]);
    const int ncx = (int)Acx.size(), ncu = (int)Acu.size();
    if (qcx.size() != Acx.size() || mxGetM(prhs[2]) * mxGetN(prhs[2]) != Acx.size() || qcu.size() != Acu.size() ||
        mxGetM(prhs[5]) * mxGetN(prhs[5]) != Acu.size())
        mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "set_cone_constraints: Ac, qc and c of one side must have the same length");
    check(tinympc_set_cone_constraints(g_handle, Acx.data(), qcx.data(), ncx ? real_doubles(prhs[2]) : nullptr, ncx, Acu.data(),
                                       qcu.data(), ncu ? real_doubles(prhs[5]) : nullptr, ncu),
          "TinyMPC:SetConeConstraintsFailed");
}

void verb_reset(int, mxArray *[], int nrhs, const mxArray *prhs[]) {
    need_args(nrhs, 1, "reset");
    check(tinympc_reset(&g_handle, as_int(prhs[0])));
}

void verb_update_settings(int, mxArray *[], int nrhs, const mxArray *prhs[]) {  // 14 scalars + verbose
    need_args(nrhs, 15, "update_settings");
    need_solver();
    check(tinympc_update_settings(g_handle, mxGetScalar(prhs[0]), mxGetScalar(prhs[1]), as_int(prhs[2]), as_int(prhs[3]),
                                  as_int(prhs[4]), as_int(prhs[5]), as_int(prhs[6]), as_int(prhs[7]), as_int(prhs[8]),
                                  as_int(prhs[9]), as_int(prhs[10]), mxGetScalar(prhs[11]), mxGetScalar(prhs[12]),
                                  as_int(prhs[13]), as_int(prhs[14])));
}

void verb_print_problem_data(int, mxArray *[], int nrhs, const mxArray *[]) {
    need_args(nrhs, 0, "print_problem_data");
    need_solver();
    check(tinympc_print_problem_data(g_handle));
}

struct Verb {
    const char *name;
    void (*fn)(int, mxArray *[], int, const mxArray *[]);
};

const Verb kVerbs[] = {
    {"setup", verb_setup}, {"set_x0", verb_set_x0}, {"set_x_ref", verb_set_x_ref}, {"set_u_ref", verb_set_u_ref},
    {"solve", verb_solve}, {"get_solution", verb_get_solution}, {"get_stats", verb_get_stats}, {"codegen", verb_codegen},
    {"reset", verb_reset}, {"set_bound_constraints", verb_set_bound_constraints},
    {"set_sensitivity_matrices", verb_set_sensitivity_matrices}, {"set_cache_terms", verb_set_cache_terms},
    {"codegen_with_sensitivity", verb_codegen_with_sensitivity}, {"update_settings", verb_update_settings},
    {"print_problem_data", verb_print_problem_data}, {"set_linear_constraints", verb_set_linear_constraints},
    {"set_cone_constraints", verb_set_cone_constraints},
    {"compute_cache_terms", verb_compute_cache_terms}, {"solve_lqr", verb_solve_lqr},
    {"compute_sensitivity", verb_compute_sensitivity},
    {"prepare", verb_prepare}, {"session_begin", verb_session_begin}, {"session_step", verb_session_step}, {"session_end", verb_session_end}, {"set_resident", verb_set_resident},
};

}  // namespace

extern "C" void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 1) mexErrMsgIdAndTxt("TinyMPC:InvalidInput", "At least one input argument required");
    char *verb = mxArrayToString(prhs[0]);
    const std::string name(verb ? verb : "");
    mxFree(verb);
    for (const Verb &v : kVerbs) {
        if (name == v.name) {
            v.fn(nlhs, plhs, nrhs - 1, prhs + 1);
            return;
        }
    }
    mexErrMsgIdAndTxt("TinyMPC:InvalidFunction", "Unknown function: %s", name.c_str());
}
