// tests/mock_mex/mock_mex.cpp -- implementation of the mock MEX API (TEST INFRASTRUCTURE ONLY).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
    int kind;  // 0 double, 1 int32, 2 char
    size_t m, n;
    std::vector<double> d;
    std::vector<int> i;
    std::string s;
};

namespace {
std::string g_id, g_msg;
struct MexError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
}  // namespace

extern "C" {
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) {
    mxArray *a = new mxArray_tag{0, m, n, std::vector<double>(m * n ? m * n : 1, 0.0), {}, {}};
    return a;
}
mxArray *mxCreateDoubleScalar(double v) {
    mxArray *a = mxCreateDoubleMatrix(1, 1, mxREAL);
    a->d[0] = v;
    return a;
}
mxArray *mxCreateInt32Matrix(mwSize m, mwSize n) { return new mxArray_tag{1, m, n, {}, std::vector<int>(m * n ? m * n : 1, 0), {}}; }
mxArray *mxCreateString(const char *s) { return new mxArray_tag{2, 1, strlen(s), {}, {}, std::string(s)}; }
void mxDestroyArray(mxArray *a) { delete a; }
int mxIsDouble(const mxArray *a) { return a->kind == 0; }
int mxIsComplex(const mxArray *) { return 0; }
int mxIsInt32(const mxArray *a) { return a->kind == 1; }
size_t mxGetM(const mxArray *a) { return a->m; }
size_t mxGetN(const mxArray *a) { return a->n; }
double *mxGetPr(const mxArray *a) { return const_cast<double *>(a->d.data()); }
void *mxGetData(const mxArray *a) { return a->kind == 1 ? (void *)a->i.data() : (void *)a->d.data(); }
double mxGetScalar(const mxArray *a) { return a->kind == 1 ? (double)a->i[0] : a->d[0]; }
char *mxArrayToString(const mxArray *a) {
    char *p = (char *)malloc(a->s.size() + 1);
    memcpy(p, a->s.c_str(), a->s.size() + 1);
    return p;
}
void mxFree(void *p) { free(p); }
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_id = id;
    g_msg = buf;
    throw MexError(buf);
}
int mexPrintf(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    int n = vprintf(fmt, ap);
    va_end(ap);
    return n;
}
int mock_mex_call(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    g_id.clear();
    g_msg.clear();
    try {
        mexFunction(nlhs, plhs, nrhs, prhs);
    } catch (const MexError &) {
        return 1;
    }
    return 0;
}
const char *mock_mex_last_id(void) { return g_id.c_str(); }
const char *mock_mex_last_msg(void) { return g_msg.c_str(); }
}
