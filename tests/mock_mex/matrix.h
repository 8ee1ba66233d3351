/* tests/mock_mex/mex.h -- minimal stand-in for MATLAB's mex.h / matrix.h, TEST INFRASTRUCTURE ONLY.
 * Just enough of the MEX C API for tinympc-matlab_amd/matlab/tinympc_matlab_mex.cpp (this repo's own
 * shim) to compile and be driven verb by verb from tests/test_mex_shim.py. mexErrMsgIdAndTxt throws a
 * C++ exception that mock_mex_call() converts into a return code + recorded identifier/message
 * (in MATLAB it long-jumps back into the interpreter). */
#ifndef MOCK_MEX_H
#define MOCK_MEX_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef struct mxArray_tag mxArray;

mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateInt32Matrix(mwSize m, mwSize n);   /* mock-only convenience */
mxArray *mxCreateString(const char *s);
void mxDestroyArray(mxArray *a);
int mxIsDouble(const mxArray *a);
int mxIsComplex(const mxArray *a);
int mxIsInt32(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
double mxGetScalar(const mxArray *a);
char *mxArrayToString(const mxArray *a);
void mxFree(void *p);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
int mexPrintf(const char *fmt, ...);

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
/* mock driver: returns 0, or 1 when mexErrMsgIdAndTxt fired (see mock_mex_last_id/msg) */
int mock_mex_call(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
const char *mock_mex_last_id(void);
const char *mock_mex_last_msg(void);
#ifdef __cplusplus
}
#endif
#endif
