// tools/asan_host_stubs.cpp -- the two host helpers of tinympc_capi.hip (tinympc_host.h) for the sanitizer build of the
// host-only emitter (tools/asan_check.py): tinympc_codegen.hip is compiled there with g++ -fsanitize=address,undefined,
// without the rest of the library (which needs hipcc and the HIP runtime).
#include <cstdarg>
#include <cstdio>
#include <string>

#include "tinympc_host.h"

namespace tinympc {
std::string &last_error_slot() {
    thread_local std::string slot;
    return slot;
}
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}
}  // namespace tinympc
extern "C" const char *tinympc_last_error(void) { return tinympc::last_error_slot().c_str(); }
