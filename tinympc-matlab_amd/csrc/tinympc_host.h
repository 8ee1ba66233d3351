// tinympc_host.h -- host-side helpers shared by the translation units behind the C ABI.
#pragma once
#include <string>

namespace tinympc {

// Per-thread message behind tinympc_last_error() (defined in tinympc_capi.hip).
std::string &last_error_slot();

// Record a printf-style message and return `code` (so call sites read `return fail(code, ...)`).
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace tinympc
