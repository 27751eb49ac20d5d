// capi.hip — library-level pieces of the C ABI (version, thread-local error string).
#include <stdarg.h>

#include "jm_common.h"

namespace jm {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace jm

extern "C" int jm_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* jm_last_error(void) { return jm::g_err; }
