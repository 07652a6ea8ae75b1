// Error string + ABI version for libmarius_hip.so.
#include "common.h"

namespace marius {
static thread_local char g_last_error[512] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}
}  // namespace marius

extern "C" int marius_hip_abi_version(void) { return 1; }
extern "C" const char* marius_hip_last_error(void) { return marius::g_last_error; }
