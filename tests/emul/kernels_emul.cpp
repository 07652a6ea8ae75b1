// TEST INFRASTRUCTURE: harness of the CPU build (tests/emul/build_emul.py appends one #include per transformed kernel file): the three entry points of
// error.hip the bindings expect of any libmarius_hip.
#include <algorithm>
#include <numeric>

#include "common.h"

extern "C" const char* marius_hip_last_error(void) { return marius::g_last_error; }
extern "C" int marius_hip_abi_version(void) { return MARIUS_HIP_ABI_VERSION; }
extern "C" int marius_config_reload(void) {
    marius::g_env = marius::read_env();
    return MARIUS_OK;
}

