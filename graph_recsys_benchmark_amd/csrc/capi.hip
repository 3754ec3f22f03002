// Library-wide entry points of the C ABI (include/peahip.h): version, thread-local error text, device probe.
#include <cstring>

#include "common.h"

namespace pea {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char *get_error() { return g_err; }

}  // namespace pea

extern "C" const char *pea_version(void) { return "peahip 0.1.0 (gfx950)"; }

extern "C" const char *pea_last_error(void) { return pea::get_error(); }

extern "C" int pea_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}
