// Library identification and per-thread error reporting of libigx.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "igx_host.hpp"

namespace igx {

char *error_buffer()
{
    static thread_local char buf[512] = "no error";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(error_buffer(), 512, "%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return IGX_OK;
}

int tuning_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    return (end && *end == '\0') ? (int)x : dflt;
}

}  // namespace igx

extern "C" int igx_version(void) { return 110; }

extern "C" const char *igx_last_error_string(void) { return igx::error_buffer(); }
