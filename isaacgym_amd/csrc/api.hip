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

// 100 * major + 10 * minor.  1.2 added igx_motion_fetch_amp_obs, igx_motion_amp_obs_history and
// igx_extract_positions_from_rigid_body_states (entry points only: every earlier signature and struct is unchanged).
extern "C" int igx_version(void) { return 120; }

extern "C" const char *igx_last_error_string(void) { return igx::error_buffer(); }
