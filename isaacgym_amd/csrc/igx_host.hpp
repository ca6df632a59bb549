// Host-side helpers shared by the C-ABI entry points of libigx.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/igx.h"

namespace igx {

// Last error text of the calling thread; written by every failing entry point.
char *error_buffer();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Converts the status of the launch that was just enqueued into the ABI's
// return convention (0 or the positive hipError_t) without synchronising.
int launch_status(const char *what);

// Integer launch-tuning knob: environment variable `name` if set, else `dflt`.
int tuning_int(const char *name, int dflt);

}  // namespace igx

#define IGX_REQUIRE_PTR(ptr, name)                                              \
    do {                                                                        \
        if ((ptr) == nullptr) return igx::fail(IGX_E_NULL, "%s: %s is NULL", __func__, name); \
    } while (0)

#define IGX_REQUIRE_ALIGNED(ptr, name)                                          \
    do {                                                                        \
        if (!igx::aligned16(ptr))                                               \
            return igx::fail(IGX_E_ALIGN, "%s: %s must be 16-byte aligned", __func__, name); \
    } while (0)
