// Error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <functional>
#include <string>

#include "rumi_orb.h"

namespace rumi {
extern thread_local std::string g_lastError;
void set_error(const char *fmt, const char *a, const char *b, int line);
}  // namespace rumi

// Any failing HIP call ends the C-ABI function with RUMI_E_NO_DEVICE and a message for rumi_last_error().
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            rumi::set_error("HIP error: %s -> %s (line %d)", #expr, hipGetErrorString(e_), __LINE__); \
            return RUMI_E_NO_DEVICE;                                                           \
        }                                                                                      \
    } while (0)
