// Error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <utility>

#include "rumi_orb.h"

namespace rumi {
extern thread_local std::string g_lastError;
void set_error(const char *fmt, const char *a, const char *b, int line);

// The opt-in for more than 64 KiB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) is state of the FUNCTION, shared by every thread and
// stream of the process.  Handles on several host threads launch the same kernels with different sizes, so the limit only ever grows: a thread
// with a smaller problem must not lower it between another thread's request and launch.  One entry per (function, device).
inline hipError_t raise_lds_limit(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> cur;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    size_t &c = cur[{fn, dev}];
    if (bytes <= c) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) c = bytes;
    return e;
}
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
