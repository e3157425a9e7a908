// One error policy for every facade class (ORBextractor, ORBmatcher, Optimizer, FrameFrustum, ORBVocabulary, Sim3Scoring, Sim3Solver).
//
// The reference's members have no error channel beyond their return values, and a SLAM node must not be killed from inside a library:
//   * every C-ABI call that does not return RUMI_OK is REPORTED (call site, status, rumi_last_error()) -- to stderr by default, to the
//     handler a maintainer installs with rumi_facade::set_error_handler otherwise (raise the node's own alarm, count, abort, ...);
//   * the member then returns the reference's own failure value (-1 matches / 0 inliers / plain return), never a silently degraded result
//     without a report, and never std::abort();
//   * RUMI_E_CAPACITY from a per-thread arena (more features / map points / edges than it was created for) is not an error yet: the arena
//     is re-created twice as large and the call repeated (up to 16x the initial size);
//   * rumi_facade::last_status() is the calling thread's most recent non-OK status (RUMI_OK after clear_status()), for call sites that
//     want to tell "no matches" from "no GPU".
#pragma once
#include <cstdio>

#include "rumi_orb.h"

namespace rumi_facade {

using ErrorHandler = void (*)(const char *where, int status, const char *message);

inline ErrorHandler &error_handler_slot() { static ErrorHandler h = nullptr; return h; }
inline void set_error_handler(ErrorHandler h) { error_handler_slot() = h; }
inline int &status_slot() { thread_local int s = RUMI_OK; return s; }
inline int last_status() { return status_slot(); }
inline void clear_status() { status_slot() = RUMI_OK; }

inline void report(const char *where, int status, const char *message = nullptr) {
    status_slot() = status;
    const char *msg = message ? message : rumi_last_error();
    if (error_handler_slot()) error_handler_slot()(where, status, msg ? msg : "");
    else std::fprintf(stderr, "[rumi] %s: status %d: %s\n", where, status, msg ? msg : "");
}

// Runs `call` (a C-ABI call returning a status); on RUMI_E_CAPACITY grows the arena through `grow` and repeats once per doubling.
template <class Call, class Grow> inline int guarded(const char *where, Grow grow, Call call) {
    int rc = call();
    while (rc == RUMI_E_CAPACITY && grow()) rc = call();
    if (rc != RUMI_OK) report(where, rc);
    return rc;
}
inline bool no_growth() { return false; }

}  // namespace rumi_facade

#define RUMI_GUARDED(where, grow, call) ::rumi_facade::guarded(where, grow, [&]() -> int { return (call); })
