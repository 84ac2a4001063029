// Host-side helpers for the C ABI entry points (argument checks, launch status).
#pragma once

#include <hip/hip_runtime.h>

#include "mvster_hip.h"

namespace mvster {

inline hipStream_t as_stream(mvster_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipGetLastError() after a launch also clears the sticky state so that one failed call does not
// poison the next one made from the same host thread.
inline int launch_status() { return hipGetLastError() == hipSuccess ? MVSTER_OK : MVSTER_ERR_LAUNCH; }

inline int grid_for(long long items, int per_block) { return (int)((items + per_block - 1) / per_block); }

// Largest plane (elements) addressable with the 32-bit in-plane offsets the kernels use.
constexpr long long kMaxPlane = 1LL << 30;

}  // namespace mvster

#define MVSTER_REQUIRE_PTR(p) \
    do {                      \
        if ((p) == nullptr) return MVSTER_ERR_NULL_POINTER; \
    } while (0)

#define MVSTER_REQUIRE_SHAPE(cond) \
    do {                           \
        if (!(cond)) return MVSTER_ERR_BAD_SHAPE; \
    } while (0)
