// wfa_fast.hpp — register-resident short-read kernel (placeholder until the general path is verified).
#pragma once
#include <hip/hip_runtime.h>
#include "wfa_common.hpp"

namespace wfa {

inline bool fast_supported(const WfaDevConfig&, int, bool) { return false; }

inline int launch_fast(const WfaDevConfig&, int, hipStream_t, const uint32_t*, const WfaPairMeta*, const uint32_t*,
                       uint32_t, int32_t*, int32_t*, uint32_t*, uint32_t*) {
  return 0;
}

}  // namespace wfa
