// Optional per-launch HIP-event timing of the MFMA kernels (bench.py's `roofline` object).
// Disabled by default: zero overhead unless tf_profile_enable(1) was called.
#pragma once
#include <hip/hip_runtime.h>

namespace tf {
struct ProfScope {
  int slot;
  hipStream_t stream;
  ProfScope(int kind, double flops, double bytes, hipStream_t s);   // records the start event when profiling is on
  ~ProfScope();                                                     // records the stop event
};
}  // namespace tf
