// Optional per-launch HIP-event timing of the MFMA kernels (bench.py's `roofline` object).
// Disabled by default: zero overhead unless tf_profile_enable(n) was called (n = 1: every launch, n > 1: every n-th launch;
// an event pair costs ~5 us of stream time, so bench.py samples 1 launch in 7 to keep the timed region undisturbed).
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
