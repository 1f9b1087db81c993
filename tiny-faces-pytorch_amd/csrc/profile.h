// Optional per-launch HIP-event timing of the MFMA kernels (bench.py's `roofline` object).
// Disabled by default: zero overhead unless tf_profile_enable(n) was called (n = 1: every launch, n > 1: every n-th launch;
// an event pair costs ~5 us of stream time, so bench.py samples 1 launch in 7 to keep the timed region undisturbed).
#pragma once
#include <hip/hip_runtime.h>

namespace tf {
struct ProfScope {
  int slot;
  hipStream_t stream;
  // shape: optional GEMM view of the launch (M = pixels, N = output channels, K = reduction length, taps, mode, epilogue flags)
  // so that tf_profile_shapes can aggregate per layer shape, not only per kernel kind
  ProfScope(int kind, double flops, double bytes, hipStream_t s, int M = 0, int N = 0, int K = 0, int taps = 0, int mode = 0, int epi = 0,
            double exec_flops = -1.0);   // exec_flops: what the kernel executed (padding, zero taps); < 0: same as flops
  ~ProfScope();                                                     // records the stop event
};
}  // namespace tf
