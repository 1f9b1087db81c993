// Optional per-launch HIP-event timing of the MFMA kernels (bench.py's `roofline` object).
// Disabled by default: zero overhead unless tf_profile_enable(n) was called (n = 1: every launch, n > 1: every n-th launch;
// an event pair costs ~5 us of stream time, so bench.py samples 1 launch in 7 to keep the timed region undisturbed).
#pragma once
#include <hip/hip_runtime.h>

namespace tf {
// Two ways to time a sampled launch:
//  * bracket (rounds 1-2, still the default of a scope): hipEventRecord before and after -> record, dispatch, kernel, record: the ~5-7 us of
//    inter-packet latency of the queue are inside the interval, and the two event packets perturb the stream;
//  * kernel (r3, `kernel_mode = true` + TF_LAUNCH_TIMED at the launch site): the launch itself carries both events
//    (hipExtLaunchKernelGGL startEvent / stopEvent), the interval is the dispatch's own begin / end time stamp -- what rocprofv3 reports
//    for the same kernel -- and no packet is added to the stream.  A launch that also carries a fork's completion event (conv_pwx) falls
//    back to the bracket.  TINYFACES_PROFILE_BRACKET=1 forces the bracket everywhere (A/B of the two clocks).
struct ProfScope {
  int slot;
  hipStream_t stream;
  int state;        // 0 inactive, 1 bracket open, 2 kernel events pending, 3 kernel events handed to the launch
  // shape: optional GEMM view of the launch (M = pixels, N = output channels, K = reduction length, taps, mode, epilogue flags)
  // so that tf_profile_shapes can aggregate per layer shape, not only per kernel kind
  ProfScope(int kind, double flops, double bytes, hipStream_t s, int M = 0, int N = 0, int K = 0, int taps = 0, int mode = 0, int epi = 0,
            double exec_flops = -1.0, bool kernel_mode = false);   // exec_flops: what the kernel executed (padding, zero taps); < 0: same as flops
  ~ProfScope();                                                     // bracket: records the stop event
  void begin_bracket();                                             // a pending kernel-mode scope turns into a bracket from here on
  static bool take_launch_events(hipEvent_t* a, hipEvent_t* b);     // the innermost pending kernel-mode scope of this thread, once
  static void fall_back_to_bracket();                               // the launch cannot carry the events: bracket it instead
};
}  // namespace tf
