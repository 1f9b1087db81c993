// Shared device helpers for the tiny-faces gfx950 kernels (CDNA4, wave64).
#pragma once
// TF_EXPERIMENTAL (build.py --experimental / TINYFACES_BUILD_EXPERIMENTAL=1): the kernels that were built to parity, measured and LOST stay in the tree for
// re-measurement but are not part of the default library: conv_pwx (BatchNorm prologues: tf_conv2d_bnbwd / tf_conv2d_bnfwd), the output-channel slices of
// conv_pws and the in-LDS BN prologue of the ring-less pointwise kernel (tf_conv_args.bnf).  DESIGN.md section 7 rows 14-16, 44, 45, 54.
#ifdef TF_EXPERIMENTAL
#define TF_EXP 1
#else
#define TF_EXP 0
#endif
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/tinyfaces_hip.h"

#define TF_LAUNCH_WITH_STOP_EVENT(kernel, grid, block, lds, stream, ...)                                                  \
  do {                                                                                                                  \
    hipEvent_t ev__ = tf::take_next_stop_event();                                                                       \
    if (ev__) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ev__, 0, __VA_ARGS__);                   \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                             \
  } while (0)

// A launch under a kernel-mode tf::ProfScope (profile.h): the dispatch carries the scope's two events when it is sampled; a fork's
// completion event (TF_LAUNCH_WITH_STOP_EVENT semantics) takes precedence and the scope falls back to a bracket.
#define TF_LAUNCH_TIMED(kernel, grid, block, lds, stream, ...)                                                            \
  do {                                                                                                                  \
    hipEvent_t f__ = tf::take_next_stop_event(), a__ = nullptr, b__ = nullptr;                                          \
    if (f__) {                                                                                                          \
      tf::ProfScope::fall_back_to_bracket();                                                                            \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, f__, 0, __VA_ARGS__);                            \
    } else if (tf::ProfScope::take_launch_events(&a__, &b__)) {                                                         \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, a__, b__, 0, __VA_ARGS__);                                \
    } else {                                                                                                            \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                \
    }                                                                                                                   \
  } while (0)

#define TF_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return TF_ERR_LAUNCH;           \
  } while (0)

namespace tf {

// Completion event of the NEXT kernel launched through TF_LAUNCH_WITH_STOP_EVENT on this host thread (one-shot).  The executor
// uses it to hand a producer kernel's OWN completion signal to the weight-gradient stream (hipExtLaunchKernelGGL stopEvent): a
// separate hipEventRecord costs a barrier packet -- an ~8 us bubble on the data-gradient chain, 94 times per training step
// (profiles/r02_step_timeline.txt).
void set_next_stop_event(hipEvent_t e);
hipEvent_t take_next_stop_event();

constexpr int kWave = 64;

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// float -> bfloat16, round-to-nearest-even (same as torch's float->bfloat16; NaN stays NaN): the gfx950 conversion instruction
// v_cvt_pk_bf16_f32 (two values per instruction).  The integer sequence it replaces (add 0x7fff + lsb, NaN test) cost ~6 VALU
// operations and, through the NaN select, a pair of exec-mask updates PER ELEMENT in every epilogue.
typedef __bf16 tf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(tf_f32x2{lo, hi}, tf_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPer16B = 4;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
struct bf16_t { uint16_t v; };
template <> struct Elem<bf16_t> {
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(p->v); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
};

// IEEE half (configs[4]: fp16 MFMA operands for the hard-setting evaluation): round-to-nearest-even, overflow -> inf like torch
struct f16_t { uint16_t v; };
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) { return (uint32_t)f32_to_f16(lo) | ((uint32_t)f32_to_f16(hi) << 16); }
template <> struct Elem<f16_t> {
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float load(const f16_t* p) { return f16_to_f32(p->v); }
  __device__ static __forceinline__ void store(f16_t* p, float v) { p->v = f32_to_f16(v); }
};
// two floats -> one 32-bit word of the 2-byte storage type
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }

// unpack a 16-byte register into floats and back
template <typename T> __device__ __forceinline__ void unpack16(const uint4& r, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& r, float* f) {
  f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& r, float* f) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& r, float* f) {
  f[0] = f16_to_f32((uint16_t)r.x); f[1] = f16_to_f32((uint16_t)(r.x >> 16));
  f[2] = f16_to_f32((uint16_t)r.y); f[3] = f16_to_f32((uint16_t)(r.y >> 16));
  f[4] = f16_to_f32((uint16_t)r.z); f[5] = f16_to_f32((uint16_t)(r.z >> 16));
  f[6] = f16_to_f32((uint16_t)r.w); f[7] = f16_to_f32((uint16_t)(r.w >> 16));
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
  return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float* f) {
  return make_uint4(pack_f16x2(f[0], f[1]), pack_f16x2(f[2], f[3]), pack_f16x2(f[4], f[5]), pack_f16x2(f[6], f[7]));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// PRECONDITION of lane_xor_sum / lane_group_sum / wave_sum(float): ALL 64 lanes active (full exec mask).  A DPP row rotation with
// bound_ctrl = false contributes 0 for a disabled source lane and the row swaps read whatever an inactive lane's register holds, so
// a divergent caller gets a silently wrong sum.  Every call site is wave-uniform (criterion.hip, the statistic epilogues of
// conv_dma / conv3x3h, behind block-uniform conditions); a divergent caller must use the __shfl_xor form (wave_sum(double) below).
// Their summation order differs from the xor butterfly, i.e. results differ from it at ulp level.
// v + (v of lane ^ O) for O = 8, 16, 32 without the LDS crossbar (`__shfl_xor` is ds_bpermute_b32: an LDS instruction plus an
// lgkmcnt wait per level): a DPP row rotation folded into the add (8), and the gfx950 row-swap instructions
// v_permlane16_swap / v_permlane32_swap (with both operands = v they leave "my half" and "the other half" in the two results).
template <int O> __device__ __forceinline__ float lane_xor_sum(float v) {
  if constexpr (O == 8) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  } else if constexpr (O == 16) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
  } else if constexpr (O == 32) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
  } else {
    return v + __shfl_xor(v, O, 64);
  }
}
// sum over the lanes that share (lane % FROM): the levels FROM, 2*FROM, ... 32
template <int FROM> __device__ __forceinline__ float lane_group_sum(float v) {
  if constexpr (FROM <= 32) { v = lane_xor_sum<FROM>(v); if constexpr (FROM < 32) v = lane_group_sum<FROM * 2>(v); }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  // inside a 16-lane row: four rotate-and-add steps (every lane ends with the row total), then the two row swaps
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return lane_group_sum<16>(v);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// counter-based RNG (splitmix64 finaliser over a 4-word key) -> uniform [0,1) double / u32
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __host__ __forceinline__ uint64_t hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  return mix64(mix64(mix64(mix64(a) ^ b) ^ c) ^ d);
}
__device__ __host__ __forceinline__ double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }


// Function attributes (hipFuncSetAttribute: dynamic LDS above 64 KiB) and the CU count belong to a DEVICE, not to the process: a host
// that drives more than one GPU from one process sets them once per device.  `PerDevice` is the flag array of one launch site.
struct PerDevice {
  bool done[64] = {};
  // true exactly once per device (the caller then sets its attributes on the current device)
  bool first() { int dev = 0; (void)hipGetDevice(&dev); bool& d = done[dev & 63]; const bool f = !d; d = true; return f; }
};
inline int device_cus() {
  static int cus[64] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  int& n = cus[dev & 63];
  if (n == 0) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; n = v; }
  return n;
}

}  // namespace tf
