// BatchNorm (training mode) consumers that finalize the batch statistics themselves.
//
// A training-mode BN forces   conv -> [global reduction] -> normalise -> conv   on the GPU.  With a separate finalize
// kernel that is three launches per BN and every launch in a dependent chain costs ~5 us on MI355X even when it moves
// a few KiB (profiles/r01d_conv_dma_pipeline_ablation.txt: an empty 768-block launch is 5.4 us).  Here the producer
// (conv epilogue / colstats) leaves R <= TF_STAT_ROWS partial rows per BN in a region of its own, and every block of the
// elementwise consumer re-derives the coefficients of ITS 64-channel slice from those rows (R*nk*64 floats out of L2/MALL;
// with 256-channel slices the ~2048 blocks re-read 64 MB and cost more than the finalize launch they replaced) before
// streaming its rows; the blocks with blockIdx.x == 0 also publish what later kernels need (scale/shift/mean/invstd +
// running statistics forward, dgamma/dbeta backward).  Regions are zeroed once per pass.
// Reference semantics: torch.nn.BatchNorm2d in training mode (torchvision resnet101 as used by model.py:17-23).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "tuning.h"

namespace {

constexpr int kSlice = 64;       // channels per block: 128-byte row segments in bf16; the statistic re-read is rows*nk*64 floats per block

struct FwdBn {
  const float* stat;             // [R][2][C]: sum, sum of squares over the batch
  const float* gamma; const float* beta;
  float* scale; float* shift; float* mean; float* invstd; float* rmean; float* rvar;
  const float* sshift;           // [C] what the producer subtracted before summing (NULL: nothing)
};
struct BwdBn {
  const float* stat;             // [R][nk][C]: row 0 = sum gz, row kidx = sum gz*x
  const float* gamma; const float* mean; const float* invstd;
  float* dgamma; float* dbeta;
  int nk, kidx;
};

// same arithmetic as bn_finalize_kernel (elementwise.hip).  RT = rows rounded up to 4/8/16 at compile time so that all
// 2*RT loads are in flight together (a rolled loop pays one L2/MALL round trip per row); rows >= R are read from row R-1
// and multiplied by zero, never out of bounds.
template <int RT>
__device__ __forceinline__ void fwd_table(const FwdBn& d, int R, int C, float count, float eps, float mom, int c0, int cs, float* sc_l,
                                          float* sh_l, bool writer) {
  const int cl = threadIdx.x;
  if (cl >= cs) return;
  const int c = c0 + cl;
  const float ga = d.gamma[c], be = d.beta[c];
  float sv[RT], qv[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int rr = r < R ? r : R - 1;
    const float keep = r < R ? 1.f : 0.f;
    sv[r] = d.stat[(size_t)(rr * 2) * C + c] * keep;
    qv[r] = d.stat[(size_t)(rr * 2 + 1) * C + c] * keep;
  }
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int r = 0; r < RT; ++r) { s += (double)sv[r]; q += (double)qv[r]; }
  // s, q = sums of (x - m0), (x - m0)^2 with the producer's shift m0 (0 without one): var = E[(x-m0)^2] - E[x-m0]^2 cancels only
  // ((mean - m0) / std)^2 of the bits of the fp32 sums instead of (mean / std)^2 (r3; m0 = the BN's running mean in the executor)
  const double m0 = d.sshift ? (double)d.sshift[c] : 0.0;
  const double dm = s / count;
  const double mean = m0 + dm;
  double var = q / count - dm * dm;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = ga * invstd;
  const float sh = be - (float)mean * sc;
  sc_l[cl] = sc; sh_l[cl] = sh;
  if (writer) {
    d.scale[c] = sc; d.shift[c] = sh; d.mean[c] = (float)mean; d.invstd[c] = invstd;
    if (d.rmean) {
      const double unbiased = count > 1.f ? var * count / (count - 1.0) : var;
      d.rmean[c] = (1.f - mom) * d.rmean[c] + mom * (float)mean;
      d.rvar[c] = (1.f - mom) * d.rvar[c] + mom * (float)unbiased;
    }
  }
}

// same arithmetic as bn_bwd_finalize_kernel: g_x = A*gz + B*x + D
template <int RT>
__device__ __forceinline__ void bwd_table(const BwdBn& d, int R, int C, float count, int c0, int cs, float* A_l, float* B_l, float* D_l,
                                          bool writer) {
  const int cl = threadIdx.x;
  if (cl >= cs) return;
  const int c = c0 + cl;
  float av[RT], bv[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int rr = r < R ? r : R - 1;
    const float keep = r < R ? 1.f : 0.f;
    av[r] = d.stat[(size_t)(rr * d.nk) * C + c] * keep;
    bv[r] = d.stat[(size_t)(rr * d.nk + d.kidx) * C + c] * keep;
  }
  const double mu = d.mean[c], is = d.invstd[c], ga = d.gamma[c];
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < RT; ++r) { s1 += (double)av[r]; s2 += (double)bv[r]; }
  const double dg = (s2 - mu * s1) * is;
  const double A = ga * is;
  A_l[cl] = (float)A;
  B_l[cl] = (float)(-A * is * dg / count);
  D_l[cl] = (float)(-A * s1 / count + A * mu * is * dg / count);
  if (writer) {
    if (d.dgamma) d.dgamma[c] = (float)dg;
    if (d.dbeta) d.dbeta[c] = (float)s1;
  }
}

template <int EPS>
__device__ __forceinline__ void lds_coef(const float* p, float (&out)[EPS]) {
#pragma unroll
  for (int j = 0; j < EPS; j += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + j);
    out[j] = v.x; out[j + 1] = v.y; out[j + 2] = v.z; out[j + 3] = v.w;
  }
}

// thread -> (16-byte channel slot inside the slice, row lane); rows advance by gridDim.x * rows-per-pass
template <typename T>
struct RowMap {
  static constexpr int EPS = tf::Elem<T>::kPer16B;
  int c0, cs, slot, rl, rpp;
  __device__ RowMap(int C) {
    cs = C < kSlice ? C : kSlice;
    c0 = blockIdx.y * cs;
    const int spr = cs / EPS;
    slot = threadIdx.x % spr; rl = threadIdx.x / spr; rpp = 256 / spr;
  }
  __device__ size_t off(size_t row, int C) const { return (row * C + c0 + slot * EPS) * sizeof(T); }
};

// y = relu(bn(x))  with the statistics finalized in-kernel
template <typename T, int RT>
__global__ void __launch_bounds__(256) bn_relu_fused_kernel(const T* __restrict__ x, FwdBn d, int R, size_t M, int C, float count, float eps,
                                                            float mom, T* __restrict__ y) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  __shared__ __attribute__((aligned(16))) float sc_l[kSlice], sh_l[kSlice];
  const RowMap<T> m(C);
  const size_t step = (size_t)gridDim.x * m.rpp;
  size_t row = (size_t)blockIdx.x * m.rpp + m.rl;
  // r3: the rows of the first pass are requested BEFORE the coefficient table is derived: the data does not depend on the
  // statistics, so the two memory round trips of the kernel (statistic rows, then data) overlap instead of adding up -- these
  // launches are latency chains, not bandwidth (6.7 us for 12.6 MB; most blocks own exactly one pass)
  const bool h0 = row < M, h1 = row + step < M;
  const uint4 zq = make_uint4(0, 0, 0, 0);
  uint4 p0 = h0 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + m.off(row, C)) : zq;
  uint4 p1 = h1 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + m.off(row + step, C)) : zq;
  fwd_table<RT>(d, R, C, count, eps, mom, m.c0, m.cs, sc_l, sh_l, blockIdx.x == 0);
  __syncthreads();
  float a[EPS], b[EPS];
  lds_coef<EPS>(sc_l + m.slot * EPS, a); lds_coef<EPS>(sh_l + m.slot * EPS, b);
  for (; row + step < M; row += 2 * step) {
    const size_t o0 = m.off(row, C), o1 = m.off(row + step, C);
    const uint4 q0 = p0, q1 = p1;
    if (row + 2 * step < M) p0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + m.off(row + 2 * step, C));
    if (row + 3 * step < M) p1 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + m.off(row + 3 * step, C));
    float f0[EPS], f1[EPS];
    tf::unpack16<T>(q0, f0); tf::unpack16<T>(q1, f1);
#pragma unroll
    for (int j = 0; j < EPS; ++j) { f0[j] = fmaxf(f0[j] * a[j] + b[j], 0.f); f1[j] = fmaxf(f1[j] * a[j] + b[j], 0.f); }
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + o0) = tf::pack16<T>(f0);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + o1) = tf::pack16<T>(f1);
  }
  if (row < M) {
    const size_t o0 = m.off(row, C);
    float f0[EPS];
    tf::unpack16<T>(p0, f0);                            // the tail row is always the pending first prefetch
#pragma unroll
    for (int j = 0; j < EPS; ++j) f0[j] = fmaxf(f0[j] * a[j] + b[j], 0.f);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + o0) = tf::pack16<T>(f0);
  }
}

// y = relu(bn3(x) + (bnd(r) | r))
template <typename T, bool DS, int RT>
__global__ void __launch_bounds__(256) bn_add_relu_fused_kernel(const T* __restrict__ x, FwdBn d1, const T* __restrict__ r, FwdBn d2, int R,
                                                                size_t M, int C, float count, float eps, float mom, T* __restrict__ y) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  __shared__ __attribute__((aligned(16))) float sc1[kSlice], sh1[kSlice], sc2[kSlice], sh2[kSlice];
  const RowMap<T> m(C);
  const size_t step = (size_t)gridDim.x * m.rpp;
  size_t row = (size_t)blockIdx.x * m.rpp + m.rl;
  auto ldq = [&](const T* p, size_t r_) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p) + m.off(r_, C)); };
  const uint4 zq = make_uint4(0, 0, 0, 0);
  // first pass requested before the tables (see bn_relu_fused_kernel)
  uint4 px0 = row < M ? ldq(x, row) : zq, pr0 = row < M ? ldq(r, row) : zq;
  uint4 px1 = row + step < M ? ldq(x, row + step) : zq, pr1 = row + step < M ? ldq(r, row + step) : zq;
  fwd_table<RT>(d1, R, C, count, eps, mom, m.c0, m.cs, sc1, sh1, blockIdx.x == 0);
  if (DS) fwd_table<RT>(d2, R, C, count, eps, mom, m.c0, m.cs, sc2, sh2, blockIdx.x == 0);
  __syncthreads();
  float a1[EPS], b1[EPS], a2[EPS], b2[EPS];
  lds_coef<EPS>(sc1 + m.slot * EPS, a1); lds_coef<EPS>(sh1 + m.slot * EPS, b1);
  if (DS) { lds_coef<EPS>(sc2 + m.slot * EPS, a2); lds_coef<EPS>(sh2 + m.slot * EPS, b2); }
  auto one = [&](const uint4& xq, const uint4& rq, size_t o) {
    float xf[EPS], rf[EPS];
    tf::unpack16<T>(xq, xf); tf::unpack16<T>(rq, rf);
#pragma unroll
    for (int j = 0; j < EPS; ++j) xf[j] = fmaxf(xf[j] * a1[j] + b1[j] + (DS ? rf[j] * a2[j] + b2[j] : rf[j]), 0.f);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + o) = tf::pack16<T>(xf);
  };
  for (; row + step < M; row += 2 * step) {
    const size_t o0 = m.off(row, C), o1 = m.off(row + step, C);
    const uint4 x0 = px0, x1 = px1, r0 = pr0, r1 = pr1;
    if (row + 2 * step < M) { px0 = ldq(x, row + 2 * step); pr0 = ldq(r, row + 2 * step); }
    if (row + 3 * step < M) { px1 = ldq(x, row + 3 * step); pr1 = ldq(r, row + 3 * step); }
    one(x0, r0, o0); one(x1, r1, o1);
  }
  if (row < M) one(px0, pr0, m.off(row, C));
}

// out = A*g' + B*x + D,  g' = g*(y>0) when MASK;  coefficients from the BN-backward sums, finalized in-kernel
template <typename T, bool MASK, int RT>
__global__ void __launch_bounds__(256) bn_bwd_apply_fused_kernel(const T* __restrict__ g, const T* __restrict__ y, const T* __restrict__ x, BwdBn d,
                                                                 int R, size_t M, int C, float count, T* __restrict__ out) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  __shared__ __attribute__((aligned(16))) float A_l[kSlice], B_l[kSlice], D_l[kSlice];
  const RowMap<T> m(C);
  const size_t step = (size_t)gridDim.x * m.rpp;
  const uint4 z = make_uint4(0, 0, 0, 0);
  size_t row = (size_t)blockIdx.x * m.rpp + m.rl;
  auto ldr = [&](const T* p, size_t r_) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p) + m.off(r_, C)); };
  // first pass requested before the table (see bn_relu_fused_kernel): 93 of these launches sit on the data-gradient chain
  const bool h0 = row < M, h1 = row + step < M;
  uint4 pg0 = h0 ? ldr(g, row) : z, px0 = h0 ? ldr(x, row) : z, py0 = (MASK && h0) ? ldr(y, row) : z;
  uint4 pg1 = h1 ? ldr(g, row + step) : z, px1 = h1 ? ldr(x, row + step) : z, py1 = (MASK && h1) ? ldr(y, row + step) : z;
  bwd_table<RT>(d, R, C, count, m.c0, m.cs, A_l, B_l, D_l, blockIdx.x == 0);
  __syncthreads();
  float A[EPS], B[EPS], D[EPS];
  lds_coef<EPS>(A_l + m.slot * EPS, A); lds_coef<EPS>(B_l + m.slot * EPS, B); lds_coef<EPS>(D_l + m.slot * EPS, D);
  auto one = [&](const uint4& gq, const uint4& yq, const uint4& xq, size_t o) {
    float gf[EPS], xf[EPS], yf[EPS];
    tf::unpack16<T>(gq, gf); tf::unpack16<T>(xq, xf);
    if (MASK) {
      tf::unpack16<T>(yq, yf);
#pragma unroll
      for (int j = 0; j < EPS; ++j) if (!(yf[j] > 0.f)) gf[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPS; ++j) gf[j] = A[j] * gf[j] + B[j] * xf[j] + D[j];
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + o) = tf::pack16<T>(gf);
  };
  for (; row + step < M; row += 2 * step) {
    const size_t o0 = m.off(row, C), o1 = m.off(row + step, C);
    const uint4 g0 = pg0, g1 = pg1, x0 = px0, x1 = px1, y0 = py0, y1 = py1;
    if (row + 2 * step < M) { pg0 = ldr(g, row + 2 * step); px0 = ldr(x, row + 2 * step); if (MASK) py0 = ldr(y, row + 2 * step); }
    if (row + 3 * step < M) { pg1 = ldr(g, row + 3 * step); px1 = ldr(x, row + 3 * step); if (MASK) py1 = ldr(y, row + 3 * step); }
    one(g0, y0, x0, o0); one(g1, y1, x1, o1);
  }
  if (row < M) one(pg0, py0, px0, m.off(row, C));
}

inline dim3 fused_grid(int64_t M, int C, int dtype) {
  const int eps = dtype == TF_BF16 ? 8 : 4;
  const int cs = C < kSlice ? C : kSlice;
  const int rpp = 256 / (cs / eps);
  const int slices = C / cs;
  int64_t gx = (M + rpp - 1) / rpp;
  // r6: a block re-derives the BN coefficients of its channel slice from the statistic rows (rows x 2..3 x C floats + gamma / beta: 16+ KB at C = 256)
  // before it touches its 8-row passes of 4 KB: with ~2k blocks the coefficient reads outweighed a layer-3 tensor itself.  Two caps, by tensor size.
  const double bytes = (double)M * C * (dtype == TF_F32 ? 4 : 2);
  const int total_cap = bytes < tf::tuning().ew_small_mb * 1048576.0 ? tf::tuning().ew_blocks_small : tf::tuning().ew_blocks;
  const int64_t cap = std::max(1, total_cap / slices);   // ~2k blocks: enough loads in flight, bounded coefficient re-derivation
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)slices);
}

inline bool fused_shape_ok(int C, int dtype) {
  const int eps = dtype == TF_BF16 ? 8 : 4;
  if (C % eps) return false;
  const int cs = C < kSlice ? C : kSlice;
  return C % cs == 0 && 256 % (cs / eps) == 0;
}

}  // namespace

#define DISPATCH_T1(dtype, ...)                                 \
  do {                                                          \
    if ((dtype) == TF_BF16) { using T = tf::bf16_t; __VA_ARGS__; } \
    else if ((dtype) == TF_F32) { using T = float; __VA_ARGS__; }  \
    else return TF_ERR_UNSUPPORTED;                             \
  } while (0)
// T from dtype, RT (compile-time row bound) from rows
#define DISPATCH_T(dtype, ...)                                                        \
  do {                                                                                \
    if (rows <= 4) { constexpr int RT = 4; DISPATCH_T1(dtype, __VA_ARGS__); }         \
    else if (rows <= 8) { constexpr int RT = 8; DISPATCH_T1(dtype, __VA_ARGS__); }    \
    else { constexpr int RT = 16; DISPATCH_T1(dtype, __VA_ARGS__); }                  \
  } while (0)

extern "C" int tf_bn_relu_fused(int dtype, const void* x, const tf_bn_fwd_desc* bn, int rows, int64_t M, int C, float count, float eps,
                                float momentum, void* y, void* stream) {
  if (!x || !y || !bn || !bn->stat || !bn->gamma || !bn->beta || !bn->scale || !bn->shift || !bn->mean || !bn->invstd) return TF_ERR_ARG;
  if (rows < 1 || rows > TF_STAT_ROWS || !fused_shape_ok(C, dtype)) return TF_ERR_ARG;
  const FwdBn d{bn->stat, bn->gamma, bn->beta, bn->scale, bn->shift, bn->mean, bn->invstd, bn->running_mean, bn->running_var, bn->stat_shift};
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_fused_kernel<T, RT>), fused_grid(M, C, dtype), dim3(256), 0, (hipStream_t)stream, (const T*)x, d, rows,
                                       (size_t)M, C, count, eps, momentum, (T*)y));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_add_relu_fused(int dtype, const void* x, const tf_bn_fwd_desc* bn, const void* r, const tf_bn_fwd_desc* bn_r, int rows,
                                    int64_t M, int C, float count, float eps, float momentum, void* y, void* stream) {
  if (!x || !r || !y || !bn || !bn->stat || !bn->gamma || !bn->beta || !bn->scale || !bn->shift || !bn->mean || !bn->invstd) return TF_ERR_ARG;
  if (bn_r && (!bn_r->stat || !bn_r->gamma || !bn_r->beta || !bn_r->scale || !bn_r->shift || !bn_r->mean || !bn_r->invstd)) return TF_ERR_ARG;
  if (rows < 1 || rows > TF_STAT_ROWS || !fused_shape_ok(C, dtype)) return TF_ERR_ARG;
  const FwdBn d1{bn->stat, bn->gamma, bn->beta, bn->scale, bn->shift, bn->mean, bn->invstd, bn->running_mean, bn->running_var, bn->stat_shift};
  FwdBn d2 = d1;
  if (bn_r) d2 = FwdBn{bn_r->stat, bn_r->gamma, bn_r->beta, bn_r->scale, bn_r->shift, bn_r->mean, bn_r->invstd, bn_r->running_mean, bn_r->running_var, bn_r->stat_shift};
  if (bn_r) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_add_relu_fused_kernel<T, true, RT>), fused_grid(M, C, dtype), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                         d1, (const T*)r, d2, rows, (size_t)M, C, count, eps, momentum, (T*)y));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_add_relu_fused_kernel<T, false, RT>), fused_grid(M, C, dtype), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                         d1, (const T*)r, d2, rows, (size_t)M, C, count, eps, momentum, (T*)y));
  }
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_bwd_apply_fused(int dtype, const void* g, const void* y, const void* x, const tf_bn_bwd_desc* bn, int rows, int64_t M, int C,
                                     float count, void* out, void* stream) {
  if (!g || !x || !out || !bn || !bn->stat || !bn->gamma || !bn->mean || !bn->invstd) return TF_ERR_ARG;
  if (rows < 1 || rows > TF_STAT_ROWS || bn->nk < 2 || bn->kidx < 1 || bn->kidx >= bn->nk || !fused_shape_ok(C, dtype)) return TF_ERR_ARG;
  const BwdBn d{bn->stat, bn->gamma, bn->mean, bn->invstd, bn->dgamma, bn->dbeta, bn->nk, bn->kidx};
  if (y) {
    DISPATCH_T(dtype, TF_LAUNCH_WITH_STOP_EVENT((bn_bwd_apply_fused_kernel<T, true, RT>), fused_grid(M, C, dtype), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)g, (const T*)y, (const T*)x, d, rows, (size_t)M, C, count, (T*)out));
  } else {
    DISPATCH_T(dtype, TF_LAUNCH_WITH_STOP_EVENT((bn_bwd_apply_fused_kernel<T, false, RT>), fused_grid(M, C, dtype), dim3(256), 0, (hipStream_t)stream,
                                                (const T*)g, (const T*)y, (const T*)x, d, rows, (size_t)M, C, count, (T*)out));
  }
  TF_CHECK_LAUNCH();
  return TF_OK;
}
