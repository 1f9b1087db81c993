// Weight gradient of a convolution as a split-K MFMA GEMM whose reduction dimension is the
// PIXEL index:  dW[co][tap][ci] = sum_p dY[p][co] * Xhat[gather(p,tap)][ci].
// (autograd backward of the convs of tinyfaces/models/model.py:90-106, triggered at
//  tinyfaces/trainer.py:86.)
//
// NHWC keeps channels contiguous, so the reduction index is the STRIDED one for both operands.
// gfx950 has exactly the tool for this: ds_read_b64_tr_b16 -- a 16-lane group reads a
// [4 pixel][16 channel] block that was stored pixel-major and each lane receives the 4 pixels
// of ITS channel.  Tiles are therefore staged exactly as they sit in HBM (coalesced 16-byte
// loads, 16-byte LDS writes) and transposed for free on the read.  Row pitch is padded by
// 32 bytes and the k<->pixel assignment of a k-step is (h*16 + g*4 + j) so that the two
// lane groups served per LDS cycle hit 8 distinct 32-byte bank windows.
// The f32 instantiation (parity path) reads scalars (ds_read_b32) for v_mfma_f32_16x16x4_f32.
// Split-K partial tiles are combined with fp32 atomics straight into the OIHW gradient
// (coalesced 64-byte runs along ci for 1x1 convs).
#include "common.h"
#include "tuning.h"
#include "profile.h"

int tf_wgrad_dma_launch(const tf_wgrad_args* a, hipStream_t stream);
int tf_wgrad3x3_launch(const tf_wgrad_args* a, hipStream_t stream);       // wgrad3x3.hip   // wgrad_dma.hip
int tf_wgrad3x3_group_launch(const tf_wgrad_args* a, int n, hipStream_t stream);       // wgrad3x3.hip
int tf_wgrad_pw_group_launch(const tf_wgrad_args* a, int n, hipStream_t stream);        // wgrad_group.hip

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgK {
  const char* x; const char* dy; float* dw;
  const float* pro_scale; const float* pro_shift;
  int H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int M, OHW, ldx, lddy, dw_ld, pro_relu, ci_stride, tap_stride;
  int nco, nci, ntaps, splitk, chunk;
};

template <typename T> struct WgTraits;
template <> struct WgTraits<tf::bf16_t> { static constexpr int PK = 64; };
template <> struct WgTraits<float> { static constexpr int PK = 32; };

// fragment of 16 channels x 32 pixels (one k-step) from a pixel-major LDS tile
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int pitch, int pk0, int c0) {
  const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
  const char* p = tile + (pk0 + g * 4 + (i >> 2)) * pitch + (c0 + (i & 3) * 4) * 2;
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * pitch));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename T, int BC>   // BC = channels per tile side (co and ci): 128 or 64
__global__ void __launch_bounds__(256) wgrad_kernel(const WgK a) {
  constexpr int PK = WgTraits<T>::PK;
  constexpr int EPS = tf::Elem<T>::kPer16B;
  constexpr int ROWB = BC * (int)sizeof(T);           // payload bytes per pixel row
  constexpr int PITCH = ROWB + 32;
  constexpr int SPR = ROWB / 16;                      // 16-byte slots per row
  constexpr int RPP = 256 / SPR;                      // rows per pass
  constexpr int NL = PK / RPP;                        // loads per thread per operand
  constexpr int TILEB = PK * PITCH;
  constexpr int WC = BC / 2, NF = WC / 16;            // wave tile: WC x WC channels
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* pro_sc = reinterpret_cast<float*>(smem + 4 * TILEB);
  float* pro_sh = pro_sc + a.Cin;

  int b = blockIdx.x;
  const int ks = b % a.splitk; b /= a.splitk;
  const int tci = b % a.nci; b /= a.nci;
  const int tco = b % a.nco; b /= a.nco;
  const int tap = b;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int co0 = tco * BC, ci0 = tci * BC;
  const int pbeg = ks * a.chunk, pend = min(a.M, pbeg + a.chunk);
  const int tid = threadIdx.x, slot = tid % SPR, lrow = tid / SPR;

  if (a.pro_scale) {
    for (int c = tid; c < a.Cin; c += 256) { pro_sc[c] = a.pro_scale[c]; pro_sh[c] = a.pro_shift[c]; }
  }

  uint4 yr[NL], xr[NL];
  unsigned okmask = 0;
  auto issue = [&](int p0) {
    okmask = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int p = p0 + lrow + i * RPP;
      yr[i] = make_uint4(0, 0, 0, 0); xr[i] = make_uint4(0, 0, 0, 0);
      if (p < pend) {
        if (co0 + slot * EPS + EPS <= a.lddy)
          yr[i] = *reinterpret_cast<const uint4*>(a.dy + ((size_t)p * a.lddy + co0 + slot * EPS) * sizeof(T));
        const int n = p / a.OHW, rem = p - n * a.OHW, oh = rem / a.OW, ow = rem - oh * a.OW;
        const int ih = oh * a.stride - a.pad + kh, iw = ow * a.stride - a.pad + kw;
        if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && ci0 + slot * EPS + EPS <= a.ldx) {
          xr[i] = *reinterpret_cast<const uint4*>(a.x + (((size_t)n * a.H * a.W + (size_t)ih * a.W + iw) * a.ldx + ci0 + slot * EPS) * sizeof(T));
          okmask |= 1u << i;
        }
      }
    }
  };
  auto commit = [&](int buf) {
    char* ys = smem + buf * 2 * TILEB;
    char* xs = ys + TILEB;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      uint4 v = xr[i];
      if (a.pro_scale && ((okmask >> i) & 1u)) {
        float f[EPS];
        tf::unpack16<T>(v, f);
        const int c = ci0 + slot * EPS;
#pragma unroll
        for (int j = 0; j < EPS; ++j) {
          float t = f[j] * pro_sc[c + j] + pro_sh[c + j];
          f[j] = a.pro_relu ? fmaxf(t, 0.f) : t;
        }
        v = tf::pack16<T>(f);
      }
      const int off = (lrow + i * RPP) * PITCH + slot * 16;
      *reinterpret_cast<uint4*>(ys + off) = yr[i];
      *reinterpret_cast<uint4*>(xs + off) = v;
    }
  };

  f32x4 acc[NF][NF];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < NF; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave = tid >> 6, wco = wave & 1, wci = wave >> 1;
  const int l = tid & 63, li = l & 15, lg = l >> 4;

  const int nst = pend > pbeg ? (pend - pbeg + PK - 1) / PK : 0;
  if (nst > 0) issue(pbeg);
  __syncthreads();
  if (nst > 0) commit(0);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) issue(pbeg + (st + 1) * PK);
    const char* ys = smem + buf * 2 * TILEB;
    const char* xs = ys + TILEB;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int k0 = 0; k0 < PK; k0 += 32) {
        bf16x8 fy[NF], fx[NF];
#pragma unroll
        for (int n = 0; n < NF; ++n) fy[n] = frag_tr(ys, PITCH, k0, wco * WC + n * 16);
#pragma unroll
        for (int m = 0; m < NF; ++m) fx[m] = frag_tr(xs, PITCH, k0, wci * WC + m * 16);
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < NF; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[n], fx[m], acc[n][m], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < PK / 4; ++kk) {
        float fy[NF], fx[NF];
        const int row = (kk * 4 + lg) * PITCH;
#pragma unroll
        for (int n = 0; n < NF; ++n) fy[n] = *reinterpret_cast<const float*>(ys + row + (wco * WC + n * 16 + li) * 4);
#pragma unroll
        for (int m = 0; m < NF; ++m) fx[m] = *reinterpret_cast<const float*>(xs + row + (wci * WC + m * 16 + li) * 4);
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
          for (int m = 0; m < NF; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[n], fx[m], acc[n][m], 0, 0, 0);
      }
    }
    if (st + 1 < nst) commit(buf ^ 1);
    __syncthreads();
  }
  if (nst == 0) return;
  // D[co][ci]: lane holds co = (l>>4)*4 + r, ci = l&15
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < NF; ++m) {
      const int ci = ci0 + wci * WC + m * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wco * WC + n * 16 + lg * 4 + r;
        if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + (size_t)co * a.dw_ld + (size_t)ci * a.ci_stride + (size_t)tap * a.tap_stride, acc[n][m][r]);
      }
    }
}

template <typename T, int BC>
int launch_wgrad(const tf_wgrad_args* A, hipStream_t stream) {
  constexpr int PK = WgTraits<T>::PK;
  WgK k;
  k.x = (const char*)A->x; k.dy = (const char*)A->dy; k.dw = A->dw_oihw; k.pro_scale = A->pro_scale; k.pro_shift = A->pro_shift;
  k.H = A->H; k.W = A->W; k.Cin = A->Cin; k.OH = A->OH; k.OW = A->OW; k.Cout = A->Cout; k.KH = A->KH; k.KW = A->KW;
  k.stride = A->stride; k.pad = A->pad; k.M = A->N * A->OH * A->OW; k.OHW = A->OH * A->OW;
  k.ldx = A->ldx; k.lddy = A->lddy; k.dw_ld = A->dw_ld; k.pro_relu = A->pro_relu;
  if (A->packed) { k.ci_stride = 1; k.tap_stride = A->Cin; } else { k.ci_stride = A->KH * A->KW; k.tap_stride = 1; }
  k.nco = (A->Cout + BC - 1) / BC; k.nci = (A->Cin + BC - 1) / BC; k.ntaps = A->KH * A->KW;
  const int tiles = k.nco * k.nci * k.ntaps;
  int sk = A->splitk;
  if (sk <= 0) {
    sk = (640 + tiles - 1) / tiles;                         // ~640 blocks measured best on the layer shapes (scripts/microbench.py)
    const int maxsk = (k.M + 4 * PK - 1) / (4 * PK);        // at least 4 stages per block
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
  }
  k.chunk = (((k.M + sk - 1) / sk) + PK - 1) / PK * PK;
  k.splitk = (k.M + k.chunk - 1) / k.chunk;
  constexpr int PITCH = BC * (int)sizeof(T) + 32;
  const size_t lds = (size_t)4 * PK * PITCH + (A->pro_scale ? (size_t)A->Cin * 8 : 0);
  static tf::PerDevice attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const double es = sizeof(T), Md = k.M;
  tf::ProfScope prof(8 + (sizeof(T) == 2 ? 2 : 0) + (BC == 128 ? 1 : 0), 2.0 * Md * A->Cout * A->Cin * k.ntaps,
                     (Md * A->Cout + (double)A->N * A->H * A->W * A->Cin) * es + (double)A->Cout * A->Cin * k.ntaps * 4, stream);
  hipLaunchKernelGGL((wgrad_kernel<T, BC>), dim3(tiles * k.splitk), dim3(256), lds, stream, k);
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

}  // namespace

extern "C" int tf_conv2d_wgrad(const tf_wgrad_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !a->x || !a->dy || !a->dw_oihw) return TF_ERR_ARG;
  if (a->dtype != TF_BF16 && a->dtype != TF_F32) return TF_ERR_UNSUPPORTED;
  const int eps = a->dtype == TF_BF16 ? 8 : 4;
  if (a->ldx % eps || a->lddy % eps || a->ldx < a->Cin || a->lddy < a->Cout) return TF_ERR_ARG;   // 16-byte slots
  if (a->pro_scale && (!a->pro_shift || a->Cin % eps)) return TF_ERR_ARG;
  // tile: 0 = auto (bf16, no prologue: the all-taps kernel for 3x3 / stride 1 / pad 1, else the per-tap LDS-DMA pipeline),
  //       1 = force the per-tap DMA kernel, 3 = force the all-taps kernel, 64 / 128 = register-staged kernel
  if ((a->tile == 0 || a->tile == 3) && a->dtype == TF_BF16 && !a->pro_scale) {
    const bool w3_off = tf::tuning().wgrad3_off;          // A/B knob
    const int rc = (w3_off && a->tile == 0) ? TF_ERR_UNSUPPORTED : tf_wgrad3x3_launch(a, stream);
    if (rc != TF_ERR_UNSUPPORTED || a->tile == 3) return rc;
  }
  if ((a->tile == 0 || a->tile == 1) && a->dtype == TF_BF16 && !a->pro_scale) return tf_wgrad_dma_launch(a, stream);
  if (a->tile == 1 || a->tile == 3) return TF_ERR_UNSUPPORTED;
  const bool small = a->tile ? a->tile == 64 : true;   // 64x64 tiles: 4x fewer split-K partials per MFMA flop than 128x128
  if (a->dtype == TF_BF16) return small ? launch_wgrad<tf::bf16_t, 64>(a, stream) : launch_wgrad<tf::bf16_t, 128>(a, stream);
  return small ? launch_wgrad<float, 64>(a, stream) : launch_wgrad<float, 128>(a, stream);
}

// r4: a GROUP of weight gradients in one launch, every output tile reduced over all pixels in-block (no split-K, no atomics: dw is
// OVERWRITTEN).  All problems pointwise (any channel counts, one pixel count), or all the same 3x3 / stride 1 / pad 1 shape.
extern "C" int tf_conv2d_wgrad_group(const tf_wgrad_args* probs, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!probs || n <= 0) return TF_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    const tf_wgrad_args& a = probs[i];
    if (!a.x || !a.dy || !a.dw_oihw) return TF_ERR_ARG;
    if (a.dtype != TF_BF16 || a.pro_scale) return TF_ERR_UNSUPPORTED;
    if (a.ldx % 8 || a.lddy % 8 || a.ldx < a.Cin || a.lddy < a.Cout) return TF_ERR_ARG;
    if (a.KH != probs[0].KH || a.KW != probs[0].KW) return TF_ERR_UNSUPPORTED;
  }
  if (probs[0].KH == 1 && probs[0].KW == 1) return tf_wgrad_pw_group_launch(probs, n, stream);
  if (probs[0].KH == 3 && probs[0].KW == 3) return tf_wgrad3x3_group_launch(probs, n, stream);
  return TF_ERR_UNSUPPORTED;
}

namespace {
__global__ void unpack_dw_kernel(const float* __restrict__ p, int Cout, int Cin, int taps, float* __restrict__ out) {
  const size_t total = (size_t)Cout * Cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const int ci = (int)((i / taps) % Cin);
    const int co = (int)(i / ((size_t)taps * Cin));
    out[i] = p[((size_t)co * taps + tap) * Cin + ci];
  }
}
}  // namespace
extern "C" int tf_unpack_dw(const float* packed, int Cout, int Cin, int taps, float* dw_oihw, void* stream) {
  if (!packed || !dw_oihw) return TF_ERR_ARG;
  const size_t total = (size_t)Cout * Cin * taps;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpack_dw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed, Cout, Cin, taps, dw_oihw);
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}
