// Convolution as MFMA implicit GEMM on NHWC activations (gfx950, wave64).
// Replaces the nn.Conv2d / BatchNorm2d / ReLU / residual-add chain of
// tinyfaces/models/model.py:90-106 and the torchvision Bottleneck (see tinyfaces_hip.h).
//
// GEMM view:  D[channel][pixel] = sum_k W[channel][k] * X[pixel][k],  k = (tap, cin).
//   * Both operands are K-contiguous (weights pre-packed [Cout][taps][Cin]; NHWC pixels), so
//     every MFMA fragment is one 16-byte ds_read_b128 per lane.
//   * Weights are the MFMA "A" (row) operand and pixels the "B" (column) operand: the 16x16
//     accumulator then holds 4 CONSECUTIVE CHANNELS of one pixel per lane -> 8/16-byte vector
//     stores into the NHWC output and float4 loads of the per-channel epilogue vectors.
//   * Stage = 128 bytes of K per row (64 bf16 / 32 f32).  LDS tiles are [rows][8 x 16B slots]
//     with slot ^= h(row) (XOR swizzle found by exhaustive search: conflict-free for the
//     ds_read_b128 lane groups of gfx950 in both the bf16 and the f32 fragment pattern).
//   * Global -> register -> LDS double buffering: the loads of stage s+1 are issued before the
//     MFMAs of stage s and written to the other LDS buffer after them; one barrier per stage.
//     The register hop is where the fused prologue (BN scale/shift + ReLU of the producer
//     layer, padding kept at exactly 0) is applied, so un-normalised conv outputs never make
//     an extra HBM round trip.
//   * f32 path: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) with a permuted-K fragment so
//     the same 16-byte LDS reads serve it; it is the 1e-3 parity path, bf16 is the fast path.
//   * blockIdx is remapped so that each XCD (private L2) owns a contiguous range of tiles:
//     the channel tiles that re-read one pixel tile stay on one L2.
#include <cstdlib>
#include "common.h"
#include "profile.h"

int tf_conv_dma_launch(const tf_conv_args* a, int tile, int depth, hipStream_t stream);   // conv_dma.hip
bool tf_conv3x3h_applicable(const tf_conv_args* a, bool forced);                              // conv3x3h.hip
int tf_conv3x3h_mtiles(const tf_conv_args* a);
int tf_conv3x3h_launch(const tf_conv_args* a, hipStream_t stream);
bool tf_conv_pwx_applicable(const tf_conv_args* a);                                            // conv_pwx.hip
int tf_conv_pwx_mtiles(const tf_conv_args* a);
int tf_conv_pwx_launch(const tf_conv_args* a, const tf_bn_bwd_desc* pro, const void* pro_x2, void* pro_out, int pro_rows, float pro_count, hipStream_t stream);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvK {
  const char* x; const char* w; char* y;
  const float* pro_scale; const float* pro_shift;
  const float* epi_scale; const float* epi_shift;
  const char* aux; const char* aux2; const char* aux3;
  const float* mask_scale; const float* mask_shift;
  float* stat_out;
  int H, W, Cin, OH, OW, KW, stride, pad, sshift;
  int M, OHW, ldy, Ktot, cpt, nstages, ntiles, mode, epi, pro_relu;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) << 2) ^ (((row >> 3) & 1) * 6); }
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ swz(row)) << 4); }

template <typename T> struct Mma;
template <> struct Mma<tf::bf16_t> {
  static constexpr int KCH = 64;
  // one stage: 2 k-steps of 16x16x32
  template <int NF, int MF>
  __device__ static __forceinline__ void stage(const char* xs, const char* ws, int xrow0, int wrow0, f32x4 (&acc)[NF][MF]) {
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[MF], wf[NF];
#pragma unroll
      for (int m = 0; m < MF; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(xs + lds_off(xrow0 + m * 16 + r, ks * 4 + g));
#pragma unroll
      for (int n = 0; n < NF; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(ws + lds_off(wrow0 + n * 16 + r, ks * 4 + g));
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[n][m], 0, 0, 0);
    }
  }
};
template <> struct Mma<float> {
  static constexpr int KCH = 32;
  // one stage: 8 steps of 16x16x4; step j consumes k = g*8 + j in every lane group g (same permutation for both operands)
  template <int NF, int MF>
  __device__ static __forceinline__ void stage(const char* xs, const char* ws, int xrow0, int wrow0, f32x4 (&acc)[NF][MF]) {
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    f32x4 xf[MF][2], wf[NF][2];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h) xf[m][h] = *reinterpret_cast<const f32x4*>(xs + lds_off(xrow0 + m * 16 + r, 2 * g + h));
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int h = 0; h < 2; ++h) wf[n][h] = *reinterpret_cast<const f32x4*>(ws + lds_off(wrow0 + n * 16 + r, 2 * g + h));
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < MF; ++m)
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][j >> 2][j & 3], xf[m][j >> 2][j & 3], acc[n][m], 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ void load4(const char* base, size_t idx, float* f);
template <> __device__ __forceinline__ void load4<float>(const char* base, size_t idx, float* f) {
  const float4 v = *reinterpret_cast<const float4*>(base + idx * 4);
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ __forceinline__ void load4<tf::bf16_t>(const char* base, size_t idx, float* f) {
  const uint2 v = *reinterpret_cast<const uint2*>(base + idx * 2);
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(char* base, size_t idx, const float* f);
template <> __device__ __forceinline__ void store4<float>(char* base, size_t idx, const float* f) {
  *reinterpret_cast<float4*>(base + idx * 4) = make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ __forceinline__ void store4<tf::bf16_t>(char* base, size_t idx, const float* f) {
  *reinterpret_cast<uint2*>(base + idx * 2) = make_uint2(tf::pack_bf16x2(f[0], f[1]), tf::pack_bf16x2(f[2], f[3]));
}

template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvK a) {
  constexpr int KCH = Mma<T>::KCH;
  constexpr int EPS = tf::Elem<T>::kPer16B;         // elements per 16-byte slot
  constexpr int XR = BM / 32, WR = BN / 32;          // rows of each tile loaded per thread
  constexpr int WM = BM / 2, WN = BN / 2, MF = WM / 16, NF = WN / 16;
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, BUF = XBYTES + WBYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* pro_sc = reinterpret_cast<float*>(smem + 2 * BUF);
  float* pro_sh = pro_sc + a.Cin;

  // XCD-aware tile order (bijective for any grid size)
  int logical;
  {
    const int nb = gridDim.x, b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, slot = tid & 7, lrow = tid >> 3;

  if (a.pro_scale) {
    for (int c = tid; c < a.Cin; c += 256) { pro_sc[c] = a.pro_scale[c]; pro_sh[c] = a.pro_shift[c]; }
  }

  // per-row gather bases (pixel -> n, oh, ow), fixed for the whole K loop
  int rb_n[XR], rb_h[XR], rb_w[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int p = m0 + lrow + i * 32;
    if (p < a.M) {
      const int n = p / a.OHW, rem = p - n * a.OHW, oh = rem / a.OW, ow = rem - oh * a.OW;
      rb_n[i] = n * a.H * a.W;
      if (a.mode == 0) { rb_h[i] = oh * a.stride - a.pad; rb_w[i] = ow * a.stride - a.pad; }
      else             { rb_h[i] = oh + a.pad;            rb_w[i] = ow + a.pad; }
    } else { rb_n[i] = 0; rb_h[i] = -(1 << 28); rb_w[i] = -(1 << 28); }
  }

  uint4 xr[XR], wr[WR];
  unsigned okmask = 0;
  auto issue = [&](int st) {
    const int tap = st / a.cpt, cin0 = (st - tap * a.cpt) * KCH;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      int ih, iw; bool ok;
      if (a.mode == 0) {
        ih = rb_h[i] + kh; iw = rb_w[i] + kw;
        ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
      } else {
        const int th = rb_h[i] - kh, tw = rb_w[i] - kw, smask = (1 << a.sshift) - 1;
        ih = th >> a.sshift; iw = tw >> a.sshift;
        ok = th >= 0 && tw >= 0 && !(th & smask) && !(tw & smask) && ih < a.H && iw < a.W;
      }
      xr[i] = make_uint4(0, 0, 0, 0);
      if (ok) {
        const size_t e = ((size_t)(rb_n[i] + ih * a.W + iw)) * a.Cin + cin0 + slot * EPS;
        xr[i] = *reinterpret_cast<const uint4*>(a.x + e * sizeof(T));
        okmask |= 1u << i;
      }
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const size_t e = (size_t)(n0 + lrow + i * 32) * a.Ktot + (size_t)st * KCH + slot * EPS;
      wr[i] = *reinterpret_cast<const uint4*>(a.w + e * sizeof(T));
    }
  };
  auto commit = [&](int st, int buf) {
    char* xs = smem + buf * BUF;
    char* ws = xs + XBYTES;
    const int cin0 = (st % a.cpt) * KCH + slot * EPS;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      uint4 v = xr[i];
      if (a.pro_scale && ((okmask >> i) & 1u)) {       // BN + ReLU of the producer, never on padding
        float f[EPS];
        tf::unpack16<T>(v, f);
#pragma unroll
        for (int j = 0; j < EPS; ++j) {
          float t = f[j] * pro_sc[cin0 + j] + pro_sh[cin0 + j];
          f[j] = a.pro_relu ? fmaxf(t, 0.f) : t;
        }
        v = tf::pack16<T>(f);
      }
      *reinterpret_cast<uint4*>(xs + lds_off(lrow + i * 32, slot)) = v;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) *reinterpret_cast<uint4*>(ws + lds_off(lrow + i * 32, slot)) = wr[i];
  };

  f32x4 acc[NF][MF];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < MF; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  issue(0);
  __syncthreads();                    // pro_sc / pro_sh visible
  commit(0, 0);
  __syncthreads();
  for (int st = 0; st < a.nstages; ++st) {
    const int buf = st & 1;
    if (st + 1 < a.nstages) issue(st + 1);
    Mma<T>::template stage<NF, MF>(smem + buf * BUF, smem + buf * BUF + XBYTES, wm * WM, wn * WN, acc);
    if (st + 1 < a.nstages) commit(st + 1, buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int l = tid & 63, pr = l & 15, g = l >> 4;
  float* red = reinterpret_cast<float*>(smem);        // [2][BN] cross-wave stat scratch (tiles are dead now)
  const bool want_stats = a.epi & (TF_EPI_STATS | TF_EPI_STATS2);
  float s1[NF][4], s2[NF][4];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[n][r] = 0.f; s2[n][r] = 0.f; }

#pragma unroll
  for (int n = 0; n < NF; ++n) {
    const int c = n0 + wn * WN + n * 16 + g * 4;
    const bool cok = c < a.ldy;
    float es[4] = {1.f, 1.f, 1.f, 1.f}, eh[4] = {0.f, 0.f, 0.f, 0.f}, ms[4], mh[4];
    if (cok && (a.epi & TF_EPI_AFFINE)) {
      const float4 s = *reinterpret_cast<const float4*>(a.epi_scale + c), h = *reinterpret_cast<const float4*>(a.epi_shift + c);
      es[0] = s.x; es[1] = s.y; es[2] = s.z; es[3] = s.w; eh[0] = h.x; eh[1] = h.y; eh[2] = h.z; eh[3] = h.w;
    }
    if (cok && (a.epi & TF_EPI_MASK)) {
      const float4 s = *reinterpret_cast<const float4*>(a.mask_scale + c), h = *reinterpret_cast<const float4*>(a.mask_shift + c);
      ms[0] = s.x; ms[1] = s.y; ms[2] = s.z; ms[3] = s.w; mh[0] = h.x; mh[1] = h.y; mh[2] = h.z; mh[3] = h.w;
    }
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int p = m0 + wm * WM + m * 16 + pr;
      float v[4] = {acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]};
      if (a.epi & TF_EPI_STATS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[n][r] += v[r]; s2[n][r] += v[r] * v[r]; }   // rows >= M are exactly 0
      }
      if (p < a.M && cok) {
        const size_t o = (size_t)p * a.ldy + c;
        float ax[4];
        if (a.epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) load4<T>(a.aux, o, ax);
        if (a.epi & TF_EPI_AFFINE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * es[r] + eh[r];
        }
        if (a.epi & TF_EPI_RES) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += ax[r];
        }
        if (a.epi & TF_EPI_MASK) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (ax[r] * ms[r] + mh[r] > 0.f) ? v[r] : 0.f;
        }
        if (a.epi & TF_EPI_JOIN) {
          float y2[4], g3[4];
          load4<T>(a.aux2, o, y2); load4<T>(a.aux3, o, g3);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (y2[r] > 0.f) ? g3[r] : 0.f;
        }
        if (a.epi & TF_EPI_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (a.epi & TF_EPI_STATS2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[n][r] += v[r]; s2[n][r] += v[r] * ax[r]; }
        }
        store4<T>(a.y, o, v);
      }
    }
  }
  if (want_stats) {                                   // block-uniform
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1[n][r] += __shfl_xor(s1[n][r], o, 64); s2[n][r] += __shfl_xor(s2[n][r], o, 64); }
    if (wm == 1 && pr == 0) {
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl = wn * WN + n * 16 + g * 4 + r;
          red[cl] = s1[n][r]; red[BN + cl] = s2[n][r];
        }
    }
    __syncthreads();
    if (wm == 0 && pr == 0) {
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl = wn * WN + n * 16 + g * 4 + r, c = n0 + cl;
          if (c < a.ldy) {
            a.stat_out[((size_t)mt * 2 + 0) * a.ldy + c] = s1[n][r] + red[cl];
            a.stat_out[((size_t)mt * 2 + 1) * a.ldy + c] = s2[n][r] + red[BN + cl];
          }
        }
    }
  }
}

template <typename T, int BM, int BN>
int launch_conv(const tf_conv_args* A, hipStream_t stream) {
  constexpr int KCH = Mma<T>::KCH;
  ConvK k;
  k.x = (const char*)A->x; k.w = (const char*)A->w; k.y = (char*)A->y;
  k.pro_scale = A->pro_scale; k.pro_shift = A->pro_shift; k.epi_scale = A->epi_scale; k.epi_shift = A->epi_shift;
  k.aux = (const char*)A->aux; k.aux2 = (const char*)A->aux2; k.aux3 = (const char*)A->aux3;
  k.mask_scale = A->mask_scale; k.mask_shift = A->mask_shift; k.stat_out = A->stat_out;
  k.H = A->H; k.W = A->W; k.Cin = A->Cin; k.OH = A->OH; k.OW = A->OW; k.KW = A->KW; k.stride = A->stride; k.pad = A->pad;
  k.sshift = A->stride == 2 ? 1 : 0;
  k.M = A->N * A->OH * A->OW; k.OHW = A->OH * A->OW; k.ldy = A->ldy;
  k.cpt = A->Cin / KCH; k.Ktot = A->KH * A->KW * A->Cin; k.nstages = A->KH * A->KW * k.cpt;
  k.ntiles = (A->Cout + BN - 1) / BN; k.mode = A->mode; k.epi = A->epi; k.pro_relu = A->pro_relu;
  const int mtiles = (k.M + BM - 1) / BM;
  const size_t lds = 2 * (size_t)(BM + BN) * 128 + (A->pro_scale ? (size_t)A->Cin * 8 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  {
    // algorithmic work of this launch: 2*M*Cout*K flops; each operand / result touched once
    const double es = sizeof(T), M = k.M, Kt = k.Ktot;
    const double in_px = (double)A->N * A->H * A->W;
    double bytes = (in_px * A->Cin + (double)A->Cout * Kt + M * A->Cout) * es;
    if (A->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) bytes += M * A->Cout * es;
    if (A->epi & TF_EPI_JOIN) bytes += 2 * M * A->Cout * es;
    tf::ProfScope prof((sizeof(T) == 2 ? 3 : 0) + (BM == 64 ? 2 : (BN == 64 ? 1 : 0)), 2.0 * M * A->Cout * Kt, bytes, stream);
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN>), dim3(mtiles * k.ntiles), dim3(256), lds, stream, k);
  }
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

// tile codes: 1/2/3 = register-staged 128x128 / 128x64 / 64x64 (pixels x channels); 11/12/13 = LDS-DMA pipeline with a
// 3-deep ring (13: the ring depth then follows K, see tf_conv_dma_launch), 21/22/23 = 4-deep ring, 32 = ring-less 128x64;
// x4/x5/x6 = LDS-DMA pipeline on 32x32x16 fragments, 128x128 / 128x64 / 64x128, ring depth 3 (14..16), ring-less (34..36) or 2 (44..46).
// 50 = halo-resident 3x3 / stride 1 kernel (conv3x3h.hip).  0 = auto.
int pick_tile(const tf_conv_args* a) {
  if (a->tile) return a->tile;
  // 60 = conv_pwx (8 waves, 64 pixels x all output channels, register-staged pixel operand): only on request so far
  // (TINYFACES_PWX_FWD=1: the pointwise convs with 128 / 256 output channels and K >= 128, A/B knob)
  static const bool pwx_fwd = getenv("TINYFACES_PWX_FWD") != nullptr;
  if (pwx_fwd && tf_conv_pwx_applicable(a) && a->Cin >= 512) return 60;
  // measured on the bs=12 500x500 layer shapes (scripts/microbench.py): 128 pixels x 64 channels wins on every
  // layer (more, smaller tiles -> more blocks in flight per CU); 64x64 only when even that leaves CUs idle.
  const long M = (long)a->N * a->OH * a->OW;
  const long t2 = ((M + 127) / 128) * ((a->Cout + 63) / 64);
  if (!a->pro_scale) {
    // LDS-DMA pipeline.  bf16 convs of <= 4 K-stages (K <= 256 pointwise) are dispatch + epilogue bound: 128-pixel tiles without a
    // ring halve their block count at the same 4 blocks per CU (+0.6 % on the step, A/B on one box: 1036.6 -> 1043.3 img/s);
    // everything else 64x64 with the ring depth chosen by K.  The choice lives HERE so that tf_conv_mtiles agrees with the launch.
    static const bool t12_off = getenv("TINYFACES_T12_SHORTK_OFF") != nullptr;
    const int nst = a->KH * a->KW * (a->Cin / 64);
    // r3: ... except where the launch is big enough to be throughput-bound: from M = 16 384 pixels on, 64 pixels x 128 channels on
    // 32x32x16 fragments with a 2-deep ring moves the same bytes faster (1920x2560 pyramid level: 256 -> 1024 at M = 19 200 35.6 -> 30.7 us,
    // 128 -> 512 at M = 76 800 49.1 -> 40.8 us, 64 -> 256 at M = 307 200 103 -> 95 us; profiles/r03_microbench_eval.txt)
    static const bool t46_off = getenv("TINYFACES_T46_SHORTK_OFF") != nullptr;
    // (r4: a 128 x 128 / eight-wave pointwise kernel, conv_pw8, was built for the K >= 512 GEMMs and measured no faster on any layer shape:
    //  profiles/r04_conv_pw8_negative.txt -- removed again)
    if (!t46_off && a->dtype != TF_F32 && nst <= 4 && M >= 16384 && a->Cout % 128 == 0 && a->Cout >= 256) return 46;
    if (!t12_off && a->dtype != TF_F32 && nst <= 4) return 32;
    // 32x32x16 fragments (64 pixels x 128 channels per block, 32 x 64 per wave, 2-deep ring) win where a launch still has several
    // blocks per CU AND a long K loop: 3x3 convs / K >= 576 with M >= 16 384 pixels -- layer 2 at bs = 12 (26.3 vs 32.9 us forward,
    // 25.0 vs 30.6 us data gradient) and layers 2-3 of the evaluation pyramid (41.0 vs 50.6, 40.2 vs 44.8, 21.3 vs 23.8 us);
    // at M = 12 288 (layer 3, bs = 12) they lose: one block per CU, nothing hides a wave's LDS latency
    // (profiles/r02a_microbench_mma32_tiles.txt, profiles/r02a_microbench_eval_tiles.txt)
    // 3x3 / stride 1 with >= 128 output channels and enough tiles to fill the chip: the halo-resident kernel (conv3x3h.hip), which
    // moves each input byte into LDS once per 64-channel chunk instead of once per tap
    if (tf_conv3x3h_applicable(a, false)) return 50;
    static const bool mma32_off = getenv("TINYFACES_MMA32_OFF") != nullptr;
    if (!mma32_off && a->dtype != TF_F32 && nst >= 9 && M >= 16384 && a->Cout % 128 == 0) return 46;
    return 13;
  }
  return t2 >= 256 ? 2 : 3;                       // producer-BN prologue needs the register-staged kernel
}
int tile_bm(int t) { return ((t % 10) == 3 || (t % 10) == 6) ? 64 : 128; }

}  // namespace

extern "C" int tf_conv_mtiles(const tf_conv_args* a) {
  const long M = (long)a->N * a->OH * a->OW;
  const int t = pick_tile(a);
  if (t == 50) { const int mt = tf_conv3x3h_mtiles(a); return mt > tf_get_stat_rows() ? tf_get_stat_rows() : mt; }
  if (t == 60) { const int mt = tf_conv_pwx_mtiles(a); return mt > tf_get_stat_rows() ? tf_get_stat_rows() : mt; }
  const int bm = tile_bm(t);
  const int mt = (int)((M + bm - 1) / bm);
  return (t >= 10 && mt > tf_get_stat_rows()) ? tf_get_stat_rows() : mt;     // the DMA kernel folds its tiles into <= TF_STAT_ROWS rows
}

extern "C" int tf_conv2d(const tf_conv_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !a->x || !a->w || !a->y) return TF_ERR_ARG;
  const int kch = a->dtype == TF_F32 ? 32 : 64;
  if (a->dtype != TF_BF16 && a->dtype != TF_F32 && a->dtype != TF_F16) return TF_ERR_UNSUPPORTED;
  if (a->dtype == TF_F16 && (a->pro_scale || (a->tile && a->tile < 10))) return TF_ERR_UNSUPPORTED;   // fp16: the LDS-DMA kernel only
  if (a->Cin % kch != 0 || a->ldy % 4 != 0 || a->ldy < a->Cout) return TF_ERR_ARG;
  if (a->stride != 1 && a->stride != 2) return TF_ERR_UNSUPPORTED;
  if ((a->epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) && !a->stat_out) return TF_ERR_ARG;
  {
    const int ns = !!(a->epi & TF_EPI_STATS) + !!(a->epi & TF_EPI_STATS2) + !!(a->epi & TF_EPI_STATS3);
    if (ns > 1) return TF_ERR_ARG;
  }
  if ((a->epi & TF_EPI_MASK2) && !a->aux2) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_STATS3) && !a->aux3) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_JOIN) && (a->epi & (TF_EPI_MASK2 | TF_EPI_STATS3))) return TF_ERR_ARG;     // aux2 / aux3 have one meaning per launch
  if ((a->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) && !a->aux) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_JOIN) && (!a->aux2 || !a->aux3)) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_AFFINE) && (!a->epi_scale || !a->epi_shift)) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_MASK) && (!a->mask_scale || !a->mask_shift)) return TF_ERR_ARG;
  if (a->pro_scale && !a->pro_shift) return TF_ERR_ARG;
  const int t = pick_tile(a);
  if (a->bnf && !(t == 32 && a->mode == 0 && a->KH == 1 && a->KW == 1 && a->stride == 1 && a->Cin <= 256)) return TF_ERR_UNSUPPORTED;
  if (t == 50) return tf_conv3x3h_applicable(a, true) ? tf_conv3x3h_launch(a, stream) : TF_ERR_UNSUPPORTED;
  if (t == 60) return tf_conv_pwx_launch(a, nullptr, nullptr, nullptr, 0, 0.f, stream);
  if (t >= 10) {
    if (a->pro_scale) return TF_ERR_UNSUPPORTED;
    return tf_conv_dma_launch(a, t % 10, t >= 40 ? 2 : (t >= 30 ? 1 : (t >= 20 ? 4 : 3)), stream);
  }
  if (a->epi & (TF_EPI_MASK2 | TF_EPI_STATS3)) return TF_ERR_UNSUPPORTED;      // only the LDS-DMA kernel implements them
  if (a->dtype == TF_BF16) {
    if (t == 1) return launch_conv<tf::bf16_t, 128, 128>(a, stream);
    if (t == 2) return launch_conv<tf::bf16_t, 128, 64>(a, stream);
    return launch_conv<tf::bf16_t, 64, 64>(a, stream);
  }
  if (t == 1) return launch_conv<float, 128, 128>(a, stream);
  if (t == 2) return launch_conv<float, 128, 64>(a, stream);
  return launch_conv<float, 64, 64>(a, stream);
}

// tf_bn_bwd_apply_fused + tf_conv2d (pointwise, mode 0 or 1) in ONE launch: the conv's pixel operand is A*x + B*x2 + D with the
// BatchNorm-backward coefficients of `bn` (derived in-kernel from its statistic rows), the applied tensor also goes to `applied_out`
// (the weight gradient's operand).  conv_pwx.hip; TF_ERR_UNSUPPORTED for shapes it does not take (the caller runs the two kernels).
extern "C" int tf_conv2d_bnbwd(const tf_conv_args* a, const tf_bn_bwd_desc* bn, const void* x2, void* applied_out, int rows, float count, void* stream_) {
  if (!a || !a->x || !a->w || !a->y || !bn || !x2) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) && !a->stat_out) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) && !a->aux) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_MASK) && (!a->mask_scale || !a->mask_shift)) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_MASK2) && !a->aux2) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_STATS3) && !a->aux3) return TF_ERR_ARG;
  if (rows < 1 || rows > TF_STAT_ROWS) return TF_ERR_ARG;
  return tf_conv_pwx_launch(a, bn, x2, applied_out, rows, count, (hipStream_t)stream_);
}
