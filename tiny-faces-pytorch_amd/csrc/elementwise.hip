// HBM-bound companions of the conv engine (all NHWC "pixels x channels", 16-byte vector
// accesses, fp32 math): weight packing, stem im2col, max-pool, BN statistics / finalize /
// backward, residual join, bilinear head upsample + crop + add and their gradients.
// Each replaces the torch op cited in tinyfaces_hip.h (tinyfaces/models/model.py:90-126).
#include <algorithm>

#include <cstdlib>
#include "common.h"
#include "tuning.h"

namespace {

using tf::bf16_t;

// ---------------------------------------------------------------- weight packing
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int transpose,
                                   T* __restrict__ out, int rows_pad, int cols_pad) {
  // normal:    out[co][tap][ci]  rows = co (pad rows_pad), inner = ci (pad cols_pad)
  // transpose: out[ci][tap][co]  rows = ci,                 inner = co
  const int taps = KH * KW;
  const size_t total = (size_t)rows_pad * taps * cols_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int inner = (int)(i % cols_pad);
    const int tap = (int)((i / cols_pad) % taps);
    const int row = (int)(i / ((size_t)cols_pad * taps));
    const int co = transpose ? inner : row, ci = transpose ? row : inner;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap];
    tf::Elem<T>::store(out + i, v);
  }
}

struct PackJobs { tf_pack_job j[64]; };
template <typename T>
__global__ void __launch_bounds__(256) pack_batched_kernel(const PackJobs jobs) {
  const tf_pack_job& J = jobs.j[blockIdx.y];
  const int taps = J.taps;
  const size_t total = (size_t)J.rows_pad * taps * J.cols_pad;
  T* out = reinterpret_cast<T*>(J.dst);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int inner = (int)(i % J.cols_pad);
    const int tap = (int)((i / J.cols_pad) % taps);
    const int row = (int)(i / ((size_t)J.cols_pad * taps));
    const int co = J.transpose ? inner : row, ci = J.transpose ? row : inner;
    float v = 0.f;
    if (co < J.cout && ci < J.cin) v = J.src[((size_t)co * J.cin + ci) * taps + tap];
    tf::Elem<T>::store(out + i, v);
  }
}

// Both operand layouts of one weight in one pass: the OIHW fp32 master is read once (coalesced runs of ci*taps floats per
// output channel) into an LDS tile, then written as [co][tap][ci] (forward operand) and/or [ci][tap][co] (data-gradient
// operand), each in full segments along its own fastest axis.  The per-element kernel above reads the source with a
// taps*4-byte (normal) or Cin*taps*4-byte (transposed) lane stride: 1.5 TB/s of traffic for 330 us per training step.
struct Pack2Jobs { tf_pack2_job j[40]; int tile0[41]; };
template <typename T>
__global__ void __launch_bounds__(256) pack2_kernel(const Pack2Jobs jobs, int njobs) {
  extern __shared__ float tile[];
  int ji = 0;
  while (ji + 1 < njobs && (int)blockIdx.x >= jobs.tile0[ji + 1]) ++ji;
  const tf_pack2_job& J = jobs.j[ji];
  const int taps = J.taps;
  const int TR = taps == 1 ? 64 : 32, TC = TR;                 // co x ci tile
  const int pitch = TC * taps + 1;
  const int ext_co = max(J.dst ? J.rows_pad : 0, J.dst_t ? J.cols_pad_t : 0);
  const int ext_ci = max(J.dst ? J.cols_pad : 0, J.dst_t ? J.rows_pad_t : 0);
  const int tiles_ci = (ext_ci + TC - 1) / TC;
  const int t = (int)blockIdx.x - jobs.tile0[ji];
  const int co0 = (t / tiles_ci) * TR, ci0 = (t % tiles_ci) * TC;
  (void)ext_co;
  // ---- load: rows of (TC*taps) consecutive floats
  const int run = TC * taps;
  if (ci0 + TC <= J.cin && ((J.cin * taps) & 3) == 0) {
    // r6: interior tiles (every trunk weight: channel counts are multiples of the tile) read 16 bytes per lane, four loads in flight per thread --
    // the scalar loop below moved the 111 MB of fp32 masters at ~1 TB/s (190 us of second-queue time per step)
    const int run4 = run >> 2;
#pragma unroll 4
    for (int e = threadIdx.x; e < TR * run4; e += 256) {
      const int r = e / run4, k = (e - r * run4) << 2;
      const int co = co0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (co < J.cout) v = *reinterpret_cast<const float4*>(J.src + ((size_t)co * J.cin + ci0) * taps + k);
      float* d = tile + r * pitch + k;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
  for (int e = threadIdx.x; e < TR * run; e += 256) {
    const int r = e / run, k = e - r * run;
    const int co = co0 + r, ci = ci0 + k / taps;
    float v = 0.f;
    if (co < J.cout && ci < J.cin) v = J.src[((size_t)co * J.cin + ci0) * taps + k];
    tile[r * pitch + k] = v;
  }
  }
  __syncthreads();
  // ---- forward operand [co][tap][ci]: ci fastest, 4 consecutive channels per lane (8-byte bf16 / 16-byte fp32 stores)
  auto store4 = [](T* p, float a, float b, float c_, float d) {
    if constexpr (sizeof(T) == 2) {
      uint2 q;
      q.x = tf::pack2<T>(a, b);
      q.y = tf::pack2<T>(c_, d);
      *reinterpret_cast<uint2*>(p) = q;
    } else {
      *reinterpret_cast<float4*>(p) = make_float4(a, b, c_, d);
    }
  };
  const int Q = TC / 4;
  if (J.dst) {
    T* out = reinterpret_cast<T*>(J.dst);
    for (int e = threadIdx.x; e < TR * taps * Q; e += 256) {
      const int q = e % Q, tap = (e / Q) % taps, r = e / (Q * taps);
      const int co = co0 + r, ci = ci0 + 4 * q;
      if (co < J.rows_pad && ci < J.cols_pad) {
        const float* t0 = tile + r * pitch + (4 * q) * taps + tap;
        store4(out + ((size_t)co * taps + tap) * J.cols_pad + ci, t0[0], t0[taps], t0[2 * taps], t0[3 * taps]);
      }
    }
  }
  // ---- data-gradient operand [ci][tap][co]: co fastest (LDS column walk, odd pitch: conflict-free)
  if (J.dst_t) {
    T* out = reinterpret_cast<T*>(J.dst_t);
    const int QR = TR / 4;
    for (int e = threadIdx.x; e < TC * taps * QR; e += 256) {
      const int q = e % QR, tap = (e / QR) % taps, cl = e / (QR * taps);
      const int co = co0 + 4 * q, ci = ci0 + cl;
      if (ci < J.rows_pad_t && co < J.cols_pad_t) {
        const float* t0 = tile + (4 * q) * pitch + cl * taps + tap;
        store4(out + ((size_t)ci * taps + tap) * J.cols_pad_t + co, t0[0], t0[pitch], t0[2 * pitch], t0[3 * pitch]);
      }
    }
  }
}

// ---------------------------------------------------------------- stem im2col
// x NCHW fp32 [N][3][H][W] -> col [M][ldc], k = c*49 + kh*7 + kw (== OIHW order of conv1.weight), zero padded
// A block owns a 16x16 tile of output pixels: its 37x37x3 input patch is loaded once into LDS (coalesced rows), then
// the 256 x (ldc/EPS) 16-byte chunks of the tile are assembled from LDS and stored with consecutive lanes on consecutive
// chunks of a pixel row (pixel rows of one tile line are contiguous in col).  The per-element global gather this
// replaces was 2.5x slower (profiles/r01d: 177 us for the 288 MB matrix of a bs=12 500x500 batch).
constexpr int kI2cT = 16, kI2cP = 2 * kI2cT + 5;
template <typename T>
__global__ void __launch_bounds__(256) stem_im2col_kernel(const float* __restrict__ x, int N, int H, int W, int OH, int OW,
                                                          T* __restrict__ col, int ldc) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  __shared__ float patch[3][kI2cP][kI2cP + 1];
  const int tw = (OW + kI2cT - 1) / kI2cT, th = (OH + kI2cT - 1) / kI2cT;
  const int n = blockIdx.x / (tw * th), tr = (blockIdx.x / tw) % th, tc = blockIdx.x % tw;
  const int oh0 = tr * kI2cT, ow0 = tc * kI2cT;
  const int ih0 = oh0 * 2 - 3, iw0 = ow0 * 2 - 3;
  for (int e = threadIdx.x; e < 3 * kI2cP * kI2cP; e += 256) {
    const int c = e / (kI2cP * kI2cP), r = e - c * kI2cP * kI2cP, py = r / kI2cP, px = r - py * kI2cP;
    const int ih = ih0 + py, iw = iw0 + px;
    float v = 0.f;
    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[(((size_t)n * 3 + c) * H + ih) * W + iw];
    patch[c][py][px] = v;
  }
  __syncthreads();
  const int spr = ldc / EPS;
  for (int e = threadIdx.x; e < kI2cT * kI2cT * spr; e += 256) {
    const int s = e % spr, pl = e / spr, oy = pl / kI2cT, ox = pl - oy * kI2cT;
    const int oh = oh0 + oy, ow = ow0 + ox;
    if (oh >= OH || ow >= OW) continue;
    float f[EPS];
#pragma unroll
    for (int j = 0; j < EPS; ++j) {
      const int k = s * EPS + j;
      float v = 0.f;
      if (k < 147) {
        const int c = k / 49, r = k - c * 49, kh = r / 7, kw = r - kh * 7;
        v = patch[c][oy * 2 + kh][ox * 2 + kw];
      }
      f[j] = v;
    }
    const size_t p = ((size_t)n * OH + oh) * OW + ow;
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(col) + (p * spr + s) * 16) = tf::pack16<T>(f);
  }
}

// ---------------------------------------------------------------- max-pool 3x3 s2 p1 (+ fused BN/ReLU of the stem)
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int OH, int OW) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  const int spr = C / EPS;
  const size_t total = (size_t)N * OH * OW * spr;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int s = (int)(i % spr);
    const size_t p = i / spr;
    const int ow = (int)(p % OW), oh = (int)((p / OW) % OH), n = (int)(p / ((size_t)OW * OH));
    float best[EPS]; int bi[EPS];
#pragma unroll
    for (int j = 0; j < EPS; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    float fs[EPS], fh[EPS];
    if (sc) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { fs[j] = sc[s * EPS + j]; fh[j] = sh[s * EPS + j]; }
    }
    // r6: the nine taps are requested together (clamped addresses, the border taps skipped afterwards): inside the bounds test each load was a
    // dependent round trip of its own
    uint4 tq[9];
    bool tok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
        const bool v = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        tok[kh * 3 + kw] = v;
        const size_t o = v ? (((size_t)n * H + ih) * W + iw) * spr + s : 0;
        tq[kh * 3 + kw] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + o * 16);
      }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (!tok[k]) continue;
      float f[EPS];
      tf::unpack16<T>(tq[k], f);
#pragma unroll
      for (int j = 0; j < EPS; ++j) {
        float t = f[j];
        if (sc) t = fmaxf(t * fs[j] + fh[j], 0.f);
        if (t > best[j]) { best[j] = t; bi[j] = k; }     // first max wins (torch CPU max_pool2d)
      }
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + i * 16) = tf::pack16<T>(best);
    if (idx) {                                     // one EPS-byte store per chunk
      uint8_t ib[EPS];
#pragma unroll
      for (int j = 0; j < EPS; ++j) ib[j] = (uint8_t)bi[j];
      if constexpr (EPS == 8) { uint2 q; __builtin_memcpy(&q, ib, 8); *reinterpret_cast<uint2*>(idx + i * EPS) = q; }
      else { uint32_t q; __builtin_memcpy(&q, ib, 4); *reinterpret_cast<uint32_t*>(idx + i * EPS) = q; }
    }
  }
}

// gz[n,ih,iw,c] = (relu mask of BN(x)) * sum over windows whose arg-max is this position of g[window]
template <typename T, bool STATS>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* __restrict__ g, const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                                          const float* __restrict__ sc, const float* __restrict__ sh, int N, int H, int W,
                                                          int C, int OH, int OW, T* __restrict__ gz, float* __restrict__ stat_out, int srows) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  const int spr = C / EPS;
  // STATS (r4): the column sums  sum gz, sum gz * x  of the stem's BatchNorm backward (what tf_colstats(gz, NULL, x) computes in a second
  // pass over both tensors) are taken here, where both values are in registers: a thread keeps its channel chunk over the grid stride
  // (gridDim.x * 256 is a multiple of C / EPS), lanes of a wave with the same chunk are summed by shuffles, the four waves through LDS,
  // and the block folds its row into stat_out[blockIdx % srows][2][C] like the colstats kernel does
  float s1[EPS], s2[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const size_t total = (size_t)N * H * W * spr;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int s = (int)(i % spr);
    const size_t p = i / spr;
    const int iw = (int)(p % W), ih = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
    float acc[EPS];
#pragma unroll
    for (int j = 0; j < EPS; ++j) acc[j] = 0.f;
    // windows (oh, ow) with ih = 2*oh - 1 + kh  ->  oh = (ih + 1 - kh) / 2: an odd row belongs to the windows kh = 0 and kh = 2, an even row to
    // kh = 1 only (likewise the columns): 1, 2, 2 or 4 windows per pixel.  r6: the up to four (gradient, arg-max) pairs and the BN input are
    // REQUESTED TOGETHER (clamped addresses, selected afterwards) -- the first form loaded them inside the window loops, one dependent round trip
    // per window (2.1 TB/s for the 250 MB of the stem's max-pool backward).  Same sums in the same (kh, kw) order.
    const int kh0 = (ih + 1) & 1, kw0 = (iw + 1) & 1;             // first window tap of this row / column; the second one (if any) is + 2
    uint4 gq[4];
    uint2 iq[4];
    bool wv[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) {
        const int kh = kh0 + 2 * a, kw = kw0 + 2 * bq;
        const int oh = (ih + 1 - kh) >> 1, ow = (iw + 1 - kw) >> 1;
        const bool v = kh < 3 && kw < 3 && ih + 1 - kh >= 0 && iw + 1 - kw >= 0 && oh < OH && ow < OW;
        wv[a * 2 + bq] = v;
        const size_t o = v ? (((size_t)n * OH + oh) * OW + ow) * spr + s : 0;
        gq[a * 2 + bq] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(g) + o * 16);
        if constexpr (EPS == 8) iq[a * 2 + bq] = *reinterpret_cast<const uint2*>(idx + o * EPS);
        else { iq[a * 2 + bq].x = *reinterpret_cast<const uint32_t*>(idx + o * EPS); iq[a * 2 + bq].y = 0u; }
      }
    const uint4 xv = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + i * 16);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) {
        if (!wv[a * 2 + bq]) continue;
        const int code = (kh0 + 2 * a) * 3 + kw0 + 2 * bq;
        float gf[EPS];
        tf::unpack16<T>(gq[a * 2 + bq], gf);
        uint8_t ib[8];
        __builtin_memcpy(ib, &iq[a * 2 + bq], 8);
#pragma unroll
        for (int j = 0; j < EPS; ++j) if (ib[j] == code) acc[j] += gf[j];
      }
    float xf[EPS];
    tf::unpack16<T>(xv, xf);
#pragma unroll
    for (int j = 0; j < EPS; ++j) if (!(xf[j] * sc[s * EPS + j] + sh[s * EPS + j] > 0.f)) acc[j] = 0.f;
    const uint4 outv = tf::pack16<T>(acc);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(gz) + i * 16) = outv;
    if constexpr (STATS) {
      float r[EPS];
      tf::unpack16<T>(outv, r);               // the ROUNDED gradient: what the separate pass would read back
#pragma unroll
      for (int j = 0; j < EPS; ++j) { s1[j] += r[j]; s2[j] += r[j] * xf[j]; }
    }
  }
  if constexpr (STATS) {
    __shared__ float red[4][2][256];           // C <= 256
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = threadIdx.x % spr;
#pragma unroll
    for (int j = 0; j < EPS; ++j) {
      for (int o = spr; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    }
    if (lane < spr) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { red[wave][0][s * EPS + j] = s1[j]; red[wave][1][s * EPS + j] = s2[j]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
      const int k = e / C, c = e - k * C;
      const float t = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
      if ((int)gridDim.x <= srows) stat_out[(size_t)blockIdx.x * 2 * C + e] = t;
      else atomicAdd(&stat_out[(size_t)(blockIdx.x % srows) * 2 * C + e], t);          // rows are zero on entry (the finalize clears them)
    }
  }
}

// ---------------------------------------------------------------- column statistics  (M x C matrix)
// out[blk][k][c]:  k=0: sum g'   k=1: sum g'*a   k=2: sum g'*b      with g' = g * (y > 0) if y given
// (a == nullptr -> only k=0;  "sum x, sum x^2" is obtained with g = a = x)
// Specialised on (mask present, number of sums) and unrolled over two row groups so that all 2..8 16-byte loads of an
// iteration are in flight together (the runtime-flag form issued them one dependent group at a time: 2x slower).
template <typename T, bool HASY, int NK>
__global__ void __launch_bounds__(256) colstats_kernel(const T* __restrict__ g, const T* __restrict__ y, const T* __restrict__ a,
                                                       const T* __restrict__ b, int M, int C, int ld, int rows_per_block,
                                                       float* __restrict__ out, int srows) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  const int ct = C / EPS;                         // threads across channels (<= 256)
  const int rt = 256 / ct;                        // rows in flight
  const int tc = threadIdx.x % ct, tr = threadIdx.x / ct;
  extern __shared__ float red[];                  // [rt][NK][C]
  float s0[EPS], s1[EPS], s2[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  auto ld16 = [&](const T* p, size_t o) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p) + o); };
  auto accum = [&](const uint4& gq, const uint4& yq, const uint4& aq, const uint4& bq) {
    float gf[EPS], f[EPS];
    tf::unpack16<T>(gq, gf);
    if (HASY) {
      tf::unpack16<T>(yq, f);
#pragma unroll
      for (int j = 0; j < EPS; ++j) if (!(f[j] > 0.f)) gf[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPS; ++j) s0[j] += gf[j];
    if (NK > 1) {
      tf::unpack16<T>(aq, f);
#pragma unroll
      for (int j = 0; j < EPS; ++j) s1[j] += gf[j] * f[j];
    }
    if (NK > 2) {
      tf::unpack16<T>(bq, f);
#pragma unroll
      for (int j = 0; j < EPS; ++j) s2[j] += gf[j] * f[j];
    }
  };
  if (tr < rt) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    int r = r0 + tr;
    for (; r + rt < r1; r += 2 * rt) {
      const size_t o0 = ((size_t)r * ld + tc * EPS) * sizeof(T), o1 = ((size_t)(r + rt) * ld + tc * EPS) * sizeof(T);
      const uint4 g0 = ld16(g, o0), g1 = ld16(g, o1);
      const uint4 y0 = HASY ? ld16(y, o0) : z, y1 = HASY ? ld16(y, o1) : z;
      const uint4 a0 = NK > 1 ? ld16(a, o0) : z, a1 = NK > 1 ? ld16(a, o1) : z;
      const uint4 b0 = NK > 2 ? ld16(b, o0) : z, b1 = NK > 2 ? ld16(b, o1) : z;
      accum(g0, y0, a0, b0);
      accum(g1, y1, a1, b1);
    }
    if (r < r1) {
      const size_t o0 = ((size_t)r * ld + tc * EPS) * sizeof(T);
      accum(ld16(g, o0), HASY ? ld16(y, o0) : z, NK > 1 ? ld16(a, o0) : z, NK > 2 ? ld16(b, o0) : z);
    }
#pragma unroll
    for (int j = 0; j < EPS; ++j) {
      red[(tr * NK + 0) * C + tc * EPS + j] = s0[j];
      if (NK > 1) red[(tr * NK + 1) * C + tc * EPS + j] = s1[j];
      if (NK > 2) red[(tr * NK + 2) * C + tc * EPS + j] = s2[j];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NK * C; e += 256) {
    float t = 0.f;
    for (int r = 0; r < rt; ++r) t += red[r * NK * C + e];
    if ((int)gridDim.x <= srows) out[(size_t)blockIdx.x * NK * C + e] = t;
    else atomicAdd(&out[(size_t)(blockIdx.x % srows) * NK * C + e], t);      // rows are zero on entry (finalize clears)
  }
}

// ---------------------------------------------------------------- BN finalize (forward, batch statistics)
// Partials come as [nblk][nk][ld] rows (conv epilogue tiles or colstats blocks).  A block owns 32 channels and
// sums the nblk rows with 32 row-lanes per channel (coalesced 128-byte reads), then 32 threads finalize.
constexpr int kFinCh = 32, kFinLanes = 32;
template <int NK>
__device__ __forceinline__ void reduce_partial_rows(float* __restrict__ partial, int nblk, int nk, const int* kidx, int ld, int C,
                                                    double (*out)[kFinCh], int clear) {
  __shared__ double red[NK][kFinLanes][kFinCh];
  const int tc = threadIdx.x % kFinCh, tr = threadIdx.x / kFinCh;
  const int c = blockIdx.x * kFinCh + tc;
  double acc[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) acc[k] = 0.0;
  if (c < C) {
    for (int b = tr; b < nblk; b += kFinLanes) {
#pragma unroll
      for (int k = 0; k < NK; ++k) acc[k] += (double)partial[((size_t)b * nk + kidx[k]) * ld + c];
    }
  }
  if (clear && c < C) {                 // leave the buffer zeroed for the next producer (own channels only: race-free)
    for (int b = tr; b < nblk; b += kFinLanes)
      for (int k = 0; k < nk; ++k) partial[((size_t)b * nk + k) * ld + c] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) red[k][tr][tc] = acc[k];
  __syncthreads();
  if (tr < NK) {                      // lane tr finalizes statistic tr for channel tc
    double t = 0.0;
    for (int r = 0; r < kFinLanes; ++r) t += red[tr][r][tc];
    out[tr][tc] = t;
  }
  __syncthreads();
}

// partial[blk][2][ld] (sum, sumsq) -> scale/shift for y = x*scale+shift, saved mean/invstd, running stats
__global__ void __launch_bounds__(kFinCh* kFinLanes) bn_finalize_kernel(float* __restrict__ partial, int nblk, int ld, int C, float count, int clear,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                        float momentum, float* __restrict__ scale, float* __restrict__ shift,
                                                                        float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                                        float* __restrict__ running_mean, float* __restrict__ running_var) {
  __shared__ double sums[2][kFinCh];
  const int kidx[2] = {0, 1};
  reduce_partial_rows<2>(partial, nblk, 2, kidx, ld, C, sums, clear);
  const int c = blockIdx.x * kFinCh + threadIdx.x;
  if (threadIdx.x >= kFinCh || c >= C) return;
  const double s = sums[0][threadIdx.x], q = sums[1][threadIdx.x];
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc; shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean; invstd_out[c] = invstd;
  if (running_mean) {
    const double unbiased = count > 1.f ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// eval-mode BN fold: scale = gamma/sqrt(var+eps), shift = beta - mean*scale
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                               const float* __restrict__ rv, float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
}

// BN backward finalize: partial[blk][nk][ld] with k0 = sum gz, kidx = sum gz*x  ->
//   dgamma, dbeta and the affine form  g_x = A*gz + B*x + D
__global__ void __launch_bounds__(kFinCh* kFinLanes) bn_bwd_finalize_kernel(float* __restrict__ partial, int nblk, int nk, int kidx_, int ld, int C,
                                                                            int clear, float count, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                            const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                                            float* __restrict__ dbeta, float* __restrict__ cA, float* __restrict__ cB,
                                                                            float* __restrict__ cD) {
  __shared__ double sums[2][kFinCh];
  const int kidx[2] = {0, kidx_};
  reduce_partial_rows<2>(partial, nblk, nk, kidx, ld, C, sums, clear);
  const int c = blockIdx.x * kFinCh + threadIdx.x;
  if (threadIdx.x >= kFinCh || c >= C) return;
  const double s1 = sums[0][threadIdx.x], s2 = sums[1][threadIdx.x];
  const double mu = mean[c], is = invstd[c], ga = gamma[c];
  const double dg = (s2 - mu * s1) * is;         // sum gz * xhat
  dgamma[c] = (float)dg; dbeta[c] = (float)s1;
  const double A = ga * is;
  cA[c] = (float)A;
  cB[c] = (float)(-A * is * dg / count);
  cD[c] = (float)(-A * s1 / count + A * mu * is * dg / count);
}

// out = A*g' + B*x + D   (g' = g*(y>0) when y given).  BN input gradient, materialised for dgrad / wgrad.
// The grid stride (gridDim*256) is a multiple of the 16-byte slots per row, so a thread keeps ONE channel chunk:
// its per-channel coefficients are loaded once, the loop body is 3 loads + 1 store of 16 bytes.
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y, const T* __restrict__ x,
                                                           const float* __restrict__ cA, const float* __restrict__ cB,
                                                           const float* __restrict__ cD, size_t M, int C, T* __restrict__ out) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  const int spr = C / EPS;
  const size_t total = M * spr;
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int s = (int)(i0 % spr);
  float A[EPS], B[EPS], D[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { A[j] = cA[s * EPS + j]; B[j] = cB[s * EPS + j]; D[j] = cD[s * EPS + j]; }
  for (size_t i = i0; i < total; i += (size_t)gridDim.x * 256) {
    float gf[EPS], xf[EPS], yf[EPS];
    tf::unpack16<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(g) + i * 16), gf);
    tf::unpack16<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + i * 16), xf);
    if (y) {
      tf::unpack16<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(y) + i * 16), yf);
#pragma unroll
      for (int j = 0; j < EPS; ++j) if (!(yf[j] > 0.f)) gf[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPS; ++j) gf[j] = A[j] * gf[j] + B[j] * xf[j] + D[j];
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + i * 16) = tf::pack16<T>(gf);
  }
}

// EPS consecutive per-channel coefficients as 16-byte loads (a conditional per-element form does not vectorise: 3x slower kernel)
template <int EPS>
__device__ __forceinline__ void load_coef(const float* __restrict__ p, float (&out)[EPS]) {
#pragma unroll
  for (int j = 0; j < EPS; j += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + j);
    out[j] = v.x; out[j + 1] = v.y; out[j + 2] = v.z; out[j + 3] = v.w;
  }
}

// y = relu(x*s1+h1 + (r*s2+h2  |  r))      block output of a Bottleneck in training mode
template <typename T, bool DS>
__global__ void __launch_bounds__(256) bn_add_relu_kernel(const T* __restrict__ x, const float* __restrict__ s1, const float* __restrict__ h1,
                                                          const T* __restrict__ r, const float* __restrict__ s2, const float* __restrict__ h2,
                                                          size_t M, int C, T* __restrict__ y) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  const int spr = C / EPS;
  const size_t total = M * spr;
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int s = (int)(i0 % spr);
  float a1[EPS], b1[EPS], a2[EPS], b2[EPS];
  load_coef<EPS>(s1 + s * EPS, a1); load_coef<EPS>(h1 + s * EPS, b1);
  if (DS) {
    load_coef<EPS>(s2 + s * EPS, a2); load_coef<EPS>(h2 + s * EPS, b2);
  }
  for (size_t i = i0; i < total; i += (size_t)gridDim.x * 256) {
    float xf[EPS], rf[EPS];
    tf::unpack16<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + i * 16), xf);
    tf::unpack16<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(r) + i * 16), rf);
#pragma unroll
    for (int j = 0; j < EPS; ++j) xf[j] = fmaxf(xf[j] * a1[j] + b1[j] + (DS ? rf[j] * a2[j] + b2[j] : rf[j]), 0.f);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + i * 16) = tf::pack16<T>(xf);
  }
}

// y = relu(x*s+h): materialised BN+ReLU (input of the 3x3 conv, so that it can use the LDS-DMA pipeline)
template <typename T>
__global__ void __launch_bounds__(256) bn_relu_kernel(const T* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh, size_t M,
                                                      int C, T* __restrict__ y) {
  constexpr int EPS = tf::Elem<T>::kPer16B;
  const int spr = C / EPS;
  const size_t total = M * spr;
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int s = (int)(i0 % spr);
  float a[EPS], b[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { a[j] = sc[s * EPS + j]; b[j] = sh[s * EPS + j]; }
  for (size_t i = i0; i < total; i += (size_t)gridDim.x * 256) {
    float f[EPS];
    tf::unpack16<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + i * 16), f);
#pragma unroll
    for (int j = 0; j < EPS; ++j) f[j] = fmaxf(f[j] * a[j] + b[j], 0.f);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + i * 16) = tf::pack16<T>(f);
  }
}

// ---------------------------------------------------------------- head: bilinear ConvTranspose2d(k4,s2,p1) + crop + add
// four consecutive elements (ldc % 4 == 0, 4-element groups: 8- / 16-byte aligned) in one load
template <typename T> __device__ __forceinline__ void load4(const T* p, float* f) {
  if constexpr (sizeof(T) == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
  } else {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    float g[8];
    tf::unpack16<T>(make_uint4(q.x, q.y, 0u, 0u), g);
    f[0] = g[0]; f[1] = g[1]; f[2] = g[2]; f[3] = g[3];
  }
}

// out NCHW fp32 [B][C][H3][W3] = s3[(b,y,x)][c] + sum_{ky,kx} s4[(b,i,j)][c] * wup[c][ky][kx],  y = 2i-1+ky, x = 2j-1+kx
// (model.py:104-126; only the channel diagonal of score4_upsample.weight is non-zero, model.py:61-65)
constexpr int kUpPx = 64;                        // pixels per block (256-byte runs in every output plane; 128: slower, 2 blocks per CU)
constexpr int kUpT = 512;                        // threads of upsample_add_kernel: 8 waves share one 41 KB staging tile (3 blocks = 24 waves per CU)
template <typename T>
__global__ void __launch_bounds__(kUpT) upsample_add_kernel(const T* __restrict__ s3, const T* __restrict__ s4, const float* __restrict__ wup,
                                                           int B, int C, int ldc, int H3, int W3, int H4, int W4, float* __restrict__ out) {
  // block: 64 consecutive pixels of one image x all channels; LDS transpose for coalesced NCHW rows.
  // r6 rewrite (the kernel ran at 0.6-0.8 TB/s: 110 us for the 63 MB of the 1920 x 2560 pyramid level, the LAST kernel of every evaluation forward):
  //   * 16-byte operand loads, and ALL of a thread's loads (its s3 chunk + up to four s4 taps, for each of its four (pixel, chunk) items) are requested
  //     before the first one is used -- the old loop walked eight items one after the other, five dependent round trips each;
  //   * the 16 bilinear weights of a channel come from an LDS table ordered [pixel parity][tap][channel] (two float4 reads per tap) instead of
  //     16 scalar global loads per item;
  //   * the staging tile is [pixel][ldc + 1]: the chunk-major writes and the pixel-major reads of the transposition are both (nearly) conflict-free.
  // Same sums in the same order as before: bit-identical outputs.
  constexpr int EPS = tf::Elem<T>::kPer16B;
  extern __shared__ float up_smem[];
  float* const wtab = up_smem;                     // [py][px][a][b][ldc]
  const int pitch = ldc + 1;
  float* const tile = up_smem + 16 * ldc;          // [kUpPx][pitch]
  const int hw = H3 * W3;
  const int tiles_per_img = (hw + kUpPx - 1) / kUpPx;
  const int b = blockIdx.x / tiles_per_img, p0 = (blockIdx.x % tiles_per_img) * kUpPx;
  (void)B;
  const int chunks = ldc / EPS, total = kUpPx * chunks;
  constexpr int NI = (kUpPx * 16 + kUpT - 1) / kUpT;      // one batch covers a 2-byte tile (16 chunks per pixel); fp32 takes two
  uint4 q3[NI], q4[NI][4];
  bool ok[NI], tv[NI][4];
  int ipx[NI], ich[NI], ipar[NI];
  auto request = [&](int e0) {                     // every load of a batch of NI (pixel, chunk) items, none of them waited for
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int e = e0 + k * kUpT;
      const int px = e / chunks, ch = e - px * chunks, p = p0 + px;
      ok[k] = e < total && p < hw;
      const int pp = ok[k] ? p : 0;
      const int y = pp / W3, x = pp - y * W3;
      ipx[k] = px; ich[k] = ch; ipar[k] = ((y & 1) * 2 + (x & 1)) * 4;
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
      q3[k] = ok[k] ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(s3 + ((size_t)b * hw + pp) * ldc) + ch * 16) : z;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int ky = ((y + 1) & 1) + 2 * a, ty = y + 1 - ky, i = ty >> 1;
          const int kx = ((x + 1) & 1) + 2 * bb, tx = x + 1 - kx, jx = tx >> 1;
          const bool v = ok[k] && ty >= 0 && i < H4 && tx >= 0 && jx < W4;
          tv[k][a * 2 + bb] = v;
          q4[k][a * 2 + bb] = v ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(s4 + (((size_t)b * H4 + i) * W4 + jx) * ldc) + ch * 16) : z;
        }
    }
  };
  auto finish = [&]() {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (!ok[k]) continue;
      float v[EPS], t[EPS];
      tf::unpack16<T>(q3[k], v);
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        if (!tv[k][tp]) continue;
        tf::unpack16<T>(q4[k][tp], t);
        const float* wt = wtab + (ipar[k] + tp) * ldc + ich[k] * EPS;
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] += t[j] * wt[j];
      }
      float* dst = tile + ipx[k] * pitch + ich[k] * EPS;
#pragma unroll
      for (int j = 0; j < EPS; ++j) dst[j] = v[j];
    }
  };
  // the first batch is requested BEFORE the weight table is built: the table's own (gathering) loads and the operand loads travel together, and a
  // block pays one memory round trip in front of its arithmetic, not two
  request(threadIdx.x);
  for (int e = threadIdx.x; e < 16 * ldc; e += kUpT) {
    const int c = e % ldc, k = e / ldc;
    const int ky = ((((k >> 3) & 1) + 1) & 1) + 2 * ((k >> 1) & 1), kx = ((((k >> 2) & 1) + 1) & 1) + 2 * (k & 1);
    wtab[e] = c < C ? wup[c * 16 + ky * 4 + kx] : 0.f;
  }
  __syncthreads();                                 // the weight table
  finish();
  for (int e0 = threadIdx.x + NI * kUpT; e0 < total; e0 += NI * kUpT) { request(e0); finish(); }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int h0 = 0; h0 < kUpPx; h0 += 64) {
    if ((hw & 3) == 0) {
      // rows of 4 floats stay 16-byte aligned in every channel plane: a lane stores four pixels of one channel, a wave 4 channels x 64 pixels = 1 KiB
      const int q = lane & 15, cw = lane >> 4;
      if (p0 + h0 + 4 * q < hw) {                    // (hw % 4 == 0: a group of four is inside the image or outside as a whole)
        float* o = out + (size_t)b * C * hw + p0 + h0 + 4 * q;
        const float* src = tile + (h0 + 4 * q) * pitch;
#pragma unroll 2
        for (int c = wave * 4 + cw; c < C; c += kUpT / 16)
          *reinterpret_cast<float4*>(o + (size_t)c * hw) = make_float4(src[c], src[pitch + c], src[2 * pitch + c], src[3 * pitch + c]);
      }
    } else if (p0 + h0 + lane < hw) {
      float* o = out + (size_t)b * C * hw + p0 + h0 + lane;
      const float* src = tile + (h0 + lane) * pitch;
#pragma unroll 4
      for (int c = wave; c < C; c += kUpT / 64) o[(size_t)c * hw] = src[c];
    }
  }
}

// backward of the head: g NCHW fp32 -> g3 NHWC [B*H3*W3][ldc] (transpose), g4 NHWC [B*H4*W4][ldc] (transposed upsample).
// Block (b, i, 32-channel chunk): the four g rows y = 2i-1 .. 2i+2 of its channels go through LDS (coalesced NCHW row
// reads); from them it writes g3 rows 2i, 2i+1 and g4 row i with the channel axis across lanes.  (One thread per output
// element read g with a H3*W3*4-byte lane stride: 150 us for 40 MB.)
constexpr int kUpC = 32;
template <typename T>
__global__ void __launch_bounds__(256) upsample_add_bwd_kernel(const float* __restrict__ g, const float* __restrict__ wup, int B, int C, int ldc,
                                                               int H3, int W3, int H4, int W4, T* __restrict__ g3, T* __restrict__ g4) {
  extern __shared__ float slab[];                  // [4][kUpC][pitch]
  const int pitch = W3 | 1;
  const int chunks = ldc / kUpC;
  const int cc = blockIdx.x % chunks, i = (blockIdx.x / chunks) % H4, b = blockIdx.x / (chunks * H4);
  const int c0 = cc * kUpC;
  (void)B;
  // r4: a wave per (slab row, channel) line of W3 floats, eight lines per pass -- the flat-index loop of rounds 1-3 issued ONE load per
  // iteration (31 dependent round trips per thread, two runtime divisions each): 80 us for 48 MB
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int x0 = 0; x0 < W3; x0 += 64) {
      const int x = x0 + lane;
#pragma unroll 8
      for (int rc = wave; rc < 4 * kUpC; rc += 4) {
        const int r = rc / kUpC, cl = rc % kUpC;
        const int y = 2 * i - 1 + r, c = c0 + cl;
        float v = 0.f;
        if (x < W3 && (unsigned)y < (unsigned)H3 && c < C) v = g[(((size_t)b * C + c) * H3 + y) * W3 + x];
        if (x < W3) slab[(r * kUpC + cl) * pitch + x] = v;
      }
    }
  }
  // r6: the 16 upsample weights of the block's 32 channels, once, into LDS (the g4 loop read 16 scalars per element from global memory)
  float* const wl = slab + 4 * kUpC * pitch;       // [kUpC][16]
  for (int e = threadIdx.x; e < kUpC * 16; e += 256) wl[e] = (c0 + e / 16 < C) ? wup[(size_t)c0 * 16 + e] : 0.f;
  __syncthreads();
  // r6: a thread finishes EPS consecutive channels of one pixel and stores 16 bytes (the first form stored one 2-byte element per lane: 64-byte runs
  // assembled from 32 store lanes)
  constexpr int EPS = tf::Elem<T>::kPer16B, GPP = kUpC / EPS;       // channel groups per pixel
  // g3 rows 2i, 2i+1 (slab rows 1, 2)
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int y = 2 * i + rr;
    if (y >= H3) continue;
    for (int e = threadIdx.x; e < W3 * GPP; e += 256) {
      const int x = e / GPP, cl = (e % GPP) * EPS;
      float f[EPS];
#pragma unroll
      for (int q = 0; q < EPS; ++q) f[q] = slab[((1 + rr) * kUpC + cl + q) * pitch + x];
      *reinterpret_cast<uint4*>(g3 + (((size_t)b * H3 + y) * W3 + x) * ldc + c0 + cl) = tf::pack16<T>(f);
    }
  }
  // g4 row i
  for (int e = threadIdx.x; e < W4 * GPP; e += 256) {
    const int jx = e / GPP, cl = (e % GPP) * EPS;
    float f[EPS];
#pragma unroll
    for (int q = 0; q < EPS; ++q) {
      float v = 0.f;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const int x = 2 * jx - 1 + kx;
          if ((unsigned)x < (unsigned)W3) v += slab[(ky * kUpC + cl + q) * pitch + x] * wl[(cl + q) * 16 + ky * 4 + kx];     // rows outside H3 hold zeros
        }
      }
      f[q] = v;
    }
    *reinterpret_cast<uint4*>(g4 + (((size_t)b * H4 + i) * W4 + jx) * ldc + c0 + cl) = tf::pack16<T>(f);
  }
}

// sum over blocks of partial[blk][nk][ld] row k -> out[c]  (bias gradients)
__global__ void __launch_bounds__(kFinCh* kFinLanes) reduce_partials_kernel(float* __restrict__ partial, int nblk, int nk, int k, int ld, int C,
                                                                            float* __restrict__ out, int clear) {
  __shared__ double sums[1][kFinCh];
  const int kidx[1] = {k};
  reduce_partial_rows<1>(partial, nblk, nk, kidx, ld, C, sums, clear);
  const int c = blockIdx.x * kFinCh + threadIdx.x;
  if (threadIdx.x < kFinCh && c < C) out[c] = (float)sums[0][threadIdx.x];
}

inline unsigned grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                  \
  do {                                                          \
    if ((dtype) == TF_BF16) { using T = tf::bf16_t; __VA_ARGS__; } \
    else if ((dtype) == TF_F32) { using T = float; __VA_ARGS__; }  \
    else if ((dtype) == TF_F16) { using T = tf::f16_t; __VA_ARGS__; } \
    else return TF_ERR_UNSUPPORTED;                             \
  } while (0)

extern "C" int tf_pack_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, int transpose, int dtype, void* out,
                              int rows_pad, int cols_pad, void* stream) {
  if (!w_oihw || !out) return TF_ERR_ARG;
  const size_t total = (size_t)rows_pad * KH * KW * cols_pad;
  DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, KH, KW,
                                       transpose, (T*)out, rows_pad, cols_pad));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_pack_weights_batched(int dtype, const tf_pack_job* host_jobs, int njobs, void* stream) {
  if (njobs < 0 || (njobs > 0 && !host_jobs)) return TF_ERR_ARG;
  for (int j0 = 0; j0 < njobs; j0 += 64) {
    PackJobs pj;
    const int n = njobs - j0 < 64 ? njobs - j0 : 64;
    for (int k = 0; k < n; ++k) pj.j[k] = host_jobs[j0 + k];
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_batched_kernel<T>, dim3(48, n), dim3(256), 0, (hipStream_t)stream, pj));
  }
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_pack_weights_tiled(int dtype, const tf_pack2_job* host_jobs, int njobs, void* stream) {
  if (njobs < 0 || (njobs > 0 && !host_jobs)) return TF_ERR_ARG;
  for (int j0 = 0; j0 < njobs; j0 += 40) {
    Pack2Jobs pj;
    const int n = njobs - j0 < 40 ? njobs - j0 : 40;
    int tiles = 0, max_taps = 1;
    for (int k = 0; k < n; ++k) {
      const tf_pack2_job& J = host_jobs[j0 + k];
      if (!J.src || (!J.dst && !J.dst_t) || J.taps < 1 || J.taps > 9) return TF_ERR_ARG;
      if ((J.dst && J.cols_pad % 4) || (J.dst_t && J.cols_pad_t % 4)) return TF_ERR_ARG;      // 4 elements per store
      pj.j[k] = J;
      pj.tile0[k] = tiles;
      const int T_ = J.taps == 1 ? 64 : 32;
      const int ext_co = std::max(J.dst ? J.rows_pad : 0, J.dst_t ? J.cols_pad_t : 0);
      const int ext_ci = std::max(J.dst ? J.cols_pad : 0, J.dst_t ? J.rows_pad_t : 0);
      tiles += ((ext_co + T_ - 1) / T_) * ((ext_ci + T_ - 1) / T_);
      if (J.taps > max_taps) max_taps = J.taps;
    }
    pj.tile0[n] = tiles;
    const int T_ = max_taps == 1 ? 64 : 32;
    const size_t lds = (size_t)T_ * (T_ * max_taps + 1) * 4;
    const size_t lds1 = (size_t)64 * 65 * 4;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack2_kernel<T>, dim3(tiles), dim3(256), lds > lds1 ? lds : lds1, (hipStream_t)stream, pj, n));
  }
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_stem_im2col(const float* x_nchw, int N, int H, int W, int dtype, void* col, int ldc, void* stream) {
  if (!x_nchw || !col || ldc < 147 || ldc % 8) return TF_ERR_ARG;
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  const size_t total = (size_t)N * OH * OW * (ldc / (dtype == TF_F32 ? 4 : 8));
  (void)total;
  const unsigned tiles = (unsigned)N * ((OH + kI2cT - 1) / kI2cT) * ((OW + kI2cT - 1) / kI2cT);
  DISPATCH_T(dtype, hipLaunchKernelGGL(stem_im2col_kernel<T>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, x_nchw, N, H, W, OH, OW, (T*)col,
                                       ldc));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_maxpool_fwd(int dtype, const void* x, int N, int H, int W, int C, const float* scale, const float* shift, void* y,
                              uint8_t* argmax, void* stream) {
  if (!x || !y || C % 8) return TF_ERR_ARG;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * OH * OW * (C / (dtype == TF_F32 ? 4 : 8));
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, N, H, W, C,
                                       scale, shift, (T*)y, argmax, OH, OW));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_maxpool_bwd(int dtype, const void* g, const uint8_t* argmax, const void* x, const float* scale, const float* shift, int N,
                              int H, int W, int C, void* gz, void* stream) {
  if (!g || !argmax || !x || !scale || !shift || !gz || C % 8) return TF_ERR_ARG;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * H * W * (C / (dtype == TF_F32 ? 4 : 8));
  DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T, false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)g, argmax,
                                       (const T*)x, scale, shift, N, H, W, C, OH, OW, (T*)gz, (float*)nullptr, 0));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

// tf_maxpool_bwd + tf_colstats(gz, NULL, x) in one pass (r4): also folds  sum gz, sum gz * x  per channel into stat_out[rows][2][C]
// (rows = *rows_out <= tf_get_stat_rows(), zero on entry like every folded statistic region); C <= 256 and 256 % (C / elements per 16 bytes) == 0
extern "C" int tf_maxpool_bwd_stats(int dtype, const void* g, const uint8_t* argmax, const void* x, const float* scale, const float* shift, int N,
                                    int H, int W, int C, void* gz, float* stat_out, int* rows_out, void* stream) {
  const int eps = dtype == TF_F32 ? 4 : 8;
  if (!g || !argmax || !x || !scale || !shift || !gz || !stat_out || !rows_out || C % 8 || C > 256 || 256 % (C / eps)) return TF_ERR_ARG;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)N * H * W * (C / eps);
  // (every block folds 2 * C sums into the rows with atomics; capping the grid below the plain kernel's 8192 blocks was measured slower --
  //  103 us at 8192, 116 us at 1024 -- the longer grid-stride loops cost more than the atomics: TINYFACES_POOL_STATS_BLOCKS re-measures)
  const unsigned cap = (unsigned)tf::tuning().pool_stats_blocks;
  unsigned grid = grid_for(total);
  if (grid > cap && cap >= 1) grid = cap;
  const int srows = tf_get_stat_rows();
  if ((int)grid > srows && srows > TF_STAT_ROWS) return TF_ERR_UNSUPPORTED;       // unfolded (bit-reproducible) rows: the two-pass form
  *rows_out = (int)grid <= srows ? (int)grid : srows;
  DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)g, argmax,
                                       (const T*)x, scale, shift, N, H, W, C, OH, OW, (T*)gz, stat_out, srows));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

// rows handled by one colstats block: aim at ~1024 blocks (chip-filling), multiple of the rows in flight
static int colstats_rows(int M, int C, int dtype) {
  const int eps = dtype == TF_F32 ? 4 : 8;
  int rt = 256 / (C / eps);
  if (rt < 1) rt = 1;
  int rows = (M + 1023) / 1024;
  rows = (rows + rt - 1) / rt * rt;
  return rows < rt ? rt : rows;
}
extern "C" int tf_colstats_blocks(int M, int C, int dtype) {       // partial ROWS written (blocks are folded into <= TF_STAT_ROWS)
  const int rows = colstats_rows(M, C, dtype);
  const int nblk = (M + rows - 1) / rows;
  return nblk > tf_get_stat_rows() ? tf_get_stat_rows() : nblk;
}

extern "C" int tf_colstats(int dtype, const void* g, const void* y, const void* a, const void* b, int M, int C, int ld, float* partial,
                           void* stream) {
  const int eps = dtype == TF_F32 ? 4 : 8;
  if (!g || !partial || C % eps || C / eps > 256 || 256 % (C / eps)) return TF_ERR_ARG;
  const int nk = b ? 3 : (a ? 2 : 1);
  const int rt = 256 / (C / eps);
  const int rows = colstats_rows(M, C, dtype);
  const int nblk = (M + rows - 1) / rows;
  const size_t lds = (size_t)rt * nk * C * 4;
  if (lds > 64 * 1024) return TF_ERR_UNSUPPORTED;
#define COLSTATS_LAUNCH(HASY, NK)                                                                                                   \
  DISPATCH_T(dtype, hipLaunchKernelGGL((colstats_kernel<T, HASY, NK>), dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const T*)g, \
                                       (const T*)y, (const T*)a, (const T*)b, M, C, ld, rows, partial, tf_get_stat_rows()))
  if (y) { if (nk == 3) COLSTATS_LAUNCH(true, 3); else if (nk == 2) COLSTATS_LAUNCH(true, 2); else COLSTATS_LAUNCH(true, 1); }
  else   { if (nk == 3) COLSTATS_LAUNCH(false, 3); else if (nk == 2) COLSTATS_LAUNCH(false, 2); else COLSTATS_LAUNCH(false, 1); }
#undef COLSTATS_LAUNCH
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_finalize(const float* partial, int nblk, int ld, int C, float count, const float* gamma, const float* beta, float eps,
                              float momentum, float* scale, float* shift, float* mean, float* invstd, float* running_mean,
                              float* running_var, int clear, void* stream) {
  if (!partial || !gamma || !beta || !scale || !shift || !mean || !invstd) return TF_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinLanes), 0, (hipStream_t)stream, const_cast<float*>(partial), nblk, ld, C, count, clear,
                     gamma, beta, eps, momentum, scale, shift, mean, invstd, running_mean, running_var);
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                          float* scale, float* shift, void* stream) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, C,
                     scale, shift);
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_bwd_finalize(const float* partial, int nblk, int nk, int kidx, int ld, int C, float count, const float* gamma,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* cA, float* cB, float* cD,
                                  int clear, void* stream) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinLanes), 0, (hipStream_t)stream, const_cast<float*>(partial), nblk, nk, kidx, ld, C, clear, count,
                     gamma, mean, invstd, dgamma, dbeta, cA, cB, cD);
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_bwd_apply(int dtype, const void* g, const void* y, const void* x, const float* cA, const float* cB, const float* cD,
                               int64_t M, int C, void* out, void* stream) {
  if (!g || !x || !out || C % 8) return TF_ERR_ARG;
  const size_t total = (size_t)M * (C / (dtype == TF_F32 ? 4 : 8));
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)g, (const T*)y,
                                       (const T*)x, cA, cB, cD, (size_t)M, C, (T*)out));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_add_relu(int dtype, const void* x, const float* s1, const float* h1, const void* r, const float* s2, const float* h2,
                              int64_t M, int C, void* y, void* stream) {
  if (!x || !r || !y || !s1 || !h1 || C % 8) return TF_ERR_ARG;
  const size_t total = (size_t)M * (C / (dtype == TF_F32 ? 4 : 8));
  if (s2 && h2) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_add_relu_kernel<T, true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, s1,
                                         h1, (const T*)r, s2, h2, (size_t)M, C, (T*)y));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_add_relu_kernel<T, false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, s1,
                                         h1, (const T*)r, s2, h2, (size_t)M, C, (T*)y));
  }
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_bn_relu(int dtype, const void* x, const float* scale, const float* shift, int64_t M, int C, void* y, void* stream) {
  if (!x || !y || !scale || !shift || C % 8) return TF_ERR_ARG;
  const size_t total = (size_t)M * (C / (dtype == TF_F32 ? 4 : 8));
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_relu_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, scale, shift,
                                       (size_t)M, C, (T*)y));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_upsample_add_crop(int dtype, const void* s3, const void* s4, const float* wup_diag, int B, int C, int ldc, int H3, int W3,
                                    int H4, int W4, float* out_nchw, void* stream) {
  if (!s3 || !s4 || !wup_diag || !out_nchw || ldc % 4 || ldc < C) return TF_ERR_ARG;
  const int tiles = (H3 * W3 + kUpPx - 1) / kUpPx;
  const size_t lds = ((size_t)16 * ldc + (size_t)kUpPx * (ldc + 1)) * 4;   // weight table + [kUpPx pixels][ldc + 1] staging tile
  if (lds > 160 * 1024 || ldc % (dtype == TF_F32 ? 4 : 8)) return TF_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    static tf::PerDevice attr;
    if (attr.first()) { DISPATCH_T(dtype, (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upsample_add_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(upsample_add_kernel<T>, dim3(B * tiles), dim3(kUpT), lds, (hipStream_t)stream, (const T*)s3, (const T*)s4,
                                       wup_diag, B, C, ldc, H3, W3, H4, W4, out_nchw));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_upsample_add_crop_bwd(int dtype, const float* g_nchw, const float* wup_diag, int B, int C, int ldc, int H3, int W3, int H4,
                                        int W4, void* g3, void* g4, void* stream) {
  if (!g_nchw || !wup_diag || !g3 || !g4) return TF_ERR_ARG;
  if (ldc % kUpC || ldc < C || 2 * H4 < H3) return TF_ERR_ARG;
  const size_t lds = ((size_t)4 * kUpC * (W3 | 1) + kUpC * 16) * 4;        // four slab rows + the weight table
  if (lds > 160 * 1024) return TF_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    DISPATCH_T(dtype, (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upsample_add_bwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                160 * 1024));
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(upsample_add_bwd_kernel<T>, dim3((unsigned)(B * H4 * (ldc / kUpC))), dim3(256), lds, (hipStream_t)stream,
                                       g_nchw, wup_diag, B, C, ldc, H3, W3, H4, W4, (T*)g3, (T*)g4));
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_reduce_partials(const float* partial, int nblk, int nk, int k, int ld, int C, float* out, int clear, void* stream) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinLanes), 0, (hipStream_t)stream, const_cast<float*>(partial), nblk, nk, k, ld, C, out, clear);
  TF_CHECK_LAUNCH();
  return TF_OK;
}
