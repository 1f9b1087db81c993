// stem_conv: conv1 of the trunk (7x7 / stride 2 / pad 3, 3 -> 64 channels: tinyfaces/models/model.py:90, torchvision ResNet.conv1) computed
// STRAIGHT from the NCHW fp32 image (gfx950, 2-byte MFMA operand types).
//
// Rounds 1-3 ran it as tf_stem_im2col + a pointwise GEMM (K = 147 padded to 192): the im2col matrix of a bs = 12 500x500 batch is 288 MB --
// written once (132 us), read by the GEMM (96 us) and again by the weight gradient -- for a conv whose input is 36 MB and whose output is
// 96 MB.  Here a block owns a tile of 4 x 32 output pixels: it stages the 13 x 69 x 3 input patch in LDS (converted to the operand type: the
// same rounding the im2col applied), the 64 x 160 weight slab once for all its tiles, and every lane GATHERS its MFMA pixel fragments from
// the patch -- k = (c, kh, kw) is the OIHW order of conv1.weight, so the eight consecutive k of a 16x16x32 fragment are eight 2-byte LDS
// reads at offsets taken from a 160-entry table (one ds_read_b128 per k-step).  K-steps: 5 of 32 (k >= 147 reads a zero element; the
// sixth step of the padded K = 192 is all zero and is skipped).  Epilogue: BatchNorm batch statistics (sum, sum of squares per channel;
// training) or folded BN + ReLU (evaluation), output staged through LDS for full-line NHWC stores.
// MFMA roles as everywhere in this library: weights = row operand, pixels = column operand -> a lane's accumulator holds 4 consecutive
// channels of one pixel.
#include <cstdlib>
#include "common.h"
#include "tuning.h"
#include "profile.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 4, TW = 32;                       // output tile (pixels)
constexpr int PR = 2 * TH + 5, PCOLS = 2 * TW + 5;   // input patch rows / columns
constexpr int PC = PCOLS + 3;                        // patch pitch (elements): 72 = 6 segments of 12, rows 16-byte aligned
constexpr int PATCH = 3 * PR * PC;                   // + ZPAD zero elements behind it: k >= 147 gathers from there, whatever the pixel's base offset
constexpr int ZPAD = 512;                            // > the largest pixel base offset (2 * (TH - 1) * PC + 2 * (TW - 1) = 494)
constexpr int KSTEPS = 5, KUSED = 32 * KSTEPS;       // k < 160 (147 real)
constexpr int WP = KUSED + 8;                        // weight row pitch in LDS (elements): 336 bytes -> the 16 rows of a fragment read hit distinct bank groups
constexpr int SP = 64 + 8;                           // staging pitch (elements)
constexpr int SEG = 12, NSEG = PC / SEG;              // patch staging: thread -> (row of the 3 x PR rows, segment of 12 columns); 234 of 256 threads
static_assert(NSEG * SEG == PC && 3 * PR * NSEG <= 256, "patch staging layout");

template <typename T> struct Mma;
template <> struct Mma<tf::bf16_t> {
  typedef bf16x8 frag;
  __device__ static __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ uint16_t cvt(float f) { return tf::f32_to_bf16(f); }
};
template <> struct Mma<tf::f16_t> {
  typedef f16x8 frag;
  __device__ static __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ uint16_t cvt(float f) { return tf::f32_to_f16(f); }
};

struct StemK {
  const float* x; const char* w; char* y;
  const float* scale; const float* shift; float* stat_out;
  int N, H, W, OH, OW, ldw, tiles_w, tiles_h, ntiles, srows;
};

// EPI: 0 plain, 1 batch statistics (stat_out[rows][2][64] += sum, sum^2; rows zero on entry), 2 folded BN + ReLU (scale / shift)
template <typename T, int EPI>
__global__ void __launch_bounds__(256, 3) stem_conv_kernel(const StemK a) {
  __shared__ __attribute__((aligned(16))) uint16_t patch[PATCH + ZPAD];
  __shared__ __attribute__((aligned(16))) uint16_t wl[64 * WP];
  __shared__ __attribute__((aligned(16))) uint16_t koff[KUSED];
  __shared__ __attribute__((aligned(16))) uint16_t stg[TH * TW * SP];
  __shared__ float red[4][2][64];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, r = l & 15, g = l >> 4;

  // ---- once per block: weight slab (64 x 160 of the packed [>= 64][ldw] matrix) and the k -> patch-offset table
  for (int e = tid; e < 64 * (KUSED / 8); e += 256) {
    const int row = e / (KUSED / 8), s = e - row * (KUSED / 8);
    *reinterpret_cast<uint4*>(&wl[row * WP + s * 8]) = *reinterpret_cast<const uint4*>(a.w + ((size_t)row * a.ldw + s * 8) * 2);
  }
  for (int k = tid; k < KUSED; k += 256) {
    int o = PATCH;                                   // the zero element
    if (k < 147) { const int c = k / 49, q = k - c * 49, kh = q / 7, kw = q - kh * 7; o = (c * PR + kh) * PC + kw; }
    koff[k] = (uint16_t)o;
  }
  for (int e = tid; e < ZPAD; e += 256) patch[PATCH + e] = 0;

  float s1[4][4], s2[4][4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[n][i] = 0.f; s2[n][i] = 0.f; }
  float esc[4][4], esh[4][4];
  if (EPI == 2) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) { esc[n][i] = a.scale[n * 16 + g * 4 + i]; esh[n][i] = a.shift[n * 16 + g * 4 + i]; }
  }

  // the patch of a tile: thread -> one 12-column segment of one of the 3 x PR patch rows (tile-invariant: channel, row, first column), so a
  // tile costs one 64-bit base address and twelve loads at consecutive addresses; out-of-image elements are loaded from a clamped address
  // and zeroed afterwards (no exec-mask juggling); columns >= PCOLS are the zero pad of the pitch
  const int prow = tid / NSEG, pseg = tid - prow * NSEG;
  const bool pactive = prow < 3 * PR;
  const int pc_ = pactive ? prow / PR : 0, py_ = pactive ? prow - pc_ * PR : 0, px0_ = pseg * SEG;
  auto load_patch = [&](int tile, float (&v)[SEG]) {
    const int n = tile / (a.tiles_w * a.tiles_h), t2 = tile - n * a.tiles_w * a.tiles_h, tr = t2 / a.tiles_w, tc = t2 - tr * a.tiles_w;
    const int ih = tr * TH * 2 - 3 + py_, iw0 = tc * TW * 2 - 3 + px0_;
    const bool rowok = pactive && (unsigned)ih < (unsigned)a.H;
    const float* rowp = a.x + (((size_t)n * 3 + pc_) * a.H + (rowok ? ih : 0)) * a.W;
#pragma unroll
    for (int q = 0; q < SEG; ++q) {
      const int iw = iw0 + q;
      const int iwc = iw < 0 ? 0 : (iw >= a.W ? a.W - 1 : iw);
      const float t = rowp[iwc];
      v[q] = (rowok && iw == iwc && px0_ + q < PCOLS) ? t : 0.f;
    }
  };

  float pv[SEG];
  int tile = blockIdx.x;
  if (tile < a.ntiles) load_patch(tile, pv);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();                                 // previous tile's patch / staging fully consumed (first pass: tables written)
    if (pactive) {
      uint32_t w2[SEG / 2];
#pragma unroll
      for (int q = 0; q < SEG / 2; ++q) w2[q] = tf::pack2<T>(pv[2 * q], pv[2 * q + 1]);
      uint2* dst = reinterpret_cast<uint2*>(&patch[prow * PC + px0_]);
#pragma unroll
      for (int q = 0; q < SEG / 4; ++q) dst[q] = make_uint2(w2[2 * q], w2[2 * q + 1]);
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) load_patch(next, pv);       // in flight under this tile's gathers and MFMAs

    // ---- 32 pixels x 64 channels per wave: 2 pixel fragments x 4 channel fragments, 5 k-steps
    f32x4 acc[4][2];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[n][m] = f32x4(0.f);
    int pbase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int p = wave * 32 + m * 16 + r;          // pixel of the tile: row p / TW, column p % TW
      pbase[m] = (p / TW) * 2 * PC + (p % TW) * 2;
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const uint4 ko = *reinterpret_cast<const uint4*>(&koff[ks * 32 + g * 8]);
      const uint32_t kw_[4] = {ko.x, ko.y, ko.z, ko.w};
      typename Mma<T>::frag xf[2], wf[4];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t lo = patch[pbase[m] + (kw_[j] & 0xffffu)], hi = patch[pbase[m] + (kw_[j] >> 16)];
          q[j] = lo | (hi << 16);
        }
        const uint4 qq = make_uint4(q[0], q[1], q[2], q[3]);
        xf[m] = __builtin_bit_cast(typename Mma<T>::frag, qq);
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) wf[n] = *reinterpret_cast<const typename Mma<T>::frag*>(&wl[(n * 16 + r) * WP + ks * 32 + g * 8]);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[n][m] = Mma<T>::mma(wf[n], xf[m], acc[n][m]);
      __builtin_amdgcn_sched_barrier(0);             // one k-step of gathers live at a time (hoisting all five costs 80 registers and spills)
    }

    // ---- epilogue: lane holds channels n*16 + g*4 + {0..3} of pixels wave*32 + m*16 + r
    const int n_img = tile / (a.tiles_w * a.tiles_h), t2 = tile - n_img * a.tiles_w * a.tiles_h, tr = t2 / a.tiles_w, tc = t2 - tr * a.tiles_w;
    const int oh0 = tr * TH, ow0 = tc * TW;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int p = wave * 32 + m * 16 + r;
      const bool valid = oh0 + p / TW < a.OH && ow0 + p % TW < a.OW;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[n][m][i];
        if (EPI == 1) {
          if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1[n][i] += v[i]; s2[n][i] += v[i] * v[i]; }
          }
        }
        if (EPI == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * esc[n][i] + esh[n][i], 0.f);
        }
        const uint2 o = make_uint2(tf::pack2<T>(v[0], v[1]), tf::pack2<T>(v[2], v[3]));
        *reinterpret_cast<uint2*>(&stg[p * SP + n * 16 + g * 4]) = o;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < (TH * TW * 8) / 256; ++ps) {
      const int e = tid + ps * 256, p = e >> 3, ch = e & 7;
      const int oh = oh0 + p / TW, ow = ow0 + p % TW;
      if (oh < a.OH && ow < a.OW)
        *reinterpret_cast<uint4*>(a.y + ((((size_t)n_img * a.OH + oh) * a.OW + ow) * 64 + ch * 8) * 2) = *reinterpret_cast<const uint4*>(&stg[p * SP + ch * 8]);
    }
  }

  if (EPI == 1) {
    // column sums of the block: over the 16 pixel lanes of a fragment (xor 1, 2, 4, 8), then the four waves through LDS
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1[n][i] += __shfl_xor(s1[n][i], o, 64); s2[n][i] += __shfl_xor(s2[n][i], o, 64); }
      }
    if (r == 0) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) { red[wave][0][n * 16 + g * 4 + i] = s1[n][i]; red[wave][1][n * 16 + g * 4 + i] = s2[n][i]; }
    }
    __syncthreads();
    if (tid < 128) {
      const int k = tid >> 6, c = tid & 63;
      const float t = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
      if ((int)gridDim.x <= a.srows) a.stat_out[((size_t)blockIdx.x * 2 + k) * 64 + c] = t;
      else atomicAdd(&a.stat_out[((size_t)(blockIdx.x % a.srows) * 2 + k) * 64 + c], t);
    }
  }
}

template <typename T>
int launch(const StemK& k, int epi, unsigned grid, hipStream_t stream) {
  if (epi == 1) TF_LAUNCH_TIMED((stem_conv_kernel<T, 1>), dim3(grid), dim3(256), 0, stream, k);
  else if (epi == 2) TF_LAUNCH_TIMED((stem_conv_kernel<T, 2>), dim3(grid), dim3(256), 0, stream, k);
  else TF_LAUNCH_TIMED((stem_conv_kernel<T, 0>), dim3(grid), dim3(256), 0, stream, k);
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}


// ---------------------------------------------------------------- weight gradient of conv1, straight from the image
// dW[co][k] = sum over output pixels of g[px][co] * patch_px[k], k = (c, kh, kw): a GEMM whose reduction index is the pixel.  Rounds 1-3 ran it
// over the 288 MB im2col matrix.  Here a block walks the same 4 x 32-pixel tiles as the forward kernel and keeps its 64 x 160 partial result in
// MFMA accumulators over ALL its tiles.  Both MFMA operands must hold eight consecutive PIXELS per lane:
//   * the gradient tile is staged as it lies in memory (pixel-major rows of 128 bytes, 16-byte slots swizzled like csrc/wgrad_dma.hip) and read
//     with the transposing LDS read ds_read_b64_tr_b16: a lane gets pixels {g*4 .. g*4+3} and {16 + g*4 .. 16 + g*4+3} of its channel;
//   * eight consecutive output pixels of one row read input columns 2*pw + kw: every second element of a patch row.  The patch is staged split
//     by column parity (q = kw & 1) and in four copies shifted by s = kw >> 1 elements, so that the eight values are eight CONSECUTIVE elements
//     of copy (s, q): the lane's two pixel quads (the order the transposing read imposes) are two aligned ds_read_b64.  The 32 output pixels
//     of a tile row are one 32-deep k-step.
// A wave owns 2 channel fragments x 5 k fragments (16 x 16 each): 4 + 10 LDS reads per 10 MFMAs and step.  The partial sums go to the fp32 OIHW
// gradient with atomics (rows zero on entry, like the split-K weight gradients).
constexpr int CROW = 3 * PR;                         // 39 patch rows (channel, row)
constexpr int CW = TW + 8;                           // row pitch of a shifted / parity copy: 32 elements used, 80-byte rows spread the fragment reads over the banks
constexpr int COPY = CROW * CW;                      // one (shift, parity) copy
constexpr int NCOPY = 8;
constexpr int KFR = KUSED / 16;                      // 10 k fragments

// 16-byte slot swizzle of the pixel-major gradient tile and the transposing fragment read (the scheme of csrc/wgrad_dma.hip; lane mapping
// recorded by tests/test_gpu_small_ops.py::test_probe_tr16): 16 channels (c0 multiple of 16) x 32 pixels, lane element e = h*4 + j <-> pixel
// pk0 + h*16 + g*4 + j
__device__ __forceinline__ int wsw(int row) { return ((row >> 1) & 3) << 1; }
typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ typename Mma<T>::frag frag_tr(const char* tile, int pk0, int c0);
template <> __device__ __forceinline__ bf16x8 frag_tr<tf::bf16_t>(const char* tile, int pk0, int c0) {
  const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
  const int row = pk0 + g * 4 + (i >> 2);
  const int slot = (c0 >> 3) + ((i & 3) >> 1);
  const char* p = tile + row * 128 + ((slot ^ wsw(row)) << 4) + ((i & 1) << 3);
  typedef __attribute__((address_space(3))) bf16x4_ lds_v4;
  const bf16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(p));
  const bf16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(p + 16 * 128));       // wsw(row + 16) == wsw(row)
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <> __device__ __forceinline__ f16x8 frag_tr<tf::f16_t>(const char* tile, int pk0, int c0) {       // a 16-bit move: the element type does not matter
  return __builtin_bit_cast(f16x8, frag_tr<tf::bf16_t>(tile, pk0, c0));
}

struct StemWK {
  const float* x; const char* g; float* dw;
  const char* xc; const float* cA; const float* cB; const float* cD;     // APPLY: the conv output and the BN-backward coefficients
  int N, H, W, OH, OW, tiles_w, tiles_h, ntiles;
};

// APPLY (the training graph): the operand is  cA * g + cB * x_conv + cD  (tf_bn_bwd_apply of the stem's BatchNorm), formed while the gradient
// tile is staged -- the applied tensor has no other reader (conv1 has no data gradient), so the 288 MB apply pass disappears
template <typename T, bool APPLY>
__global__ void __launch_bounds__(256, 2) stem_wgrad_kernel(const StemWK a) {     // (1.5 blocks per CU are launched: registers over occupancy)
  __shared__ __attribute__((aligned(16))) char gt[TH * TW * 128];               // [pixel][64 channels], slot-swizzled
  __shared__ __attribute__((aligned(16))) uint16_t cp[NCOPY * COPY + CW];      // + one zero row (k >= 147)
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, r = l & 15, g = l >> 4;
  const int cf0 = (wave & 1) * 2, kf0 = (wave >> 1) * 5;

  for (int e = tid; e < CW; e += 256) cp[NCOPY * COPY + e] = 0;
  // element offset of this lane's k (one per owned k fragment) inside the copies: ((s * 2 + q) * CROW + c * PR + kh) * CW
  int kbase[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int k = (kf0 + j) * 16 + r;
    int o = NCOPY * COPY;
    if (k < 147) { const int c = k / 49, t = k - c * 49, kh = t / 7, kw = t - kh * 7; o = (((kw >> 1) * 2 + (kw & 1)) * CROW + c * PR + kh) * CW; }
    kbase[j] = o;
  }
  const bool kzero[5] = {kbase[0] == NCOPY * COPY, kbase[1] == NCOPY * COPY, kbase[2] == NCOPY * COPY, kbase[3] == NCOPY * COPY, kbase[4] == NCOPY * COPY};

  f32x4 acc[2][5];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[n][j] = f32x4(0.f);

  // staging roles: patch -- thread -> 12-column segment of one patch row (as in the forward kernel); gradient -- thread -> 16-byte chunk
  // (8 channels) of pixel tid / 8 + 32 * i
  const int prow = tid / NSEG, pseg = tid - prow * NSEG;
  const bool pactive = prow < CROW;
  const int pc_ = pactive ? prow / PR : 0, py_ = pactive ? prow - pc_ * PR : 0, px0_ = pseg * SEG;
  const int gch = tid & 7, gpx = tid >> 3;
  float fA[8], fB[8], fD[8];
  if constexpr (APPLY) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { fA[j] = a.cA[gch * 8 + j]; fB[j] = a.cB[gch * 8 + j]; fD[j] = a.cD[gch * 8 + j]; }
  }
  auto load_tile = [&](int tile, float (&v)[SEG], uint4 (&gq)[4], uint4 (&xq)[4]) {
    const int n = tile / (a.tiles_w * a.tiles_h), t2 = tile - n * a.tiles_w * a.tiles_h, tr = t2 / a.tiles_w, tc = t2 - tr * a.tiles_w;
    const int ih = tr * TH * 2 - 3 + py_, iw0 = tc * TW * 2 - 3 + px0_;
    const bool rowok = pactive && (unsigned)ih < (unsigned)a.H;
    const float* rowp = a.x + (((size_t)n * 3 + pc_) * a.H + (rowok ? ih : 0)) * a.W;
#pragma unroll
    for (int q = 0; q < SEG; ++q) {
      const int iw = iw0 + q;
      const int iwc = iw < 0 ? 0 : (iw >= a.W ? a.W - 1 : iw);
      const float t = rowp[iwc];
      v[q] = (rowok && iw == iwc && px0_ + q < PCOLS) ? t : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = gpx + 32 * i, oh = tr * TH + p / TW, ow = tc * TW + p % TW;
      const bool ok = oh < a.OH && ow < a.OW;
      const size_t o = ((((size_t)n * a.OH + (ok ? oh : 0)) * a.OW + (ok ? ow : 0)) * 64 + gch * 8) * 2;
      const uint4 t = *reinterpret_cast<const uint4*>(a.g + o);
      gq[i] = ok ? t : make_uint4(0, 0, 0, 0);
      if constexpr (APPLY) xq[i] = *reinterpret_cast<const uint4*>(a.xc + o);
    }
  };

  float pv[SEG];
  uint4 gq[4], xq[4];                               // (xq: unused and optimised away without APPLY)
  bool gok[4];
  auto tile_ok = [&](int tile) {
    const int n = tile / (a.tiles_w * a.tiles_h), t2 = tile - n * a.tiles_w * a.tiles_h, tr = t2 / a.tiles_w, tc = t2 - tr * a.tiles_w;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int p = gpx + 32 * i; gok[i] = tr * TH + p / TW < a.OH && tc * TW + p % TW < a.OW; }
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) load_tile(tile, pv, gq, xq);
  for (; tile < a.ntiles; tile += gridDim.x) {
    if constexpr (APPLY) tile_ok(tile);
    __syncthreads();                                 // the previous tile's fragments are read
    if (pactive) {
      // the thread's 12 columns = 6 even + 6 odd; in copy (sft, par) they are the 6 consecutive elements i0 .. i0 + 5, i0 = pseg * 6 - sft:
      // 32-bit writes where the pair is aligned (even sft), 16-bit at the two ends otherwise
      uint16_t ev[2][SEG / 2];
#pragma unroll
      for (int q = 0; q < SEG; ++q) ev[q & 1][q >> 1] = Mma<T>::cvt(pv[q]);
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int sft = 0; sft < 4; ++sft) {
          uint16_t* row = &cp[((sft * 2 + par) * CROW + prow) * CW];
          const int i0 = pseg * (SEG / 2) - sft;
          if ((sft & 1) == 0) {
#pragma unroll
            for (int e = 0; e < SEG / 2; e += 2) {
              const int i = i0 + e;
              if (i >= 0 && i < TW) *reinterpret_cast<uint32_t*>(row + i) = (uint32_t)ev[par][e] | ((uint32_t)ev[par][e + 1] << 16);
            }
          } else {
            if (i0 >= 0 && i0 < TW) row[i0] = ev[par][0];
#pragma unroll
            for (int e = 1; e + 1 < SEG / 2; e += 2) {
              const int i = i0 + e;
              if (i >= 0 && i < TW) *reinterpret_cast<uint32_t*>(row + i) = (uint32_t)ev[par][e] | ((uint32_t)ev[par][e + 1] << 16);
            }
            if (i0 + SEG / 2 - 1 >= 0 && i0 + SEG / 2 - 1 < TW) row[i0 + SEG / 2 - 1] = ev[par][SEG / 2 - 1];
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = gpx + 32 * i;
      uint4 v = gq[i];
      if constexpr (APPLY) {
        float gf[8], xf[8];
        tf::unpack16<T>(gq[i], gf);
        tf::unpack16<T>(xq[i], xf);
#pragma unroll
        for (int j = 0; j < 8; ++j) gf[j] = gok[i] ? fA[j] * gf[j] + fB[j] * xf[j] + fD[j] : 0.f;
        v = tf::pack16<T>(gf);
      }
      *reinterpret_cast<uint4*>(gt + p * 128 + ((gch ^ wsw(p)) << 4)) = v;
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < a.ntiles) load_tile(next, pv, gq, xq);

#pragma unroll
    for (int st = 0; st < TH; ++st) {                // k-step = the 32 pixels of tile row st
      typename Mma<T>::frag af[2], bf[5];
#pragma unroll
      for (int n = 0; n < 2; ++n) af[n] = frag_tr<T>(gt, st * TW, (cf0 + n) * 16);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint16_t* bp = &cp[kbase[j] + (kzero[j] ? 0 : 2 * st * CW) + g * 4];
        const uint2 lo = *reinterpret_cast<const uint2*>(bp), hi = *reinterpret_cast<const uint2*>(bp + (kzero[j] ? 0 : 16));
        const uint4 qq = make_uint4(lo.x, lo.y, hi.x, hi.y);
        bf[j] = __builtin_bit_cast(typename Mma<T>::frag, qq);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[n][j] = Mma<T>::mma(af[n], bf[j], acc[n][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // lane: channels (cf0 + n) * 16 + g * 4 + {0..3}, k = (kf0 + j) * 16 + r
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int k = (kf0 + j) * 16 + r;
      if (k < 147) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&a.dw[(size_t)((cf0 + n) * 16 + g * 4 + i) * 147 + k], acc[n][j][i]);
      }
    }
}

}  // namespace

// conv1 (7x7 / stride 2 / pad 3, 3 -> 64; model.py:90) from the NCHW fp32 image: y [N*OH*OW][64] of `dtype` (TF_BF16 | TF_F16; TF_F32 keeps
// tf_stem_im2col + tf_conv2d).  w_packed: conv1.weight as tf_pack_weight writes it for the im2col GEMM ([>= 64 rows][ldw], k = c*49 + kh*7 + kw,
// ldw >= 160, multiple of 8).  epi: 0 | TF_EPI_STATS (stat_out[rows][2][64] += sum, sum of squares per channel, rows = *host_rows_out <=
// tf_get_stat_rows(), zero on entry; TF_ERR_UNSUPPORTED with unfolded rows) | TF_EPI_AFFINE|TF_EPI_RELU (y = relu(conv * scale + shift)).
extern "C" int tf_stem_conv(int dtype, const float* x_nchw, int N, int H, int W, const void* w_packed, int ldw, void* y, int epi, const float* scale,
                            const float* shift, float* stat_out, int* host_rows_out, void* stream_) {
  if (!x_nchw || !w_packed || !y || N < 1 || H < 1 || W < 1 || ldw < KUSED || ldw % 8) return TF_ERR_ARG;
  if (dtype != TF_BF16 && dtype != TF_F16) return TF_ERR_UNSUPPORTED;
  int mode = 0;
  if (epi == TF_EPI_STATS) mode = 1;
  else if (epi == (TF_EPI_AFFINE | TF_EPI_RELU)) mode = 2;
  else if (epi != 0) return TF_ERR_UNSUPPORTED;
  if (mode == 1 && (!stat_out || !host_rows_out)) return TF_ERR_ARG;
  if (mode == 2 && (!scale || !shift)) return TF_ERR_ARG;
  StemK k;
  k.x = x_nchw; k.w = (const char*)w_packed; k.y = (char*)y; k.scale = scale; k.shift = shift; k.stat_out = stat_out;
  k.N = N; k.H = H; k.W = W; k.OH = (H + 6 - 7) / 2 + 1; k.OW = (W + 6 - 7) / 2 + 1; k.ldw = ldw;
  k.tiles_w = (k.OW + TW - 1) / TW; k.tiles_h = (k.OH + TH - 1) / TH;
  const long nt = (long)N * k.tiles_w * k.tiles_h;
  if (nt > 0x7fffffffL) return TF_ERR_ARG;
  k.ntiles = (int)nt; k.srows = tf_get_stat_rows();
  // three blocks per CU, each walking several tiles: the weight slab and the offset table are staged once per block
  unsigned grid = 256 * 3;
  if ((long)grid > nt) grid = (unsigned)nt;
  if (mode == 1) {
    if ((int)grid > k.srows && k.srows > TF_STAT_ROWS) return TF_ERR_UNSUPPORTED;
    *host_rows_out = (int)grid <= k.srows ? (int)grid : k.srows;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const double M = (double)N * k.OH * k.OW;
  const double bytes = (double)N * 3 * H * W * 4 + 64.0 * 147 * 2 + M * 64 * 2;
  tf::ProfScope prof(dtype == TF_BF16 ? 20 : 21, 2.0 * M * 64 * 147, bytes, stream, (int)M, 64, 147, 49, 0, epi, 2.0 * M * 64 * KUSED, true);   // 20 / 21 = stem_conv bf16 / f16
  return dtype == TF_BF16 ? launch<tf::bf16_t>(k, mode, grid, stream) : launch<tf::f16_t>(k, mode, grid, stream);
}

// weight gradient of conv1 straight from the image (r4): dw_oihw[64][147] (fp32, conv1.weight's OIHW order) += sum over output pixels of
// g[px][co] * patch[px][k]; g = the gradient w.r.t. the conv output, [N*OH*OW][64] of `dtype` (TF_BF16 | TF_F16); the image is rounded to
// `dtype` like the forward operand.  dw must be zero (or hold what is to be accumulated into) on entry.  Replaces tf_stem_im2col +
// tf_conv2d_wgrad over the 147-column matrix.  With x_conv != NULL the gradient operand is  cA * g + cB * x_conv + cD  per channel (rounded
// to `dtype`): tf_bn_bwd_apply of the stem's BatchNorm folded into the staging of the gradient tile.
extern "C" int tf_stem_wgrad(int dtype, const float* x_nchw, int N, int H, int W, const void* g, const void* x_conv, const float* cA, const float* cB,
                             const float* cD, float* dw_oihw, void* stream_) {
  if (!x_nchw || !g || !dw_oihw || N < 1 || H < 1 || W < 1) return TF_ERR_ARG;
  if (dtype != TF_BF16 && dtype != TF_F16) return TF_ERR_UNSUPPORTED;
  const bool apply = x_conv != nullptr;
  if (apply && (!cA || !cB || !cD)) return TF_ERR_ARG;
  StemWK k;
  k.x = x_nchw; k.g = (const char*)g; k.dw = dw_oihw; k.xc = (const char*)x_conv; k.cA = cA; k.cB = cB; k.cD = cD;
  k.N = N; k.H = H; k.W = W; k.OH = (H + 6 - 7) / 2 + 1; k.OW = (W + 6 - 7) / 2 + 1;
  k.tiles_w = (k.OW + TW - 1) / TW; k.tiles_h = (k.OH + TH - 1) / TH;
  const long nt = (long)N * k.tiles_w * k.tiles_h;
  if (nt > 0x7fffffffL) return TF_ERR_ARG;
  k.ntiles = (int)nt;
  // 384 blocks: 1.5 per CU -- more blocks overlap their staging and MFMA phases better but every block adds 9408 atomics (microbenchmark at
  // bs = 12: 256 blocks 118 us, 384: 91, 512: 95, 768: 126)
  const unsigned want = (unsigned)tf::tuning().stem_wgrad_blocks;
  unsigned grid = want < 1 ? 1 : want;
  if ((long)grid > nt) grid = (unsigned)nt;
  hipStream_t stream = (hipStream_t)stream_;
  const double M = (double)N * k.OH * k.OW;
  tf::ProfScope prof(22, 2.0 * M * 64 * 147, (double)N * 3 * H * W * 4 + M * 64 * 2 * (apply ? 2 : 1) + 64.0 * 147 * 4, stream, (int)M, 64, 147, 49, 2, 0,
                     2.0 * M * 64 * KUSED, true);   // 22 = stem_wgrad
  if (dtype == TF_BF16) {
    if (apply) TF_LAUNCH_TIMED((stem_wgrad_kernel<tf::bf16_t, true>), dim3(grid), dim3(256), 0, stream, k);
    else TF_LAUNCH_TIMED((stem_wgrad_kernel<tf::bf16_t, false>), dim3(grid), dim3(256), 0, stream, k);
  } else {
    if (apply) TF_LAUNCH_TIMED((stem_wgrad_kernel<tf::f16_t, true>), dim3(grid), dim3(256), 0, stream, k);
    else TF_LAUNCH_TIMED((stem_wgrad_kernel<tf::f16_t, false>), dim3(grid), dim3(256), 0, stream, k);
  }
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}
