// conv_pws (r5): WAVE-AUTONOMOUS streaming pointwise conv for the short-K / large-M launches of the trunk -- layer 1 at bs = 12
// (M = 187 500 pixels, 64 <-> 256 channels), the stem-side layers of the large pyramid levels (M = 76 800 / 307 200).
//
// Why.  These launches are HBM streams with a tiny GEMM in the middle (120 MB moved for 6 GFLOP).  The tiled LDS-DMA kernel runs them as 5860
// short-lived blocks, each of which fetches the same 16 KiB weight tile again, waits out a full memory round trip with nothing behind it, parks
// its accumulators in a block-wide staging tile and dies.  The memory skeleton of the form below -- csrc/probe.hip kind 11,
// profiles/r05_conv_pws.txt -- moves the 64 -> 256 layer-1 tensor pair in 24 us (4.95 TB/s); this kernel needs 30 us, conv_dma 32-39.
//
// How.  A PERSISTENT block of two to four waves loads the whole weight matrix (K x N x 2 bytes <= 64 KiB) into LDS once; where it is <= 32 KiB
// every wave then copies its 32 MFMA fragments into 128 VGPRs and never reads the slab again.  After that every wave is a pipeline of its own
// over tiles of 16 pixels dealt round-robin: LDS-DMA of the tile's input rows and of up to three epilogue operands (residual / mask / statistic
// tensors) into a PRIVATE 2-slot ring; ONE counted `s_waitcnt vmcnt` per tile -- the VM counter retires in order and counts stores, so the
// count leaves the previous tile's output stores and the next tile's DMAs in flight --; 16x16x32 MFMAs (accumulators: 4 consecutive channels of
// one pixel per lane, in AGPRs); accumulators -> a private fp32 staging tile of 8 pixels -> every lane finishes 8 channels of one pixel
// (affine / residual / ReLU / masks, statistic sums in registers across ALL its tiles) and stores 16 bytes.  No block barrier after the weight
// load, no LDS shared between waves but the read-only weights: the only synchronisation of the steady state is a wave's own vmcnt / lgkmcnt
// (ISA audit: tests/test_cabi.py::test_conv_pws_tile_loop_keeps_its_ring_rules).  Statistic sums meet in LDS once at the end: one atomic per
// block and channel.
// Epilogue sets (compile-time, as in conv_dma): STATS | AFFINE+RELU | AFFINE | AFFINE+RES+RELU | MASK+STATS2 | none, and the hand-over sets of the
// data gradient of conv1 (RES+MASK2+STATS3, RES+MASK2, RES: three operand rings per wave -> two waves per block).
// Output-channel slices (N > 256: 256 -> 512 / 1024, 128 -> 512): a block keeps a 128-channel slice resident and the nsl blocks that walk the
// same pixel tiles sit on one XCD.  Parity-green and slower than the tiled kernel (64 KiB of weights leave room for two waves x one tile in
// flight): on request (tile = 70) or with TINYFACES_PWS_SLICED=1 only.
// Replaces nn.Conv2d (1x1) + BN statistics / folded BN (+ residual + ReLU) of the torchvision Bottleneck (tinyfaces/models/model.py:90-101)
// and the data gradient of conv3 for those shapes.  bf16 and fp16 operands.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "tuning.h"
#include "lds_dma.h"
#include "profile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ uint4 g_pws_zero[8];

constexpr int EPS = 8;

struct PwK {
  const char* x; const char* w; char* y; const char* aux; const char* aux2; const char* aux3;
  const float* epi_scale; const float* epi_shift; const float* mask_scale; const float* mask_shift;
  float* stat_out; const float* stat_shift; float* stat_shift_out;
  int M, K, ntiles, srows, ldy;
  int nsl, nb;       // output-channel slices of N = NF * 16 channels (1: the whole matrix is resident), blocks per slice: grid = nsl * nb
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) << 2) ^ (((row >> 3) & 1) * 6); }
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ swz(row)) << 4); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T> struct Fr;
template <> struct Fr<tf::bf16_t> {
  typedef bf16x8 t;
  __device__ static __forceinline__ f32x4 mma(t a, t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Fr<tf::f16_t> {
  typedef f16x8 t;
  __device__ static __forceinline__ f32x4 mma(t a, t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// KS = K / 64 (64-deep k stages), NF = N / 16 (16-channel accumulator tiles per wave)
// NW = waves per block: as many (<= 4) as have room for their private rings beside the weight slab
constexpr int pws_naux(int epic) { return ((epic & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) ? 1 : 0) + ((epic & TF_EPI_MASK2) ? 1 : 0) + ((epic & TF_EPI_STATS3) ? 1 : 0); }
constexpr int pws_per(int ks, int n, int naux) { return 2 * ks * 2048 + 2 * naux * 16 * n * 2 + 8 * (n + 4) * 4; }      // private LDS of a wave
constexpr bool pws_wreg(int ks, int nf) { return ks * 2 * nf <= 32; }

template <typename T, int KS, int NF, int EPIC, int NW>
__global__ void __launch_bounds__(NW * 64) conv_pws_kernel(const PwK a) {
  typedef typename Fr<T>::t frag;
  constexpr int N = NF * 16, K = KS * 64;
  constexpr int NAUX = ((EPIC & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) ? 1 : 0) + ((EPIC & TF_EPI_MASK2) ? 1 : 0) + ((EPIC & TF_EPI_STATS3) ? 1 : 0);
  constexpr bool HAS_AUX = NAUX > 0;
  constexpr bool HAS_STATS = (EPIC & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) != 0;
  constexpr bool WREG = pws_wreg(KS, NF);           // the whole weight matrix as MFMA fragments in <= 128 registers of every wave
  constexpr int SLAB = KS * N * 128;                    // weights: [k stage][N rows][128 B], swizzled like every operand tile of the conv kernels
  constexpr int XT = KS * 2048;                     // input tile of a wave: [k stage][16 pixels][128 B]
  constexpr int AUXT = 16 * N * 2;                  // epilogue operand of a tile: 16 pixel rows, plain
  constexpr int PITCH = N + 4, STG = 8 * PITCH * 4; // staging: 8 pixels x N fp32
  constexpr int PER = 2 * XT + 2 * NAUX * AUXT + STG;
  constexpr int NIX = 2 * KS, NI1 = AUXT / 1024, NIA = NAUX * NI1;  // DMA instructions per tile
  constexpr int CPR = N / EPS, RPP = 64 / CPR, NPASS = 8 / RPP;     // 16-byte chunks per pixel row, rows per pass, passes per 8 pixels
  constexpr int NST = 2 * NPASS;                                    // output stores of a wave per tile (they count on the VM counter like the DMAs)
  static_assert(NST + 2 * (NIX + NIA) <= 63, "conv_pws: a tile's stores and two tiles' DMAs must fit the 6-bit VM counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t smem_u = tf::lds_addr_uniform(smem);
  const char* zero = reinterpret_cast<const char*>(g_pws_zero) + (lane & 7) * 16;
  // output-channel slices (r5b: N > 256, e.g. 256 -> 1024 of layer 3 as 8 slices of 128): a block keeps ONE slice of the weights resident and
  // walks the pixel tiles of its share.  Blocks are dealt to the 8 XCDs round-robin by the hardware; the mapping below puts the nsl blocks
  // that walk the SAME tiles on the same XCD, so the input rows they all read come from HBM once and from that XCD's L2 nsl - 1 times.
  int slice = 0, bi = blockIdx.x;
  if (a.nsl > 1) {
    if ((a.nb & 7) == 0) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; slice = j % a.nsl; bi = (j / a.nsl) * 8 + xcd; }
    else { slice = blockIdx.x % a.nsl; bi = blockIdx.x / a.nsl; }
  }
  const int n0 = slice * N;

  // ---- the weights, once: N x KS rows of 128 B, 8 rows per wave-level DMA
  for (int i = wave; i < KS * N / 8; i += NW) {
    const int ks = i / (N / 8), rb = i - ks * (N / 8), row = rb * 8 + (lane >> 3);
    tf::dma16_hidden(a.w + ((size_t)(n0 + row) * K + ks * 64) * sizeof(T) + (((lane & 7) ^ swz(row)) << 4), smem_u + ks * (N * 128) + rb * 1024);
  }

  // ---- per-lane constants of the store phase: lane -> 8 channels c0 .. c0+7 of pixel rows (lane / CPR) + RPP * pass
  const int chunk = lane % CPR, prow = lane / CPR, c0 = chunk * EPS, cg = n0 + c0;
  float es[EPS], eh[EPS], ms[EPS], mh[EPS], sft[EPS], s1[EPS], s2[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { es[j] = 1.f; eh[j] = 0.f; ms[j] = 0.f; mh[j] = 0.f; sft[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  if constexpr ((EPIC & TF_EPI_AFFINE) != 0) {
#pragma unroll
    for (int j = 0; j < EPS; ++j) { es[j] = a.epi_scale[cg + j]; eh[j] = a.epi_shift[cg + j]; }
  }
  if constexpr ((EPIC & TF_EPI_MASK) != 0) {
#pragma unroll
    for (int j = 0; j < EPS; ++j) { ms[j] = a.mask_scale[cg + j]; mh[j] = a.mask_shift[cg + j]; }
  }
  if constexpr ((EPIC & TF_EPI_STATS) != 0) {
    if (a.stat_shift) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) sft[j] = a.stat_shift[cg + j];
    }
  }

  // ---- this wave's private LDS and its tile walk
  const uint32_t mine_u = smem_u + SLAB + wave * PER;
  char* const mine = smem + SLAB + wave * PER;
  const int stride = a.nb * NW;
  int t = bi * NW + wave;
  auto issue = [&](int tile, int slot) {
    const int p0 = tile * 16;
    const bool live = tile < a.ntiles;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 8 + (lane >> 3);
        const bool ok = live && p0 + row < a.M;
        tf::dma16_hidden(ok ? a.x + ((size_t)(p0 + row) * K + ks * 64) * sizeof(T) + (((lane & 7) ^ swz(row)) << 4) : zero,
                         mine_u + slot * XT + ks * 2048 + i * 1024);
      }
    if constexpr (HAS_AUX) {
#pragma unroll
      for (int q = 0; q < NAUX; ++q) {
        const char* src = (q == 0 ? a.aux : (q == 1 ? a.aux2 : a.aux3)) + (size_t)n0 * sizeof(T);
#pragma unroll
        for (int i = 0; i < NI1; ++i) {
          const int byte = i * 1024 + lane * 16, row = byte / (N * 2);
          const bool ok = live && p0 + row < a.M;
          tf::dma16_hidden(ok ? src + (size_t)p0 * a.ldy * sizeof(T) + (size_t)row * (a.ldy - N) * sizeof(T) + byte : zero,
                           mine_u + 2 * XT + (slot * NAUX + q) * AUXT + i * 1024);
        }
      }
    }
  };
  issue(t, 0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIX + NIA) : "memory");     // the weight pieces this wave requested (they were issued before the first tile)
  __syncthreads();                                                      // everyone's pieces of the weights are in LDS: the only block barrier of the loop

  const int r = lane & 15, g = lane >> 4;
  // r5b: where the matrix is <= 32 KiB a wave keeps its fragments in REGISTERS for all its tiles (128 VGPRs; the accumulators live in AGPRs): the
  // slab is read once per wave instead of once per 16-pixel tile, 34 instead of 66 KiB of LDS traffic per tile -- the LDS pipe, not HBM, was
  // what bounded the 64 -> 256 / 256 -> 64 launches.  (Fragments straight from global memory, no slab, two blocks per CU at <= 256 registers:
  // measured equal in the step and slower alone, profiles/r05_conv_pws.txt section 5.)
  frag wreg[WREG ? KS * 2 : 1][WREG ? NF : 1];
  if constexpr (WREG) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int n = 0; n < NF; ++n) wreg[ks * 2 + kk][n] = *reinterpret_cast<const frag*>(smem + ks * (N * 128) + lds_off(n * 16 + r, kk * 4 + g));
  }
  float* const stg = reinterpret_cast<float*>(mine + 2 * XT + 2 * NAUX * AUXT);
  int slot = 0;
  bool first = true;
  for (; t < a.ntiles; t += stride) {
    issue(t + stride, slot ^ 1);                    // (past the end: zero-page pieces into the free slot, the wait count stays a constant)
    // this tile's pieces landed.  The VM counter retires in order and counts stores too: behind this tile's DMAs the wave issued the NST
    // output stores of the previous tile and the next tile's DMAs, and none of those need to be back -- waiting for the stores (the first
    // form of this loop: vmcnt(NIX + NIA)) put the write acknowledge of every tile on the critical path of the next one.  The count is
    // exact: a tile that is followed by another one is never the ragged last tile of the tensor, so all its NST stores were issued.
    if (first) { wait_vmcnt<NIX + NIA>(); first = false; }
    else wait_vmcnt<NST + NIX + NIA>();
    f32x4 acc[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xs = mine + slot * XT;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const frag xf = *reinterpret_cast<const frag*>(xs + ks * 2048 + lds_off(r, kk * 4 + g));
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          if constexpr (WREG) acc[n] = Fr<T>::mma(wreg[ks * 2 + kk][n], xf, acc[n]);
          else {
            const frag wf = *reinterpret_cast<const frag*>(smem + ks * (N * 128) + lds_off(n * 16 + r, kk * 4 + g));
            acc[n] = Fr<T>::mma(wf, xf, acc[n]);
          }
        }
      }
    // ---- epilogue, 8 pixels at a time: lane (pixel r, channel quad g of every 16-channel tile) parks, lane (row prow, chunk) finishes
    const int p0 = t * 16;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((r >> 3) == half) {
#pragma unroll
        for (int n = 0; n < NF; ++n) *reinterpret_cast<f32x4*>(stg + (r & 7) * PITCH + n * 16 + g * 4) = acc[n];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (one wave: LDS executes its accesses in order; the wait orders the compiler)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int row8 = ps * RPP + prow, row = half * 8 + row8, p = p0 + row;
        float v[EPS];
        {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row8 * PITCH + c0), hi = *reinterpret_cast<const f32x4*>(stg + row8 * PITCH + c0 + 4);
          v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        }
        float ax[EPS], y2[EPS], x3[EPS];
        const char* auxb = mine + 2 * XT + slot * NAUX * AUXT + row * (N * 2) + chunk * 16;
        if constexpr (HAS_AUX) tf::unpack16<T>(*reinterpret_cast<const uint4*>(auxb), ax);
        if constexpr ((EPIC & TF_EPI_MASK2) != 0) tf::unpack16<T>(*reinterpret_cast<const uint4*>(auxb + AUXT), y2);
        if constexpr ((EPIC & TF_EPI_STATS3) != 0) tf::unpack16<T>(*reinterpret_cast<const uint4*>(auxb + (NAUX - 1) * AUXT), x3);
        if (p < a.M) {
          if constexpr ((EPIC & TF_EPI_STATS) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) { const float d = v[j] - sft[j]; s1[j] += d; s2[j] += d * d; }
          }
          if constexpr ((EPIC & TF_EPI_AFFINE) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) v[j] = v[j] * es[j] + eh[j];
          }
          if constexpr ((EPIC & TF_EPI_RES) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) v[j] += ax[j];
          }
          if constexpr ((EPIC & TF_EPI_MASK) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) v[j] = (ax[j] * ms[j] + mh[j] > 0.f) ? v[j] : 0.f;
          }
          if constexpr ((EPIC & TF_EPI_MASK2) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) v[j] = (y2[j] > 0.f) ? v[j] : 0.f;
          }
          if constexpr ((EPIC & TF_EPI_RELU) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if constexpr ((EPIC & TF_EPI_STATS2) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * ax[j]; }
          }
          if constexpr ((EPIC & TF_EPI_STATS3) != 0) {
#pragma unroll
            for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * x3[j]; }
          }
          *reinterpret_cast<uint4*>(a.y + ((size_t)p * a.ldy + cg) * sizeof(T)) = tf::pack16<T>(v);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the staging tile is consumed before the other half overwrites it
    }
    slot ^= 1;
  }
  wait_vmcnt<0>();

  // ---- statistic sums: lanes sharing a chunk -> one value per wave, the four waves meet in LDS, ONE atomic per block, sum and channel
  if constexpr (HAS_STATS) {
#pragma unroll
    for (int j = 0; j < EPS; ++j) { s1[j] = tf::lane_group_sum<(CPR < 64 ? CPR : 64)>(s1[j]); s2[j] = tf::lane_group_sum<(CPR < 64 ? CPR : 64)>(s2[j]); }
    __syncthreads();                                 // every wave left its loop: the private regions can be reused
    float* red = reinterpret_cast<float*>(smem + SLAB);     // [NW][2][N]
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { red[(wave * 2 + 0) * N + lane * EPS + j] = s1[j]; red[(wave * 2 + 1) * N + lane * EPS + j] = s2[j]; }
    }
    __syncthreads();
    const int srow = bi % a.srows;
    for (int e = tid; e < 2 * N; e += NW * 64) {
      const int k = e / N, c = e - k * N;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) v += red[(wv * 2 + k) * N + c];
      atomicAdd(&a.stat_out[((size_t)srow * 2 + k) * a.ldy + n0 + c], v);
    }
    if ((EPIC & TF_EPI_STATS) && a.stat_shift && a.stat_shift_out && bi == 0)
      for (int c = tid; c < N; c += NW * 64) a.stat_shift_out[n0 + c] = a.stat_shift[n0 + c];
  }
}

template <typename T, int KS, int NF, int EPIC>
int launch_one(const tf_conv_args* A, const PwK& k, hipStream_t stream) {
  constexpr int N = NF * 16;
  constexpr int NAUX = ((EPIC & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) ? 1 : 0) + ((EPIC & TF_EPI_MASK2) ? 1 : 0) + ((EPIC & TF_EPI_STATS3) ? 1 : 0);
  constexpr size_t per = 2 * KS * 2048 + 2 * NAUX * 16 * N * 2 + 8 * (N + 4) * 4;
  constexpr size_t slab = (size_t)KS * N * 128;        // (<= 32 KiB: register-resident, see the kernel)
  constexpr int NW = slab + 4 * per <= 160 * 1024 ? 4 : (slab + 3 * per <= 160 * 1024 ? 3 : 2);     // as many waves (<= 4) as have room for their private rings beside the weights
  constexpr size_t lds = slab + NW * per;
  static_assert(lds <= 160 * 1024, "conv_pws: the weight slice and two waves' rings must fit the LDS of a CU");
  {
  static tf::PerDevice attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pws_kernel<T, KS, NF, EPIC, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // persistent grid: as many blocks as fit (LDS decides), at most one wave per tile
  const int cus = tf::device_cus();               // (per device: ADVICE r5)
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  PwK kk = k;
  kk.nsl = A->Cout / N;
  kk.nb = cus * (per_cu > 2 ? 2 : per_cu) / kk.nsl;
  if (kk.nb < 1) kk.nb = 1;
  const int need = (k.ntiles + NW - 1) / NW;
  if (kk.nb > need) kk.nb = need;
  if (kk.nsl > 1 && kk.nb > 8) kk.nb &= ~7;                    // (the XCD mapping of the slices wants a multiple of 8)
  const int blocks = kk.nb * kk.nsl;
  const double M = k.M, es = sizeof(T);
  const double NT = A->Cout;
  const double bytes = (M * k.K + NT * k.K + M * NT * (1 + NAUX)) * es;
  const double alg_k = A->alg_k > 0 ? A->alg_k : k.K, alg_n = A->alg_n > 0 ? A->alg_n : NT;
  tf::ProfScope prof(23, 2.0 * M * alg_n * alg_k, bytes, stream, k.M, A->Cout, k.K, 1, A->mode, A->epi, 2.0 * M * NT * k.K, true);     // 23 = conv_pws
  TF_LAUNCH_TIMED((conv_pws_kernel<T, KS, NF, EPIC, NW>), dim3(blocks), dim3(NW * 64), lds, stream, kk);
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
  }
}

template <typename T, int KS, int NF>
int launch_epi(const tf_conv_args* A, const PwK& k, hipStream_t stream) {
  const bool train = std::is_same<T, tf::bf16_t>::value;
  switch (A->epi) {
    case TF_EPI_AFFINE | TF_EPI_RELU: return launch_one<T, KS, NF, TF_EPI_AFFINE | TF_EPI_RELU>(A, k, stream);
    case TF_EPI_AFFINE: return launch_one<T, KS, NF, TF_EPI_AFFINE>(A, k, stream);
    case TF_EPI_AFFINE | TF_EPI_RES | TF_EPI_RELU: return launch_one<T, KS, NF, TF_EPI_AFFINE | TF_EPI_RES | TF_EPI_RELU>(A, k, stream);
    default: break;
  }
  if constexpr (std::is_same<T, tf::bf16_t>::value) {       // the training sets: bf16 only (fp16 is inference only)
    if (A->epi == TF_EPI_STATS) return launch_one<T, KS, NF, TF_EPI_STATS>(A, k, stream);
    if (A->epi == (TF_EPI_MASK | TF_EPI_STATS2)) return launch_one<T, KS, NF, TF_EPI_MASK | TF_EPI_STATS2>(A, k, stream);
    if (A->epi == 0) return launch_one<T, KS, NF, 0>(A, k, stream);
    if constexpr (KS == 1 || NF == 8) {                    // the hand-over data gradients of conv1 (planes -> 4 planes / planes): layers 1-3
      if (A->epi == (TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3)) return launch_one<T, KS, NF, TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3>(A, k, stream);
      if (A->epi == (TF_EPI_RES | TF_EPI_MASK2)) return launch_one<T, KS, NF, TF_EPI_RES | TF_EPI_MASK2>(A, k, stream);
      if (A->epi == TF_EPI_RES) return launch_one<T, KS, NF, TF_EPI_RES>(A, k, stream);
    }
  }
  (void)train;
  return TF_ERR_UNSUPPORTED;
}

template <typename T>
int launch(const tf_conv_args* A, hipStream_t stream) {
  PwK k;
  k.x = (const char*)A->x; k.w = (const char*)A->w; k.y = (char*)A->y; k.aux = (const char*)A->aux; k.aux2 = (const char*)A->aux2; k.aux3 = (const char*)A->aux3;
  k.epi_scale = A->epi_scale; k.epi_shift = A->epi_shift; k.mask_scale = A->mask_scale; k.mask_shift = A->mask_shift;
  k.stat_out = A->stat_out; k.stat_shift = A->stat_shift; k.stat_shift_out = A->stat_shift_out;
  k.M = A->N * A->OH * A->OW; k.K = A->Cin; k.ntiles = (k.M + 15) / 16; k.srows = tf_get_stat_rows(); k.ldy = A->ldy;
  k.nsl = 1; k.nb = 1;
  const int ks = A->Cin / 64, nf = A->Cout / 16;
  if (nf > 16) {                                                            // sliced: 128 output channels per block (experimental build only)
#if TF_EXP
    if (ks == 4) return launch_epi<T, 4, 8>(A, k, stream);                  // 256 -> 1024: conv3 of layer 3 and the data gradient of its conv1
    if (ks == 2) return launch_epi<T, 2, 8>(A, k, stream);                  // 128 -> 512: the same of layer 2
#endif
    return TF_ERR_UNSUPPORTED;
  }
  if (ks == 1 && nf == 16) return launch_epi<T, 1, 16>(A, k, stream);     // 64 -> 256: conv3 / downsample of layer 1
  if (ks == 4 && nf == 4) return launch_epi<T, 4, 4>(A, k, stream);       // 256 -> 64: conv1 of layer 1, the data gradient of its conv3
  if (ks == 1 && nf == 4) return launch_epi<T, 1, 4>(A, k, stream);       // 64 -> 64: conv1 of layer1.0
  if (ks == 4 && nf == 8) return launch_epi<T, 4, 8>(A, k, stream);       // 256 -> 128: conv1 of layer2.0
  return TF_ERR_UNSUPPORTED;
}

bool epi_ok(const tf_conv_args* a) {
  switch (a->epi) {
    case TF_EPI_AFFINE | TF_EPI_RELU: case TF_EPI_AFFINE: case TF_EPI_AFFINE | TF_EPI_RES | TF_EPI_RELU: return true;
    case TF_EPI_STATS: case TF_EPI_MASK | TF_EPI_STATS2: case 0: return a->dtype == TF_BF16;
    case TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3: case TF_EPI_RES | TF_EPI_MASK2: case TF_EPI_RES:
      return a->dtype == TF_BF16 && (a->Cin == 64 || ((a->Cin == 256 || a->Cin == 128) && a->Cout % 128 == 0));
    default: return false;
  }
}

}  // namespace

// pointwise (1x1, stride 1, pad 0) conv / data gradient with 2-byte operands, (Cin, Cout) in {(64, 256), (256, 64), (64, 64), (256, 128)} or
// (256 | 128, a multiple of 128 above 256: sliced),
// ldy == Cout, one of the epilogue sets above, and enough pixels that the launch is a stream (M >= 16 384)
bool tf_conv_pws_applicable(const tf_conv_args* a) {
  if (a->dtype != TF_BF16 && a->dtype != TF_F16) return false;
  if (a->pro_scale || a->bnf) return false;
  if (a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->H != a->OH || a->W != a->OW) return false;
  if (a->ldy != a->Cout || !epi_ok(a)) return false;
  // statistic sums are folded into <= TF_STAT_ROWS rows by atomics: the bit-reproducible flow (tf_set_stat_rows(0): one row per tile) keeps conv_dma
  if ((a->epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) && tf_get_stat_rows() > TF_STAT_ROWS) return false;
  const int ks = a->Cin / 64, nf = a->Cout / 16;
  if (a->Cin % 64 || a->Cout % 16) return false;
  const long M = (long)a->N * a->OH * a->OW;
  if (nf > 16) {
    // sliced: 256 -> 512 / 1024, 128 -> 512 in slices of 128 output channels.  Built, parity-green (tile = 70 on request), and SLOWER than the
    // tiled kernel on every layer-2/3 shape (alone: 39.8 vs 24.9 us on the layer-3 hand-over gradient; in the step: -5 %, profiles/r05_conv_pws.txt):
    // two waves per CU hold one 24 KiB tile in flight each.  The dispatcher takes it only with TINYFACES_PWS_SLICED=1.
    const bool sliced = tf::tuning().pws_sliced;
    if (!TF_EXP || !(ks == 4 || ks == 2) || a->Cout % 128 != 0 || M < 8192) return false;
    return sliced || a->tile == 70;
  }
  if (!((ks == 1 && nf == 16) || (ks == 4 && nf == 4) || (ks == 1 && nf == 4) || (ks == 4 && nf == 8))) return false;
  return M >= 16384;
}
int tf_conv_pws_launch(const tf_conv_args* a, hipStream_t stream) {
  if (!tf_conv_pws_applicable(a)) return TF_ERR_UNSUPPORTED;
  return a->dtype == TF_BF16 ? launch<tf::bf16_t>(a, stream) : launch<tf::f16_t>(a, stream);
}
