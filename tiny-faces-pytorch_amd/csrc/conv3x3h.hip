// conv3x3h: the 3x3 / stride 1 / pad 1 convolution (forward AND data gradient) with a halo-resident input tile (gfx950).
//
// Why a second conv kernel.  The generic LDS-DMA kernel (conv_dma.hip) treats a 3x3 conv as a GEMM over an im2col view: every
// K-stage (one tap x 64 channels) DMAs a fresh [pixels][64] slab of the input, so each input byte crosses the L2 -> L1 -> LDS
// path nine times: on the layer-3 shape (12 x 32 x 32 pixels, 256 -> 256 channels) 442 MB of LDS-DMA per launch at 64 x 64
// tiles, and a wave gets 8 MFMAs per barrier behind the address arithmetic of a generic gather (28-30 us per launch whatever
// the tile: profiles/r01d_conv_dma_pipeline_ablation.txt, profiles/r02a_microbench_mma32_tiles.txt).
// Here the block's input patch is loaded ONCE per 64-channel chunk, with its one-pixel halo:
//   * output tile = 4 rows x 32 columns of one image (128 pixels) x 128 output channels;
//   * input frame = 6 x 34 pixels x 64 channels = 204 rows of 128 bytes in LDS; a tap is a CONSTANT ROW SHIFT of the fragment
//     reads inside that frame ((kh-1)*34 + (kw-1)), so the nine taps of a chunk need no new input traffic at all;
//   * K order = chunk-major, tap-minor: stage st = chunk*9 + tap streams only the 128 x 64 weight slab of (tap, chunk)
//     (16 KiB, 3-deep ring); the next chunk's frame (26 KiB, 2-deep ring) is requested at the first tap of the current one.
//   LDS-DMA bytes per block and stage: 16 KiB + 26/9 KiB = 19 KiB instead of 32 KiB (128 x 128 im2col) or 2 x 16 KiB (64 x 64).
// Eight waves: 2 (pixels) x 2 (channels) x 2 (K halves).  Each wave owns a 64 x 64 accumulator block on 32x32x16 fragments
// (16 MACs per LDS byte read) and multiplies HALF of every 64-deep stage (32 of the 64 channels); the two K halves are added
// through two fp32 staging tiles in the epilogue.  Two waves per SIMD with the same MFMA work as the four-wave form: one
// wave's DMA issue / LDS latency / barrier wait is covered by its partner's MFMAs (the four-wave 128 x 128 form had nothing
// to cover them: profiles/r02a_microbench_mma32_tiles.txt).
// The K loop is written for instruction count (see below): 23 us per launch = 628 TFLOP/s on the layer-3 shape
// (profiles/r02_conv3x3h_trace.txt: stage stamps, ablations, the versions that did not work).
// r6 (profiles/r06_conv3x3h.txt): a K stage takes 880 cycles where its 16 MFMAs per SIMD need 512 (MFMAs alone 535, + fragment reads 600-700,
// + barrier and counted wait ~100, + the DMA ~190): the parts of a stage add up, they do not overlap.  Six forms were built to parity and
// measured against this one -- K half 1 multiplying behind the barrier while K half 0 reads ("ping-pong"), a ring of 4 / 5 slots, dedicated
// loader waves (8 MFMA waves that never touch vector memory + 4 loaders, with and without a deeper ring), the DMA instructions spread between
// the MFMAs, a tile of 8 rows x 64 channels (12.8 instead of 19.4 KiB of operands per stage) -- and every one is equal or slower.  A probe
// without any barrier (csrc/probe.hip kind 12) shows the law underneath: vector-memory instructions of OTHER waves make no progress while a
// CU's matrix pipes are saturated, and every global_load_lds a wave issues between its own MFMAs costs its SIMD about one MFMA slot.
// Epilogue = conv_dma's (fp32 tile in LDS, 16 output bytes per thread, every TF_EPI_* flag), for 512 threads.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "tuning.h"
#include "debug_api.h"
#include "profile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ uint4 g_zero_rows[8];     // 128 zero bytes: source of halo pixels outside the image

constexpr int TR = 4, TC = 32, FW = TC + 2, FR = (TR + 2) * FW;      // output tile rows x cols, frame width, frame rows (204)
constexpr int BM = TR * TC, BN = 128, NT = 512;
constexpr int XPASS = 4, XBUF = XPASS * 64 * 128;                     // frame buffer: 256 rows of 128 B (rows >= FR unused)
constexpr int WPASS = 2, WBUF = BN * 128;                             // weight stage: 128 channels x 64 k
constexpr int W_AT = 2 * XBUF;
constexpr int NSW = 3;                                               // 3-deep weight ring (9 taps = 3 turns)
constexpr int PITCH = BN + 4, STG_FLOATS = (BM / 2) * PITCH;          // r3: the epilogue stages HALF the tile (64 pixels) at a time
constexpr int RING_BYTES = W_AT + NSW * WBUF;                          // 112 KiB
// r3: 112 KiB instead of 132.  LDS capacity is what the two streams of the backward pass fight over (profiles/r03_contention.txt):
// with the two K-half staging tiles of the WHOLE 128-pixel tile (2 x 66 KiB) a block left 28 KiB of its CU, so not even one 48 KiB
// weight-gradient block could sit beside it and the two kernels time-sliced the CU; two passes of 64 pixels (2 x 33 KiB, overlaying
// the rings) cost two more barriers per block.
constexpr int LDS_BYTES = 2 * STG_FLOATS * 4 > RING_BYTES ? 2 * STG_FLOATS * 4 : RING_BYTES;
// r6: the tile HEIGHT is a template parameter of the kernel (the constants above are those of TR = 4, the training tile).  A launch whose
// 4-row tiles need a second, nearly empty round of blocks -- 300 blocks on 256 CUs at the 1920 x 2560 pyramid level (M = 19 200) -- takes
// 6-row tiles instead: 200 blocks of 1.5x the work in ONE round (launch(): the height with the smaller rounds x rows product).
template <int TR_> struct Geo {
  static constexpr int TR = TR_, FR = (TR_ + 2) * FW, BM = TR_ * TC;
  static constexpr int XPASS = (FR + 63) / 64, XBUF = XPASS * 64 * 128, W_AT = 2 * XBUF;
  static constexpr int STG_FLOATS = (BM / 2) * PITCH, RING_BYTES = W_AT + NSW * WBUF;
  static constexpr int LDS_BYTES = 2 * STG_FLOATS * 4 > RING_BYTES ? 2 * STG_FLOATS * 4 : RING_BYTES;
};
static_assert(Geo<4>::XPASS == XPASS && Geo<4>::LDS_BYTES == LDS_BYTES && Geo<6>::LDS_BYTES <= 160 * 1024, "conv3x3h geometry");

struct HK {
  const char* x; const char* w; char* y;
  const float* epi_scale; const float* epi_shift;
  const char* aux; const char* aux2; const char* aux3;
  const float* mask_scale; const float* mask_shift;
  float* stat_out; const float* stat_shift; float* stat_shift_out;
  int H, W, C, ldy, Ktot, cpt, nst, ntiles, epi, srows, mtiles, rtiles, ctiles, sign, dbg;
  unsigned long long* trace;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) << 2) ^ (((row >> 3) & 1) * 6); }

template <typename T> struct Frag;
template <> struct Frag<tf::bf16_t> {
  typedef bf16x8 t;
  __device__ static __forceinline__ f32x16 mma(t a, t b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Frag<tf::f16_t> {
  typedef f16x8 t;
  __device__ static __forceinline__ f32x16 mma(t a, t b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// r5: the stage wait ALSO drains the wave's LDS reads (lgkmcnt(0)).  hipcc sinks the last MFMAs of a stage -- and the wait for the two
// fragment reads that feed them -- below the NEXT stage's wait + barrier (ISA of every instantiation): a wave then passed the barrier with
// reads of ring slot s % 3 / the old frame still in flight, and behind that barrier another wave issues the DMA that REFILLS exactly that
// slot.  The reads normally return long before the DMA lands, but with three pyramid levels side by side on one GPU they sometimes
// did not: c2 of a layer-3 bottleneck came out a few values different in ~1 of 4 evaluation pyramids (scripts/diag_arena_diff.py walks the
// arena to the first tensor that differs; tests/test_gpu_model.py::test_eval_pyramid_is_bit_reproducible).  With the reads drained before
// the barrier the refill is ordered behind every read of the slot it overwrites.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// TRACE: debugging instantiation (tf_debug_conv3x3h_trace): every wave of the first 8 blocks keeps s_memtime stamps of the five
// points of each K stage in LDS and dumps them after the loop, and the TINYFACES_CONV3H_DBG ablation bits (1: no steady-state
// DMA, 2: no LDS reads / MFMAs) are honoured; never launched unless a trace buffer was registered.
constexpr int TRACE_BYTES = 8 * 64 * 8 * 8;
// EPIC (r4): the epilogue flag set as a compile-time constant (-1: read a.epi), like conv_dma_kernel's: the three sets a training step and
// an evaluation forward use get their own instantiation, the dead modes of the generic epilogue fold away.
template <typename T, bool TRACE, int EPIC = -1, int TR_ = 4>
__global__ void __launch_bounds__(NT, 2) conv3x3h_kernel(const HK a) {
  const int epi_flags = EPIC >= 0 ? EPIC : a.epi;
  typedef typename Frag<T>::t frag;
  typedef Geo<TR_> G;                                  // (the names below shadow the TR = 4 constants of the file scope)
  constexpr int TR = G::TR, FR = G::FR, XPASS = G::XPASS, XBUF = G::XBUF, W_AT = G::W_AT, STG_FLOATS = G::STG_FLOATS, RING_BYTES = G::RING_BYTES;
  constexpr int MH = TR / 2;                           // tile rows (= 32-pixel fragments) per pixel half
  constexpr int EPS = 8, TRACE_AT = RING_BYTES;       // (the stamps are dumped before the epilogue overlays them)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long t_start = 0, r_start = 0;
  if constexpr (TRACE) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
  auto stamp = [&](int st, int k) {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) reinterpret_cast<unsigned long long*>(smem + TRACE_AT)[((threadIdx.x >> 6) * 64 + st) * 8 + k] = t;
    }
  };

  int logical;
  {
    const int nb = gridDim.x, b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
  const int per_img = a.rtiles * a.ctiles;
  const int img = mt / per_img, trem = mt - img * per_img, rb = trem / a.ctiles, cb = trem - rb * a.ctiles;
  const int r0 = rb * TR, c0 = cb * TC, n0 = nt * BN;
  const int tid = threadIdx.x, lrow = tid >> 3, pslot = tid & 7, wave = tid >> 6;

  // ---- DMA roles (fixed for the whole K loop).  Physical 16-byte slot pslot of LDS row f receives logical slot pslot ^ swz(f).
  // The K loop is ISSUE-bound if written naively (a first version with run-time tap / ring iterators spent ~210 instructions and
  // 13 branches per stage for 8 MFMAs per wave: 1000 cycles per stage, whatever the pipelining -- profiles/r02_conv3x3h_trace.txt),
  // so everything a stage needs is either a compile-time function of the tap (the nine taps of a chunk are unrolled; 9 is a
  // multiple of the 3-deep weight ring, so the ring slot is tap % 3) or precomputed here.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const char* zero = reinterpret_cast<const char*>(g_zero_rows) + pslot * 16;
  const char* xptr[XPASS]; int xstep[XPASS];       // frame row f = lrow + 64 i of the NEXT chunk to request, advanced per chunk
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int f = lrow + i * 64;
    const int fr = f / FW, fc = f - fr * FW;
    const int ih = r0 - 1 + fr, iw = c0 - 1 + fc;
    const bool ok = f < FR && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    xptr[i] = ok ? a.x + (((size_t)img * a.H + ih) * a.W + iw) * a.C * sizeof(T) + ((pslot ^ swz(f)) << 4) : zero;
    xstep[i] = ok ? 128 : 0;
  }
  const char* wptr[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int row = lrow + i * 64;
    wptr[i] = a.w + (size_t)(n0 + row) * a.Ktot * sizeof(T) + ((pslot ^ swz(row)) << 4);
  }
  int x_buf = 0;                                     // frame buffer the next issue_x fills
  auto issue_x = [&]() {
    char* dst = smem + x_buf * XBUF + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) { dma16(xptr[i], dst + i * 8192); xptr[i] += xstep[i]; }
    x_buf ^= 1;
  };
  auto issue_w = [&](auto SLOT, int kofs_bytes) {    // weight stage (tap, chunk) at K offset tap*C + chunk*64 -> ring slot SLOT
    char* dst = smem + W_AT + decltype(SLOT)::value * WBUF + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < WPASS; ++i) dma16(wptr[i] + kofs_bytes, dst + i * 8192);
  };
  // request the first frame and the first two weight stages NOW: they travel while the fragment offsets below are computed
  const int cpt = a.cpt, tapC = a.C * (int)sizeof(T);      // bytes between two taps of one weight row
  issue_x();
  issue_w(std::integral_constant<int, 0>{}, 0);
  issue_w(std::integral_constant<int, 1>{}, tapC);

  // ---- MFMA roles: wave = (K half kg, pixel half wm, channel half wn); 2 x MH fragments of 32 channels x 32 pixels (MH = 2 tile rows at TR = 4)
  const int kg = wave >> 2, wm = wave & 1, wn = (wave >> 1) & 1;
  const int l = tid & 63, r32 = l & 31, h = l >> 5;
  // fragment byte offsets, all of them: x fragment m of tap t at k-step j lives at frame row fbase(m) + shift(t) (a tap is a constant
  // row shift inside the frame; the swizzle follows the shifted row), in frame buffer 0 -- + XBUF selects buffer 1
  int xo[9][MH][2], woff[2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int shift = a.sign * ((t / 3 - 1) * FW + (t % 3 - 1));
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      const int f = (wm * MH + m + 1) * FW + r32 + 1 + shift, sz = swz(f);      // output pixel (tile row wm*MH+m, column r32)
#pragma unroll
      for (int j = 0; j < 2; ++j) xo[t][m][j] = f * 128 + ((((kg * 2 + j) * 2 + h) ^ sz) << 4);
    }
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int row = wn * 64 + n * 32 + r32;
#pragma unroll
    for (int j = 0; j < 2; ++j) woff[n][j] = W_AT + row * 128 + ((((kg * 2 + j) * 2 + h) ^ swz(row)) << 4);
  }
  f32x16 acc[2][MH];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[n][m] = f32x16(0.f);

  int st = 0;
  for (int chunk = 0; chunk < cpt; ++chunk) {
    const bool more = chunk + 1 < cpt;              // another chunk follows
    const int kchunk = chunk * 128;                 // byte offset of this chunk inside a tap
    auto stage = [&](auto TAP) {
      constexpr int tap = decltype(TAP)::value;
      stamp(st, 0);
      // weight stage st has landed once only what was issued after it is still in flight: stage st+1 (WPASS), plus the next
      // chunk's frame (XPASS) when the previous stage was a first tap (the frame is issued BEFORE that stage's weights, so every
      // earlier frame / stage has retired by then: loads retire in order); the very last stage has nothing younger
      if (tap == 8) { if (more) wait_vmcnt<WPASS>(); else wait_vmcnt<0>(); }
      else if (tap == 1) { if (more) wait_vmcnt<WPASS + XPASS>(); else wait_vmcnt<WPASS>(); }
      else wait_vmcnt<WPASS>();
      stamp(st, 1);
      __builtin_amdgcn_s_barrier();                 // everyone's pieces landed; everyone finished reading stage st-1's buffers
      stamp(st, 2);
      auto issue = [&]() {
        if (!TRACE || !(a.dbg & 1)) {
          if (tap == 0 && more) issue_x();
          // stage st+2 = tap+2 of this chunk, or tap+2-9 of the next one
          if (tap < 7) issue_w(std::integral_constant<int, (tap + 2) % 3>{}, (tap + 2) * tapC + kchunk);
          else if (more) issue_w(std::integral_constant<int, (tap + 2) % 3>{}, (tap + 2 - 9) * tapC + kchunk + 128);
        }
      };
      issue();               // (issuing after the first MFMA group instead was measured: no difference, profiles/r02_conv3x3h_trace.txt)
      stamp(st, 3);
      if (!TRACE || !(a.dbg & 2)) {
        const char* wb = smem + (tap % 3) * WBUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          frag xf[MH], wf[2];
#pragma unroll
          for (int m = 0; m < MH; ++m) xf[m] = *reinterpret_cast<const frag*>(smem + xo[tap][m][j]);
#pragma unroll
          for (int n = 0; n < 2; ++n) wf[n] = *reinterpret_cast<const frag*>(wb + woff[n][j]);
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < MH; ++m) acc[n][m] = Frag<T>::mma(wf[n], xf[m], acc[n][m]);
        }
      }
      stamp(st, 4);
      ++st;
    };
    stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
    stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});
    // the next chunk's frame sits in the other buffer (r6: by addition -- the 40 KiB buffer of the 6-row tile is not a power of two, the XOR of
    // rounds 2-5 only toggled the 32 KiB one)
    const int other = (chunk & 1) ? -XBUF : XBUF;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) xo[t][m][j] += other;
  }
  // (r5, measured and removed: TWO taps per barrier -- the 18 taps of two channel chunks unrolled into 9 two-tap stages, a 2-slot ring of
  //  two-tap weight slabs (128 KiB), the next frame requested behind the weights so that one counted wait leaves it in flight; parity-green,
  //  208 VGPRs: 1296.9 / 1292.2 against 1297.3 / 1292.5 img/s, evaluation 5.16 against 4.93 ms per image.  Halving the barriers and counted
  //  waits of the K loop buys nothing: DESIGN.md 7 row 51.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (r5: no fragment read in flight when the rings become the staging tile)
  __builtin_amdgcn_s_barrier();                     // all waves done reading the rings -> reuse them as the staging tile
  if constexpr (TRACE) {
    if (logical < 8 && a.trace) {
      const unsigned long long t_loop = __builtin_amdgcn_s_memtime(), r_loop = __builtin_amdgcn_s_memrealtime();
      __syncthreads();
      unsigned long long* out = a.trace + (size_t)logical * (8 * 64 * 8);
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + TRACE_AT);
      for (int e = tid; e < 8 * 64 * 8; e += NT) out[e] = src[e];
      __syncthreads();
      if ((tid & 63) == 0) {                          // stage slot 63 of each wave: start / end of the K loop on both clocks
        unsigned long long* q = out + (wave * 64 + 63) * 8;
        q[0] = t_start; q[1] = r_start; q[2] = t_loop; q[3] = r_loop; q[4] = blockIdx.x;
      }
    }
  }

  unsigned long long e_t[5] = {0, 0, 0, 0, 0};
  if constexpr (TRACE) e_t[0] = __builtin_amdgcn_s_memtime();
  // ---------------- epilogue, phase 1: one fp32 [BM][BN+4] tile in LDS per K half (both halves write at once; phase 2 adds them)
  // 32x32 accumulator: lane l holds pixel l & 31, channels 8*g + 4*(l >> 5) + {0..3} for g = 0..3 (registers 4g .. 4g+3)
  float* stg = reinterpret_cast<float*>(smem);
  auto park = [&](int half) {                        // the waves that own pixel half `half` (wm) write their accumulators
    if (wm != half) return;
    float* mine = stg + kg * STG_FLOATS;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(mine + (m * 32 + r32) * PITCH + wn * 64 + n * 32 + g * 8 + h * 4) =
              f32x4{acc[n][m][4 * g], acc[n][m][4 * g + 1], acc[n][m][4 * g + 2], acc[n][m][4 * g + 3]};
  };
  park(0);
  __syncthreads();
  if constexpr (TRACE) e_t[1] = __builtin_amdgcn_s_memtime();

  // ---------------- phase 2: one 16-byte output chunk (8 consecutive channels) of one pixel per thread and pass; a pass of
  // 512 threads = one row of the output tile (32 pixels x 16 chunks)
  constexpr int CPR = BN / EPS;                      // 16
  const int chunk8 = tid % CPR, tc = tid / CPR;
  const int ch0 = n0 + chunk8 * EPS;
  const bool cok = ch0 < a.ldy;
  float es[EPS], eh[EPS], ms[EPS], mh[EPS], s1[EPS], s2[EPS], sft[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { es[j] = 1.f; eh[j] = 0.f; ms[j] = 0.f; mh[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; sft[j] = 0.f; }
  if (cok) {
    if ((epi_flags & TF_EPI_STATS) && a.stat_shift) {    // sums of (x - shift), (x - shift)^2: see tf_conv_args.stat_shift
#pragma unroll
      for (int j = 0; j < EPS; ++j) sft[j] = a.stat_shift[ch0 + j];
    }
    if (epi_flags & TF_EPI_AFFINE) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { es[j] = a.epi_scale[ch0 + j]; eh[j] = a.epi_shift[ch0 + j]; }
    }
    if (epi_flags & TF_EPI_MASK) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { ms[j] = a.mask_scale[ch0 + j]; mh[j] = a.mask_shift[ch0 + j]; }
    }
  }
#pragma unroll
  for (int ps = 0; ps < TR; ++ps) {
    if (ps == TR / 2) {                              // second pixel half: rows 2, 3 of the tile replace rows 0, 1 in the staging tiles
      __syncthreads();
      park(1);
      __syncthreads();
    }
    const int oh = r0 + ps, ow = c0 + tc;
    if (!(oh < a.H && ow < a.W && cok)) continue;   // tile pixels outside the image hold meaningless sums: no store, no statistics
    const int row = (ps % (TR / 2)) * TC + tc;
    float v[EPS];
#pragma unroll
    for (int j = 0; j < EPS; j += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(stg + row * PITCH + chunk8 * EPS + j) +
                      *reinterpret_cast<const f32x4*>(stg + STG_FLOATS + row * PITCH + chunk8 * EPS + j);
      v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
    }
    if (epi_flags & TF_EPI_STATS) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { const float t = v[j] - sft[j]; s1[j] += t; s2[j] += t * t; }
    }
    const size_t o = ((((size_t)img * a.H + oh) * a.W + ow) * a.ldy + ch0) * sizeof(T);
    float ax[EPS];
    if (epi_flags & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux + o), ax);
    if (epi_flags & TF_EPI_AFFINE) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) v[j] = v[j] * es[j] + eh[j];
    }
    if (epi_flags & TF_EPI_RES) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) v[j] += ax[j];
    }
    if (epi_flags & TF_EPI_MASK) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) v[j] = (ax[j] * ms[j] + mh[j] > 0.f) ? v[j] : 0.f;
    }
    if (epi_flags & TF_EPI_JOIN) {
      float y2[EPS], g3[EPS];
      tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
      tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), g3);
#pragma unroll
      for (int j = 0; j < EPS; ++j) v[j] += (y2[j] > 0.f) ? g3[j] : 0.f;
    }
    if (epi_flags & TF_EPI_MASK2) {
      float y2[EPS];
      tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
#pragma unroll
      for (int j = 0; j < EPS; ++j) v[j] = (y2[j] > 0.f) ? v[j] : 0.f;
    }
    if (epi_flags & TF_EPI_RELU) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (epi_flags & TF_EPI_STATS2) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * ax[j]; }
    }
    if (epi_flags & TF_EPI_STATS3) {
      float x3[EPS];
      tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), x3);
#pragma unroll
      for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * x3[j]; }
    }
    *reinterpret_cast<uint4*>(a.y + o) = tf::pack16<T>(v);
  }
  if constexpr (TRACE) e_t[2] = __builtin_amdgcn_s_memtime();
  if (epi_flags & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) {       // block-uniform: column sums of the tile
#pragma unroll
    for (int j = 0; j < EPS; ++j) { s1[j] = tf::lane_group_sum<CPR>(s1[j]); s2[j] = tf::lane_group_sum<CPR>(s2[j]); }       // DPP / row swaps, no LDS crossbar
    __syncthreads();                                 // staging tile fully consumed
    float* red = reinterpret_cast<float*>(smem);     // [8 waves][2][BN]
    const int lane = tid & 63;
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { red[(wave * 2 + 0) * BN + lane * EPS + j] = s1[j]; red[(wave * 2 + 1) * BN + lane * EPS + j] = s2[j]; }
    }
    __syncthreads();
    if ((epi_flags & TF_EPI_STATS) && a.stat_shift && a.stat_shift_out && mt == 0) {
      for (int cl = tid; cl < BN; cl += NT)
        if (n0 + cl < a.ldy) a.stat_shift_out[n0 + cl] = a.stat_shift[n0 + cl];
    }
    for (int e = tid; e < 2 * BN; e += NT) {
      const int k = e / BN, cl = e - k * BN, c = n0 + cl;
      if (c < a.ldy) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv) v += red[(wv * 2 + k) * BN + cl];
        // same folding of partial rows as conv_dma: tile mt accumulates into row mt % TF_STAT_ROWS when there are more tiles than rows
        if (a.mtiles > a.srows) atomicAdd(&a.stat_out[((size_t)(mt % a.srows) * 2 + k) * a.ldy + c], v);
        else a.stat_out[((size_t)mt * 2 + k) * a.ldy + c] = v;
      }
    }
  }
  if constexpr (TRACE) {
    if (logical < 8 && a.trace && (tid & 63) == 0) {  // stage slot 62 of each wave: epilogue stamps (start, tile staged, stores issued, statistics done) + all stores retired
      e_t[3] = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      e_t[4] = __builtin_amdgcn_s_memtime();
      unsigned long long* q = a.trace + (size_t)logical * (8 * 64 * 8) + (wave * 64 + 62) * 8;
      for (int i = 0; i < 5; ++i) q[i] = e_t[i];
    }
  }
}

unsigned long long* g_trace = nullptr;
int g_last_tile_rows = 0;          // tile height of the most recent launch (tf_debug_conv3x3h_tile_rows: which instantiation a test just ran)
int dbg_flags() { return tf::tuning().conv3h_dbg; }
int min_blocks() {
  return tf::tuning().conv3h_minblocks;
}

template <typename T, bool TRACE, int EPIC = -1, int TR_ = 4>
void launch_var(const HK& k, hipStream_t stream) {
  typedef Geo<TR_> G;
  static tf::PerDevice attr_set;                     // per DEVICE: the attribute belongs to the current device's copy of the function
  if (attr_set.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3h_kernel<T, TRACE, EPIC, TR_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  TF_LAUNCH_TIMED((conv3x3h_kernel<T, TRACE, EPIC, TR_>), dim3(k.mtiles * k.ntiles), dim3(NT),
                  TRACE && G::RING_BYTES + TRACE_BYTES > G::LDS_BYTES ? G::RING_BYTES + TRACE_BYTES : G::LDS_BYTES, stream, k);
}

template <typename T>
int launch(const tf_conv_args* A, hipStream_t stream) {
  HK k;
  k.x = (const char*)A->x; k.w = (const char*)A->w; k.y = (char*)A->y;
  k.epi_scale = A->epi_scale; k.epi_shift = A->epi_shift;
  k.aux = (const char*)A->aux; k.aux2 = (const char*)A->aux2; k.aux3 = (const char*)A->aux3;
  k.mask_scale = A->mask_scale; k.mask_shift = A->mask_shift; k.stat_out = A->stat_out;
  k.stat_shift = A->stat_shift; k.stat_shift_out = A->stat_shift_out;
  k.H = A->OH; k.W = A->OW; k.C = A->Cin; k.ldy = A->ldy; k.Ktot = 9 * A->Cin; k.cpt = A->Cin / 64; k.nst = 9 * k.cpt;
  k.ntiles = A->Cout / BN; k.epi = A->epi; k.srows = tf_get_stat_rows();
  // tile height (r6): evaluation launches (folded BN + ReLU) whose 4-row tiles leave a mostly empty last round of blocks take 6-row tiles when
  // that makes rounds x rows smaller -- 1 x 120 x 160 (the 1920 x 2560 level): 300 blocks = 2 rounds x 4 rows against 200 blocks = 1 round x 6 rows.
  // The statistic epilogues of the training step keep TR = 4 (tf_conv3x3h_mtiles sizes their partial rows).  TINYFACES_CONV3H_TR6=0: never.
  k.ctiles = (A->OW + TC - 1) / TC;
  bool tall = false;
  if (A->epi == (TF_EPI_AFFINE | TF_EPI_RELU) && !g_trace && !tf::tuning().epi_spec_off && tf::tuning().conv3h_tr6) {
    const long cus = tf::device_cus();
    auto cost = [&](int tr) { const long blocks = (long)A->N * ((A->OH + tr - 1) / tr) * k.ctiles * k.ntiles; return ((blocks + cus - 1) / cus) * tr; };
    tall = cost(6) < cost(4);
  }
  const int tr = tall ? 6 : TR;
  g_last_tile_rows = tr;
  k.rtiles = (A->OH + tr - 1) / tr; k.mtiles = A->N * k.rtiles * k.ctiles;
  k.sign = A->mode == 0 ? 1 : -1;                    // forward reads pixel + (kh-1, kw-1); the data gradient reads pixel - (kh-1, kw-1)
  k.dbg = dbg_flags(); k.trace = g_trace;
  const double es = sizeof(T), M = (double)A->N * A->OH * A->OW, Kt = k.Ktot;
  double bytes = (M * A->Cin + (double)A->Cout * Kt + M * A->Cout) * es;
  if (A->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) bytes += M * A->Cout * es;
  if (A->epi & TF_EPI_JOIN) bytes += 2 * M * A->Cout * es;
  if (A->epi & TF_EPI_MASK2) bytes += M * A->Cout * es;
  if (A->epi & TF_EPI_STATS3) bytes += M * A->Cout * es;
  tf::ProfScope prof(A->dtype == TF_BF16 ? 6 : 7, 2.0 * M * A->Cout * Kt, bytes, stream, (int)M, A->Cout, k.Ktot, 9, A->mode, A->epi, -1.0, true);   // 6 = conv3x3h bf16, 7 = f16
  const bool spec_off = tf::tuning().epi_spec_off;       // A/B knob (shared with conv_dma)
  if (g_trace) launch_var<T, true>(k, stream);
  else if (spec_off) launch_var<T, false>(k, stream);
  else if (A->epi == TF_EPI_STATS) launch_var<T, false, TF_EPI_STATS>(k, stream);                                           // training forward
  else if (A->epi == (TF_EPI_MASK | TF_EPI_STATS2)) launch_var<T, false, TF_EPI_MASK | TF_EPI_STATS2>(k, stream);           // training data gradient
  else if (A->epi == (TF_EPI_AFFINE | TF_EPI_RELU) && tall) launch_var<T, false, TF_EPI_AFFINE | TF_EPI_RELU, 6>(k, stream);
  else if (A->epi == (TF_EPI_AFFINE | TF_EPI_RELU)) launch_var<T, false, TF_EPI_AFFINE | TF_EPI_RELU>(k, stream);           // evaluation (folded BN + ReLU)
  else launch_var<T, false>(k, stream);
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

}  // namespace

// 3x3, stride 1, pad 1, 2-byte operands, 64-multiples of input channels, 128-multiples of output channels (layers 2 and 3 of
// the trunk, forward and data gradient); `forced` skips the "enough blocks to fill the chip" test (tile code 50 from a caller)
bool tf_conv3x3h_applicable(const tf_conv_args* a, bool forced) {
  if (a->dtype != TF_BF16 && a->dtype != TF_F16) return false;
  if (a->KH != 3 || a->KW != 3 || a->stride != 1 || a->pad != 1 || a->H != a->OH || a->W != a->OW) return false;
  if (a->Cin % 64 != 0 || a->Cout % BN != 0 || a->pro_scale) return false;
  if (forced) return true;
  const bool off = tf::tuning().conv3h_off;
  if (off) return false;
  // >= 4 channel chunks (36 stages): with the 18 stages of layer 2 (128 channels) the prologue / epilogue weigh too much and the
  // 64 x 128 im2col tile wins (A/B on one box: 1116 / 1114 img/s without layer 2, 1107 / 1111 with)
  const int min_cin = tf::tuning().conv3h_mincin;
  if (a->Cin < min_cin) return false;
  const long blocks = (long)a->N * ((a->OH + TR - 1) / TR) * ((a->OW + TC - 1) / TC) * (a->Cout / BN);
  return blocks >= min_blocks();
}
// debugging: register (or clear, nullptr) a device buffer of 8 blocks x 8 waves x 64 stages x 8 u64 for the stage stamps of the
// NEXT launches (scripts/trace_conv3x3h.py); not part of the product path
extern "C" int tf_debug_conv3x3h_trace(void* device_buf) { g_trace = (unsigned long long*)device_buf; return TF_OK; }
extern "C" int tf_debug_conv3x3h_tile_rows(void) { return g_last_tile_rows; }
int tf_conv3x3h_mtiles(const tf_conv_args* a) { return a->N * ((a->OH + TR - 1) / TR) * ((a->OW + TC - 1) / TC); }
int tf_conv3x3h_launch(const tf_conv_args* a, hipStream_t stream) {
  return a->dtype == TF_BF16 ? launch<tf::bf16_t>(a, stream) : launch<tf::f16_t>(a, stream);
}
