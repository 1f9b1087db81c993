// wgrad_group: the weight gradients of a GROUP of pointwise convolutions in ONE launch, every output tile reduced over ALL pixels
// inside its block (gfx950).
//
//   dW_g[co][ci] = sum over all M pixels of dY_g[p][co] * X_g[p][ci]          g = 0 .. nprob-1     (bf16 operands, fp32 accumulate)
//
// Why it exists (profiles/r03_layer_table.md): one pointwise weight gradient of layer 3 is a GEMM with a 256 x 1024
// output and a 12 288-long reduction -- 16 output tiles of 128 x 128 for 256 CUs.  wgrad_dma_kernel therefore splits the reduction 16 ways
// and adds the slices with fp32 atomics: 12 pixel stages per block, i.e. launch / prologue / epilogue as long as the loop, and 64 x 64
// tiles whose operands cross the LDS twice per MAC (0.066 of the bf16 MFMA peak in situ).  The 22 identity bottlenecks of layer 3 have
// IDENTICAL shapes, and nothing but the optimizer reads a weight gradient: the executor keeps their dY / X tensors alive (csrc/detnet.hip,
// 0.9 GB of 288) and differentiates a whole group of bottlenecks in one launch -- 32 tiles per bottleneck, 256 per group of eight.
//   * no split-K: a block walks all 384 pixel stages of its tile; no atomics, no memset, no partial-tile round trip, no reduce kernel:
//     the epilogue is 64 plain stores per lane, once per 0.4 GFLOP;
//   * 128 x 128 tiles, 4 waves, 64 x 64 per wave on mfma_f32_16x16x32_bf16: 16 MFMAs per 16 transposing fragment reads
//     (ds_read_b64_tr_b16: the reduction index -- the pixel -- is the strided one of both NHWC operands) instead of 4 per 8;
//   * operands HBM/L2 -> LDS by global_load_lds_dwordx4 (no register hop), 32-pixel stages of 16 KiB, GP_NS-deep ring (default 4:
//     64 KiB per block), counted vmcnt + raw s_barrier; the fragments of stage st+1 are read while the MFMAs of stage st run
//     (two named fragment sets: a one-block-per-CU launch has ONE wave per SIMD and nobody else hides its LDS latency);
//   * the tiles of one problem are consecutive in the XCD-remapped block order, so the 16 blocks that stream the same dY / X rows
//     share one XCD's L2 (the operands of a group -- 0.6 GB -- do not fit the 256 MB Infinity Cache: they come from HBM once).
// LDS image of an operand sub-tile: [32 pixels][8 x 16 B] (64 channels), 16-byte slots XOR-swizzled on the SOURCE address with
// ((row >> 1) & 3) << 1 -- the layout wgrad_dma.hip found conflict-free for the transposing reads.
// Replaces the autograd weight gradient of the conv1 / conv3 of every identity Bottleneck of layer 3 (tinyfaces/trainer.py:86 through
// torchvision's Bottleneck); bf16 only (the fp32 parity path keeps wgrad.hip).
#include <cstdlib>
#include "common.h"
#include "tuning.h"
#include "profile.h"
#include "lds_dma.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ uint4 g_gzero_page[8];

constexpr int GP_PK = 32;                    // pixels per stage = one MFMA k-step
constexpr int GP_SUB = GP_PK * 128;          // one 64-channel operand sub-tile of a stage: 4 KiB
constexpr int GP_STAGE = 4 * GP_SUB;         // dY[co 0-63], dY[co 64-127], X[ci 0-63], X[ci 64-127]: 16 KiB
constexpr int GP_L = 4;                      // DMA instructions per thread and stage
constexpr int GP_MAX = 48;                   // problems per launch (the table travels as kernel arguments)

struct GpProb { const char* x; const char* dy; float* dw; int Cin, Cout, ldx, lddy, dw_ld, nci, tile0, pad_; };
struct GpK { int M, nprob, ntiles, pad_; GpProb p[GP_MAX]; };

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int fsw(int row) { return ((row >> 1) & 3) << 1; }

struct Frags { bf16x8 y[4], x[4]; };
// the eight 16-channel x 32-pixel fragments of a wave's 64 x 64 tile from one stage: off[n] = byte offset of fragment n's low half
// inside a sub-tile (the same for both operands), the +16-pixel half 2 KiB further (fsw(row + 16) == fsw(row))
__device__ __forceinline__ void read_frags(Frags& f, const char* ysub, const char* xsub, const int (&off)[4]) {
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ysub + off[n]));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ysub + off[n] + 16 * 128));
    f.y[n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(xsub + off[m]));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(xsub + off[m] + 16 * 128));
    f.x[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
__device__ __forceinline__ void mma_frags(f32x4 (&acc)[4][4], const Frags& f) {
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.y[n], f.x[m], acc[n][m], 0, 0, 0);
}

template <int NS>
__global__ void __launch_bounds__(256, 2) wgrad_group_kernel(const GpK a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware bijective remap: the dispatcher deals consecutive block ids round-robin to the 8 XCDs; give every XCD a contiguous range
  // of tiles, so that the tiles of one problem (same dY, same X) meet in one L2
  int logical;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int pi = 0;
  while (pi + 1 < a.nprob && logical >= a.p[pi + 1].tile0) ++pi;
  const GpProb P = a.p[pi];
  const int t = logical - P.tile0;
  const int tco = t / P.nci, tci = t - tco * P.nci;
  const int co0 = tco * 128, ci0 = tci * 128;
  const int tid = threadIdx.x, lrow = tid >> 3, pslot = tid & 7;
  const int wave_byte = (tid & ~63) * 16;
  const char* zero = reinterpret_cast<const char*>(g_gzero_page) + pslot * 16;

  // this thread's 16-byte piece of row `lrow` of each of the four sub-tiles of a stage
  const int ls = pslot ^ fsw(lrow);          // logical slot that must land in physical slot pslot of row lrow
  const char* src[4];
  bool colok[4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int cy = co0 + s * 64 + ls * 8, cx = ci0 + s * 64 + ls * 8;
    colok[s] = cy + 8 <= P.lddy; colok[2 + s] = cx + 8 <= P.ldx;
    src[s] = P.dy + ((size_t)lrow * P.lddy + cy) * 2;
    src[2 + s] = P.x + ((size_t)lrow * P.ldx + cx) * 2;
  }
  const size_t ystep = (size_t)GP_PK * P.lddy * 2, xstep = (size_t)GP_PK * P.ldx * 2;
  int prow = lrow, is_slot = 0;
  const uint32_t lds0 = tf::lds_addr_uniform(smem + wave_byte);      // this wave's 1 KiB (8 rows) of sub-tile 0 of ring slot 0
  // the DMA is issued from inline asm (lds_dma.h): hipcc must not see a pending LDS write, or it drains the ring with a
  // `s_waitcnt vmcnt(0)` in front of the first transposing fragment read of every stage
  auto issue = [&]() {
    const uint32_t st = lds0 + (uint32_t)(is_slot * GP_STAGE);
    const bool in = prow < a.M;
    const void* g[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      g[s] = reinterpret_cast<const void*>((in && colok[s]) ? reinterpret_cast<uintptr_t>(src[s]) : reinterpret_cast<uintptr_t>(zero));
      src[s] += s < 2 ? ystep : xstep;
    }
    tf::dma16_hidden4(g[0], g[1], g[2], g[3], st, st + GP_SUB, st + 2 * GP_SUB, st + 3 * GP_SUB);
    prow += GP_PK;
    if (++is_slot == NS) is_slot = 0;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave = tid >> 6, wco = wave & 1, wci = wave >> 1;
  const int l = tid & 63, li = l & 15, lg = l >> 4;
  int off[4];
  {
    const int row = lg * 4 + (li >> 2), fs = (li & 3) >> 1, fhalf = (li & 1) << 3;
#pragma unroll
    for (int n = 0; n < 4; ++n) off[n] = row * 128 + (((2 * n + fs) ^ fsw(row)) << 4) + fhalf;
  }
  const int ybase = wco * GP_SUB, xbase = 2 * GP_SUB + wci * GP_SUB;

  const int nst = (a.M + GP_PK - 1) / GP_PK;
  // prologue: stages 0 .. NS-2 in flight, stage 0 landed and read
#pragma unroll
  for (int j = 0; j < NS - 1; ++j) issue();            // past the end the rows come from the zero page: the DMA count stays uniform
  wait_vm<GP_L*(NS - 2)>();
  __builtin_amdgcn_s_barrier();
  Frags fa, fb;
  read_frags(fa, smem + ybase, smem + xbase, off);
  int rs = 1;                                           // ring slot of the NEXT stage to read
  // iteration st: stage st+1 landed -> barrier -> DMA of stage st+NS-1 into the slot stage st-1 left -> fragments of stage st+1
  // are read WHILE the MFMAs of stage st (fragments already in registers) run.  Two iterations per trip: named fragment sets.
  auto step = [&](Frags& cur, Frags& nxt) {
    wait_vm<GP_L*(NS - 3)>();                           // this wave's pieces of stage st+1 landed (stages st+2 .. may be in flight)
    __builtin_amdgcn_s_barrier();                       // everyone's pieces landed; everyone finished the MFMAs of stage st-1
    issue();
    const char* sb = smem + rs * GP_STAGE;
    read_frags(nxt, sb + ybase, sb + xbase, off);
    __builtin_amdgcn_sched_barrier(0);                  // the 16 fragment reads are ISSUED before the first MFMA (hipcc would hoist the
    mma_frags(acc, cur);                                //  register-only MFMAs above the barrier and expose the LDS latency every stage)
    if (++rs == NS) rs = 0;
  };
  int st = 0;
  for (; st + 2 <= nst - 1; st += 2) { step(fa, fb); step(fb, fa); }
  if (st < nst - 1) { step(fa, fb); mma_frags(acc, fb); }
  else mma_frags(acc, fa);
  wait_vm<0>();                                         // the zero-page pieces issued past the end, before the LDS is released

  // D[co][ci]: lane holds co = (l >> 4) * 4 + r, ci = l & 15 of every 16 x 16 tile.  Single writer: plain stores.
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wco * 64 + n * 16 + lg * 4 + r;
      if (co < P.Cout) {
        float* drow = P.dw + (size_t)co * P.dw_ld;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int ci = ci0 + wci * 64 + m * 16 + li;
          if (ci < P.Cin) drow[ci] = acc[n][m][r];
        }
      }
    }
}


// ---- the fast form (r4b): every problem has 128-multiples of channels and the pixel count is a multiple of NS x 32.
// The SQ pass over the generic kernel above (profiles/r04_wgrad_group.txt) says where its time goes: matrix pipe busy 34 % of a wave's cycles,
// wave ACTIVE (issuing) 50 %, 742 cycles per 32-pixel stage against 256 of MFMA -- one wave per SIMD, ~100 non-MFMA instructions per stage
// (64-bit source pointers selected against the zero page and advanced per lane, ring-slot arithmetic on the 8 fragment addresses, M0
// juggling), issued in a burst BEHIND the 16 back-to-back MFMAs instead of underneath them.  Here:
//   * the DMA addresses are a SCALAR base (advanced with two SALU instructions per operand and stage) + a loop-invariant 32-bit lane
//     offset (`global_load_lds_dwordx4 voffset, s[base]`): no per-lane address arithmetic, no zero-page select (full tiles only; the
//     stages issued past the end re-read the last one into slots nobody consumes);
//   * the loop is unrolled over the ring (NS steps per trip), so every LDS address -- fragment reads and DMA destinations -- is a
//     loop-invariant register plus an immediate;
//   * each MFMA is followed by one fragment read of the NEXT stage (sched_group_barrier pairs): the LDS latency and the issue slots of the
//     reads hide under the matrix pipe of the same wave.
template <int NS>
__global__ void __launch_bounds__(256, NS <= 4 ? 2 : 1) wgrad_group_fast_kernel(const GpK a) {
  static_assert(NS == 4 || NS == 8, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int logical;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int pi = 0;
  while (pi + 1 < a.nprob && logical >= a.p[pi + 1].tile0) ++pi;
  const GpProb P = a.p[pi];
  const int t = logical - P.tile0;
  const int tco = t / P.nci, tci = t - tco * P.nci;
  const int tid = threadIdx.x, lrow = tid >> 3, pslot = tid & 7;
  const int ls = pslot ^ fsw(lrow);
  // loop-invariant lane offsets (bytes) of this thread's 16-byte pieces: row lrow, logical slot ls of the two 64-channel halves
  const uint32_t voy0 = (uint32_t)((lrow * P.lddy + ls * 8) * 2), voy1 = voy0 + 128u;
  const uint32_t vox0 = (uint32_t)((lrow * P.ldx + ls * 8) * 2), vox1 = vox0 + 128u;
  // scalar bases of the stage about to be issued, advanced by one 32-pixel stage per issue and clamped at the last stage
  const char* ysb = P.dy + (size_t)tco * 256;
  const char* xsb = P.x + (size_t)tci * 256;
  const uint32_t ystep = (uint32_t)(GP_PK * P.lddy * 2), xstep = (uint32_t)(GP_PK * P.ldx * 2);
  const int nst = a.M / GP_PK;
  int issued = 0;
  const uint32_t lds0 = tf::lds_addr_uniform(smem + (tid & ~63) * 16);
  auto issue = [&](int slot) {
    const uint32_t st = lds0 + (uint32_t)(slot * GP_STAGE);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %6\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %6\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voy0), "v"(voy1), "v"(vox0), "v"(vox1), "s"(ysb), "s"(xsb), "s"(st), "s"(st + GP_SUB), "s"(st + 2 * GP_SUB), "s"(st + 3 * GP_SUB)
        : "memory");
    if (++issued < nst) { ysb += ystep; xsb += xstep; }       // (scalar; past the end: the last stage again, into a slot nobody reads)
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave = tid >> 6, wco = wave & 1, wci = wave >> 1;
  const int l = tid & 63, li = l & 15, lg = l >> 4;
  // fragment bases: slot 0, operand sub-tile of this wave; everything else is an immediate (NS = 8: slots 4-7 through a second set of
  // registers 64 KiB further, the DS offset field has 16 bits)
  const char* yb[4]; const char* xb[4];
  {
    const int row = lg * 4 + (li >> 2), fs = (li & 3) >> 1, fhalf = (li & 1) << 3;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int o = row * 128 + (((2 * n + fs) ^ fsw(row)) << 4) + fhalf;
      yb[n] = smem + wco * GP_SUB + o; xb[n] = smem + 2 * GP_SUB + wci * GP_SUB + o;
    }
  }
  auto read = [&](Frags& f, int slot) {
    const int so = slot * GP_STAGE;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(yb[n] + so));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(yb[n] + so + 16 * 128));
      f.y[n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(xb[m] + so));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(xb[m] + so + 16 * 128));
      f.x[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

#pragma unroll
  for (int j = 0; j < NS - 1; ++j) issue(j);
  wait_vm<GP_L*(NS - 2)>();
  __builtin_amdgcn_s_barrier();
  Frags fa, fb;
  read(fa, 0);
  // step j of a trip (stage st = trip * NS + j): the fragments of stage st are in `cur`; stage st+1 must have landed -> barrier ->
  // DMA of stage st + NS - 1 into the slot stage st-1 left -> reads of stage st+1 into `nxt`, one behind each MFMA of stage st
  auto step = [&](Frags& cur, Frags& nxt, int j) {
    wait_vm<GP_L*(NS - 3)>();
    __builtin_amdgcn_s_barrier();
    issue((j + NS - 1) % NS);
    read(nxt, (j + 1) % NS);
    mma_frags(acc, cur);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA ...
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // ... one fragment read of the next stage
    }
  };
  for (int trip = 0; trip < nst / NS; ++trip) {
#pragma unroll
    for (int j = 0; j < NS; j += 2) { step(fa, fb, j); step(fb, fa, j + 1); }
  }
  wait_vm<0>();

  const int co0 = tco * 128, ci0 = tci * 128;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* drow = P.dw + (size_t)(co0 + wco * 64 + n * 16 + lg * 4 + r) * P.dw_ld + ci0 + wci * 64 + li;
#pragma unroll
      for (int m = 0; m < 4; ++m) drow[m * 16] = acc[n][m][r];
    }
}

}  // namespace

// n pointwise problems (1x1, stride 1, pad 0, bf16, no prologue, all with the same pixel count) in one launch; dw is OVERWRITTEN.
// TF_ERR_UNSUPPORTED when a problem does not qualify (the caller runs tf_conv2d_wgrad per problem instead).
int tf_wgrad_pw_group_launch(const tf_wgrad_args* A, int n, hipStream_t stream) {
  if (n <= 0 || n > GP_MAX) return TF_ERR_UNSUPPORTED;
  GpK k;
  k.M = A[0].N * A[0].OH * A[0].OW; k.nprob = n; k.pad_ = 0;
  int tiles = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < n; ++i) {
    const tf_wgrad_args& q = A[i];
    if (q.dtype != TF_BF16 || q.pro_scale || q.packed) return TF_ERR_UNSUPPORTED;
    if (q.KH != 1 || q.KW != 1 || q.stride != 1 || q.pad != 0 || q.H != q.OH || q.W != q.OW) return TF_ERR_UNSUPPORTED;
    if (q.N * q.OH * q.OW != k.M || q.ldx % 8 || q.lddy % 8 || q.ldx < q.Cin || q.lddy < q.Cout) return TF_ERR_UNSUPPORTED;
    GpProb& p = k.p[i];
    p.x = (const char*)q.x; p.dy = (const char*)q.dy; p.dw = q.dw_oihw; p.Cin = q.Cin; p.Cout = q.Cout; p.ldx = q.ldx; p.lddy = q.lddy;
    p.dw_ld = q.dw_ld ? q.dw_ld : q.Cin; p.nci = (q.Cin + 127) / 128; p.tile0 = tiles; p.pad_ = 0;
    tiles += ((q.Cout + 127) / 128) * p.nci;
    flops += 2.0 * k.M * q.Cout * q.Cin;
    bytes += ((double)k.M * q.Cout + (double)k.M * q.Cin) * 2 + (double)q.Cout * q.Cin * 4;
  }
  k.ntiles = tiles;
  bool full = k.M % GP_PK == 0;
  for (int i = 0; i < n; ++i) full = full && A[i].Cout % 128 == 0 && A[i].Cin % 128 == 0;
  // Measured on the layer-3 group of eight (256 tiles, M = 12 288; profiles/r04_wgrad_group.txt): generic kernel 161 us with a 4-deep ring,
  // 147 us with 8; fast kernel 148 / 136 us (4 / 8 slots) = 0.30 of the bf16 MFMA peak.  What bounds it is the ISSUE of the LDS-DMA itself:
  // a 1 KiB piece stalls the issuing wave ~100 cycles (MI355X_MICROARCH.md: 60-185) and a 128 x 128 x 32 stage is four pieces per wave
  // against 256 cycles of MFMA, one wave per SIMD -- the matrix pipe idles while its only wave feeds the DMA queue (SQ pass: MFMA busy
  // 38 %, issue-stalled 44 %).  The step does not see the difference any more (A/B 1177.7 / 1177.7 img/s): the second stream is off the
  // critical path since the grouping.
  const size_t lds = (size_t)4 * GP_STAGE;
  static tf::PerDevice attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_fast_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_fast_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // kind 18 = grouped pointwise weight gradient (bench.py tables); the GEMM view is the SUM over the group
  tf::ProfScope prof(18, flops, bytes, stream, k.M, A[0].Cout, A[0].Cin, 1, 2, 0, -1.0, true);
  const int fast_ns = tf::tuning().wgradg_fast;      // 0: the generic kernel (A/B); 4 / 8: ring depth of the fast one
  if (full && fast_ns == 4 && (k.M / GP_PK) % 4 == 0) {
    TF_LAUNCH_TIMED((wgrad_group_fast_kernel<4>), dim3(tiles), dim3(256), (size_t)4 * GP_STAGE, stream, k);
  } else if (full && fast_ns == 8 && (k.M / GP_PK) % 8 == 0) {
    TF_LAUNCH_TIMED((wgrad_group_fast_kernel<8>), dim3(tiles), dim3(256), (size_t)8 * GP_STAGE, stream, k);
  } else {
    TF_LAUNCH_TIMED((wgrad_group_kernel<4>), dim3(tiles), dim3(256), lds, stream, k);       // ragged shapes: zero-page selects, 64 KiB ring
  }
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}
