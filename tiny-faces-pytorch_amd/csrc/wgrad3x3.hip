// wgrad3x3: weight gradient of a 3x3 / stride 1 / pad 1 convolution with ALL NINE TAPS per block (gfx950).
//
//   dW[co][tap][ci] += sum_p dY[p][co] * X[p + shift(tap)][ci]
//
// The reduction runs over the ZERO-PADDED frame: pixel index p' walks the (H+2) x (W+2) raster of every image, dY rows at pad
// positions are zero (DMA from a zero page), X rows at pad positions are zero.  In that raster the input pixel of tap (kh, kw) is
// always p' + (kh-1)*(W+2) + (kw-1): one constant row shift per tap, no border masks, no per-pixel address tables -- the price is
// (H+2)(W+2)/(HW) - 1 wasted MFMA work (13 % at 32x32, 6 % at 63x63, 3 % at 125x125).
//
// What it buys over wgrad_dma_kernel<0> (one block per tap, csrc/wgrad_dma.hip):
//   * the dY tile (the operand all nine taps share) and the X rows are DMA-ed ONCE per 64-pixel stage instead of nine times:
//     16 KiB of LDS-DMA per 9 x 262 144 MACs instead of per 262 144;
//   * X lives in a 512-row circular LDS image (64 KiB): a stage adds 64 new rows, the (W+2)+1 rows of halo on either side are
//     already there, and every tap reads its fragments with ds_read_b64_tr_b16 from rows shifted by its constant;
//   * 8 waves per block (2 x 4 over the 64 x 64 output tile: 32 co x 16 ci each), two per SIMD, so that one wave's fragment
//     reads / address arithmetic overlap its partner's MFMAs (the 4-wave first version of this kernel spent 2.5 us per stage,
//     5x its MFMA time: profiles/r02_wgrad3x3.txt); per stage a wave issues 36 MFMAs behind ONE barrier against 22 fragment reads;
//   * nine accumulator sets (72 registers per wave) and ONE epilogue instead of nine.
// Split-K over the padded pixel range.  Epilogue, two forms:
//   * partial workspace given: every block stores its 9 x 64 x 64 fp32 tile with plain stores ([slice][tile][tap][co][ci]) and a
//     second kernel sums the slices straight into dW (OIHW or packed) -- no memset, no transposing copy, no atomics: with one
//     block per CU the atomic form issues 256 x 36 864 = 9.4 M L2 atomics per launch, which alone cost 31 us (measured);
//   * no workspace: fp32 atomics into dW (the C-ABI default, tests).
// bf16 only, no prologue.
#include <cstdlib>
#include "common.h"
#include "tuning.h"
#include "profile.h"
#include "lds_dma.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ uint4 g_w3zero_page[8];

struct W3K {
  const char* x; const char* dy; float* dw; float* partial;
  int N, H, W, Cin, Cout, ldx, lddy, dw_ld, ci_stride, tap_stride;
  int Hp, Wp, HWp, Mp;              // padded frame: H+2, W+2, their product, N * HWp
  int nco, nci, splitk, chunk, hb;  // hb = halo in 64-row chunks on either side: ceil((Wp + 1) / 64)
  int q64, r64, small_frame, dbg;   // 64 = q64 * Wp + r64; small_frame: a frame of <= q64 + 1 rows (loop in PadPos::advance)
  int direct;                       // 1: single writer per output element (splitk == 1): plain stores instead of fp32 atomics
};
// r4: a GROUP of problems of identical shape in one launch (the conv2 weight gradients of the identity bottlenecks of layer 3): block ->
// (problem, tile); x / dy / dw per problem.  With splitk = 1 every tile reduces over the WHOLE padded raster in-block: no slices, no
// partial tiles, no summing kernel, no atomics (tf_conv2d_wgrad_group, csrc/wgrad.hip).
constexpr int W3_MAXG = 24;
struct W3G { const char* x[W3_MAXG]; const char* dy[W3_MAXG]; float* dw[W3_MAXG]; };

constexpr int PK = 64, NS = 3, YT = PK * 128, XROWS = 512, XBYTES = XROWS * 128, L = 2, NT = 512;
constexpr int TILE_FLOATS = 9 * 64 * 64;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// r4: issued from inline asm (lds_dma.h).  Through __builtin_amdgcn_global_load_lds hipcc saw a pending LDS write and put an
// `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 of EVERY stage (the transposing read carries no memory operand the
// wait-count pass could disambiguate): the ring was drained once per stage, each stage paid the full latency of the DMA issued just
// before it.  The counted vmcnt + barrier of the loop is what orders the fragment reads behind the DMA.
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  tf::dma16_hidden(gsrc, tf::lds_addr_uniform(lds_wave_base));
}
// same source-side swizzle as wgrad_dma.hip: conflict-free ds_read_b64_tr_b16 for ANY eight consecutive rows (the shifted reads of
// the taps start at arbitrary rows; 512 is a multiple of 8, so the ring wrap keeps the pattern)
__device__ __forceinline__ int fsw(int row) { return ((row >> 1) & 3) << 1; }

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// (image, padded row, padded column) of a padded pixel index, advanced 64 at a time
struct PadPos {
  int n, hp, wp;
  __device__ __forceinline__ void init(int r, const W3K& a) {       // r >= -2 * PK (the halo in front of the first pixel)
    const int kk = (2 * PK + a.HWp - 1) / a.HWp;
    const int rr = r + kk * a.HWp;
    const int q = rr / a.HWp;
    n = q - kk;
    const int rem = rr - q * a.HWp;
    hp = rem / a.Wp; wp = rem - hp * a.Wp;
  }
  // + 64 pixels = + q64 rows + r64 columns (q64 = 64 / Wp, r64 = 64 % Wp: launch constants), carries by select: no loop, no
  // divergent branch for any frame of more than q64 + 1 rows; smaller frames (H <= 2 at W <= 30 ...) take the loop
  template <bool SMALL>
  __device__ __forceinline__ void advance(const W3K& a) {
    wp += a.r64;
    const int c = wp >= a.Wp ? 1 : 0;
    wp -= c * a.Wp;
    hp += a.q64 + c;
    if (SMALL) { while (hp >= a.Hp) { hp -= a.Hp; ++n; } }
    else { const int c2 = hp >= a.Hp ? 1 : 0; hp -= c2 * a.Hp; n += c2; }
  }
  __device__ __forceinline__ bool interior(const W3K& a) const {       // (bitwise: no short-circuit control flow)
    return ((unsigned)n < (unsigned)a.N) & ((unsigned)(hp - 1) < (unsigned)a.H) & ((unsigned)(wp - 1) < (unsigned)a.W);
  }
  __device__ __forceinline__ int pixel(const W3K& a) const { return (n * a.H + (hp - 1)) * a.W + (wp - 1); }
};

typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int STG_PITCH = 68;                            // floats per staged [co] row of 64 ci: 272 B, 16-byte aligned, conflict-light

__device__ __forceinline__ bf16x8 read_tr_pair(const char* base, int off_lo, int off_hi) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off_lo));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off_hi));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// The K loop is written for INSTRUCTION COUNT (the first version re-derived every fragment address per tap and stage and walked
// the padded raster with data-dependent loops: ~410 instructions, 8 divergent branches per stage and wave for 36 MFMAs):
//   * the 18 X-fragment ring offsets of a lane (9 taps x 2 k-steps) are registers that advance by 64 rows (8 KiB) per stage
//     modulo the 64 KiB ring; the swizzle term is stage-invariant (64 and the 512-row wrap are multiples of 8 rows);
//   * the 4 dY-fragment offsets are fixed; the ring slot is a scalar add;
//   * the DMA source walk is branch-free (PadPos::advance).
template <bool SMALL, bool GROUPED>
__device__ __forceinline__ void wgrad3x3_body(const W3K& a, const char* const gx, const char* const gdy, float* const gdw, int b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const yring = smem;
  char* const xring = smem + NS * YT;
  const int ks = b % a.splitk; b /= a.splitk;
  const int tci = b % a.nci; b /= a.nci;
  const int tco = b;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int pb = ks * a.chunk, pe = min(a.Mp, pb + a.chunk);
  const int nst = (pe - pb + PK - 1) / PK;               // >= 1: the host sizes splitk so that every slice owns pixels
  const int x0 = pb - PK * a.hb;                         // padded index of ring row 0
  const int D = 2 * a.hb;                                // stage st needs X chunks st .. st + D
  const int tid = threadIdx.x, r = tid >> 3, pslot = tid & 7;      // one row of every 64-row group per thread
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* zero = reinterpret_cast<const char*>(g_w3zero_page) + pslot * 16;

  // physical slot pslot of row r receives logical slot pslot ^ fsw(r) (swizzle on the SOURCE)
  const int ls = pslot ^ fsw(r);
  PadPos ypos, xpos;
  ypos.init(pb + r, a);
  xpos.init(x0 + r, a);
  int yrow = pb + r;                                     // padded index of this thread's dY row in the NEXT stage to issue
  const int ycol = co0 + ls * 8, xcol = ci0 + ls * 8;
  const bool ycok = ycol + 8 <= a.lddy, xcok = xcol + 8 <= a.ldx;
  const char* const ybase = gdy + (size_t)ycol * 2;
  const char* const xbase = gx + (size_t)xcol * 2;
  int ys_slot = 0, xc = 0;                               // dY ring slot / X chunk index of the next issue (scalar)
  auto issue_y = [&]() {
    const bool ok = (yrow < pe) & ycok & ypos.interior(a);
    const unsigned off = ok ? (unsigned)ypos.pixel(a) * (unsigned)(a.lddy * 2) : 0u;        // (the launch checks that a tensor is < 4 GiB)
    dma16((ok ? ybase : zero) + off, yring + ys_slot * YT + wave_u * 1024);
    ypos.template advance<SMALL>(a); yrow += PK;
    if (++ys_slot == NS) ys_slot = 0;
  };
  auto issue_x = [&]() {
    const bool ok = xcok & xpos.interior(a);
    const unsigned off = ok ? (unsigned)xpos.pixel(a) * (unsigned)(a.ldx * 2) : 0u;
    dma16((ok ? xbase : zero) + off, xring + (xc & 7) * (PK * 128) + wave_u * 1024);
    xpos.template advance<SMALL>(a);
    ++xc;
  };

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave = tid >> 6, wco = wave & 1, wci = wave >> 1;      // 2 x 4 waves: 32 output channels x 16 input channels each

  // fragment offsets (bytes).  Transposing read of 16 channels x 32 pixels: lane (i = l & 15, g = l >> 4) reads rows
  // g*4 + (i >> 2) and +16 of the 32-pixel k-step, 8 bytes at 16-byte slot (c0 >> 3) + ((i & 3) >> 1), half (i & 1)
  const int l = tid & 63, li = l & 15, lg = l >> 4;
  const int frow = lg * 4 + (li >> 2), fhalf = (li & 1) << 3, fs = (li & 3) >> 1;
  int yoff[2][2];                                        // [k-step][n]: inside a 64-row dY stage tile; the +16-row half is +2048
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int row = k * 32 + frow, slot = ((wco * 32 + n * 16) >> 3) + fs;
      yoff[k][n] = row * 128 + ((slot ^ fsw(row)) << 4) + fhalf;
    }
  int xoff[9][2];                                        // [tap][k-step]: inside the 512-row ring, for the CURRENT stage
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int shift = (t / 3 - 1) * a.Wp + (t % 3 - 1);
      const int row = (PK * a.hb + k * 32 + shift + frow) & (XROWS - 1);        // stage 0: ring row of padded pixel pb + k*32 + shift
      const int slot = ((wci * 16) >> 3) + fs;
      xoff[t][k] = row * 128 + ((slot ^ fsw(row)) << 4) + fhalf;
    }

  // prologue: the halo chunks 0 .. D-1 first, then the (dY stage, X chunk) pairs of stages 0 and 1.  From then on every loop
  // iteration issues exactly one pair (L = 2 DMA instructions per thread), so vmcnt(L) == "everything but the youngest pair landed"
  for (int c = 0; c < D; ++c) issue_x();
#pragma unroll
  for (int j = 0; j < NS - 1; ++j) { issue_y(); issue_x(); }       // past the slice the rows come from the zero page: the count stays uniform
  int cs = 0;
  for (int st = 0; st < nst; ++st) {
    wait_vm<L*(NS - 2)>();                               // pair (st+1) may still be in flight; pair st and all older ones landed
    __builtin_amdgcn_s_barrier();                        // everyone's pieces of stage st landed; stage st-1 fully consumed
    issue_y(); issue_x();                                // pair st + 2 (dY slot (st+2) % 3, X chunk st + D + 2)
    if (!(a.dbg & 1)) {
      const char* ys = yring + cs * YT;
      // r4: ALL 22 fragment reads of a k-step are issued before its first MFMA, and the reads of the second k-step before the MFMAs of the
      // first (two fragment sets, 88 registers).  The r2-r3 loop read tap t+1 while the two MFMAs of tap t ran: 32 cycles of cover for an
      // LDS round trip of > 100 -- every tap of every stage waited (1.3 us per stage against 0.48 us of MFMA time for the two waves of a
      // SIMD, profiles/r04_wgrad_group.txt).  The full-K grouped form runs 217 stages per block: the loop IS the kernel now.
      bf16x8 fy0[2], fy1[2], fx0[9], fx1[9];
#pragma unroll
      for (int n = 0; n < 2; ++n) fy0[n] = read_tr_pair(ys, yoff[0][n], yoff[0][n] + 16 * 128);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int o = xoff[t][0];
        fx0[t] = read_tr_pair(xring, o, (o + 16 * 128) & (XBYTES - 1));                     // fsw(row + 16) == fsw(row)
        xoff[t][0] = (o + PK * 128) & (XBYTES - 1);                                         // next stage: 64 rows further round the ring
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) fy1[n] = read_tr_pair(ys, yoff[1][n], yoff[1][n] + 16 * 128);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int o = xoff[t][1];
        fx1[t] = read_tr_pair(xring, o, (o + 16 * 128) & (XBYTES - 1));
        xoff[t][1] = (o + PK * 128) & (XBYTES - 1);
      }
      __builtin_amdgcn_sched_barrier(0);                   // (hipcc would sink the reads back in front of their consumers)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy0[n], fx0[t], acc[t][n], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy1[n], fx1[t], acc[t][n], 0, 0, 0);
    }
    if (++cs == NS) cs = 0;
  }
  wait_vm<0>();                                          // drain the pairs issued past the end before the LDS is released

  if (a.partial) {
    // [slice][tile][tap][co 64][ci 64] fp32.  Through LDS, three taps at a time, so that a lane stores 16 contiguous bytes and a
    // wave 1 KiB: the direct form (72 4-byte stores per lane, 16 lanes per 64-byte run) was bound by store ISSUE
    // (147 KiB per block at ~7 B/clk: as long as the whole K loop).
    if (a.dbg & 2) return;
    float* out = a.partial + ((size_t)ks * (a.nco * a.nci) + (size_t)tco * a.nci + tci) * TILE_FLOATS;
    float* stg = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t0 = 0; t0 < 9; t0 += 3) {
      __syncthreads();                                   // rings (first round) / previous round fully read
#pragma unroll
      for (int tl = 0; tl < 3; ++tl)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            stg[(tl * 64 + wco * 32 + n * 16 + lg * 4 + q) * STG_PITCH + wci * 16 + li] = acc[t0 + tl][n][q];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int e4 = tid + k * NT;                     // float4 index inside the 3 x 64 x 64 round
        const int row = e4 >> 4, c4 = e4 & 15;           // row = tl * 64 + co
        const f32x4v v = *reinterpret_cast<const f32x4v*>(stg + row * STG_PITCH + c4 * 4);
        *reinterpret_cast<f32x4v*>(out + (size_t)t0 * 4096 + (size_t)e4 * 4) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int ci = ci0 + wci * 16 + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + wco * 32 + n * 16 + lg * 4 + q;
        if (co < a.Cout && ci < a.Cin) {
          float* d = gdw + (size_t)co * a.dw_ld + (size_t)ci * a.ci_stride + (size_t)t * a.tap_stride;
          if (GROUPED || a.direct) *d = acc[t][n][q]; else atomicAdd(d, acc[t][n][q]);
        }
      }
    }
}

template <bool SMALL>
__global__ void __launch_bounds__(NT, 2) wgrad3x3_kernel(const W3K a) {
  wgrad3x3_body<SMALL, false>(a, a.x, a.dy, a.dw, blockIdx.x);
}
// grouped form: splitk == 1, no partial workspace; the tiles of one problem are consecutive in the XCD-remapped block order (they
// stream the same dY / X rows: one XCD's L2 serves all 16 of them)
template <bool SMALL>
__global__ void __launch_bounds__(NT, 2) wgrad3x3_group_kernel(const W3K a, const W3G g) {
  const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int per = a.nco * a.nci;
  const int grp = logical / per;
  wgrad3x3_body<SMALL, true>(a, g.x[grp], g.dy[grp], g.dw[grp], logical - grp * per);
}

// dW += sum over the slices of the partial tiles.  256 threads = (256 / SG) float4 columns x SG slice groups; SG is chosen so that a
// thread sums >= 8 slices (SG = 2 for the 16 slices of layer 3, 16 for the 256 slices of layer 1's single tile): enough loads in
// flight per thread, and a one-tile layer still spreads over hundreds of blocks.  The SG partial sums meet in LDS.
template <int SG>
__global__ void __launch_bounds__(256) wgrad3x3_reduce_kernel(const W3K a) {
  constexpr int COLS = 256 / SG;
  __shared__ float4 part[SG][COLS + 1];
  const int ntiles = a.nco * a.nci;
  const size_t total4 = (size_t)ntiles * TILE_FLOATS / 4;              // a multiple of 256
  const int col = threadIdx.x % COLS, sg = threadIdx.x / COLS;
  const size_t e4 = (size_t)blockIdx.x * COLS + col;
  const size_t stride = (size_t)ntiles * TILE_FLOATS;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e4 < total4) {
    const float* src = a.partial + e4 * 4;
#pragma unroll 4
    for (int ks = sg; ks < a.splitk; ks += SG) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)ks * stride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (SG > 1) {
    part[sg][col] = s;
    __syncthreads();
    if (sg != 0) return;
#pragma unroll
    for (int g = 1; g < SG; ++g) { const float4 v = part[g][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  }
  if (e4 >= total4) return;
  const size_t e = e4 * 4;
  const int tile = (int)(e / TILE_FLOATS), rem = (int)(e - (size_t)tile * TILE_FLOATS);
  const int t = rem / 4096, co = (tile / a.nci) * 64 + (rem % 4096) / 64, ci = (tile % a.nci) * 64 + rem % 64;
  if (co >= a.Cout) return;
  float* d = a.dw + (size_t)co * a.dw_ld + (size_t)t * a.tap_stride;
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (ci + j < a.Cin) d[(size_t)(ci + j) * a.ci_stride] += v[j];
}

// shape checks + split-K plan shared by the launch and the workspace query
bool w3_plan(const tf_wgrad_args* A, W3K& k) {
  if (A->dtype != TF_BF16 || A->pro_scale) return false;
  if (A->KH != 3 || A->KW != 3 || A->stride != 1 || A->pad != 1 || A->OH != A->H || A->OW != A->W) return false;
  if (A->W + 3 > 2 * PK) return false;                         // halo of at most two 64-row chunks per side (512-row ring)
  k.x = (const char*)A->x; k.dy = (const char*)A->dy; k.dw = A->dw_oihw; k.partial = nullptr;
  k.N = A->N; k.H = A->H; k.W = A->W; k.Cin = A->Cin; k.Cout = A->Cout; k.ldx = A->ldx; k.lddy = A->lddy; k.dw_ld = A->dw_ld;
  if (A->packed) { k.ci_stride = 1; k.tap_stride = A->Cin; } else { k.ci_stride = 9; k.tap_stride = 1; }
  k.Hp = A->H + 2; k.Wp = A->W + 2; k.HWp = k.Hp * k.Wp;
  const long long mp = (long long)A->N * k.HWp;
  if (mp > (1ll << 30)) return false;
  if ((long long)A->N * A->H * A->W * 2 * (A->ldx > A->lddy ? A->ldx : A->lddy) >= (1ll << 32)) return false;       // 32-bit byte offsets in the DMA walk
  k.Mp = (int)mp;
  k.hb = (k.Wp + 1 + PK - 1) / PK;
  k.q64 = PK / k.Wp; k.r64 = PK % k.Wp; k.small_frame = k.Hp <= k.q64 + 1; k.dbg = 0; k.direct = 0;
  k.nco = (A->Cout + 63) / 64; k.nci = (A->Cin + 63) / 64;
  const int tiles = k.nco * k.nci;
  int sk = A->splitk;
  if (sk <= 0) {
    // one block per CU (88 KiB of LDS): split the padded pixel range until ~256 blocks exist, but keep >= 6 stages per block
    // so that the 2*hb halo chunks and the nine-tap epilogue stay a small part of a block
    const int target = tf::tuning().wgrad3_blocks;
    sk = (target + tiles - 1) / tiles;
    const int maxsk = (k.Mp + 6 * PK - 1) / (6 * PK);
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
  }
  k.chunk = (((k.Mp + sk - 1) / sk) + PK - 1) / PK * PK;
  k.splitk = (k.Mp + k.chunk - 1) / k.chunk;
  return true;
}

}  // namespace

// bytes of the partial-tile workspace the two-phase epilogue needs for this call (0: the all-taps kernel does not apply)
extern "C" size_t tf_wgrad_workspace_bytes(const tf_wgrad_args* A) {
  W3K k;
  if (!A || !w3_plan(A, k)) return 0;
  return (size_t)k.splitk * k.nco * k.nci * TILE_FLOATS * sizeof(float);
}

// 3x3, stride 1, pad 1, same-size output, bf16, no prologue, W <= 125.  TF_ERR_UNSUPPORTED otherwise (the caller falls back).
int tf_wgrad3x3_launch(const tf_wgrad_args* A, hipStream_t stream) {
  W3K k;
  if (!w3_plan(A, k)) return TF_ERR_UNSUPPORTED;
  const size_t need = (size_t)k.splitk * k.nco * k.nci * TILE_FLOATS * sizeof(float);
  const bool atomics_only = tf::tuning().wgrad3_atomics;      // A/B knob
  if (A->partial_ws && A->partial_ws_bytes >= need && !atomics_only) k.partial = (float*)A->partial_ws;
  const int dbg = tf::tuning().wgrad3_dbg;     // timing ablation: 1 = no MFMA loop body, 2 = no partial stores (results invalid)
  k.dbg = dbg;
  const size_t lds = (size_t)NS * YT + XBYTES;
  static tf::PerDevice attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const double Md = (double)A->N * A->H * A->W;
  // kind 16 = wgrad3x3 (its own row in bench.py's tables); the bracket spans BOTH launches of the two-phase form: the summing
  // kernel is a mandatory part of the gradient
  tf::ProfScope prof(16, 2.0 * Md * A->Cout * A->Cin * 9, (Md * A->Cout + Md * A->Cin) * 2 + (double)A->Cout * A->Cin * 9 * 4, stream, (int)Md,
                     A->Cout, A->Cin * 9, 9, 2, 0);
  if (k.small_frame) hipLaunchKernelGGL(wgrad3x3_kernel<true>, dim3(k.nco * k.nci * k.splitk), dim3(NT), lds, stream, k);
  else hipLaunchKernelGGL(wgrad3x3_kernel<false>, dim3(k.nco * k.nci * k.splitk), dim3(NT), lds, stream, k);
  if (k.partial) {
    const size_t total4 = (size_t)k.nco * k.nci * TILE_FLOATS / 4;
    if (k.splitk >= 128)     hipLaunchKernelGGL(wgrad3x3_reduce_kernel<16>, dim3((unsigned)(total4 / 16)), dim3(256), 0, stream, k);
    else if (k.splitk >= 32) hipLaunchKernelGGL(wgrad3x3_reduce_kernel<4>, dim3((unsigned)(total4 / 64)), dim3(256), 0, stream, k);
    else if (k.splitk >= 16) hipLaunchKernelGGL(wgrad3x3_reduce_kernel<2>, dim3((unsigned)(total4 / 128)), dim3(256), 0, stream, k);
    else                     hipLaunchKernelGGL(wgrad3x3_reduce_kernel<1>, dim3((unsigned)(total4 / 256)), dim3(256), 0, stream, k);
  }
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

// n problems of the SAME shape (3x3 / stride 1 / pad 1, bf16) in one launch, every tile reduced over the whole padded raster in-block;
// dw is OVERWRITTEN (plain stores).  TF_ERR_UNSUPPORTED when the shape does not qualify or the problems differ in shape.
int tf_wgrad3x3_group_launch(const tf_wgrad_args* A, int n, hipStream_t stream) {
  if (n <= 0 || n > W3_MAXG) return TF_ERR_UNSUPPORTED;
  tf_wgrad_args first = A[0];
  first.splitk = 1; first.partial_ws = nullptr; first.partial_ws_bytes = 0;
  W3K k;
  if (!w3_plan(&first, k) || k.splitk != 1) return TF_ERR_UNSUPPORTED;
  W3G g;
  for (int i = 0; i < n; ++i) {
    const tf_wgrad_args& q = A[i];
    if (q.dtype != first.dtype || q.pro_scale || q.N != first.N || q.H != first.H || q.W != first.W || q.Cin != first.Cin || q.Cout != first.Cout ||
        q.KH != 3 || q.KW != 3 || q.stride != 1 || q.pad != 1 || q.OH != q.H || q.OW != q.W || q.ldx != first.ldx || q.lddy != first.lddy ||
        q.dw_ld != first.dw_ld || q.packed != first.packed) return TF_ERR_UNSUPPORTED;
    g.x[i] = (const char*)q.x; g.dy[i] = (const char*)q.dy; g.dw[i] = q.dw_oihw;
  }
  for (int i = n; i < W3_MAXG; ++i) { g.x[i] = nullptr; g.dy[i] = nullptr; g.dw[i] = nullptr; }
  k.direct = 1;
  const size_t lds = (size_t)NS * YT + XBYTES;
  static tf::PerDevice attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_group_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_group_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const double Md = (double)first.N * first.H * first.W;
  // kind 19 = grouped all-taps 3x3 weight gradient; the GEMM view is the SUM over the group
  tf::ProfScope prof(19, 2.0 * Md * first.Cout * first.Cin * 9 * n, ((Md * first.Cout + Md * first.Cin) * 2 + (double)first.Cout * first.Cin * 9 * 4) * n, stream,
                     (int)Md, first.Cout, first.Cin * 9, 9, 2, 0, -1.0, true);
  const dim3 grid(k.nco * k.nci * n);
  if (k.small_frame) TF_LAUNCH_TIMED((wgrad3x3_group_kernel<true>), grid, dim3(NT), lds, stream, k, g);
  else TF_LAUNCH_TIMED((wgrad3x3_group_kernel<false>), grid, dim3(NT), lds, stream, k, g);
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}
