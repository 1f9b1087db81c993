#pragma once
// conv_dma (implementation header, r4: included by conv_dma.hip (fp32 + dispatch), conv_dma_bf16.hip and conv_dma_f16.hip so that the
// three dtypes compile in parallel): the implicit-GEMM convolution with an LDS-DMA operand pipeline (gfx950).
//
// Same GEMM view, MFMA fragments and XOR-swizzled LDS image as conv.hip; what changes is how
// operands reach LDS and how results leave:
//   * global_load_lds_dwordx4: every lane DMAs 16 bytes straight into LDS (no VGPR round
//     trip), a wave fills 1 KiB of the linear tile image per instruction.  The swizzle is
//     applied on the SOURCE address (lane l fills LDS slot l%8 of row l/8 with logical slot
//     (l%8)^h(row)); im2col padding / rows past M read a 128-byte zero page instead of branching.
//   * NS-deep LDS ring, counted `s_waitcnt vmcnt(N)` + raw s_barrier: NS-1 K-stages stay in
//     flight across barriers, so the ~1-2 us L2/HBM latency of a stage is overlapped with the
//     MFMAs of the previous ones instead of being paid once per stage (the register-staged
//     kernel is latency-bound at one stage in flight: profiles/r01a_microbench_layer_shapes.txt).
//   * epilogue through LDS: accumulators are parked as an fp32 [pixels][channels] tile, then
//     every thread handles 16 output bytes of one pixel row -> full 128-byte lines for the
//     store AND for the residual / mask operands, all epilogue math on 8 consecutive channels.
// Used whenever no producer-BN prologue has to be applied on the fly (eval forward, every
// data-gradient, conv1 / downsample / heads in training); conv.hip keeps the prologue path.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "tuning.h"
#include "profile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero_page[8];     // 128 zero bytes: source of padded / out-of-range rows

struct DmaK {
  const char* x; const char* w; char* y;
  const float* epi_scale; const float* epi_shift;
  const char* aux; const char* aux2; const char* aux3;
  const float* mask_scale; const float* mask_shift;
  float* stat_out; const float* stat_shift; float* stat_shift_out;
  int H, W, Cin, OH, OW, KW, stride, pad, sshift;
  int M, OHW, ldy, Ktot, cpt, nstages, ntiles, mode, epi, srows, mtiles, dbg;
  int scat, sc_hw, sc_w, sc_OH, sc_OW;   // scattered rows (stride-2 data gradients): GEMM row p = (n, h, w) of a half-resolution raster -> output pixel (n, 2h + ph, 2w + pw)
  int ph, pw, kh0, kw0;                  // scat == 2 (KIND 2): parity class of the output pixels and its first tap (taps kh0, kh0+2, .. x kw0, kw0+2, ..)
  int inplace;                           // r6, scat == 1 with aux == y: the scattered rows ACCUMULATE into a raster another launch already wrote (no initialisation;
                                         // statistic sums take the increment only)
  // pro == 1 (ring-less pointwise, 2-byte types): x := relu(bn(x)) applied to the pixel tile in LDS after its DMA landed (tf_conv_args.bnf)
  int pro, pf_rows; float pf_count, pf_eps, pf_mom;
  const float* pf_stat; const float* pf_gamma; const float* pf_beta; const float* pf_sshift;
  float* pf_scale; float* pf_shift; float* pf_mean; float* pf_invstd; float* pf_rmean; float* pf_rvar; char* pf_out;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) << 2) ^ (((row >> 3) & 1) * 6); }
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ swz(row)) << 4); }

template <typename T> struct MmaD;
template <> struct MmaD<tf::bf16_t> {
  static constexpr int KCH = 64;
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <> struct MmaD<tf::f16_t> {              // BASELINE.json configs[4]: fp16 MFMA (v_mfma_f32_16x16x32_f16), same fragment layout as bf16
  static constexpr int KCH = 64;
  template <int NF, int MF>
  __device__ static __forceinline__ void stage(const char* xs, const char* ws, int xrow0, int wrow0, f32x4 (&acc)[NF][MF]) {
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 xf[MF], wf[NF];
#pragma unroll
      for (int m = 0; m < MF; ++m) xf[m] = *reinterpret_cast<const f16x8*>(xs + lds_off(xrow0 + m * 16 + r, ks * 4 + g));
#pragma unroll
      for (int n = 0; n < NF; ++n) wf[n] = *reinterpret_cast<const f16x8*>(ws + lds_off(wrow0 + n * 16 + r, ks * 4 + g));
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[n], xf[m], acc[n][m], 0, 0, 0);
    }
  }
};
template <> struct MmaD<float> {
  static constexpr int KCH = 32;
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

// ---- 32x32x16 fragments (bf16 / fp16): a wave owns MF x NF fragments of 32 pixels x 32 channels (64 x 64 for the 128 x 128 block
// tile).  Per 16-deep k-step a wave reads MF + NF fragments (1 KiB each) for MF*NF MFMAs of 16 384 MACs: 16 MACs per LDS byte at
// 2 x 2 fragments, twice the 16x16x32 / 32 x 32-wave-tile form above -- the LDS read port is what bounded that form
// (profiles/r01d_conv_dma_pipeline_ablation.txt).  Lane l holds row l & 31 and the 8 k-values of 16-byte slot 2*ks + (l >> 5); the
// XOR swizzle h(row) is conflict-free for this read pattern too (checked exhaustively over the four ds_read_b128 lane groups).
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename T> struct Mma32;
template <> struct Mma32<tf::bf16_t> {
  typedef bf16x8 frag;
  __device__ static __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<tf::f16_t> {
  typedef f16x8 frag;
  __device__ static __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <typename T, int NF, int MF>
__device__ __forceinline__ void stage32(const char* xs, const char* ws, int xrow0, int wrow0, f32x16 (&acc)[NF][MF]) {
  typedef typename Mma32<T>::frag frag;
  const int l = threadIdx.x & 63, r = l & 31, h = l >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    frag xf[MF], wf[NF];
#pragma unroll
    for (int m = 0; m < MF; ++m) xf[m] = *reinterpret_cast<const frag*>(xs + lds_off(xrow0 + m * 32 + r, ks * 2 + h));
#pragma unroll
    for (int n = 0; n < NF; ++n) wf[n] = *reinterpret_cast<const frag*>(ws + lds_off(wrow0 + n * 32 + r, ks * 2 + h));
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int m = 0; m < MF; ++m) acc[n][m] = Mma32<T>::mma(wf[n], xf[m], acc[n][m]);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 16-byte DMA: global (per-lane address) -> LDS (wave-uniform base + lane*16)
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// KIND: 1 = pointwise GEMM (1x1, stride 1, pad 0: forward or data-gradient, no per-stage address logic at all),
//       0 = generic forward gather, 2 = generic transposed gather (data gradient of a KxK / strided conv)
// waves-per-EU 5: 92-96 VGPRs, accumulators in VGPRs, no spills (the default allocation is 90 + 16..24 AGPRs = 4 waves, which
// caps the 2-deep / ring-less variants at 4 blocks per CU).  Measured on one box: 4 and 5 waves tie (1034 img/s), 6 waves
// spill 32-92 bytes and lose 8 %; a software-pipelined fragment loop (reads of stage s+1 under the MFMAs of stage s) added
// +0.2 % -- at 4-5 blocks per CU the other blocks already cover a wave's LDS latency, so the simple stage() stays.
// MMA = 16: 16x16 fragments (fp32: 16x16x4, bf16/fp16: 16x16x32), up to five blocks of waves per SIMD;
// MMA = 32: 32x32x16 fragments (bf16/fp16), accumulators of a 64 x 64 wave tile = 64 registers -> two waves per SIMD
//           (__launch_bounds__(256, 2): up to 256 VGPR+AGPR per lane, no spills; profiles/r02*_kernel_resources.txt)
// EPIC (r4): -1 = the epilogue flags are read at run time (a.epi); >= 0 = they are THIS compile-time constant.  Every 16x16-fragment
// instantiation of rounds 1-3 spilled (12-18 VGPRs, 20-52 B of scratch per lane, profiles/r03_kernel_resources.txt) because the generic
// epilogue keeps the coefficient vectors of all its modes alive (affine 16 + mask 16 + statistic shift 8 registers) next to the prefetched
// residual / mask operands (32) and the accumulators; a training step uses three flag sets on 90 % of its launches (STATS;
// MASK | STATS2; RES | MASK2 | STATS3), which get an instantiation each for the hot pointwise tiles: dead modes fold away.
template <typename T, int BM, int BN, int NS, int KIND, int MMA = 16, int EPIC = -1>
__global__ void __launch_bounds__(256, MMA == 32 ? 2 : (NS <= 2 && BM == 128 && EPIC == (TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3) ? 3 : (NS == 1 && BM == 128 ? 4 : 5))) conv_dma_kernel(const DmaK a) {
  // (the hand-over instantiation of the ring-less 128 x 64 tile keeps three operand tiles in registers: 3 blocks per CU without spills --
  //  768 resident blocks, exactly two rounds of the 1536 tiles of a layer-3 launch -- instead of 4 with 8 spilled registers)
  const int epi_flags = EPIC >= 0 ? EPIC : a.epi;
  constexpr int KCH = MmaD<T>::KCH;
  constexpr int EPS = tf::Elem<T>::kPer16B;
  constexpr int XR = BM / 32, WR = BN / 32;
  constexpr int WM = BM / 2, WN = BN / 2, MF = WM / MMA, NF = WN / MMA;
  static_assert(MMA == 16 || (sizeof(T) == 2 && WM % 32 == 0 && WN % 32 == 0), "32x32x16 fragments: 2-byte operands, 32-multiples");
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, BUF = XBYTES + WBYTES;
  constexpr int L = XR + WR;                        // DMA instructions per thread per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int logical;
  {
    const int nb = gridDim.x, b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  // output row of GEMM row p: identity, or (r3) the even-even pixel of the 2x larger raster for the data gradient of a 1x1 / stride-2 conv.
  // That gradient is nonzero ONLY there; the generic transposed gather (KIND 2) visits all four parities of the output raster and reads the
  // zero page for three of them: 4x the rows, DMAs and MFMAs (layer2.0 / layer3.0 downsample: 141 / 136 us -> see launch_kind).
  auto orow = [&](int p) -> size_t {
    if (!a.scat) return (size_t)p;
    const int n = p / a.sc_hw, rem = p - n * a.sc_hw, h = rem / a.sc_w, w = rem - h * a.sc_w;
    return ((size_t)n * a.sc_OH + 2 * h + a.ph) * a.sc_OW + 2 * w + a.pw;
  };
  const int tid = threadIdx.x, lrow = tid >> 3, pslot = tid & 7;
  const int wave_byte = (tid & ~63) * 16;            // LDS byte offset of this wave's 1 KiB piece inside a 256-thread pass

  // Per-row gather state, fixed for the whole K loop.  Physical LDS slot pslot of row r must receive logical slot
  // pslot^h(r) (swizzle on the SOURCE).  Padded taps / rows past M read the zero page.
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + pslot * 16;
  const char* rowptr[XR];       // KIND 1: running source pointer;  else: pointer of tap (0,0) / channel chunk 0
  int rstep[XR];                // KIND 1: bytes to advance per stage (0 for rows on the zero page)
  int rb_h[XR], rb_w[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int row = lrow + i * 32;
    const int xs_ = (pslot ^ swz(row)) * 16;
    const int p = m0 + row;
    rowptr[i] = zero; rstep[i] = 0; rb_h[i] = -(1 << 28); rb_w[i] = -(1 << 28);
    if (p < a.M) {
      if (KIND == 1) {
        rowptr[i] = a.x + (size_t)p * a.Cin * sizeof(T) + xs_;
        rstep[i] = KCH * (int)sizeof(T);
      } else {
        int n, oh, ow;
        if (KIND == 2 && a.scat == 2) {              // row of the parity class's half-resolution raster -> its pixel of the output raster
          n = p / a.sc_hw; const int rem = p - n * a.sc_hw, sh = rem / a.sc_w;
          oh = 2 * sh + a.ph; ow = 2 * (rem - sh * a.sc_w) + a.pw;
        } else {
          n = p / a.OHW; const int rem = p - n * a.OHW; oh = rem / a.OW; ow = rem - oh * a.OW;
        }
        if (KIND == 0) { rb_h[i] = oh * a.stride - a.pad; rb_w[i] = ow * a.stride - a.pad; }
        else           { rb_h[i] = oh + a.pad;            rb_w[i] = ow + a.pad; }
        rowptr[i] = a.x + (size_t)n * a.H * a.W * a.Cin * sizeof(T) + xs_;
      }
    }
  }
  const char* wsrc[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int row = lrow + i * 32;
    wsrc[i] = a.w + ((size_t)(n0 + row) * a.Ktot + (pslot ^ swz(row)) * EPS) * sizeof(T);
  }

  // Ring-less variants (K <= 256: the wide 1x1 convs and the hand-over data gradients) are epilogue-bound: their residual /
  // mask / statistic operands (up to three tensors as large as the output) used to be requested only after the K loop, with the
  // whole HBM latency exposed once per block.  Request them NOW: they travel while the K stages are DMA-ed and multiplied.
  // (Older loads retire first, so the counted vmcnt waits of the K loop are unaffected.)
  // (r4: also the 2-deep form of the 128 x 64 tile for the hand-over class: TINYFACES_HANDOVER_TILE=42)
  constexpr bool PREF = (NS == 1 && KIND == 1) || (NS == 2 && KIND == 1 && BM == 128 && BN == 64 && MMA == 16 && EPIC == (TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3));
  // r4, hand-over instantiation (RES | MASK2 | STATS3 known at compile time): two of its three epilogue operands are COLD -- aux2 (the
  // previous block's output y) and aux3 (its conv3 output) were written in the forward pass -- and the third, the residual gradient, was
  // written a few launches ago.  It has the registers to request all three up front since its flag set is a compile-time constant (152 of
  // the 168 a 3-wave bound allows; +0.5 % on the step against a late residual).  The requests are issued BEHIND the DMAs of the first K
  // stage and that stage is waited for with vmcnt(NPF): the K loop starts after an L2 latency, not after the HBM latency of the cold
  // operands (loads retire in order, so every later wait covers them).  For the count to be exact the requests are unconditional
  // (clamped addresses, the zero is selected afterwards).
  constexpr bool PF_COLD3 = PREF && EPIC >= 0 && (EPIC & TF_EPI_STATS3) && (EPIC & TF_EPI_RES) && !(EPIC & (TF_EPI_MASK | TF_EPI_STATS2 | TF_EPI_JOIN));
  constexpr int P_CPR = BN / EPS, P_RPP = 256 / P_CPR, P_PASSES = BM / P_RPP;
  constexpr int NPF = PF_COLD3 ? 3 * P_PASSES : 0;
  uint4 pf1[P_PASSES], pf2[P_PASSES];      // (the third operand, JOIN's aux3, stays a late load: registers)
  uint4 pf3[PF_COLD3 ? P_PASSES : 1];
  auto prefetch = [&]() {
    const int pchunk = tid % P_CPR, prl = tid / P_CPR, pc0 = n0 + pchunk * EPS;
    const bool w1 = epi_flags & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2), w2 = epi_flags & (TF_EPI_JOIN | TF_EPI_MASK2);
#pragma unroll
    for (int ps = 0; ps < P_PASSES; ++ps) {
      const int p = m0 + prl + ps * P_RPP;
      const bool ok = p < a.M && pc0 < a.ldy;
      const size_t o = (orow(ok ? p : 0) * a.ldy + (ok ? pc0 : 0)) * sizeof(T);
      if constexpr (PF_COLD3) {
        const uint4 t1 = *reinterpret_cast<const uint4*>(a.aux3 + o), t2 = *reinterpret_cast<const uint4*>(a.aux2 + o), t3 = *reinterpret_cast<const uint4*>(a.aux + o);
        const uint4 z = make_uint4(0, 0, 0, 0);
        pf1[ps] = ok ? t1 : z; pf2[ps] = ok ? t2 : z; pf3[ps] = ok ? t3 : z;
      } else {
        pf1[ps] = (ok && w1) ? *reinterpret_cast<const uint4*>(a.aux + o) : make_uint4(0, 0, 0, 0);
        pf2[ps] = (ok && w2) ? *reinterpret_cast<const uint4*>(a.aux2 + o) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  const bool pf_early = PF_COLD3 && (a.dbg & 32);    // TF_CONV_DBG=32 (A/B): requests in front of the first DMAs, first stage waited with vmcnt(0)
  if constexpr (PREF && !PF_COLD3) prefetch();
  if constexpr (PF_COLD3) { if (pf_early) prefetch(); }

  // stage iterator (scalar): stages are issued in order, so (slot, chunk, kw, kh) advance incrementally
  const bool par = KIND == 2 && a.scat == 2;       // parity class: only the taps kh0 + 2i, kw0 + 2j exist for these output pixels
  const int tstep = par ? 2 : 1;
  int is_slot = 0, is_c = 0, is_kw = par ? a.kw0 : 0, is_kh = par ? a.kh0 : 0;
  auto issue = [&]() {
    char* xs = smem + is_slot * BUF;
    char* ws = xs + XBYTES;
    if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < XR; ++i) { dma16(rowptr[i], xs + i * 4096 + wave_byte); rowptr[i] += rstep[i]; }
    } else {
      const int cin_b = is_c * KCH * (int)sizeof(T);
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        int ih, iw; bool ok;
        if (KIND == 0) {
          ih = rb_h[i] + is_kh; iw = rb_w[i] + is_kw;
          ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        } else {
          const int th = rb_h[i] - is_kh, tw = rb_w[i] - is_kw, smask = (1 << a.sshift) - 1;
          ih = th >> a.sshift; iw = tw >> a.sshift;
          ok = (th | tw) >= 0 && !((th | tw) & smask) && ih < a.H && iw < a.W;
        }
        const int pix = ok ? ih * a.W + iw : 0;
        const uintptr_t real = reinterpret_cast<uintptr_t>(rowptr[i]) + (size_t)pix * a.Cin * sizeof(T) + cin_b;
        const uintptr_t src = ok ? real : reinterpret_cast<uintptr_t>(zero);
        dma16(reinterpret_cast<const void*>(src), xs + i * 4096 + wave_byte);
      }
    }
    if (par) {                                     // the weight slab of (tap, chunk): taps are not consecutive in K
      const size_t wo = ((size_t)(is_kh * a.KW + is_kw) * a.Cin + (size_t)is_c * KCH) * sizeof(T);
#pragma unroll
      for (int i = 0; i < WR; ++i) dma16(wsrc[i] + wo, ws + i * 4096 + wave_byte);
    } else {
#pragma unroll
      for (int i = 0; i < WR; ++i) { dma16(wsrc[i], ws + i * 4096 + wave_byte); wsrc[i] += KCH * sizeof(T); }
    }
    if (++is_slot == NS) is_slot = 0;
    if (KIND != 1 && ++is_c == a.cpt) { is_c = 0; is_kw += tstep; if (is_kw >= a.KW) { is_kw = par ? a.kw0 : 0; is_kh += tstep; } }
  };

  typedef typename std::conditional<MMA == 32, f32x16, f32x4>::type acc_t;
  acc_t acc[NF][MF];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < MF; ++m) acc[n][m] = acc_t(0.f);
  const int wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  auto compute = [&](const char* xs) {
    if constexpr (MMA == 32) stage32<T, NF, MF>(xs, xs + XBYTES, wm * WM, wn * WN, acc);
    else MmaD<T>::template stage<NF, MF>(xs, xs + XBYTES, wm * WM, wn * WN, acc);
  };

  const int nst = a.nstages;
  if constexpr (NS == 1) {
    // r3: the BatchNorm + ReLU in front of this conv (tf_conv_args.bnf), applied to the pixel tile IN LDS after its DMA landed: the
    // thread that requested a 16-byte piece reads it back, activates its 8 channels and stores it again (and, for the first channel
    // tile, to bnf_out: the weight gradient's operand), one more barrier per stage.  This kernel waits for every stage's DMA anyway
    // (ring-less, 1-4 stages, latency-bound with 4 blocks per CU), so the fix-up rides in time other blocks spend waiting, and the
    // separate bn_relu launch (one per bottleneck on the forward chain) disappears.  Coefficients: bn_fused.hip fwd_table, all Cin <= 256
    // channels per block, in a 2 KiB table behind the staging tile; block 0 publishes scale / shift / mean / invstd + running statistics.
    constexpr bool FIX = TF_EXP && KIND == 1 && sizeof(T) == 2 && EPIC < 0;      // (the specialised instantiations are never launched with a prologue)
    constexpr int TAB_AT = (BUF > BM * (BN + 4) * 4 ? BUF : BM * (BN + 4) * 4);
    float* ctab = reinterpret_cast<float*>(smem + TAB_AT);
    if constexpr (FIX) {
      if (a.pro) {
        const bool writer = logical == 0;
        for (int c = tid; c < a.Cin; c += 256) {
          double s = 0.0, q = 0.0;
          for (int r = 0; r < a.pf_rows; ++r) { s += (double)a.pf_stat[(size_t)(r * 2) * a.Cin + c]; q += (double)a.pf_stat[(size_t)(r * 2 + 1) * a.Cin + c]; }
          const double m0s = a.pf_sshift ? (double)a.pf_sshift[c] : 0.0;
          const double dm = s / a.pf_count, mean = m0s + dm;
          double var = q / a.pf_count - dm * dm;
          if (var < 0.0) var = 0.0;
          const float invstd = (float)(1.0 / sqrt(var + (double)a.pf_eps));
          const float sc = a.pf_gamma[c] * invstd, sh = a.pf_beta[c] - (float)mean * sc;
          ctab[c] = sc; ctab[256 + c] = sh;
          if (writer) {
            a.pf_scale[c] = sc; a.pf_shift[c] = sh; a.pf_mean[c] = (float)mean; a.pf_invstd[c] = invstd;
            if (a.pf_rmean) {
              const double unbiased = a.pf_count > 1.f ? var * a.pf_count / (a.pf_count - 1.0) : var;
              a.pf_rmean[c] = (1.f - a.pf_mom) * a.pf_rmean[c] + a.pf_mom * (float)mean;
              a.pf_rvar[c] = (1.f - a.pf_mom) * a.pf_rvar[c] + a.pf_mom * (float)unbiased;
            }
          }
        }
      }
    }
    for (int st = 0; st < nst; ++st) {
      if (st) __builtin_amdgcn_s_barrier();         // everyone finished reading the previous stage
      issue();
      if constexpr (PF_COLD3) {
        if (st == 0 && !pf_early) { prefetch(); wait_vmcnt<NPF>(); } else wait_vmcnt<0>();
      } else wait_vmcnt<0>();
      // (r3: __syncthreads also publishes the coefficient table written above; it implies vmcnt(0), which the specialised instantiations
      //  -- never launched with a prologue -- must not pay: the raw barrier behind the counted wait is the ring variants' protocol)
      if constexpr (EPIC >= 0) __builtin_amdgcn_s_barrier(); else __syncthreads();
      if constexpr (FIX) {
        if (a.pro) {
#pragma unroll
          for (int i = 0; i < XR; ++i) {
            const int row = lrow + i * 32, p = m0 + row;
            if (p < a.M) {
              char* piece = smem + row * 128 + pslot * 16;
              const int k0 = st * 64 + ((pslot ^ swz(row)) << 3);
              float f[8];
              tf::unpack16<T>(*reinterpret_cast<const uint4*>(piece), f);
#pragma unroll
              for (int j = 0; j < 8; j += 4) {
                const f32x4 vs = *reinterpret_cast<const f32x4*>(ctab + k0 + j), vh = *reinterpret_cast<const f32x4*>(ctab + 256 + k0 + j);
#pragma unroll
                for (int q = 0; q < 4; ++q) f[j + q] = fmaxf(f[j + q] * vs[q] + vh[q], 0.f);
              }
              const uint4 o = tf::pack16<T>(f);
              *reinterpret_cast<uint4*>(piece) = o;
              if (a.pf_out && nt == 0) *reinterpret_cast<uint4*>(a.pf_out + ((size_t)p * a.Cin + k0) * sizeof(T)) = o;
            }
          }
          __syncthreads();
        }
      }
      compute(smem);
    }
  } else {
#pragma unroll
  for (int j = 0; j < NS - 1; ++j)
    if (j < nst && !(a.dbg & 8)) issue();
  if constexpr (PF_COLD3) { if (!pf_early) prefetch(); }      // (NS == 2: behind the DMAs of stage 0)
  int cs = 0;                                        // ring slot being computed
  for (int st = 0; st < nst; ++st) {
    // stage st has landed once at most min(NS-2, nst-1-st) younger stages are still in flight
    const int younger = nst - 1 - st;
    if (PF_COLD3 && st == 0 && !pf_early) wait_vmcnt<NPF>();
    else if (younger >= NS - 2) wait_vmcnt<L*(NS - 2)>();
    else if (NS > 3 && younger == 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                   // everyone's pieces of stage st landed; ring slot (st-1)%NS is free
    if (st + NS - 1 < nst && !(a.dbg & 1)) issue();
    const char* xs = smem + cs * BUF;
    if (!(a.dbg & 2)) compute(xs);
    if (++cs == NS) cs = 0;
  }
  }
  __builtin_amdgcn_s_barrier();                     // all waves done reading the ring -> reuse it as the staging tile
  if (a.dbg & 4) { if (acc[0][0][0] == 123.456f) a.y[0] = 1; return; }

  // ---------------- epilogue, phase 1: accumulators -> fp32 [BM][BN+4] tile in LDS
  constexpr int PITCH = BN + 4;
  float* stg = reinterpret_cast<float*>(smem);
  if constexpr (MMA == 32) {
    // 32x32 accumulator: lane l holds pixel l & 31, channels 8*g + 4*(l >> 5) + {0..3} for g = 0..3 (registers 4g .. 4g+3)
    const int l = tid & 63, pr = l & 31, h = l >> 5;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(stg + (wm * WM + m * 32 + pr) * PITCH + wn * WN + n * 32 + g * 8 + h * 4) =
              f32x4{acc[n][m][4 * g], acc[n][m][4 * g + 1], acc[n][m][4 * g + 2], acc[n][m][4 * g + 3]};
  } else {
    const int l = tid & 63, pr = l & 15, g = l >> 4;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int m = 0; m < MF; ++m)
        *reinterpret_cast<f32x4*>(stg + (wm * WM + m * 16 + pr) * PITCH + wn * WN + n * 16 + g * 4) = acc[n][m];
  }
  __syncthreads();

  // ---------------- phase 2: one 16-byte output chunk (EPS consecutive channels) of one pixel row per thread
  constexpr int CPR = BN / EPS, RPP = 256 / CPR, PASSES = BM / RPP;
  const int chunk = tid % CPR, rlane = tid / CPR;
  const int c0 = n0 + chunk * EPS;
  const bool cok = c0 < a.ldy;
  float es[EPS], eh[EPS], ms[EPS], mh[EPS], s1[EPS], s2[EPS], sft[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { es[j] = 1.f; eh[j] = 0.f; ms[j] = 0.f; mh[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; sft[j] = 0.f; }
  if (cok) {
    if ((epi_flags & TF_EPI_STATS) && a.stat_shift) {    // sums of (x - shift), (x - shift)^2: see tf_conv_args.stat_shift
#pragma unroll
      for (int j = 0; j < EPS; ++j) sft[j] = a.stat_shift[c0 + j];
    }
    if (epi_flags & TF_EPI_AFFINE) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { es[j] = a.epi_scale[c0 + j]; eh[j] = a.epi_shift[c0 + j]; }
    }
    if (epi_flags & TF_EPI_MASK) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { ms[j] = a.mask_scale[c0 + j]; mh[j] = a.mask_shift[c0 + j]; }
    }
  }
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int row = rlane + ps * RPP;
    const int p = m0 + row;
    float v[EPS];
#pragma unroll
    for (int j = 0; j < EPS; j += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(stg + row * PITCH + chunk * EPS + j);
      v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
    }
    if ((epi_flags & TF_EPI_STATS) && p < a.M) {        // (rows >= M hold exact zeros, but not after the shift)
#pragma unroll
      for (int j = 0; j < EPS; ++j) { const float t = v[j] - sft[j]; s1[j] += t; s2[j] += t * t; }
    }
    if (p < a.M && cok) {
      const size_t o = (orow(p) * a.ldy + c0) * sizeof(T);
      float ax[EPS];
      if (epi_flags & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) {
        if constexpr (PF_COLD3) tf::unpack16<T>(pf3[PF_COLD3 ? ps : 0], ax);
        else if constexpr (PREF) tf::unpack16<T>(pf1[ps], ax);
        else tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux + o), ax);
      }
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
        if constexpr (PREF) tf::unpack16<T>(pf2[ps], y2); else tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), g3);
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] += (y2[j] > 0.f) ? g3[j] : 0.f;
      }
      if (epi_flags & TF_EPI_MASK2) {
        float y2[EPS];
        if constexpr (PREF) tf::unpack16<T>(pf2[ps], y2); else tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
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
        if constexpr (PF_COLD3) tf::unpack16<T>(pf1[ps], x3); else tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), x3);
#pragma unroll
        for (int j = 0; j < EPS; ++j) {
          // in place (r6): the launch that wrote the raster already counted ax = its (masked) value here; this launch adds what IT adds.
          // (ax is zero wherever the mask is: v - ax == mask ? acc : 0)
          const float d = a.inplace ? v[j] - ax[j] : v[j];
          s1[j] += d; s2[j] += d * x3[j];
        }
      }
      *reinterpret_cast<uint4*>(a.y + o) = tf::pack16<T>(v);
    }
  }
  if (epi_flags & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) {       // block-uniform: column sums of the tile
    // lanes sharing a chunk inside a wave differ in the lane bits >= log2(CPR)
    if (a.dbg & 16) {                                // A/B knob (TF_CONV_DBG=16): the ds_bpermute form
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
      }
    } else {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { s1[j] = tf::lane_group_sum<CPR>(s1[j]); s2[j] = tf::lane_group_sum<CPR>(s2[j]); }     // DPP / row swaps, no LDS crossbar
    }
    __syncthreads();                                 // staging tile fully consumed
    float* red = reinterpret_cast<float*>(smem);     // [4 waves][2][BN]
    const int lane = tid & 63;
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { red[(wave * 2 + 0) * BN + lane * EPS + j] = s1[j]; red[(wave * 2 + 1) * BN + lane * EPS + j] = s2[j]; }
    }
    __syncthreads();
    if ((epi_flags & TF_EPI_STATS) && a.stat_shift && a.stat_shift_out && mt == 0) {   // the shift this launch used, once per channel
      for (int cl = tid; cl < BN; cl += 256)
        if (n0 + cl < a.ldy) a.stat_shift_out[n0 + cl] = a.stat_shift[n0 + cl];
    }
    for (int e = tid; e < 2 * BN; e += 256) {
      const int k = e / BN, cl = e - k * BN, c = n0 + cl;
      if (c < a.ldy) {
        // at most TF_STAT_ROWS partial rows per launch: tile mt accumulates into row mt % TF_STAT_ROWS (fp32 atomics;
        // rows start at zero: the finalize kernels clear what they consumed), so the finalize reads 64 rows, not thousands
        const float v = red[(0 * 2 + k) * BN + cl] + red[(1 * 2 + k) * BN + cl] + red[(2 * 2 + k) * BN + cl] + red[(3 * 2 + k) * BN + cl];
        // (several launches of a parity-decomposed gradient fold into the same rows: always accumulate there)
        if (a.mtiles > a.srows || a.scat) atomicAdd(&a.stat_out[((size_t)(mt % a.srows) * 2 + k) * a.ldy + c], v);
        else a.stat_out[((size_t)mt * 2 + k) * a.ldy + c] = v;
      }
    }
  }
}

// pcls: -1 = the whole launch; 0..3 = parity class (ph = pcls >> 1, pw = pcls & 1) of a 3x3 / stride-2 / pad-1 data gradient (KIND 2)
template <typename T, int BM, int BN, int NS, int KIND, int MMA = 16>
int launch_kind(const tf_conv_args* A, hipStream_t stream, int pcls = -1) {
  constexpr int KCH = MmaD<T>::KCH;
  DmaK k;
  k.x = (const char*)A->x; k.w = (const char*)A->w; k.y = (char*)A->y;
  k.epi_scale = A->epi_scale; k.epi_shift = A->epi_shift;
  k.aux = (const char*)A->aux; k.aux2 = (const char*)A->aux2; k.aux3 = (const char*)A->aux3;
  k.mask_scale = A->mask_scale; k.mask_shift = A->mask_shift; k.stat_out = A->stat_out;
  k.stat_shift = A->stat_shift; k.stat_shift_out = A->stat_shift_out;
  k.pro = 0; k.pf_rows = 0; k.pf_count = k.pf_eps = k.pf_mom = 0.f;
  k.pf_stat = k.pf_gamma = k.pf_beta = k.pf_sshift = nullptr; k.pf_scale = k.pf_shift = k.pf_mean = k.pf_invstd = k.pf_rmean = k.pf_rvar = nullptr; k.pf_out = nullptr;
  if (A->bnf && !TF_EXP) return TF_ERR_UNSUPPORTED;      // the in-LDS BatchNorm prologue is an experimental-build feature (common.h)
  if (A->bnf) {
    if (!(NS == 1 && KIND == 1 && sizeof(T) == 2) || A->Cin > 256) return TF_ERR_UNSUPPORTED;
    const tf_bn_fwd_desc* d = A->bnf;
    if (!d->stat || !d->gamma || !d->beta || !d->scale || !d->shift || !d->mean || !d->invstd || A->bnf_rows < 1) return TF_ERR_ARG;
    k.pro = 1; k.pf_rows = A->bnf_rows; k.pf_count = A->bnf_count; k.pf_eps = A->bnf_eps; k.pf_mom = A->bnf_momentum;
    k.pf_stat = d->stat; k.pf_gamma = d->gamma; k.pf_beta = d->beta; k.pf_sshift = d->stat_shift;
    k.pf_scale = d->scale; k.pf_shift = d->shift; k.pf_mean = d->mean; k.pf_invstd = d->invstd; k.pf_rmean = d->running_mean; k.pf_rvar = d->running_var;
    k.pf_out = (char*)A->bnf_out;
  }
  k.H = A->H; k.W = A->W; k.Cin = A->Cin; k.OH = A->OH; k.OW = A->OW; k.KW = A->KW; k.stride = A->stride; k.pad = A->pad;
  k.sshift = A->stride == 2 ? 1 : 0;
  k.M = A->N * A->OH * A->OW; k.OHW = A->OH * A->OW; k.ldy = A->ldy;
  k.scat = 0; k.sc_hw = k.sc_w = k.sc_OH = k.sc_OW = 1; k.ph = k.pw = k.kh0 = k.kw0 = 0;
  if (KIND == 1 && A->mode == 1 && A->stride == 2) {      // scattered pointwise data gradient (see launch()): rows = the GRADIENT raster
    k.scat = 1; k.M = A->N * A->H * A->W; k.sc_hw = A->H * A->W; k.sc_w = A->W; k.sc_OH = A->OH; k.sc_OW = A->OW;
  }
  k.inplace = (k.scat == 1 && (A->epi & TF_EPI_RES) && A->aux == (const void*)A->y) ? 1 : 0;
  k.cpt = A->Cin / KCH; k.Ktot = A->KH * A->KW * A->Cin; k.nstages = A->KH * A->KW * k.cpt;
  if (KIND == 2 && pcls >= 0) {                           // one parity class of the output raster (see launch())
    k.scat = 2; k.ph = pcls >> 1; k.pw = pcls & 1;
    const int soh = (A->OH - k.ph + 1) / 2, sow = (A->OW - k.pw + 1) / 2;
    k.sc_hw = soh * sow; k.sc_w = sow; k.sc_OH = A->OH; k.sc_OW = A->OW; k.M = A->N * soh * sow;
    // taps with (oh + pad - kh) even: kh = kh0, kh0 + 2, ...
    k.kh0 = ((k.ph + A->pad) & 1); k.kw0 = ((k.pw + A->pad) & 1);
    const int nkh = (A->KH - k.kh0 + 1) / 2, nkw = (A->KW - k.kw0 + 1) / 2;
    k.nstages = nkh * nkw * k.cpt;
  }
  k.ntiles = (A->Cout + BN - 1) / BN; k.mode = A->mode; k.epi = A->epi;
  const int mtiles = (k.M + BM - 1) / BM;
  k.mtiles = mtiles; k.srows = tf_get_stat_rows();
  k.dbg = tf::tuning().conv_dbg;
  size_t lds = (size_t)NS * (BM + BN) * 128;
  const size_t stg = (size_t)BM * (BN + 4) * 4;
  if (stg > lds) lds = stg;
  if (k.pro) lds += 2048;                            // scale / shift of up to 256 input channels behind the staging tile
  static tf::PerDevice attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<T, BM, BN, NS, KIND, MMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  {
    const double es = sizeof(T), M = k.M, Kt = k.Ktot;
    double bytes = ((double)A->N * A->H * A->W * A->Cin + (double)A->Cout * Kt + (double)A->N * A->OH * A->OW * A->Cout) * es;
    if (A->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) bytes += M * A->Cout * es;
    if (A->epi & TF_EPI_JOIN) bytes += 2 * M * A->Cout * es;
    if (A->epi & TF_EPI_MASK2) bytes += M * A->Cout * es;
    if (A->epi & TF_EPI_STATS3) bytes += M * A->Cout * es;
    // algorithmic work = 2 x the MACs of the FORWARD convolution this launch belongs to, on unpadded channels:
    // a data gradient (mode 1) has as many MACs as its forward conv -- one per (forward output pixel, tap, cin, cout) -- although a
    // stride-2 one executes the zero-inserted gather over its 4x larger output raster; padded K / N (stem, heads) count as what they hold
    const double alg_k = A->alg_k > 0 ? A->alg_k : Kt, alg_n = A->alg_n > 0 ? A->alg_n : A->Cout;
    const double alg_m = A->mode == 1 ? (double)A->N * A->H * A->W : M;
    // executed = the GEMM this launch runs (M rows x Cout x the K it walks); a parity-class launch walks only the taps that exist for its
    // pixels, so its executed work IS algorithmic work (the four classes add up to the forward conv's MACs, borders aside)
    const double exec_fl = 2.0 * M * A->Cout * (double)k.nstages * KCH;
    const double alg_fl = pcls >= 0 ? exec_fl : 2.0 * alg_m * alg_n * alg_k;
    if (pcls >= 0) bytes = ((double)A->N * A->H * A->W * A->Cin + (double)A->Cout * Kt) * es / 4 + M * A->Cout * es * (1 + ((A->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) ? 1 : 0));
    tf::ProfScope prof(A->dtype == TF_F32 ? 12 : (A->dtype == TF_BF16 ? 13 : 15), alg_fl, bytes, stream, k.M, A->Cout, k.Ktot,
                       A->KH * A->KW, A->mode, A->epi, exec_fl, true);   // 12 = conv_dma f32, 13 = conv_dma bf16, 15 = conv_dma f16
    if (k.scat == 1 && !k.inplace) {
      prof.begin_bracket();                          // the initialisation of the raster is part of this launch's cost
      // the three other parities of the output raster: zero, or the residual operand itself (y = 0 + aux there)
      const size_t ybytes = (size_t)A->N * A->OH * A->OW * A->ldy * sizeof(T);
      const hipError_t e = (A->epi & TF_EPI_RES) ? hipMemcpyAsync(A->y, A->aux, ybytes, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(A->y, 0, ybytes, stream);
      if (e != hipSuccess) return TF_ERR_LAUNCH;
    }
    // r4: the three flag sets of a training step get instantiations of their own for the hot pointwise bf16 tiles (EPIC, see the kernel)
    // which (tile, ring, fragment) combinations the executor dispatches (pick_tile, csrc/conv.hip): 64 x 128 / 2-deep / 32x32x16 fragments,
    // the ring-less 128 x 64 tile, 64 x 64 with 1-3 slots -- the other instantiations exist for explicit tile requests only
    constexpr bool SPEC = sizeof(T) == 2 && ((MMA == 32 && BM == 64 && BN == 128 && NS == 2) || (MMA == 16 && BM == 128 && BN == 64 && NS == 1 && KIND != 2) || (MMA == 16 && BM == 128 && BN == 64 && NS == 2 && KIND == 1) ||
                                             (MMA == 16 && BM == 64 && BN == 64 && NS <= 3));
    constexpr bool TRAIN = std::is_same<T, tf::bf16_t>::value;          // the training flag sets: bf16 only (fp16 is inference only)
    const bool spec_off = tf::tuning().epi_spec_off;       // A/B knob
    const dim3 grid(mtiles * k.ntiles);
    bool done = false;
    if constexpr (SPEC) {
      if (!spec_off && !k.pro) {
        auto go = [&](auto epic) {
          constexpr int E = decltype(epic)::value;
          static tf::PerDevice set;
          if (set.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<T, BM, BN, NS, KIND, MMA, E>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          TF_LAUNCH_TIMED((conv_dma_kernel<T, BM, BN, NS, KIND, MMA, E>), grid, dim3(256), lds, stream, k);
          done = true;
        };
        if constexpr (TRAIN) {
          if (A->epi == TF_EPI_STATS) go(std::integral_constant<int, TF_EPI_STATS>{});
          else if (A->epi == (TF_EPI_MASK | TF_EPI_STATS2)) go(std::integral_constant<int, TF_EPI_MASK | TF_EPI_STATS2>{});
          else if (KIND == 1 && A->epi == (TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3)) go(std::integral_constant<int, TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3>{});
          else if (KIND == 1 && A->epi == (TF_EPI_MASK2 | TF_EPI_STATS3)) go(std::integral_constant<int, TF_EPI_MASK2 | TF_EPI_STATS3>{});      // r6: the hand-over of layer2.0 (its residual arrives in place, behind it)
        }
        if (!done && KIND != 2) {                      // the folded-BN epilogues of the evaluation graph (forward convs only)
          if (A->epi == (TF_EPI_AFFINE | TF_EPI_RELU)) go(std::integral_constant<int, TF_EPI_AFFINE | TF_EPI_RELU>{});
          else if (KIND == 1 && A->epi == (TF_EPI_AFFINE | TF_EPI_RES | TF_EPI_RELU)) go(std::integral_constant<int, TF_EPI_AFFINE | TF_EPI_RES | TF_EPI_RELU>{});
        }
      }
    }
    if (!done) TF_LAUNCH_TIMED((conv_dma_kernel<T, BM, BN, NS, KIND, MMA>), grid, dim3(256), lds, stream, k);
  }
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

template <typename T, int BM, int BN, int NS, int MMA = 16>
int launch(const tf_conv_args* A, hipStream_t stream) {
  const bool pointwise = A->KH == 1 && A->KW == 1 && A->stride == 1 && A->pad == 0 && A->H == A->OH && A->W == A->OW;
  if (pointwise) return launch_kind<T, BM, BN, NS, 1, MMA>(A, stream);
  // r3: the data gradient of a 1x1 / stride-2 / pad-0 conv (the downsample branches of layer2.0 and layer3.0) is the pointwise GEMM over the
  // gradient's own pixels, scattered to the even-even positions of the 2x larger output raster (the rest is zero, or the residual
  // operand): the pointwise kernel with a row map instead of the transposed gather over all four parities.  Epilogues that reduce over
  // the output raster (statistics) or read a mask there keep the generic kernel; TINYFACES_SCATTER_DGRAD_OFF=1: A/B knob.
  const bool scat_off = tf::tuning().scatter_dgrad_off;
  if (!scat_off && A->mode == 1 && A->KH == 1 && A->KW == 1 && A->stride == 2 && A->pad == 0 && A->OH >= 2 * A->H - 1 && A->OW >= 2 * A->W - 1 &&
      A->ldy == A->Cout &&
      (!(A->epi & ~(TF_EPI_RES | TF_EPI_AFFINE)) ||
       // r6: IN PLACE (aux == y): the raster already holds another launch's result and the scattered rows are added to it -- then the ReLU mask
       // and the BN-backward sums of a hand-over may ride along: they see exactly the rows this launch changes (the executor's stride-2
       // downsample gradients: no zero fill / residual copy of the 2x larger raster, no second read of it by the hand-over)
       ((A->epi & TF_EPI_RES) && A->aux == (const void*)A->y && !(A->epi & ~(TF_EPI_RES | TF_EPI_MASK2 | TF_EPI_STATS3)))))
    return launch_kind<T, BM, BN, NS, 1, MMA>(A, stream);
  // r3: the data gradient of a 3x3 / stride-2 / pad-1 conv (conv2 of layer2.0 / layer3.0), by PARITY CLASS of the output pixel.  An output
  // pixel (ih, iw) only receives the taps with (ih + 1 - kh) and (iw + 1 - kw) even: 1, 2, 2 or 4 of the 9; the generic transposed gather
  // walks all 9 for every pixel and reads the zero page for the rest (4x the stages, DMAs and MFMAs: 134 and 113 us per launch at
  // bs = 12, 8-13x over their roofline, profiles/r02_layer_table.md).  Four launches, each a gather over its class's half-resolution
  // raster with its own tap list; rows scatter back to (2h + ph, 2w + pw); statistic sums fold into the same rows by atomics.
  const bool par_off = tf::tuning().parity_dgrad_off;
  if (!par_off && A->mode == 1 && A->stride == 2 && A->KH == 3 && A->KW == 3 && A->pad == 1 && A->OH >= 2 && A->OW >= 2 &&
      (tf_get_stat_rows() <= TF_STAT_ROWS || !(A->epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)))) {
    for (int pc = 0; pc < 4; ++pc) {
      const int rc = launch_kind<T, BM, BN, NS, 2, MMA>(A, stream, pc);
      if (rc != TF_OK) return rc;
    }
    return TF_OK;
  }
  return A->mode == 0 ? launch_kind<T, BM, BN, NS, 0, MMA>(A, stream) : launch_kind<T, BM, BN, NS, 2, MMA>(A, stream);
}

// the 2-byte operand types (bf16, fp16) share every tile / ring-depth decision
template <typename T>
int launch_half(const tf_conv_args* a, int tile, int depth, hipStream_t stream) {
  // tiles 4..6: 32x32x16 fragments (64 x 64 / 64 x 32 / 32 x 64 wave tiles), ring depth 1..3
  if (tile == 4) return depth == 1 ? launch<T, 128, 128, 1, 32>(a, stream) : depth == 2 ? launch<T, 128, 128, 2, 32>(a, stream) : launch<T, 128, 128, 3, 32>(a, stream);
  if (tile == 5) return depth == 1 ? launch<T, 128, 64, 1, 32>(a, stream) : depth == 2 ? launch<T, 128, 64, 2, 32>(a, stream) : launch<T, 128, 64, 3, 32>(a, stream);
  if (tile == 6) return depth == 1 ? launch<T, 64, 128, 1, 32>(a, stream) : depth == 2 ? launch<T, 64, 128, 2, 32>(a, stream) : launch<T, 64, 128, 3, 32>(a, stream);
  if (tile == 1) return launch<T, 128, 128, 3>(a, stream);
  if (tile == 2) {
    if (depth == 1) return launch<T, 128, 64, 1>(a, stream);       // tile code 32: ring-less, short K (see pick_tile)
    if (depth == 2) return launch<T, 128, 64, 2>(a, stream);       // tile code 42 (r4: A/B form of the hand-over data gradient)
    return depth == 4 ? launch<T, 128, 64, 4>(a, stream) : launch<T, 128, 64, 3>(a, stream);
  }
  if (depth == 3) {
    // convs of up to 16 K-stages (every 1x1 of the trunk, K <= 1024) are dispatch + prologue + epilogue bound rather than
    // K-loop bound: a 2-deep ring is 32 KiB of LDS, so five blocks fit a CU instead of three and more of those phases
    // overlap.  A/B on one box, img/s: 956 (3-deep everywhere), 989 (<= 4 stages), 996 (<= 8), 1008 (<= 16), 1001 (all).
    const int ns2_max = tf::tuning().ns2_maxstages;
    const int nst = a->KH * a->KW * (a->Cin / 64);
    // ... and no ring at all up to 4 stages (17 KiB of LDS, 8 blocks/CU): 1007 -> 1013 img/s
    const int ns1_max = tf::tuning().ns1_maxstages;
    if (nst <= ns1_max) return launch<T, 64, 64, 1>(a, stream);
    if (nst <= ns2_max) return launch<T, 64, 64, 2>(a, stream);
    return launch<T, 64, 64, 3>(a, stream);
  }
  return launch<T, 64, 64, 4>(a, stream);
}

}  // namespace
