// conv_pwx (r5 rewrite): PIXEL-STREAMING pointwise conv / data gradient for narrow outputs (Cout = 128 or 256) and a long reduction
// (K = 128 ... 1024) whose pixel operand is TRANSFORMED on its way to the MFMAs -- the BatchNorm passes of a training step ride on the
// operand path of the conv that consumes them instead of being launches (and HBM round trips) of their own:
//
//   PRO 0   x                                                     plain pointwise conv (A/B knob, C ABI)
//   PRO 2   x' = A*g + B*c + D            side output x'          tf_bn_bwd_apply_fused + the data gradient of conv3     (tf_conv2d_bnbwd)
//   PRO 3   x' = relu(S*c + H + r)        side output x' (= y)    tf_bn_add_relu_fused (identity residual) + conv1 of the NEXT bottleneck
//   PRO 4   x' = relu(S*c + H + S2*r + H2)  side output x'        the same behind a downsample bottleneck (residual = bn_d(conv_d))   (tf_conv2d_bnfwd)
//
// (torchvision Bottleneck: bn3 -> += identity -> relu -> conv1 of the next block, and their autograd backward; tinyfaces/models/model.py:90-101.)
//
// Why a rewrite.  Round 3's kernel staged the pixel operand global -> VGPR -> LDS ONE stage ahead.  One block owns 64 pixels x all output
// channels, so M = 12 288 (layer 3 at bs = 12) is 192 blocks on 256 CUs with 16 KB of HBM reads in flight each: by Little's law
// (192 x 16 KB / ~1.5 us) that is ~2 TB/s, and the SQ pass of r3 says the same (profiles/r03_pmc_sq_pointwise.txt: 30.8 us for 81 MB,
// waves waiting 59 % of their cycles, matrix pipes 8 % busy).  The fused form lost to the two launches it replaced (31.6 vs 29.8 us) for
// that reason alone.  Here:
//   * the RAW operand tiles (g and c, or c and r: 2 x 8 KB per 64-deep stage) travel by LDS-DMA into a ring of NSX slots (5-6 stages =
//     64-80 KB of HBM reads in flight per block), issued by waves 4-7 ONLY; the weight stages (L2-resident, BN x 128 B) by waves 0-3
//     ONLY.  vmcnt is a per-wave counter and retires in order: with both streams in one wave the depth of the pixel ring would be
//     capped by the depth of the weight ring (a wait for weight stage s also waits for every pixel stage issued before it);
//   * a wave transforms exactly the 2 x 1 KB regions its OWN DMAs wrote (its own vmcnt is the only wait it needs): ds_read raw ->
//     fp32 math with the coefficients of the 8 channels of its k slot -> ds_write IN PLACE (the slot's first 8 KB become the MFMA
//     operand tile) + 16-byte global stores of the side output (8 lanes = one 128-byte line);
//   * ONE barrier per stage as before (it publishes the transformed tile of stage s+1 and frees the slots of stage s);
//   * the coefficient table (A, B, D / S, H of all K channels, bn_fused.hip arithmetic) is derived by the weight waves while the pixel
//     ring fills; block 0 publishes dgamma / dbeta, or scale / shift / mean / invstd + the running statistics.
// MFMA roles (8 waves = 4 channel groups x 2 K halves on 32x32x16, halves merged through fp32 staging tiles) and the epilogue (every
// TF_EPI_* flag, compile-time flag set EPIC for the executor's instantiations) are those of the r3 kernel / conv_dma.
// bf16 only (the fp32 parity path keeps the unfused kernels).
#include "common.h"
#if TF_EXP
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "tuning.h"
#include "lds_dma.h"
#include "profile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef tf::bf16_t T;

__device__ uint4 g_pwx_zero[8];

constexpr int BM = 64, NT = 512, EPS = 8;
constexpr int XOP = BM * 128;                     // 8 KiB: one operand of a pixel stage, 64 pixels x 64 k

struct FwdTab {                                   // tf_bn_fwd_desc
  const float* stat; const float* gamma; const float* beta;
  float* scale; float* shift; float* mean; float* invstd; float* rmean; float* rvar;
  const float* sshift;
};
struct PK {
  const char* x; const char* xc; const char* w; char* y; char* t1;
  const float* epi_scale; const float* epi_shift;
  const char* aux; const char* aux2; const char* aux3;
  const float* mask_scale; const float* mask_shift;
  float* stat_out; const float* stat_shift; float* stat_shift_out;
  const float* pstat; const float* pgamma; const float* pmean; const float* pinvstd; float* pdgamma; float* pdbeta;
  FwdTab f1, f2;
  float feps, fmom;
  int prows, pnk, pkidx; float pcount;
  int M, K, ldy, nst, ntiles, epi, srows, mtiles;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) << 2) ^ (((row >> 3) & 1) * 6); }
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ swz(row)) << 4); }
// Every DMA of this kernel is issued through inline asm (lds_dma.h): hipcc models the builtin as a pending LDS WRITE and puts
// `s_waitcnt vmcnt(0)` in front of the first ds_write that follows one -- the in-place transform below -- which drains the pixel ring
// in every stage (seen in the ISA of the first r5 build).  Hidden, the counted waits below are the only thing that orders LDS accesses
// behind a DMA.  For the same reason the barriers of the K loop are raw s_barrier behind an explicit lgkmcnt(0): __syncthreads() expands to
// `s_waitcnt vmcnt(0) lgkmcnt(0)` + s_barrier.
__device__ __forceinline__ void block_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The coefficient table is derived by 256 threads for up to 1024 channels: FOUR channels per thread and pass, with every load of the pass
// (4 channels x 8 rows x 2 statistic planes + the per-channel vectors) in flight together.  A rolled loop pays one L2 / MALL round trip per
// row and per vector -- ~16 dependent round trips per channel, which is what the r3 kernel's table cost.  Rows >= R are read from row R - 1
// and multiplied by zero, never out of bounds (bn_fused.hip fwd_table / bwd_table); channels >= K are clamped and not written.
constexpr int TCH = 4;
__device__ __forceinline__ void stat_sums4(const float* stat, int R, int nk, int ka, int kb, int K, const int (&c)[TCH], double (&sa)[TCH], double (&sb)[TCH]) {
#pragma unroll
  for (int i = 0; i < TCH; ++i) { sa[i] = 0.0; sb[i] = 0.0; }
  for (int r0 = 0; r0 < R; r0 += 8) {
    float av[TCH][8], bv[TCH][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int rr = r0 + r < R ? r0 + r : R - 1;
      const float keep = r0 + r < R ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < TCH; ++i) {
        av[i][r] = stat[(size_t)(rr * nk + ka) * K + c[i]] * keep;
        bv[i][r] = stat[(size_t)(rr * nk + kb) * K + c[i]] * keep;
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < TCH; ++i) { sa[i] += (double)av[i][r]; sb[i] += (double)bv[i][r]; }
  }
}

// scale / shift of a training-mode BatchNorm from its statistic rows: the arithmetic of bn_fused.hip fwd_table.  sc_out / sh_out: LDS [K]
__device__ __forceinline__ void fwd_table4(const FwdTab& d, int R, int K, const int (&c)[TCH], const bool (&ok)[TCH], float count, float eps, float mom,
                                           bool writer, float* sc_out, float* sh_out) {
  float ga[TCH], be[TCH], m0[TCH], rm[TCH], rv[TCH];
  // (unconditional loads through a substitute pointer: a branch per optional vector would put a full wait between the batches)
  const float* ssp = d.sshift ? d.sshift : d.gamma;
  const float* rmp = d.rmean ? d.rmean : d.gamma;
  const float* rvp = d.rmean ? d.rvar : d.gamma;
  const float has_shift = d.sshift ? 1.f : 0.f;
#pragma unroll
  for (int i = 0; i < TCH; ++i) {
    ga[i] = d.gamma[c[i]]; be[i] = d.beta[c[i]];
    m0[i] = ssp[c[i]] * has_shift;
    rm[i] = rmp[c[i]]; rv[i] = rvp[c[i]];
  }
  double s[TCH], q[TCH];
  stat_sums4(d.stat, R, 2, 0, 1, K, c, s, q);
#pragma unroll
  for (int i = 0; i < TCH; ++i) {
    if (!ok[i]) continue;
    const double dm = s[i] / count;
    const double mean = (double)m0[i] + dm;
    double var = q[i] / count - dm * dm;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = ga[i] * invstd;
    const float sh = be[i] - (float)mean * sc;
    sc_out[c[i]] = sc; sh_out[c[i]] = sh;
    if (writer) {
      d.scale[c[i]] = sc; d.shift[c[i]] = sh; d.mean[c[i]] = (float)mean; d.invstd[c[i]] = invstd;
      if (d.rmean) {
        const double unbiased = count > 1.f ? var * count / (count - 1.0) : var;
        d.rmean[c[i]] = (1.f - mom) * rm[i] + mom * (float)mean;
        d.rvar[c[i]] = (1.f - mom) * rv[i] + mom * (float)unbiased;
      }
    }
  }
}

template <int BN, int PRO, int NSW, int NSX, int EPIC>
__global__ void __launch_bounds__(NT) conv_pwx_kernel(const PK a) {
  constexpr int NOP = PRO ? 2 : 1;                // raw operands of a pixel stage
  constexpr int NCOEF = PRO == 2 ? 3 : (PRO == 3 ? 2 : (PRO == 4 ? 4 : 0));
  constexpr int WBUF = BN * 128, XSLOT = NOP * XOP;
  constexpr int X_AT = NSW * WBUF, COEF_AT = X_AT + NSX * XSLOT;
  constexpr int WPW = BN / 32;                    // weight DMAs per thread of waves 0-3 and stage (BN rows x 8 pieces / 256 threads)
  constexpr int XPW = 2 * NOP;                    // pixel DMAs per thread of waves 4-7 and stage
  constexpr int WCH = BN / 4, NF = WCH / 32;      // channels per wave, 32-channel fragments per wave
  constexpr int PITCH = BN + 4, STG = 32 * PITCH; // one K half of a 32-pixel pass, floats
  static_assert(NSW >= 2 && NSX >= 2, "ring depths");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int logical;
  {
    const int nb = gridDim.x, b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool is_w = wave_u < 4;                   // wave-uniform role: 0-3 stream weights, 4-7 stream (and transform) pixels
  const int role = wave_u & 3;
  const int nst = a.nst, K = a.K;
  const char* zero = reinterpret_cast<const char*>(g_pwx_zero) + (l & 7) * 16;

  // ---- DMA roles.  A wave-level DMA moves 64 x 16 B = 8 rows of a 128-byte-row tile: lane l -> row (l >> 3), physical slot (l & 7), which
  //      receives logical slot (l & 7) ^ swz(row) (the swizzle is applied on the source address)
  const char* wsrc[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int row = (i * 4 + role) * 8 + (l >> 3);
    wsrc[i] = a.w + (size_t)(n0 + row) * K * sizeof(T) + (((l & 7) ^ swz(row)) << 4);
  }
  const char* xsrc[NOP][2];
  int xstep[2];
  bool rvalid[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = role * 16 + j * 8 + (l >> 3);
    rvalid[j] = m0 + row < a.M;
    xstep[j] = rvalid[j] ? 128 : 0;
    const size_t o = ((size_t)(m0 + row) * K) * sizeof(T) + (((l & 7) ^ swz(row)) << 4);
    xsrc[0][j] = rvalid[j] ? a.x + o : zero;
    if constexpr (NOP == 2) xsrc[1][j] = rvalid[j] ? a.xc + o : zero;
  }
  const uint32_t smem_u = tf::lds_addr_uniform(smem);
  auto issue_w = [&](int stage, int slot) {       // waves 0-3
    const uint32_t dst = smem_u + slot * WBUF + role * 1024;
    const bool live = stage < nst;                // past the end: the same NUMBER of DMAs (zeros into a free slot) keeps every wait count a constant
#pragma unroll
    for (int i = 0; i < WPW; i += 2)
      tf::dma16_hidden2(live ? wsrc[i] + (size_t)stage * 128 : zero, dst + i * 4096, live ? wsrc[i + 1] + (size_t)stage * 128 : zero, dst + (i + 1) * 4096);
  };
  auto issue_x = [&](int stage, int slot) {       // waves 4-7
    const uint32_t dst = smem_u + X_AT + slot * XSLOT + role * 2048;
    const bool live = stage < nst;
#pragma unroll
    for (int op = 0; op < NOP; ++op)
      tf::dma16_hidden2(live ? xsrc[op][0] + (size_t)stage * xstep[0] : zero, dst + op * XOP, live ? xsrc[op][1] + (size_t)stage * xstep[1] : zero, dst + op * XOP + 1024);
  };
  if (is_w) {
#pragma unroll
    for (int st = 0; st < NSW - 1; ++st) issue_w(st, st);
  } else {
#pragma unroll
    for (int st = 0; st < NSX - 1; ++st) issue_x(st, st);
  }

  // ---- coefficient table, derived by the weight waves (their ordinary loads would drain the deep pixel ring of the other four)
  float* coef = reinterpret_cast<float*>(smem + COEF_AT);       // [NCOEF][K]
  if constexpr (PRO != 0) {
    if (is_w) {
      const bool writer = logical == 0;
      for (int cb = tid; cb < K; cb += 256 * TCH) {
        int c[TCH]; bool ok[TCH];
#pragma unroll
        for (int i = 0; i < TCH; ++i) { ok[i] = cb + i * 256 < K; c[i] = ok[i] ? cb + i * 256 : K - 1; }
        if constexpr (PRO == 2) {
          float mu_[TCH], is_[TCH], ga_[TCH];
#pragma unroll
          for (int i = 0; i < TCH; ++i) { mu_[i] = a.pmean[c[i]]; is_[i] = a.pinvstd[c[i]]; ga_[i] = a.pgamma[c[i]]; }
          double s1[TCH], s2[TCH];
          stat_sums4(a.pstat, a.prows, a.pnk, 0, a.pkidx, K, c, s1, s2);
#pragma unroll
          for (int i = 0; i < TCH; ++i) {
            if (!ok[i]) continue;
            const double mu = mu_[i], is = is_[i], ga = ga_[i];
            const double dg = (s2[i] - mu * s1[i]) * is;
            const double A = ga * is;
            coef[c[i]] = (float)A;
            coef[K + c[i]] = (float)(-A * is * dg / a.pcount);
            coef[2 * K + c[i]] = (float)(-A * s1[i] / a.pcount + A * mu * is * dg / a.pcount);
            if (writer) {
              if (a.pdgamma) a.pdgamma[c[i]] = (float)dg;
              if (a.pdbeta) a.pdbeta[c[i]] = (float)s1[i];
            }
          }
        } else {
          fwd_table4(a.f1, a.prows, K, c, ok, a.pcount, a.feps, a.fmom, writer, coef, coef + K);
          if constexpr (PRO == 4) fwd_table4(a.f2, a.prows, K, c, ok, a.pcount, a.feps, a.fmom, writer, coef + 2 * K, coef + 3 * K);
        }
      }
    }
  }

  // ---- the transform of ONE pixel stage by the wave that DMA-ed it: lane -> k slot ls = l & 7 of rows role*16 + (l >> 3) and + 8
  const int ls = l & 7;
  int toff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int row = role * 16 + j * 8 + (l >> 3); toff[j] = lds_off(row, ls); }
  char* t1dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = role * 16 + j * 8 + (l >> 3);
    t1dst[j] = (PRO != 0 && a.t1 && rvalid[j] && nt == 0) ? a.t1 + ((size_t)(m0 + row) * K + ls * 8) * sizeof(T) : nullptr;
  }
  auto transform = [&](int stage, int slot) {
    if constexpr (PRO != 0) {
      char* xb = smem + X_AT + slot * XSLOT;
      const int k0 = stage * 64 + ls * 8;
      float cf[NCOEF][EPS];
#pragma unroll
      for (int q = 0; q < NCOEF; ++q)
#pragma unroll
        for (int e = 0; e < EPS; e += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(coef + q * K + k0 + e);
          cf[q][e] = v[0]; cf[q][e + 1] = v[1]; cf[q][e + 2] = v[2]; cf[q][e + 3] = v[3];
        }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float p[EPS], q2[EPS];
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(xb + toff[j]), p);
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(xb + XOP + toff[j]), q2);
        if constexpr (PRO == 2) {
#pragma unroll
          for (int e = 0; e < EPS; ++e) p[e] = cf[0][e] * p[e] + cf[1][e] * q2[e] + cf[2][e];
        } else if constexpr (PRO == 3) {
#pragma unroll
          for (int e = 0; e < EPS; ++e) p[e] = fmaxf(p[e] * cf[0][e] + cf[1][e] + q2[e], 0.f);
        } else {
#pragma unroll
          for (int e = 0; e < EPS; ++e) p[e] = fmaxf(p[e] * cf[0][e] + cf[1][e] + (q2[e] * cf[2][e] + cf[3][e]), 0.f);
        }
        const uint4 out = tf::pack16<T>(p);
        *reinterpret_cast<uint4*>(xb + toff[j]) = out;
        if (t1dst[j]) *reinterpret_cast<uint4*>(t1dst[j] + (size_t)stage * 128) = out;
      }
    }
  };

  // pixel stage 0 has landed for its wave once only the NSX - 2 stages issued after it are outstanding (side-output stores are NOT counted:
  // a wait that is too strict by the stores in flight is safe, vmcnt retires in order)
  if (!is_w) wait_vmcnt<(NSX - 2) * XPW>();
  block_barrier_lds();                              // the coefficient table is complete
  if (!is_w) transform(0, 0);
  else wait_vmcnt<(NSW - 2) * WPW>();               // weight stage 0 (the table's loads were older: already waited for)

  // ---- MFMA roles: wave = (K half kg, channel group wn); 2 pixel fragments x NF channel fragments of 32 x 32
  const int kg = wave >> 2, wn = wave & 3;
  const int r32 = l & 31, h = l >> 5;
  int xo[2][2], wo[NF][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) xo[m][j] = lds_off(m * 32 + r32, (kg * 2 + j) * 2 + h);
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int j = 0; j < 2; ++j) wo[n][j] = lds_off(wn * WCH + n * 32 + r32, (kg * 2 + j) * 2 + h);
  f32x16 acc[NF][2];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[n][m] = f32x16(0.f);

  auto compute = [&](int wslot, int xslot) {
    const char* wb = smem + wslot * WBUF;
    const char* xb = smem + X_AT + xslot * XSLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 xf[2], wf[NF];
#pragma unroll
      for (int m = 0; m < 2; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(xb + xo[m][j]);
#pragma unroll
      for (int n = 0; n < NF; ++n) wf[n] = *reinterpret_cast<const bf16x8*>(wb + wo[n][j]);
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n], xf[m], acc[n][m], 0, 0, 0);
    }
  };
  block_barrier_lds();                              // tile 0 transformed, weight stage 0 landed (every wave waited for its own part)

  // ---- K loop.  Stage s lives in weight slot s % NSW and pixel slot s % NSX; the slots of stage s - 1 were released by the barrier that
  //      ended iteration s - 1 and receive stages s + NSW - 1 / s + NSX - 1 at the top of iteration s.
  int wslot = 0, xslot = 0;
  for (int s = 0; s < nst; ++s) {
    if (is_w) issue_w(s + NSW - 1, wslot == 0 ? NSW - 1 : wslot - 1);
    else issue_x(s + NSX - 1, xslot == 0 ? NSX - 1 : xslot - 1);
    compute(wslot, xslot);
    if (++wslot == NSW) wslot = 0;
    if (++xslot == NSX) xslot = 0;
    if (s + 1 < nst) {
      if (is_w) wait_vmcnt<(NSW - 2) * WPW>();      // weight stage s + 1: only the NSW - 2 younger stages may be outstanding
      else { wait_vmcnt<(NSX - 2) * XPW>(); transform(s + 1, xslot); }
      block_barrier_lds();                          // publishes tile s + 1, releases the slots of stage s
    }
  }
  wait_vmcnt<0>();                                  // dummy DMAs and side-output stores: nothing may be pending on the LDS reuse below
  __syncthreads();                                  // all waves done reading the rings -> reuse them as the staging tiles

  // ---------------- epilogue: two passes of 32 pixels; per pass the two K halves park their 32 x BN fp32 tiles, then every thread
  // handles 16 output bytes (8 channels) of one pixel: conv_dma's phase 2
  const int epi = EPIC >= 0 ? EPIC : a.epi;
  float* stg = reinterpret_cast<float*>(smem);
  constexpr int CPR = BN / EPS, RPP = NT / CPR, SUB = 32 / RPP;     // chunks per row, rows per sub-pass, sub-passes per 32-pixel pass
  const int chunk = tid % CPR, rl = tid / CPR;
  const int c0 = n0 + chunk * EPS;
  const bool cok = c0 < a.ldy;
  float es[EPS], eh[EPS], ms[EPS], mh[EPS], s1[EPS], s2[EPS], sft[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { es[j] = 1.f; eh[j] = 0.f; ms[j] = 0.f; mh[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; sft[j] = 0.f; }
  if (cok) {
    if ((epi & TF_EPI_STATS) && a.stat_shift) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) sft[j] = a.stat_shift[c0 + j];
    }
    if (epi & TF_EPI_AFFINE) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { es[j] = a.epi_scale[c0 + j]; eh[j] = a.epi_shift[c0 + j]; }
    }
    if (epi & TF_EPI_MASK) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { ms[j] = a.mask_scale[c0 + j]; mh[j] = a.mask_shift[c0 + j]; }
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                      // pass 0 fully consumed
    {
      // 32x32 accumulator: lane l holds pixel l & 31, channels 8*g + 4*(l >> 5) + {0..3} for g = 0..3 (registers 4g .. 4g+3)
      float* mine = stg + kg * STG;
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(mine + r32 * PITCH + wn * WCH + n * 32 + g * 8 + h * 4) =
              f32x4{acc[n][half][4 * g], acc[n][half][4 * g + 1], acc[n][half][4 * g + 2], acc[n][half][4 * g + 3]};
    }
    __syncthreads();
#pragma unroll
    for (int sp = 0; sp < SUB; ++sp) {
      const int row = sp * RPP + rl;                // 0..31 inside the pass
      const int p = m0 + half * 32 + row;
      float v[EPS];
#pragma unroll
      for (int j = 0; j < EPS; j += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(stg + row * PITCH + chunk * EPS + j) +
                        *reinterpret_cast<const f32x4*>(stg + STG + row * PITCH + chunk * EPS + j);
        v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
      }
      if (!(p < a.M && cok)) continue;
      if (epi & TF_EPI_STATS) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) { const float t = v[j] - sft[j]; s1[j] += t; s2[j] += t * t; }
      }
      const size_t o = ((size_t)p * a.ldy + c0) * sizeof(T);
      float ax[EPS];
      if (epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux + o), ax);
      if (epi & TF_EPI_AFFINE) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = v[j] * es[j] + eh[j];
      }
      if (epi & TF_EPI_RES) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] += ax[j];
      }
      if (epi & TF_EPI_MASK) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = (ax[j] * ms[j] + mh[j] > 0.f) ? v[j] : 0.f;
      }
      if (epi & TF_EPI_JOIN) {
        float y2[EPS], g3[EPS];
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), g3);
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] += (y2[j] > 0.f) ? g3[j] : 0.f;
      }
      if (epi & TF_EPI_MASK2) {
        float y2[EPS];
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = (y2[j] > 0.f) ? v[j] : 0.f;
      }
      if (epi & TF_EPI_RELU) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (epi & TF_EPI_STATS2) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * ax[j]; }
      }
      if (epi & TF_EPI_STATS3) {
        float x3[EPS];
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), x3);
#pragma unroll
        for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * x3[j]; }
      }
      *reinterpret_cast<uint4*>(a.y + o) = tf::pack16<T>(v);
    }
  }
  if (epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) {       // block-uniform: column sums of the tile
    // lanes sharing a chunk inside a wave differ in the lane bits >= log2(CPR)  (full exec mask here: see common.h)
#pragma unroll
    for (int j = 0; j < EPS; ++j) { s1[j] = tf::lane_group_sum<(CPR < 64 ? CPR : 64)>(s1[j]); s2[j] = tf::lane_group_sum<(CPR < 64 ? CPR : 64)>(s2[j]); }
    __syncthreads();                                 // staging tiles fully consumed
    float* red = reinterpret_cast<float*>(smem);     // [8 waves][2][BN]
    if (l < CPR) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { red[(wave * 2 + 0) * BN + l * EPS + j] = s1[j]; red[(wave * 2 + 1) * BN + l * EPS + j] = s2[j]; }
    }
    __syncthreads();
    if ((epi & TF_EPI_STATS) && a.stat_shift && a.stat_shift_out && mt == 0) {
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
}

// ring depths by (BN, PRO): 160 KiB of LDS = weight ring + pixel ring + coefficient table (<= 16 KiB at K = 1024)
template <int BN, int PRO> struct Rings {
  static constexpr int NSW = BN == 256 ? 2 : 3;
  static constexpr int NSX = PRO == 0 ? (BN == 256 ? 8 : 10) : (BN == 256 ? (PRO == 4 ? 4 : 5) : (PRO == 4 ? 5 : 6));
};

template <int BN, int PRO, int EPIC>
int launch(const tf_conv_args* A, const PK& k, hipStream_t stream) {
  constexpr int NSW = Rings<BN, PRO>::NSW, NSX = Rings<BN, PRO>::NSX;
  constexpr int NOP = PRO ? 2 : 1, NCOEF = PRO == 2 ? 3 : (PRO == 3 ? 2 : (PRO == 4 ? 4 : 0));
  const size_t ring = (size_t)NSW * BN * 128 + (size_t)NSX * NOP * XOP + (size_t)NCOEF * k.K * 4;
  const size_t stg = (size_t)2 * 32 * (BN + 4) * 4;
  const size_t lds = ring > stg ? ring : stg;
  if (lds > 160 * 1024) return TF_ERR_UNSUPPORTED;
  static tf::PerDevice attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pwx_kernel<BN, PRO, NSW, NSX, EPIC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const double M = k.M, Kt = k.K;
  double bytes = (M * Kt * (PRO ? 3.0 : 1.0) + (double)A->Cout * Kt + M * A->Cout) * 2;     // PRO: two operands read, the transformed one written
  if (A->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) bytes += M * A->Cout * 2;
  if (A->epi & TF_EPI_JOIN) bytes += 2 * M * A->Cout * 2;
  if (A->epi & TF_EPI_MASK2) bytes += M * A->Cout * 2;
  if (A->epi & TF_EPI_STATS3) bytes += M * A->Cout * 2;
  const double alg_k = A->alg_k > 0 ? A->alg_k : Kt, alg_n = A->alg_n > 0 ? A->alg_n : A->Cout;
  tf::ProfScope prof(17, 2.0 * M * alg_n * alg_k, bytes, stream, k.M, A->Cout, k.K, 1, A->mode, A->epi, 2.0 * M * A->Cout * Kt, true);   // 17 = conv_pwx bf16
  TF_LAUNCH_TIMED((conv_pwx_kernel<BN, PRO, NSW, NSX, EPIC>), dim3(k.mtiles * k.ntiles), dim3(NT), lds, stream, k);     // (a fork of the executor may ride on this launch)
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

template <int PRO, int EPIC>
int launch_bn(const tf_conv_args* A, const PK& k, bool wide, hipStream_t stream) {
  return wide ? launch<256, PRO, EPIC>(A, k, stream) : launch<128, PRO, EPIC>(A, k, stream);
}

void fill_common(PK& k, const tf_conv_args* A) {
  k.x = (const char*)A->x; k.xc = nullptr; k.w = (const char*)A->w; k.y = (char*)A->y; k.t1 = nullptr;
  k.epi_scale = A->epi_scale; k.epi_shift = A->epi_shift;
  k.aux = (const char*)A->aux; k.aux2 = (const char*)A->aux2; k.aux3 = (const char*)A->aux3;
  k.mask_scale = A->mask_scale; k.mask_shift = A->mask_shift; k.stat_out = A->stat_out;
  k.stat_shift = A->stat_shift; k.stat_shift_out = A->stat_shift_out;
  k.pstat = k.pgamma = k.pmean = k.pinvstd = nullptr; k.pdgamma = k.pdbeta = nullptr;
  k.f1 = FwdTab{}; k.f2 = FwdTab{}; k.feps = 0.f; k.fmom = 0.f;
  k.prows = 0; k.pnk = 0; k.pkidx = 0; k.pcount = 0.f;
  k.M = A->N * A->OH * A->OW; k.K = A->Cin; k.ldy = A->ldy; k.nst = A->Cin / 64; k.epi = A->epi;
  k.srows = tf_get_stat_rows(); k.mtiles = (k.M + BM - 1) / BM;
  k.ntiles = A->Cout / (A->Cout % 256 == 0 ? 256 : 128);
}
FwdTab tab_of(const tf_bn_fwd_desc* d) {
  return FwdTab{d->stat, d->gamma, d->beta, d->scale, d->shift, d->mean, d->invstd, d->running_mean, d->running_var, d->stat_shift};
}

}  // namespace

// pointwise (1x1, stride 1, pad 0) bf16 conv / data gradient with Cout a multiple of 128 (one block = 64 pixels x 128 or 256 channels),
// Cin a multiple of 64 in [128, 1024].
bool tf_conv_pwx_applicable(const tf_conv_args* a) {
  if (a->dtype != TF_BF16 || a->pro_scale || a->bnf) return false;
  if (a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->H != a->OH || a->W != a->OW) return false;
  if (a->Cin % 64 != 0 || a->Cin < 128 || a->Cin > 1024 || a->Cout % 128 != 0 || a->ldy != a->Cout) return false;
  return true;
}
int tf_conv_pwx_mtiles(const tf_conv_args* a) { return (a->N * a->OH * a->OW + BM - 1) / BM; }

// `pro` != NULL: the pixel operand is A*x + B*x2 + D with the coefficients of a BatchNorm backward derived from `pro` (its statistic rows),
// x2 = pro_x2 the BN's input, and the applied tensor is also written to pro_out (may be NULL) -- tf_bn_bwd_apply_fused + tf_conv2d in one launch.
int tf_conv_pwx_launch(const tf_conv_args* A, const tf_bn_bwd_desc* pro, const void* pro_x2, void* pro_out, int pro_rows, float pro_count,
                       hipStream_t stream) {
  if (!tf_conv_pwx_applicable(A)) return TF_ERR_UNSUPPORTED;
  if (pro && (!pro->stat || !pro->gamma || !pro->mean || !pro->invstd || !pro_x2 || pro_rows < 1 || pro->nk < 2 || pro->kidx < 1 || pro->kidx >= pro->nk))
    return TF_ERR_ARG;
  PK k;
  fill_common(k, A);
  const bool wide = A->Cout % 256 == 0;
  if (!pro) return launch_bn<0, -1>(A, k, wide, stream);
  k.xc = (const char*)pro_x2; k.t1 = (char*)pro_out;
  k.pstat = pro->stat; k.pgamma = pro->gamma; k.pmean = pro->mean; k.pinvstd = pro->invstd; k.pdgamma = pro->dgamma; k.pdbeta = pro->dbeta;
  k.prows = pro_rows; k.pnk = pro->nk; k.pkidx = pro->kidx; k.pcount = pro_count;
  const bool spec_off = tf::tuning().epi_spec_off;
  if (!spec_off && A->epi == (TF_EPI_MASK | TF_EPI_STATS2)) return launch_bn<2, TF_EPI_MASK | TF_EPI_STATS2>(A, k, wide, stream);     // the executor's conv3 data gradient
  return launch_bn<2, -1>(A, k, wide, stream);
}

// the conv's pixel operand is y = relu(bn(x) + (bn_res(res) | res)) with the batch statistics of `bn` (and `bn_res`) finalized in-kernel
// exactly like tf_bn_add_relu_fused (scale / shift / mean / invstd + running statistics published by the first block); y is also written
// to y_out [M][Cin].
int tf_conv_pwx_launch_fwd(const tf_conv_args* A, const tf_bn_fwd_desc* bn, const void* res, const tf_bn_fwd_desc* bn_res, void* y_out, int rows,
                           float count, float eps, float momentum, hipStream_t stream) {
  if (!tf_conv_pwx_applicable(A)) return TF_ERR_UNSUPPORTED;
  PK k;
  fill_common(k, A);
  k.xc = (const char*)res; k.t1 = (char*)y_out;
  k.f1 = tab_of(bn);
  if (bn_res) k.f2 = tab_of(bn_res);
  k.feps = eps; k.fmom = momentum; k.prows = rows; k.pcount = count;
  const bool wide = A->Cout % 256 == 0;
  const bool spec_off = tf::tuning().epi_spec_off;
  const bool spec = !spec_off && A->epi == TF_EPI_STATS;                                  // the executor's training-mode conv1
  if (bn_res) return spec ? launch_bn<4, TF_EPI_STATS>(A, k, wide, stream) : launch_bn<4, -1>(A, k, wide, stream);
  return spec ? launch_bn<3, TF_EPI_STATS>(A, k, wide, stream) : launch_bn<3, -1>(A, k, wide, stream);
}

#else   // default build: the fused BatchNorm prologues are not part of the library (common.h: TF_EXPERIMENTAL)
bool tf_conv_pwx_applicable(const tf_conv_args*) { return false; }
int tf_conv_pwx_mtiles(const tf_conv_args*) { return 0; }
int tf_conv_pwx_launch(const tf_conv_args*, const tf_bn_bwd_desc*, const void*, void*, int, float, hipStream_t) { return TF_ERR_UNSUPPORTED; }
int tf_conv_pwx_launch_fwd(const tf_conv_args*, const tf_bn_fwd_desc*, const void*, const tf_bn_fwd_desc*, void*, int, float, float, float, hipStream_t) { return TF_ERR_UNSUPPORTED; }
#endif
