// conv_pwx: pointwise conv / data gradient for NARROW outputs (Cout = 128 or 256) with a long reduction (K = 256 ... 1024), whose
// pixel operand is REGISTER-staged so that a BatchNorm-backward "apply" can ride on it (round 3).
//
// Why.  In the backward pass of a bottleneck the chain runs   bn_bwd_apply (C = 4 planes)  ->  dgrad conv3 (K = 4 planes -> planes):
// the apply kernel reads g and c3, writes T1 = A*g + B*c3 + D (75 MB at layer 3, 14 us, HBM-bound), and the data gradient reads T1
// straight back (15 us).  The LDS-DMA operand path of conv_dma cannot apply anything on the fly (no register stage), which is why
// rounds 1-2 kept the elementwise kernel.  Here
//   * one block owns 64 pixels x ALL output channels (BN = Cout: 128 / 256), so the pixel operand is touched ONCE: 512 threads =
//     512 16-byte pieces of a 64-pixel x 64-channel stage, one per thread: global -> VGPR (g and c3) -> A*g + B*c + D -> ds_write
//     into the swizzled LDS tile, and the same registers go to T1 in global memory (the weight gradient of conv3 still wants it);
//   * weights stream by LDS-DMA through a 3-deep ring (32 KiB stages, counted vmcnt, one raw barrier per stage) exactly like
//     conv3x3h; 8 waves = 4 channel groups x 2 K halves on 32x32x16 fragments (two waves per SIMD: one wave's DMA issue and LDS
//     latency under its partner's MFMAs), K halves merged through two fp32 staging tiles, 64 pixels in two passes (66 KiB);
//   * the coefficients A, B, D of all K channels are derived in-kernel from the BN's backward statistic rows (the arithmetic of
//     bn_fused.hip bwd_table) into a 12 KiB LDS table while the first weight stages are in flight; block 0 publishes dgamma / dbeta.
// The pixel-operand loads are inline asm: hipcc waits vmcnt(0) for any ordinary load while LDS-DMAs are in flight, which would drain
// the weight ring every stage (cdna_hip_programming.md 5, "three .s-level traps" (b)); their waits are counted by hand.  Stores to
// T1 are issued inside the K loop; counted waits EXCLUDE them (a wait that is one op too strict is safe, loads retire in order).
// PRO = 0: the same kernel without the apply (plain pointwise conv: conv1 of a bottleneck, K = 4 planes -> planes).
// Epilogue = conv_dma's (every TF_EPI_* flag).  bf16 only (the fp32 parity path keeps the unfused kernels).
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "profile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef tf::bf16_t T;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));       // asm-friendly 16-byte register quad (HIP's uint4 is a struct)

__device__ uint4 g_pwx_zero[8];

constexpr int BM = 64, NT = 512, NSW = 3, EPS = 8;
constexpr int XBUF = BM * 128;                    // 8 KiB: 64 pixels x 64 k

struct PK {
  const char* x; const char* xc; const char* w; char* y; char* t1;
  const float* epi_scale; const float* epi_shift;
  const char* aux; const char* aux2; const char* aux3;
  const float* mask_scale; const float* mask_shift;
  float* stat_out; const float* stat_shift; float* stat_shift_out;
  const float* pstat; const float* pgamma; const float* pmean; const float* pinvstd; float* pdgamma; float* pdbeta;
  int prows, pnk, pkidx; float pcount;
  int M, K, ldy, nst, ntiles, epi, srows, mtiles;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) << 2) ^ (((row >> 3) & 1) * 6); }
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ swz(row)) << 4); }
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BN, int PRO>
__global__ void __launch_bounds__(NT, 2) conv_pwx_kernel(const PK a) {
  constexpr int WBUF = BN * 128, W_RING = NSW * WBUF, X_AT = W_RING, COEF_AT = X_AT + 2 * XBUF;
  constexpr int WPASS = BN / 64;                  // weight DMAs per thread and stage
  constexpr int NX = PRO == 2 ? 2 : 1;            // pixel-operand loads per thread and stage
  constexpr int WCH = BN / 4, NF = WCH / 32;      // channels per wave, 32-channel fragments per wave
  constexpr int PITCH = BN + 4, STG = 32 * PITCH; // one K half of a 32-pixel pass, floats
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int logical;
  {
    const int nb = gridDim.x, b = blockIdx.x, q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lrow = tid >> 3, pslot = tid & 7, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // ---- weight DMA roles: rows lrow + 64 i of the BN x 64 stage; physical slot pslot of row r receives logical slot pslot ^ swz(r)
  const char* wptr[WPASS];
#pragma unroll
  for (int i = 0; i < WPASS; ++i) {
    const int row = lrow + i * 64;
    wptr[i] = a.w + (size_t)(n0 + row) * a.K * sizeof(T) + ((pslot ^ swz(row)) << 4);
  }
  auto issue_w = [&](int stage) {
    char* dst = smem + (stage % NSW) * WBUF + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < WPASS; ++i) dma16(wptr[i] + (size_t)stage * 128, dst + i * 8192);
  };
  const int nst = a.nst;
  issue_w(0);
  issue_w(1);

  // ---- coefficient table of the BN-backward apply (bn_fused.hip bwd_table): A, B, D for every K channel
  float* cA = reinterpret_cast<float*>(smem + COEF_AT);
  float* cB = cA + a.K;
  float* cD = cB + a.K;
  if constexpr (PRO == 2) {
    const bool writer = logical == 0;
    for (int c = tid; c < a.K; c += NT) {
      double s1 = 0.0, s2 = 0.0;
      for (int r = 0; r < a.prows; ++r) {
        s1 += (double)a.pstat[(size_t)(r * a.pnk) * a.K + c];
        s2 += (double)a.pstat[(size_t)(r * a.pnk + a.pkidx) * a.K + c];
      }
      const double mu = a.pmean[c], is = a.pinvstd[c], ga = a.pgamma[c];
      const double dg = (s2 - mu * s1) * is;
      const double A = ga * is;
      cA[c] = (float)A;
      cB[c] = (float)(-A * is * dg / a.pcount);
      cD[c] = (float)(-A * s1 / a.pcount + A * mu * is * dg / a.pcount);
      if (writer) {
        if (a.pdgamma) a.pdgamma[c] = (float)dg;
        if (a.pdbeta) a.pdbeta[c] = (float)s1;
      }
    }
  }

  // ---- pixel operand role: ONE 16-byte piece per thread and stage: row lrow (0..63), logical k slot ls
  const int ls = pslot ^ swz(lrow);
  const int prow = m0 + lrow;
  const bool pvalid = prow < a.M;
  const char* zero = reinterpret_cast<const char*>(g_pwx_zero) + pslot * 16;
  const char* xsrc = pvalid ? a.x + ((size_t)prow * a.K + ls * 8) * sizeof(T) : zero;
  const char* csrc = (PRO == 2 && pvalid) ? a.xc + ((size_t)prow * a.K + ls * 8) * sizeof(T) : zero;
  const int xstep = pvalid ? 128 : 0;
  char* t1dst = (PRO == 2 && a.t1 && pvalid && nt == 0) ? a.t1 + ((size_t)prow * a.K + ls * 8) * sizeof(T) : nullptr;
  u32x4 xg = u32x4(0u), xc = u32x4(0u);
  auto load_x = [&](int stage) {
    const char* p = xsrc + (size_t)stage * xstep;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xg) : "v"(p) : "memory");
    if constexpr (PRO == 2) {
      const char* q = csrc + (size_t)stage * xstep;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xc) : "v"(q) : "memory");
    }
  };
  // the loaded piece -> (apply) -> LDS tile (buffer stage & 1), and T1
  auto put_x = [&](int stage) {
    uint4 out = make_uint4(xg[0], xg[1], xg[2], xg[3]);
    if constexpr (PRO == 2) {
      float gf[EPS], cf[EPS], A[EPS], B[EPS], D[EPS];
      tf::unpack16<T>(out, gf); tf::unpack16<T>(make_uint4(xc[0], xc[1], xc[2], xc[3]), cf);
      const int k0 = stage * 64 + ls * 8;
#pragma unroll
      for (int j = 0; j < EPS; j += 4) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(cA + k0 + j), vb = *reinterpret_cast<const f32x4*>(cB + k0 + j),
                    vd = *reinterpret_cast<const f32x4*>(cD + k0 + j);
        A[j] = va[0]; A[j + 1] = va[1]; A[j + 2] = va[2]; A[j + 3] = va[3];
        B[j] = vb[0]; B[j + 1] = vb[1]; B[j + 2] = vb[2]; B[j + 3] = vb[3];
        D[j] = vd[0]; D[j + 1] = vd[1]; D[j + 2] = vd[2]; D[j + 3] = vd[3];
      }
#pragma unroll
      for (int j = 0; j < EPS; ++j) gf[j] = A[j] * gf[j] + B[j] * cf[j] + D[j];
      out = tf::pack16<T>(gf);
      if (t1dst) *reinterpret_cast<uint4*>(t1dst + (size_t)stage * 128) = out;
    }
    *reinterpret_cast<uint4*>(smem + X_AT + (stage & 1) * XBUF + lrow * 128 + pslot * 16) = out;
  };

  load_x(0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(xg), "+v"(xc) : : "memory");
  __syncthreads();                                  // the coefficient table is complete (and everything issued so far has landed)
  put_x(0);

  // ---- MFMA roles: wave = (K half kg, channel group wn); 2 pixel fragments x NF channel fragments of 32 x 32
  const int kg = wave >> 2, wn = wave & 3;
  const int l = tid & 63, r32 = l & 31, h = l >> 5;
  int xo[2][2], wo[NF][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) xo[m][j] = X_AT + lds_off(m * 32 + r32, (kg * 2 + j) * 2 + h);
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int j = 0; j < 2; ++j) wo[n][j] = lds_off(wn * WCH + n * 32 + r32, (kg * 2 + j) * 2 + h);
  f32x16 acc[NF][2];
#pragma unroll
  for (int n = 0; n < NF; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[n][m] = f32x16(0.f);

  // One MFMA stage: 8 fragment reads, 8 MFMAs (NF = 2) per wave
  auto compute = [&](int s, int wslot) {
    const char* wb = smem + wslot * WBUF;
    const char* xb = smem + (s & 1) * XBUF;
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
  // The asm loads of the pixel operand are invisible to hipcc: between the load statement and the wait statement it believes xg / xc
  // hold their values and may COPY them (a first version branched around two wait statements; on one path the register allocator put
  // the phi copies BEFORE the wait: stale pieces for the cache lines that landed last).  So the body below is straight-line for every
  // stage but the last -- ONE load statement, ONE wait statement, the same vmcnt immediate every time: when no weight stage is left to
  // request, the same number of DMAs re-reads the zero page into the ring slot that has just been released -- and the last stage is peeled.
  int wslot = 0;                                    // ring slot of stage s
  for (int s = 0; s + 1 < nst; ++s) {
    load_x(s + 1);
    // weight stage s has landed once only what was issued after it is in flight: stage s+1's DMAs and the loads just issued
    wait_vmcnt<WPASS + NX>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // my piece of the pixel tile of stage s is in LDS
    __builtin_amdgcn_s_barrier();                   // everyone's pieces landed; everyone finished reading stage s-1's buffers
    if (s + 2 < nst) issue_w(s + 2);
    else {                                          // keep the count uniform: WPASS DMAs of zeros into the free slot
      char* dst = smem + ((s + 2) % NSW) * WBUF + wave_u * 1024;
#pragma unroll
      for (int i = 0; i < WPASS; ++i) dma16(zero, dst + i * 8192);
    }
    compute(s, wslot);
    // my pixel piece of stage s+1: older than the WPASS DMAs issued above
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xg), "+v"(xc) : "n"(WPASS) : "memory");
    put_x(s + 1);
    if (++wslot == NSW) wslot = 0;
  }
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  compute(nst - 1, wslot);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (T1 stores: nothing may be pending on the LDS-side reuse below; cheap)
  __builtin_amdgcn_s_barrier();                     // all waves done reading the rings -> reuse them as the staging tiles

  // ---------------- epilogue: two passes of 32 pixels; per pass the two K halves park their 32 x BN fp32 tiles, then every thread
  // handles 16 output bytes (8 channels) of one pixel: conv_dma's phase 2
  float* stg = reinterpret_cast<float*>(smem);
  constexpr int CPR = BN / EPS, RPP = NT / CPR, SUB = 32 / RPP;     // chunks per row, rows per sub-pass, sub-passes per 32-pixel pass
  const int chunk = tid % CPR, rl = tid / CPR;
  const int c0 = n0 + chunk * EPS;
  const bool cok = c0 < a.ldy;
  float es[EPS], eh[EPS], ms[EPS], mh[EPS], s1[EPS], s2[EPS], sft[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) { es[j] = 1.f; eh[j] = 0.f; ms[j] = 0.f; mh[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; sft[j] = 0.f; }
  if (cok) {
    if ((a.epi & TF_EPI_STATS) && a.stat_shift) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) sft[j] = a.stat_shift[c0 + j];
    }
    if (a.epi & TF_EPI_AFFINE) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { es[j] = a.epi_scale[c0 + j]; eh[j] = a.epi_shift[c0 + j]; }
    }
    if (a.epi & TF_EPI_MASK) {
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
      if (a.epi & TF_EPI_STATS) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) { const float t = v[j] - sft[j]; s1[j] += t; s2[j] += t * t; }
      }
      const size_t o = ((size_t)p * a.ldy + c0) * sizeof(T);
      float ax[EPS];
      if (a.epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux + o), ax);
      if (a.epi & TF_EPI_AFFINE) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = v[j] * es[j] + eh[j];
      }
      if (a.epi & TF_EPI_RES) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] += ax[j];
      }
      if (a.epi & TF_EPI_MASK) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = (ax[j] * ms[j] + mh[j] > 0.f) ? v[j] : 0.f;
      }
      if (a.epi & TF_EPI_JOIN) {
        float y2[EPS], g3[EPS];
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), g3);
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] += (y2[j] > 0.f) ? g3[j] : 0.f;
      }
      if (a.epi & TF_EPI_MASK2) {
        float y2[EPS];
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux2 + o), y2);
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = (y2[j] > 0.f) ? v[j] : 0.f;
      }
      if (a.epi & TF_EPI_RELU) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (a.epi & TF_EPI_STATS2) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * ax[j]; }
      }
      if (a.epi & TF_EPI_STATS3) {
        float x3[EPS];
        tf::unpack16<T>(*reinterpret_cast<const uint4*>(a.aux3 + o), x3);
#pragma unroll
        for (int j = 0; j < EPS; ++j) { s1[j] += v[j]; s2[j] += v[j] * x3[j]; }
      }
      *reinterpret_cast<uint4*>(a.y + o) = tf::pack16<T>(v);
    }
  }
  if (a.epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) {       // block-uniform: column sums of the tile
    // lanes sharing a chunk inside a wave differ in the lane bits >= log2(CPR)  (full exec mask here: see common.h)
#pragma unroll
    for (int j = 0; j < EPS; ++j) { s1[j] = tf::lane_group_sum<(CPR < 64 ? CPR : 64)>(s1[j]); s2[j] = tf::lane_group_sum<(CPR < 64 ? CPR : 64)>(s2[j]); }
    __syncthreads();                                 // staging tiles fully consumed
    float* red = reinterpret_cast<float*>(smem);     // [8 waves][2][BN]
    const int lane = tid & 63;
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < EPS; ++j) { red[(wave * 2 + 0) * BN + lane * EPS + j] = s1[j]; red[(wave * 2 + 1) * BN + lane * EPS + j] = s2[j]; }
    }
    __syncthreads();
    if ((a.epi & TF_EPI_STATS) && a.stat_shift && a.stat_shift_out && mt == 0) {
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

template <int BN, int PRO>
int launch(const tf_conv_args* A, const PK& k, hipStream_t stream) {
  constexpr int WBUF = BN * 128;
  const size_t ring = (size_t)NSW * WBUF + 2 * XBUF + (PRO == 2 ? (size_t)3 * k.K * 4 : 0);
  const size_t stg = (size_t)2 * 32 * (BN + 4) * 4;
  const size_t lds = ring > stg ? ring : stg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pwx_kernel<BN, PRO>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const double M = k.M, Kt = k.K;
  double bytes = (M * Kt * (PRO == 2 ? 3.0 : 1.0) + (double)A->Cout * Kt + M * A->Cout) * 2;     // PRO 2: g and c read, T1 written
  if (A->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) bytes += M * A->Cout * 2;
  if (A->epi & TF_EPI_JOIN) bytes += 2 * M * A->Cout * 2;
  if (A->epi & TF_EPI_MASK2) bytes += M * A->Cout * 2;
  if (A->epi & TF_EPI_STATS3) bytes += M * A->Cout * 2;
  const double alg_k = A->alg_k > 0 ? A->alg_k : Kt, alg_n = A->alg_n > 0 ? A->alg_n : A->Cout;
  tf::ProfScope prof(17, 2.0 * M * alg_n * alg_k, bytes, stream, k.M, A->Cout, k.K, 1, A->mode, A->epi, 2.0 * M * A->Cout * Kt, true);   // 17 = conv_pwx bf16
  TF_LAUNCH_TIMED((conv_pwx_kernel<BN, PRO>), dim3(k.mtiles * k.ntiles), dim3(NT), lds, stream, k);     // (a fork of the executor may ride on this launch)
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

}  // namespace

// pointwise (1x1, stride 1, pad 0) bf16 conv / data gradient with Cout a multiple of 128 (one block = 64 pixels x 128 or 256 channels),
// Cin a multiple of 64 with at least two 64-deep stages; `pro` != NULL: the pixel operand is A*x + B*x2 + D with the coefficients of a
// BatchNorm backward derived from `pro` (tf_bn_bwd_desc: the statistic rows of tf_conv_args-style producers), x2 = pro_x2 the BN's
// input, and the applied tensor is also written to pro_out (may be NULL) -- i.e. tf_bn_bwd_apply_fused + tf_conv2d in one launch.
bool tf_conv_pwx_applicable(const tf_conv_args* a) {
  if (a->dtype != TF_BF16 || a->pro_scale) return false;
  if (a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->H != a->OH || a->W != a->OW) return false;
  if (a->Cin % 64 != 0 || a->Cin < 128 || a->Cin > 1024 || a->Cout % 128 != 0 || a->ldy != a->Cout) return false;
  return true;
}
int tf_conv_pwx_mtiles(const tf_conv_args* a) { return (a->N * a->OH * a->OW + BM - 1) / BM; }
int tf_conv_pwx_launch(const tf_conv_args* A, const tf_bn_bwd_desc* pro, const void* pro_x2, void* pro_out, int pro_rows, float pro_count,
                       hipStream_t stream) {
  if (!tf_conv_pwx_applicable(A)) return TF_ERR_UNSUPPORTED;
  if (pro && (!pro->stat || !pro->gamma || !pro->mean || !pro->invstd || !pro_x2 || pro_rows < 1 || pro->nk < 2 || pro->kidx < 1 || pro->kidx >= pro->nk))
    return TF_ERR_ARG;
  PK k;
  k.x = (const char*)A->x; k.xc = (const char*)pro_x2; k.w = (const char*)A->w; k.y = (char*)A->y; k.t1 = (char*)pro_out;
  k.epi_scale = A->epi_scale; k.epi_shift = A->epi_shift;
  k.aux = (const char*)A->aux; k.aux2 = (const char*)A->aux2; k.aux3 = (const char*)A->aux3;
  k.mask_scale = A->mask_scale; k.mask_shift = A->mask_shift; k.stat_out = A->stat_out;
  k.stat_shift = A->stat_shift; k.stat_shift_out = A->stat_shift_out;
  k.pstat = pro ? pro->stat : nullptr; k.pgamma = pro ? pro->gamma : nullptr; k.pmean = pro ? pro->mean : nullptr;
  k.pinvstd = pro ? pro->invstd : nullptr; k.pdgamma = pro ? pro->dgamma : nullptr; k.pdbeta = pro ? pro->dbeta : nullptr;
  k.prows = pro_rows; k.pnk = pro ? pro->nk : 0; k.pkidx = pro ? pro->kidx : 0; k.pcount = pro_count;
  k.M = A->N * A->OH * A->OW; k.K = A->Cin; k.ldy = A->ldy; k.nst = A->Cin / 64; k.epi = A->epi;
  k.srows = tf_get_stat_rows(); k.mtiles = (k.M + BM - 1) / BM;
  const bool wide = A->Cout % 256 == 0;
  k.ntiles = A->Cout / (wide ? 256 : 128);
  if (wide) return pro ? launch<256, 2>(A, k, stream) : launch<256, 0>(A, k, stream);
  return pro ? launch<128, 2>(A, k, stream) : launch<128, 0>(A, k, stream);
}
