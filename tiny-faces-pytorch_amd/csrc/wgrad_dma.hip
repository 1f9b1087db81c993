// wgrad_dma: the pixel-reduction weight-gradient GEMM (see wgrad.hip) with an LDS-DMA operand ring.
//   dW[co][tap][ci] += sum over a pixel range of dY[p][co] * X[gather(p,tap)][ci]      (bf16 operands, fp32 accumulate)
// Both operands are pixel-major 128-byte rows (64 channels) that go HBM/L2 -> LDS by global_load_lds_dwordx4 with no
// register hop; 3-deep ring, counted vmcnt + raw s_barrier (two pixel-stages in flight across barriers).  The LDS image is
// linear [pixel][8 x 16 B]; the 16-byte slots are XOR-swizzled by ((row>>1)&3)<<1 ON THE SOURCE so that the
// ds_read_b64_tr_b16 transposing reads of the two 16-lane groups served per LDS cycle (8 pixel rows x 32 bytes) cover all
// 64 banks exactly once.  No producer-BN prologue: the executor hands materialised activations.
#include <cstdlib>
#include "common.h"
#include "tuning.h"
#include "profile.h"
#include "lds_dma.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ uint4 g_wzero_page[8];

struct WdK {
  const char* x; const char* dy; float* dw;
  int H, W, Cin, OH, OW, Cout, KW, stride, pad;
  int M, OHW, ldx, lddy, dw_ld, ci_stride, tap_stride;
  int nco, nci, splitk, chunk;
};

constexpr int PK = 64, TILEB = PK * 128, BUF = 2 * TILEB, L = 4;     // ring depth NS: template parameter (3 default; 2 = 32 KiB of LDS per block)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// r4: issued from inline asm (lds_dma.h).  Through __builtin_amdgcn_global_load_lds hipcc saw a pending LDS write and put an
// `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 of EVERY stage (the transposing read carries no memory operand the
// wait-count pass could disambiguate): the ring was drained once per stage, each stage paid the full latency of the DMA issued just
// before it.  The counted vmcnt + barrier of the loop is what orders the fragment reads behind the DMA.
// HIDDEN = false keeps the builtin form of rounds 1-3 for the A/B (TINYFACES_DMA_BUILTIN=1).
template <bool HIDDEN> __device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  if constexpr (HIDDEN) {
    tf::dma16_hidden(gsrc, tf::lds_addr_uniform(lds_wave_base));
  } else {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
  }
}
__device__ __forceinline__ int fsw(int row) { return ((row >> 1) & 3) << 1; }

// 16 channels (c0 multiple of 16) x 32 pixels (one MFMA k-step) from a pixel-major swizzled tile; k <-> pixel = h*16 + g*4 + j
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int pk0, int c0) {
  const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
  const int row = pk0 + g * 4 + (i >> 2);
  const int slot = (c0 >> 3) + ((i & 3) >> 1);
  const char* p = tile + row * 128 + ((slot ^ fsw(row)) << 4) + ((i & 1) << 3);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * 128));    // fsw(row+16) == fsw(row)
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// KIND 1: pointwise (1x1, stride 1, pad 0): X row of pixel p is row p.  KIND 0: generic tap gather.
template <int KIND, int NS = 3, bool HIDDEN = true>
__global__ void __launch_bounds__(256) wgrad_dma_kernel(const WdK a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int b = blockIdx.x;
  const int ks = b % a.splitk; b /= a.splitk;
  const int tci = b % a.nci; b /= a.nci;
  const int tco = b % a.nco; b /= a.nco;
  const int tap = b;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int pbeg = ks * a.chunk, pend = min(a.M, pbeg + a.chunk);
  const int tid = threadIdx.x, lrow = tid >> 3, pslot = tid & 7;
  const int wave_byte = (tid & ~63) * 16;
  const char* zero = reinterpret_cast<const char*>(g_wzero_page) + pslot * 16;

  // per-row running state (rows lrow and lrow+32 of every 64-pixel stage)
  int prow[2];                              // pixel index of the row in the NEXT stage to issue
  const char* yptr[2]; const char* xptr[2]; // running source pointers (valid while the row is inside [pbeg, pend))
  bool ycol[2], xcol[2];
  int rn[2], roh[2], row_[2];               // KIND 0: (n, oh, ow) of prow
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = lrow + i * 32;
    const int ls = pslot ^ fsw(r);          // logical 16-byte slot that must land in physical slot pslot of row r
    prow[i] = pbeg + r;
    ycol[i] = co0 + ls * 8 + 8 <= a.lddy;
    xcol[i] = ci0 + ls * 8 + 8 <= a.ldx;
    yptr[i] = a.dy + ((size_t)prow[i] * a.lddy + co0 + ls * 8) * 2;
    if (KIND == 1) {
      xptr[i] = a.x + ((size_t)prow[i] * a.ldx + ci0 + ls * 8) * 2;
    } else {
      xptr[i] = a.x + (size_t)(ci0 + ls * 8) * 2;
      const int p = prow[i];
      rn[i] = p / a.OHW; const int rem = p - rn[i] * a.OHW; roh[i] = rem / a.OW; row_[i] = rem - roh[i] * a.OW;
    }
  }
  int is_slot = 0;
  auto issue = [&]() {
    char* ys = smem + is_slot * BUF;
    char* xs = ys + TILEB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool in = prow[i] < pend;
      const uintptr_t ysrc = (in && ycol[i]) ? reinterpret_cast<uintptr_t>(yptr[i]) : reinterpret_cast<uintptr_t>(zero);
      dma16<HIDDEN>(reinterpret_cast<const void*>(ysrc), ys + i * 4096 + wave_byte);
      uintptr_t xsrc;
      if (KIND == 1) {
        xsrc = (in && xcol[i]) ? reinterpret_cast<uintptr_t>(xptr[i]) : reinterpret_cast<uintptr_t>(zero);
        xptr[i] += (size_t)PK * a.ldx * 2;
      } else {
        const int ih = roh[i] * a.stride - a.pad + kh, iw = row_[i] * a.stride - a.pad + kw;
        const bool ok = in && xcol[i] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        const int pix = ok ? (rn[i] * a.H + ih) * a.W + iw : 0;
        xsrc = ok ? reinterpret_cast<uintptr_t>(xptr[i]) + (size_t)pix * a.ldx * 2 : reinterpret_cast<uintptr_t>(zero);
        // advance (n, oh, ow) by PK pixels
        row_[i] += PK;
        while (row_[i] >= a.OW) { row_[i] -= a.OW; if (++roh[i] == a.OH) { roh[i] = 0; ++rn[i]; } }
      }
      dma16<HIDDEN>(reinterpret_cast<const void*>(xsrc), xs + i * 4096 + wave_byte);
      yptr[i] += (size_t)PK * a.lddy * 2;
      prow[i] += PK;
    }
    if (++is_slot == NS) is_slot = 0;
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave = tid >> 6, wco = wave & 1, wci = wave >> 1;

  const int nst = pend > pbeg ? (pend - pbeg + PK - 1) / PK : 0;
#pragma unroll
  for (int j = 0; j < NS - 1; ++j)
    if (j < nst) issue();
  int cs = 0;
  for (int st = 0; st < nst; ++st) {
    const int younger = nst - 1 - st;
    if (younger >= NS - 2) wait_vm<L*(NS - 2)>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (st + NS - 1 < nst) issue();
    const char* ys = smem + cs * BUF;
    const char* xs = ys + TILEB;
#pragma unroll
    for (int k0 = 0; k0 < PK; k0 += 32) {
      bf16x8 fy[2], fx[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) fy[n] = frag_tr(ys, k0, wco * 32 + n * 16);
#pragma unroll
      for (int m = 0; m < 2; ++m) fx[m] = frag_tr(xs, k0, wci * 32 + m * 16);
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[n], fx[m], acc[n][m], 0, 0, 0);
    }
    if (++cs == NS) cs = 0;
  }
  if (nst == 0) return;
  const int l = tid & 63, li = l & 15, lg = l >> 4;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int ci = ci0 + wci * 32 + m * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wco * 32 + n * 16 + lg * 4 + r;
        if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + (size_t)co * a.dw_ld + (size_t)ci * a.ci_stride + (size_t)tap * a.tap_stride, acc[n][m][r]);
      }
    }
}

}  // namespace

// bf16, no prologue.  Returns TF_ERR_UNSUPPORTED for anything else (the caller falls back to the register-staged kernel).
int tf_wgrad_dma_launch(const tf_wgrad_args* A, hipStream_t stream) {
  if (A->dtype != TF_BF16 || A->pro_scale) return TF_ERR_UNSUPPORTED;
  WdK k;
  k.x = (const char*)A->x; k.dy = (const char*)A->dy; k.dw = A->dw_oihw;
  k.H = A->H; k.W = A->W; k.Cin = A->Cin; k.OH = A->OH; k.OW = A->OW; k.Cout = A->Cout; k.KW = A->KW; k.stride = A->stride; k.pad = A->pad;
  k.M = A->N * A->OH * A->OW; k.OHW = A->OH * A->OW; k.ldx = A->ldx; k.lddy = A->lddy; k.dw_ld = A->dw_ld;
  if (A->packed) { k.ci_stride = 1; k.tap_stride = A->Cin; } else { k.ci_stride = A->KH * A->KW; k.tap_stride = 1; }
  k.nco = (A->Cout + 63) / 64; k.nci = (A->Cin + 63) / 64;
  const int ntaps = A->KH * A->KW, tiles = k.nco * k.nci * ntaps;
  int sk = A->splitk;
  if (sk <= 0) {
    const int target = tf::tuning().wgrad_blocks;
    sk = (target + tiles - 1) / tiles;                      // ~512 blocks measured best (scripts/microbench.py wgrad)
    const int maxsk = (k.M + 4 * PK - 1) / (4 * PK);
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
  }
  k.chunk = (((k.M + sk - 1) / sk) + PK - 1) / PK * PK;
  k.splitk = (k.M + k.chunk - 1) / k.chunk;
  // ring depth: the weight gradients share every CU's 160 KiB of LDS with the data-gradient chain on the other stream
  // (profiles/r03_contention.txt); TINYFACES_WGRAD_NS=2 trades a shallower ring for a third less LDS per block
  const int ns = tf::tuning().wgrad_ns;
  const size_t lds = (size_t)ns * BUF;
  auto lds3 = [] { return (size_t)3 * BUF; };
  const bool pointwise = ntaps == 1 && A->stride == 1 && A->pad == 0 && A->H == A->OH && A->W == A->OW;
  const double Md = k.M;
  tf::ProfScope prof(14, 2.0 * Md * A->Cout * A->Cin * ntaps,
                     (Md * A->Cout + (double)A->N * A->H * A->W * A->Cin) * 2 + (double)A->Cout * A->Cin * ntaps * 4, stream, k.M, A->Cout,
                     A->Cin * ntaps, ntaps, 2, 0, -1.0, true);
  const bool builtin_dma = tf::tuning().dma_builtin;       // A/B: the compiler-visible DMA of rounds 1-3 (drained every stage)
  const dim3 grid(tiles * k.splitk);
  if (builtin_dma) {
    if (pointwise) TF_LAUNCH_TIMED((wgrad_dma_kernel<1, 3, false>), grid, dim3(256), lds3(), stream, k);
    else TF_LAUNCH_TIMED((wgrad_dma_kernel<0, 3, false>), grid, dim3(256), lds3(), stream, k);
  } else if (pointwise) {
    if (ns == 2) TF_LAUNCH_TIMED((wgrad_dma_kernel<1, 2>), grid, dim3(256), lds, stream, k);
    else TF_LAUNCH_TIMED((wgrad_dma_kernel<1, 3>), grid, dim3(256), lds, stream, k);
  } else {
    if (ns == 2) TF_LAUNCH_TIMED((wgrad_dma_kernel<0, 2>), grid, dim3(256), lds, stream, k);
    else TF_LAUNCH_TIMED((wgrad_dma_kernel<0, 3>), grid, dim3(256), lds, stream, k);
  }
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}
