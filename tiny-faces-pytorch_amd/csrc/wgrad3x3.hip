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
//   * per stage a wave issues 72 MFMAs (9 taps x 2 k-steps x 2x2 fragments) behind ONE barrier, against 40 fragment reads:
//     the kernel is MFMA-bound by construction (1152 MFMA cycles vs ~690 LDS cycles per CU-stage), where the per-tap kernel
//     was LDS-bound (profiles/r01e_layer_table.md rows 1, 10, 11: 14-17x over roofline);
//   * nine accumulator sets (144 registers) and one atomic epilogue instead of nine.
// Split-K over the padded pixel range; fp32 atomics into dW (packed [Cout][tap][Cin] or OIHW).  bf16 only, no prologue.
#include <cstdlib>
#include "common.h"
#include "profile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ uint4 g_w3zero_page[8];

struct W3K {
  const char* x; const char* dy; float* dw;
  int N, H, W, Cin, Cout, ldx, lddy, dw_ld, ci_stride, tap_stride;
  int Hp, Wp, HWp, Mp;              // padded frame: H+2, W+2, their product, N * HWp
  int nco, nci, splitk, chunk, hb;  // hb = halo in 64-row chunks on either side: ceil((Wp + 1) / 64)
};

constexpr int PK = 64, NS = 3, YT = PK * 128, XROWS = 512, XBYTES = XROWS * 128, L = 4;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}
// same source-side swizzle as wgrad_dma.hip: conflict-free ds_read_b64_tr_b16 for ANY eight consecutive rows (the shifted reads of
// the taps start at arbitrary rows; 512 is a multiple of 8, so the ring wrap keeps the pattern)
__device__ __forceinline__ int fsw(int row) { return ((row >> 1) & 3) << 1; }

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// dY fragment: 16 channels (c0 multiple of 16) x 32 pixels of a linear 64-row stage tile; k <-> pixel = h*16 + g*4 + j
__device__ __forceinline__ bf16x8 frag_y(const char* tile, int pk0, int c0) {
  const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
  const int row = pk0 + g * 4 + (i >> 2);
  const int slot = (c0 >> 3) + ((i & 3) >> 1);
  const char* p = tile + row * 128 + ((slot ^ fsw(row)) << 4) + ((i & 1) << 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * 128));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// X fragment from the circular image: ring row of the fragment's first pixel = rowbase (any integer, wrapped here)
__device__ __forceinline__ bf16x8 frag_x(const char* ring, int rowbase, int c0) {
  const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
  const int row = (rowbase + g * 4 + (i >> 2)) & (XROWS - 1);
  const int slot = (c0 >> 3) + ((i & 3) >> 1);
  const int off = row * 128 + ((slot ^ fsw(row)) << 4) + ((i & 1) << 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ring + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ring + ((off + 16 * 128) & (XBYTES - 1))));   // fsw(row+16) == fsw(row)
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

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
  __device__ __forceinline__ void advance(const W3K& a) {
    wp += PK;
    while (wp >= a.Wp) { wp -= a.Wp; if (++hp == a.Hp) { hp = 0; ++n; } }
  }
  __device__ __forceinline__ bool interior(const W3K& a) const {
    return (unsigned)n < (unsigned)a.N && (unsigned)(hp - 1) < (unsigned)a.H && (unsigned)(wp - 1) < (unsigned)a.W;
  }
  __device__ __forceinline__ size_t pixel(const W3K& a) const { return ((size_t)n * a.H + (hp - 1)) * a.W + (wp - 1); }
};

__global__ void __launch_bounds__(256, 1) wgrad3x3_kernel(const W3K a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const yring = smem;
  char* const xring = smem + NS * YT;
  int b = blockIdx.x;
  const int ks = b % a.splitk; b /= a.splitk;
  const int tci = b % a.nci; b /= a.nci;
  const int tco = b;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int pb = ks * a.chunk, pe = min(a.Mp, pb + a.chunk);
  const int nst = pe > pb ? (pe - pb + PK - 1) / PK : 0;
  if (nst == 0) return;
  const int x0 = pb - PK * a.hb;                         // padded index of ring row 0
  const int D = 2 * a.hb;                                // stage st needs X chunks st .. st + D
  const int tid = threadIdx.x, lrow = tid >> 3, pslot = tid & 7;
  const int wave_byte = (tid & ~63) * 16;
  const char* zero = reinterpret_cast<const char*>(g_w3zero_page) + pslot * 16;

  // per-thread DMA rows: lrow and lrow + 32 of every 64-row group; physical slot pslot receives logical slot pslot ^ fsw(row)
  PadPos ypos[2], xpos[2];
  int yrow[2];                                           // padded index of the dY row in the NEXT stage to issue
  int ycol[2], xcol[2];
  bool ycok[2], xcok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = lrow + i * 32;
    const int ls = pslot ^ fsw(r);
    yrow[i] = pb + r;
    ypos[i].init(pb + r, a);
    xpos[i].init(x0 + r, a);
    ycol[i] = co0 + ls * 8; xcol[i] = ci0 + ls * 8;
    ycok[i] = ycol[i] + 8 <= a.lddy; xcok[i] = xcol[i] + 8 <= a.ldx;
  }
  int ys_slot = 0, xc = 0;                               // dY ring slot / X chunk index of the next issue
  auto issue_y = [&]() {
    char* ys = yring + ys_slot * YT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = yrow[i] < pe && ycok[i] && ypos[i].interior(a);
      const uintptr_t src = ok ? reinterpret_cast<uintptr_t>(a.dy) + (ypos[i].pixel(a) * a.lddy + ycol[i]) * 2 : reinterpret_cast<uintptr_t>(zero);
      dma16(reinterpret_cast<const void*>(src), ys + i * 4096 + wave_byte);
      ypos[i].advance(a); yrow[i] += PK;
    }
    if (++ys_slot == NS) ys_slot = 0;
  };
  auto issue_x = [&]() {
    char* xs = xring + ((xc * PK) & (XROWS - 1)) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = xcok[i] && xpos[i].interior(a);
      const uintptr_t src = ok ? reinterpret_cast<uintptr_t>(a.x) + (xpos[i].pixel(a) * a.ldx + xcol[i]) * 2 : reinterpret_cast<uintptr_t>(zero);
      dma16(reinterpret_cast<const void*>(src), xs + i * 4096 + wave_byte);
      xpos[i].advance(a);
    }
    ++xc;
  };

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[t][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave = tid >> 6, wco = wave & 1, wci = wave >> 1;

  // prologue: the halo chunks 0 .. D-1 first, then the (dY stage, X chunk) pairs of stages 0 and 1.  From then on every loop
  // iteration issues exactly one pair (L = 4 DMA instructions per thread), so vmcnt(L) == "everything but the youngest pair landed"
  for (int c = 0; c < D; ++c) issue_x();
#pragma unroll
  for (int j = 0; j < NS - 1; ++j) {
    issue_y();                                           // always issued: past the slice the rows come from the zero page (keeps the count uniform)
    issue_x();
  }
  int cs = 0;
  for (int st = 0; st < nst; ++st) {
    wait_vm<L*(NS - 2)>();                               // pair (st+1) may still be in flight; pair st and all older ones landed
    __builtin_amdgcn_s_barrier();                        // everyone's pieces of stage st landed; stage st-1 fully consumed
    issue_y(); issue_x();                                // pair st + 2 (dY slot (st+2) % 3, X chunk st + D + 2)
    const char* ys = yring + cs * YT;
    const int rb0 = PK * (st + a.hb);                    // ring row of padded pixel pb + 64 st
#pragma unroll
    for (int k0 = 0; k0 < PK; k0 += 32) {
      bf16x8 fy[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) fy[n] = frag_y(ys, k0, wco * 32 + n * 16);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int shift = (t / 3 - 1) * a.Wp + (t % 3 - 1);
        bf16x8 fx[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) fx[m] = frag_x(xring, rb0 + k0 + shift, wci * 32 + m * 16);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[t][n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[n], fx[m], acc[t][n][m], 0, 0, 0);
      }
    }
    if (++cs == NS) cs = 0;
  }
  wait_vm<0>();                                          // drain the pairs issued past the end before the LDS is released

  const int l = tid & 63, li = l & 15, lg = l >> 4;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int ci = ci0 + wci * 32 + m * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wco * 32 + n * 16 + lg * 4 + r;
          if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + (size_t)co * a.dw_ld + (size_t)ci * a.ci_stride + (size_t)t * a.tap_stride, acc[t][n][m][r]);
        }
      }
}

}  // namespace

// 3x3, stride 1, pad 1, same-size output, bf16, no prologue, W <= 125.  TF_ERR_UNSUPPORTED otherwise (the caller falls back).
int tf_wgrad3x3_launch(const tf_wgrad_args* A, hipStream_t stream) {
  if (A->dtype != TF_BF16 || A->pro_scale) return TF_ERR_UNSUPPORTED;
  if (A->KH != 3 || A->KW != 3 || A->stride != 1 || A->pad != 1 || A->OH != A->H || A->OW != A->W) return TF_ERR_UNSUPPORTED;
  if (A->W + 3 > 2 * PK) return TF_ERR_UNSUPPORTED;           // halo of at most two 64-row chunks per side (512-row ring)
  W3K k;
  k.x = (const char*)A->x; k.dy = (const char*)A->dy; k.dw = A->dw_oihw;
  k.N = A->N; k.H = A->H; k.W = A->W; k.Cin = A->Cin; k.Cout = A->Cout; k.ldx = A->ldx; k.lddy = A->lddy; k.dw_ld = A->dw_ld;
  if (A->packed) { k.ci_stride = 1; k.tap_stride = A->Cin; } else { k.ci_stride = 9; k.tap_stride = 1; }
  k.Hp = A->H + 2; k.Wp = A->W + 2; k.HWp = k.Hp * k.Wp;
  const long long mp = (long long)A->N * k.HWp;
  if (mp > (1ll << 30)) return TF_ERR_UNSUPPORTED;
  k.Mp = (int)mp;
  k.hb = (k.Wp + 1 + PK - 1) / PK;
  k.nco = (A->Cout + 63) / 64; k.nci = (A->Cin + 63) / 64;
  const int tiles = k.nco * k.nci;
  int sk = A->splitk;
  if (sk <= 0) {
    // one block per CU (88 KiB of LDS): split the padded pixel range until ~256 blocks exist, but keep >= 6 stages per block
    // so that the 2*hb halo chunks and the nine-tap atomic epilogue stay a small part of a block
    static const int target = [] { const char* e = getenv("TINYFACES_WGRAD3_BLOCKS"); return e ? atoi(e) : 256; }();
    sk = (target + tiles - 1) / tiles;
    const int maxsk = (k.Mp + 6 * PK - 1) / (6 * PK);
    if (sk > maxsk) sk = maxsk;
    if (sk < 1) sk = 1;
  }
  k.chunk = (((k.Mp + sk - 1) / sk) + PK - 1) / PK * PK;
  k.splitk = (k.Mp + k.chunk - 1) / k.chunk;
  const size_t lds = (size_t)NS * YT + XBYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const double Md = (double)A->N * A->H * A->W;
  tf::ProfScope prof(14, 2.0 * Md * A->Cout * A->Cin * 9, (Md * A->Cout + Md * A->Cin) * 2 + (double)A->Cout * A->Cin * 9 * 4, stream, (int)Md,
                     A->Cout, A->Cin * 9, 9, 2, 0);
  hipLaunchKernelGGL(wgrad3x3_kernel, dim3(tiles * k.splitk), dim3(256), lds, stream, k);
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}
