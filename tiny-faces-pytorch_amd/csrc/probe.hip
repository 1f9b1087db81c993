// probe: interference kernels for the two-stream contention measurement (scripts/contention.py; profiles/r03_contention.txt).
//
// The backward pass runs the data-gradient chain and the weight gradients on two HIP streams; inside the step both run ~2x slower
// than alone (profiles/r02_layer_table.md).  rocprofv3 --pmc serialises dispatches, so counters cannot see which resource the two
// queues fight over.  These kernels each hog ONE resource of a CU for a chosen time; the victim (a real conv / weight-gradient
// launch on another stream) is timed beside each of them:
//   0 park    : waves that only s_sleep; with `lds_bytes` of dynamic LDS they take LDS capacity + wave slots and nothing else
//   1 l2      : 16-byte global loads over a small (L2-resident) window      -> vector-memory path / L2 bandwidth
//   2 hbm     : 16-byte global loads over a large window                     -> HBM bandwidth
//   3 mfma    : dependent-free v_mfma_f32_32x32x16_bf16 chains               -> the matrix pipes
//   4 ldsdma  : global_load_lds_dwordx4 from an L2-resident window           -> the LDS-DMA path (TA -> LDS write port)
//   5 atomic  : fp32 atomicAdd to an L2-resident window                      -> the L2 atomic units
//   6 ldsread : ds_read_b128 loops                                           -> the LDS read port
//   7 / 8 gridbar : `iters` device-wide barriers inside ONE launch (7: one atomic arrival counter, 8: per-group counters + a global one;
//               agent-scope release / acquire fences); with a
//               window > 4 KiB every thread also writes 16 bytes before and reads another block's 16 bytes after each barrier
//               -> what a persistent ("cooperative") conv -> BN -> conv kernel would pay INSTEAD of a kernel boundary (r4)
// Not part of the product path (debug symbol of the C ABI, like tf_debug_conv3x3h_trace).
#include <hip/hip_ext.h>
#include "common.h"
#include "debug_api.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

template <int KIND>
__global__ void __launch_bounds__(256) probe_kernel(char* buf, size_t window, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const size_t nchunk = window / 16;
  size_t idx = ((size_t)blockIdx.x * blockDim.x + tid) % nchunk;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  if constexpr (KIND == 0) {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(32);
    if (iters < 0) smem[tid] = 1;
  } else if constexpr (KIND == 1 || KIND == 2) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint4 v = *reinterpret_cast<const uint4*>(buf + idx * 16);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        idx = (idx + step) % nchunk;
      }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1.f;
  } else if constexpr (KIND == 3) {
    f32x16 a0 = f32x16(0.f), a1 = f32x16(0.f), a2 = f32x16(0.f), a3 = f32x16(0.f);
    bf16x8 x, w;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(tid & 3); w[j] = (__bf16)0.5f; }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a3, 0, 0, 0);
    }
    if (a0[0] + a1[1] + a2[2] + a3[3] == 123.456f) sink[0] = 1.f;
  } else if constexpr (KIND == 4) {
    char* dst = smem + (tid & ~63) * 16;             // wave-uniform base; 4 KiB per block and pass
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dma16(buf + idx * 16, dst + (u & 3) * 4096);
        idx = (idx + step) % nchunk;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (iters < 0) sink[0] = smem[tid];
  } else if constexpr (KIND == 5) {
    float* f = reinterpret_cast<float*>(buf);
    const size_t nf = window / 4;
    size_t j = ((size_t)blockIdx.x * blockDim.x + tid) % nf;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { atomicAdd(f + j, 1.0f); j = (j + step) % nf; }
    }
  } else if constexpr (KIND == 6) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = reinterpret_cast<const float4*>(smem);
    int o = tid;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float4 v = p[o]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; o = (o + 256) & 1023; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = 1.f;
  } else if constexpr (KIND == 7 || KIND == 8) {
    // buf[0..3]: arrival counter (zeroed by the launcher), buf[4..7]: error flag (a barrier that did not complete), data from byte 4096 on.
    // All blocks must be co-resident (the launcher refuses more than 4 per CU); a spin that runs too long gives up instead of hanging.
    unsigned* ctr = reinterpret_cast<unsigned*>(buf);
    const unsigned nb = gridDim.x;
    const bool data = window > 4096 + (size_t)nb * 256 * 16;
    uint4* slots = reinterpret_cast<uint4*>(buf + 4096);
    unsigned chk = 0;
    volatile int& dead = *reinterpret_cast<volatile int*>(smem);      // (dynamic LDS: a static __shared__ would push the 160 KiB attribute over the limit)
    if (tid == 0) dead = 0;
    for (int i = 0; i < iters; ++i) {
      if (data) slots[(size_t)blockIdx.x * 256 + tid] = make_uint4(i, blockIdx.x, tid, 7);
      __syncthreads();
      if (dead) break;
      if (tid == 0) {
        __threadfence();                                   // release: this block's stores visible device-wide (L2 write-back across XCDs)
        unsigned want;
        if constexpr (KIND == 7) {                         // one arrival counter: nb same-address device-scope atomics per barrier
          atomicAdd(ctr, 1u);
          want = (unsigned)(i + 1) * nb;
        } else {                                           // two levels: 8 group counters 128 B apart (block % 8 ~ the XCD a block lands on), the
          const unsigned G = 8, gs = nb / G;               // last arrival of a group bumps the global one: nb/8 + 8 serialised atomics per barrier
          if (atomicAdd(ctr + 32 * (1 + blockIdx.x % G), 1u) == (unsigned)(i + 1) * gs - 1) atomicAdd(ctr, 1u);
          want = (unsigned)(i + 1) * G;
        }
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { ctr[1] = 1u; dead = 1; break; }
        }
        __threadfence();                                   // acquire
      }
      __syncthreads();
      if (dead) break;
      if (data) {
        const uint4 v = slots[(size_t)((blockIdx.x + nb / 2 + 1) % nb) * 256 + tid];
        chk += (v.x < (unsigned)i);                        // stale = older than this round (a faster block may already have written round i + 1)
      }
    }
    if (chk) ctr[2] = chk;                                 // a stale read after the barrier (must stay 0)
  }
}

// kind 9 (r5): how fast can a block of a pixel-stationary GEMM STREAM its operand?  Every block owns 64 rows of a [rows][2048 B] matrix
// (M x 1024 bf16: the layer-3 activations) and moves them into LDS by LDS-DMA in 8 KiB stages with DEPTH stages in flight; `seg` is the
// run of contiguous bytes a stage takes from one row: 128 = the GEMM pattern (64 rows x one 64-deep k step), 512 / 2048 = fewer rows,
// longer runs (2048: four whole rows per stage).  Same bytes, same requests per stage -- only their addresses differ.
template <int DEPTH>
__global__ void __launch_bounds__(256) stream_probe_kernel(const char* buf, int seg, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = 2048, ROWS = 64, STAGE = 8192, NST = ROWS * RB / STAGE;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = buf + (size_t)blockIdx.x * ROWS * RB;
  const int ppr = seg / 16, rps = STAGE / seg, cgs = RB / seg;        // pieces per row run, rows per stage, column groups
  auto issue = [&](int t) {
    const int rg = t % (ROWS / rps), cg = t / (ROWS / rps);
    char* dst = smem + (t % (DEPTH + 1)) * STAGE + wave * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = tid + 256 * u, r = p / ppr, c = p - r * ppr;
      typedef __attribute__((address_space(3))) void lds_void;
      typedef __attribute__((address_space(1))) const void glb_void;
      __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)(rg * rps + r) * RB + (size_t)(cg % cgs) * seg + c * 16), (lds_void*)(dst + u * 4096), 16, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < DEPTH; ++t) issue(t);
  for (int t = 0; t < NST; ++t) {
    if (t + DEPTH < NST) issue(t + DEPTH);
    else { issue(0); }                                                  // keep the count uniform (re-reads a cached stage)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DEPTH) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (seg < 0) sink[0] = smem[tid];
}

// kind 10 (r5): the EPILOGUE of the hand-over data gradient as a kernel of its own -- no GEMM, no LDS.  A block owns the (128-row x 128-byte)
// tile (mt, nt) of [M][2048 B] matrices (row stride 2 KiB, 16 tiles across a row), loads it from NIN matrices (16 loads of 16 bytes per
// thread and matrix, all issued before the first use), combines them and stores the tile of a further matrix: 1536 tiles at M = 12 288.
// What the memory system gives a kernel with THIS access pattern, free of everything else the conv kernel does.
template <int NIN>
__global__ void __launch_bounds__(256) tile_probe_kernel(const char* buf, size_t mat_bytes, int ntiles, int rows_total) {
  const int mt = blockIdx.x / ntiles, nt = blockIdx.x - mt * ntiles;
  const int t = threadIdx.x, r0 = t >> 3, pc = t & 7;
  uint4 v[NIN][4];
#pragma unroll
  for (int i = 0; i < NIN; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = mt * 128 + r0 + 32 * j;
      v[i][j] = row < rows_total ? *reinterpret_cast<const uint4*>(buf + (size_t)i * mat_bytes + (size_t)row * 2048 + nt * 128 + pc * 16) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint4 o = v[0][j];
#pragma unroll
    for (int i = 1; i < NIN; ++i) { o.x += v[i][j].x; o.y ^= v[i][j].y; o.z += v[i][j].z; o.w ^= v[i][j].w; }
    const int row = mt * 128 + r0 + 32 * j;
    if (row < rows_total) *reinterpret_cast<uint4*>(const_cast<char*>(buf) + (size_t)NIN * mat_bytes + (size_t)row * 2048 + nt * 128 + pc * 16) = o;
  }
}

// kind 11 (r5): the memory skeleton of a WAVE-AUTONOMOUS streaming pointwise conv (short K, large M: layer 1, the large pyramid levels).  Every wave
// of a persistent block owns tiles of 16 pixels: it pulls the tile's input rows (16 x IN bytes) through a private 2-slot LDS ring by LDS-DMA, reads them
// back, writes a 16 x OUT-byte result tile into a private staging area and stores it with 16-byte stores -- no block barrier, only its own counted waits.
// IN / OUT in bytes per pixel: 128 / 512 = the 64 -> 256 conv of layer 1, 512 / 128 = 256 -> 64.  Grid = blocks of NW waves, tiles dealt round-robin.
template <int IN, int OUT>
__global__ void __launch_bounds__(512) wave_stream_probe_kernel(const char* src, char* dst, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XT = 16 * IN, ST = 16 * OUT, PER = 2 * XT + ST;       // ring of two input tiles + one staging tile per wave
  constexpr int NI = XT / 1024;                                        // DMA instructions per tile (1 KiB each)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  char* mine = smem + wave * PER;
  const int stride = gridDim.x * nw;
  int t = blockIdx.x * nw + wave;
  auto issue = [&](int tile, int slot) {
    const char* g = src + (size_t)(tile < ntiles ? tile : 0) * XT + lane * 16;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      typedef __attribute__((address_space(3))) void lds_void;
      typedef __attribute__((address_space(1))) const void glb_void;
      __builtin_amdgcn_global_load_lds((glb_void*)(g + i * 1024), (lds_void*)(mine + slot * XT + i * 1024), 16, 0, 0);
    }
  };
  issue(t, 0);
  int slot = 0;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (; t < ntiles; t += stride) {
    issue(t + stride, slot ^ 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");         // this tile landed; the next one (and older stores) may fly
#pragma unroll
    for (int i = 0; i < NI; ++i) { const uint4 v = *reinterpret_cast<const uint4*>(mine + slot * XT + i * 1024 + lane * 16); acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w; }
    char* stg = mine + 2 * XT;
#pragma unroll
    for (int i = 0; i < ST / 1024; ++i) *reinterpret_cast<uint4*>(stg + i * 1024 + lane * 16) = acc;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < ST / 1024; ++i)
      *reinterpret_cast<uint4*>(dst + (size_t)t * ST + i * 1024 + lane * 16) = *reinterpret_cast<const uint4*>(stg + i * 1024 + lane * 16);
    slot ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// kind 12 (r6): does LDS-DMA traffic slow the matrix pipe?  One block of 768 threads per CU: waves 0-7 (two per SIMD) issue nothing but
// v_mfma_f32_32x32x16_bf16 on register operands (8 per round and wave = the 512 cycles of matrix-pipe time of a conv3x3h K stage), optionally with the
// 8 ds_read_b128 of a stage; waves 8-11 (one per SIMD) stream `dma_per_round` 1 KiB LDS-DMA instructions each per round from an L2-resident window
// into a 64 KiB LDS ring (5 each = the 20 KiB of a stage), never waiting for more than the ring needs.  No barrier anywhere: the two groups share
// nothing but the CU.  mode bits: 1 MFMA waves run, 2 loader waves run, 4 MFMA waves also read fragments from LDS, 8 the loaders use global_load into
// registers instead of LDS-DMA, 16 the loaders store to LDS with ds_write_b128 instead (no memory traffic).  rounds = iters >> 8, mode = iters & 255.
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(768) mix_probe_kernel(const char* buf, size_t window, int rounds, int dma_per_round, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 64 KiB ring (loaders) + 64 KiB fragment area (readers)
  const int lane = threadIdx.x & 63, wave_hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // bit 64: the loaders are the OLDEST waves of the block (hardware waves 0-3) instead of the youngest (8-11)
  const int wave = (MODE & 64) ? (wave_hw < 4 ? wave_hw + 8 : wave_hw - 4) : wave_hw;
  if (wave < 8) {
    if (!(MODE & 1)) return;
    pf32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = pf32x16(0.f);
    pbf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[i] = pbf16x8((__bf16)(1.0f + lane)); fb[i] = pbf16x8((__bf16)(0.5f + i)); }
    const char* frag = smem + 65536 + wave * 8192 + lane * 16;
    if (MODE & 128) {
      // bit 128: NO loader waves -- every MFMA wave issues its own share of the DMA, one instruction behind every (8 / dma_per_round)-th MFMA
      const char* base = buf + ((size_t)blockIdx.x * (1u << 20)) % (window - (1u << 20)) + lane * 16;
      unsigned ofs = wave * 65536u;
      auto dma1 = [&]() {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef __attribute__((address_space(1))) const void glb_void;
        __builtin_amdgcn_global_load_lds((glb_void*)(base + (ofs & ((1u << 20) - 1))), (lds_void*)(smem + ((ofs >> 10) & 7) * 8192 + wave * 1024), 16, 0, 0);
        ofs += 1024;
      };
      for (int r = 0; r < rounds; ++r) {
        if (MODE & 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { fa[i] = *reinterpret_cast<const pbf16x8*>(frag + i * 1024); fb[i] = *reinterpret_cast<const pbf16x8*>(frag + 4096 + i * 1024); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], acc[i], 0, 0, 0);
          if (dma_per_round >= 4 || (dma_per_round == 3 && i < 3) || (dma_per_round == 2 && (i & 1) == 0) || (dma_per_round == 1 && i == 0)) dma1();
          acc[(i + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(i + 1) & 3], fb[i], acc[(i + 2) & 3], 0, 0, 0);
          if (dma_per_round >= 8 || (dma_per_round > 4 && i < dma_per_round - 4)) dma1();
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
    for (int r = 0; r < rounds; ++r) {
      if (MODE & 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i] = *reinterpret_cast<const pbf16x8*>(frag + i * 1024); fb[i] = *reinterpret_cast<const pbf16x8*>(frag + 4096 + i * 1024); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], acc[i], 0, 0, 0);
        acc[(i + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(i + 1) & 3], fb[i], acc[(i + 2) & 3], 0, 0, 0);
      }
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v += acc[i][0] + acc[i][7];
    if (v == 123.456f) sink[0] = v;
  } else {
    if (!(MODE & 2) || (MODE & 128)) return;
    if (MODE & 32) __builtin_amdgcn_s_setprio(3);     // bit 32: the loaders win every issue arbitration
    const int q = wave - 8;
    // each CU walks its own 1 MiB of the window again and again (L2-resident after the first pass)
    const char* base = buf + ((size_t)blockIdx.x * (1u << 20)) % (window - (1u << 20)) + lane * 16;
    uint4 keep = make_uint4(0, 0, 0, 0);
    unsigned ofs = q * 65536u;
    for (int r = 0; r < rounds; ++r) {
      for (int i = 0; i < dma_per_round; ++i) {
        const char* g = base + (ofs & ((1u << 20) - 1));
        char* dst = smem + ((ofs >> 10) & 15) * 4096 + q * 1024;
        if (MODE & 8) { const uint4 v = *reinterpret_cast<const uint4*>(g); keep.x ^= v.x; keep.y += v.y; keep.z ^= v.z; keep.w += v.w; }
        else if (MODE & 16) { *reinterpret_cast<uint4*>(dst + lane * 16) = keep; keep.x += 1; }
        else {
          typedef __attribute__((address_space(3))) void lds_void;
          typedef __attribute__((address_space(1))) const void glb_void;
          __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)dst, 16, 0, 0);
        }
        ofs += 1024;
      }
      if (!(MODE & 24)) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // at most two rounds in flight per loader
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (keep.x == 0x12345u && keep.y == 77u) sink[1] = (float)keep.z;
  }
}
template <int MODE>
int launch_mix(int blocks, char* buf, size_t window, int iters, int dma, hipStream_t s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_probe_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(mix_probe_kernel<MODE>, dim3(blocks), dim3(768), 128 * 1024, s, buf, window, iters >> 8, dma, reinterpret_cast<float*>(buf + window - 4096));
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}


// kind 13 (r6): what does an edge of a dependent-kernel chain cost WITHOUT the AQL barrier bit?  `iters` kernels of `blocks` x 256 threads;
// kernel k writes 16 B per thread into slab k % 2, then (one lane per block, behind an agent-scope release) bumps the completion counter of ITS XCD in
// row k; before that it waits -- one lane per block, bounded spin with s_sleep, then an agent-scope acquire -- until the eight counters of row k - 1 add
// up to the grid, and checks one value its predecessor wrote.  mode 0: plain launches on one stream, no in-kernel wait (the kernel boundary);
// mode 1: the same launches WITH the in-kernel wait (what the protocol costs on top when the boundary is there anyway); mode 2: hipExtLaunchKernelGGL with
// hipExtAnyOrderLaunch on one stream (hip_ext.h says the flag is not supported on gfx9xx: this measures whether it does anything); mode 3: kernels
// alternate between TWO streams with no event between them (the only way to really drop the barrier bit if mode 2 is ignored).
// ctl[0] = spins that gave up, ctl[1] = stale reads, ctl[2] = waits that found the predecessor already complete.
__global__ void __launch_bounds__(256) chain_probe_kernel(unsigned* counters, unsigned* ctl, uint4* slabs, size_t slab_elems, int k, int wait, unsigned expect) {
  __shared__ int ok;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (wait && k > 0) {
    if (threadIdx.x == 0) {
      const unsigned* row = counters + (size_t)(k - 1) * 32;
      int spins = 0; unsigned sum = 0;
      for (;;) {
        sum = 0;
        for (int x = 0; x < 8; ++x) sum += __hip_atomic_load(row + x * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sum >= expect || ++spins > 200000) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (sum < expect) atomicAdd(ctl + 0, 1u);
      if (spins == 0) atomicAdd(ctl + 2, 1u);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      ok = 1;
    }
    __syncthreads();
    const uint4 v = slabs[(size_t)((k - 1) & 1) * slab_elems + i % slab_elems];
    if (v.x != (unsigned)(k - 1)) atomicAdd(ctl + 1, 1u);
  }
  slabs[(size_t)(k & 1) * slab_elems + i % slab_elems] = make_uint4((unsigned)k, (unsigned)i, 0u, 0u);
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __hip_atomic_fetch_add(counters + (size_t)k * 32 + (xcc & 7) * 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int KIND>
int launch(int blocks, int lds, char* buf, size_t window, int iters, hipStream_t s) {
  static tf::PerDevice attr;
  if (attr.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  hipLaunchKernelGGL(probe_kernel<KIND>, dim3(blocks), dim3(256), lds, s, buf, window, iters, reinterpret_cast<float*>(buf));
  return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
}

}  // namespace

// kind: see the header comment; `blocks` workgroups of 256 threads with `lds_bytes` of dynamic LDS (>= 16 KiB is forced for the
// kinds that touch it); `buf` / `window_bytes`: the device window the memory kinds walk (multiple of 16, >= 4 KiB)
extern "C" int tf_debug_probe(int kind, int blocks, int lds_bytes, void* buf, size_t window_bytes, int iters, void* stream_) {
  if (blocks <= 0 || iters < 0 || lds_bytes < 0 || lds_bytes > 160 * 1024) return TF_ERR_ARG;
  if (kind != 0 && kind != 3 && kind != 6 && (!buf || window_bytes < 4096 || window_bytes % 16)) return TF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream_;
  char* b = (char*)buf;
  if ((kind == 4 || kind == 6) && lds_bytes < 16 * 1024) lds_bytes = 16 * 1024;
  switch (kind) {
    case 0: return launch<0>(blocks, lds_bytes, b, 4096, iters, s);
    case 1: return launch<1>(blocks, lds_bytes, b, window_bytes, iters, s);
    case 2: return launch<2>(blocks, lds_bytes, b, window_bytes, iters, s);
    case 3: return launch<3>(blocks, lds_bytes, b, 4096, iters, s);
    case 4: return launch<4>(blocks, lds_bytes, b, window_bytes, iters, s);
    case 5: return launch<5>(blocks, lds_bytes, b, window_bytes, iters, s);
    case 6: return launch<6>(blocks, lds_bytes, b, 4096, iters, s);
    case 9: {                                               // iters = seg | depth << 16; window >= blocks * 64 * 2048
      const int seg = iters & 0xffff, depth = iters >> 16;
      if ((seg != 128 && seg != 256 && seg != 512 && seg != 1024 && seg != 2048) || window_bytes < (size_t)blocks * 64 * 2048) return TF_ERR_ARG;
      const size_t l = (size_t)(depth + 1) * 8192;
      if (depth == 2) hipLaunchKernelGGL(stream_probe_kernel<2>, dim3(blocks), dim3(256), l, s, b, seg, reinterpret_cast<float*>(b));
      else if (depth == 4) hipLaunchKernelGGL(stream_probe_kernel<4>, dim3(blocks), dim3(256), l, s, b, seg, reinterpret_cast<float*>(b));
      else if (depth == 8) hipLaunchKernelGGL(stream_probe_kernel<8>, dim3(blocks), dim3(256), l, s, b, seg, reinterpret_cast<float*>(b));
      else return TF_ERR_ARG;
      return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
    }
    case 10: {                                              // iters = number of input matrices (1..3); blocks = (rows / 128) * 16; window >= (iters + 1) * rows * 2048
      const int rows = blocks / 16 * 128;
      const size_t mat = (size_t)rows * 2048;
      if (iters < 1 || iters > 3 || blocks % 16 || window_bytes < (size_t)(iters + 1) * mat) return TF_ERR_ARG;
      if (iters == 1) hipLaunchKernelGGL(tile_probe_kernel<1>, dim3(blocks), dim3(256), 0, s, b, mat, 16, rows);
      else if (iters == 2) hipLaunchKernelGGL(tile_probe_kernel<2>, dim3(blocks), dim3(256), 0, s, b, mat, 16, rows);
      else hipLaunchKernelGGL(tile_probe_kernel<3>, dim3(blocks), dim3(256), 0, s, b, mat, 16, rows);
      return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
    }
    case 11: {                                              // iters = pixels | shape << 28 (0: 128 -> 512 B, 1: 512 -> 128 B per pixel) ; lds_bytes = waves per block (4 / 8)
      const int shape = iters >> 28, px = iters & 0x0fffffff, nw = lds_bytes == 4 ? 4 : 8;
      const int in = shape ? 512 : 128, out = shape ? 128 : 512, ntiles = px / 16;
      if (window_bytes < (size_t)px * (in + out)) return TF_ERR_ARG;
      const size_t l = (size_t)nw * (2 * 16 * in + 16 * out);
      static tf::PerDevice a0, a1;
      if (!shape) {
        if (a0.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wave_stream_probe_kernel<128, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((wave_stream_probe_kernel<128, 512>), dim3(blocks), dim3(nw * 64), l, s, b, b + (size_t)px * in, ntiles);
      } else {
        if (a1.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wave_stream_probe_kernel<512, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((wave_stream_probe_kernel<512, 128>), dim3(blocks), dim3(nw * 64), l, s, b, b + (size_t)px * in, ntiles);
      }
      return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
    }
    case 12: {                                              // iters = rounds << 8 | mode; lds_bytes = LDS-DMA instructions per loader wave and round (default 5)
      if (window_bytes < (4u << 20)) return TF_ERR_ARG;
      const int dma = lds_bytes > 0 && lds_bytes <= 32 ? lds_bytes : 5;
      switch (iters & 255) {
        case 1: return launch_mix<1>(blocks, b, window_bytes, iters, dma, s);   case 2: return launch_mix<2>(blocks, b, window_bytes, iters, dma, s);
        case 3: return launch_mix<3>(blocks, b, window_bytes, iters, dma, s);   case 5: return launch_mix<5>(blocks, b, window_bytes, iters, dma, s);
        case 7: return launch_mix<7>(blocks, b, window_bytes, iters, dma, s);   case 10: return launch_mix<10>(blocks, b, window_bytes, iters, dma, s);
        case 11: return launch_mix<11>(blocks, b, window_bytes, iters, dma, s); case 15: return launch_mix<15>(blocks, b, window_bytes, iters, dma, s);
        case 18: return launch_mix<18>(blocks, b, window_bytes, iters, dma, s); case 19: return launch_mix<19>(blocks, b, window_bytes, iters, dma, s);
        case 23: return launch_mix<23>(blocks, b, window_bytes, iters, dma, s);
        case 35: return launch_mix<35>(blocks, b, window_bytes, iters, dma, s); case 39: return launch_mix<39>(blocks, b, window_bytes, iters, dma, s);
        case 129: return launch_mix<129>(blocks, b, window_bytes, iters, dma, s); case 133: return launch_mix<133>(blocks, b, window_bytes, iters, dma, s);
        case 67: return launch_mix<67>(blocks, b, window_bytes, iters, dma, s); case 99: return launch_mix<99>(blocks, b, window_bytes, iters, dma, s);
        case 103: return launch_mix<103>(blocks, b, window_bytes, iters, dma, s); case 43: return launch_mix<43>(blocks, b, window_bytes, iters, dma, s);
      }
      return TF_ERR_ARG;
    }
    case 13: {                                              // iters = kernels in the chain (<= 4096); lds_bytes = mode 0..3; window >= 1 MiB + 2 slabs
      const int n = iters, mode = lds_bytes;
      const size_t slab_elems = (size_t)blocks * 256;
      if (n < 1 || n > 4096 || mode < 0 || mode > 3 || window_bytes < (1u << 20) + 2 * slab_elems * 16) return TF_ERR_ARG;
      unsigned* counters = (unsigned*)b;                    // n rows of 32 words (8 counters 16 B apart) = 512 KiB at most
      unsigned* ctl = (unsigned*)(b + (1u << 20) - 64);
      uint4* slabs = (uint4*)(b + (1u << 20));
      if (hipMemsetAsync(b, 0, 1u << 20, s) != hipSuccess) return TF_ERR_LAUNCH;
      static hipStream_t s2 = nullptr;
      if (mode == 3 && !s2 && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return TF_ERR_LAUNCH;
      if (mode == 3) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return TF_ERR_LAUNCH; (void)hipEventRecord(e, s); (void)hipStreamWaitEvent(s2, e, 0); (void)hipEventDestroy(e); }
      for (int k = 0; k < n; ++k) {
        hipStream_t q = mode == 3 && (k & 1) ? s2 : s;
        if (mode == 2) hipExtLaunchKernelGGL(chain_probe_kernel, dim3(blocks), dim3(256), 0, q, nullptr, nullptr, hipExtAnyOrderLaunch, counters, ctl, slabs, slab_elems, k, 1, (unsigned)blocks);
        else hipLaunchKernelGGL(chain_probe_kernel, dim3(blocks), dim3(256), 0, q, counters, ctl, slabs, slab_elems, k, mode >= 1 ? 1 : 0, (unsigned)blocks);
      }
      if (mode == 3) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return TF_ERR_LAUNCH; (void)hipEventRecord(e, s2); (void)hipStreamWaitEvent(s, e, 0); (void)hipEventDestroy(e); }
      return hipGetLastError() == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
    }
    case 7: case 8:
      if (blocks > 1024 || blocks % 8) return TF_ERR_ARG;   // every block must be resident at once (256 CUs x 4)
      if (hipMemsetAsync(b, 0, 4096, s) != hipSuccess) return TF_ERR_LAUNCH;
      return kind == 7 ? launch<7>(blocks, 64, b, window_bytes, iters, s) : launch<8>(blocks, 64, b, window_bytes, iters, s);
  }
  return TF_ERR_ARG;
}
// `repeat` launches of the same probe from ONE host call (a C loop: the host side of a Python loop costs ~7 us per launch and hides the
// GPU-side cost of a dependent launch, which is what scripts/floor.py wants to see)
extern "C" int tf_debug_probe_chain(int kind, int blocks, int lds_bytes, void* buf, size_t window_bytes, int iters, int repeat, void* stream_) {
  for (int i = 0; i < repeat; ++i) {
    const int rc = tf_debug_probe(kind, blocks, lds_bytes, buf, window_bytes, iters, stream_);
    if (rc != TF_OK) return rc;
  }
  return TF_OK;
}
