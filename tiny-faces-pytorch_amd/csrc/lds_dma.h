// LDS-DMA (global_load_lds_dwordx4) issued from inline asm, i.e. HIDDEN from hipcc's wait-count bookkeeping (gfx950).
//
// Why (r4, found in the ISA of wgrad_dma_kernel / wgrad3x3_kernel): hipcc models __builtin_amdgcn_global_load_lds as a pending LDS
// WRITE on the VM counter.  A plain LDS load (ds_read_b128 through a __shared__ pointer) carries a memory operand the wait-count pass can
// disambiguate, and the counted `s_waitcnt vmcnt(N)` pipelines of conv_dma / conv3x3h survive.  The transposing read
// __builtin_amdgcn_ds_read_tr16_b64_* carries none: in front of the first one behind a DMA the compiler inserts `s_waitcnt vmcnt(0)` --
// every stage of a "3-deep ring" then waits for the DMA it has JUST issued (the full L2 / HBM round trip), and the ring is a
// single buffer with extra steps.  Both weight-gradient kernels of rounds 1-3 ran like that.
// With the DMA inside an asm statement the compiler neither counts it nor orders LDS reads behind it: the kernel's own counted
// `s_waitcnt vmcnt(N)` + barrier (cdna_hip_programming.md 5.7 item 1, "no VGPR destination: register-safe") is the ONLY thing that
// orders a fragment read behind a DMA -- exactly what the pipeline was written for.  M0 (the wave-uniform LDS destination) is
// compiler-reserved: saved and restored inside the statement; `s_nop 0` covers the SALU-write-M0 -> LDS-DMA hazard.
#pragma once
#include <stdint.h>

namespace tf {

// wave-uniform 32-bit LDS byte address of a pointer into __shared__ memory (M0 operand of the DMA)
__device__ __forceinline__ uint32_t lds_addr_uniform(const void* lds_ptr) {
  typedef __attribute__((address_space(3))) const char lds_char;
  const uint32_t a = (uint32_t)(uintptr_t)(lds_char*)lds_ptr;
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
}

// one 16-byte piece per lane: LDS[lds_base + lane * 16 .. +16) <- *gsrc (per-lane global address)
__device__ __forceinline__ void dma16_hidden(const void* gsrc, uint32_t lds_base_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base_uniform)
      : "memory");
}

// two / four pieces behind ONE save / restore of M0
__device__ __forceinline__ void dma16_hidden2(const void* g0, uint32_t l0, const void* g1, uint32_t l1) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "s"(l0), "s"(l1)
      : "memory");
}
__device__ __forceinline__ void dma16_hidden4(const void* g0, const void* g1, const void* g2, const void* g3, uint32_t l0, uint32_t l1,
                                              uint32_t l2, uint32_t l3) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %6\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b32 m0, %7\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, off\n\t"
      "s_mov_b32 m0, %8\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(l0), "s"(l1), "s"(l2), "s"(l3)
      : "memory");
}

}  // namespace tf
