// conv_dma: dispatch by operand type + the fp32 instantiations; the kernel lives in conv_dma_impl.h, the 2-byte types in
// conv_dma_bf16.hip / conv_dma_f16.hip (one translation unit per type: they compile in parallel).
#include "conv_dma_impl.h"

int tf_conv_dma_launch_bf16(const tf_conv_args* a, int tile, int depth, hipStream_t stream);
int tf_conv_dma_launch_f16(const tf_conv_args* a, int tile, int depth, hipStream_t stream);

// tile: 1 = 128x128, 2 = 128x64, 3 = 64x64 (pixels x channels); depth: ring stages (3 or 4; 1 = ring-less 128x64)
int tf_conv_dma_launch(const tf_conv_args* a, int tile, int depth, hipStream_t stream) {
  // (64x128, 64x256, 128x128x4 and 128x256 tiles were measured and lost to 64x64x3 on every layer shape:
  //  profiles/r01c_microbench_wide_tiles.txt; they were removed again.)
  if (tile > 6 || (tile > 3 && a->dtype == TF_F32)) return TF_ERR_UNSUPPORTED;
  if (a->dtype == TF_BF16) return tf_conv_dma_launch_bf16(a, tile, depth, stream);
  if (a->dtype == TF_F16) return tf_conv_dma_launch_f16(a, tile, depth, stream);
  if (tile == 1) return launch<float, 128, 128, 3>(a, stream);
  if (tile == 2) return launch<float, 128, 64, 3>(a, stream);
  return launch<float, 64, 64, 4>(a, stream);
}
