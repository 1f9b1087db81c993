// conv_dma, fp16 operands (BASELINE.json configs[4]: inference only): see conv_dma_impl.h
#include "conv_dma_impl.h"

int tf_conv_dma_launch_f16(const tf_conv_args* a, int tile, int depth, hipStream_t stream) { return launch_half<tf::f16_t>(a, tile, depth, stream); }
