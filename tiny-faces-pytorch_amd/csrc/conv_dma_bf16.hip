// conv_dma, bf16 operands (training + bf16 inference): see conv_dma_impl.h
#include "conv_dma_impl.h"

int tf_conv_dma_launch_bf16(const tf_conv_args* a, int tile, int depth, hipStream_t stream) { return launch_half<tf::bf16_t>(a, tile, depth, stream); }
