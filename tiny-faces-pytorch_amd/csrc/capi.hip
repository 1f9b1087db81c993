// Library identity + the list of symbols a binding must resolve (checked by the CPU test-suite).
#include "common.h"
#include "debug_api.h"

static const char* const kSymbols[] = {
    "tf_version", "tf_build_id", "tf_symbol_count", "tf_symbol_name",
    "tf_targets_workspace_bytes", "tf_dense_overlap_targets", "tf_dense_overlap_iou", "tf_pairwise_iou_distance",
    "tf_nms_workspace_bytes", "tf_nms_f64", "tf_nms_batched_workspace_bytes", "tf_nms_f64_batched",
    "tf_decode_workspace_bytes", "tf_decode_compact",
    "tf_criterion_workspace_bytes", "tf_criterion_fwd_bwd",
    "tf_sgd_step", "tf_image_prepare",
    "tf_conv_mtiles", "tf_conv2d", "tf_pack_weight", "tf_pack_weights_batched", "tf_pack_weights_tiled", "tf_conv2d_wgrad", "tf_conv2d_wgrad_group", "tf_wgrad_workspace_bytes", "tf_unpack_dw",
    "tf_stem_im2col", "tf_stem_conv", "tf_stem_wgrad", "tf_maxpool_fwd", "tf_maxpool_bwd", "tf_maxpool_bwd_stats", "tf_colstats_blocks", "tf_colstats",
    "tf_bn_finalize", "tf_bn_fold", "tf_bn_bwd_finalize", "tf_bn_bwd_apply", "tf_bn_relu", "tf_bn_add_relu",
    "tf_bn_relu_fused", "tf_bn_add_relu_fused", "tf_bn_bwd_apply_fused",
#if TF_EXP
    "tf_conv2d_bnbwd", "tf_conv2d_bnfwd",       // experimental build only (common.h)
#endif
    "tf_upsample_add_crop", "tf_upsample_add_crop_bwd", "tf_reduce_partials",
    "tf_detnet_num_params", "tf_detnet_param_name", "tf_detnet_param_numel", "tf_detnet_workspace_bytes",
    "tf_detnet_out_shape", "tf_detnet_param_region_bytes", "tf_detnet_forward", "tf_detnet_backward", "tf_detnet_ctx_create", "tf_detnet_ctx_destroy", "tf_detnet_forward_ctx", "tf_detnet_backward_ctx", "tf_comm_available", "tf_comm_unique_id", "tf_comm_init", "tf_comm_destroy", "tf_comm_rank", "tf_comm_world", "tf_allreduce_bucket", "tf_comm_join", "tf_comm_allreduce_hook",
    "tf_detnet_set_dual_stream", "tf_detnet_set_grad_events", "tf_detnet_set_grad_callback",
    "tf_set_stat_rows", "tf_get_stat_rows", "tf_profile_enable", "tf_profile_collect", "tf_profile_shapes",
};

namespace tf {
static thread_local hipEvent_t g_next_stop_event = nullptr;
void set_next_stop_event(hipEvent_t e) { g_next_stop_event = e; }
hipEvent_t take_next_stop_event() { hipEvent_t e = g_next_stop_event; g_next_stop_event = nullptr; return e; }
}  // namespace tf

extern "C" int tf_version(void) { return 600; }   // r6: tf_build_id; r5: tf_conv2d_bnfwd; r4: context + hooks + communicator entry points
#ifndef TF_BUILD_ID
#define TF_BUILD_ID "unstamped"
#endif
// digest of the sources this library was built from (build.py: sha256 over csrc/* + include/tinyfaces_hip.h + the compiler flags): what a
// committed profile is stamped with, so that a measurement taken with another binary is recognised as stale (bench.py `traffic_stale`)
extern "C" const char* tf_build_id(void) { return TF_EXP ? TF_BUILD_ID "+x" : TF_BUILD_ID; }      // "+x": built with TF_EXPERIMENTAL
static int g_stat_rows = 8;
extern "C" int tf_set_stat_rows(int rows) { g_stat_rows = rows <= 0 ? (1 << 30) : (rows > TF_STAT_ROWS ? TF_STAT_ROWS : rows); return TF_OK; }
extern "C" int tf_get_stat_rows(void) { return g_stat_rows; }
extern "C" int tf_symbol_count(void) { return (int)(sizeof(kSymbols) / sizeof(kSymbols[0])); }
extern "C" const char* tf_symbol_name(int i) { return (i >= 0 && i < tf_symbol_count()) ? kSymbols[i] : nullptr; }

// Hardware probe used by the GPU test-suite: records what ds_read_b64_tr_b16 returns when LDS
// holds the identity (element e = e) and lane l supplies the address of elements [4l, 4l+4).
namespace {
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void probe_tr16_kernel(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + threadIdx.x * 4));
  const uint2 raw = __builtin_bit_cast(uint2, v);
  out[threadIdx.x * 4 + 0] = (unsigned short)(raw.x & 0xffff); out[threadIdx.x * 4 + 1] = (unsigned short)(raw.x >> 16);
  out[threadIdx.x * 4 + 2] = (unsigned short)(raw.y & 0xffff); out[threadIdx.x * 4 + 3] = (unsigned short)(raw.y >> 16);
}
}  // namespace
extern "C" int tf_probe_tr16(unsigned short* out256, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out256);
  TF_CHECK_LAUNCH();
  return TF_OK;
}
