// tf_conv2d: argument checks, tile selection (pick_tile) and dispatch of the convolution kernels (gfx950).
// Replaces the nn.Conv2d / BatchNorm2d / ReLU / residual-add chain of tinyfaces/models/model.py:90-106 and the torchvision Bottleneck
// (see tinyfaces_hip.h).  GEMM view:  D[channel][pixel] = sum_k W[channel][k] * X[pixel][k],  k = (tap, cin).
// The kernels: conv_dma_impl.h (LDS-DMA implicit GEMM: every conv but ...), conv3x3h.hip (... the 3x3 / stride 1 convs with >= 256 input
// channels), conv_pwx.hip (a BatchNorm pass fused into the operand path of the pointwise conv / data gradient that consumes it).  r4: the register-staged kernel of round 1
// (conv_igemm, tile codes 1-3, the only one with a producer-BN prologue) was dead on the executor's path since round 1's LDS-DMA
// kernel and was removed: tf_conv2d returns TF_ERR_UNSUPPORTED for tile codes < 10 and for pro_scale (tf_conv2d_wgrad keeps its prologue).
#include <cstdlib>
#include "common.h"
#include "tuning.h"
#include "profile.h"

int tf_conv_dma_launch(const tf_conv_args* a, int tile, int depth, hipStream_t stream);   // conv_dma.hip
bool tf_conv3x3h_applicable(const tf_conv_args* a, bool forced);                              // conv3x3h.hip
int tf_conv3x3h_mtiles(const tf_conv_args* a);
int tf_conv3x3h_launch(const tf_conv_args* a, hipStream_t stream);
bool tf_conv_pwx_applicable(const tf_conv_args* a);                                            // conv_pwx.hip
int tf_conv_pwx_mtiles(const tf_conv_args* a);
int tf_conv_pwx_launch(const tf_conv_args* a, const tf_bn_bwd_desc* pro, const void* pro_x2, void* pro_out, int pro_rows, float pro_count, hipStream_t stream);
bool tf_conv_pws_applicable(const tf_conv_args* a);                                            // conv_pws.hip
int tf_conv_pws_launch(const tf_conv_args* a, hipStream_t stream);
int tf_conv_pwx_launch_fwd(const tf_conv_args* a, const tf_bn_fwd_desc* bn, const void* res, const tf_bn_fwd_desc* bn_res, void* y_out, int rows, float count,
                           float eps, float momentum, hipStream_t stream);

namespace {

int pick_tile(const tf_conv_args* a) {
  if (a->tile) return a->tile;
  // 60 = conv_pwx (8 waves, 64 pixels x all output channels, register-staged pixel operand): only on request so far
  // (TINYFACES_PWX_FWD=1: the pointwise convs with 128 / 256 output channels and K >= 128, A/B knob)
  // 70 = conv_pws (r5): wave-autonomous streaming kernel for the short-K / large-M pointwise launches (layer 1, the large pyramid levels);
  // TINYFACES_PWS_OFF=1: the tiled kernel as in rounds 1-4
  const bool pws_off = tf::tuning().pws_off;
  if (!pws_off && tf_conv_pws_applicable(a)) return 70;
  const bool pwx_fwd = tf::tuning().pwx_fwd;
  if (pwx_fwd && tf_conv_pwx_applicable(a) && a->Cin >= 512) return 60;
  // measured on the bs=12 500x500 layer shapes (scripts/microbench.py): 128 pixels x 64 channels wins on every
  // layer (more, smaller tiles -> more blocks in flight per CU); 64x64 only when even that leaves CUs idle.
  const long M = (long)a->N * a->OH * a->OW;
  {
    // LDS-DMA pipeline.  bf16 convs of <= 4 K-stages (K <= 256 pointwise) are dispatch + epilogue bound: 128-pixel tiles without a
    // ring halve their block count at the same 4 blocks per CU (+0.6 % on the step, A/B on one box: 1036.6 -> 1043.3 img/s);
    // everything else 64x64 with the ring depth chosen by K.  The choice lives HERE so that tf_conv_mtiles agrees with the launch.
    const bool t12_off = tf::tuning().t12_shortk_off;
    const int nst = a->KH * a->KW * (a->Cin / 64);
    // r3: ... except where the launch is big enough to be throughput-bound: from M = 16 384 pixels on, 64 pixels x 128 channels on
    // 32x32x16 fragments with a 2-deep ring moves the same bytes faster (1920x2560 pyramid level: 256 -> 1024 at M = 19 200 35.6 -> 30.7 us,
    // 128 -> 512 at M = 76 800 49.1 -> 40.8 us, 64 -> 256 at M = 307 200 103 -> 95 us; profiles/r03_microbench_eval.txt)
    const bool t46_off = tf::tuning().t46_shortk_off;
    // (r4: a 128 x 128 / eight-wave pointwise kernel, conv_pw8, was built for the K >= 512 GEMMs and measured no faster on any layer shape:
    //  profiles/r04_conv_pw8_negative.txt -- removed again)
    // (not with the in-LDS BN prologue `bnf`: only the ring-less 128 x 64 tile implements it -- ADVICE r3)
    // (TINYFACES_SHORTK_BIG_TILE: A/B knob -- another tile code for these launches, e.g. 44 = 128 x 128 / 45 = 128 x 64 on the same fragments)
    const int shortk_big = tf::tuning().shortk_big_tile;
    // (TINYFACES_T46_HANDOVER_MIN_M: A/B knob -- the M from which the hand-over data gradients (RES + MASK2 [+ STATS3]: four M x Cout tensors per launch) take it)
    const long hand_min_m = tf::tuning().t46_handover_min_m;
    const long big_min_m = (a->epi & (TF_EPI_MASK2 | TF_EPI_STATS3)) ? hand_min_m : 16384L;
    if (!t46_off && !a->bnf && a->dtype != TF_F32 && nst <= 4 && M >= big_min_m && a->Cout % 128 == 0 && a->Cout >= 256) return shortk_big;
    // (TINYFACES_SHORTK_TILE: A/B knob -- another tile code for these launches, e.g. 42 = the same 128 x 64 tile with a 2-slot ring)
    const int shortk_tile = tf::tuning().shortk_tile;
    if (!t12_off && a->dtype != TF_F32 && nst <= 4) return shortk_tile;
    // 32x32x16 fragments (64 pixels x 128 channels per block, 32 x 64 per wave, 2-deep ring) win where a launch still has several
    // blocks per CU AND a long K loop: 3x3 convs / K >= 576 with M >= 16 384 pixels -- layer 2 at bs = 12 (26.3 vs 32.9 us forward,
    // 25.0 vs 30.6 us data gradient) and layers 2-3 of the evaluation pyramid (41.0 vs 50.6, 40.2 vs 44.8, 21.3 vs 23.8 us);
    // at M = 12 288 (layer 3, bs = 12) they lose: one block per CU, nothing hides a wave's LDS latency
    // (profiles/r02a_microbench_mma32_tiles.txt, profiles/r02a_microbench_eval_tiles.txt)
    // 3x3 / stride 1 with >= 128 output channels and enough tiles to fill the chip: the halo-resident kernel (conv3x3h.hip), which
    // moves each input byte into LDS once per 64-channel chunk instead of once per tap
    if (tf_conv3x3h_applicable(a, false)) return 50;
    const bool mma32_off = tf::tuning().mma32_off;
    if (!mma32_off && a->dtype != TF_F32 && nst >= 9 && M >= 16384 && a->Cout % 128 == 0) return 46;
    return 13;
  }
}
int tile_bm(int t) { return ((t % 10) == 3 || (t % 10) == 6) ? 64 : 128; }

}  // namespace

extern "C" int tf_conv_mtiles(const tf_conv_args* a) {
  const long M = (long)a->N * a->OH * a->OW;
  const int t = pick_tile(a);
  if (t == 50) { const int mt = tf_conv3x3h_mtiles(a); return mt > tf_get_stat_rows() ? tf_get_stat_rows() : mt; }
  if (t == 60) { const int mt = tf_conv_pwx_mtiles(a); return mt > tf_get_stat_rows() ? tf_get_stat_rows() : mt; }
  // 70 = conv_pws: its blocks add into row blockIdx % rows of the FOLDED statistic rows.  An explicit tile = 70 on a launch the kernel does not
  // take (tf_conv2d then returns TF_ERR_UNSUPPORTED) must not size the caller's buffer from it -- with tf_set_stat_rows(0) that would be 1 << 30
  // rows: such a request is an argument error here too.
  if (t == 70) return tf_conv_pws_applicable(a) ? tf_get_stat_rows() : TF_ERR_UNSUPPORTED;
  const int bm = tile_bm(t);
  const int mt = (int)((M + bm - 1) / bm);
  return (t >= 10 && mt > tf_get_stat_rows()) ? tf_get_stat_rows() : mt;     // the DMA kernel folds its tiles into <= TF_STAT_ROWS rows
}

extern "C" int tf_conv2d(const tf_conv_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !a->x || !a->w || !a->y) return TF_ERR_ARG;
  const int kch = a->dtype == TF_F32 ? 32 : 64;
  if (a->dtype != TF_BF16 && a->dtype != TF_F32 && a->dtype != TF_F16) return TF_ERR_UNSUPPORTED;
  if (a->pro_scale) return TF_ERR_UNSUPPORTED;    // the producer-BN prologue went with the register-staged kernel (r4)
  if (a->Cin % kch != 0 || a->ldy % 4 != 0 || a->ldy < a->Cout) return TF_ERR_ARG;
  if (a->stride != 1 && a->stride != 2) return TF_ERR_UNSUPPORTED;
  if ((a->epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) && !a->stat_out) return TF_ERR_ARG;
  {
    const int ns = !!(a->epi & TF_EPI_STATS) + !!(a->epi & TF_EPI_STATS2) + !!(a->epi & TF_EPI_STATS3);
    if (ns > 1) return TF_ERR_ARG;
  }
  if ((a->epi & TF_EPI_MASK2) && !a->aux2) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_STATS3) && !a->aux3) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_JOIN) && (a->epi & (TF_EPI_MASK2 | TF_EPI_STATS3))) return TF_ERR_ARG;     // aux2 / aux3 have one meaning per launch
  if ((a->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) && !a->aux) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_JOIN) && (!a->aux2 || !a->aux3)) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_AFFINE) && (!a->epi_scale || !a->epi_shift)) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_MASK) && (!a->mask_scale || !a->mask_shift)) return TF_ERR_ARG;
  const int t = pick_tile(a);
  if (a->bnf && !(t == 32 && a->mode == 0 && a->KH == 1 && a->KW == 1 && a->stride == 1 && a->Cin <= 256)) return TF_ERR_UNSUPPORTED;
  if (t == 50) return tf_conv3x3h_applicable(a, true) ? tf_conv3x3h_launch(a, stream) : TF_ERR_UNSUPPORTED;
  if (t == 60) return tf_conv_pwx_launch(a, nullptr, nullptr, nullptr, 0, 0.f, stream);
  if (t == 70) return tf_conv_pws_launch(a, stream);
  if (t >= 10) {
    return tf_conv_dma_launch(a, t % 10, t >= 40 ? 2 : (t >= 30 ? 1 : (t >= 20 ? 4 : 3)), stream);
  }
  return TF_ERR_UNSUPPORTED;                      // tile codes 1-3: the register-staged kernel of round 1, removed in r4
}

// tf_bn_bwd_apply_fused + tf_conv2d (pointwise, mode 0 or 1) in ONE launch: the conv's pixel operand is A*x + B*x2 + D with the
// BatchNorm-backward coefficients of `bn` (derived in-kernel from its statistic rows), the applied tensor also goes to `applied_out`
// (the weight gradient's operand).  conv_pwx.hip; TF_ERR_UNSUPPORTED for shapes it does not take (the caller runs the two kernels).
#if TF_EXP
extern "C" int tf_conv2d_bnbwd(const tf_conv_args* a, const tf_bn_bwd_desc* bn, const void* x2, void* applied_out, int rows, float count, void* stream_) {
  if (!a || !a->x || !a->w || !a->y || !bn || !x2) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) && !a->stat_out) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) && !a->aux) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_MASK) && (!a->mask_scale || !a->mask_shift)) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_MASK2) && !a->aux2) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_STATS3) && !a->aux3) return TF_ERR_ARG;
  if (rows < 1 || rows > TF_STAT_ROWS) return TF_ERR_ARG;
  return tf_conv_pwx_launch(a, bn, x2, applied_out, rows, count, (hipStream_t)stream_);
}

// r5: tf_bn_add_relu_fused + the pointwise tf_conv2d that consumes its output (conv1 of the NEXT bottleneck), in ONE launch: the conv's pixel
// operand is y = relu(bn(a->x) + (bn_res(res) | res)) with the batch statistics finalized in-kernel; y also goes to y_out (the block output:
// residual of the next block, operand of conv1's weight gradient, ReLU mask of the backward pass).  conv_pwx.hip; TF_ERR_UNSUPPORTED for
// shapes it does not take (the caller runs the two kernels).
extern "C" int tf_conv2d_bnfwd(const tf_conv_args* a, const tf_bn_fwd_desc* bn, const void* res, const tf_bn_fwd_desc* bn_res, void* y_out, int rows,
                               float count, float eps, float momentum, void* stream_) {
  if (!a || !a->x || !a->w || !a->y || !bn || !res || !y_out) return TF_ERR_ARG;
  if (!bn->stat || !bn->gamma || !bn->beta || !bn->scale || !bn->shift || !bn->mean || !bn->invstd) return TF_ERR_ARG;
  if (bn_res && (!bn_res->stat || !bn_res->gamma || !bn_res->beta || !bn_res->scale || !bn_res->shift || !bn_res->mean || !bn_res->invstd)) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_STATS | TF_EPI_STATS2 | TF_EPI_STATS3)) && !a->stat_out) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_RES | TF_EPI_MASK | TF_EPI_STATS2)) && !a->aux) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_AFFINE) && (!a->epi_scale || !a->epi_shift)) return TF_ERR_ARG;
  if ((a->epi & TF_EPI_MASK) && (!a->mask_scale || !a->mask_shift)) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_MASK2 | TF_EPI_JOIN)) && !a->aux2) return TF_ERR_ARG;
  if ((a->epi & (TF_EPI_STATS3 | TF_EPI_JOIN)) && !a->aux3) return TF_ERR_ARG;
  if (rows < 1 || rows > TF_STAT_ROWS) return TF_ERR_ARG;
  return tf_conv_pwx_launch_fwd(a, bn, res, bn_res, y_out, rows, count, eps, momentum, (hipStream_t)stream_);
}
#endif
