/* tinyfaces_hip.h -- C ABI of libtinyfaces_hip.so (gfx950 / MI355X).
 *
 * The reference (varunagrawal/tiny-faces-pytorch) has no FFI of its own: its hot path is
 * Python calling torch / torchvision / numpy.  Each entry point below replaces the
 * framework call(s) cited next to it (reference file:line, relative to /root/reference).
 * The Python package tiny-faces-pytorch_amd/tinyfaces binds them with ctypes
 * (tinyfaces/_hip.py) -- see INTEGRATION.md for the stub a maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - the caller owns all memory; kernels never allocate or free; scratch is passed in
 *     (`ws`, sized by the matching *_workspace_bytes());
 *   - everything is enqueued on `stream` (a hipStream_t passed as void*), no host sync
 *     unless stated; stateless and therefore thread-safe per stream;
 *   - return 0 (TF_OK) or a negative TF_ERR_* code; no exceptions cross the boundary.
 *   - activations of the network kernels are NHWC ("pixels x channels" matrices), dtype
 *     TF_F32, TF_BF16 or (inference) TF_F16; maps handed to / from the reference call surface are NCHW float32.
 */
#ifndef TINYFACES_HIP_H
#define TINYFACES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TF_OK 0
#define TF_ERR_ARG (-1)
#define TF_ERR_LAUNCH (-2)
#define TF_ERR_UNSUPPORTED (-3)
#define TF_ERR_WORKSPACE (-4)

#define TF_F32 0
#define TF_BF16 1
#define TF_F16 2      /* IEEE half operands, fp32 accumulation: inference only (BASELINE.json configs[4]); the training entry points refuse it */

int tf_version(void);
/* r6: digest (16 hex digits of a sha256) of the sources and compiler flags the library was built from; profiles/ files are stamped with it */
const char* tf_build_id(void);
/* number of exported symbols a binding must resolve; names via tf_symbol_name(i) */
int tf_symbol_count(void);
const char* tf_symbol_name(int i);

/* ---- dense_overlap + heat-map target assignment ------------------------------------
 * Replaces compute_dense_overlap (tinyfaces/datasets/dense_overlap.py:4-75) fused with
 * DataProcessor.get_padding / get_regression / get_heatmaps
 * (tinyfaces/datasets/processor.py:114-277).  float64 arithmetic, IoU rounded to 14
 * decimals exactly like the reference; the 63x63x25xG IoU tensor is never materialised.
 *   boxes        [total][4] f64 (x1,y1,x2,y2), degenerate boxes already removed
 *                (processor.py:228-232 is done by the host wrapper)
 *   box_offsets  [B+1] i32   image b owns boxes [box_offsets[b], box_offsets[b+1])
 *   templates    [nt][tstride] f64 (x1,y1,x2,y2,...)
 *   paste_boxes  [B][4] i32 (x1,y1,x2,y2) or NULL (= whole image, no padding)
 *   flips        [B] i32 or NULL: pad mask mirrored in x (tinyfaces/datasets/wider_face.py:165)
 *   noise        f64 uniform[0,1) tie-break draws of processor.py:195, per image laid out
 *                (y,x,t,g) at element offset noise_offsets[b]; NULL -> counter RNG(seed)
 *   class_map    [B][nt][vsy][vsx] f32 in {-1,0,1};  reg_map [B][4nt][vsy][vsx] f32
 */
size_t tf_targets_workspace_bytes(int total_boxes);
int tf_dense_overlap_targets(const double* boxes, const int32_t* box_offsets, int B,
                             const double* templates, int nt, int tstride,
                             int vsy, int vsx, int ofy, int ofx, int sty, int stx,
                             const int32_t* paste_boxes, const int32_t* flips,
                             const double* noise, const int64_t* noise_offsets, uint64_t seed,
                             double pos_thresh, double neg_thresh,
                             float* class_map, float* reg_map,
                             void* ws, size_t ws_bytes, void* stream);
/* test hook: the raw rounded IoU tensor [vsy][vsx][nt][G] f64 for ONE image */
int tf_dense_overlap_iou(const double* boxes, int G, const double* templates, int nt, int tstride,
                         int vsy, int vsx, int ofy, int ofx, int sty, int stx,
                         double* iou_out, void* stream);

/* ---- template clustering: distance matrix of the k-medoids ------------------------------
 * Replaces compute_distances (tinyfaces/clustering/cluster.py:28-37, a Python double loop over jaccard_index,
 * tinyfaces/metrics.py:8-40): out[i][j] = 1 - IoU(boxes[i], boxes[j]) in float64, plain areas, IoU = 0 when the union is not
 * positive; bit-exact with the reference's arithmetic.  boxes [n][4] (x1,y1,x2,y2), out [n][n]. */
int tf_pairwise_iou_distance(const double* boxes, int n, double* out, void* stream);

/* ---- greedy NMS, float64 -----------------------------------------------------------
 * Replaces torchvision.ops.nms as called at tinyfaces/evaluation.py:84 (float64 boxes and
 * scores): stable descending sort, areas without +1, strict `>` on IoU.  keep_out holds the
 * surviving INPUT indices in descending-score order; *num_keep (device int32) their count. */
size_t tf_nms_workspace_bytes(int n);
int tf_nms_f64(const double* boxes /*[n][4]*/, const double* scores /*[n]*/, int n, double iou_thresh,
               int64_t* keep_out /*[n]*/, int32_t* num_keep, void* ws, size_t ws_bytes, void* stream);
/* Batched multi-scale form (BASELINE.json configs[4]): S <= TF_NMS_MAX_SEGMENTS independent candidate lists -- the multi-scale
 * candidates of each image of an evaluation batch (the reference runs evaluation.py:80-84 once per image), or one list per pyramid
 * level -- laid out back to back; segment s = rows [host_seg_offsets[s], host_seg_offsets[s+1]) (HOST int32 array, [0] == 0).
 * One NMS per segment with the semantics of tf_nms_f64, all of them in the same three launches.  The survivors of segment s are
 * written to keep_out[host_seg_offsets[s] ...] as indices into the CONCATENATED input (descending score inside the segment),
 * num_keep[s] (device) their count.  Largest segment <= 524 160 boxes.  ws: tf_nms_batched_workspace_bytes(). */
#define TF_NMS_MAX_SEGMENTS 64
size_t tf_nms_batched_workspace_bytes(const int32_t* host_seg_offsets, int num_segments);
int tf_nms_f64_batched(const double* boxes /*[n][4]*/, const double* scores /*[n]*/, const int32_t* host_seg_offsets /*[S+1]*/,
                       int num_segments, double iou_thresh, int64_t* keep_out /*[n]*/, int32_t* num_keep /*[S]*/,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- score map -> boxes: sigmoid + threshold + ORDERED compaction + refinement ------
 * Replaces evaluation.py:61-78 + get_bboxes / regression_refinement
 * (tinyfaces/models/utils.py:4-100).  score [5nt][H][W] f32 (one image, NCHW).
 * Candidates are appended to dets[*count ...] as rows (x1,y1,x2,y2,score) f64 in the
 * reference's order: C-order over (y, x, template).  valid_x[W] / valid_t[nt] (u8) express
 * the template mask; the reference's defect D1 (utils.py:44 masks the W axis) is
 * reproduced by the caller passing it through valid_x. */
size_t tf_decode_workspace_bytes(int H, int W, int nt);
int tf_decode_compact(const float* score, int nt, int H, int W,
                      const double* templates, int tstride,
                      const uint8_t* valid_x, const uint8_t* valid_t,
                      float prob_thresh, double scale, int sty, int stx, int ofy, int ofx,
                      double* dets /*[cap][5]*/, int32_t* count /*device, in/out*/, int cap,
                      void* ws, size_t ws_bytes, void* stream);

/* ---- detection criterion: OHEM + balance sampling + masked SoftMargin / SmoothL1 ----
 * Replaces DetectionCriterion.forward and its autograd backward
 * (tinyfaces/models/loss.py:59-93, tinyfaces/models/utils.py:103-163) with no host
 * round trip.  output [B][5nt][H][W] f32; class_map [B][nt][H][W] f32 is mined IN PLACE
 * (loss.py:62); label_out receives the labels after balance sampling; grad_out =
 * d(total)/d(output); loss_out[0] = sum cls, loss_out[1] = sum reg (unweighted).
 * pos_keep/neg_keep: optional u8 [B][nt*H*W] keep flags indexed by the C-order RANK of the
 * positive / negative label inside its image (injects the reference's np.random
 * permutation); NULL -> uniformly random subset from the counter RNG(seed). */
size_t tf_criterion_workspace_bytes(int B, int nt, int H, int W);
int tf_criterion_fwd_bwd(const float* output, float* class_map, const float* reg_map,
                         int B, int nt, int H, int W, float ohem_thresh, int max_pos, int max_neg,
                         float reg_weight, const uint8_t* pos_keep, const uint8_t* neg_keep, uint64_t seed,
                         float* label_out, float* grad_out, double* loss_out /*[2]*/,
                         int32_t* counts_out /*[B][2] or NULL*/, void* ws, size_t ws_bytes, void* stream);

/* ---- fused SGD step (momentum, weight decay) over a flat fp32 segment ---------------
 * Replaces torch.optim.SGD.step as configured at main.py:67-70 for one parameter group. */
int tf_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n,
                float lr, float momentum, float weight_decay, float grad_scale, void* stream);

/* ---- convolution as MFMA implicit GEMM (NHWC) --------------------------------------
 * Replaces every nn.Conv2d on the path (tinyfaces/models/model.py:25-32,90-106 and the
 * torchvision Bottleneck convs) plus the BN / ReLU / residual passes fused around them.
 *   x  [N][H][W][Cin]  (dtype), w packed [CoutPad][KH*KW][Cin] K-contiguous (tf_pack_weight),
 *   y  [M][ldy] with M = N*OH*OW, ldy >= Cout, multiple of 4.
 * mode 0: y[p,co] = sum x[gather(p,tap),ci] w[co,tap,ci]            (forward conv)
 * mode 1: transposed gather: "x" is dY [N][H][W][Cin'=Cout_fwd] of a conv with (stride,pad)
 *         and y is dX [N][OH][OW][Cout'=Cin_fwd]  (data gradient)
 * prologue (per input channel, fwd only): x <- relu?(x*pro_scale + pro_shift), padding stays 0
 * epilogue flags (TF_EPI_*):
 *   AFFINE  v = v*epi_scale[c] + epi_shift[c]     RES   v += aux[p,c]      RELU  v = max(v,0)
 *   STATS   stat_out[mtile][0][c] += v, [1][c] += v*v  (raw accumulator, before AFFINE)
 *   MASK    v = (aux[p,c]*mask_scale[c]+mask_shift[c] > 0) ? v : 0        (dgrad through ReLU(BN(aux)))
 *   STATS2  stat_out[mtile][0][c] = sum v, [1][c] = sum v*aux[p,c]        (after MASK)
 *   JOIN    v += (aux2[p,c] > 0) ? aux3[p,c] : 0                          (residual-join gradient)
 *   MASK2   v = (aux2[p,c] > 0) ? v : 0    (after RES: gradient through the ReLU that produced aux2; LDS-DMA kernel only)
 *   STATS3  stat_out[mtile][0][c] = sum v, [1][c] = sum v*aux3[p,c]       (after MASK2; LDS-DMA kernel only)
 *           RES|MASK2|STATS3 on the last data-gradient conv of a Bottleneck hands the next block (in backward order)
 *           its already-masked output gradient together with the BN3-backward sums.
 */
#define TF_EPI_AFFINE 1
#define TF_EPI_RES 2
#define TF_EPI_RELU 4
#define TF_EPI_STATS 8
#define TF_EPI_MASK 16
#define TF_EPI_STATS2 32
#define TF_EPI_JOIN 64
#define TF_EPI_MASK2 128
#define TF_EPI_STATS3 256
/* partial-sum rows of one launch are folded (atomics) into at most this many rows; consumers pass clear=1 to the
 * last finalize that reads them so that the buffer is zero again for the next producer */
#define TF_STAT_ROWS 16
/* rows <= 0: do not fold (one partial row per tile / block, plain stores: bit-reproducible statistics; the executor then
 * uses the separate finalize kernels); 1..TF_STAT_ROWS: folded rows (fp32 atomics); default 8 */
int tf_set_stat_rows(int rows);
int tf_get_stat_rows(void);

struct tf_bn_fwd_desc;
typedef struct tf_conv_args {
  int dtype, mode;
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int ldy;            /* row stride (elements) of y, aux, aux2, aux3 */
  int epi;
  int pro_relu;
  const void* x; const void* w; void* y;
  const float* pro_scale; const float* pro_shift;
  const float* epi_scale; const float* epi_shift;
  const void* aux; const void* aux2; const void* aux3;
                      /* r6: mode 1, 1x1, stride 2 (the data gradient of a downsample conv) with TF_EPI_RES and aux == y: IN PLACE -- the launch adds its rows to the
                         even-even pixels of a raster that already holds another launch's result, touches no other pixel, and TF_EPI_MASK2 / TF_EPI_STATS3
                         then apply to those rows only (the sums take the increment).  With aux != y the raster is initialised first (zero, or aux). */
  const float* mask_scale; const float* mask_shift;
  float* stat_out;    /* [mtiles][2][ldy] fp32 partial sums, mtiles = tf_conv_mtiles() */
  int tile;           /* 0 = auto (recommended).  Else a kernel / tile code, pixels x channels: 11 128x128, 12 128x64, 13 64x64 on the LDS-DMA
                         kernel (3-deep ring; 2x = 4-deep, 3x = ring-less, 4x = 2-deep; x4 / x5 / x6 = 128x128 / 128x64 / 64x128 on 32x32x16
                         fragments), 50 = halo-resident 3x3 kernel, 60 = conv_pwx, 70 = conv_pws (r5: wave-streaming pointwise kernel with resident weights).
                         conv_pws takes EXACTLY: 1x1 / stride 1 / pad 0, bf16 or fp16, no prologue, ldy == Cout, M >= 16 384 pixels,
                         (Cin, Cout) in {(64, 256), (256, 64), (64, 64), (256, 128)}, epilogue sets AFFINE[+RELU], AFFINE+RES+RELU (both types) and, bf16
                         only, none, STATS, MASK+STATS2, RES[+MASK2[+STATS3]]; statistic epilogues only with folded rows (tf_get_stat_rows() <=
                         TF_STAT_ROWS).  0 picks it for those launches (TINYFACES_PWS_OFF=1: never); tile = 70 on anything else is TF_ERR_UNSUPPORTED from
                         tf_conv2d and from tf_conv_mtiles (negative return).  Output-channel slices (Cout > 256) exist in the TF_EXPERIMENTAL build only.
                         Codes 1-3 (the register-staged kernel of round 1) were
                         removed in r4: TF_ERR_UNSUPPORTED, like a prologue (pro_scale != NULL) -- tf_conv2d_wgrad keeps its prologue. */
  /* TF_EPI_STATS only (r3): per-channel value subtracted from every output BEFORE it enters the two sums, so that the consumer computes
   * var = E[(x-s)^2] - E[x-s]^2 around a shift s close to the mean instead of E[x^2] - mean^2 (which loses (mean/std)^2 of the
   * significant bits in fp32).  The executor passes the BN's running mean.  stat_shift_out [Cout] receives the shift that was used
   * (written by the launch's first pixel tile) -- the consumer reads it from there, never from a buffer that is updated meanwhile.
   * NULL: no shift (sums of x and x^2 as before). */
  const float* stat_shift; float* stat_shift_out;
  /* r3: training-mode BatchNorm + ReLU of the conv's INPUT applied inside the conv (x := relu(bn(x)) with the batch statistics of `bnf`
   * finalized in-kernel, exactly tf_bn_relu_fused followed by this conv): bf16, 1x1 / stride 1 with Cin <= 256 only (the ring-less
   * LDS-DMA kernel fixes its pixel tile up in LDS after the DMA landed); bnf_out [M][Cin] receives the activated tensor (the weight
   * gradient's operand; may be NULL).  TF_ERR_UNSUPPORTED for other shapes: run tf_bn_relu_fused + tf_conv2d.  NULL: off. */
  const struct tf_bn_fwd_desc* bnf; void* bnf_out; int bnf_rows; float bnf_count, bnf_eps, bnf_momentum;
  int alg_k, alg_n;   /* measurement hooks only: the UNPADDED reduction length (taps * channels) and output-channel count when the
                         operands are zero-padded (stem: 147 of 192, heads: 125 of 128); 0 = Cin*KH*KW / Cout */
} tf_conv_args;

int tf_conv_mtiles(const tf_conv_args* a);   /* rows of stat_out this launch writes (<= TF_STAT_ROWS for the DMA kernel) */
int tf_conv2d(const tf_conv_args* a, void* stream);

/* OIHW fp32 -> packed [CoutPad][KH*KW][CinPad] (dtype); transpose=1 packs the data-gradient
 * operand [CinPad'][KH*KW][Cout] (roles swapped).  Pads are zero-filled. */
int tf_pack_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, int transpose,
                   int dtype, void* out, int rows_pad, int cols_pad, void* stream);
/* the same for many weights in one or two launches (job table passed as kernel arguments, no H2D copy) */
typedef struct tf_pack_job { const float* src; void* dst; int cout, cin, taps, transpose, rows_pad, cols_pad; } tf_pack_job;
int tf_pack_weights_batched(int dtype, const tf_pack_job* host_jobs, int njobs, void* stream);
/* both layouts of a weight from ONE read of the fp32 master (LDS-tiled): dst = [rows_pad >= cout][taps][cols_pad >= cin]
 * (forward operand), dst_t = [rows_pad_t >= cin][taps][cols_pad_t >= cout] (data-gradient operand); either may be NULL;
 * padding is written as zeros.  A training forward packs both, so the backward call packs nothing. */
typedef struct tf_pack2_job {
  const float* src; void* dst; void* dst_t;
  int cout, cin, taps, rows_pad, cols_pad, rows_pad_t, cols_pad_t;
} tf_pack2_job;
int tf_pack_weights_tiled(int dtype, const tf_pack2_job* host_jobs, int njobs, void* stream);

/* weight gradient: dW[co][ci][kh][kw] (+)= sum_p dY[p,co] * xhat[gather(p,tap),ci] (fp32 atomics
 * into dw_oihw, which the caller zeroes).  Same prologue semantics as tf_conv2d. */
typedef struct tf_wgrad_args {
  int dtype;
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int ldx, lddy;      /* row strides of x and dy */
  int pro_relu;
  const void* x; const void* dy; float* dw_oihw;
  const float* pro_scale; const float* pro_shift;
  int dw_ld;          /* elements between consecutive co rows of dw (= Cin*KH*KW normally) */
  int splitk;         /* 0 = auto */
  int tile;           /* 0 = auto; 64 or 128 = channels per tile side */
  int packed;         /* 1: dw is [Cout][KH*KW][Cin] (coalesced atomics; tf_unpack_dw -> OIHW) */
  void* partial_ws;   /* optional scratch of >= tf_wgrad_workspace_bytes(a) bytes: the all-taps 3x3 kernel then reduces its split-K */
  size_t partial_ws_bytes;   /* slices through it (plain stores + one summing kernel) instead of fp32 atomics; NULL / too small: atomics */
  /* tile: 0 = auto, 1 = per-tap LDS-DMA kernel, 3 = all-taps 3x3 kernel (stride 1, pad 1, bf16), 64 / 128 = register-staged kernel */
} tf_wgrad_args;
int tf_conv2d_wgrad(const tf_wgrad_args* a, void* stream);
size_t tf_wgrad_workspace_bytes(const tf_wgrad_args* a);   /* 0 when the all-taps kernel does not apply to `a` */
/* r4: the weight gradients of a GROUP of convolutions in ONE launch, every output tile reduced over ALL pixels inside its block: no
 * split-K, no atomics, no partial-tile workspace -- dw_oihw of every problem is OVERWRITTEN (plain stores), the caller need not zero it.
 * Replaces the autograd weight gradients (tinyfaces/trainer.py:86) of the 22 identity Bottlenecks of layer 3, whose shapes are identical:
 * one such gradient alone has 16 output tiles for 256 CUs; a group of eight bottlenecks has 256.
 *   all problems 1x1 / stride 1 / pad 0 (any channel counts, ONE pixel count N*OH*OW), n <= 48: 128 x 128 tiles (csrc/wgrad_group.hip);
 *   all problems the SAME 3x3 / stride 1 / pad 1 shape, n <= 24: the all-taps kernel with splitk = 1 (csrc/wgrad3x3.hip).
 * bf16, no prologue.  TF_ERR_UNSUPPORTED: nothing was launched, call tf_conv2d_wgrad per problem (into zeroed gradients). */
int tf_conv2d_wgrad_group(const tf_wgrad_args* problems, int n, void* stream);
/* [Cout][taps][Cin] fp32 -> OIHW fp32 (overwrites) */
int tf_unpack_dw(const float* packed, int Cout, int Cin, int taps, float* dw_oihw, void* stream);


/* ---- HBM-bound companions of the conv engine (NHWC, dtype TF_F32 | TF_BF16) ---------- */
/* conv1 (7x7 s2 p3, 3->64; model.py:90): x NCHW fp32 -> im2col [N*OH*OW][ldc], k = c*49+kh*7+kw
 * (the OIHW order of conv1.weight), zero padded to ldc (>= 147, multiple of 8). */
int tf_stem_im2col(const float* x_nchw, int N, int H, int W, int dtype, void* col, int ldc, void* stream);
/* r4: conv1 straight from the NCHW fp32 image, no im2col matrix (dtype TF_BF16 | TF_F16; TF_F32 keeps tf_stem_im2col + tf_conv2d):
 * y [N*OH*OW][64] = conv(x, W) with w_packed = conv1.weight as tf_pack_weight* writes it for the im2col GEMM ([>= 64 rows][ldw],
 * k = c*49 + kh*7 + kw, ldw >= 160 and a multiple of 8).  epi: 0 | TF_EPI_STATS (stat_out[rows][2][64] += per-channel sum and sum of
 * squares of the fp32 results, rows = *host_rows_out <= tf_get_stat_rows(), zero on entry; TF_ERR_UNSUPPORTED with unfolded rows) |
 * TF_EPI_AFFINE|TF_EPI_RELU (y = relu(conv * scale + shift): the folded BatchNorm of the evaluation graph). */
int tf_stem_conv(int dtype, const float* x_nchw, int N, int H, int W, const void* w_packed, int ldw, void* y, int epi,
                 const float* scale, const float* shift, float* stat_out, int* host_rows_out, void* stream);
/* r4: weight gradient of conv1 straight from the image (autograd of model.py:90; dtype TF_BF16 | TF_F16): dw_oihw[64][147] (fp32, the OIHW
 * order of conv1.weight) += sum over output pixels of g[px][co] * patch[px][k]; g [N*OH*OW][64] of `dtype`; dw holds zeros (or what is to be
 * accumulated into) on entry.  Replaces tf_stem_im2col + tf_conv2d_wgrad over the 147-column matrix.  x_conv != NULL (with cA, cB, cD: 64
 * floats each): the gradient operand is  cA * g + cB * x_conv + cD  per channel, rounded to `dtype` -- tf_bn_bwd_apply of the stem's BatchNorm
 * (x_conv = the conv output the statistics were taken of) folded into the kernel, whose output has no other reader. */
int tf_stem_wgrad(int dtype, const float* x_nchw, int N, int H, int W, const void* g, const void* x_conv, const float* cA, const float* cB,
                  const float* cD, float* dw_oihw, void* stream);
/* nn.MaxPool2d(3,2,1) (model.py:93) with the stem's BN+ReLU fused in front when scale/shift are
 * given (training: the un-normalised conv output is read once).  argmax (u8, optional) feeds _bwd. */
int tf_maxpool_fwd(int dtype, const void* x, int N, int H, int W, int C, const float* scale, const float* shift,
                   void* y, uint8_t* argmax, void* stream);
int tf_maxpool_bwd(int dtype, const void* g, const uint8_t* argmax, const void* x, const float* scale, const float* shift,
                   int N, int H, int W, int C, void* gz, void* stream);
/* r4: tf_maxpool_bwd that also takes the column sums of the stem's BatchNorm backward (autograd of model.py:91-93) in the same pass:
 * stat_out[rows][2][C] += (sum gz, sum gz * x), rows = *host_rows_out <= tf_get_stat_rows(), zero on entry -- what
 * tf_colstats(gz, NULL, x) would compute from a second read of both tensors.  TF_ERR_UNSUPPORTED with unfolded rows (tf_set_stat_rows(0)). */
int tf_maxpool_bwd_stats(int dtype, const void* g, const uint8_t* argmax, const void* x, const float* scale, const float* shift,
                         int N, int H, int W, int C, void* gz, float* stat_out, int* host_rows_out, void* stream);
/* per-channel sums over the rows of an [M][ld] matrix, block partials [nblk][nk][C]:
 * k0 = sum g', k1 = sum g'*a, k2 = sum g'*b, g' = g*(y>0) when y != NULL.  nblk = tf_colstats_blocks(). */
int tf_colstats_blocks(int M, int C, int dtype);
int tf_colstats(int dtype, const void* g, const void* y, const void* a, const void* b, int M, int C, int ld,
                float* partial, void* stream);
/* nn.BatchNorm2d in training mode (torchvision Bottleneck / model.py:91): partial (sum, sumsq)
 * -> y = x*scale + shift, saved mean / invstd, running statistics (momentum, unbiased var). */
int tf_bn_finalize(const float* partial, int nblk, int ld, int C, float count, const float* gamma, const float* beta,
                   float eps, float momentum, float* scale, float* shift, float* mean, float* invstd,
                   float* running_mean, float* running_var, int clear, void* stream);
/* eval-mode BN as a per-channel affine (folded into the conv epilogue) */
int tf_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
               int C, float* scale, float* shift, void* stream);
/* BN backward: partial sums (k0 = sum gz, kidx = sum gz*x) -> dgamma, dbeta and g_x = A*gz + B*x + D */
int tf_bn_bwd_finalize(const float* partial, int nblk, int nk, int kidx, int ld, int C, float count, const float* gamma,
                       const float* mean, const float* invstd, float* dgamma, float* dbeta, float* cA, float* cB,
                       float* cD, int clear, void* stream);
int tf_bn_bwd_apply(int dtype, const void* g, const void* y, const void* x, const float* cA, const float* cB,
                    const float* cD, int64_t M, int C, void* out, void* stream);
/* y = relu(x*scale + shift): BN + ReLU materialised for the 3x3 conv's LDS-DMA operand pipeline */
int tf_bn_relu(int dtype, const void* x, const float* scale, const float* shift, int64_t M, int C, void* y, void* stream);
/* Bottleneck output in training mode: y = relu(x*s1+h1 + (r*s2+h2 | r)) */
int tf_bn_add_relu(int dtype, const void* x, const float* s1, const float* h1, const void* r, const float* s2,
                   const float* h2, int64_t M, int C, void* y, void* stream);
/* ---- training-mode BN consumers that finalize the batch statistics in-kernel (no separate finalize launch) ----
 * The producer (tf_conv2d with TF_EPI_STATS/STATS2, tf_colstats) leaves `rows` = tf_get_stat_rows() <= TF_STAT_ROWS
 * partial rows [rows][nk][C] in a region that was zero before it ran; every block re-derives the coefficients of its
 * 64-channel slice; the first row-block of each slice publishes scale/shift/mean/invstd (+ running statistics update)
 * or dgamma/dbeta.  Same arithmetic as tf_bn_finalize / tf_bn_bwd_finalize (torch.nn.BatchNorm2d training semantics of
 * the torchvision trunk built at tinyfaces/models/model.py:17-23).                                                    */
typedef struct tf_bn_fwd_desc {
  const float* stat;                 /* [rows][2][C] sum, sum of squares */
  const float* gamma; const float* beta;
  float* scale; float* shift; float* mean; float* invstd;        /* published */
  float* running_mean; float* running_var;                        /* updated in place (may be NULL) */
  const float* stat_shift;           /* [C] the shift the producer subtracted (tf_conv_args.stat_shift_out); NULL: none */
} tf_bn_fwd_desc;
typedef struct tf_bn_bwd_desc {
  const float* stat;                 /* [rows][nk][C]: k = 0 sum gz, k = kidx sum gz*x */
  const float* gamma; const float* mean; const float* invstd;
  float* dgamma; float* dbeta;       /* published (may be NULL) */
  int nk, kidx;
} tf_bn_bwd_desc;
int tf_bn_relu_fused(int dtype, const void* x, const tf_bn_fwd_desc* bn, int rows, int64_t M, int C, float count,
                     float eps, float momentum, void* y, void* stream);
int tf_bn_add_relu_fused(int dtype, const void* x, const tf_bn_fwd_desc* bn, const void* r, const tf_bn_fwd_desc* bn_r /* NULL: identity */,
                         int rows, int64_t M, int C, float count, float eps, float momentum, void* y, void* stream);
int tf_bn_bwd_apply_fused(int dtype, const void* g, const void* y /* NULL: no ReLU mask */, const void* x, const tf_bn_bwd_desc* bn,
                          int rows, int64_t M, int C, float count, void* out, void* stream);
#ifdef TF_EXPERIMENTAL
/* EXPERIMENTAL BUILD ONLY (build.py --experimental): parity-green, measured slower than the two launches on every layer,
 * not part of the default library.
 * r3: tf_bn_bwd_apply_fused + the pointwise tf_conv2d that consumes its output, in ONE launch (csrc/conv_pwx.hip): the conv's pixel
 * operand is A*x + B*x2 + D (a->x = the incoming gradient g, x2 = the BatchNorm's input, coefficients from `bn`'s statistic rows as in
 * tf_bn_bwd_apply_fused, dgamma / dbeta published), the applied tensor is also written to applied_out [M][Cin] (NULL: not kept).
 * Replaces the backward of BatchNorm2d followed by the data gradient of the 1x1 conv in front of it (torchvision Bottleneck.bn3 / conv3
 * under autograd, tinyfaces/trainer.py:86).  bf16, 1x1 / stride 1, Cin a multiple of 64 in [128, 1024], Cout a multiple of 128 with
 * ldy == Cout; TF_ERR_UNSUPPORTED otherwise (run the two calls). */
int tf_conv2d_bnbwd(const tf_conv_args* a, const tf_bn_bwd_desc* bn, const void* x2, void* applied_out, int rows, float count, void* stream);
/* r5: tf_bn_add_relu_fused + the pointwise tf_conv2d that consumes its output, in ONE launch (csrc/conv_pwx.hip): the conv's pixel operand
 * is y = relu(bn(a->x) + (bn_res(res) | res)) -- a->x = the raw output of the previous bottleneck's conv3, `bn` its bn3 (batch statistics
 * finalized in-kernel: scale / shift / mean / invstd published, running statistics updated, exactly like tf_bn_add_relu_fused), res = the
 * residual (bn_res != NULL: the raw downsample conv output and its BatchNorm) -- and y is also written to y_out [M][Cin]: the block output.
 * Replaces bn3 -> += identity -> relu of a torchvision Bottleneck followed by conv1 of the NEXT one (tinyfaces/models/model.py:90-101).
 * Same shape limits and error codes as tf_conv2d_bnbwd. */
int tf_conv2d_bnfwd(const tf_conv_args* a, const tf_bn_fwd_desc* bn, const void* res, const tf_bn_fwd_desc* bn_res, void* y_out, int rows, float count,
                    float eps, float momentum, void* stream);
#endif /* TF_EXPERIMENTAL */
/* score4_upsample (frozen bilinear ConvTranspose2d k4 s2 p1, model.py:34-40,107) + crop (:110-124)
 * + add (:126); wup_diag [C][4][4] = the channel diagonal of the (C,C,4,4) weight; output NCHW fp32. */
int tf_upsample_add_crop(int dtype, const void* s3, const void* s4, const float* wup_diag, int B, int C, int ldc,
                         int H3, int W3, int H4, int W4, float* out_nchw, void* stream);
int tf_upsample_add_crop_bwd(int dtype, const float* g_nchw, const float* wup_diag, int B, int C, int ldc,
                             int H3, int W3, int H4, int W4, void* g3, void* g4, void* stream);
int tf_reduce_partials(const float* partial, int nblk, int nk, int k, int ld, int C, float* out, int clear, void* stream);

/* ---- the detector network as one native graph executor -----------------------------
 * Replaces DetectionModel.forward (tinyfaces/models/model.py:89-128) and its autograd
 * backward (tinyfaces/trainer.py:78,86): ResNet-101 trunk minus layer4, heads, upsample+add.
 * params / grads: tables of device pointers in tf_detnet_param_name(i) order (fp32, the
 * layouts of the reference state_dict: OIHW conv weights, [C] BN vectors).
 * ws: scratch of tf_detnet_workspace_bytes() bytes, carved deterministically per call; the
 * activations a backward needs live there until tf_detnet_backward is called.            */
int tf_detnet_num_params(void);
const char* tf_detnet_param_name(int i);      /* state_dict key, e.g. "model.layer3.22.bn3.running_var" */
int64_t tf_detnet_param_numel(int i, int num_out);
size_t tf_detnet_workspace_bytes(int dtype, int N, int H, int W, int num_out, int training);
int tf_detnet_out_shape(int H, int W, int* H3, int* W3);
/* flags: TF_DETNET_WEIGHTS_READY (eval only) = the front of `ws` (tf_detnet_param_region_bytes bytes, whose layout depends on
 * dtype/num_out/training but not on N,H,W) still holds the packed weights + folded BN of an earlier eval forward with the
 * SAME parameter values: skip re-packing.  The image pyramid of evaluation.py:49-82 runs 3-5 forwards per image on
 * constant weights; the caller owns the guarantee.                                                                       */
#define TF_DETNET_WEIGHTS_READY 1
size_t tf_detnet_param_region_bytes(int dtype, int num_out, int training);
int tf_detnet_forward(int dtype, int training, const float* x_nchw, int N, int H, int W, int num_out,
                      void* const* params, float bn_eps, float bn_momentum,
                      float* out_nchw, void* ws, size_t ws_bytes, int flags, void* stream);
/* ---- executor context + gradient-ready hooks (r4) ------------------------------------------------------------------------------------
 * tf_detnet_ctx owns what the executor needs beyond `ws`: the second HIP stream of the device it was created on (weight gradients of the
 * backward pass run there, concurrently with the data-gradient chain; the weights of layer 3 are packed there beside the start of a
 * training forward) and its pool of fork / join events.  One context per model (or per thread that drives models): two contexts never
 * share a stream or an event, a context must only be used on the device that was current when it was created.  NULL selects the
 * process-wide default context of the current device (the behaviour of rounds 1-3).
 * tf_detnet_hooks is the data-parallel interface of ONE backward call (the reference has no distributed path; this serves the 8-GPU
 * data-parallel row of SURVEY.md section 8e): when the gradients of all bottlenecks >= blocks[k] (and of the heads) are enqueued
 * (blocks[k] = -1: at the very end, stem included), events[k] -- a hipEvent_t, may be NULL -- is recorded on the stream that carries them
 * and fn(blocks[k], stream, user) is called on the calling thread: work the callee enqueues on `stream` (or orders behind it) sees the
 * bucket final, e.g. tf_comm_allreduce_hook below.  BN gamma / beta gradients of a block are written on the caller's stream BEFORE the
 * fork that precedes the event's stream position, so one event covers them too.  single_stream = 1: no second stream (A/B, race tests). */
typedef struct tf_detnet_ctx tf_detnet_ctx;
int tf_detnet_ctx_create(tf_detnet_ctx** out);       /* binds to the CURRENT device */
int tf_detnet_ctx_destroy(tf_detnet_ctx* ctx);       /* waits for the context's streams, then frees them */
typedef void (*tf_grad_ready_fn)(int block, void* stream, void* user);
typedef struct tf_detnet_hooks {
  const int* blocks; void* const* events; int n;
  tf_grad_ready_fn fn; void* user;
  int single_stream;
} tf_detnet_hooks;
int tf_detnet_forward_ctx(tf_detnet_ctx* ctx, int single_stream, int dtype, int training, const float* x_nchw, int N, int H, int W, int num_out,
                          void* const* params, float bn_eps, float bn_momentum, float* out_nchw, void* ws, size_t ws_bytes, int flags, void* stream);
/* grad_flat (optional): when every entry of `grads` lies inside [grad_flat, grad_flat + grad_flat_bytes) the ranges that are accumulated
 * into are zeroed with at most two memsets instead of one per weight gradient. */
int tf_detnet_backward_ctx(tf_detnet_ctx* ctx, const tf_detnet_hooks* hooks /* NULL: none */, int dtype, const float* x_nchw, int N, int H, int W,
                           int num_out, void* const* params, void* const* grads, const float* gout_nchw,
                           void* grad_flat, size_t grad_flat_bytes, void* ws, size_t ws_bytes, void* stream);
/* Context-free forms of rounds 1-3, kept for callers that drive ONE model from ONE thread: they use the default context of the current
 * device and hooks registered PROCESS-WIDE with the three setters below (not thread-safe; the Python surface no longer uses them). */
int tf_detnet_set_dual_stream(int on);               /* 1 (default): second stream; 0: everything on the caller's stream */
int tf_detnet_set_grad_events(const int* blocks, void* const* events, int n);      /* n = 0 clears */
int tf_detnet_set_grad_callback(tf_grad_ready_fn fn, void* user);                  /* NULL: off */
int tf_detnet_backward(int dtype, const float* x_nchw, int N, int H, int W, int num_out,
                       void* const* params, void* const* grads, const float* gout_nchw,
                       void* grad_flat, size_t grad_flat_bytes,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- gradient exchange of the data-parallel path over RCCL (the reference has no distributed code) -------------
 * One process per GPU; the gradients are SUMMED over the ranks bucket by bucket while the backward pass runs, the 1/world goes into
 * tf_sgd_step's grad_scale.  librccl.so is resolved at run time (the copy already mapped into the process first): tf_comm_available() == 0
 * on a host without it, and every other entry then returns TF_ERR_UNSUPPORTED.
 *   tf_comm_unique_id  rank 0 draws the 128-byte identifier of a new communicator (HOST memory); ship it to the other ranks out of band
 *   tf_comm_init       collective over all ranks; binds to the CURRENT device; owns a communication stream (default priority) + events
 *   tf_allreduce_bucket  buf[0..n) <- sum over ranks, in place, on the communicator's stream, ordered behind the current tail of `after`
 *                        (the executor stream that carries the bucket: tf_detnet_hooks) without holding that stream up
 *   tf_comm_join       `stream` waits for every collective issued so far (call it on the training stream before tf_sgd_step)
 *   tf_comm_allreduce_hook  a ready-made tf_grad_ready_fn: tf_detnet_hooks.fn = tf_comm_allreduce_hook, .user = a tf_comm_plan that maps
 *                        the registered blocks to element ranges of the flat gradient; plan.rc keeps the first error, plan.issued counts,
 *                        plan.status[k] says which buckets were issued (a caller with another exchange at hand reduces the others itself) */
#define TF_COMM_ID_BYTES 128
typedef struct tf_comm tf_comm;
int tf_comm_available(void);
int tf_comm_unique_id(void* host_id_out /*[TF_COMM_ID_BYTES]*/);
int tf_comm_init(const void* host_id /*[TF_COMM_ID_BYTES]*/, int rank, int world, tf_comm** out);
int tf_comm_destroy(tf_comm* comm);
int tf_comm_rank(const tf_comm* comm);
int tf_comm_world(const tf_comm* comm);
int tf_allreduce_bucket(tf_comm* comm, float* buf, size_t n, void* after_stream);
int tf_comm_join(tf_comm* comm, void* stream);
typedef struct tf_comm_plan {
  void* comm;                    /* tf_comm* */
  float* grad_flat;              /* the flat fp32 gradient the buckets are slices of */
  int n; const int* blocks;      /* the blocks registered in tf_detnet_hooks (backward order; -1 = the end of the pass) */
  const int64_t* start; const int64_t* end;      /* element range of bucket k */
  int rc, issued;                /* out: first error (TF_OK), collectives issued since the caller reset it */
  int* status;                   /* r6, optional [n], out: 1 = bucket k's collective was issued, < 0 = its error code, untouched = hook not reached */
} tf_comm_plan;
void tf_comm_allreduce_hook(int block, void* stream, void* user /* tf_comm_plan* */);

/* ---- image preparation in front of the detector -----------------------------------
 * One pass from the decoded uint8 RGB image to the normalised fp32 CHW tensor: PIL BILINEAR resize (bit-exact with Pillow's
 * 8-bit two-pass resample; tinyfaces/datasets/wider_face.py:136-146 and tinyfaces/evaluation.py:46 through
 * torchvision.transforms.functional.resize), crop + paste on the mean colour (tinyfaces/datasets/processor.py:41-76),
 * horizontal flip (tinyfaces/datasets/wider_face.py:155-157), ToTensor + Normalize (main.py:44-46).  Only the pixels of the
 * window are resampled.  Training: OH = OW = 500; evaluation: crop = the whole resized image, paste (0,0), OH x OW = RH x RW. */
typedef struct tf_image_prepare_args {
  const unsigned char* img;              /* device, [H][W][3] uint8 */
  int H, W;                              /* decoded size */
  int RH, RW;                            /* size after the resize (== H, W: no resize) */
  int crop_y, crop_x, crop_h, crop_w;    /* window of the RESIZED image */
  int paste_y, paste_x;                  /* top-left corner of the window in the output */
  int flip;                              /* 1: the finished buffer is mirrored (np.fliplr) */
  int OH, OW;                            /* output size */
  float mean[3], std[3];                 /* Normalize */
  unsigned char bg[3];                   /* colour outside the window: (mean * 255) truncated = 123, 116, 103 */
  float* out;                            /* device, fp32 [3][OH][OW] */
} tf_image_prepare_args;
int tf_image_prepare(const tf_image_prepare_args* a, void* stream);

/* ---- measurement hooks (bench.py `roofline`) ------------------------------------------
 * While enabled, every MFMA kernel launch (conv_igemm / wgrad) is bracketed by HIP events on
 * the stream it is launched on.  tf_profile_collect blocks until they completed and writes
 * rows of 6 doubles to HOST memory: kind, launches, total_ms, ALGORITHMIC flops (2 x the MACs of the
 * forward convolution the launch belongs to, unpadded channels: a stride-2 data gradient counts the
 * forward conv's MACs, not the zero-inserted gather it executes), algorithmic bytes, EXECUTED flops
 * (2*M*N*K of the GEMM the kernel ran, padding and zero taps included).  kind 0..5 = conv_igemm (dtype*3 + tile-1), 8..11 = wgrad (8 + dtype*2 + (tile==128)),
 * 12/13/15 = conv_dma f32/bf16/f16, 14 = wgrad_dma bf16, 6/7 = conv3x3h bf16/f16, 16 = wgrad3x3 bf16 (both launches of its two-phase form), 17 = conv_pwx,
 * 18 / 19 = the grouped pointwise / 3x3 weight gradients (tf_conv2d_wgrad_group: flops and bytes of the whole group). */
int tf_profile_enable(int every);   /* 0 = off, 1 = bracket every launch, n = every n-th launch (sampling keeps the timed region undisturbed) */
int tf_profile_collect(double* host_out, int max_rows);
/* per layer shape, for the records consumed by the LAST tf_profile_collect: rows of 12 doubles
 * (kind, M pixels, N output channels, K reduction length, taps, mode (0 fwd, 1 dgrad, 2 wgrad), epilogue flags, launches, total_ms,
 *  algorithmic flops, algorithmic bytes, executed flops) */
int tf_profile_shapes(double* host_out, int max_rows);
/* (r4: the debugging / measurement probes -- tf_debug_conv3x3h_trace, tf_debug_probe, tf_debug_probe_chain, tf_probe_tr16 -- are not part of
 * this ABI any more: tiny-faces-pytorch_amd/csrc/debug_api.h, bound by the test-suite and scripts/ only.) */

#ifdef __cplusplus
}
#endif
#endif /* TINYFACES_HIP_H */
