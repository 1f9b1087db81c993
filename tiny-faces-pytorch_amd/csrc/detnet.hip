// The detector network as a native graph executor: every kernel launch of a forward or a
// backward pass is issued from here (one C-ABI call per pass, no Python in the loop), with all
// activations carved from one caller-provided arena.
// Replaces DetectionModel.forward (tinyfaces/models/model.py:89-128) over the torchvision
// ResNet-101 trunk (Bottleneck x [3,4,23], stride on the 3x3) and the autograd backward that
// tinyfaces/trainer.py:86 triggers.
//
// Data layout in HBM: activations NHWC ("pixels x channels"), dtype bf16 (fast) or fp32
// (parity); BN vectors, statistics and all gradients fp32; weights re-packed once per training
// step (or once per constant_weights() session in eval) from the fp32 OIHW master copy into
// K-contiguous [Cout][tap][Cin] (forward / wgrad operand) and [Cin][tap][Cout] (data-gradient
// operand), both from one read of the master (tf_pack_weights_tiled).
//
// Fusion plan (what never makes an HBM round trip, and what never costs a launch):
//   eval : BN folded to a per-channel affine inside every conv epilogue, + residual + ReLU.
//   train: conv epilogues fold per-tile (sum, sumsq) into the BN's own statistic rows; the
//          elementwise consumer (bn_relu_fused -> a1/a2, bn_add_relu_fused -> y) finalizes them
//          in-kernel, so a bottleneck is 6 launches.  a1/a2 are materialised on purpose: every
//          conv and every weight gradient then runs on the LDS-DMA pipeline (no prologue).
//   backward: dgrad epilogues apply the ReLU mask and fold the BN-backward sums; the last data
//          gradient of a block hands the previous block g*(y>0) AND its BN3-backward sums
//          (RES|MASK2|STATS3); bn_bwd_apply_fused finalizes in-kernel; weight gradients run on
//          the context's second stream against per-block operand buffers (r4) -- per launch for
//          layers 1-2, deferred and grouped (tf_conv2d_wgrad_group: 8 + 7 + 7 blocks) for the
//          identity bottlenecks of layer 3.
//   stem (r4): conv1 and its weight gradient straight from the NCHW image (stem_conv.hip), the
//          BN-backward sums inside the max-pool backward, the apply inside the weight gradient.
//   tf_set_stat_rows(0) (unfolded, bit-reproducible statistics) falls back to the separate
//   finalize kernels and the unmasked gradient flow: the A/B and race-screen path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "common.h"
#include "tuning.h"

extern "C" {
int tf_stem_im2col(const float*, int, int, int, int, void*, int, void*);
int tf_stem_conv(int, const float*, int, int, int, const void*, int, void*, int, const float*, const float*, float*, int*, void*);
int tf_stem_wgrad(int, const float*, int, int, int, const void*, const void*, const float*, const float*, const float*, float*, void*);
int tf_maxpool_fwd(int, const void*, int, int, int, int, const float*, const float*, void*, uint8_t*, void*);
int tf_maxpool_bwd(int, const void*, const uint8_t*, const void*, const float*, const float*, int, int, int, int, void*, void*);
int tf_maxpool_bwd_stats(int, const void*, const uint8_t*, const void*, const float*, const float*, int, int, int, int, void*, float*, int*, void*);
int tf_colstats_blocks(int, int, int);
int tf_colstats(int, const void*, const void*, const void*, const void*, int, int, int, float*, void*);
int tf_bn_finalize(const float*, int, int, int, float, const float*, const float*, float, float, float*, float*, float*, float*, float*,
                   float*, int, void*);
int tf_bn_fold(const float*, const float*, const float*, const float*, float, int, float*, float*, void*);
int tf_bn_bwd_finalize(const float*, int, int, int, int, int, float, const float*, const float*, const float*, float*, float*, float*, float*,
                       float*, int, void*);
int tf_bn_bwd_apply(int, const void*, const void*, const void*, const float*, const float*, const float*, int64_t, int, void*, void*);
int tf_bn_add_relu(int, const void*, const float*, const float*, const void*, const float*, const float*, int64_t, int, void*, void*);
int tf_bn_relu(int, const void*, const float*, const float*, int64_t, int, void*, void*);
int tf_upsample_add_crop(int, const void*, const void*, const float*, int, int, int, int, int, int, int, float*, void*);
int tf_upsample_add_crop_bwd(int, const float*, const float*, int, int, int, int, int, int, int, void*, void*, void*);
int tf_reduce_partials(const float*, int, int, int, int, int, float*, int, void*);
int tf_conv2d_wgrad_group(const tf_wgrad_args*, int, void*);
}
// (conv_pwx.hip; TF_ERR_UNSUPPORTED in the default build: the C entry points tf_conv2d_bnbwd / tf_conv2d_bnfwd exist in the experimental build only)
int tf_conv_pwx_launch(const tf_conv_args*, const tf_bn_bwd_desc*, const void*, void*, int, float, hipStream_t);
int tf_conv_pwx_launch_fwd(const tf_conv_args*, const tf_bn_fwd_desc*, const void*, const tf_bn_fwd_desc*, void*, int, float, float, float, hipStream_t);

namespace {

constexpr int kStemK = 192;     // 147 taps*channels padded to 3 x 64
// r4: conv1 AND its weight gradient straight from the image (csrc/stem_conv.hip: tf_stem_conv, tf_stem_wgrad) for the 2-byte operand types:
// no 288 MB im2col matrix in a training step or an evaluation forward.  fp32 and the unfolded-statistics mode keep im2col + GEMM.
// TINYFACES_STEM_DIRECT_OFF=1: the path of rounds 1-3; TINYFACES_STEM_WGRAD_IM2COL=1: only the weight gradient over the im2col matrix (built on
// the second stream at the top of the backward pass).
bool stem_direct_mode(int dtype, bool training, bool fused) {
  const bool off = tf::tuning().stem_direct_off;
  return !off && dtype != TF_F32 && (!training || fused);
}
constexpr int kHeadLd = 128;    // 125 outputs padded

struct ConvUnit {               // one conv + (optional) BN, with indices into the parameter table
  std::string name;             // e.g. "model.layer1.0.conv1"
  int cin, cout, k, stride, pad;
  int w, gamma, beta, rmean, rvar;   // param-table indices (-1 if absent)
  int bias;                          // heads only
};
struct Block { ConvUnit c1, c2, c3, ds; bool has_ds; int planes, stride, cin; };
struct Arch {
  ConvUnit stem;
  std::vector<Block> blocks;    // 30 bottlenecks
  int layer_end[3];             // index of the last block of layer1/2/3
  ConvUnit head3, head4;
  int upsample_w;
  std::vector<std::string> names;
};

int add_param(Arch& a, const std::string& n) { a.names.push_back(n); return (int)a.names.size() - 1; }

ConvUnit make_unit(Arch& a, const std::string& conv, const std::string& bn, int cin, int cout, int k, int stride, int pad) {
  ConvUnit u;
  u.name = conv; u.cin = cin; u.cout = cout; u.k = k; u.stride = stride; u.pad = pad; u.bias = -1;
  u.w = add_param(a, conv + ".weight");
  u.gamma = add_param(a, bn + ".weight"); u.beta = add_param(a, bn + ".bias");
  u.rmean = add_param(a, bn + ".running_mean"); u.rvar = add_param(a, bn + ".running_var");
  return u;
}

const Arch& arch() {
  static Arch a = [] {
    Arch a;
    a.stem = make_unit(a, "model.conv1", "model.bn1", 3, 64, 7, 2, 3);
    const int nblk[3] = {3, 4, 23}, planes[3] = {64, 128, 256};
    int inpl = 64;
    for (int L = 0; L < 3; ++L) {
      for (int b = 0; b < nblk[L]; ++b) {
        Block B;
        const std::string p = "model.layer" + std::to_string(L + 1) + "." + std::to_string(b);
        B.planes = planes[L]; B.stride = (b == 0 && L > 0) ? 2 : 1; B.cin = inpl;
        B.c1 = make_unit(a, p + ".conv1", p + ".bn1", inpl, planes[L], 1, 1, 0);
        B.c2 = make_unit(a, p + ".conv2", p + ".bn2", planes[L], planes[L], 3, B.stride, 1);
        B.c3 = make_unit(a, p + ".conv3", p + ".bn3", planes[L], planes[L] * 4, 1, 1, 0);
        B.has_ds = (b == 0);
        if (B.has_ds) B.ds = make_unit(a, p + ".downsample.0", p + ".downsample.1", inpl, planes[L] * 4, 1, B.stride, 0);
        inpl = planes[L] * 4;
        a.blocks.push_back(B);
      }
      a.layer_end[L] = (int)a.blocks.size() - 1;
    }
    auto head = [&](const std::string& n, int cin) {
      ConvUnit u; u.name = n; u.cin = cin; u.cout = -1; u.k = 1; u.stride = 1; u.pad = 0;
      u.w = add_param(a, n + ".weight"); u.bias = add_param(a, n + ".bias");
      u.gamma = u.beta = u.rmean = u.rvar = -1;
      return u;
    };
    a.head3 = head("score_res3", 512);
    a.head4 = head("score_res4", 1024);
    a.upsample_w = add_param(a, "score4_upsample.weight");
    return a;
  }();
  return a;
}

// r4: the weight gradients of the identity bottlenecks of layer 3 (22 blocks of identical shape) are differentiated in GROUPS of up to
// this many bottlenecks per launch (tf_conv2d_wgrad_group: full-K tiles, no split-K / atomics / partial tiles); their dY operands then
// live in per-block buffers instead of the two parity sets.  TINYFACES_WGRAD_GROUP=0: the per-block launches of rounds 1-3.
inline int wgrad_group_size() {
  const int g = tf::tuning().wgrad_group;
  return g;
}
inline bool wgrad_group_mode(int dtype, int training) { return training && dtype == TF_BF16 && wgrad_group_size() > 0; }

inline int down2(int n) { return (n - 1) / 2 + 1; }   // every stride-2 stage of the trunk: ceil(n/2)
inline size_t esize(int dtype) { return dtype == TF_F32 ? 4 : 2; }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Arena {
  char* base; size_t cap, off; bool ok;
  Arena(void* b, size_t c) : base((char*)b), cap(c), off(0), ok(true) {}
  void* get(size_t bytes) {
    const size_t o = off; off += up256(bytes);
    if (off > cap && base) ok = false;
    return base ? base + o : nullptr;
  }
  float* f32(size_t n) { return (float*)get(n * 4); }
};

// per-BN scratch: scale/shift (forward affine), mean/invstd, backward coefficients
// fst / bst: this BN's own statistic rows ([TF_STAT_ROWS][2][C] forward sums, [TF_STAT_ROWS][3][C] backward sums) for the
// consumers that finalize in-kernel (bn_fused.hip); all of them are zeroed by ONE memset per pass
struct BnBuf { float *scale, *shift, *mean, *invstd, *cA, *cB, *cD, *fst, *bst; int C; };
BnBuf bn_alloc(Arena& ar, int C) {
  BnBuf b; float* p = ar.f32((size_t)7 * C);
  b.scale = p; b.shift = p + C; b.mean = p + 2 * C; b.invstd = p + 3 * C; b.cA = p + 4 * C; b.cB = p + 5 * C; b.cD = p + 6 * C;
  b.fst = b.bst = nullptr; b.C = C;
  return b;
}

struct Plan {                    // everything a forward carves; backward re-derives the same pointers
  int dtype, N, H, W, nout, training;
  int H1, W1, H2, W2;            // stem conv out, maxpool out
  void *col, *cstem, *pool, *a1tmp; uint8_t* pool_idx; BnBuf bn_stem;
  void *wstem;
  struct Blk {
    int Hin, Win, Hout, Wout;
    void *c1, *c2, *c3, *d, *y;
    void *a1, *a2;                   // relu(bn1(c1)), relu(bn2(c2)) materialised in training (operands of conv2/conv3 and of their weight gradients)
    void *w1, *w2, *w3, *wd;         // packed forward weights
    void *w1t, *w2t, *w3t, *wdt;     // packed data-gradient (transposed) weights, training only
    void *gT1, *gT2, *gU1, *gT3;     // r4: this block's own g_c3 / g_c2 / g_c1 / downsample-branch gradient (alive until its weight gradients ran)
    BnBuf b1, b2, b3, bd;
  };
  std::vector<Blk> blk;
  void *w_h3, *w_h4, *w_h3t, *w_h4t, *s3, *s4; float *hbias3, *hbias4, *ones, *wup_diag;
  float *partial, *partial_b; size_t partial_floats;      // per-tile partial rows of the forward / backward pass (each behind its statistic region)
  float *stat_fwd, *stat_bwd; size_t stat_fwd_floats, stat_bwd_floats;   // per-BN statistic regions (fused finalize)
  // backward-only
  void *g3, *g4, *G0, *G1, *T4, *R3, *wt; float* dwp; size_t dwp_floats;
  int H3, W3, H4, W4;
  size_t total, param_bytes;
};

size_t packed_bytes(int dtype, int rows, int taps, int cols) { return (size_t)((rows + 127) / 128 * 128) * taps * cols * esize(dtype); }

void build_plan(Plan& P, Arena& ar, int dtype, int N, int H, int W, int nout, int training) {
  const Arch& A = arch();
  const size_t es = esize(dtype);
  P.dtype = dtype; P.N = N; P.H = H; P.W = W; P.nout = nout; P.training = training;
  P.H1 = down2(H); P.W1 = down2(W); P.H2 = down2(P.H1); P.W2 = down2(P.W1);
  const size_t M1 = (size_t)N * P.H1 * P.W1, M2 = (size_t)N * P.H2 * P.W2;
  // rows of per-tile partial sums the unfolded (tf_set_stat_rows(0)) flow and colstats need: known from the shapes alone
  size_t max_partial = 0;
  {
    auto part = [&](size_t M, int C) { size_t t = ((M + 63) / 64) * 2 * (size_t)C; if (t > max_partial) max_partial = t; };
    part(M1, 64);
    int h = P.H2, w = P.W2;
    for (size_t i = 0; i < A.blocks.size(); ++i) {
      const Block& B = A.blocks[i];
      const int ho = B.stride == 2 ? down2(h) : h, wo = B.stride == 2 ? down2(w) : w;
      part((size_t)N * h * w, B.planes); part((size_t)N * ho * wo, B.planes * 4);
      h = ho; w = wo;
      if ((int)i == A.layer_end[1]) part((size_t)N * h * w, kHeadLd);
    }
    if (max_partial < (size_t)1100 * 3 * 1024) max_partial = (size_t)1100 * 3 * 1024;   // colstats: up to ~1024 blocks x 3 sums x 1024 channels
  }
  P.partial_floats = max_partial + 4096;
  // ---- parameter-derived buffers first: their offsets depend on (dtype, nout, training) only, never on the image size, so an
  //      eval-mode caller may keep them across forwards of different sizes (TF_DETNET_WEIGHTS_READY)
  P.bn_stem = bn_alloc(ar, 64);
  P.wstem = ar.get(packed_bytes(dtype, 64, 1, kStemK));
  P.blk.resize(A.blocks.size());
  size_t max_wt = 0;
  for (size_t i = 0; i < A.blocks.size(); ++i) {
    const Block& B = A.blocks[i];
    Plan::Blk& b = P.blk[i];
    const int pl = B.planes, c4 = pl * 4;
    b.w1 = ar.get(packed_bytes(dtype, pl, 1, B.cin)); b.w2 = ar.get(packed_bytes(dtype, pl, 9, pl));
    b.w3 = ar.get(packed_bytes(dtype, c4, 1, pl));
    b.wd = B.has_ds ? ar.get(packed_bytes(dtype, c4, 1, B.cin)) : nullptr;
    b.w1t = training ? ar.get(packed_bytes(dtype, B.cin, 1, pl)) : nullptr;
    b.w2t = training ? ar.get(packed_bytes(dtype, pl, 9, pl)) : nullptr;
    b.w3t = training ? ar.get(packed_bytes(dtype, pl, 1, c4)) : nullptr;
    b.wdt = (training && B.has_ds) ? ar.get(packed_bytes(dtype, B.cin, 1, c4)) : nullptr;
    b.b1 = bn_alloc(ar, pl); b.b2 = bn_alloc(ar, pl); b.b3 = bn_alloc(ar, c4);
    if (B.has_ds) b.bd = bn_alloc(ar, c4);
    const size_t wtb = packed_bytes(dtype, pl, 9, pl);
    if (wtb > max_wt) max_wt = wtb;
    if (packed_bytes(dtype, B.cin, 1, c4) > max_wt) max_wt = packed_bytes(dtype, B.cin, 1, c4);
  }
  P.w_h3 = ar.get(packed_bytes(dtype, kHeadLd, 1, 512)); P.w_h4 = ar.get(packed_bytes(dtype, kHeadLd, 1, 1024));
  P.w_h3t = training ? ar.get(packed_bytes(dtype, 512, 1, kHeadLd)) : nullptr;
  P.w_h4t = training ? ar.get(packed_bytes(dtype, 1024, 1, kHeadLd)) : nullptr;
  P.hbias3 = ar.f32(kHeadLd); P.hbias4 = ar.f32(kHeadLd); P.ones = ar.f32(kHeadLd);
  P.wup_diag = ar.f32((size_t)nout * 16);
  P.param_bytes = ar.off;
  // ---- per-BN statistic regions, contiguous so that one memset per pass clears them
  {
    std::vector<BnBuf*> bns;
    for (size_t i = 0; i < A.blocks.size(); ++i) {
      Plan::Blk& b = P.blk[i];
      bns.push_back(&b.b1); bns.push_back(&b.b2); bns.push_back(&b.b3);
      if (A.blocks[i].has_ds) bns.push_back(&b.bd);
    }
    size_t nf = 0, nb = 0;
    // forward region of a BN: TF_STAT_ROWS x (sum, sum of squares) + ONE row holding the shift the producer subtracted (r3)
    for (BnBuf* q : bns) { nf += (size_t)(TF_STAT_ROWS * 2 + 1) * q->C; nb += (size_t)TF_STAT_ROWS * 3 * q->C; }
    P.stat_fwd_floats = training ? nf : 0; P.stat_bwd_floats = training ? nb : 0;
    // r4: [forward statistic rows | forward partial rows] and [backward statistic rows | backward partial rows] are contiguous pairs, so that
    // ONE memset per pass clears the statistic region AND the head of the pass's partial buffer (two dispatches per pass in rounds 1-3)
    P.stat_fwd = ar.f32(P.stat_fwd_floats); P.partial = ar.f32(P.partial_floats);
    P.stat_bwd = ar.f32(P.stat_bwd_floats); P.partial_b = training ? ar.f32(P.partial_floats) : nullptr;
    size_t of = 0, ob = 0;
    for (BnBuf* q : bns) {
      q->fst = training && P.stat_fwd ? P.stat_fwd + of : nullptr; q->bst = training && P.stat_bwd ? P.stat_bwd + ob : nullptr;
      of += (size_t)(TF_STAT_ROWS * 2 + 1) * q->C; ob += (size_t)TF_STAT_ROWS * 3 * q->C;
    }
  }
  // ---- activations
  P.col = ar.get(M1 * kStemK * es);
  P.cstem = ar.get(M1 * 64 * es);
  P.pool = ar.get(M2 * 64 * es);
  P.pool_idx = (uint8_t*)ar.get(training ? M2 * 64 : 0);
  int h = P.H2, w = P.W2;
  size_t max_act = M1 * 64 * es;
  for (size_t i = 0; i < A.blocks.size(); ++i) {
    const Block& B = A.blocks[i];
    Plan::Blk& b = P.blk[i];
    b.Hin = h; b.Win = w; b.Hout = B.stride == 2 ? down2(h) : h; b.Wout = B.stride == 2 ? down2(w) : w;
    const size_t Min = (size_t)N * h * w, Mout = (size_t)N * b.Hout * b.Wout;
    const int pl = B.planes, c4 = pl * 4;
    b.c1 = ar.get(Min * pl * es); b.c2 = ar.get(Mout * pl * es);
    b.a1 = training ? ar.get(Min * pl * es) : nullptr; b.a2 = training ? ar.get(Mout * pl * es) : nullptr;
    b.c3 = training ? ar.get(Mout * c4 * es) : nullptr;
    b.d = B.has_ds ? ar.get(Mout * c4 * es) : nullptr;
    b.y = ar.get(Mout * c4 * es);
    // r4: every block owns its backward operands (g_c3, g_c2, g_c1, downsample-branch gradient).  Rounds 1-3 shared two parity sets and
    // made the chain wait for the weight gradients of block i+2 before it reused their buffers: with the second stream behind by a group
    // launch the chain stalled there (79 us at the top of layer 2, profiles/r04_step_timeline.txt).  0.85 GB of 288 (bf16, bs = 12).
    b.gT1 = b.gT2 = b.gU1 = b.gT3 = nullptr;
    if (training) {
      b.gT1 = ar.get(Mout * c4 * es); b.gT2 = ar.get(Mout * pl * es); b.gU1 = ar.get(Min * pl * es);
      if (B.has_ds) b.gT3 = ar.get(Mout * c4 * es);
    }
    if (Min * (size_t)B.cin * es > max_act) max_act = Min * B.cin * es;
    if (Mout * c4 * es > max_act) max_act = Mout * c4 * es;
    h = b.Hout; w = b.Wout;
  }
  const Plan::Blk& l2 = P.blk[A.layer_end[1]];
  const Plan::Blk& l3 = P.blk[A.layer_end[2]];
  P.H3 = l2.Hout; P.W3 = l2.Wout; P.H4 = l3.Hout; P.W4 = l3.Wout;
  const size_t M3 = (size_t)N * P.H3 * P.W3, M4 = (size_t)N * P.H4 * P.W4;
  P.a1tmp = nullptr;
  P.s3 = ar.get(M3 * kHeadLd * es); P.s4 = ar.get(M4 * kHeadLd * es);
  if (training) {
    P.g3 = ar.get(M3 * kHeadLd * es); P.g4 = ar.get(M4 * kHeadLd * es);
    P.G0 = ar.get(max_act); P.G1 = ar.get(max_act); P.T4 = ar.get(max_act); P.R3 = ar.get(max_act);
    if (packed_bytes(dtype, 1024, 1, kHeadLd) > max_wt) max_wt = packed_bytes(dtype, 1024, 1, kHeadLd);
    P.wt = ar.get(max_wt);
    // scratch of the 3x3 weight gradients: the partial [slice][tile][tap][64][64] tiles of the all-taps kernel (one block per CU:
    // at most 256 + 16 tiles of 144 KiB) or, for the stride-2 convs, the packed [Cout][tap][Cin] gradient of the per-tap kernel
    P.dwp_floats = (size_t)(256 + 16) * 9 * 64 * 64;
    P.dwp = ar.f32(P.dwp_floats);
  } else {
    P.g3 = P.g4 = P.G0 = P.G1 = P.T4 = P.R3 = P.wt = nullptr; P.dwp = nullptr; P.dwp_floats = 0;
  }
  P.total = ar.off;
}

__global__ void head_vectors_kernel(const float* b3, const float* b4, const float* wup, int nout, float* hb3, float* hb4, float* ones,
                                    float* diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kHeadLd) { hb3[i] = i < nout ? b3[i] : 0.f; hb4[i] = i < nout ? b4[i] : 0.f; ones[i] = 1.f; }
  if (i < nout * 16) { const int c = i / 16, k = i % 16; diag[i] = wup[((size_t)c * nout + c) * 16 + k]; }
}

struct Ctx {
  int dtype; hipStream_t stream; void* const* params; void* const* grads; int rc;
  bool grads_zeroed = false;
  bool skip_fold = false;
  std::vector<tf_pack_job> jobs;
  std::vector<tf_pack2_job> jobs2;
  hipStream_t side = nullptr;                 // weight gradients run here, concurrently with the data-gradient chain
  hipStream_t gside = nullptr;                // r4: the GROUPED weight gradients of layer 3 (two launches of 150-300 us per group): a queue of their own, so that the
                                              // per-block gradients of layer 3.0 / layers 1-2 -- whose completion the chain waits for two blocks later (parity buffers) --
                                              // do not queue up behind a group
  std::vector<hipEvent_t>* events = nullptr; size_t ev_next = 0;
  hipEvent_t next_event() {
    if (ev_next == events->size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { chk(TF_ERR_LAUNCH); return nullptr; } events->push_back(e); }
    return (*events)[ev_next++];
  }
  // fork without a packet of its own on `stream`: the NEXT kernel launched through TF_LAUNCH_WITH_STOP_EVENT (the fused BN-backward
  // apply that produces a weight gradient's dY operand) carries the event as its completion signal; join_side() then makes the
  // weight-gradient stream wait for it.  hipEventRecord costs a barrier packet = an ~8 us bubble on the data-gradient chain
  // (profiles/r02_step_timeline.txt), three to four times per bottleneck.
  hipEvent_t pending = nullptr;
  static bool kernel_events() { return !tf::tuning().fork_by_record; }
  void arm_fork() { if (!side || !kernel_events()) return; pending = next_event(); if (pending) tf::set_next_stop_event(pending); }
  void fork_armed(hipStream_t to = nullptr) {
    if (!side) return;
    if (!to) to = side;
    if (tf::take_next_stop_event()) pending = nullptr;     // armed but no kernel took it (the producer refused its arguments): plain fork
    if (pending) { (void)hipStreamWaitEvent(to, pending, 0); pending = nullptr; }
    else { hipEvent_t e = next_event(); if (e) { (void)hipEventRecord(e, stream); (void)hipStreamWaitEvent(to, e, 0); } }
  }
  hipStream_t gstream() const { return gside ? gside : wstream(); }
  hipEvent_t mark(hipStream_t s) { if (!side || !s) return nullptr; hipEvent_t e = next_event(); if (e) (void)hipEventRecord(e, s); return e; }
  // everything enqueued on `stream` so far becomes a dependency of what is enqueued on `side` next
  void fork() { if (!side) return; hipEvent_t e = next_event(); if (e) { (void)hipEventRecord(e, stream); (void)hipStreamWaitEvent(side, e, 0); } }
  hipEvent_t mark_side() { if (!side) return nullptr; hipEvent_t e = next_event(); if (e) (void)hipEventRecord(e, side); return e; }
  void wait_on_main(hipEvent_t e) { if (e) (void)hipStreamWaitEvent(stream, e, 0); }
  hipStream_t wstream() const { return side ? side : stream; }
  void flush_packs(hipStream_t s = nullptr) {
    if (!s) s = stream;
    if (!jobs.empty()) { chk(tf_pack_weights_batched(dtype, jobs.data(), (int)jobs.size(), s)); jobs.clear(); }
    if (!jobs2.empty()) { chk(tf_pack_weights_tiled(dtype, jobs2.data(), (int)jobs2.size(), s)); jobs2.clear(); }
  }
  const float* P(int i) const { return (const float*)params[i]; }
  float* G(int i) const { return grads ? (float*)grads[i] : nullptr; }
  void chk(int r) { if (r != TF_OK && rc == TF_OK) rc = r; }
};

void conv_fill(tf_conv_args& a, int dtype, int mode, int N, int H, int W, int Cin, int OH, int OW, int Cout, int k, int stride, int pad,
               int ldy, const void* x, const void* w, void* y) {
  memset(&a, 0, sizeof(a));
  a.dtype = dtype; a.mode = mode; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.KH = k; a.KW = k;
  a.stride = stride; a.pad = pad; a.ldy = ldy; a.x = x; a.w = w; a.y = y;
}

void pack(Ctx& c, const ConvUnit& u, int cout, void* out, bool transpose, int cin_override = 0, int cols_pad_override = 0,
          int k_override = 0) {
  const int cin = cin_override ? cin_override : u.cin;
  const int k = k_override ? k_override : u.k;
  tf_pack_job j;
  j.src = c.P(u.w); j.dst = out; j.cout = cout; j.cin = cin; j.taps = k * k; j.transpose = transpose ? 1 : 0;
  if (!transpose) { j.rows_pad = (cout + 127) / 128 * 128; j.cols_pad = cols_pad_override ? cols_pad_override : cin; }
  else            { j.rows_pad = (cin + 127) / 128 * 128;  j.cols_pad = cols_pad_override ? cols_pad_override : cout; }
  c.jobs.push_back(j);
}

// forward operand [cout pad 128][taps][cin] and (training) data-gradient operand [cin pad 128][taps][cout] in one job
void pack2(Ctx& c, const ConvUnit& u, int cout, void* out, void* out_t, int cin_override = 0, int cols_pad_override = 0, int k_override = 0) {
  const int cin = cin_override ? cin_override : u.cin;
  const int k = k_override ? k_override : u.k;
  tf_pack2_job j;
  j.src = c.P(u.w); j.dst = out; j.dst_t = out_t; j.cout = cout; j.cin = cin; j.taps = k * k;
  j.rows_pad = (cout + 127) / 128 * 128; j.cols_pad = cols_pad_override ? cols_pad_override : cin;
  j.rows_pad_t = (cin + 127) / 128 * 128; j.cols_pad_t = cout;
  c.jobs2.push_back(j);
}

// BN after a conv: eval -> fold running stats; train -> finalize batch partials (+ running update)
void bn_forward(Ctx& c, const ConvUnit& u, int C, BnBuf& b, bool training, const tf_conv_args* conv, float* partial, float count, float eps,
                float mom) {
  if (!training) {
    if (!c.skip_fold) c.chk(tf_bn_fold(c.P(u.gamma), c.P(u.beta), c.P(u.rmean), c.P(u.rvar), eps, C, b.scale, b.shift, c.stream));
  } else {
    c.chk(tf_bn_finalize(partial, tf_conv_mtiles(conv), conv->ldy, C, count, c.P(u.gamma), c.P(u.beta), eps, mom, b.scale, b.shift, b.mean,
                         b.invstd, (float*)c.params[u.rmean], (float*)c.params[u.rvar], 1, c.stream));
  }
}

// r3: the conv that produces a BN's statistics subtracts the BN's running mean before summing (sum (x - s), sum (x - s)^2) and records
// s in the extra row of the BN's statistic region, where the consumer's table picks it up (fused flow only; bn_fused.hip fwd_table).
// E[x^2] - mean^2 from fp32 sums loses (mean / std)^2 of the significant bits; around the running mean it loses ((mean - s) / std)^2.
void stat_shift(tf_conv_args& a, const Ctx& c, const ConvUnit& u, const BnBuf& b, bool fused) {
  const bool off = tf::tuning().stat_shift_off;       // A/B + parity knob
  if (!fused || off || !b.fst || u.rmean < 0) return;
  a.stat_shift = (const float*)c.params[u.rmean];
  a.stat_shift_out = b.fst + (size_t)TF_STAT_ROWS * 2 * b.C;
}

tf_bn_fwd_desc fwd_desc(const Ctx& c, const ConvUnit& u, const BnBuf& b) {
  tf_bn_fwd_desc d;
  d.stat = b.fst; d.gamma = c.P(u.gamma); d.beta = c.P(u.beta);
  d.scale = b.scale; d.shift = b.shift; d.mean = b.mean; d.invstd = b.invstd;
  d.running_mean = (float*)c.params[u.rmean]; d.running_var = (float*)c.params[u.rvar];
  d.stat_shift = b.fst ? b.fst + (size_t)TF_STAT_ROWS * 2 * b.C : nullptr;      // zero (memset) unless the producer recorded a shift
  return d;
}
tf_bn_bwd_desc bwd_desc(const Ctx& c, const ConvUnit& u, const BnBuf& b, const float* stat, int nk, int kidx) {
  tf_bn_bwd_desc d;
  d.stat = stat; d.gamma = c.P(u.gamma); d.mean = b.mean; d.invstd = b.invstd; d.dgamma = c.G(u.gamma); d.dbeta = c.G(u.beta);
  d.nk = nk; d.kidx = kidx;
  return d;
}

}  // namespace

extern "C" int tf_detnet_num_params(void) { return (int)arch().names.size(); }
extern "C" const char* tf_detnet_param_name(int i) {
  const Arch& a = arch();
  return (i >= 0 && i < (int)a.names.size()) ? a.names[i].c_str() : nullptr;
}
extern "C" int64_t tf_detnet_param_numel(int i, int nout) {
  const Arch& a = arch();
  auto unit = [&](const ConvUnit& u, int cout) -> int64_t {
    if (i == u.w) return (int64_t)cout * u.cin * u.k * u.k;
    if (i == u.gamma || i == u.beta || i == u.rmean || i == u.rvar || i == u.bias) return cout;
    return -1;
  };
  int64_t r;
  if ((r = unit(a.stem, 64)) >= 0) return r;
  for (const Block& B : a.blocks) {
    if ((r = unit(B.c1, B.planes)) >= 0) return r;
    if ((r = unit(B.c2, B.planes)) >= 0) return r;
    if ((r = unit(B.c3, B.planes * 4)) >= 0) return r;
    if (B.has_ds && (r = unit(B.ds, B.planes * 4)) >= 0) return r;
  }
  if ((r = unit(a.head3, nout)) >= 0) return r;
  if ((r = unit(a.head4, nout)) >= 0) return r;
  if (i == a.upsample_w) return (int64_t)nout * nout * 16;
  return -1;
}

extern "C" int tf_detnet_out_shape(int H, int W, int* H3, int* W3) {
  if (H3) *H3 = down2(down2(down2(H)));
  if (W3) *W3 = down2(down2(down2(W)));
  return TF_OK;
}

extern "C" size_t tf_detnet_workspace_bytes(int dtype, int N, int H, int W, int nout, int training) {
  Plan P; Arena ar(nullptr, 0);
  build_plan(P, ar, dtype, N, H, W, nout, training);
  return P.total + 4096;
}

extern "C" size_t tf_detnet_param_region_bytes(int dtype, int nout, int training) {
  Plan P; Arena ar(nullptr, 0);
  build_plan(P, ar, dtype, 1, 32, 32, nout, training);
  return P.param_bytes;
}

// The executor's second stream (weight gradients of the backward pass; r3: the packing of the layer-3 weights beside the start of the
// forward pass), one per device, created on first use at the DEFAULT priority.
// Rounds 1-2 created it at the LOWEST priority ("the weight gradients are off the critical chain"; +0.4 % on the step, A/B r3: 1135.7 vs
// 1131.6 img/s).  r3 found what that costs as soon as the process has a few more busy queues: with three more streams carrying a token
// kernel per step -- or with RCCL in the process and its collectives overlapping the backward pass, i.e. the data-parallel path -- the
// kernels of BOTH queues stretch ~1.6x and the step takes 16.7-17.6 ms instead of 10.6 (profiles/r03_stream_priority.txt; GPU_MAX_HW_QUEUES,
// creation order and the number of streams of our own made no difference, the priority does).  The round-2 CU-mask experiment
// (hipExtStreamCreateWithCUMask: 603 img/s whatever the mask) looked the same.  TINYFACES_SIDE_PRIO_LOW=1 brings the low priority back.
// ---- r4: the executor's own state lives in an explicit CONTEXT (tf_detnet_ctx): the second stream of the device it was created on (weight
// gradients of the backward pass; the packing of the layer-3 weights beside the start of the forward pass), the optional group stream, the
// pool of fork / join events, and -- per backward call -- the caller's gradient-ready hooks (tf_detnet_hooks).  Rounds 1-3 kept all of this in
// process-wide statics behind an ABI documented as stateless (events created on whichever device ran
// first).  The entry points without a context (tf_detnet_forward / tf_detnet_backward / tf_detnet_set_*) remain as wrappers over ONE
// default context per device and thread-unsafe process-wide hooks, for callers that drive one model from one thread.
// The second stream is created at the DEFAULT priority.  Rounds 1-2 created it at the LOWEST priority ("the weight gradients are off the
// critical chain"; +0.4 % on the step); r3 found what that costs as soon as the process has a few more busy queues -- RCCL with its
// collectives overlapping the backward pass, i.e. the data-parallel path -- the kernels of BOTH queues stretch ~1.6x
// (profiles/r03_stream_priority.txt).  TINYFACES_SIDE_PRIO_LOW=1 brings the low priority back.
struct tf_detnet_ctx {
  int device = -1;
  hipStream_t side = nullptr, gside = nullptr;
  hipEvent_t pack_fork = nullptr, pack_join = nullptr, pack_join0 = nullptr;
  std::vector<hipEvent_t> events;
};
namespace {
hipStream_t make_stream() {
  hipStream_t s = nullptr;
  int least = 0, greatest = 0;
  const bool low_prio = tf::tuning().side_prio_low;      // A/B knob (the default of rounds 1-2)
  if (!low_prio || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
      hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) {
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
  }
  return s;
}
// streams and events are created lazily, on the device the context belongs to (the caller has made it current: the binding wraps every
// call in torch.cuda.device(x.device))
hipStream_t ctx_side(tf_detnet_ctx* x) { if (x && !x->side) x->side = make_stream(); return x ? x->side : nullptr; }
hipStream_t ctx_gside(tf_detnet_ctx* x) { if (x && !x->gside) x->gside = make_stream(); return x ? x->gside : nullptr; }
tf_detnet_ctx* default_ctx() {            // one per device: what the context-free entry points use
  static tf_detnet_ctx* per_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!per_dev[dev]) { per_dev[dev] = new tf_detnet_ctx; per_dev[dev]->device = dev; }
  return per_dev[dev];
}
}  // namespace
extern "C" int tf_detnet_ctx_create(tf_detnet_ctx** out) {
  if (!out) return TF_ERR_ARG;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return TF_ERR_LAUNCH;
  tf_detnet_ctx* x = new tf_detnet_ctx; x->device = dev;
  *out = x;
  return TF_OK;
}
extern "C" int tf_detnet_ctx_destroy(tf_detnet_ctx* x) {
  if (!x) return TF_OK;
  if (x->side) (void)hipStreamSynchronize(x->side);
  if (x->gside) (void)hipStreamSynchronize(x->gside);
  for (hipEvent_t e : x->events) (void)hipEventDestroy(e);
  if (x->pack_fork) (void)hipEventDestroy(x->pack_fork);
  if (x->pack_join) (void)hipEventDestroy(x->pack_join);
  if (x->pack_join0) (void)hipEventDestroy(x->pack_join0);
  if (x->side) (void)hipStreamDestroy(x->side);
  if (x->gside) (void)hipStreamDestroy(x->gside);
  delete x;
  return TF_OK;
}
// legacy process-wide switches (tf_detnet_set_dual_stream / _grad_events / _grad_callback): hooks of the context-free entry points
static bool g_force_single = false;      // tf_detnet_set_dual_stream(0): everything on the caller's stream

extern "C" int tf_detnet_forward_ctx(tf_detnet_ctx* xctx, int single_stream, int dtype, int training, const float* x, int N, int H, int W, int nout,
                                     void* const* params, float eps, float mom, float* out, void* ws, size_t ws_bytes, int flags, void* stream_) {
  if (!xctx) xctx = default_ctx();
  if (!x || !params || !out || !ws || nout <= 0 || nout > kHeadLd) return TF_ERR_ARG;
  if (dtype != TF_BF16 && dtype != TF_F32 && dtype != TF_F16) return TF_ERR_UNSUPPORTED;
  if (dtype == TF_F16 && training) return TF_ERR_UNSUPPORTED;       // fp16 operands: the inference graph only (BASELINE.json configs[4])
  const Arch& A = arch();
  Plan P; Arena ar(ws, ws_bytes);
  build_plan(P, ar, dtype, N, H, W, nout, training);
  if (!ar.ok) return TF_ERR_WORKSPACE;
  Ctx c{dtype, (hipStream_t)stream_, params, nullptr, TF_OK};
  tf_conv_args a;
  const bool tr = training != 0;
  // eval only: the packed weights, folded BN affines and head vectors at the front of `ws` are already those of `params`
  const bool ready = !tr && (flags & TF_DETNET_WEIGHTS_READY);

  c.skip_fold = ready;
  // statistics folded into <= TF_STAT_ROWS rows: the elementwise consumers finalize them in-kernel (bn_fused.hip);
  // unfolded (tf_set_stat_rows(0), bit-reproducible sums): separate finalize kernels on the shared partial buffer
  const int srows = tf_get_stat_rows();
  const bool g_unfused_env = tf::tuning().unfused_bn;     // A/B knob
  const bool fused = tr && srows <= TF_STAT_ROWS && !g_unfused_env;
  // statistic rows start at zero: the per-BN regions and, right behind them in the arena, the head of the shared partial buffer -- ONE memset
  if (tr && hipMemsetAsync(P.stat_fwd, 0, (size_t)((char*)P.partial - (char*)P.stat_fwd) + (size_t)TF_STAT_ROWS * 3 * 1024 * 4, c.stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
  // ---- stem: conv1 (direct, or im2col + GEMM) (+BN+ReLU) + maxpool
  const int M1 = N * P.H1 * P.W1;
  // r3 experiment, NEGATIVE, kept behind TINYFACES_PACK_SIDE=1: the weight re-packing of a training step (three launches, ~170 us: 111 MB
  // of masters read, 2 x 55 MB written) on a second stream BESIDE the stem's im2col (~130 us) -- both are HBM-bound, side by side they
  // take as long as back to back and the fork / join events cost a little: 1153 / 1153 img/s against 1161 / 1159 inline (A/B on one box).
  bool pack_joined = false;
  hipStream_t g_pack_stream = nullptr;             // = the context's second stream: idle during the forward pass, no further hardware queue
  const bool pack_side_env = tf::tuning().pack_side;
  // r3, second form: only the weights of layer 3 and of the heads (62 % of the bytes) go to the second stream, forked at the top of the step
  // and joined in front of the first layer-3 bottleneck: they are packed beside the stem and layers 1-2, whose launches are latency-bound
  // at bs = 12, instead of in front of them.  TINYFACES_PACK_SPLIT_OFF=1: everything inline.
  const bool pack_split_off = tf::tuning().pack_split_off;
  const bool single_env = tf::tuning().single_stream;
  const bool pack_split = tr && !ready && !pack_side_env && !pack_split_off && !single_env && !single_stream && xctx;
  bool pack_side = tr && !ready && (pack_side_env || pack_split);
  if (pack_side) {
    g_pack_stream = ctx_side(xctx);
    if (g_pack_stream && !xctx->pack_fork && (hipEventCreateWithFlags(&xctx->pack_fork, hipEventDisableTiming) != hipSuccess ||
                                              hipEventCreateWithFlags(&xctx->pack_join, hipEventDisableTiming) != hipSuccess ||
                                              hipEventCreateWithFlags(&xctx->pack_join0, hipEventDisableTiming) != hipSuccess)) g_pack_stream = nullptr;
  }
  if (!g_pack_stream) pack_side = false;
  const bool stem_direct = stem_direct_mode(dtype, tr, fused);
  // r4 experiment, NEGATIVE, opt-in (TINYFACES_PACK_FORK_LATE=1): the layer-3 packing forked BEHIND the stem conv instead of at the top of the
  // step (beside it the conv takes 99 instead of 56 us): 1273.0 / 1274.2 against 1276.1 / 1276.1 img/s -- the packing then overlaps layer 1
  const bool fork_late_env = tf::tuning().pack_fork_late;
  const bool pack_late = pack_side && pack_split && stem_direct && fork_late_env;
  if (pack_side && !pack_late) {           // everything enqueued so far (the previous step's SGD: the masters) precedes the packing
    if (hipEventRecord(xctx->pack_fork, c.stream) != hipSuccess || hipStreamWaitEvent(g_pack_stream, xctx->pack_fork, 0) != hipSuccess) c.chk(TF_ERR_LAUNCH);
  }
  if (!stem_direct) c.chk(tf_stem_im2col(x, N, H, W, dtype, P.col, kStemK, c.stream));
  // ---- every weight of the pass re-packed from the fp32 master copy in two launches
  if (!ready) {
  // every operand of the pass (and, in training, of the backward pass) from ONE read of the fp32 masters
  pack2(c, A.stem, 64, P.wstem, nullptr, 147, kStemK, 1);     // conv1.weight flattened OIHW == im2col k order
  for (size_t i = 0; i < A.blocks.size(); ++i) {
    const Block& B = A.blocks[i];
    Plan::Blk& b = P.blk[i];
    // stem + layers 1-2 are needed first: inline, behind the im2col.  r4 experiment, NEUTRAL (1277 / 1281 against 1279 / 1275 img/s), opt-in
    // (TINYFACES_PACK_FIRST_SIDE=1): packed on the second stream as well, first in its order, beside the stem's im2col, the caller's
    // stream waiting for them in front of the stem conv
    if (pack_side && pack_split && (int)i == A.layer_end[1] + 1) {
      const bool first_inline = !tf::tuning().pack_first_side;
      if (first_inline) c.flush_packs();
      else {
        c.flush_packs(g_pack_stream);
        if (hipEventRecord(xctx->pack_join0, g_pack_stream) != hipSuccess || hipStreamWaitEvent(c.stream, xctx->pack_join0, 0) != hipSuccess) c.chk(TF_ERR_LAUNCH);
      }
    }
    pack2(c, B.c1, B.planes, b.w1, b.w1t); pack2(c, B.c2, B.planes, b.w2, b.w2t); pack2(c, B.c3, B.planes * 4, b.w3, b.w3t);
    if (B.has_ds) pack2(c, B.ds, B.planes * 4, b.wd, b.wdt);
  }
  {
    tf_pack2_job j;
    j.src = c.P(A.head3.w); j.dst = P.w_h3; j.dst_t = P.w_h3t; j.cout = nout; j.cin = 512; j.taps = 1;
    j.rows_pad = kHeadLd; j.cols_pad = 512; j.rows_pad_t = 512; j.cols_pad_t = kHeadLd;
    c.jobs2.push_back(j);
    j.src = c.P(A.head4.w); j.dst = P.w_h4; j.dst_t = P.w_h4t; j.cin = 1024; j.cols_pad = 1024; j.rows_pad_t = 1024;
    c.jobs2.push_back(j);
  }
  if (!pack_late) {
    c.flush_packs(pack_side ? g_pack_stream : nullptr);
    if (pack_side) {
      if (hipEventRecord(xctx->pack_join, g_pack_stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
      if (!pack_split && hipStreamWaitEvent(c.stream, xctx->pack_join, 0) != hipSuccess) c.chk(TF_ERR_LAUNCH);
    }
  }
  }
  if (stem_direct) {
    if (tr) {
      int rows = 0;
      c.chk(tf_stem_conv(dtype, x, N, H, W, P.wstem, kStemK, P.cstem, TF_EPI_STATS, nullptr, nullptr, P.partial, &rows, c.stream));
      c.chk(tf_bn_finalize(P.partial, rows, 64, 64, (float)M1, c.P(A.stem.gamma), c.P(A.stem.beta), eps, mom, P.bn_stem.scale, P.bn_stem.shift,
                           P.bn_stem.mean, P.bn_stem.invstd, (float*)c.params[A.stem.rmean], (float*)c.params[A.stem.rvar], 1, c.stream));
    } else {
      bn_forward(c, A.stem, 64, P.bn_stem, false, nullptr, nullptr, 0, eps, mom);
      c.chk(tf_stem_conv(dtype, x, N, H, W, P.wstem, kStemK, P.cstem, TF_EPI_AFFINE | TF_EPI_RELU, P.bn_stem.scale, P.bn_stem.shift, nullptr, nullptr,
                         c.stream));
    }
  } else {
    conv_fill(a, dtype, 0, 1, 1, M1, kStemK, 1, M1, 64, 1, 1, 0, 64, P.col, P.wstem, P.cstem);
    a.alg_k = 147;                                  // 7 x 7 x 3 taps*channels, zero-padded to kStemK for the 64-deep K stages
    if (tr) { a.epi = TF_EPI_STATS; a.stat_out = P.partial; }
    else {
      bn_forward(c, A.stem, 64, P.bn_stem, false, nullptr, nullptr, 0, eps, mom);
      a.epi = TF_EPI_AFFINE | TF_EPI_RELU; a.epi_scale = P.bn_stem.scale; a.epi_shift = P.bn_stem.shift;
    }
    c.chk(tf_conv2d(&a, c.stream));
    if (tr) bn_forward(c, A.stem, 64, P.bn_stem, true, &a, P.partial, (float)M1, eps, mom);
  }
  if (pack_late && !ready) {               // the layer-3 + head packing: forked behind the stem conv, joined in front of layer 3
    if (hipEventRecord(xctx->pack_fork, c.stream) != hipSuccess || hipStreamWaitEvent(g_pack_stream, xctx->pack_fork, 0) != hipSuccess) c.chk(TF_ERR_LAUNCH);
    c.flush_packs(g_pack_stream);
    if (hipEventRecord(xctx->pack_join, g_pack_stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
  }
  c.chk(tf_maxpool_fwd(dtype, P.cstem, N, P.H1, P.W1, 64, tr ? P.bn_stem.scale : nullptr, tr ? P.bn_stem.shift : nullptr, P.pool,
                       tr ? P.pool_idx : nullptr, c.stream));

  // ---- bottlenecks
  const void* yin = P.pool;
  bool tail_deferred = false;             // the previous bottleneck left c3 raw: this one's conv1 produces y on its way in
  for (size_t i = 0; i < A.blocks.size(); ++i) {
    const Block& B = A.blocks[i];
    Plan::Blk& b = P.blk[i];
    const int pl = B.planes, c4 = pl * 4;
    const int Min = N * b.Hin * b.Win, Mout = N * b.Hout * b.Wout;
    if (pack_side && pack_split && (int)i == A.layer_end[1] + 1) { pack_joined = true; if (hipStreamWaitEvent(c.stream, xctx->pack_join, 0) != hipSuccess) c.chk(TF_ERR_LAUNCH); }
    // conv1 1x1
    conv_fill(a, dtype, 0, N, b.Hin, b.Win, B.cin, b.Hin, b.Win, pl, 1, 1, 0, pl, yin, b.w1, b.c1);
    if (tr) { a.epi = TF_EPI_STATS; a.stat_out = fused ? b.b1.fst : P.partial; stat_shift(a, c, B.c1, b.b1, fused); }
    else { bn_forward(c, B.c1, pl, b.b1, false, nullptr, nullptr, 0, eps, mom); a.epi = TF_EPI_AFFINE | TF_EPI_RELU; a.epi_scale = b.b1.scale; a.epi_shift = b.b1.shift; }
    if (tail_deferred) {
      // r5: the PREVIOUS bottleneck's  y = relu(bn3(c3) + residual)  rides on this conv's operand path (tf_conv2d_bnfwd, conv_pwx.hip): the
      // pixel stages are transformed once in LDS on their way to the MFMAs and y (= yin: three more readers) is the launch's side output
      const Block& Bp = A.blocks[i - 1];
      Plan::Blk& bp = P.blk[i - 1];
      const tf_bn_fwd_desc d3 = fwd_desc(c, Bp.c3, bp.b3);
      tf_bn_fwd_desc dd; if (Bp.has_ds) dd = fwd_desc(c, Bp.ds, bp.bd);
      const void* resid = Bp.has_ds ? bp.d : (i >= 2 ? P.blk[i - 2].y : P.pool);
      a.x = bp.c3;
      const int rc = tf_conv_pwx_launch_fwd(&a, &d3, resid, Bp.has_ds ? &dd : nullptr, bp.y, srows, (float)Min, eps, mom, c.stream);
      if (rc == TF_ERR_UNSUPPORTED) {              // a shape the fused kernel does not take after all: the two launches it stands for
        c.chk(tf_bn_add_relu_fused(dtype, bp.c3, &d3, resid, Bp.has_ds ? &dd : nullptr, srows, Min, B.cin, (float)Min, eps, mom, bp.y, c.stream));
        a.x = yin;
        c.chk(tf_conv2d(&a, c.stream));
      } else c.chk(rc);
      tail_deferred = false;
    } else c.chk(tf_conv2d(&a, c.stream));
    // a1 = relu(bn1(c1)), materialised on purpose: every consumer (conv2, its weight gradient) uses the LDS-DMA pipeline
    if (fused) {
      const tf_bn_fwd_desc d = fwd_desc(c, B.c1, b.b1);
      c.chk(tf_bn_relu_fused(dtype, b.c1, &d, srows, Min, pl, (float)Min, eps, mom, b.a1, c.stream));
    } else if (tr) {
      bn_forward(c, B.c1, pl, b.b1, true, &a, P.partial, (float)Min, eps, mom);
      c.chk(tf_bn_relu(dtype, b.c1, b.b1.scale, b.b1.shift, Min, pl, b.a1, c.stream));
    }
    // conv2 3x3 (stride here)
    conv_fill(a, dtype, 0, N, b.Hin, b.Win, pl, b.Hout, b.Wout, pl, 3, B.stride, 1, pl, tr ? b.a1 : b.c1, b.w2, b.c2);
    if (tr) { a.epi = TF_EPI_STATS; a.stat_out = fused ? b.b2.fst : P.partial; stat_shift(a, c, B.c2, b.b2, fused); }
    else { bn_forward(c, B.c2, pl, b.b2, false, nullptr, nullptr, 0, eps, mom); a.epi = TF_EPI_AFFINE | TF_EPI_RELU; a.epi_scale = b.b2.scale; a.epi_shift = b.b2.shift; }
    c.chk(tf_conv2d(&a, c.stream));
    if (tr && !fused) bn_forward(c, B.c2, pl, b.b2, true, &a, P.partial, (float)Mout, eps, mom);
    // downsample 1x1 (stride)
    if (B.has_ds) {
      conv_fill(a, dtype, 0, N, b.Hin, b.Win, B.cin, b.Hout, b.Wout, c4, 1, B.stride, 0, c4, yin, b.wd, b.d);
      if (tr) { a.epi = TF_EPI_STATS; a.stat_out = fused ? b.bd.fst : P.partial; stat_shift(a, c, B.ds, b.bd, fused); }
      else { bn_forward(c, B.ds, c4, b.bd, false, nullptr, nullptr, 0, eps, mom); a.epi = TF_EPI_AFFINE; a.epi_scale = b.bd.scale; a.epi_shift = b.bd.shift; }
      c.chk(tf_conv2d(&a, c.stream));
      if (tr && !fused) bn_forward(c, B.ds, c4, b.bd, true, &a, P.partial, (float)Mout, eps, mom);
    }
    // a2 = relu(bn2(c2)): a launch of its own.  Folding it into conv3 (r3, tf_conv_args.bnf: the ring-less pointwise kernel activates its
    // pixel tile in LDS and writes a2 for the weight gradient; bit-identical, tests/test_gpu_conv.py) removes 33 launches from the
    // forward chain but LOSES 1.3 % on the step (A/B 1101 vs 1116 img/s): each of the 4-16 channel tiles of a pixel tile repeats the
    // activation and the table derivation, and the extra barrier per stage sits in a launch that is latency-bound already.
    // TINYFACES_BNF=1 turns it on (kept for the eval-sized shapes where pixel tiles >> channel tiles).
    const bool bnf_on = TF_EXP && tf::tuning().bnf;
    const bool bnf = fused && bnf_on && dtype != TF_F32 && pl <= 256;
    tf_bn_fwd_desc d2;
    if (fused) d2 = fwd_desc(c, B.c2, b.b2);
    if (fused && !bnf) {
      c.chk(tf_bn_relu_fused(dtype, b.c2, &d2, srows, Mout, pl, (float)Mout, eps, mom, b.a2, c.stream));
    } else if (tr && !fused) {
      c.chk(tf_bn_relu(dtype, b.c2, b.b2.scale, b.b2.shift, Mout, pl, b.a2, c.stream));
    }
    // conv3 1x1 (+ BN + residual + ReLU)
    conv_fill(a, dtype, 0, N, b.Hout, b.Wout, pl, b.Hout, b.Wout, c4, 1, 1, 0, c4, bnf ? b.c2 : (tr ? b.a2 : b.c2), b.w3, tr ? b.c3 : b.y);
    if (bnf) { a.bnf = &d2; a.bnf_out = b.a2; a.bnf_rows = srows; a.bnf_count = (float)Mout; a.bnf_eps = eps; a.bnf_momentum = mom; }
    if (tr) { a.epi = TF_EPI_STATS; a.stat_out = fused ? b.b3.fst : P.partial; stat_shift(a, c, B.c3, b.b3, fused); }
    else {
      bn_forward(c, B.c3, c4, b.b3, false, nullptr, nullptr, 0, eps, mom);
      a.epi = TF_EPI_AFFINE | TF_EPI_RES | TF_EPI_RELU; a.epi_scale = b.b3.scale; a.epi_shift = b.b3.shift; a.aux = B.has_ds ? b.d : yin;
    }
    c.chk(tf_conv2d(&a, c.stream));
    // r5: the block's tail (bn3 + residual + ReLU) is NOT launched when the next bottleneck's conv1 can apply it on its operand path
    // (bf16 training, conv1 with 128 / 256 output channels and 128 ... 1024 input channels: layers 2 and 3).  OPT-IN (TINYFACES_PWX_FWD=1): measured
    // r5, it LOSES -- alone 37.4 us against 29.1 for the two launches at layer 3, 45.5 against 40.7 at layer 2 (profiles/r05_conv_pwx.txt); in the
    // step 1238-1240 img/s against 1277-1279 without it (same box).  DESIGN.md 7.
    const bool pwx_fwd_off = !(TF_EXP && tf::tuning().pwx_fwd);
    if (fused && !pwx_fwd_off && dtype == TF_BF16 && i + 1 < A.blocks.size()) {
      const Block& Bn = A.blocks[i + 1];
      tail_deferred = Bn.cin == c4 && Bn.planes % 128 == 0 && Bn.cin % 64 == 0 && Bn.cin >= 128 && Bn.cin <= 1024;
    }
    if (tail_deferred) {
      // nothing here
    } else if (fused) {
      const tf_bn_fwd_desc d3 = fwd_desc(c, B.c3, b.b3);
      tf_bn_fwd_desc dd; if (B.has_ds) dd = fwd_desc(c, B.ds, b.bd);
      c.chk(tf_bn_add_relu_fused(dtype, b.c3, &d3, B.has_ds ? b.d : yin, B.has_ds ? &dd : nullptr, srows, Mout, c4, (float)Mout, eps, mom, b.y,
                                 c.stream));
    } else if (tr) {
      bn_forward(c, B.c3, c4, b.b3, true, &a, P.partial, (float)Mout, eps, mom);
      c.chk(tf_bn_add_relu(dtype, b.c3, b.b3.scale, b.b3.shift, B.has_ds ? b.d : yin, B.has_ds ? b.bd.scale : nullptr,
                           B.has_ds ? b.bd.shift : nullptr, Mout, c4, b.y, c.stream));
    }
    yin = b.y;
  }

  // ---- heads + bilinear upsample + crop + add
  const void* res3 = P.blk[A.layer_end[1]].y;
  const void* res4 = P.blk[A.layer_end[2]].y;
  if (!ready)
    hipLaunchKernelGGL(head_vectors_kernel, dim3((nout * 16 + 255) / 256), dim3(256), 0, c.stream, c.P(A.head3.bias), c.P(A.head4.bias),
                       c.P(A.upsample_w), nout, P.hbias3, P.hbias4, P.ones, P.wup_diag);
  conv_fill(a, dtype, 0, N, P.H3, P.W3, 512, P.H3, P.W3, kHeadLd, 1, 1, 0, kHeadLd, res3, P.w_h3, P.s3);
  a.epi = TF_EPI_AFFINE; a.epi_scale = P.ones; a.epi_shift = P.hbias3; a.alg_n = nout;
  c.chk(tf_conv2d(&a, c.stream));
  conv_fill(a, dtype, 0, N, P.H4, P.W4, 1024, P.H4, P.W4, kHeadLd, 1, 1, 0, kHeadLd, res4, P.w_h4, P.s4);
  a.epi = TF_EPI_AFFINE; a.epi_scale = P.ones; a.epi_shift = P.hbias4; a.alg_n = nout;
  c.chk(tf_conv2d(&a, c.stream));
  c.chk(tf_upsample_add_crop(dtype, P.s3, P.s4, P.wup_diag, N, nout, kHeadLd, P.H3, P.W3, P.H4, P.W4, out, c.stream));
  if (pack_side && pack_split && !pack_joined) (void)hipStreamWaitEvent(c.stream, xctx->pack_join, 0);      // (never leave the caller's stream un-joined)
  if (hipGetLastError() != hipSuccess && c.rc == TF_OK) c.rc = TF_ERR_LAUNCH;
  return c.rc;
}
extern "C" int tf_detnet_forward(int dtype, int training, const float* x, int N, int H, int W, int nout, void* const* params, float eps,
                                 float mom, float* out, void* ws, size_t ws_bytes, int flags, void* stream_) {
  return tf_detnet_forward_ctx(nullptr, g_force_single ? 1 : 0, dtype, training, x, N, H, W, nout, params, eps, mom, out, ws, ws_bytes, flags, stream_);
}

// ------------------------------------------------------------------------------------------------
// backward
namespace {

// g_x for a BN whose output-gradient sums are in `partial`: finalize (dgamma, dbeta, coefficients)
void bn_backward_coefs(Ctx& c, const ConvUnit& u, int C, BnBuf& b, const float* partial, int nblk, int nk, int kidx, int ld, float count,
                       int clear = 1) {
  c.chk(tf_bn_bwd_finalize(partial, nblk, nk, kidx, ld, C, count, c.P(u.gamma), b.mean, b.invstd, c.G(u.gamma), c.G(u.beta), b.cA, b.cB,
                           b.cD, clear, c.stream));
}

tf_wgrad_args wgrad_args(const Ctx& c, const ConvUnit& u, int cout, int N, int H, int W, int OH, int OW, const void* x, int ldx, const void* dy,
                         int lddy, int cin_override = 0, int k_override = 0, int dw_ld = 0) {
  tf_wgrad_args w;
  memset(&w, 0, sizeof(w));
  const int cin = cin_override ? cin_override : u.cin, k = k_override ? k_override : u.k;
  w.dtype = c.dtype; w.N = N; w.H = H; w.W = W; w.Cin = cin; w.OH = OH; w.OW = OW; w.Cout = cout; w.KH = k; w.KW = k;
  w.stride = u.stride; w.pad = u.pad; w.ldx = ldx; w.lddy = lddy; w.x = x; w.dy = dy; w.dw_oihw = c.G(u.w);
  w.dw_ld = dw_ld ? dw_ld : cin * k * k;
  return w;
}

void wgrad(Ctx& c, const ConvUnit& u, int cout, int N, int H, int W, int OH, int OW, const void* x, int ldx, const void* dy, int lddy,
           const BnBuf* pro, int cin_override = 0, int k_override = 0, int dw_ld = 0, float* packed_scratch = nullptr, size_t scratch_floats = 0) {
  // timing-ablation knob (RESULTS INVALID): the step without any weight gradient = what the data-gradient chain costs when it owns the GPU
  const bool skip = tf::tuning().dbg_skip_wgrad;
  if (skip) return;
  const int cin = cin_override ? cin_override : u.cin, k = k_override ? k_override : u.k;
  tf_wgrad_args w = wgrad_args(c, u, cout, N, H, W, OH, OW, x, ldx, dy, lddy, cin_override, k_override, dw_ld);
  if (pro) { w.pro_scale = pro->scale; w.pro_shift = pro->shift; w.pro_relu = 1; }
  if (k > 1 && packed_scratch && c.grads_zeroed) {
    // 3x3 / stride 1: the all-taps kernel sums its split-K slices through the scratch straight into the (already zeroed) OIHW
    // gradient: no memset, no atomics, no transposing copy
    w.partial_ws = packed_scratch; w.partial_ws_bytes = scratch_floats * 4;
    if (tf_wgrad_workspace_bytes(&w) != 0 && tf_wgrad_workspace_bytes(&w) <= w.partial_ws_bytes) {
      const bool w3_off = tf::tuning().wgrad3_off;
      if (!w3_off) { c.chk(tf_conv2d_wgrad(&w, c.wstream())); return; }
    }
    w.partial_ws = nullptr; w.partial_ws_bytes = 0;
  }
  if (k > 1 && packed_scratch) {          // 3x3 (stride 2): coalesced atomics into [Cout][tap][Cin], then one transposing copy to OIHW
    float* oihw = w.dw_oihw;
    w.dw_oihw = packed_scratch; w.packed = 1;
    if (hipMemsetAsync(packed_scratch, 0, (size_t)cout * w.dw_ld * 4, c.wstream()) != hipSuccess) c.chk(TF_ERR_LAUNCH);   // scratch: always
    c.chk(tf_conv2d_wgrad(&w, c.wstream()));
    c.chk(tf_unpack_dw(packed_scratch, cout, cin, k * k, oihw, c.wstream()));
    return;
  }
  if (!c.grads_zeroed && hipMemsetAsync(w.dw_oihw, 0, (size_t)cout * w.dw_ld * 4, c.wstream()) != hipSuccess) c.chk(TF_ERR_LAUNCH);
  c.chk(tf_conv2d_wgrad(&w, c.wstream()));
}

}  // namespace

// gradient-ready hooks (data-parallel overlap): after the weight gradients of bottleneck `block` (backward order: the LAST block of a
// bucket) are enqueued, the caller's event is recorded on the stream that carries them and / or the caller's function is called with that
// stream, so that the bucket's all-reduce starts while the rest of the backward pass runs.  block -1 = the very end (stem done).
// r4: the hooks are an ARGUMENT of the backward call (tf_detnet_hooks), not a process-wide registration.
static std::vector<std::pair<int, hipEvent_t>> g_grad_events;      // legacy registration (tf_detnet_set_grad_events), used by tf_detnet_backward only
extern "C" int tf_detnet_set_grad_events(const int* blocks, void* const* events, int n) {
  g_grad_events.clear();
  if (n < 0 || (n > 0 && (!blocks || !events))) return TF_ERR_ARG;       // (events[k] itself may be NULL)
  for (int k = 0; k < n; ++k) g_grad_events.emplace_back(blocks[k], (hipEvent_t)events[k]);      // a NULL event: the callback only
  return TF_OK;
}
static tf_grad_ready_fn g_grad_cb = nullptr;
static void* g_grad_cb_user = nullptr;
extern "C" int tf_detnet_set_grad_callback(tf_grad_ready_fn fn, void* user) { g_grad_cb = fn; g_grad_cb_user = user; return TF_OK; }
static void record_grad_events(const tf_detnet_hooks* h, int block, hipStream_t s, int& rc) {
  if (!h) return;
  bool registered = false;
  for (int k = 0; k < h->n; ++k)
    if (h->blocks[k] == block) {
      registered = true;
      hipEvent_t e = h->events ? (hipEvent_t)h->events[k] : nullptr;
      if (e && hipEventRecord(e, s) != hipSuccess && rc == TF_OK) rc = TF_ERR_LAUNCH;
    }
  if (registered && h->fn) h->fn(block, (void*)s, h->user);
}
static bool grad_event_registered(const tf_detnet_hooks* h, int block) {
  if (!h) return false;
  for (int k = 0; k < h->n; ++k) if (h->blocks[k] == block) return true;
  return false;
}
// 1 = weight gradients on a second stream (default), 0 = everything on the caller's stream (A/B + race tests)
extern "C" int tf_detnet_set_dual_stream(int on) { g_force_single = !on; return TF_OK; }

extern "C" int tf_detnet_backward_ctx(tf_detnet_ctx* xctx, const tf_detnet_hooks* hooks, int dtype, const float* x, int N, int H, int W, int nout,
                                      void* const* params, void* const* grads, const float* gout, void* grad_flat, size_t grad_flat_bytes,
                                      void* ws, size_t ws_bytes, void* stream_) {
  if (!x || !params || !grads || !gout || !ws) return TF_ERR_ARG;
  if (hooks && (hooks->n < 0 || (hooks->n > 0 && !hooks->blocks))) return TF_ERR_ARG;
  if (!xctx) xctx = default_ctx();
  if (dtype != TF_BF16 && dtype != TF_F32) return TF_ERR_UNSUPPORTED;
  const Arch& A = arch();
  Plan P; Arena ar(ws, ws_bytes);
  build_plan(P, ar, dtype, N, H, W, nout, 1);
  if (!ar.ok) return TF_ERR_WORKSPACE;
  Ctx c{dtype, (hipStream_t)stream_, params, grads, TF_OK};   // (grads_zeroed / jobs default-initialised)
  const bool g_single_env = tf::tuning().single_stream;
  if (!g_single_env && !(hooks && hooks->single_stream) && xctx) {
    hipStream_t g_side = ctx_side(xctx);
    c.side = g_side; c.events = &xctx->events;
    // r4, measured NEGATIVE, opt-in (TINYFACES_GROUP_STREAM=1): the grouped launches on a third queue, so that the per-block gradients of
    // layer 3.0 / layers 1-2 do not queue up behind a group: 1170 / 1164 img/s against 1177 / 1181 on the second stream (A/B on one box,
    // main-queue idle time of the backward pass 508 instead of 379 us): a third busy queue costs more than the queueing it removes.
    const bool gstream_on = tf::tuning().group_stream;
    if (g_side && gstream_on) c.gside = ctx_gside(xctx);
  }
  tf_conv_args a;
  const int M3 = N * P.H3 * P.W3, M4 = N * P.H4 * P.W4;
  const void* res3 = P.blk[A.layer_end[1]].y;
  const void* res4 = P.blk[A.layer_end[2]].y;

  const int srows = tf_get_stat_rows();
  const bool g_unfused_env = tf::tuning().unfused_bn;
  const bool fused = srows <= TF_STAT_ROWS && !g_unfused_env;   // see tf_detnet_forward
  // statistic rows start at zero: the per-BN backward regions + the head of this pass's partial buffer right behind them, ONE memset
  if (hipMemsetAsync(P.stat_bwd, 0, (size_t)((char*)P.partial_b - (char*)P.stat_bwd) + (size_t)TF_STAT_ROWS * 3 * 1024 * 4, c.stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
  // the forward pass took conv1 straight from the image.  Its weight gradient does too (tf_stem_wgrad); with TINYFACES_STEM_WGRAD_IM2COL=1 it
  // reduces over the im2col matrix as in rounds 1-3, built here on the second stream (beside the head's backward)
  const bool stem_wgrad_im2col = tf::tuning().stem_wgrad_im2col;
  const bool stem_direct = stem_direct_mode(dtype, true, fused);
  if (stem_direct && stem_wgrad_im2col) {
    if (c.side) { c.fork(); c.chk(tf_stem_im2col(x, N, H, W, dtype, P.col, kStemK, c.side)); }
    else c.chk(tf_stem_im2col(x, N, H, W, dtype, P.col, kStemK, c.stream));
  }
  const bool group_on = wgrad_group_mode(dtype, 1) && fused;
  const int first_id = A.layer_end[1] + 2, last_id = A.layer_end[2];           // the identity bottlenecks of layer 3: blocks 8 .. 29
  if (grad_flat && grad_flat_bytes) {                     // one memset for every weight gradient (atomics accumulate into them) ...
    // r4: ... except where nothing accumulates: the grouped weight gradients OVERWRITE theirs, and the BatchNorm gradients in between are
    // published with plain stores (bn_fused.hip) -- the 22 identity bottlenecks of layer 3 are 98 of the 111 MB of the flat gradient.
    // With the table in executor order (DetectionModel.flatten_parameters) they are ONE range: [layer3.1.conv1.weight, score_res3.weight).
    char* lo = group_on ? (char*)c.G(A.blocks[first_id].c1.w) : nullptr;
    char* hi = group_on ? (char*)c.G(A.head3.w) : nullptr;
    char* g0 = (char*)grad_flat; char* g1 = g0 + grad_flat_bytes;
    bool split = lo && hi && lo >= g0 && hi <= g1 && lo < hi;
    for (int i = first_id; split && i <= last_id; ++i) {          // every trained tensor of these blocks must lie inside the range
      const Block& B = A.blocks[i];
      const int idx[9] = {B.c1.w, B.c1.gamma, B.c1.beta, B.c2.w, B.c2.gamma, B.c2.beta, B.c3.w, B.c3.gamma, B.c3.beta};
      for (int q : idx) { char* t = (char*)c.G(q); split = split && t >= lo && t < hi; }
    }
    for (int q = 0; split && q < (int)A.names.size(); ++q) {       // ... and nothing else may
      char* t = (char*)c.G(q);
      if (!t || t < lo || t >= hi) continue;
      bool mine = false;
      for (int i = first_id; i <= last_id && !mine; ++i) {
        const Block& B = A.blocks[i];
        mine = q == B.c1.w || q == B.c1.gamma || q == B.c1.beta || q == B.c2.w || q == B.c2.gamma || q == B.c2.beta || q == B.c3.w || q == B.c3.gamma || q == B.c3.beta;
      }
      split = split && mine;
    }
    const bool split_off = tf::tuning().grad_memset_full;     // A/B + safety knob
    if (split && !split_off) {
      if (lo > g0 && hipMemsetAsync(g0, 0, (size_t)(lo - g0), c.stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
      if (g1 > hi && hipMemsetAsync(hi, 0, (size_t)(g1 - hi), c.stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
    } else if (hipMemsetAsync(grad_flat, 0, grad_flat_bytes, c.stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
    c.grads_zeroed = true;
  }
  // (the transposed weight operands w*t were packed by the training forward, together with the forward operands)

  // ---- heads
  c.chk(tf_upsample_add_crop_bwd(dtype, gout, P.wup_diag, N, nout, kHeadLd, P.H3, P.W3, P.H4, P.W4, P.g3, P.g4, c.stream));
  {
    const int nb3 = tf_colstats_blocks(M3, kHeadLd, dtype), nb4 = tf_colstats_blocks(M4, kHeadLd, dtype);
    c.chk(tf_colstats(dtype, P.g3, nullptr, nullptr, nullptr, M3, kHeadLd, kHeadLd, P.partial_b, c.stream));
    c.chk(tf_reduce_partials(P.partial_b, nb3, 1, 0, kHeadLd, nout, c.G(A.head3.bias), 1, c.stream));
    c.chk(tf_colstats(dtype, P.g4, nullptr, nullptr, nullptr, M4, kHeadLd, kHeadLd, P.partial_b, c.stream));
    c.chk(tf_reduce_partials(P.partial_b, nb4, 1, 0, kHeadLd, nout, c.G(A.head4.bias), 1, c.stream));
  }
  {
    ConvUnit h3 = A.head3, h4 = A.head4;
    c.fork();
    wgrad(c, h3, nout, N, P.H3, P.W3, P.H3, P.W3, res3, 512, P.g3, kHeadLd, nullptr);
    wgrad(c, h4, nout, N, P.H4, P.W4, P.H4, P.W4, res4, 1024, P.g4, kHeadLd, nullptr);
  }
  // score4_upsample.weight has lr 0 (model.py:84): its gradient is defined as zero here
  {
    char* u = (char*)c.G(A.upsample_w);
    const bool covered = c.grads_zeroed && u >= (char*)grad_flat && u + (size_t)nout * nout * 16 * 4 <= (char*)grad_flat + grad_flat_bytes;     // (the flat buffer's memset)
    if (u && !covered && hipMemsetAsync(u, 0, (size_t)nout * nout * 16 * 4, c.stream) != hipSuccess) c.chk(TF_ERR_LAUNCH);
  }

  // Buffer roles: Gcur/Gnext ping-pong the gradient w.r.t. a block output / input; T1 = g_c3 then g_c1;
  // T2 = gz2 -> g_c2; T3 = g_d; T4 = downsample-branch input gradient; R3 = gradient w.r.t. res3 from the head.
  void *Gcur = P.G0, *Gnext = P.G1;
  conv_fill(a, dtype, 1, N, P.H4, P.W4, kHeadLd, P.H4, P.W4, 1024, 1, 1, 0, 1024, P.g4, P.w_h4t, Gcur);
  a.alg_k = nout;
  if (fused) {
    // fused flow: Gcur always carries gz = g_y * (y > 0) of the block about to be processed, and (unless that block has a
    // downsample branch) its BN3-backward sums are accumulated by the conv that produces it (MASK2 | STATS3)
    const size_t last = A.blocks.size() - 1;
    a.epi = TF_EPI_MASK2; a.aux2 = P.blk[last].y;
    if (!A.blocks[last].has_ds) { a.epi |= TF_EPI_STATS3; a.aux3 = P.blk[last].c3; a.stat_out = P.blk[last].b3.bst; }
  }
  c.chk(tf_conv2d(&a, c.stream));
  conv_fill(a, dtype, 1, N, P.H3, P.W3, kHeadLd, P.H3, P.W3, 512, 1, 1, 0, 512, P.g3, P.w_h3t, P.R3);
  a.alg_k = nout;
  c.chk(tf_conv2d(&a, c.stream));

  // one fork per weight gradient (default) or, TINYFACES_FORK_PER_BLOCK=1, one per block: measured 948 vs 943 img/s
  const bool fork_each = !tf::tuning().fork_per_block;
  // Every fork makes its producer kernel carry a completion signal, which costs ~6 us of main-queue bubble behind that kernel
  // (profiles/r02b_step_timeline.txt: 90 x 5.9 us).  The 22 identity bottlenecks of layer 3 therefore fork ONCE, behind the
  // BN1-backward apply, when the dY operands of all three weight gradients exist: conv3's and conv2's gradients start ~100 us
  // later and overlap the next bottleneck instead, far from the end of the pass (layer 1 / 2 and the stem keep one fork per
  // gradient so that the tail of the weight-gradient stream stays short).  TINYFACES_L3_FORK_PER_WGRAD=1: the old schedule.
  const bool l3_single_fork = !tf::tuning().l3_fork_per_wgrad;
  // r4: GROUPED weight gradients of the identity bottlenecks of layer 3.  Their three gradients are not launched
  // per block any more: the block's dY operands go to buffers of its own (Plan::Blk::gT1 / gT2 / gU1), the problems are queued, and when
  // a group is complete -- behind the BN1-backward apply of its LAST (lowest) block, the kernel that produces the group's last operand --
  // ONE fork hands the second stream two launches (tf_conv2d_wgrad_group): the 2 x n pointwise problems (128 x 128 tiles, each reduced
  // over all 12 288 pixels in-block) and the n 3x3 problems (all-taps kernel, splitk = 1).  No split-K, no atomics, no partial tiles, no
  // reduce kernel, 3 forks instead of 22 for these blocks; the gradient-ready events of the group's blocks fire behind the group.
  std::vector<int> group_close;                                                 // block index that closes each group (descending)
  if (group_on) {
    const int nid = last_id - first_id + 1, gs = wgrad_group_size(), ng = (nid + gs - 1) / gs;
    for (int g = 1; g <= ng; ++g) group_close.push_back(last_id + 1 - (int)(((long long)nid * g + ng - 1) / ng));      // balanced: 8 + 7 + 7
  }
  std::vector<tf_wgrad_args> pend_pw, pend_c3;
  std::vector<int> pend_blocks;
  auto flush_group = [&]() {
    if (pend_blocks.empty()) return;
    const bool skip = tf::tuning().dbg_skip_wgrad;
    if (!skip) {
      // the 3x3 group first: 16 tiles per problem = half a machine for a group of eight; the pointwise launch behind it fills the CUs its
      // tail leaves (both are enqueued behind the same fork)
      // A refused group (e.g. a layer-3 width the all-taps plan does not take: inputs wider than ~2000 px) falls back to the per-problem
      // split-K kernels, which ACCUMULATE with fp32 atomics -- and the split memset above skipped exactly these tensors, so on a
      // persistent flat gradient they would be added onto the previous step's (all-reduced) values: zero each one here, on the stream
      // the fallback launches run on (tests/test_gpu_model.py::test_refused_wgrad_group_falls_back_into_zeroed_gradients)
      auto fallback = [&](const std::vector<tf_wgrad_args>& v) {
        for (const tf_wgrad_args& w : v) {
          if (hipMemsetAsync(w.dw_oihw, 0, (size_t)w.Cout * w.dw_ld * 4, c.gstream()) != hipSuccess) c.chk(TF_ERR_LAUNCH);
          c.chk(tf_conv2d_wgrad(&w, c.gstream()));
        }
      };
      const bool force_refuse = tf::tuning().dbg_group_refuse;      // test knob: exercise the fallback at any size
      int rc = force_refuse ? TF_ERR_UNSUPPORTED : tf_conv2d_wgrad_group(pend_c3.data(), (int)pend_c3.size(), c.gstream());
      if (rc == TF_ERR_UNSUPPORTED) fallback(pend_c3);
      else c.chk(rc);
      // r5: the pointwise group in `pw_split` launches (default 1).  A group of eight bottlenecks is 256 tiles = one block on EVERY CU for
      // ~150 us, and beside it the chain's kernels crawl (profiles/r05_step_timeline.txt: the data gradient that normally takes 22 us
      // takes 118-130 us while the group runs); split, each launch leaves CUs to the chain.  TINYFACES_WGRADG_SPLIT=n.
      const int pw_split = tf::tuning().wgradg_split;
      const int npw = (int)pend_pw.size(), per = (npw + pw_split - 1) / pw_split;
      rc = TF_OK;
      for (int at = 0; at < npw && rc == TF_OK; at += per)
        rc = force_refuse ? TF_ERR_UNSUPPORTED : tf_conv2d_wgrad_group(pend_pw.data() + at, npw - at < per ? npw - at : per, c.gstream());
      if (rc == TF_ERR_UNSUPPORTED) fallback(pend_pw);        // (the first launch already refuses: all problems of a group have the same kind of shape)
      else c.chk(rc);
    }
    // a gradient-ready event of a grouped block promises "every gradient of the blocks >= it, and of the heads": the heads and layer3.x
    // per-block launches live on the second stream, so the group's stream first orders itself behind that stream's position
    if (c.gside && !pend_blocks.empty()) { hipEvent_t e = c.mark(c.side); if (e) (void)hipStreamWaitEvent(c.gside, e, 0); }
    for (int blk : pend_blocks) record_grad_events(hooks, blk, c.gstream(), c.rc);
    pend_pw.clear(); pend_c3.clear(); pend_blocks.clear();
  };
  // ---- bottlenecks in reverse
  for (int i = (int)A.blocks.size() - 1; i >= 0; --i) {
    const Block& B = A.blocks[i];
    Plan::Blk& b = P.blk[i];
    const int pl = B.planes, c4 = pl * 4;
    const int Min = N * b.Hin * b.Win, Mout = N * b.Hout * b.Wout;
    const void* yin = i == 0 ? P.pool : P.blk[i - 1].y;
    const void* extra = (i == A.layer_end[1] + 1) ? P.R3 : nullptr;     // the block whose INPUT is res3
    const bool grouped = group_on && i >= first_id && i <= last_id;
    bool closes_group = false;
    if (grouped) for (int gc : group_close) closes_group |= gc == i;
    const bool late = !grouped && fused && fork_each && l3_single_fork && i > A.layer_end[1] && !B.has_ds;    // the three gradients behind ONE fork
    void *T1 = b.gT1, *T2 = b.gT2, *U1 = b.gU1, *T3 = b.gT3;      // the block's own buffers: nothing on this stream ever waits for a weight gradient
    // (1) per-channel sums for bn3 (and the downsample BN) with gz = g_y * (y > 0)
    const int nb = tf_colstats_blocks(Mout, c4, dtype);
    const int nk = B.has_ds ? 3 : 2;
    if (!fused) c.chk(tf_colstats(dtype, Gcur, b.y, b.c3, b.d, Mout, c4, c4, P.partial_b, c.stream));
    else if (B.has_ds) c.chk(tf_colstats(dtype, Gcur, nullptr, b.c3, b.d, Mout, c4, c4, b.b3.bst, c.stream));   // Gcur is already masked
    // (2) g_c3 -> T1.  r3: in bf16 the apply can ride on the operand path of the data gradient that consumes it (tf_conv2d_bnbwd,
    //     conv_pwx.hip): steps (2) and (4) in ONE launch, T1 its side output for the weight gradient.  Measured (scripts/microbench_pwx.py,
    //     profiles/r03_conv_pwx.txt): layer 2 (M = 47 628, K = 512 -> 128) 37.2 us against 43.1 for the two launches; layer 3 (M = 12 288,
    //     K = 1024 -> 256) 31.6 against 29.8 -- 192 one-per-CU blocks pay the coefficient table and the register-staged operand where the
    //     elementwise kernel has thousands of threads in flight.  So: the identity bottlenecks of layer 2 only (TINYFACES_PWX_ALL=1: every
    //     eligible one, TINYFACES_PWX_OFF=1: none; the step is the same within noise either way, 1160 img/s).
    // Measured r5 (profiles/r05_conv_pwx.txt): alone 27.8 us against 29.9 for the two launches at layer 3 but 46.8 against 40.9 at layer 2 (one block
    // per CU with the deep rings); in the step: off 1287-1294, layer 2 only 1280-1287, layers 2 + 3 1277-1279 img/s.  So the fused form is
    // OPT-IN now: TINYFACES_PWX_BWD=1 (layer 2) / TINYFACES_PWX_ALL=1 (layers 2 and 3).
    const bool pwx_all = TF_EXP && tf::tuning().pwx_all;
    const bool pwx_off = !TF_EXP || tf::tuning().pwx_off || (!tf::tuning().pwx_bwd && !pwx_all);
    bool fused24 = false;
    if (fused && !pwx_off && dtype == TF_BF16 && !B.has_ds && pl % 128 == 0 && (pl == 128 || pwx_all)) {
      const tf_bn_bwd_desc d = bwd_desc(c, B.c3, b.b3, b.b3.bst, nk, 1);
      conv_fill(a, dtype, 1, N, b.Hout, b.Wout, c4, b.Hout, b.Wout, pl, 1, 1, 0, pl, Gcur, b.w3t, T2);
      a.epi = TF_EPI_MASK | TF_EPI_STATS2; a.aux = b.c2; a.mask_scale = b.b2.scale; a.mask_shift = b.b2.shift; a.stat_out = b.b2.bst;
      if (fork_each && !late && !grouped) c.arm_fork();
      const int rc = tf_conv_pwx_launch(&a, &d, b.c3, T1, srows, (float)Mout, c.stream);
      if (rc == TF_OK) fused24 = true;
      else if (rc != TF_ERR_UNSUPPORTED) c.chk(rc);
      else if (tf::take_next_stop_event()) c.pending = nullptr;       // the armed event was not consumed: arm again below
    }
    if (fused24) {
      // nothing: T1 and T2 are on their way
    } else if (fused) {
      const tf_bn_bwd_desc d = bwd_desc(c, B.c3, b.b3, b.b3.bst, nk, 1);
      if (fork_each && !late && !grouped) c.arm_fork();
      c.chk(tf_bn_bwd_apply_fused(dtype, Gcur, nullptr, b.c3, &d, srows, Mout, c4, (float)Mout, T1, c.stream));
    } else {
      bn_backward_coefs(c, B.c3, c4, b.b3, P.partial_b, nb, nk, 1, c4, (float)Mout, B.has_ds ? 0 : 1);   // the last reader clears the partial rows
      if (B.has_ds) bn_backward_coefs(c, B.ds, c4, b.bd, P.partial_b, nb, nk, 2, c4, (float)Mout, 1);
      c.chk(tf_bn_bwd_apply(dtype, Gcur, b.y, b.c3, b.b3.cA, b.b3.cB, b.b3.cD, Mout, c4, T1, c.stream));
    }
    // (3) wgrad conv3 (its input is relu(bn2(c2)), materialised in the forward)
    auto wg3 = [&]() { wgrad(c, B.c3, c4, N, b.Hout, b.Wout, b.Hout, b.Wout, b.a2, pl, T1, c4, nullptr); };
    if (grouped) pend_pw.push_back(wgrad_args(c, B.c3, c4, N, b.Hout, b.Wout, b.Hout, b.Wout, b.a2, pl, T1, c4));
    else if (fork_each && !late) { c.fork_armed(); wg3(); }
    // (4) dgrad conv3 -> gz2 in T2 (masked by relu(bn2(c2))) + BN-backward sums
    if (!fused24) {
      conv_fill(a, dtype, 1, N, b.Hout, b.Wout, c4, b.Hout, b.Wout, pl, 1, 1, 0, pl, T1, b.w3t, T2);
      a.epi = TF_EPI_MASK | TF_EPI_STATS2; a.aux = b.c2; a.mask_scale = b.b2.scale; a.mask_shift = b.b2.shift;
      a.stat_out = fused ? b.b2.bst : P.partial_b;
      c.chk(tf_conv2d(&a, c.stream));
    }
    // (5) g_c2 in place
    if (fused) {
      const tf_bn_bwd_desc d = bwd_desc(c, B.c2, b.b2, b.b2.bst, 2, 1);
      if (fork_each && !late && !grouped) c.arm_fork();
      c.chk(tf_bn_bwd_apply_fused(dtype, T2, nullptr, b.c2, &d, srows, Mout, pl, (float)Mout, T2, c.stream));
    } else {
      bn_backward_coefs(c, B.c2, pl, b.b2, P.partial_b, tf_conv_mtiles(&a), 2, 1, pl, (float)Mout);
      c.chk(tf_bn_bwd_apply(dtype, T2, nullptr, b.c2, b.b2.cA, b.b2.cB, b.b2.cD, Mout, pl, T2, c.stream));
    }
    // (6) wgrad conv2 (input relu(bn1(c1)))
    auto wg2 = [&]() { wgrad(c, B.c2, pl, N, b.Hin, b.Win, b.Hout, b.Wout, b.a1, pl, T2, pl, nullptr, 0, 0, 0, P.dwp, P.dwp_floats); };
    if (grouped) pend_c3.push_back(wgrad_args(c, B.c2, pl, N, b.Hin, b.Win, b.Hout, b.Wout, b.a1, pl, T2, pl));
    else if (fork_each && !late) { c.fork_armed(); wg2(); }
    // (7) dgrad conv2 -> gz1 in U1 (+ sums); output spatial = conv2's input
    conv_fill(a, dtype, 1, N, b.Hout, b.Wout, pl, b.Hin, b.Win, pl, 3, B.stride, 1, pl, T2, b.w2t, U1);
    a.epi = TF_EPI_MASK | TF_EPI_STATS2; a.aux = b.c1; a.mask_scale = b.b1.scale; a.mask_shift = b.b1.shift;
    a.stat_out = fused ? b.b1.bst : P.partial_b;
    c.chk(tf_conv2d(&a, c.stream));
    // (8) g_c1 in place
    if (fused) {
      const tf_bn_bwd_desc d = bwd_desc(c, B.c1, b.b1, b.b1.bst, 2, 1);
      if (grouped ? closes_group : fork_each) c.arm_fork();       // grouped: only the kernel that completes a GROUP's operands carries a fork
      c.chk(tf_bn_bwd_apply_fused(dtype, U1, nullptr, b.c1, &d, srows, Min, pl, (float)Min, U1, c.stream));
    } else {
      bn_backward_coefs(c, B.c1, pl, b.b1, P.partial_b, tf_conv_mtiles(&a), 2, 1, pl, (float)Min);
      c.chk(tf_bn_bwd_apply(dtype, U1, nullptr, b.c1, b.b1.cA, b.b1.cB, b.b1.cD, Min, pl, U1, c.stream));
    }
    // (9) wgrad conv1 (input = block input, already activated)
    auto wg1 = [&]() { wgrad(c, B.c1, pl, N, b.Hin, b.Win, b.Hin, b.Win, yin, B.cin, U1, pl, nullptr); };
    if (grouped) {
      pend_pw.push_back(wgrad_args(c, B.c1, pl, N, b.Hin, b.Win, b.Hin, b.Win, yin, B.cin, U1, pl));
      pend_blocks.push_back(i);
      if (closes_group) { c.fork_armed(c.gstream()); flush_group(); }
    } else if (fork_each) { c.fork_armed(); if (late) { wg3(); wg2(); } wg1(); }
    // (10) gradient w.r.t. the block input -> Gnext.  Fused flow: the conv that completes it also applies the ReLU mask of
    //      the previous block's output (MASK2 with aux2 = yin) and, unless that block has a downsample branch, accumulates
    //      its BN3-backward sums (STATS3 with aux3 = its c3), so the next iteration starts at step (2).
    auto hand_over = [&](tf_conv_args& q) {
      if (!fused || i == 0) return;                        // block 0's input is the max-pool output: no ReLU in between
      const int ho_tile = tf::tuning().handover_tile;     // A/B knob: tile code of the hand-over data gradients
      if (ho_tile) q.tile = ho_tile;
      q.epi |= TF_EPI_MASK2; q.aux2 = yin;
      if (!A.blocks[i - 1].has_ds) { q.epi |= TF_EPI_STATS3; q.aux3 = P.blk[i - 1].c3; q.stat_out = P.blk[i - 1].b3.bst; }
    };
    if (B.has_ds) {
      if (fused) {
        const tf_bn_bwd_desc d = bwd_desc(c, B.ds, b.bd, b.b3.bst, 3, 2);      // the downsample BN's sums are row 2 of bn3's region
        if (fork_each) c.arm_fork();
        c.chk(tf_bn_bwd_apply_fused(dtype, Gcur, nullptr, b.d, &d, srows, Mout, c4, (float)Mout, T3, c.stream));
      } else {
        c.chk(tf_bn_bwd_apply(dtype, Gcur, b.y, b.d, b.bd.cA, b.bd.cB, b.bd.cD, Mout, c4, T3, c.stream));
      }
      if (fork_each) { c.fork_armed(); wgrad(c, B.ds, c4, N, b.Hin, b.Win, b.Hout, b.Wout, yin, B.cin, T3, c4, nullptr); }
      // r6: the stride-2 downsample gradient is nonzero on the even-even pixels of the block's input raster only.  Rounds 3-5 zeroed (or, with
      // the res3 head gradient, copied) a raster of that size, scattered into it, and the hand-over conv read it back as its residual: a
      // 96 MB fill + 49 MB copy by blit kernels at 2.6 TB/s and two more passes over those rasters.  Now the hand-over writes the raster first
      // (residual = the head gradient where there is one) and the scattered gradient ACCUMULATES IN PLACE (tf_conv2d: aux == y), applying the
      // same ReLU mask and adding its own share to the same BN-backward sums.  TINYFACES_DS_INPLACE_OFF=1: the old order.
      const bool ds_inplace = fused && i > 0 && B.stride == 2 && !tf::tuning().ds_inplace_off;
      if (ds_inplace) {
        conv_fill(a, dtype, 1, N, b.Hin, b.Win, pl, b.Hin, b.Win, B.cin, 1, 1, 0, B.cin, U1, b.w1t, Gnext);
        if (extra) { a.epi = TF_EPI_RES; a.aux = extra; }
        hand_over(a);
        c.chk(tf_conv2d(&a, c.stream));
        conv_fill(a, dtype, 1, N, b.Hout, b.Wout, c4, b.Hin, b.Win, B.cin, 1, B.stride, 0, B.cin, T3, b.wdt, Gnext);
        a.epi = TF_EPI_RES; a.aux = Gnext;
        hand_over(a);
        a.tile = 0;                                          // (the hand-over tile knob is about the pointwise conv above)
        c.chk(tf_conv2d(&a, c.stream));
      } else {
      conv_fill(a, dtype, 1, N, b.Hout, b.Wout, c4, b.Hin, b.Win, B.cin, 1, B.stride, 0, B.cin, T3, b.wdt, P.T4);
      if (extra) { a.epi = TF_EPI_RES; a.aux = extra; }
      c.chk(tf_conv2d(&a, c.stream));
      conv_fill(a, dtype, 1, N, b.Hin, b.Win, pl, b.Hin, b.Win, B.cin, 1, 1, 0, B.cin, U1, b.w1t, Gnext);
      a.epi = TF_EPI_RES; a.aux = P.T4;
      hand_over(a);
      c.chk(tf_conv2d(&a, c.stream));
      }
    } else {
      conv_fill(a, dtype, 1, N, b.Hin, b.Win, pl, b.Hin, b.Win, B.cin, 1, 1, 0, B.cin, U1, b.w1t, Gnext);
      if (fused) { a.epi = TF_EPI_RES; a.aux = Gcur; hand_over(a); }       // identity branch: Gcur is already g_y * (y > 0)
      else { a.epi = TF_EPI_JOIN; a.aux2 = b.y; a.aux3 = Gcur; }            // identity branch: + g_y * (y > 0)
      c.chk(tf_conv2d(&a, c.stream));
    }
    if (!fork_each && !grouped) {
      // alternative schedule: the block's four weight gradients start together once its data-gradient chain is enqueued
      // and overlap the NEXT block's chain (their operands live in this parity's buffers until block i-2 reuses them)
      c.fork();
      wg3(); wg2(); wg1();
      if (B.has_ds) wgrad(c, B.ds, c4, N, b.Hin, b.Win, b.Hout, b.Wout, yin, B.cin, T3, c4, nullptr);
    }
    if (!grouped) {
      // everything up to here: the weight / BN gradients of blocks >= i and of the heads.  With a group stream they are spread over two
      // queues: the event goes to the group stream, ordered behind the second stream's position (never the other way round: the second
      // stream must not wait for a group)
      if (c.gside && grad_event_registered(hooks, i)) {
        hipEvent_t e = c.mark(c.side); if (e) (void)hipStreamWaitEvent(c.gside, e, 0);
        record_grad_events(hooks, i, c.gside, c.rc);
      } else {
        record_grad_events(hooks, i, c.wstream(), c.rc);
      }
    }                                                // (grouped blocks: their events fire in flush_group)
    void* t = Gcur; Gcur = Gnext; Gnext = t;
  }

  // ---- stem
  const int M1 = N * P.H1 * P.W1;
  void* gz = P.T4;                          // main-stream scratch (its last reader, block 0's conv1 dgrad, is ahead on this stream)
  // r4: the statistic sums of the stem's BN backward ride in the max-pool backward (gz and x are in its registers): one pass over two 96 MB
  // tensors and one launch fewer at the very end of the chain (TINYFACES_POOL_STATS_OFF=1: the two-pass form)
  const bool pool_stats_off = tf::tuning().pool_stats_off;
  int nb = 0;
  if (fused && !pool_stats_off) {
    c.chk(tf_maxpool_bwd_stats(dtype, Gcur, P.pool_idx, P.cstem, P.bn_stem.scale, P.bn_stem.shift, N, P.H1, P.W1, 64, gz, P.partial_b, &nb, c.stream));
  } else {
    c.chk(tf_maxpool_bwd(dtype, Gcur, P.pool_idx, P.cstem, P.bn_stem.scale, P.bn_stem.shift, N, P.H1, P.W1, 64, gz, c.stream));
    nb = tf_colstats_blocks(M1, 64, dtype);
    c.chk(tf_colstats(dtype, gz, nullptr, P.cstem, nullptr, M1, 64, 64, P.partial_b, c.stream));
  }
  bn_backward_coefs(c, A.stem, 64, P.bn_stem, P.partial_b, nb, 2, 1, 64, (float)M1);
  // (r4: with the direct weight gradient the apply rides in that kernel's staging -- its output has no other reader)
  const bool stem_wgrad_direct = stem_direct && !stem_wgrad_im2col;
  const bool stem_apply_off = tf::tuning().stem_apply_separate;
  const bool stem_apply_fused = stem_wgrad_direct && !stem_apply_off;
  if (!stem_apply_fused) c.chk(tf_bn_bwd_apply(dtype, gz, nullptr, P.cstem, P.bn_stem.cA, P.bn_stem.cB, P.bn_stem.cD, M1, 64, gz, c.stream));
  // P.col still holds the im2col matrix of this forward (nothing else is carved from that range)
  {
    ConvUnit s = A.stem; s.stride = 1; s.pad = 0;
    c.fork();
    if (stem_wgrad_direct) {
      if (!c.grads_zeroed && hipMemsetAsync(c.G(A.stem.w), 0, (size_t)64 * 147 * 4, c.wstream()) != hipSuccess) c.chk(TF_ERR_LAUNCH);
      c.chk(tf_stem_wgrad(dtype, x, N, H, W, gz, stem_apply_fused ? P.cstem : nullptr, P.bn_stem.cA, P.bn_stem.cB, P.bn_stem.cD, c.G(A.stem.w), c.wstream()));
    } else wgrad(c, s, 64, 1, 1, M1, 1, M1, P.col, kStemK, gz, 64, nullptr, 147, 1, 147);
  }
  // (r5, measured and removed: returning WITHOUT joining this last kernel and running the SGD update of every other parameter beside it --
  //  1282.2 / 1284.0 against 1282.1 / 1281.2 img/s joined: both are HBM-bound, side by side they take as long as back to back)
  c.wait_on_main(c.mark_side());           // join: the caller's stream sees every weight gradient
  if (c.gside) c.wait_on_main(c.mark(c.gside));
  record_grad_events(hooks, -1, c.stream, c.rc);
  if (hipGetLastError() != hipSuccess && c.rc == TF_OK) c.rc = TF_ERR_LAUNCH;
  return c.rc;
}

// context-free form (rounds 1-3): the default context of the current device + the process-wide hooks registered with tf_detnet_set_*
extern "C" int tf_detnet_backward(int dtype, const float* x, int N, int H, int W, int nout, void* const* params, void* const* grads,
                                  const float* gout, void* grad_flat, size_t grad_flat_bytes, void* ws, size_t ws_bytes, void* stream_) {
  std::vector<int> blocks; std::vector<void*> events;
  for (const auto& e : g_grad_events) { blocks.push_back(e.first); events.push_back((void*)e.second); }
  tf_detnet_hooks h;
  h.blocks = blocks.data(); h.events = events.data(); h.n = (int)blocks.size(); h.fn = g_grad_cb; h.user = g_grad_cb_user; h.single_stream = g_force_single ? 1 : 0;
  return tf_detnet_backward_ctx(nullptr, &h, dtype, x, N, H, W, nout, params, grads, gout, grad_flat, grad_flat_bytes, ws, ws_bytes, stream_);
}
