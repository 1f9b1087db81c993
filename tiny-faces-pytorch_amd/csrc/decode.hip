// Score map -> candidate boxes: sigmoid, threshold, ORDERED compaction, regression refinement.
// Replaces tinyfaces/evaluation.py:61-78 + tinyfaces/models/utils.py:4-100 (see tinyfaces_hip.h).
// Compiled with -ffp-contract=off (the f64 box arithmetic must round like numpy's).
//
// HBM-read bound: each block stages a [5nt][64-pixel] slab of the NCHW map through LDS with
// fully coalesced 256-byte row reads, then walks it in the reference's (y, x, template) order.
// Ordered compaction = per-block counts -> one-block exclusive scan -> per-block ordered write
// (ballot/popcount prefix inside the wave, LDS across the 4 waves).
#include "common.h"

namespace {

constexpr int PXB = 64;   // pixels per block

struct DecParams {
  const float* score; int nt, H, W;
  const double* tpl; int tstride;
  const uint8_t* valid_x; const uint8_t* valid_t;
  float thr; double factor; int sty, stx, ofy, ofx;
  double* dets; int* count; int cap;
  int* blk; int nblk;
};

__device__ __forceinline__ bool is_candidate(const DecParams& p, float logit, int x, int t) {
  // torch.sigmoid in fp32 (evaluation.py:62), then prob > prob_thresh compared in fp32 (utils.py:46);
  // masked entries are 0.0 in the reference (utils.py:44), i.e. pass only if thr < 0.
  float prob = 1.0f / (1.0f + expf(-logit));
  if (!(p.valid_x[x] && p.valid_t[t])) prob = 0.0f;
  return prob > p.thr;
}

template <int PHASE>   // 0 = count, 1 = write
__global__ void __launch_bounds__(256) decode_kernel(DecParams p) {
  extern __shared__ float slab[];                 // [5nt][PXB]
  __shared__ int wave_cnt[4];
  __shared__ int run_base;
  const int HW = p.H * p.W;
  const int p0 = blockIdx.x * PXB;
  const int npx = min(PXB, HW - p0);
  const int nch = PHASE == 0 ? p.nt : 5 * p.nt;
  for (int e = threadIdx.x; e < nch * PXB; e += 256) {
    const int c = e / PXB, px = e - c * PXB;
    slab[e] = px < npx ? p.score[(size_t)c * HW + p0 + px] : -1e30f;
  }
  if (threadIdx.x == 0) run_base = PHASE == 0 ? 0 : p.blk[blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int total = PXB * p.nt;
  int local_total = 0;
  for (int e0 = 0; e0 < total; e0 += 256) {
    const int e = e0 + threadIdx.x;               // order: (pixel, template), template fastest
    const int px = e / p.nt, t = e - px * p.nt;
    bool c = false;
    float logit = 0.f;
    int x = 0, y = 0;
    if (e < total && px < npx) {
      const int pix = p0 + px;
      y = pix / p.W; x = pix - y * p.W;
      logit = slab[t * PXB + px];
      c = is_candidate(p, logit, x, t);
    }
    const unsigned long long bal = __ballot(c);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int v = wave_cnt[w]; all += v; if (w < wave) before += v; }
    if (PHASE == 1 && c) {
      const int pos = run_base + local_total + before + __popcll(bal & ((1ull << lane) - 1ull));
      if (pos < p.cap) {
        const double* tp = p.tpl + (size_t)t * p.tstride;
        const double cy = (double)(y * p.sty + p.ofy), cx = (double)(x * p.stx + p.ofx);   // utils.py:52-53
        const double cw = tp[2] - tp[0] + 1.0, ch = tp[3] - tp[1] + 1.0;                     // :54-55
        const float tx = slab[(1 * p.nt + t) * PXB + px], ty = slab[(2 * p.nt + t) * PXB + px];
        const float tw = slab[(3 * p.nt + t) * PXB + px], th = slab[(4 * p.nt + t) * PXB + px];
        const double rcx = cx + cw * (double)tx, rcy = cy + ch * (double)ty;                 // :81-85
        // np.exp on a float32 array is evaluated in float32 (:87): round the f64 result to f32
        const double rcw = cw * (double)(float)exp((double)tw), rch = ch * (double)(float)exp((double)th);
        double* d = p.dets + 5 * (size_t)pos;
        d[0] = (rcx - rcw / 2.0) * p.factor; d[1] = (rcy - rch / 2.0) * p.factor;          // :97-98, :73-74
        d[2] = (rcx + rcw / 2.0) * p.factor; d[3] = (rcy + rch / 2.0) * p.factor;
        d[4] = (double)logit;                                                                // :49 raw logit
      }
    }
    local_total += all;
    __syncthreads();
  }
  if (PHASE == 0 && threadIdx.x == 0) p.blk[blockIdx.x] = local_total;
}

// exclusive scan of block counts (+ current *count as base); updates *count
__global__ void __launch_bounds__(1024) decode_scan_kernel(int* blk, int nblk, int* count) {
  __shared__ int part[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = *count;
  __syncthreads();
  for (int b0 = 0; b0 < nblk; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nblk ? blk[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      int add = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    const int incl = part[threadIdx.x];
    if (i < nblk) blk[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry;   // true total; rows >= cap are not written (caller checks)
}

}  // namespace

extern "C" size_t tf_decode_workspace_bytes(int H, int W, int nt) {
  (void)nt;
  size_t nblk = ((size_t)H * W + PXB - 1) / PXB;
  return nblk * 4 + 256;
}

extern "C" int tf_decode_compact(const float* score, int nt, int H, int W, const double* templates, int tstride,
                                 const uint8_t* valid_x, const uint8_t* valid_t, float prob_thresh, double scale,
                                 int sty, int stx, int ofy, int ofx, double* dets, int32_t* count, int cap,
                                 void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!score || !templates || !valid_x || !valid_t || !dets || !count || nt <= 0 || H <= 0 || W <= 0 || scale <= 0) return TF_ERR_ARG;
  if (!ws || ws_bytes < tf_decode_workspace_bytes(H, W, nt)) return TF_ERR_WORKSPACE;
  DecParams p;
  p.score = score; p.nt = nt; p.H = H; p.W = W; p.tpl = templates; p.tstride = tstride;
  p.valid_x = valid_x; p.valid_t = valid_t; p.thr = prob_thresh; p.factor = 1.0 / scale;   // utils.py:73
  p.sty = sty; p.stx = stx; p.ofy = ofy; p.ofx = ofx; p.dets = dets; p.count = count; p.cap = cap;
  p.blk = (int*)ws; p.nblk = (H * W + PXB - 1) / PXB;
  hipLaunchKernelGGL(decode_kernel<0>, dim3(p.nblk), dim3(256), (size_t)nt * PXB * 4, stream, p);
  hipLaunchKernelGGL(decode_scan_kernel, dim3(1), dim3(1024), 0, stream, p.blk, p.nblk, count);
  hipLaunchKernelGGL(decode_kernel<1>, dim3(p.nblk), dim3(256), (size_t)5 * nt * PXB * 4, stream, p);
  TF_CHECK_LAUNCH();
  return TF_OK;
}
