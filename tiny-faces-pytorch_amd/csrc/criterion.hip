// DetectionCriterion forward + backward on device: OHEM, balance sampling, masked
// SoftMargin / SmoothL1 sums and d(total)/d(output), no host round trip.
// Replaces tinyfaces/models/loss.py:47-93 and tinyfaces/models/utils.py:103-163.
//
// Elements of one image are walked in C-order over (template, y, x) -- the order np.where /
// ravel_multi_index give in balance_sampling (utils.py:115-117,127-129) -- so "rank of a
// positive/negative label" means the same thing here and in the reference.
//   K1  OHEM in place (loss.py:59-63) + per-block counts of +1 / -1 labels
//   K2  per image: exclusive scan of the block counts, totals, and (if no keep flags are
//       injected) a uniformly random keep-subset of size <= max via rejection sampling into
//       a bitmap (one wave, counter RNG)
//   K3  ordered rank of every label (ballot prefix), sampling decision, loss terms, gradient;
//       every element of grad_out is written exactly once (no memset needed)
#include "common.h"

namespace {

constexpr int EPB = 1024;   // elements per block (4 per thread, 256-strided => coalesced)

struct CritParams {
  const float* out; float* cls; const float* reg;
  int B, nt, H, W, E, nblk;
  float ohem; int max_pos, max_neg; float reg_weight;
  const uint8_t* pos_keep; const uint8_t* neg_keep; uint64_t seed;
  float* label_out; float* grad; double* loss; int* counts_out;
  int* blkcnt;            // [B][nblk][2] counts, then exclusive offsets in place
  int* totals;            // [B][2]
  unsigned int* bitmap;   // [B][2][words]
  int words;
};

__device__ __forceinline__ float soft_margin(float s, float y) { return log1pf(expf(-s * y)); }   // ATen soft_margin_loss

__global__ void __launch_bounds__(256) crit_ohem_count_kernel(CritParams p) {
  const int b = blockIdx.y, blk = blockIdx.x;
  __shared__ int wsum[4][2];
  int npos = 0, nneg = 0;
#pragma unroll
  for (int k = 0; k < EPB / 256; ++k) {
    const int e = blk * EPB + k * 256 + threadIdx.x;
    if (e < p.E) {
      const size_t ci = (size_t)b * p.E + e;
      float y = p.cls[ci];
      const float s = p.out[(size_t)b * 5 * p.E + e];     // channels [0, nt) are the class logits
      if (soft_margin(s, y) < p.ohem) { y = 0.f; p.cls[ci] = 0.f; }     // loss.py:60-62 (in place)
      npos += (y == 1.f); nneg += (y == -1.f);
    }
  }
  npos = (int)tf::wave_sum((float)npos); nneg = (int)tf::wave_sum((float)nneg);
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = npos; wsum[threadIdx.x >> 6][1] = nneg; }
  __syncthreads();
  if (threadIdx.x < 2) {
    int* c = p.blkcnt + ((size_t)b * p.nblk + blk) * 2;
    c[threadIdx.x] = wsum[0][threadIdx.x] + wsum[1][threadIdx.x] + wsum[2][threadIdx.x] + wsum[3][threadIdx.x];
  }
}

// one block (64 threads = one wave) per image
__global__ void __launch_bounds__(64) crit_scan_sample_kernel(CritParams p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int* c = p.blkcnt + (size_t)b * p.nblk * 2;
  int carry[2] = {0, 0};
  for (int i0 = 0; i0 < p.nblk; i0 += 64) {
    const int i = i0 + lane;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = i < p.nblk ? c[2 * i + k] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      if (i < p.nblk) c[2 * i + k] = carry[k] + incl - v;
      carry[k] += __shfl(incl, 63, 64);
    }
  }
  if (lane == 0) {
    p.totals[2 * b] = carry[0]; p.totals[2 * b + 1] = carry[1];
    if (p.counts_out) { p.counts_out[2 * b] = carry[0]; p.counts_out[2 * b + 1] = carry[1]; }
  }
  // random keep-subsets (bitmaps are zeroed by the host-side memset before this kernel)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint8_t* inj = k == 0 ? p.pos_keep : p.neg_keep;
    const int n = carry[k], want = k == 0 ? p.max_pos : p.max_neg;
    if (inj || n <= want) continue;
    unsigned int* bm = p.bitmap + ((size_t)b * 2 + k) * p.words;
    int have = 0;
    for (unsigned round = 0; have < want; ++round) {
      const int need = want - have;
      bool fresh = false;
      if (lane < need) {
        const uint64_t h = tf::hash4(p.seed, ((uint64_t)b << 1) | k, round, lane);
        const unsigned int r = (unsigned int)(((h >> 32) * (uint64_t)n) >> 32);     // uniform in [0, n)
        const unsigned int bit = 1u << (r & 31);
        fresh = !(atomicOr(&bm[r >> 5], bit) & bit);
      }
      have += __popcll(__ballot(fresh));
    }
  }
}

__global__ void __launch_bounds__(256) crit_apply_kernel(CritParams p) {
  const int b = blockIdx.y, blk = blockIdx.x;
  __shared__ int wcnt[4][2];
  __shared__ double wloss[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int* off = p.blkcnt + ((size_t)b * p.nblk + blk) * 2;
  int base_pos = off[0], base_neg = off[1];
  const int npos = p.totals[2 * b], nneg = p.totals[2 * b + 1];
  const bool all_pos = npos <= p.max_pos, all_neg = nneg <= p.max_neg;     // utils.py:119,131
  const unsigned int* bm_pos = p.bitmap + ((size_t)b * 2) * p.words;
  const unsigned int* bm_neg = bm_pos + p.words;
  double lcls = 0.0, lreg = 0.0;
  for (int k = 0; k < EPB / 256; ++k) {
    const int e = blk * EPB + k * 256 + threadIdx.x;
    const bool in = e < p.E;
    const size_t ci = (size_t)b * p.E + (in ? e : 0);
    float y = in ? p.cls[ci] : 0.f;
    const bool isp = y == 1.f, isn = y == -1.f;
    const unsigned long long bp = __ballot(isp), bn = __ballot(isn);
    if (lane == 0) { wcnt[wave][0] = __popcll(bp); wcnt[wave][1] = __popcll(bn); }
    __syncthreads();
    int before_p = 0, before_n = 0, all_p = 0, all_n = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      all_p += wcnt[w][0]; all_n += wcnt[w][1];
      if (w < wave) { before_p += wcnt[w][0]; before_n += wcnt[w][1]; }
    }
    const unsigned long long lower = (1ull << lane) - 1ull;
    if (isp && !all_pos) {
      const int r = base_pos + before_p + __popcll(bp & lower);
      const bool keep = p.pos_keep ? p.pos_keep[(size_t)b * p.E + r] != 0 : (bm_pos[r >> 5] >> (r & 31)) & 1u;
      if (!keep) y = 0.f;
    }
    if (isn && !all_neg) {
      const int r = base_neg + before_n + __popcll(bn & lower);
      const bool keep = p.neg_keep ? p.neg_keep[(size_t)b * p.E + r] != 0 : (bm_neg[r >> 5] >> (r & 31)) & 1u;
      if (!keep) y = 0.f;
    }
    base_pos += all_p; base_neg += all_n;
    if (in) {
      if (p.label_out) p.label_out[ci] = y;
      const size_t oi = (size_t)b * 5 * p.E + e;
      const float s = p.out[oi];
      float g = 0.f;
      if (y != 0.f) {                                           // loss.py:77-79
        const float z = expf(-s * y);
        lcls += (double)log1pf(z);
        g = -y * z / (1.f + z);                                 // ATen soft_margin_loss_backward
      }
      p.grad[oi] = g;
#pragma unroll
      for (int c = 1; c <= 4; ++c) {                            // tx | ty | tw | th blocks (loss.py:66-67,83)
        const size_t ri = oi + (size_t)c * p.E;
        float gr = 0.f;
        if (y > 0.f) {
          const float d = p.out[ri] - p.reg[(size_t)b * 4 * p.E + (size_t)(c - 1) * p.E + e];
          const float ad = fabsf(d);
          lreg += (double)(ad < 1.f ? 0.5f * d * d : ad - 0.5f);     // SmoothL1, beta = 1
          gr = p.reg_weight * (d < -1.f ? -1.f : (d > 1.f ? 1.f : d));
        }
        p.grad[ri] = gr;
      }
    }
    __syncthreads();
  }
  lcls = tf::wave_sum(lcls); lreg = tf::wave_sum(lreg);
  if (lane == 0) { wloss[wave][0] = lcls; wloss[wave][1] = lreg; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const double v = wloss[0][threadIdx.x] + wloss[1][threadIdx.x] + wloss[2][threadIdx.x] + wloss[3][threadIdx.x];
    if (v != 0.0) atomicAdd(&p.loss[threadIdx.x], v);
  }
}

size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t tf_criterion_workspace_bytes(int B, int nt, int H, int W) {
  const size_t E = (size_t)nt * H * W, nblk = (E + EPB - 1) / EPB, words = (E + 31) / 32;
  return a256(B * nblk * 2 * 4) + a256(B * 2 * 4) + a256(B * 2 * words * 4) + 256;
}

extern "C" int tf_criterion_fwd_bwd(const float* output, float* class_map, const float* reg_map,
                                    int B, int nt, int H, int W, float ohem_thresh, int max_pos, int max_neg,
                                    float reg_weight, const uint8_t* pos_keep, const uint8_t* neg_keep, uint64_t seed,
                                    float* label_out, float* grad_out, double* loss_out, int32_t* counts_out,
                                    void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!output || !class_map || !reg_map || !grad_out || !loss_out || B <= 0 || nt <= 0 || H <= 0 || W <= 0) return TF_ERR_ARG;
  if (!ws || ws_bytes < tf_criterion_workspace_bytes(B, nt, H, W)) return TF_ERR_WORKSPACE;
  CritParams p;
  p.out = output; p.cls = class_map; p.reg = reg_map; p.B = B; p.nt = nt; p.H = H; p.W = W;
  p.E = nt * H * W; p.nblk = (p.E + EPB - 1) / EPB;
  p.ohem = ohem_thresh; p.max_pos = max_pos; p.max_neg = max_neg; p.reg_weight = reg_weight;
  p.pos_keep = pos_keep; p.neg_keep = neg_keep; p.seed = seed;
  p.label_out = label_out; p.grad = grad_out; p.loss = loss_out; p.counts_out = counts_out;
  p.words = (p.E + 31) / 32;
  char* w = (char*)ws;
  p.blkcnt = (int*)w;            w += a256((size_t)B * p.nblk * 2 * 4);
  p.totals = (int*)w;            w += a256((size_t)B * 2 * 4);
  p.bitmap = (unsigned int*)w;
  if (hipMemsetAsync(p.bitmap, 0, (size_t)B * 2 * p.words * 4, stream) != hipSuccess) return TF_ERR_LAUNCH;
  if (hipMemsetAsync(loss_out, 0, 2 * sizeof(double), stream) != hipSuccess) return TF_ERR_LAUNCH;
  dim3 grid(p.nblk, B);
  hipLaunchKernelGGL(crit_ohem_count_kernel, grid, dim3(256), 0, stream, p);
  hipLaunchKernelGGL(crit_scan_sample_kernel, dim3(B), dim3(64), 0, stream, p);
  hipLaunchKernelGGL(crit_apply_kernel, grid, dim3(256), 0, stream, p);
  TF_CHECK_LAUNCH();
  return TF_OK;
}
