// Greedy NMS in float64, index-exact vs torchvision.ops.nms' CPU kernel as called at
// tinyfaces/evaluation.py:84 (see tinyfaces_hip.h).  Compiled with -ffp-contract=off.
//
//   1 rank sort   rank[i] = #{j : s_j > s_i  or (s_j == s_i and j < i)}  -> stable descending
//                 order without a sorting network; O(N^2) compares, same order as the IoU work.
//   2 bit matrix  mask[i][w] bit j = IoU(sorted i, sorted 64w+j) > thr, j > i (upper triangle),
//                 one 64-thread wave per 64x64 block, column boxes staged in LDS.
//   3 scan        one workgroup walks the 64-box chunks in order: wave 0 resolves the chunk's
//                 diagonal 64x64 block in registers (readlane broadcast), then all waves OR the
//                 kept rows into the running `removed` bit-vector held in LDS.
//
// Batched form (tf_nms_f64_batched, configs[4] "batched multi-scale NMS"): S independent candidate lists (one per image of an
// evaluation batch, or one per pyramid level) in ONE set of three launches: segment s owns boxes [off[s], off[s+1]); blockIdx.y /
// .z selects the segment, so the serial scans of different segments run side by side on different CUs instead of back to back.
#include "common.h"

namespace {

constexpr int kMaxSeg = TF_NMS_MAX_SEGMENTS;
struct Segs {                     // passed by value as a kernel argument: no H2D copy, no device-side table to keep alive
  int off[kMaxSeg + 1];           // box offsets
  unsigned long long moff[kMaxSeg];   // offset (in 64-bit words) of the segment's bit matrix
};

// rank[i] = #{j : s_j > s_i or (s_j == s_i and j < i)} == position of i in the stable descending argsort
// (evaluation.py:84 -> torchvision nms sorts by score; ties keep input order).  32 boxes x 8 column slices per block:
// lanes with the same slice read the same LDS word (broadcast), so a wave touches two addresses per step.
__global__ void __launch_bounds__(256) nms_rank_kernel(const double* __restrict__ scores_all, const double* __restrict__ boxes_all,
                                                       const Segs sg, int* __restrict__ order_all, double* __restrict__ sboxes_all) {
  __shared__ double tile[2048];
  __shared__ int part[8][32];
  const int base = sg.off[blockIdx.y], n = sg.off[blockIdx.y + 1] - base;
  if ((int)blockIdx.x * 32 >= n) return;                 // block-uniform: the grid is sized for the largest segment
  const double* scores = scores_all + base;
  const double* boxes = boxes_all + 4 * (size_t)base;
  int* order = order_all + base;
  double* sboxes = sboxes_all + 4 * (size_t)base;
  const int li = threadIdx.x & 31, p = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + li;
  const double si = i < n ? scores[i] : 0.0;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 2048) {
    __syncthreads();
    for (int k = threadIdx.x; k < 2048; k += 256) tile[k] = (j0 + k < n) ? scores[j0 + k] : -1.0e300;
    __syncthreads();
    const int lim = min(2048, n - j0);
    // strictly-greater never counts the -1e300 padding; the tie term is masked by the index bound
#pragma unroll 4
    for (int k = p; k < lim; k += 8) {
      const double sj = tile[k];
      rank += (sj > si) || (sj == si && (j0 + k) < i);
    }
  }
  part[p][li] = rank;
  __syncthreads();
  if (p == 0 && i < n) {
    int r = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) r += part[q][li];
    order[r] = i;
    const double4 b = *reinterpret_cast<const double4*>(boxes + 4 * (size_t)i);
    *reinterpret_cast<double4*>(sboxes + 4 * (size_t)r) = b;
  }
}

__global__ void __launch_bounds__(64) nms_mask_kernel(const double* __restrict__ sboxes_all, const Segs sg, double thr,
                                                      unsigned long long* __restrict__ mask_all) {
  const int base = sg.off[blockIdx.z], n = sg.off[blockIdx.z + 1] - base, nwords = (n + 63) / 64;
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb || cb >= nwords) return;       // lower triangle never read; grid sized for the largest segment
  const double* sboxes = sboxes_all + 4 * (size_t)base;
  unsigned long long* mask = mask_all + sg.moff[blockIdx.z];
  __shared__ double cbox[64][4];
  __shared__ double carea[64];
  const int t = threadIdx.x;
  const int cj = cb * 64 + t;
  if (cj < n) {
    const double4 b = *reinterpret_cast<const double4*>(sboxes + 4 * (size_t)cj);
    cbox[t][0] = b.x; cbox[t][1] = b.y; cbox[t][2] = b.z; cbox[t][3] = b.w;
    carea[t] = (b.z - b.x) * (b.w - b.y);
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  const double4 a = *reinterpret_cast<const double4*>(sboxes + 4 * (size_t)i);
  const double iarea = (a.z - a.x) * (a.w - a.y);
  unsigned long long bits = 0;
  const int lim = min(64, n - cb * 64);
  for (int j = 0; j < lim; ++j) {
    if (cb * 64 + j <= i) continue;
    const double xx1 = fmax(a.x, cbox[j][0]), yy1 = fmax(a.y, cbox[j][1]);
    const double xx2 = fmin(a.z, cbox[j][2]), yy2 = fmin(a.w, cbox[j][3]);
    const double w = fmax(0.0, xx2 - xx1), h = fmax(0.0, yy2 - yy1);
    const double inter = w * h;
    const double ovr = inter / (iarea + carea[j] - inter);
    if (ovr > thr) bits |= 1ull << j;
  }
  mask[(size_t)i * nwords + cb] = bits;
}

// Greedy scan over the score-sorted boxes, 64 at a time.  Wave 0 walks the chain: it resolves chunk c against removed[c]
// (64 scalar steps on the diagonal word), then folds the rows it kept into removed[c+1] itself (one word per lane, wave
// OR-reduction), so the next chunk can start at once; its three global words per chunk (diagonal, next column, original
// index) are fetched one chunk AHEAD, so no memory latency sits on the chain.  Waves 1..15 trail one chunk behind and push
// the kept rows of chunk c-1 into removed[w], w >= c+1: a thread owns one word and a 13-row range, issues its (predicated)
// row loads together and ORs once into LDS.  One barrier per chunk joins the two.  PREFETCH (n <= 12288: every (word, row
// range) pair has its own thread): the trailing threads fetch the 13 rows of chunk c while wave 0 is still resolving it and
// only select + OR them once its keep bits are known, so the push carries no memory latency either.
template <bool PREFETCH>
__global__ void __launch_bounds__(1024) nms_scan_kernel(const unsigned long long* __restrict__ mask_all, const int* __restrict__ order_all,
                                                        const Segs sg, int nw_lo, int nw_hi, int64_t* __restrict__ keep_all,
                                                        int* __restrict__ num_keep_all) {
  extern __shared__ unsigned long long removed[];   // nwords + 2: [nwords] / [nwords+1] = keep bits of even / odd chunks
  const int base = sg.off[blockIdx.x], n = sg.off[blockIdx.x + 1] - base, nwords = (n + 63) / 64;
  // the two instantiations split the segments by size (nw_lo < nwords <= nw_hi); an empty segment keeps nothing
  if (n == 0) { if (threadIdx.x == 0 && nw_lo < 0) num_keep_all[blockIdx.x] = 0; return; }
  if (nwords <= nw_lo || nwords > nw_hi) return;
  const unsigned long long* mask = mask_all + sg.moff[blockIdx.x];
  const int* order = order_all + base;
  int64_t* keep = keep_all + base;
  int* num_keep = num_keep_all + blockIdx.x;
  for (int w = threadIdx.x; w < nwords + 2; w += blockDim.x) removed[w] = 0;
  __syncthreads();
  int kcount = 0;                                    // meaningful in wave 0 only
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave 0: operands of the chunk being resolved (fetched during the previous iteration)
  unsigned long long diag = 0, next = 0;
  unsigned long long pv[13];                         // PREFETCH: this thread's rows of the chunk in flight
#pragma unroll
  for (int j = 0; j < 13; ++j) pv[j] = 0;
  int oi = 0;
  if (wave == 0) {
    const int i = lane;
    if (i < n) { diag = mask[(size_t)i * nwords]; next = nwords > 1 ? mask[(size_t)i * nwords + 1] : 0ull; oi = order[i]; }
  }
  for (int c = 0; c < nwords; ++c) {
    if (wave == 0) {
      // prefetch chunk c+1
      unsigned long long pdiag = 0, pnext = 0;
      int poi = 0;
      {
        const int i2 = (c + 1) * 64 + lane;
        if (c + 1 < nwords && i2 < n) {
          pdiag = mask[(size_t)i2 * nwords + c + 1];
          pnext = (c + 2 < nwords) ? mask[(size_t)i2 * nwords + c + 2] : 0ull;
          poi = order[i2];
        }
      }
      // removed[c] is the same word in every lane: make it scalar so that the whole resolve runs on the scalar unit
      const unsigned long long remv = removed[c];
      const unsigned long long rem = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(remv >> 32)) << 32) |
                                     (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)remv);
      const int nvalid = min(64, n - c * 64);                // boxes past n do not exist
      const unsigned long long valid = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
      // walk the SURVIVORS only (one step per kept box, ~14 of 64 on the bench pyramid): the lowest candidate is kept
      // (everything that could suppress it is decided), then it strikes out the later boxes it overlaps
      unsigned long long cand = valid & ~rem, kb = 0;
      const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
      while (cand) {
        const int b = __builtin_ctzll(cand);
        // readlane returns a SIGNED int: go through unsigned or the low word sign-extends into the high word
        const unsigned long long d = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                                     (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dlo, b);
        kb |= 1ull << b;
        cand &= ~(d | (1ull << b));
      }
      const bool mine = (kb >> lane) & 1ull;
      if (mine) keep[kcount + __popcll(kb & ((1ull << lane) - 1ull))] = (int64_t)(base + oi);   // index into the concatenated input
      kcount += __popcll(kb);
      // rows kept in this chunk -> removed[c+1] (the only word the next resolve needs from this chunk)
      if (mine && next && c + 1 < nwords) atomicOr(&removed[c + 1], next);     // LDS atomics: cheaper than a 64-bit wave OR-reduction
      if (lane == 0) removed[nwords + (c & 1)] = kb;
      diag = pdiag; next = pnext; oi = poi;
    } else if (PREFETCH) {
      const int t = threadIdx.x - 64;                    // 0..959
      const int slice = t % 5, w = t / 5;                 // one word, rows 13*slice .. 13*slice+12 of every chunk
      if (c >= 1 && w >= c + 1 && w < nwords) {           // rows of chunk c-1 were fetched during the previous iteration
        const unsigned long long kb = removed[nwords + ((c - 1) & 1)];
        const unsigned int bits = (unsigned int)(kb >> (13 * slice)) & 0x1FFFu;
        unsigned long long acc = 0;
#pragma unroll
        for (int j = 0; j < 13; ++j) acc |= ((bits >> j) & 1u) ? pv[j] : 0ull;
        if (acc) atomicOr(&removed[w], acc);
      }
      if (w >= c + 2 && w < nwords) {
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          const int rl = 13 * slice + j, row = c * 64 + rl;
          pv[j] = (rl < 64 && row < n) ? mask[(size_t)row * nwords + w] : 0ull;
        }
      }
    } else if (c >= 1) {
      // trailing push of chunk c-1's kept rows into words >= c+1
      const unsigned long long kb = removed[nwords + ((c - 1) & 1)];
      const int t = threadIdx.x - 64;                    // 0..959
      const int slice = t % 5, wi = t / 5;                // 192 words per sweep; slice s owns rows 13s .. 13s+12 of the chunk
      const unsigned int bits = (unsigned int)(kb >> (13 * slice)) & 0x1FFFu;
      const size_t row0 = (size_t)(c - 1) * 64 + 13 * slice;
      if (bits) {
        for (int w = c + 1 + wi; w < nwords; w += 192) {
          unsigned long long v[13];
#pragma unroll
          for (int j = 0; j < 13; ++j) v[j] = ((bits >> j) & 1u) ? mask[(row0 + j) * nwords + w] : 0ull;
          unsigned long long acc = 0;
#pragma unroll
          for (int j = 0; j < 13; ++j) acc |= v[j];
          if (acc) atomicOr(&removed[w], acc);
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_keep = kcount;
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t nms_mask_words(const int* off, int S, unsigned long long* moff) {
  size_t words = 0;
  for (int s = 0; s < S; ++s) {
    const size_t n = (size_t)(off[s + 1] - off[s]);
    if (moff) moff[s] = words;
    words += n * ((n + 63) / 64);
  }
  return words;
}

extern "C" size_t tf_nms_batched_workspace_bytes(const int32_t* host_seg_offsets, int num_segments) {
  if (!host_seg_offsets || num_segments <= 0) return 256;
  const size_t n = (size_t)host_seg_offsets[num_segments];
  return align256(n * 4) + align256(n * 32) + align256(nms_mask_words(host_seg_offsets, num_segments, nullptr) * 8) + 256;
}

extern "C" size_t tf_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  const int32_t off[2] = {0, n};
  return tf_nms_batched_workspace_bytes(off, 1);
}

extern "C" int tf_nms_f64_batched(const double* boxes, const double* scores, const int32_t* host_seg_offsets, int num_segments,
                                  double iou_thresh, int64_t* keep_out, int32_t* num_keep, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int S = num_segments;
  if (!host_seg_offsets || S <= 0 || S > kMaxSeg || !num_keep || host_seg_offsets[0] != 0) return TF_ERR_ARG;
  Segs sg;
  int nmax = 0;
  for (int s = 0; s < S; ++s) {
    const int ns = host_seg_offsets[s + 1] - host_seg_offsets[s];
    if (ns < 0) return TF_ERR_ARG;
    if (ns > nmax) nmax = ns;
  }
  for (int s = 0; s <= kMaxSeg; ++s) sg.off[s] = host_seg_offsets[s < S ? s : S];
  for (int s = 0; s < kMaxSeg; ++s) sg.moff[s] = 0;
  const size_t words = nms_mask_words(host_seg_offsets, S, sg.moff);
  const int n = host_seg_offsets[S];
  if (n == 0) return hipMemsetAsync(num_keep, 0, 4 * (size_t)S, stream) == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
  if (!boxes || !scores || !keep_out) return TF_ERR_ARG;
  if (!ws || ws_bytes < tf_nms_batched_workspace_bytes(host_seg_offsets, S)) return TF_ERR_WORKSPACE;
  const int nwmax = (nmax + 63) / 64;
  if ((size_t)(nwmax + 2) * 8 > 64 * 1024) return TF_ERR_UNSUPPORTED;   // largest segment <= 524k boxes
  char* w = (char*)ws;
  int* order = (int*)w;                 w += align256((size_t)n * 4);
  double* sboxes = (double*)w;          w += align256((size_t)n * 32);
  unsigned long long* mask = (unsigned long long*)w;
  (void)words;
  hipLaunchKernelGGL(nms_rank_kernel, dim3((nmax + 31) / 32, S), dim3(256), 0, stream, scores, boxes, sg, order, sboxes);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nwmax, nwmax, S), dim3(64), 0, stream, sboxes, sg, iou_thresh, mask);
  // segments of <= 192 words (12 288 boxes) take the prefetching scan, larger ones the sweeping scan; a launch whose
  // size class is empty is skipped.  nw_lo = -1 also makes that launch the one that zeroes the count of empty segments.
  bool any_small = false, any_large = false;
  for (int s = 0; s < S; ++s) {
    const int nw = (host_seg_offsets[s + 1] - host_seg_offsets[s] + 63) / 64;
    if (nw <= 192) any_small = true; else any_large = true;
  }
  const size_t lds = (size_t)(nwmax + 2) * 8;
  if (any_small)
    hipLaunchKernelGGL(nms_scan_kernel<true>, dim3(S), dim3(1024), (size_t)((nwmax < 192 ? nwmax : 192) + 2) * 8, stream, mask, order, sg, -1, 192,
                       keep_out, num_keep);
  if (any_large)
    hipLaunchKernelGGL(nms_scan_kernel<false>, dim3(S), dim3(1024), lds, stream, mask, order, sg, any_small ? 192 : -1, 1 << 30, keep_out, num_keep);
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_nms_f64(const double* boxes, const double* scores, int n, double iou_thresh,
                          int64_t* keep_out, int32_t* num_keep, void* ws, size_t ws_bytes, void* stream_) {
  if (n < 0) return TF_ERR_ARG;
  const int32_t off[2] = {0, n};
  return tf_nms_f64_batched(boxes, scores, off, 1, iou_thresh, keep_out, num_keep, ws, ws_bytes, stream_);
}
