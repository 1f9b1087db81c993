// Greedy NMS in float64, index-exact vs torchvision.ops.nms' CPU kernel as called at
// tinyfaces/evaluation.py:84 (see tinyfaces_hip.h).  Compiled with -ffp-contract=off.
//
//   1 rank sort   rank[i] = #{j : s_j > s_i  or (s_j == s_i and j < i)}  -> stable descending
//                 order without a sorting network; O(N^2) compares, same order as the IoU work.
//   2 bit matrix  mask[i][w] bit j = IoU(sorted i, sorted 64w+j) > thr, j > i (upper triangle),
//                 one 64-thread wave per 64x64 block, column boxes staged in LDS.
//   3 scan        two levels (r3): per super-chunk of 1024 boxes one workgroup resolves the 16 chunks against the diagonal
//                 block held in LDS (wave 0 walks the survivors of a chunk on the scalar unit, readlane broadcast), then a
//                 chip-wide launch ORs the kept rows into the `removed` bit-vector of all later words.
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
  int woff[kMaxSeg];                  // offset (in 64-bit words) of the segment's removed / kept bit-vectors
};

// rank[i] = #{j : s_j > s_i or (s_j == s_i and j < i)} == position of i in the stable descending argsort
// (evaluation.py:84 -> torchvision nms sorts by score; ties keep input order).  128 boxes x 8 column slices per block, four boxes
// per thread: lanes with the same slice read the same LDS word (broadcast), and one LDS read feeds four compares.  The tie term
// costs nothing outside the block's own index range: for j < i the predicate is s_j >= s_i, for j > i it is s_j > s_i, so a
// 2048-score tile entirely below / above the block's 128 boxes takes ONE float64 compare per pair (r3: 0.94 -> 0.4 ms at N = 65 536).
// Q boxes per thread: 4 (128 per block) for long lists, 1 (32 per block: four times the blocks) below 16 384 boxes, where the grid
// of the wide form would leave most of the chip idle.
template <int Q>
__global__ void __launch_bounds__(256) nms_rank_kernel(const double* __restrict__ scores_all, const double* __restrict__ boxes_all,
                                                       const Segs sg, int* __restrict__ order_all, double* __restrict__ sboxes_all) {
  __shared__ double tile[2048];
  constexpr int kRankBoxes = 32 * Q;
  __shared__ int part[8][kRankBoxes];
  const int base = sg.off[blockIdx.y], n = sg.off[blockIdx.y + 1] - base;
  const int i0 = blockIdx.x * kRankBoxes;
  if (i0 >= n) return;                                    // block-uniform: the grid is sized for the largest segment
  const double* scores = scores_all + base;
  const double* boxes = boxes_all + 4 * (size_t)base;
  int* order = order_all + base;
  double* sboxes = sboxes_all + 4 * (size_t)base;
  const int li = threadIdx.x & 31, p = threadIdx.x >> 5;
  double si[Q]; int rank[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) { const int i = i0 + li + 32 * q; si[q] = i < n ? scores[i] : 0.0; rank[q] = 0; }
  for (int j0 = 0; j0 < n; j0 += 2048) {
    __syncthreads();
    for (int k = threadIdx.x; k < 2048; k += 256) tile[k] = (j0 + k < n) ? scores[j0 + k] : -1.0e300;
    __syncthreads();
    const int lim = min(2048, n - j0);
    if (j0 + 2048 <= i0) {                                // every j of the tile is below every i of the block: ties count
#pragma unroll 4
      for (int k = p; k < lim; k += 8) {
        const double sj = tile[k];
#pragma unroll
        for (int q = 0; q < Q; ++q) rank[q] += sj >= si[q];
      }
    } else if (j0 >= i0 + kRankBoxes) {                   // every j above every i: ties do not count (nor does the -1e300 padding)
#pragma unroll 4
      for (int k = p; k < lim; k += 8) {
        const double sj = tile[k];
#pragma unroll
        for (int q = 0; q < Q; ++q) rank[q] += sj > si[q];
      }
    } else {                                              // the tile that holds the block's own boxes: the full predicate
#pragma unroll 2
      for (int k = p; k < lim; k += 8) {
        const double sj = tile[k];
#pragma unroll
        for (int q = 0; q < Q; ++q) rank[q] += (sj > si[q]) || (sj == si[q] && (j0 + k) < i0 + li + 32 * q);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) part[p][li + 32 * q] = rank[q];
  __syncthreads();
  if (threadIdx.x < kRankBoxes) {
    const int i = i0 + threadIdx.x;
    if (i < n) {
      int r = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) r += part[q][threadIdx.x];
      order[r] = i;
      const double4 bx = *reinterpret_cast<const double4*>(boxes + 4 * (size_t)i);
      *reinterpret_cast<double4*>(sboxes + 4 * (size_t)r) = bx;
    }
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
    // disjoint boxes (the overwhelming majority of the N^2 / 2 pairs) have inter == 0 exactly, so ovr is 0 / union: +0, or NaN for
    // two zero-area boxes -- neither exceeds a threshold >= 0.  Skip the float64 division for them (it is half of the work per
    // pair); everything else takes the reference's exact expression, division included (index-exact parity)
    if (inter == 0.0 && thr >= 0.0) continue;
    const double ovr = inter / (iarea + carea[j] - inter);
    if (ovr > thr) bits |= 1ull << j;
  }
  mask[(size_t)i * nwords + cb] = bits;
}

// ---- greedy scan, two levels (round 3).  The keep decision of a box depends on every earlier kept box, so the scan is a
// serial chain over the score-sorted list; what is NOT serial is moving the suppression bits of the kept rows.  Round 2 walked
// the 64-box chunks in ONE workgroup with a global-memory push per chunk: 7.2 us per chunk at N = 65 536 (1024 chunks, 7.4 ms of a
// 10.8 ms NMS; profiles/r03_nms.txt).  Now:
//   * a SUPER-CHUNK = 16 chunks = 1024 boxes.  nms_resolve_kernel loads the super-chunk's diagonal block of the bit matrix
//     (1024 rows x 16 words = 128 KiB) into LDS once, then resolves its 16 chunks back to back: wave 0 walks the survivors of
//     the chunk's diagonal word (scalar unit, one step per KEPT box), all 16 waves push the kept rows into the remaining words
//     of the super-chunk -- LDS reads and LDS atomics only, two barriers per chunk, no global latency on the chain;
//   * nms_push_kernel folds the kept rows of the finished super-chunk into removed[w] of every LATER word, spread over the
//     chip (one thread per word, 16 row groups, loads issued back to back, one 64-bit atomic OR per thread).
// One resolve + one push launch per super-chunk (64 + 63 launches at N = 65 536); segments of a batched call ride along in the
// grid.  Same results bit for bit: the kept set is defined by the bit matrix alone.
constexpr int kSC = 16;                         // chunks (64-bit words) per super-chunk
constexpr int kDiagPitch = kSC + 1;             // words per LDS row: 17 (odd pitch: the per-chunk column reads spread over the banks)

__global__ void __launch_bounds__(1024) nms_resolve_kernel(const unsigned long long* __restrict__ mask_all, const int* __restrict__ order_all,
                                                           const Segs sg, int sc, const unsigned long long* __restrict__ removed_all,
                                                           unsigned long long* __restrict__ kept_all, int64_t* __restrict__ keep_all,
                                                           int* __restrict__ num_keep_all) {
  extern __shared__ unsigned long long lds[];       // [1024][17] diagonal block, then rem[16], kbs[16]
  const int base = sg.off[blockIdx.x], n = sg.off[blockIdx.x + 1] - base, nwords = (n + 63) / 64;
  const int w0 = sc * kSC;
  if (n == 0) { if (threadIdx.x == 0 && sc == 0) num_keep_all[blockIdx.x] = 0; return; }
  if (w0 >= nwords) return;
  const int nw = min(kSC, nwords - w0);
  const unsigned long long* mask = mask_all + sg.moff[blockIdx.x];
  const unsigned long long* removed = removed_all + sg.woff[blockIdx.x];
  unsigned long long* kept = kept_all + sg.woff[blockIdx.x];
  const int* order = order_all + base;
  int64_t* keep = keep_all + base;
  unsigned long long* diag = lds;
  unsigned long long* rem = lds + 1024 * kDiagPitch;
  unsigned long long* kbs = rem + kSC;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  {
    // the diagonal block: rows w0*64 .. +1023, words w0 .. w0 + nw - 1 of each (128 contiguous bytes per row).  Eight threads per
    // row, 16 bytes each -> whole lines per request (one thread per row was 16 strided 8-byte requests: 10 us of the 32 per launch).
    // Words left of a row's own chunk belong to the lower triangle, which the mask kernel never writes: never loaded, never used.
    const int part = t & 7;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int lr = ps * 128 + (t >> 3), r = w0 * 64 + lr;
      const int j = part * 2;
      if (r < n && j + 1 >= (lr >> 6) && j < nw) {
        const unsigned long long* src = mask + (size_t)r * nwords + w0 + j;
        unsigned long long v0 = src[0], v1 = (j + 1 < nw) ? src[1] : 0ull;
        diag[lr * kDiagPitch + j] = v0; diag[lr * kDiagPitch + j + 1] = v1;
      }
    }
    if (t < kSC) rem[t] = t < nw ? removed[w0 + t] : 0ull;
  }
  int kcount = sc == 0 ? 0 : num_keep_all[blockIdx.x];      // running keep count of this segment (meaningful in wave 0)
  __syncthreads();
  for (int c = 0; c < nw; ++c) {
    if (wave == 0) {
      const int i = (w0 + c) * 64 + lane;
      const unsigned long long dg = i < n ? diag[(c * 64 + lane) * kDiagPitch + c] : 0ull;
      const int oi = i < n ? order[i] : 0;
      const unsigned long long remv = rem[c];
      const unsigned long long rm = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(remv >> 32)) << 32) |
                                    (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)remv);
      const int nvalid = min(64, n - (w0 + c) * 64);
      const unsigned long long valid = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
      // walk the SURVIVORS only: the lowest candidate is kept (everything that could suppress it is decided), then it strikes out
      // the later boxes of the chunk it overlaps
      unsigned long long cand = valid & ~rm, kb = 0;
      const unsigned int dlo = (unsigned int)dg, dhi = (unsigned int)(dg >> 32);
      while (cand) {
        const int b = __builtin_ctzll(cand);
        // readlane returns a SIGNED int: go through unsigned or the low word sign-extends into the high word
        const unsigned long long d = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                                     (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dlo, b);
        kb |= 1ull << b;
        cand &= ~(d | (1ull << b));
      }
      if ((kb >> lane) & 1ull) keep[kcount + __popcll(kb & ((1ull << lane) - 1ull))] = (int64_t)(base + oi);   // index into the concatenated input
      kcount += __popcll(kb);
      if (lane == 0) kbs[c] = kb;
    }
    __syncthreads();
    {
      // push the kept rows of chunk c into the later words of this super-chunk: thread = (row of the chunk, word)
      const int row = lane, j = c + 1 + wave;
      const unsigned long long kb = kbs[c];
      if (j < nw && ((kb >> row) & 1ull)) {
        const unsigned long long v = diag[(c * 64 + row) * kDiagPitch + j];
        if (v) atomicOr(&rem[j], v);
      }
    }
    __syncthreads();
  }
  if (t < nw) kept[w0 + t] = kbs[t];
  if (t == 0) num_keep_all[blockIdx.x] = kcount;
}

// removed[w] |= OR of the kept rows of super-chunk sc, for every word w of the LATER super-chunks.
// grid: (256-word tiles, 16 chunks of the super-chunk, segments)
__global__ void __launch_bounds__(256) nms_push_kernel(const unsigned long long* __restrict__ mask_all, const Segs sg, int sc,
                                                       const unsigned long long* __restrict__ kept_all, unsigned long long* __restrict__ removed_all) {
  const int base = sg.off[blockIdx.z], n = sg.off[blockIdx.z + 1] - base, nwords = (n + 63) / 64;
  const int chunk = sc * kSC + blockIdx.y;
  const int w = (sc + 1) * kSC + blockIdx.x * 256 + threadIdx.x;
  if (chunk >= nwords || (sc + 1) * kSC + (int)blockIdx.x * 256 >= nwords) return;     // block-uniform
  const unsigned long long kbv = kept_all[sg.woff[blockIdx.z] + chunk];
  unsigned long long kb = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(kbv >> 32)) << 32) |
                          (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)kbv);
  if (kb == 0 || w >= nwords) return;
  const unsigned long long* col = mask_all + sg.moff[blockIdx.z] + (size_t)chunk * 64 * nwords + w;
  unsigned long long acc = 0;
  while (kb) {                                     // wave-uniform walk over the kept rows: the loads are independent and issue back to back
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (kb) { const int b = __builtin_ctzll(kb); kb &= kb - 1; v[u] = col[(size_t)b * nwords]; } else v[u] = 0ull;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc |= v[u];
  }
  if (acc) atomicOr(removed_all + sg.woff[blockIdx.z] + w, acc);
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
// words of one bit-vector over all segments (removed / kept: one bit per box, every segment starting on a word)
static size_t nms_vec_words(const int* off, int S, int* woff) {
  size_t words = 0;
  for (int s = 0; s < S; ++s) {
    if (woff) woff[s] = (int)words;
    words += (size_t)(off[s + 1] - off[s] + 63) / 64;
  }
  return words;
}

extern "C" size_t tf_nms_batched_workspace_bytes(const int32_t* host_seg_offsets, int num_segments) {
  if (!host_seg_offsets || num_segments <= 0) return 256;
  const size_t n = (size_t)host_seg_offsets[num_segments];
  return align256(n * 4) + align256(n * 32) + 2 * align256(nms_vec_words(host_seg_offsets, num_segments, nullptr) * 8) +
         align256(nms_mask_words(host_seg_offsets, num_segments, nullptr) * 8) + 256;
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
  for (int s = 0; s < kMaxSeg; ++s) { sg.moff[s] = 0; sg.woff[s] = 0; }
  const size_t words = nms_mask_words(host_seg_offsets, S, sg.moff);
  const size_t vwords = nms_vec_words(host_seg_offsets, S, sg.woff);
  const int n = host_seg_offsets[S];
  if (n == 0) return hipMemsetAsync(num_keep, 0, 4 * (size_t)S, stream) == hipSuccess ? TF_OK : TF_ERR_LAUNCH;
  if (!boxes || !scores || !keep_out) return TF_ERR_ARG;
  if (!ws || ws_bytes < tf_nms_batched_workspace_bytes(host_seg_offsets, S)) return TF_ERR_WORKSPACE;
  const int nwmax = (nmax + 63) / 64;
  if (nmax > 524160) return TF_ERR_UNSUPPORTED;          // largest segment: 34 GB of bit matrix; a candidate list that long is a threshold mistake
  char* w = (char*)ws;
  int* order = (int*)w;                 w += align256((size_t)n * 4);
  double* sboxes = (double*)w;          w += align256((size_t)n * 32);
  unsigned long long* removed = (unsigned long long*)w;   w += align256(vwords * 8);
  unsigned long long* kept = (unsigned long long*)w;      w += align256(vwords * 8);
  unsigned long long* mask = (unsigned long long*)w;
  (void)words;
  if (hipMemsetAsync(removed, 0, vwords * 8, stream) != hipSuccess) return TF_ERR_LAUNCH;
  if (nmax >= 16384) hipLaunchKernelGGL(nms_rank_kernel<4>, dim3((nmax + 127) / 128, S), dim3(256), 0, stream, scores, boxes, sg, order, sboxes);
  else hipLaunchKernelGGL(nms_rank_kernel<1>, dim3((nmax + 31) / 32, S), dim3(256), 0, stream, scores, boxes, sg, order, sboxes);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nwmax, nwmax, S), dim3(64), 0, stream, sboxes, sg, iou_thresh, mask);
  // two-level scan: per super-chunk of 1024 boxes one resolve launch (one workgroup per segment, the diagonal block in LDS) and,
  // while later super-chunks exist, one push launch over the chip
  static tf::PerDevice attr_set;
  const size_t lds = ((size_t)1024 * kDiagPitch + 2 * kSC) * 8;         // 139.5 KiB
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int nsc = (nwmax + kSC - 1) / kSC;
  for (int sc = 0; sc < nsc; ++sc) {
    hipLaunchKernelGGL(nms_resolve_kernel, dim3(S), dim3(1024), lds, stream, mask, order, sg, sc, removed, kept, keep_out, num_keep);
    const int later = nwmax - (sc + 1) * kSC;
    if (later > 0) hipLaunchKernelGGL(nms_push_kernel, dim3((later + 255) / 256, kSC, S), dim3(256), 0, stream, mask, sg, sc, kept, removed);
  }
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_nms_f64(const double* boxes, const double* scores, int n, double iou_thresh,
                          int64_t* keep_out, int32_t* num_keep, void* ws, size_t ws_bytes, void* stream_) {
  if (n < 0) return TF_ERR_ARG;
  const int32_t off[2] = {0, n};
  return tf_nms_f64_batched(boxes, scores, off, 1, iou_thresh, keep_out, num_keep, ws, ws_bytes, stream_);
}
