// dense_overlap anchor-IoU + heat-map target assignment, fused, float64-exact.
// Replaces tinyfaces/datasets/dense_overlap.py:4-75 + processor.py:114-277 (see tinyfaces_hip.h).
// Compiled with -ffp-contract=off: every f64 op must round exactly like numpy's.
//
// HBM-write bound by design: one thread per (template, y, x) anchor, x fastest so the
// [B][5nt][vsy][vsx] float maps are written fully coalesced; boxes are wave-uniform scalar
// loads; the (vsy,vsx,nt,G) IoU tensor only ever exists in registers.
// Three stream-ordered phases:
//   A  per anchor: best GT (first arg-max of perturbed IoU), labels, regression targets;
//      per GT: atomicMax of the perturbed-IoU bit pattern (only values > neg_thresh matter)
//   B  per anchor: recompute, atomicMin the C-order flat index of anchors that hit that max
//   C  per GT: label its best anchor +1 (processor.py:252-257) unless padded (:272-273)
#include "common.h"

namespace {

struct TgtParams {
  const double* boxes; const int32_t* box_off; const double* tpl; int nt, tstride;
  int vsy, vsx, ofy, ofx, sty, stx;
  const int32_t* paste; const int32_t* flips;
  const double* noise; const int64_t* noise_off; uint64_t seed;
  double pos_thresh, neg_thresh;
  float* cls; float* reg;
  unsigned long long* gmax; unsigned int* gidx;
};

// IoU of anchor (template t at cell y,x) with box g, rounded to 14 decimals: dense_overlap.py:32-75
__device__ __forceinline__ double iou14(double x1, double y1, double x2, double y2, double farea,
                                        double gx1, double gy1, double gx2, double gy2) {
  double bw = gx2 - gx1 + 1.0, bh = gy2 - gy1 + 1.0;
  double barea = bw * bh;
  double xx1 = fmax(x1, gx1), yy1 = fmax(y1, gy1), xx2 = fmin(x2, gx2), yy2 = fmin(y2, gy2);
  double iw = xx2 - xx1 + 1.0, ih = yy2 - yy1 + 1.0;
  double ov = 0.0;
  if (ih > 0.0 && iw > 0.0) {
    double ia = iw * ih;
    double un = farea + barea - ia;
    ov = ia / un;
  }
  return rint(ov * 1e14) / 1e14;   // np.around(x, 14) == rint(x * 1e14) / 1e14
}

__device__ __forceinline__ double noise_at(const TgtParams& p, int b, int flat, int g, int G) {
  if (p.noise) return p.noise[p.noise_off[b] + (int64_t)flat * G + g];
  return tf::u01(tf::hash4(p.seed, (uint64_t)b, (uint64_t)flat, (uint64_t)g));
}

__device__ __forceinline__ bool pad_at(const TgtParams& p, int b, int y, int x, double dx1, double dy1,
                                       double dx2, double dy2) {
  if (!p.paste) return false;
  int xs = (p.flips && p.flips[b]) ? (p.vsx - 1 - x) : x;   // fliplr of the mask (wider_face.py:165)
  double cx = (double)(p.ofx + xs * p.stx), cy = (double)(p.ofy + y * p.sty);
  const int32_t* pb = p.paste + 4 * b;
  return (cx + dx1 < (double)(pb[0] + 1)) | (cy + dy1 < (double)(pb[1] + 1)) |
         (cx + dx2 > (double)pb[2]) | (cy + dy2 > (double)pb[3]);      // processor.py:143-148
}

template <int PHASE>
__global__ void __launch_bounds__(256) targets_kernel(TgtParams p) {
  const int b = blockIdx.y;
  const int cells = p.vsy * p.vsx;
  const int idx = blockIdx.x * 256 + threadIdx.x;        // (t, y, x), x fastest
  const bool active = idx < p.nt * cells;
  const int t = active ? idx / cells : 0;
  const int cell = active ? idx - t * cells : 0;
  const int y = cell / p.vsx, x = cell - y * p.vsx;
  const int flat = cell * p.nt + t;                      // C-order over (y, x, t): reference index
  const int g0 = p.box_off[b], G = p.box_off[b + 1] - g0;
  const double* tp = p.tpl + (size_t)t * p.tstride;
  const double dx1 = tp[0], dy1 = tp[1], dx2 = tp[2], dy2 = tp[3];
  const double cx = (double)p.ofx + (double)x * ((double)p.stx / 1.0);   // dense_overlap.py:50
  const double cy = (double)p.ofy + (double)y * ((double)p.sty / 1.0);
  const double ax1 = dx1 + cx, ay1 = dy1 + cy, ax2 = dx2 + cx, ay2 = dy2 + cy;
  const double fh = dy2 - dy1 + 1.0, fw = dx2 - dx1 + 1.0;
  const double farea = fw * fh;

  double best = -1.0; int bestg = 0;
  for (int g = 0; g < G; ++g) {
    const double* bx = p.boxes + 4 * (size_t)(g0 + g);
    const double gx1 = bx[0], gy1 = bx[1], gx2 = bx[2], gy2 = bx[3];
    double v = -1.0;
    if (active) {
      double r = iou14(ax1, ay1, ax2, ay2, farea, gx1, gy1, gx2, gy2);
      v = r + 1e-6 * noise_at(p, b, flat, g, G);                      // processor.py:195
      if (v > best) { best = v; bestg = g; }                          // first arg-max (:197)
    }
    const bool hot = v > p.neg_thresh;                                 // only these can win (:255)
    if (__ballot(hot)) {
      unsigned long long bits = hot ? (unsigned long long)__double_as_longlong(v) : 0ull;
      if (PHASE == 0) {
        unsigned long long m = bits;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          unsigned long long other = __shfl_xor(m, o, 64);
          m = other > m ? other : m;
        }
        if (tf::lane_id() == 0) atomicMax(&p.gmax[g0 + g], m);
      } else {
        if (hot && bits == p.gmax[g0 + g]) atomicMin(&p.gidx[g0 + g], (unsigned int)flat);
      }
    }
  }
  if (PHASE != 0 || !active) return;

  float cls = -1.f, tx = 0.f, ty = 0.f, tw = 0.f, th = 0.f;
  if (G > 0) {
    const double* bx = p.boxes + 4 * (size_t)(g0 + bestg);
    const double gx1 = bx[0], gy1 = bx[1], gx2 = bx[2], gy2 = bx[3];
    const double fcx = (gx1 + gx2) / 2.0, fcy = (gy1 + gy2) / 2.0;     // processor.py:181-182
    const double coarse_x = (double)(p.ofx + x * p.stx), coarse_y = (double)(p.ofy + y * p.sty);
    tx = (float)((fcx - coarse_x) / fw);                               // :184-185
    ty = (float)((fcy - coarse_y) / fh);
    tw = (float)log((gx2 - gx1 + 1.0) / fw);                           // :187-191
    th = (float)log((gy2 - gy1 + 1.0) / fh);
    if (best >= p.pos_thresh) cls = 1.f;                               // :260-261
    else if (p.neg_thresh <= best && best < p.pos_thresh) cls = 0.f;   // :264-269
  }
  if (cls != -1.f && pad_at(p, b, y, x, dx1, dy1, dx2, dy2)) { cls = 0.f; tx = 0.f; }   // :272-274 (tx only: D5)
  const size_t plane = (size_t)cells;
  p.cls[((size_t)b * p.nt + t) * plane + cell] = cls;
  float* r = p.reg + ((size_t)b * 4 * p.nt + t) * plane + cell;
  r[0] = tx; r[(size_t)p.nt * plane] = ty; r[(size_t)2 * p.nt * plane] = tw; r[(size_t)3 * p.nt * plane] = th;
}

__global__ void targets_best_anchor_kernel(TgtParams p, int total) {
  int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total) return;
  if (__longlong_as_double((long long)p.gmax[gi]) > p.neg_thresh) {     // :255
    // which image owns box gi
    int b = 0;
    while (p.box_off[b + 1] <= gi) ++b;
    unsigned int flat = p.gidx[gi];
    int t = flat % p.nt, cell = flat / p.nt;
    int y = cell / p.vsx, x = cell - y * p.vsx;
    const double* tp = p.tpl + (size_t)t * p.tstride;
    if (!pad_at(p, b, y, x, tp[0], tp[1], tp[2], tp[3]))
      p.cls[((size_t)b * p.nt + t) * (size_t)(p.vsy * p.vsx) + cell] = 1.f;   // :257 then :272-273
  }
}

__global__ void iou_dump_kernel(const double* boxes, int G, const double* tpl, int nt, int tstride, int vsy, int vsx,
                                int ofy, int ofx, int sty, int stx, double* out) {
  int flat = blockIdx.x * blockDim.x + threadIdx.x;
  if (flat >= vsy * vsx * nt) return;
  int t = flat % nt, cell = flat / nt, y = cell / vsx, x = cell % vsx;
  const double* tp = tpl + (size_t)t * tstride;
  double cx = (double)ofx + (double)x * ((double)stx / 1.0), cy = (double)ofy + (double)y * ((double)sty / 1.0);
  double fh = tp[3] - tp[1] + 1.0, fw = tp[2] - tp[0] + 1.0;
  for (int g = 0; g < G; ++g)
    out[(size_t)flat * G + g] = iou14(tp[0] + cx, tp[1] + cy, tp[2] + cx, tp[3] + cy, fw * fh,
                                      boxes[4 * g], boxes[4 * g + 1], boxes[4 * g + 2], boxes[4 * g + 3]);
}

}  // namespace

extern "C" size_t tf_targets_workspace_bytes(int total_boxes) {
  size_t n = (size_t)(total_boxes > 0 ? total_boxes : 1);
  return n * 8 + n * 4 + 64;
}

extern "C" int tf_dense_overlap_targets(const double* boxes, const int32_t* box_offsets, int B,
                                        const double* templates, int nt, int tstride,
                                        int vsy, int vsx, int ofy, int ofx, int sty, int stx,
                                        const int32_t* paste_boxes, const int32_t* flips,
                                        const double* noise, const int64_t* noise_offsets, uint64_t seed,
                                        double pos_thresh, double neg_thresh,
                                        float* class_map, float* reg_map,
                                        void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || nt <= 0 || vsy <= 0 || vsx <= 0 || !box_offsets || !templates || !class_map || !reg_map) return TF_ERR_ARG;
  if (noise && !noise_offsets) return TF_ERR_ARG;
  // total boxes is only known on device; the caller sizes ws from its host copy.
  size_t nslots = (ws_bytes - 64) / 12;
  if (!ws || ws_bytes < 76) return TF_ERR_WORKSPACE;
  TgtParams p;
  p.boxes = boxes; p.box_off = box_offsets; p.tpl = templates; p.nt = nt; p.tstride = tstride;
  p.vsy = vsy; p.vsx = vsx; p.ofy = ofy; p.ofx = ofx; p.sty = sty; p.stx = stx;
  p.paste = paste_boxes; p.flips = flips; p.noise = noise; p.noise_off = noise_offsets; p.seed = seed;
  p.pos_thresh = pos_thresh; p.neg_thresh = neg_thresh; p.cls = class_map; p.reg = reg_map;
  p.gmax = (unsigned long long*)ws;
  p.gidx = (unsigned int*)((char*)ws + nslots * 8);
  if (hipMemsetAsync(p.gmax, 0, nslots * 8, stream) != hipSuccess) return TF_ERR_LAUNCH;
  if (hipMemsetAsync(p.gidx, 0xff, nslots * 4, stream) != hipSuccess) return TF_ERR_LAUNCH;
  dim3 grid((nt * vsy * vsx + 255) / 256, B);
  hipLaunchKernelGGL(targets_kernel<0>, grid, dim3(256), 0, stream, p);
  hipLaunchKernelGGL(targets_kernel<1>, grid, dim3(256), 0, stream, p);
  int total = (int)nslots;
  // total real boxes <= nslots; surplus slots keep gmax == 0 and are skipped by the threshold test
  hipLaunchKernelGGL(targets_best_anchor_kernel, dim3((total + 127) / 128), dim3(128), 0, stream, p, total);
  TF_CHECK_LAUNCH();
  return TF_OK;
}

extern "C" int tf_dense_overlap_iou(const double* boxes, int G, const double* templates, int nt, int tstride,
                                    int vsy, int vsx, int ofy, int ofx, int sty, int stx,
                                    double* iou_out, void* stream_) {
  if (G <= 0) return TF_OK;
  int n = vsy * vsx * nt;
  hipLaunchKernelGGL(iou_dump_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, boxes, G, templates, nt,
                     tstride, vsy, vsx, ofy, ofx, sty, stx, iou_out);
  TF_CHECK_LAUNCH();
  return TF_OK;
}

// ---- template clustering: the n x n distance matrix 1 - IoU of the centred ground-truth shapes
// (tinyfaces/clustering/cluster.py:28-37 over tinyfaces/metrics.py:8-40: plain areas, no +1, IoU = 0 when the union is not
// positive).  float64, every operation rounded like numpy's (this file is compiled with -ffp-contract=off): bit-exact with the
// reference's double loop, which is what makes the k-medoids that follows index-exact.
namespace {
__global__ void __launch_bounds__(256) pairwise_dist_kernel(const double* __restrict__ boxes, int n, double* __restrict__ out) {
  __shared__ double4 col[256];
  const int j0 = blockIdx.x * 256, i0 = blockIdx.y * 16;
  const int j = j0 + threadIdx.x;
  if (j < n) col[threadIdx.x] = *reinterpret_cast<const double4*>(boxes + 4 * (size_t)j);
  __syncthreads();
  if (j >= n) return;
  const double4 b = col[threadIdx.x];
  const double area_b = (b.z - b.x) * (b.w - b.y);
  for (int i = i0; i < min(n, i0 + 16); ++i) {
    const double4 a = *reinterpret_cast<const double4*>(boxes + 4 * (size_t)i);       // wave-uniform: one scalar-path load
    const double area_a = (a.z - a.x) * (a.w - a.y);
    const double xa = fmax(a.x, b.x), ya = fmax(a.y, b.y), xb = fmin(a.z, b.z), yb = fmin(a.w, b.w);
    const double inter = (xb - xa) * (yb - ya);
    const double uni = area_a + area_b - inter;
    out[(size_t)i * n + j] = 1.0 - (uni <= 0.0 ? 0.0 : inter / uni);
  }
}
}  // namespace

extern "C" int tf_pairwise_iou_distance(const double* boxes, int n, double* out, void* stream) {
  if (n < 0 || (n > 0 && (!boxes || !out))) return TF_ERR_ARG;
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(pairwise_dist_kernel, dim3((n + 255) / 256, (n + 15) / 16), dim3(256), 0, (hipStream_t)stream, boxes, n, out);
  TF_CHECK_LAUNCH();
  return TF_OK;
}

