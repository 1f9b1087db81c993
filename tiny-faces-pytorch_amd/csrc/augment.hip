// Image preparation in front of the detector: everything between the decoded uint8 image and
// the normalised fp32 NCHW tensor, in one pass per image:
//   PIL BILINEAR resize (x0.5 / x2 in training, any ratio for the evaluation pyramid)   tinyfaces/datasets/wider_face.py:136-146,
//                                                                                        tinyfaces/evaluation.py:46 (through
//                                                                                        torchvision.transforms.functional.resize)
//   500x500 crop pasted at a random place on the mean colour                            tinyfaces/datasets/processor.py:41-76
//   horizontal flip                                                                     tinyfaces/datasets/wider_face.py:155-157
//   ToTensor + Normalize                                                                main.py:44-46
// Only the pixels of the crop window are ever resampled (the reference resizes the whole image on the CPU, then crops).
//
// Bit-exactness: the resize is Pillow's two-pass 8-bit resample (src/libImaging/Resample.c: precompute_coeffs with the bilinear
// filter, normalize_coeffs_8bpc, horizontal pass then vertical pass with a uint8 intermediate).  Every output pixel re-derives
// its own coefficients in double precision exactly as Pillow does (this file is compiled with -ffp-contract=off: the
// `0.5 + w * 2^22` rounding must not become an FMA) and evaluates the horizontal pass for each of its vertical taps, so no
// intermediate image exists.  The float stage divides (x / 255, (x - mean) / std) like torch does.
#include "common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

struct Taps { int lo, n; double center, ss, ww; };

__device__ __forceinline__ double bilinear_w(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

// precompute_coeffs for output index xx of an axis resampled from in_size to out_size (in0 = 0)
__device__ __forceinline__ Taps taps_for(int in_size, int out_size, int xx) {
  Taps t;
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  t.center = (xx + 0.5) * scale;
  t.ss = 1.0 / filterscale;
  int lo = (int)(t.center - support + 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)(t.center + support + 0.5);
  if (hi > in_size) hi = in_size;
  t.lo = lo; t.n = hi - lo;
  double ww = 0.0;
  for (int x = 0; x < t.n; ++x) ww += bilinear_w((x + lo - t.center + 0.5) * t.ss);
  t.ww = ww;
  return t;
}
__device__ __forceinline__ int fixed_coef(const Taps& t, int x) {
  double w = bilinear_w((x + t.lo - t.center + 0.5) * t.ss);
  if (t.ww != 0.0) w /= t.ww;
  return w < 0 ? (int)(-0.5 + w * (double)(1 << kPrecisionBits)) : (int)(0.5 + w * (double)(1 << kPrecisionBits));
}
__device__ __forceinline__ int clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ void __launch_bounds__(256) image_prepare_kernel(const tf_image_prepare_args a) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= a.OW || oy >= a.OH) return;
  const int bx = a.flip ? a.OW - 1 - ox : ox, by = oy;           // position in the un-flipped buffer
  int px[3] = {a.bg[0], a.bg[1], a.bg[2]};
  const int wy = by - a.paste_y, wx = bx - a.paste_x;
  if (wy >= 0 && wy < a.crop_h && wx >= 0 && wx < a.crop_w) {
    const int ry = a.crop_y + wy, rx = a.crop_x + wx;            // pixel of the resized image
    const bool hres = a.RW != a.W, vres = a.RH != a.H;
    Taps th, tv;
    if (hres) th = taps_for(a.W, a.RW, rx);
    if (vres) tv = taps_for(a.H, a.RH, ry);
    auto hpass = [&](int y, int (&o)[3]) {                        // value of the horizontally resampled image at (y, rx)
      const unsigned char* row = a.img + (size_t)y * a.W * 3;
      if (!hres) { o[0] = row[rx * 3]; o[1] = row[rx * 3 + 1]; o[2] = row[rx * 3 + 2]; return; }
      int acc[3] = {1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1)};
      for (int x = 0; x < th.n; ++x) {
        const int k = fixed_coef(th, x);
        const unsigned char* p = row + (size_t)(th.lo + x) * 3;
        acc[0] += p[0] * k; acc[1] += p[1] * k; acc[2] += p[2] * k;
      }
      o[0] = clip8(acc[0]); o[1] = clip8(acc[1]); o[2] = clip8(acc[2]);
    };
    if (!vres) {
      hpass(ry, px);
    } else {
      int acc[3] = {1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1)};
      for (int y = 0; y < tv.n; ++y) {
        int h[3];
        hpass(tv.lo + y, h);
        const int k = fixed_coef(tv, y);
        acc[0] += h[0] * k; acc[1] += h[1] * k; acc[2] += h[2] * k;
      }
      px[0] = clip8(acc[0]); px[1] = clip8(acc[1]); px[2] = clip8(acc[2]);
    }
  }
  const size_t plane = (size_t)a.OH * a.OW, o = (size_t)oy * a.OW + ox;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = (float)px[c] / 255.0f;
    a.out[c * plane + o] = (x - a.mean[c]) / a.std[c];
  }
}

}  // namespace

extern "C" int tf_image_prepare(const tf_image_prepare_args* a, void* stream) {
  if (!a || !a->img || !a->out) return TF_ERR_ARG;
  if (a->H <= 0 || a->W <= 0 || a->RH <= 0 || a->RW <= 0 || a->OH <= 0 || a->OW <= 0) return TF_ERR_ARG;
  if (a->crop_h < 0 || a->crop_w < 0 || a->crop_y < 0 || a->crop_x < 0 || a->crop_y + a->crop_h > a->RH || a->crop_x + a->crop_w > a->RW)
    return TF_ERR_ARG;
  if (a->paste_y < 0 || a->paste_x < 0 || a->paste_y + a->crop_h > a->OH || a->paste_x + a->crop_w > a->OW) return TF_ERR_ARG;
  for (int c = 0; c < 3; ++c) if (!(a->std[c] != 0.f)) return TF_ERR_ARG;
  hipLaunchKernelGGL(image_prepare_kernel, dim3((a->OW + 63) / 64, (a->OH + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  TF_CHECK_LAUNCH();
  return TF_OK;
}
