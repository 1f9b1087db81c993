// Fused SGD(momentum, weight decay) step over a flat fp32 segment.
// Replaces torch.optim.SGD.step as configured at main.py:67-70 (dampening 0, no nesterov):
//   d = grad*grad_scale + wd*p ;  buf = mu*buf + d  (buf starts at 0, so step 1 gives buf = d) ;  p -= lr*buf
// HBM bound: 3 reads + 2 writes of 4 B per element, float4-vectorised, grid-stride.
#include "common.h"

namespace {
__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  int64_t n, float lr, float mu, float wd, float gs, int vec) {
  const int64_t n4 = vec ? (n >> 2) : 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    mv.x = mu * mv.x + (gv.x * gs + wd * pv.x); pv.x -= lr * mv.x;
    mv.y = mu * mv.y + (gv.y * gs + wd * pv.y); pv.y -= lr * mv.y;
    mv.z = mu * mv.z + (gv.z * gs + wd * pv.z); pv.z -= lr * mv.z;
    mv.w = mu * mv.w + (gv.w * gs + wd * pv.w); pv.w -= lr * mv.w;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(p)[i] = pv;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float mv = mu * m[i] + (g[i] * gs + wd * p[i]);
    m[i] = mv;
    p[i] -= lr * mv;
  }
}
}  // namespace

extern "C" int tf_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n,
                           float lr, float momentum, float weight_decay, float grad_scale, void* stream) {
  if (n < 0 || (n > 0 && (!param || !grad || !momentum_buf))) return TF_ERR_ARG;
  if (n == 0) return TF_OK;
  const int vec = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) == 0;   // float4 path needs 16-byte alignment
  int64_t blocks = ((vec ? (n >> 2) : n) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, n, lr,
                     momentum, weight_decay, grad_scale, vec);
  TF_CHECK_LAUNCH();
  return TF_OK;
}
