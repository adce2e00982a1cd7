// Fused Adam on flat fp32 buffers (all parameters of a model live in ONE buffer, all gradients in
// another, so the optimizer is one HBM-bound launch and the DDP all-reduce is one collective).
// Arithmetic of torch.optim.Adam (no amsgrad, no weight decay) as configured by the reference,
// satflow/models/conv_lstm.py:48-51 and pl_metnet.py:70.
#include "sf_common.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = gg[j] * gscale;
      mm[j] = b1 * mm[j] + (1.f - b1) * gj;
      vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
      pp[j] -= (lr / bc1) * mm[j] / (sqrtf(vv[j]) / bc2_sqrt + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float gj = g[i] * gscale;
    const float mj = b1 * m[i] + (1.f - b1) * gj, vj = b2 * v[i] + (1.f - b2) * gj * gj;
    m[i] = mj; v[i] = vj;
    p[i] -= (lr / bc1) * mj / (sqrtf(vj) / bc2_sqrt + eps);
  }
}
}  // namespace

extern "C" int sf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, int32_t step, float grad_scale, sfStream stream) {
  SF_REQUIRE(step >= 1, "adam: step must start at 1");
  SF_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "adam: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  const long long n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, beta1, beta2,
                     eps, bc1, bc2_sqrt, grad_scale);
  SF_CHECK_LAUNCH("adam");
  return 0;
}
