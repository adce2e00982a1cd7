// Loss kernels of the CloudGAN steps (reference satflow/models/cloudgan.py:121-189, gan/discriminators.py:70-136):
//   * GANLoss("vanilla") = nn.BCEWithLogitsLoss against a constant real / fake label, on the discriminator's patch logits;
//   * the L1 term (nowcasting_utils get_loss("l1") = nn.L1Loss) between generated and future frames;
// each with its gradient and the per-timestep means in ONE pass (the reference evaluates them timestep by timestep in a
// Python loop, cloudgan.py:137-150).  Tensors are NHWC rows [rows][stride] of which the first `c` lanes are real; rows are
// split into `groups` equal, contiguous groups (timesteps of the time-major layout).
#include "sf_common.h"

namespace {

// MODE 0: L1 against `target` (same layout);  MODE 1: BCE-with-logits against a constant label (`label` for even groups,
// `label_odd` for odd ones: the discriminator step interleaves real and generated frames, cloudgan.py:163-172);
// MODE 2: GANLoss("lsgan") = nn.MSELoss against the constant label;  MODE 3: GANLoss("wgangp") = -mean(x) for a real target
// (label > 0.5), +mean(x) for a generated one (gan/discriminators.py:94-136)
template <int MODE>
__global__ __launch_bounds__(256) void pair_loss_kernel(const float* __restrict__ pred, int ps, const float* __restrict__ target, int ts, float label, float label_odd,
                                                        long long rows_per_group, int c, float gscale, float* __restrict__ grad, int gs,
                                                        int grad_lanes, double* __restrict__ sums) {
  __shared__ double red[4];
  const int g = blockIdx.y;
  const long long total = rows_per_group * c;
  double acc = 0.0;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long row = (long long)g * rows_per_group + e / c;
    const int lane = (int)(e % c);
    const float x = pred[row * ps + lane];
    float l, d;
    if (MODE == 0) {
      const float df = x - target[row * ts + lane];
      l = fabsf(df);
      d = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
    } else if (MODE == 1) {
      // max(x, 0) - x * t + log(1 + exp(-|x|));  d/dx = sigmoid(x) - t
      const float t = (g & 1) ? label_odd : label;
      l = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      d = 1.f / (1.f + expf(-x)) - t;
    } else if (MODE == 2) {
      const float df = x - ((g & 1) ? label_odd : label);
      l = df * df;
      d = 2.f * df;
    } else {
      const float sgn = (((g & 1) ? label_odd : label) > 0.5f) ? -1.f : 1.f;
      l = sgn * x;
      d = sgn;
    }
    acc += (double)l;
    if (grad) grad[row * gs + lane] = d * gscale;
  }
  // pad lanes of the gradient are zero
  if (grad && grad_lanes > c) {
    const long long padtotal = rows_per_group * (grad_lanes - c);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < padtotal; e += (long long)gridDim.x * blockDim.x) {
      const long long row = (long long)g * rows_per_group + e / (grad_lanes - c);
      grad[row * gs + c + (int)(e % (grad_lanes - c))] = 0.f;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = red[0] + red[1] + red[2] + red[3];
    atomicAdd(sums + 1 + g, t);
    atomicAdd(sums, t);
  }
}

__global__ void pair_loss_finalize_kernel(const double* __restrict__ sums, double count_per_group, int groups, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i == 0) out[0] = (float)(sums[0] / (count_per_group * groups));
  if (i < groups) out[1 + i] = (float)(sums[1 + i] / count_per_group);
}

int launch(int mode, sfTensor pred, sfTensor target, float label, float label_odd, int64_t rows, int32_t groups, int32_t c, sfTensor grad, double* sums, float* out,
           hipStream_t st) {
  SF_REQUIRE(pred.ptr && pred.dtype == SF_F32 && sums && out && groups >= 1 && groups <= 1024 && rows % groups == 0 && c >= 1 && c <= pred.stride,
             "gan loss: pred fp32, rows %% groups == 0, c <= stride");
  SF_REQUIRE(mode >= 0 && mode <= 3, "gan loss: mode %d (0 l1, 1 vanilla, 2 lsgan, 3 wgangp)", mode);
  SF_REQUIRE(mode != 0 || (target.ptr && target.dtype == SF_F32 && c <= target.stride), "l1 loss: target");
  SF_REQUIRE(!grad.ptr || (grad.dtype == SF_F32 && c <= grad.stride && grad.c <= grad.stride), "gan loss: grad layout");
  if (sf_fill_async(sums, 0, sizeof(double) * (1 + groups), st) != hipSuccess) { sf_set_error("gan loss: memset failed"); return 2; }
  if (rows == 0) {  // an empty batch: torch's mean over nothing is NaN; never leave `out` uninitialised
    if (sf_fill_async(out, 0xff, sizeof(float) * (1 + groups), st) != hipSuccess) { sf_set_error("gan loss: memset failed"); return 2; }
    return 0;
  }
  const long long rpg = rows / groups, total = rpg * c;
  const double count = (double)total;
  const float gscale = (float)(1.0 / (count * groups));  // d(mean over ALL elements)
  int bx = (int)((total + 255) / 256); if (bx > 512) bx = 512; if (bx < 1) bx = 1;
  dim3 grid(bx, groups), block(256);
#define SF_PAIR(M_) hipLaunchKernelGGL((pair_loss_kernel<M_>), grid, block, 0, st, (const float*)pred.ptr, pred.stride, (const float*)nullptr, 0, label, label_odd, rpg, c, \
                                      gscale, (float*)grad.ptr, grad.stride, grad.c, sums)
  if (mode == 0)
    hipLaunchKernelGGL((pair_loss_kernel<0>), grid, block, 0, st, (const float*)pred.ptr, pred.stride, (const float*)target.ptr, target.stride, 0.f, 0.f, rpg, c,
                       gscale, (float*)grad.ptr, grad.stride, grad.c, sums);
  else if (mode == 1) SF_PAIR(1);
  else if (mode == 2) SF_PAIR(2);
  else SF_PAIR(3);
#undef SF_PAIR
  SF_CHECK_LAUNCH("gan loss");
  hipLaunchKernelGGL(pair_loss_finalize_kernel, dim3(1), dim3(1024), 0, st, sums, count, groups, out);
  SF_CHECK_LAUNCH("gan loss finalize");
  return 0;
}

}  // namespace

extern "C" {

int sf_l1_loss(sfTensor pred, sfTensor target, int64_t rows, int32_t groups, int32_t c, sfTensor grad, double* sums, float* out, sfStream stream) {
  return launch(0, pred, target, 0.f, 0.f, rows, groups, c, grad, sums, out, (hipStream_t)stream);
}

int sf_bce_logits_loss(sfTensor logits, float label, float label_odd, int64_t rows, int32_t groups, int32_t c, sfTensor grad, double* sums, float* out, sfStream stream) {
  sfTensor none{};
  return launch(1, logits, none, label, label_odd, rows, groups, c, grad, sums, out, (hipStream_t)stream);
}

int sf_gan_loss(int32_t mode, sfTensor pred, float label, float label_odd, int64_t rows, int32_t groups, int32_t c, sfTensor grad, double* sums, float* out,
                sfStream stream) {
  sfTensor none{};
  SF_REQUIRE(mode >= 1 && mode <= 3, "sf_gan_loss: mode %d (1 vanilla, 2 lsgan, 3 wgangp)", mode);
  return launch(mode, pred, none, label, label_odd, rows, groups, c, grad, sums, out, (hipStream_t)stream);
}

}  // extern "C"
