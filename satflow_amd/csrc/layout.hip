// Module-boundary layout conversion: the reference's NCHW-style tensors <-> the time-major NHWC,
// channel-padded activations the kernels use.  HBM-bound; the NCHW side is accessed coalesced
// along x, each thread moves one pixel's channels (16-byte pieces on the NHWC side).
#include "sf_common.h"

namespace {

struct LayoutParams {
  long long sb, st, sc, HW, pixels;
  int nb, C;
};

// TO: storage type of the NHWC tensor (fp32, or bf16 where the consumer is a bf16-operand convolution: the ConvLSTM's input frames in "bf16a" mode)
template <typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, const LayoutParams p,
                                                           TO* __restrict__ dst, int dc, int ds) {
  for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < p.pixels; pix += (long long)gridDim.x * blockDim.x) {
    const long long j = pix / p.HW, yx = pix - j * p.HW;
    const long long t = j / p.nb, b = j - t * p.nb;
    const float* s = src + b * p.sb + t * p.st + yx;
    TO* d = dst + pix * ds;
    for (int c = 0; c < dc; c += 4) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (c + k < p.C) ? s[(long long)(c + k) * p.sc] : 0.f;
      stv4<TO>(d + c, v);
    }
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, int ss, const LayoutParams p,
                                                           float* __restrict__ dst) {
  for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < p.pixels; pix += (long long)gridDim.x * blockDim.x) {
    const long long j = pix / p.HW, yx = pix - j * p.HW;
    const long long t = j / p.nb, b = j - t * p.nb;
    const float* s = src + pix * ss;
    float* d = dst + b * p.sb + t * p.st + yx;
    for (int c = 0; c < p.C; ++c) d[(long long)c * p.sc] = s[c];
  }
}

int grid_for(long long pixels) { return (int)((pixels + 255) / 256 < 4096 ? (pixels + 255) / 256 : 4096); }

}  // namespace

extern "C" {

int sf_nchw_to_nhwc(const float* src, int64_t stride_b, int64_t stride_t, int64_t stride_c, int32_t nb, int32_t nt,
                    int32_t c, int32_t h, int32_t w, sfTensor dst, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_nchw_to_nhwc: dtype %d not built", dtype);
  SF_REQUIRE(dst.dtype == SF_F32 || dst.dtype == SF_BF16, "sf_nchw_to_nhwc: dst storage %d (fp32 or bf16)", dst.dtype);
  SF_REQUIRE(dst.c % 4 == 0 && dst.stride % 4 == 0 && (((uintptr_t)dst.ptr) & (dst.dtype == SF_BF16 ? 7 : 15)) == 0 && dst.c >= c,
             "nchw_to_nhwc: dst channels %d / stride %d", dst.c, dst.stride);
  LayoutParams p{stride_b, stride_t, stride_c, (long long)h * w, (long long)h * w * nb * nt, nb, c};
  if (p.pixels == 0) return 0;
  if (dst.dtype == SF_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(grid_for(p.pixels)), dim3(256), 0, (hipStream_t)stream, src, p, (__bf16*)dst.ptr, dst.c, dst.stride);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(p.pixels)), dim3(256), 0, (hipStream_t)stream, src, p, (float*)dst.ptr, dst.c, dst.stride);
  SF_CHECK_LAUNCH("nchw_to_nhwc");
  return 0;
}

int sf_nhwc_to_nchw(sfTensor src, int32_t nb, int32_t nt, int32_t c, int32_t h, int32_t w, float* dst,
                    int64_t stride_b, int64_t stride_t, int64_t stride_c, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_nhwc_to_nchw: dtype %d not built", dtype);
  SF_F32_ONLY(src, "sf_nhwc_to_nchw");
  SF_REQUIRE(src.c >= c, "nhwc_to_nchw: src channels %d < %d", src.c, c);
  LayoutParams p{stride_b, stride_t, stride_c, (long long)h * w, (long long)h * w * nb * nt, nb, c};
  if (p.pixels == 0) return 0;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(p.pixels)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)src.ptr, src.stride, p, dst);
  SF_CHECK_LAUNCH("nhwc_to_nchw");
  return 0;
}

}  // extern "C"
