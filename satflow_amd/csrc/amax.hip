// sf_amax: max |t| of an fp32-stored NHWC tensor into a device word (sfTensor::amax) - the per-tensor power-of-two scale of the SF_F32E compute mode's
// GRADIENT operands (include/satflow_hip.h; conv3x3_f32e.hip / conv3x3_wgrad_f32e.hip read the word and derive 2^(14 - floor(log2 amax)) themselves).
// HBM-bound: one pass over the tensor, 16-byte loads, wave maximum by DPP-free shuffles, one atomic per workgroup on the float's bit pattern (non-negative
// floats order like unsigned integers; NaN / inf patterns are the largest and therefore survive: the consumer's output is then NaN, loudly).
#include "sf_common.h"

namespace {

__global__ void amax_reset_kernel(float* amax, float* acc, int reset_acc) {
  amax[0] = 0.f;
  if (acc && reset_acc) acc[0] = 0.f;
}

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ t, long long pixels, int c, int stride, unsigned* __restrict__ amax,
                                                   unsigned* __restrict__ acc) {
  const int quads = c / 4;
  const long long total = pixels * quads;
  unsigned m = 0u;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long px = e / quads;
    const int q = (int)(e - px * quads);
    // (explicit components of an integer vector: with `__builtin_bit_cast(unsigned, v[k])` on a float vector in an unrolled loop hipcc 7.2 kept
    // component 0 only - found by tests/test_f32e_gpu.py::test_amax_word)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *reinterpret_cast<const u32x4*>(t + px * stride + 4 * q);
    const unsigned a0 = v.x & 0x7fffffffu, a1 = v.y & 0x7fffffffu, a2 = v.z & 0x7fffffffu, a3 = v.w & 0x7fffffffu;
    const unsigned m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3;
    const unsigned mq = m01 > m23 ? m01 : m23;
    m = mq > m ? mq : m;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off);
    m = o > m ? o : m;
  }
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = red[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) m = red[k] > m ? red[k] : m;
    // (the words only grow: a workgroup whose maximum is already covered skips its atomics)
    if (m > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, m);
    if (acc && m > __atomic_load_n(acc, __ATOMIC_RELAXED)) atomicMax(acc, m);
  }
}

}  // namespace

extern "C" int sf_amax(sfTensor t, int64_t pixels, float* amax, float* amax_acc, int32_t reset_acc, sfStream stream) {
  SF_REQUIRE(amax && ((uintptr_t)amax & 3) == 0 && ((uintptr_t)amax_acc & 3) == 0, "sf_amax: amax null / misaligned");
  SF_REQUIRE(pixels >= 0 && t.c >= 0 && t.c % 4 == 0, "sf_amax: pixels=%lld channels=%d (a multiple of 4)", (long long)pixels, t.c);
  SF_REQUIRE(pixels == 0 || t.c == 0 || (t.ptr && t.dtype == SF_F32 && ((uintptr_t)t.ptr & 15) == 0 && t.stride % 4 == 0 && t.stride >= t.c),
             "sf_amax: an fp32-stored tensor with 16-byte aligned pixels (stride %d, channels %d)", t.stride, t.c);
  hipLaunchKernelGGL(amax_reset_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax, amax_acc, (int)reset_acc);
  SF_CHECK_LAUNCH("amax_reset");
  const long long total = (long long)pixels * (t.c / 4);
  if (total == 0) return 0;
  const long long want = (total + 255) / 256;
  const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)t.ptr, (long long)pixels, (int)t.c, (int)t.stride,
                     (unsigned*)amax, (unsigned*)amax_acc);
  SF_CHECK_LAUNCH("amax");
  return 0;
}
