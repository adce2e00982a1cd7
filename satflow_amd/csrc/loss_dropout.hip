// Training-step glue that the reference leaves to ATen and that sits on the hot path:
//   * MSE loss with its gradient and the per-forecast-frame losses in ONE pass (reference: nn.MSELoss +
//     a loop of `forecast_steps` extra MSE evaluations with .item() host syncs, satflow/models/conv_lstm.py:63-69,
//     pl_metnet.py:118-124);
//   * dropout on the encoder output (nn.Dropout(temporal_dropout), pl_metnet.py:58 -> metnet.MetNet) and the
//     sequence-consistent "RNN" dropout of the ConvGRU input, fused into one pass; masks are a counter-based hash of
//     (seed, element index), so the backward regenerates them instead of storing them.
#include "sf_common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// loss_sums[0] = sum (p-y)^2 ; loss_sums[1 + f] = per-frame sums; grad = 2 (p - y) / n
// A workgroup owns one piece (1/pieces) of one `inner`-sized frame slab, so its partial sum belongs to one frame.  In the
// 16-byte path the loads of four 1024-element strides are issued together from clamped offsets and selected afterwards
// (a load under a per-element condition is emitted as load -> vmcnt(0) -> store, one element at a time).
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, long long n,
                                                  long long inner, int frames, float gscale, float* __restrict__ grad,
                                                  double* __restrict__ sums, int pieces, long long piece_len) {
  __shared__ double red[4];
  const bool vec = (inner & 3) == 0;  // 16-byte accesses need every slab base 4-aligned
  const long long s = blockIdx.x / pieces;
  const int piece = blockIdx.x % pieces;
  const long long base = s * inner;
  const long long lo = piece * piece_len, hi = lo + piece_len < inner ? lo + piece_len : inner;
  double acc = 0.0;
  if (vec) {
    for (long long i0 = lo + threadIdx.x * 4; i0 < hi; i0 += 4096) {
      f32x4 a[4], b[4]; bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long i = i0 + u * 1024;
        ok[u] = i < hi;  // hi and i are multiples of 4
        const long long ic = ok[u] ? i : lo;
        a[u] = ld4(pred + base + ic); b[u] = ld4(target + base + ic);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (ok[u]) {
          const f32x4 d = a[u] - b[u];
          if (grad) st4(grad + base + i0 + u * 1024, d * gscale);
          acc += (double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
        }
      }
    }
  } else {
    for (long long j = lo + threadIdx.x; j < hi; j += 256) {
      const float d = pred[base + j] - target[base + j];
      if (grad) grad[base + j] = d * gscale;
      acc += (double)(d * d);
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = red[0] + red[1] + red[2] + red[3];
    atomicAdd(sums + 1 + (s % frames), t);
    atomicAdd(sums, t);
  }
}

__global__ void mse_finalize_kernel(const double* __restrict__ sums, long long n, int frames, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i == 0) out[0] = (float)(sums[0] / (double)n);
  if (i < frames) out[1 + i] = (float)(sums[1 + i] / ((double)n / frames));
}

// y = x * m1(idx) * m2(idx % period); either probability may be 0 (mask == 1).  Masks: sf_drop_scales (sf_common.h), the same
// function the fused max-pool kernels use.
__global__ __launch_bounds__(256) void dropout2_kernel(const float* __restrict__ x, long long n, const sfDrop d, long long period,
                                                       float* __restrict__ y) {
  const long long n4 = n >> 2, per4 = period >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 sc = sf_drop_scales(d, (unsigned long long)i, (unsigned long long)(d.p2 > 0.f ? i % per4 : 0));
    st4(y + i * 4, ld4(x + i * 4) * sc);
  }
}

// the same masks on a bf16-stored tensor, 8 elements (two mask groups) per thread, in place or not: the pooled encoder output behind
// sf_conv3x3_fwd_folded_pool (the pooling epilogue does not hash: ~100 cycles per mask group on a wave that is alone on its SIMD)
__global__ __launch_bounds__(256) void dropout2_bf16_kernel(const __bf16* __restrict__ x, long long n, const sfDrop d, long long period, __bf16* __restrict__ y) {
  const long long n8 = n >> 3, per4 = period >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long g = 2ull * (unsigned long long)i;
    const f32x4 a = sf_drop_scales(d, g, d.p2 > 0.f ? g % per4 : 0), b = sf_drop_scales(d, g + 1, d.p2 > 0.f ? (g + 1) % per4 : 0);
    f32x8_t v = ldv8(x + i * 8);
    v = v * f32x8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    stv8(y + i * 8, v);
  }
}

}  // namespace

extern "C" {

int sf_mse_loss(const float* pred, const float* target, int64_t n, int64_t inner, int32_t frames, float* grad, double* sums,
                float* out, sfStream stream) {
  SF_REQUIRE(n > 0 && inner > 0 && n % inner == 0 && frames >= 1 && frames <= 1023 && (n / inner) % frames == 0,
             "mse: n=%lld inner=%lld frames=%d", (long long)n, (long long)inner, frames);
  SF_REQUIRE(((((uintptr_t)pred) | ((uintptr_t)target) | ((uintptr_t)grad)) & 15) == 0, "mse: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  SF_REQUIRE(sf_fill_async(sums, 0, sizeof(double) * (1 + frames), st) == hipSuccess, "mse: memset");
  const long long slabs = n / inner;
  int pieces = (int)(inner / 4096 < 16 ? inner / 4096 : 16);
  if (pieces < 1) pieces = 1;
  long long piece_len = ((inner + pieces - 1) / pieces + 1023) / 1024 * 1024;  // multiple of 1024: pieces stay 16-byte aligned
  SF_REQUIRE(slabs * pieces < (1ll << 31), "mse: too many slabs (%lld)", (long long)slabs);
  hipLaunchKernelGGL(mse_kernel, dim3((unsigned)(slabs * pieces)), dim3(256), 0, st, pred, target, (long long)n, (long long)inner, frames,
                     2.0f / (float)n, grad, sums, pieces, piece_len);
  SF_CHECK_LAUNCH("mse");
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(1024), 0, st, sums, (long long)n, frames, out);
  SF_CHECK_LAUNCH("mse_finalize");
  return 0;
}

int sf_dropout2(const float* x, int64_t n, float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2, float* y,
                sfStream stream) {
  SF_REQUIRE(n % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f && period > 0 &&
                 period % 4 == 0, "dropout2: n=%lld p1=%f p2=%f period=%lld (n and period must be multiples of 4)", (long long)n, p1, p2, (long long)period);
  if (n == 0) return 0;
  const long long n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
  hipLaunchKernelGGL(dropout2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, sf_make_drop(p1, p2, seed1, seed2),
                     (long long)period, y);
  SF_CHECK_LAUNCH("dropout2");
  return 0;
}

int sf_dropout2_bf16(const void* x, int64_t n, float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2, void* y, sfStream stream) {
  SF_REQUIRE(n % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f && period > 0 &&
                 period % 4 == 0, "dropout2_bf16: n=%lld p1=%f p2=%f period=%lld (n a multiple of 8, period of 4)", (long long)n, p1, p2, (long long)period);
  if (n == 0) return 0;
  const long long n8 = n >> 3;
  const int blocks = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
  hipLaunchKernelGGL(dropout2_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (long long)n, sf_make_drop(p1, p2, seed1, seed2),
                     (long long)period, (__bf16*)y);
  SF_CHECK_LAUNCH("dropout2_bf16");
  return 0;
}

}  // extern "C"
