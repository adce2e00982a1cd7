// What does a bare v_mfma_f32_32x32x16_bf16 stream sustain on this chip, by operand data and by waves per SIMD?
// (Calibrates the "MFMA floor" the convolution / weight-gradient kernels are compared with: DVFS lowers the clock on
// random operands.)  hipcc --offload-arch=gfx950 -O3 mfma_stream.hip -o mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int EXTRA>
__global__ __launch_bounds__(512, 2) void k(const u32x4* __restrict__ ops, float* out, int iters, unsigned long long* cyc) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[3];
  for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, ops[(tid + 64 * i) & 4095]);
  for (int i = 0; i < 3; ++i) b[i] = __builtin_bit_cast(bf16x8, ops[(tid * 7 + 64 * i + 13) & 4095]);
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  int s = blockIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[t % 3], acc[t], 0, 0, 0);
      if constexpr (EXTRA > 0) {
#pragma unroll
        for (int e = 0; e < EXTRA; ++e) asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 7" : "+s"(s));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int t = 0; t < 9; ++t)
    for (int i = 0; i < 16; ++i) r += acc[t][i];
  out[blockIdx.x * blockDim.x + tid] = r + (float)s;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = 4000;
  std::vector<unsigned> h(4096 * 4);
  u32x4* d; float* o; unsigned long long* c;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 8);
  for (int mode = 0; mode < 3; ++mode) {
    // 0: zeros, 1: random bf16 in [-1, 1), 2: random bit patterns with sane exponents
    for (size_t i = 0; i < h.size(); ++i) {
      unsigned lo = 0, hi = 0;
      if (mode >= 1) {
        auto rb = [&]() { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; std::memcpy(&u, &f, 4); return u >> 16; };
        lo = rb(); hi = rb();
      }
      h[i] = lo | (hi << 16);
    }
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int waves = 8; waves >= 4; waves -= 4)
      for (int extra = 0; extra <= 2; ++extra) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
          if (extra == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), 0, 0, d, o, iters, c);
          else if (extra == 1) hipLaunchKernelGGL(k<8>, dim3(256), dim3(waves * 64), 0, 0, d, o, iters, c);
          else hipLaunchKernelGGL(k<24>, dim3(256), dim3(waves * 64), 0, 0, d, o, iters, c);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        const double mf = (double)iters * 36 * waves * 256, fl = mf * 32768.0;
        printf("data %d waves/CU %d salu-pairs/9mfma %2d: %.3f ms  %.0f TF/s  counter %.1f ticks per MFMA per SIMD\n", mode, waves, extra == 0 ? 0 : (extra == 1 ? 8 : 24),
               ms, fl / ms / 1e9, (double)cy / (iters * 36.0 * waves / 4));
      }
  }
  return 0;
}
