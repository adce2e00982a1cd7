// Micro-benchmark: per-CU staging rate from an L2-resident buffer into LDS, 512-thread workgroups (one per CU):
//   mode 0: LDS-DMA, 16 B per lane (global_load_lds_dwordx4), lane-contiguous 1 KiB pieces
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: LDS-DMA 4 B per lane (256 B pieces)
//   mode 3: global_load_dwordx4 -> VGPR only (no LDS write)
// Each wave moves PIECES pieces per round (issue all, wait, barrier), ROUNDS rounds.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PIECES = 8;
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t span, int rounds, float* out) {
  __shared__ __attribute__((aligned(1024))) char lds[8 * PIECES * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc = {0, 0, 0, 0};
  size_t off = ((size_t)blockIdx.x * 7919 * 1024) % span;
  for (int r = 0; r < rounds; ++r) {
    f32x4 v[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const size_t o = (off + (size_t)(wave * PIECES + i) * 1024) % span;
      char* dst = lds + (wave * PIECES + i) * 1024;
      if constexpr (MODE == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + lane * 16), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      else if constexpr (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + q * 256 + lane * 4), (__attribute__((address_space(3))) void*)(dst + q * 256), 4, 0, 0);
      } else
        v[i] = *reinterpret_cast<const f32x4*>(src + o + lane * 16);
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) *reinterpret_cast<f32x4*>(lds + (wave * PIECES + i) * 1024 + lane * 16) = v[i];
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) acc += v[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (MODE != 3) acc += *reinterpret_cast<f32x4*>(lds + ((tid * 16 + r * 1024) % (8 * PIECES * 1024)));
    off = (off + 8 * PIECES * 1024) % span;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
int main(int argc, char** argv) {
  const size_t span = (argc > 1 ? atoi(argv[1]) : 2) * (size_t)1 << 20;  // MB, L2-resident when small
  const int rounds = 2000, blocks = 256;
  char* src; float* out;
  hipMalloc(&src, span + (1 << 20)); hipMemset(src, 1, span + (1 << 20)); hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, src, span, rounds, out);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, src, span, rounds, out);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, src, span, rounds, out);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, src, span, rounds, out);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)blocks * rounds * 8 * PIECES * 1024;
    printf("span %zu MB mode %d: %.3f ms  %.2f TB/s  %.1f B/clk/CU at 2.4 GHz\n", span >> 20, mode, ms, bytes / ms / 1e9, bytes / blocks / (ms * 1e-3 * 2.4e9));
  }
  return 0;
}
