// Micro-benchmark: how fast does a burst of output stores leave a CU, and is the limit per CU or chip-wide?
// A 512-thread workgroup per CU alternates a compute phase (dependent-free bf16 MFMAs, `mfmas` per wave) with a store burst of 16 x 16-byte stores per lane
// (128 KiB per workgroup, distinct lines, like the 3x3 convolution's transposed epilogue), `rounds` times.
//   argv[1] = workgroups launched (256 = whole chip, 32 = one eighth: if a burst is limited per CU its time does not change)
//   argv[2] = MFMAs per wave per round (0: stores only)
//   argv[3] = mode: 0 plain global stores, 1 nontemporal stores, 2 stores interleaved with the MFMAs (one store per mfmas/16 MFMAs), 3 no stores (control)
//   argv[4] = phase: 1 = workgroup b starts with (b % 8) * mfmas / 8 extra MFMAs (de-phased bursts)
// Build: hipcc --offload-arch=gfx950 -O3 store_drain.hip -o store_drain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(char* __restrict__ dst, size_t per_wg, int rounds, int mfmas, int phase, float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(wave - i); }
  char* base = dst + (size_t)blockIdx.x * per_wg;
  if (phase) for (int m = 0; m < (int)(blockIdx.x % 8) * mfmas / 8; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < rounds; ++r) {
    char* p = base + (size_t)(r & 7) * (128 * 1024) + wave * 16 * 1024 + lane * 16;
    if (MODE == 2) {
      const int per = mfmas / 16 > 0 ? mfmas / 16 : 1;
      for (int s = 0; s < 16; ++s) {
        for (int m = 0; m < per; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        *reinterpret_cast<f32x4*>(p + s * 1024) = f32x4{acc[0], acc[1], acc[2], (float)r};
      }
    } else {
      for (int m = 0; m < mfmas; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const f32x4 v = f32x4{acc[s & 15], acc[1], acc[2], (float)r};
        if (MODE == 3) { if (v[0] == 12345.f) *reinterpret_cast<f32x4*>(p) = v; }
        else if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p + s * 1024));
        else *reinterpret_cast<f32x4*>(p + s * 1024) = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the convolution's next item waits for the burst (vmcnt covers stores)
    }
    __syncthreads();
  }
  if (acc[0] == 12345.678f) out[0] = acc[0];
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, mfmas = argc > 2 ? atoi(argv[2]) : 0, mode = argc > 3 ? atoi(argv[3]) : 0, phase = argc > 4 ? atoi(argv[4]) : 0;
  const int rounds = 400;
  const size_t per_wg = 8 * 128 * 1024;
  char* dst; float* out;
  hipMalloc(&dst, per_wg * blocks); hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, dst, per_wg, rounds, mfmas, phase, out);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, dst, per_wg, rounds, mfmas, phase, out);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, dst, per_wg, rounds, mfmas, phase, out);
    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, dst, per_wg, rounds, mfmas, phase, out);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  const double bytes = (double)blocks * rounds * 128 * 1024;
  printf("wgs %3d mfmas %4d mode %d phase %d: %.3f ms  %.2f us/round  %.2f TB/s written\n", blocks, mfmas, mode, phase, ms, ms * 1e3 / rounds, bytes / ms / 1e9);
  return 0;
}
