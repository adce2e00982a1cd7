// Test infrastructure only (never loaded by satflow_amd): device-side helpers the GPU tests need and the product has no use for.
//   sftest_occupy_cus: park `workgroups` one-wave workgroups, each holding `lds_bytes` of LDS (>= 80 KiB: at most one per CU, and no
//   room next to it for a kernel that needs more than 160 KiB - lds_bytes), for `microseconds` of wall time on `stream` - "something
//   else holds CUs" (an RCCL kernel on the exchange stream, another process) for tests/test_convgru_seq_gpu.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, unsigned* census) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  lds[threadIdx.x] = (char)threadIdx.x;  // the allocation is real
  __syncthreads();
  if (census && threadIdx.x == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    census[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);  // XCC id | {se, sh, cu, simd, wave} bits
  }
  const unsigned long long t0 = wall_clock64();  // constant 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (lds[threadIdx.x] != (char)threadIdx.x) __builtin_trap();
}
}  // namespace

extern "C" int sftest_occupy_cus(int32_t workgroups, int32_t lds_bytes, int32_t microseconds, void* census, void* stream) {
  if (workgroups <= 0 || lds_bytes < 64 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 2000000) return 1;
  if (hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 2;
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(64), lds_bytes, (hipStream_t)stream, (unsigned long long)microseconds * 100ull,
                     (unsigned*)census);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
