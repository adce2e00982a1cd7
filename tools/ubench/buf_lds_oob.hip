// Does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose offset fails the descriptor's range check?
// (The register form returns zeros; the weight-gradient kernel wants the same from the LDS-DMA form so that out-of-image
// halo pieces need no per-lane address select.)  Build: hipcc --offload-arch=gfx950 -O2 buf_lds_oob.hip -o buf_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, int nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds;
  unsigned voff = (l & 1) ? 0x80000000u : (unsigned)(l * 16);  // odd lanes out of range
  unsigned soff = 0;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)"
               :: "v"(voff), "s"(rs), "s"(lds0), "s"(soff) : "memory", "m0");
  __syncthreads();
  for (int i = l; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1024, o);
  std::vector<unsigned> r(256);
  hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
  int ok_even = 0, zero_odd = 0, beef_odd = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      unsigned v = r[l * 4 + j];
      if (l & 1) { zero_odd += v == 0; beef_odd += v == 0xdeadbeefu; } else ok_even += v == (unsigned)(0x1000 + l * 4 + j);
    }
  printf("even lanes correct %d/128; odd (out-of-range) lanes: zero %d/128, untouched %d/128\n", ok_even, zero_odd, beef_odd);
  return 0;
}
