// Micro-benchmark / hardware fact check: does vmcnt retire loads and stores IN ORDER with respect to each other on gfx950?
// The counted waits of conv3x3_bf16_persist4.hip, conv3x3_wgrad_bf16_dma.hip and convgru_seq.hip leave younger STORES in flight while they wait for older
// LOADS (`s_waitcnt vmcnt(N)`, N = number of stores issued after the loads).  That is only safe if a store can never be retired from the counter before an
// older load.  (LLVM's waitcnt pass assumes nothing of the kind for mixed event types; the ISA text orders "instructions of a given type".)
//
// Each wave, `iters` times: pre-set a landing register to a sentinel, issue ONE slow load (a random line of a multi-GiB buffer: TLB + HBM miss), then N fast
// 16-byte stores to a tiny, hot, wave-private buffer (L2 hits), then `s_waitcnt vmcnt(N)` and look at the landing register.  If the counter could skip over
// the older load, the sentinel would still be there; counts the violations.  Variants: LDS-DMA load (lands in LDS, checked after the wait) and plain load.
//   argv[1] = GiB of the load buffer (default 8), argv[2] = iterations per wave (default 20000), argv[3] = stores per iteration N (1..16, default 8)
// Build: hipcc --offload-arch=gfx950 -O3 vmcnt_order.hip -o vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ big, size_t big_dwords, char* __restrict__ hot, int iters, unsigned long long* bad, unsigned long long* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  char* myhot = hot + wave * (16 * 1024) + lane * 16;
  unsigned long long violations = 0, acc = 0;
  unsigned long long rng = 0x9E3779B97F4A7C15ull * (wave * 64 + lane + 1);
  for (int i = 0; i < iters; ++i) {
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    const size_t idx = (size_t)((rng >> 20) % (big_dwords / 32)) * 32 + (lane & 31);  // a random 128-byte line per half-wave
    const unsigned* a = big + idx;
    unsigned land = 0xDEADBEEFu;
    const u32x4 v = u32x4{(unsigned)i, (unsigned)lane, 3u, 4u};
    asm volatile("global_load_dword %0, %1, off" : "+v"(land) : "v"(a) : "memory");
#pragma unroll
    for (int s = 0; s < N; ++s) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(myhot + s * 1024), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(%1)\n\tv_mov_b32 %0, %0" : "+v"(land) : "n"(N) : "memory");  // only the N stores may still be out
    if (land == 0xDEADBEEFu) ++violations;  // the buffer is filled with idx ^ 0x5A5A5A5A, never the sentinel
    else if (land != ((unsigned)idx ^ 0x5A5A5A5Au)) violations += 1ull << 32;  // wrong data would be a different kind of surprise
    acc += land;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (violations) atomicAdd(bad, violations);
  if (acc == 0x1234567887654321ull) *sink = acc;
}
__global__ void fill(unsigned* big, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) big[i] = (unsigned)i ^ 0x5A5A5A5Au;
}
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? atoi(argv[1]) : 8;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000, n = argc > 3 ? atoi(argv[3]) : 8;
  const size_t dwords = gib * (1ull << 30) / 4;
  if (dwords > (1ull << 32)) { printf("at most 16 GiB (the check compares 32-bit indices)\n"); return 1; }
  unsigned* big; char* hot; unsigned long long *bad, *sink;
  const int blocks = 1024;
  if (hipMalloc(&big, dwords * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMalloc(&hot, (size_t)blocks * 4 * 16 * 1024); hipMalloc(&bad, 8); hipMalloc(&sink, 8);
  hipMemset(bad, 0, 8);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, big, dwords);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
#define RUN(NN) case NN: hipLaunchKernelGGL(k<NN>, dim3(blocks), dim3(256), 0, 0, big, dwords, hot, iters, bad, sink); break;
  switch (n) { RUN(1) RUN(2) RUN(4) RUN(8) RUN(12) RUN(16) default: printf("N in {1,2,4,8,12,16}\n"); return 1; }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  const double trials = (double)blocks * 4 * iters;
  printf("N=%d stores behind one HBM-missing load, %.0f wave-trials (x64 lanes), %.1f ms: sentinel seen %llu times, wrong data %llu times -> %s\n", n, trials, ms,
         h & 0xffffffffull, h >> 32, h ? "OUT OF ORDER: counted waits across loads and stores are NOT safe" : "loads and stores retire in order");
  return h ? 2 : 0;
}
