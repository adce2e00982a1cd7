// v_smfmac_f32_32x32x32_bf16 (2:4 structured-sparse MFMA, gfx950): (1) operand layout and index encoding by experiment - the ISA text is not on this
// machine -, (2) what a bare stream of it sustains against the dense v_mfma_f32_32x32x16_bf16 on random data.
// Why: the gradient behind a 2x2 max-pooling has exactly one non-zero per window and channel, i.e. at most 2 of any 4 consecutive pixels of an image row:
// conv4's weight gradient (dout as the sparse operand, K = the pixels of a row) qualifies for the sparse instruction with no change of the arithmetic.
//   hipcc --offload-arch=gfx950 -O3 smfmac_probe.hip -o smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one instruction: A slot `slot` of lanes with lane / 32 == half is 1, everything else 0; B[k][j] = k + 1 under the ASSUMED layout "lane (j = lane % 32,
// kb = lane / 32) holds k = 16 kb + e, e = 0..15"; idx = the given word in every lane.  C then shows which logical k the slot multiplied.
__global__ void layout(int half, int slot, unsigned idxw, float* out) {
  const int lane = threadIdx.x;
  bf16x8 a; bf16x16 b;
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)((lane / 32 == half && e == slot) ? 1.0f : 0.0f);
  for (int e = 0; e < 16; ++e) b[e] = (__bf16)(float)(16 * (lane / 32) + e + 1);
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c, (int)idxw, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
// which C element is row i: A row marker.  A slot 0 of lane l = (l % 32) + 1 in half 0 only, idx 0 (slot 0 -> k = 0 assumed), B all ones
__global__ void rows(float* out) {
  const int lane = threadIdx.x;
  bf16x8 a; bf16x16 b;
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)((lane < 32 && e == 0) ? (float)(lane + 1) : 0.0f);
  for (int e = 0; e < 16; ++e) b[e] = (__bf16)1.0f;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c, 0x4444, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

template <bool SPARSE>
__global__ __launch_bounds__(256, 1) void stream(const unsigned* __restrict__ ops, float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4]; bf16x16 b[4];
  for (int i = 0; i < 4; ++i) {
    unsigned w[8];
    for (int j = 0; j < 8; ++j) w[j] = ops[(tid * 8 + j + 512 * i) & 16383];
    std::memcpy(&b[i], w, 32); std::memcpy(&a[i], w, 16);
  }
  f32x16 acc[16];
  for (int t = 0; t < 16; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int idx = 0x4444 ^ (tid & 1 ? 0xeeee ^ 0x4444 : 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if constexpr (SPARSE) acc[t] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t], idx, 0, 0);
      else {
        bf16x8 bb; std::memcpy(&bb, &b[(t >> 2) & 3], 16);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 3], bb, acc[t], 0, 0, 0);
      }
    }
  }
  float r = 0.f;
  for (int t = 0; t < 16; ++t)
    for (int i = 0; i < 16; ++i) r += acc[t][i];
  out[blockIdx.x * blockDim.x + tid] = r;
}

int main() {
  float* o; hipMalloc(&o, 64 * 16 * 4);
  std::vector<float> h(64 * 16);
  printf("== rows: C register / lane -> (row value, expected the A row + 1)\n");
  hipLaunchKernelGGL(rows, dim3(1), dim3(64), 0, 0, o); hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost);
  for (int lane : {0, 1, 31, 32, 33}) { printf("lane %2d:", lane); for (int i = 0; i < 16; ++i) printf(" %4.0f", h[lane * 16 + i]); printf("\n"); }
  printf("== layout: for A slot s in lane half h with index nibbles n (idx word = n repeated): the logical k (= C - 1) the slot multiplied (lane 0, reg 0); -1 = none\n");
  for (int half = 0; half < 2; ++half)
    for (int slot = 0; slot < 8; ++slot) {
      printf("half %d slot %d:", half, slot);
      for (unsigned nib : {0x4u, 0x8u, 0xcu, 0x9u, 0xdu, 0xeu, 0x0u, 0x5u}) {   // (i0 | i1 << 2): (0,1) (0,2) (0,3) (1,2) (1,3) (2,3) (0,0) (1,1)
        unsigned w = 0; for (int q = 0; q < 8; ++q) w |= nib << (4 * q);
        hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, half, slot, w, o); hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost);
        // find the C value in the row that belongs to A row 0: report lane 0 reg 0 and, for orientation, the max over all
        float mx = 0.f; for (float v : h) mx = v > mx ? v : mx;
        printf("  n=%x:%3.0f", nib, mx - 1);
      }
      printf("\n");
    }
  printf("== index word: which nibble serves which slot pair (one nibble = 0xe, i.e. positions (2, 3), the others 0x4 = (0, 1)); per slot the logical k\n");
  for (int nibpos = 0; nibpos < 8; ++nibpos) {
    unsigned w = 0; for (int q = 0; q < 8; ++q) w |= (q == nibpos ? 0xeu : 0x4u) << (4 * q);
    printf("nibble %d = e:", nibpos);
    for (int slot = 0; slot < 8; ++slot) {
      hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, 0, slot, w, o); hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost);
      float mx = 0.f; for (float v : h) mx = v > mx ? v : mx;
      printf(" s%d->%2.0f", slot, mx - 1);
    }
    printf("\n");
  }
  // throughput
  std::vector<unsigned> ops(16384);
  unsigned* d; hipMalloc(&d, ops.size() * 4);
  float* so; hipMalloc(&so, 256 * 256 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& w : ops) {
      auto rb = [&]() { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; std::memcpy(&u, &f, 4); return mode ? (u >> 16) : 0u; };
      w = rb() | (rb() << 16);
    }
    hipMemcpy(d, ops.data(), ops.size() * 4, hipMemcpyHostToDevice);
    for (int sp = 0; sp < 2; ++sp) {
      const int iters = 20000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() { if (sp) hipLaunchKernelGGL(stream<true>, dim3(256), dim3(256), 0, 0, d, so, iters); else hipLaunchKernelGGL(stream<false>, dim3(256), dim3(256), 0, 0, d, so, iters); };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)iters * 16 * 4 * 256;   // wave-instructions
      printf("%s, %s operands: %.3f ms, %.2f G wave-MFMA/s = %.0f TF/s dense-equivalent (32x32x%d per instruction)\n", sp ? "v_smfmac_f32_32x32x32_bf16" : "v_mfma_f32_32x32x16_bf16  ",
             mode ? "random" : "zero", ms, insts / ms / 1e6, insts * 2.0 * 32 * 32 * (sp ? 32 : 16) / ms / 1e9, sp ? 32 : 16);
    }
  }
  return 0;
}
