// Weight gradient of the 3x3 same convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16):
//
//   dW[tap][co][ci] = sum over pixels  bf16(dout[pixel][co]) * bf16(in[pixel + tap][ci]),   fp32 accumulate
//
// GEMM view: M = co, N = ci, K = pixels.  The MFMA wants 8 consecutive K values per lane, i.e. 8
// consecutive PIXELS of one channel, while HBM holds NHWC fp32 (channel-contiguous).
//
// Workgroup = 8 waves with two roles (wave-uniform branch), one 128(co) x 32(ci) x 9(tap) slab of dW:
//   * 4 LOADER waves transpose in registers: a lane loads float4 (4 channels) of 8..9 consecutive pixels
//     of one image row (coalesced along the channel axis across lanes), converts to bf16 and writes one
//     16-byte run of 8 pixels per channel into channel-major LDS tiles of the NEXT K tile
//     (A: dout [co][8 rows][16 px];  B: in [ci][10 halo rows][16 px] + one halo pixel left/right);
//   * 4 COMPUTE waves (one per SIMD, co fragment w) run the CURRENT K tile: per tile row (= one K step of
//     16 pixels) 1 A read + 3 B reads feed 9 MFMAs; the two horizontally shifted taps are built from the
//     aligned 16-byte run plus one neighbour dword with 4 v_alignbit each - no shifted LDS copies.
// LDS layout (found by exhaustive search over pitch / xor patterns): the 4 channels of a quad sit in
// adjacent 16-byte slots, slot index xor-ed with (quad>>1)&3, quad pitch = 68 (A) / 84 (B) slots:
// conflict-free for the loaders' ds_write_b128 (8-lane groups along the quad axis) AND for the compute
// waves' ds_read_b128 lane groups.  Two LDS buffers, one barrier per K tile.
// Split-K partial slabs in the fp32 kernel's layout; the shared reduce kernel finishes (the bias gradient
// is summed in fp32 by the loaders from the unrounded values).
#include "wgrad_common.h"

namespace {

using namespace sfwgrad;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KR = 8;                          // K-tile rows
constexpr int HR = KR + 2;                     // halo rows
constexpr int A_S = 68 * 16;                   // bytes per channel quad of the dout tile (1088)
constexpr int B_S = 84 * 16;                   // bytes per channel quad of the input tile (1344)
constexpr int H_P = HR * 8 + 4;                // bytes per channel of the halo-pixel array (84)
constexpr int A_BYTES = (CO_T / 4) * A_S;      // 34816
constexpr int B_BYTES = (CI_T / 4) * B_S;      // 10752
constexpr int H_BYTES = CI_T * H_P;            // 2688
constexpr int BUF = A_BYTES + B_BYTES + H_BYTES;  // 48256
constexpr int THREADS = 512, LOADERS = 256;

__device__ __forceinline__ int quad_slot(int cq, int c) { return (c ^ ((cq >> 1) & 3)) * 16; }

__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  f32x8 v = {a0, a1, a2, a3, a4, a5, a6, a7};
  return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ unsigned bf16_bits(float x) {
  __bf16 b = (__bf16)x;
  return (unsigned)__builtin_bit_cast(unsigned short, b);
}

__global__ __launch_bounds__(THREADS, 2) void wgrad_bf16_kernel(const WgradParams p, const int xcd_groups) {
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, kh = lane >> 5;

  // block id -> (ks, cot, cit): the (cot, cit) combinations of one K slice are consecutive on one XCD
  // (blocks are dealt round-robin to the 8 XCDs), so the slice's tiles are shared through that XCD's L2.
  const int combos = gridDim.y * gridDim.z;
  int ks, cot, cit;
  {
    const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (xcd_groups) {
      const int x = id % 8, j = id / 8;
      ks = (j / combos) * 8 + x;
      const int combo = j % combos;
      cot = combo % gridDim.y; cit = combo / gridDim.y;
    } else { ks = blockIdx.x; cot = blockIdx.y; cit = blockIdx.z; }
  }
  const int my_tiles = ks < p.ntiles ? (p.ntiles - ks + p.KS - 1) / p.KS : 0;

  if (wave >= 4) {
    // =========================== loader waves ===========================
    const int lt = tid - LOADERS;
    const int a_cq = lt % 32, b_cq = lt % 8;
    const int kcb = cit * CI_T + b_cq * 4;  // input channel (concatenated padded K space)
    const float* bsrc = nullptr; int bstride = 0, bdiv = 1, bmod = 0, bch = 0;
    if (kcb < p.c0) { bsrc = p.src0; bstride = p.s0; bdiv = p.idiv0; bmod = p.imod0; bch = kcb; }
    else if (kcb - p.c0 < p.c1) { bsrc = p.src1; bstride = p.s1; bdiv = p.idiv1; bmod = p.imod1; bch = kcb - p.c0; }
    const int aco = cot * CO_T + a_cq * 4;
    const bool a_ok = aco < p.dc;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // Register-staged software pipeline: the global loads of tile i+1 are issued right after tile i has been written
    // to LDS and stay in flight across the barrier, i.e. during the compute waves' work on tile i.
    const bool b_item = lt < 8 * HR * 2;
    const int brest = lt / 8, bhalf = brest & 1, bhrow = brest >> 1;
    f32x4 va[2][8], vb[10];
    auto load_tile = [&](int i) {
      int t = ks + i * p.KS;
      const int tx = t % p.tiles_x; t /= p.tiles_x;
      const int ty = t % p.tiles_y;
      const int n = t / p.tiles_y;
      const int x0 = tx * KT_W, y0 = ty * KR;
      // dout tile: items (cq 32, row 8, half 2) -> item = lt and lt + 256.  One row pointer per item, pixel j at
      // a constant stride from it (address arithmetic was the loaders' bottleneck: ~600 VALU per tile per wave).
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int rest = (lt + u * LOADERS) / 32, half = rest & 1, row = rest >> 1;
        const int gy = y0 + row, gx0 = x0 + 8 * half;
        const float* prow = p.dout + ((size_t)(n * p.H + gy) * p.W + gx0) * p.ds + aco;
        const int lim = (a_ok && gy < p.H) ? p.W - gx0 : 0;  // pixels j < lim are inside the image
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          va[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (j < lim) va[u][j] = *reinterpret_cast<const f32x4*>(prow + (size_t)j * p.ds);
        }
      }
      // input halo tile: items (cq 8, halo row 10, half 2) = 160 items; pixels x0-1+8*half .. +9
      int ns = n / bdiv; if (bmod) ns %= bmod;
      const int gy = y0 + bhrow - 1, gx0 = x0 - 1 + 8 * bhalf;
      const float* prow = bsrc + ((long long)(ns * p.H + gy) * p.W + gx0) * bstride + bch;
      const bool rok = b_item && bsrc && gy >= 0 && gy < p.H;
      const int lo = rok ? -gx0 : 99, hi = rok ? p.W - gx0 : 0;  // pixels lo <= j < hi are inside the image
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        vb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j >= lo && j < hi) vb[j] = *reinterpret_cast<const f32x4*>(prow + (long long)j * bstride);
      }
    };
    auto store_tile = [&](int i) {
      char* la = lds + (i & 1) * BUF;
      char* lb = la + A_BYTES;
      char* lh = lb + B_BYTES;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int rest = (lt + u * LOADERS) / 32, half = rest & 1, row = rest >> 1;
        char* dst = la + a_cq * A_S + (row * 2 + half) * 64;
        if (cit == 0) {  // block-uniform: only the first ci tile's blocks report the bias gradient
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum += va[u][j];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<bf16x8*>(dst + quad_slot(a_cq, c)) =
              pack8(va[u][0][c], va[u][1][c], va[u][2][c], va[u][3][c], va[u][4][c], va[u][5][c], va[u][6][c], va[u][7][c]);
      }
      if (b_item) {
        char* dst = lb + b_cq * B_S + (bhrow * 2 + bhalf) * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          *reinterpret_cast<bf16x8*>(dst + quad_slot(b_cq, c)) = pack8(vb[1][c], vb[2][c], vb[3][c], vb[4][c], vb[5][c], vb[6][c], vb[7][c], vb[8][c]);
          // halo pixels: left neighbour in the HIGH half of its dword, right neighbour in the LOW half
          unsigned* hp = reinterpret_cast<unsigned*>(lh + (b_cq * 4 + c) * H_P + bhrow * 8);
          if (bhalf == 0) hp[0] = bf16_bits(vb[0][c]) << 16;
          else            hp[1] = bf16_bits(vb[9][c]);
        }
      }
    };
    if (my_tiles > 0) load_tile(0);
    for (int i = 0; i <= my_tiles; ++i) {
      if (i < my_tiles) store_tile(i);
      if (i + 1 < my_tiles) load_tile(i + 1);
      __syncthreads();
    }
    // bias gradient: 8 loader threads share a channel quad
    if (cit == 0) {
      float* red = reinterpret_cast<float*>(lds);
      *reinterpret_cast<f32x4*>(red + lt * 4) = bsum;
    }
    __syncthreads();
  } else {
    // =========================== compute waves ===========================
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int co_l = 32 * wave + r;  // co within the tile
    const int a_off = (co_l >> 2) * A_S + quad_slot(co_l >> 2, co_l & 3) + kh * 64;
    const int b_q = (r >> 2) * B_S + quad_slot(r >> 2, r & 3);
    const int h_off = r * H_P;

    __syncthreads();  // tile 0 staged
    for (int i = 0; i < my_tiles; ++i) {
      const char* la = lds + (i & 1) * BUF;
      const char* lb = la + A_BYTES;
      const char* lh = lb + B_BYTES;
#pragma unroll
      for (int row = 0; row < KR; ++row) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(la + a_off + row * 128);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int hrow = row + ky;
          const char* rowp = lb + b_q + hrow * 128;
          const u32x4 mid = *reinterpret_cast<const u32x4*>(rowp + kh * 64);
          // neighbours: kh = 0 -> left halo dword / first dword of the upper half; kh = 1 -> last dword of the lower half / right halo dword
          const unsigned left = kh ? *reinterpret_cast<const unsigned*>(rowp + 12) : *reinterpret_cast<const unsigned*>(lh + h_off + hrow * 8);
          const unsigned right = kh ? *reinterpret_cast<const unsigned*>(lh + h_off + hrow * 8 + 4) : *reinterpret_cast<const unsigned*>(rowp + 64);
          u32x4 b0, b2;
          b0[0] = __builtin_amdgcn_alignbit(mid[0], left, 16);
          b0[1] = __builtin_amdgcn_alignbit(mid[1], mid[0], 16);
          b0[2] = __builtin_amdgcn_alignbit(mid[2], mid[1], 16);
          b0[3] = __builtin_amdgcn_alignbit(mid[3], mid[2], 16);
          b2[0] = __builtin_amdgcn_alignbit(mid[1], mid[0], 16);
          b2[1] = __builtin_amdgcn_alignbit(mid[2], mid[1], 16);
          b2[2] = __builtin_amdgcn_alignbit(mid[3], mid[2], 16);
          b2[3] = __builtin_amdgcn_alignbit(right, mid[3], 16);
          acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b0), acc[ky * 3 + 0], 0, 0, 0);
          acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, mid), acc[ky * 3 + 1], 0, 0, 0);
          acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b2), acc[ky * 3 + 2], 0, 0, 0);
        }
      }
      __syncthreads();
    }
    // partial[ks][tap][co][ci]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = cot * CO_T + 32 * wave + frag_row(reg, kh);
        const int ci = cit * CI_T + r;
        p.partial[(((size_t)ks * 9 + tap) * p.NpT + co) * p.KpT + ci] = acc[tap][reg];
      }
    __syncthreads();  // loaders have parked their bias sums
    if (cit == 0 && tid < 128) {
      const float* red = reinterpret_cast<const float*>(lds);
      const int cq = tid / 4, c = tid % 4;
      float s = 0.f;
      for (int k = 0; k < LOADERS / 32; ++k) s += red[(k * 32 + cq) * 4 + c];
      p.partial_db[(size_t)ks * p.NpT + cot * CO_T + tid] = s;
    }
  }
}

}  // namespace

int sf_launch_wgrad_bf16(const sfwgrad::WgradParams& p, const sfwgrad::Plan& pl, hipStream_t st) {
  if ((((uintptr_t)p.dout) & 15) || p.ds % 4 || p.dc % 4 || (p.src0 && ((((uintptr_t)p.src0) & 15) || p.s0 % 4)) ||
      (p.src1 && ((((uintptr_t)p.src1) & 15) || p.s1 % 4))) {
    sf_set_error("wgrad_bf16: tensors must be 16-byte aligned with 4-aligned strides");
    return 1;
  }
  const int xcd_groups = (pl.KS % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL(wgrad_bf16_kernel, dim3(pl.KS, pl.cot, pl.cit), dim3(THREADS), 0, st, p, xcd_groups);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error("wgrad_bf16: launch failed: %s", hipGetErrorString(e)); return 2; }
  return 0;
}
