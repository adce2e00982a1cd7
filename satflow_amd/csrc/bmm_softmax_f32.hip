// Strided batched matrix product and row softmax on the exact-fp32 matrix cores: the attention products of the in-tree attention
// layers in the form the reference writes them - torch.bmm on reshaped / permuted views, softmax(dim=-1) -
//   satflow/models/layers/Discriminator.py:104-126 (SelfAttention of both discriminators),
//   satflow/models/layers/Attention.py:23-109 (SeparableAttn), :112-170 (SelfAttention), :173-223 (SelfAttention2d).
// The views the reference multiplies are arbitrary (b, row, col) strided windows of NCHW / NHWC buffers (".view(B, A, -1)" of a
// transposed convolution output and the like), so every operand carries explicit element strides; operands are read straight from
// global memory (16-byte loads along k where an operand is k-contiguous and aligned).
//   C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][k][n]  (+ beta * C[b][m][n])
// v_mfma_f32_32x32x2_f32, one 32 x 32 tile per wave, 2 x 2 waves per workgroup.  Within a k-block of 8 the half-wave h owns
// k = 4 h .. 4 h + 3 (any pairing of k values to MFMA k-slots is valid as long as A and B agree), so a k-contiguous operand is
// one float4 per lane per block.
#include "sf_common.h"

namespace {

struct BmmParams {
  const float* A; const float* B; float* C;
  long long sAb, sAm, sAk, sBb, sBk, sBn, sCb, sCm, sCn;
  int batch, M, N, K, mtiles;
  float alpha, beta;
};

template <bool AVEC, bool BVEC>
__global__ __launch_bounds__(256) void bmm_f32_kernel(const BmmParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / p.mtiles, mt = blockIdx.x - b * p.mtiles;
  const int m0 = mt * 64 + (wave >> 1) * 32, n0 = blockIdx.y * 64 + (wave & 1) * 32;
  if (m0 >= p.M || n0 >= p.N) return;  // whole wave out of range (no barriers in this kernel)
  const int m = m0 + i < p.M ? m0 + i : p.M - 1;   // clamped operand row / column; masked at the store
  const int n = n0 + i < p.N ? n0 + i : p.N - 1;
  const float* Ap = p.A + (long long)b * p.sAb + (long long)m * p.sAm;
  const float* Bp = p.B + (long long)b * p.sBb + (long long)n * p.sBn;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int K8 = p.K & ~7;
  for (int k0 = 0; k0 < K8; k0 += 8) {
    const int k = k0 + 4 * h;
    float a[4], bb[4];
    if (AVEC) { const f32x4 v = *reinterpret_cast<const f32x4*>(Ap + k); a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3]; }
    else {
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = Ap[(long long)(k + s) * p.sAk];
    }
    if (BVEC) { const f32x4 v = *reinterpret_cast<const f32x4*>(Bp + k); bb[0] = v[0]; bb[1] = v[1]; bb[2] = v[2]; bb[3] = v[3]; }
    else {
#pragma unroll
      for (int s = 0; s < 4; ++s) bb[s] = Bp[(long long)(k + s) * p.sBk];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bb[s], acc, 0, 0, 0);
  }
  if (K8 < p.K) {  // ragged tail: clamped loads, zeroed operands
    const int k = K8 + 4 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = k + s < p.K;
      const int kc = ok ? k + s : p.K - 1;
      const float av = Ap[(long long)kc * p.sAk], bv = Bp[(long long)kc * p.sBk];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? av : 0.f, ok ? bv : 0.f, acc, 0, 0, 0);
    }
  }
  // D[row = m][col = n]: lane holds column n0 + i, rows frag_row(r, h)
  if (n0 + i < p.N) {
    float* Cp = p.C + (long long)b * p.sCb + (long long)(n0 + i) * p.sCn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = m0 + frag_row(r, h);
      if (mm < p.M) {
        float* d = Cp + (long long)mm * p.sCm;
        const float v = p.alpha * acc[r];
        *d = p.beta != 0.f ? v + p.beta * *d : v;
      }
    }
  }
}

// ---- the same product with bf16 MFMA operands (fp32 data in memory, fp32 accumulate) -----------------------------------------
// What torch.bmm computes under torch.autocast: used for the attention products in the bf16 compute modes.  The 16x higher MFMA rate
// needs operand reuse the direct-from-global kernel above does not have: a workgroup owns a 128 x 128 tile of C, stages 128 x 32 blocks of A and
// B through LDS as bf16 (rounded once, RNE; [row][k] with rows padded to 80 bytes: the MFMA operand of a lane is one ds_read_b128), 2 x 2 waves
// with 2 x 2 fragments of 32 x 32 each.  The global loads of block k+1 are in flight while block k is multiplied; one barrier per block (two LDS
// buffers).  MODE 0: the operand is k-contiguous (16-byte loads along k), MODE 1: any strides (lanes walk the row index - coalesced when that is
// the contiguous one).
constexpr int BK = 32, BT = 128, LROW = 40;   // k-block, tile edge, LDS row pitch in bf16 elements

template <int MODE>
__device__ __forceinline__ void bmm_load_block(const float* __restrict__ base, long long srow, long long sk, int rows_total, int row0, int k0, int K, float (&v)[16]) {
  const int t = threadIdx.x;
  if (MODE == 0) {
    const int row = row0 + (t >> 1);
    const float* ptr = base + (long long)(row < rows_total ? row : rows_total - 1) * srow + k0 + (t & 1) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = k0 + (t & 1) * 16 + 4 * j < K;   // K % 4 == 0 in this mode: a quad is inside or outside
      const f32x4 q = *reinterpret_cast<const f32x4*>(ok ? ptr + 4 * j : base);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[4 * j + c] = ok ? q[c] : 0.f;
    }
  } else {
    const int kq = k0 + (t >> 6) * 8;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = row0 + (t & 63) + 64 * rr;
      const float* ptr = base + (long long)(row < rows_total ? row : rows_total - 1) * srow;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = kq + j < K;
        const float x = ptr[(long long)(ok ? kq + j : 0) * sk];
        v[8 * rr + j] = ok ? x : 0.f;
      }
    }
  }
}
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// OT: the 16-bit operand type (__bf16: sf_bmm_bf16; _Float16: sf_bmm_f16)
template <typename OT> struct bmm_ops;
template <> struct bmm_ops<__bf16> {
  typedef bf16x8_t v8;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct bmm_ops<_Float16> {
  typedef f16x8_t v8;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <int MODE, typename OT>
__device__ __forceinline__ void bmm_store_block(OT* __restrict__ tile, const float (&v)[16]) {
  typedef typename bmm_ops<OT>::v8 v8;
  const int t = threadIdx.x;
  if (MODE == 0) {
    OT* d = tile + (t >> 1) * LROW + (t & 1) * 16;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      f32x8_t f;
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = v[8 * hlf + c];
      *reinterpret_cast<v8*>(d + 8 * hlf) = __builtin_convertvector(f, v8);
    }
  } else {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      f32x8_t f;
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = v[8 * rr + c];
      *reinterpret_cast<v8*>(tile + ((t & 63) + 64 * rr) * LROW + (t >> 6) * 8) = __builtin_convertvector(f, v8);
    }
  }
}

template <int AMODE, int BMODE, typename OT = __bf16>
__global__ __launch_bounds__(256) void bmm_bf16_kernel(const BmmParams p) {
  typedef typename bmm_ops<OT>::v8 v8;
  __shared__ __attribute__((aligned(16))) OT lds[2][2][BT * LROW];   // [buffer][A | B][row][k]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / p.mtiles, mt = blockIdx.x - b * p.mtiles;
  const int m0 = mt * BT, n0 = blockIdx.y * BT;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const float* Ab = p.A + (long long)b * p.sAb;
  const float* Bb = p.B + (long long)b * p.sBb;
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float va[16], vb[16];
  bmm_load_block<AMODE>(Ab, p.sAm, p.sAk, p.M, m0, 0, p.K, va);
  bmm_load_block<BMODE>(Bb, p.sBn, p.sBk, p.N, n0, 0, p.K, vb);
  bmm_store_block<AMODE>(lds[0][0], va);
  bmm_store_block<BMODE>(lds[0][1], vb);
  __syncthreads();
  const int nblk = (p.K + BK - 1) / BK;
  for (int kb = 0; kb < nblk; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nblk;
    if (more) {
      bmm_load_block<AMODE>(Ab, p.sAm, p.sAk, p.M, m0, (kb + 1) * BK, p.K, va);
      bmm_load_block<BMODE>(Bb, p.sBn, p.sBk, p.N, n0, (kb + 1) * BK, p.K, vb);
    }
    const OT* ta = lds[cur][0];
    const OT* tb = lds[cur][1];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v8 fa[2], fb[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) fa[x] = *reinterpret_cast<const v8*>(ta + (wm + 32 * x + i) * LROW + ks * 16 + 8 * h);
#pragma unroll
      for (int y = 0; y < 2; ++y) fb[y] = *reinterpret_cast<const v8*>(tb + (wn + 32 * y + i) * LROW + ks * 16 + 8 * h);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = bmm_ops<OT>::mfma(fa[x], fb[y], acc[x][y]);
    }
    if (more) {
      bmm_store_block<AMODE>(lds[cur ^ 1][0], va);
      bmm_store_block<BMODE>(lds[cur ^ 1][1], vb);
    }
    __syncthreads();
  }
  // D[m][n]: the lane holds column n = .. + i, rows frag_row(r, h) (as in bmm_f32_kernel)
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int n = n0 + wn + 32 * y + i;
    if (n >= p.N) continue;
    float* Cp = p.C + (long long)b * p.sCb + (long long)n * p.sCn;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = m0 + wm + 32 * x + frag_row(r, h);
        if (mm < p.M) {
          float* d = Cp + (long long)mm * p.sCm;
          const float v = p.alpha * acc[x][y][r];
          *d = p.beta != 0.f ? v + p.beta * *d : v;
        }
      }
  }
}

// ---- softmax over contiguous rows ------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return s;
}
__device__ __forceinline__ float block_add(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return s;
}

// one workgroup per row (rows of a few thousand elements: the row stays in L2 between the passes); y may alias x
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, long long rows, int L, float* __restrict__ y) {
  __shared__ float red[4];
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* xr = x + row * L;
    float* yr = y + row * L;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < L; j += 256) m = fmaxf(m, xr[j]);
    m = block_max(m, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < L; j += 256) s += __expf(xr[j] - m);
    s = block_add(s, red);
    const float inv = 1.f / s;
    for (int j = threadIdx.x; j < L; j += 256) yr[j] = __expf(xr[j] - m) * inv;
  }
}
// short rows: one wave per row, the row in registers (L <= 256)
__global__ __launch_bounds__(256) void softmax_fwd_wave_kernel(const float* __restrict__ x, long long rows, int L, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const float* xr = x + row * L;
    float v[4];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int j = lane + 64 * t; v[t] = j < L ? xr[j] : -INFINITY; m = fmaxf(m, v[t]); }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { v[t] = lane + 64 * t < L ? __expf(v[t] - m) : 0.f; s += v[t]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.f / s;
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int j = lane + 64 * t; if (j < L) y[row * L + j] = v[t] * inv; }
  }
}
// dx = y * (g - sum_j g_j y_j); dx may alias g
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, long long rows, int L, float* __restrict__ dx) {
  __shared__ float red[4];
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* gr = g + row * L;
    const float* yr = y + row * L;
    float s = 0.f;
    for (int j = threadIdx.x; j < L; j += 256) s = __builtin_fmaf(gr[j], yr[j], s);
    s = block_add(s, red);
    for (int j = threadIdx.x; j < L; j += 256) dx[row * L + j] = yr[j] * (gr[j] - s);
  }
}
__global__ __launch_bounds__(256) void softmax_bwd_wave_kernel(const float* __restrict__ g, const float* __restrict__ y, long long rows, int L, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    float gv[4], yv[4];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t;
      gv[t] = j < L ? g[row * L + j] : 0.f; yv[t] = j < L ? y[row * L + j] : 0.f;
      s = __builtin_fmaf(gv[t], yv[t], s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int j = lane + 64 * t; if (j < L) dx[row * L + j] = yv[t] * (gv[t] - s); }
  }
}

template <typename OT>
int bmm16_launch(const char* name, const float* A, int64_t sAb, int64_t sAm, int64_t sAk, const float* B, int64_t sBb, int64_t sBk, int64_t sBn, float* C,
                        int64_t sCb, int64_t sCm, int64_t sCn, int32_t batch, int32_t M, int32_t N, int32_t K, float alpha, float beta, sfStream stream) {
  SF_REQUIRE(A && B && C && batch >= 0 && M >= 0 && N >= 0 && K >= 1, "%s: null operand or bad extents (%d x %d x %d, batch %d)", name, M, N, K, batch);
  if (batch == 0 || M == 0 || N == 0) return 0;
  BmmParams p{A, B, C, sAb, sAm, sAk, sBb, sBk, sBn, sCb, sCm, sCn, batch, M, N, K, (M + BT - 1) / BT, alpha, beta};
  SF_REQUIRE((long long)batch * p.mtiles < 2147483647LL && (N + BT - 1) / BT <= 65535, "%s: grid too large", name);
  const bool avec = sAk == 1 && ((uintptr_t)A & 15) == 0 && sAb % 4 == 0 && sAm % 4 == 0 && K % 4 == 0;
  const bool bvec = sBk == 1 && ((uintptr_t)B & 15) == 0 && sBb % 4 == 0 && sBn % 4 == 0 && K % 4 == 0;
  dim3 grid((unsigned)(batch * p.mtiles), (N + BT - 1) / BT), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (avec && bvec) hipLaunchKernelGGL((bmm_bf16_kernel<0, 0, OT>), grid, block, 0, st, p);
  else if (avec) hipLaunchKernelGGL((bmm_bf16_kernel<0, 1, OT>), grid, block, 0, st, p);
  else if (bvec) hipLaunchKernelGGL((bmm_bf16_kernel<1, 0, OT>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((bmm_bf16_kernel<1, 1, OT>), grid, block, 0, st, p);
  SF_CHECK_LAUNCH(name);
  return 0;
}

}  // namespace

extern "C" {

int sf_bmm_f32(const float* A, int64_t sAb, int64_t sAm, int64_t sAk, const float* B, int64_t sBb, int64_t sBk, int64_t sBn, float* C, int64_t sCb, int64_t sCm,
               int64_t sCn, int32_t batch, int32_t M, int32_t N, int32_t K, float alpha, float beta, sfStream stream) {
  SF_REQUIRE(A && B && C && batch >= 0 && M >= 0 && N >= 0 && K >= 1, "sf_bmm_f32: null operand or bad extents (%d x %d x %d, batch %d)", M, N, K, batch);
  if (batch == 0 || M == 0 || N == 0) return 0;
  BmmParams p{A, B, C, sAb, sAm, sAk, sBb, sBk, sBn, sCb, sCm, sCn, batch, M, N, K, (M + 63) / 64, alpha, beta};
  SF_REQUIRE((long long)batch * p.mtiles < 2147483647LL && (N + 63) / 64 <= 65535, "sf_bmm_f32: grid too large");
  const bool avec = sAk == 1 && ((uintptr_t)A & 15) == 0 && sAb % 4 == 0 && sAm % 4 == 0;
  const bool bvec = sBk == 1 && ((uintptr_t)B & 15) == 0 && sBb % 4 == 0 && sBn % 4 == 0;
  dim3 grid((unsigned)(batch * p.mtiles), (N + 63) / 64), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (avec && bvec) hipLaunchKernelGGL((bmm_f32_kernel<true, true>), grid, block, 0, st, p);
  else if (avec) hipLaunchKernelGGL((bmm_f32_kernel<true, false>), grid, block, 0, st, p);
  else if (bvec) hipLaunchKernelGGL((bmm_f32_kernel<false, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((bmm_f32_kernel<false, false>), grid, block, 0, st, p);
  SF_CHECK_LAUNCH("bmm_f32");
  return 0;
}

int sf_bmm_bf16(const float* A, int64_t sAb, int64_t sAm, int64_t sAk, const float* B, int64_t sBb, int64_t sBk, int64_t sBn, float* C, int64_t sCb, int64_t sCm,
                int64_t sCn, int32_t batch, int32_t M, int32_t N, int32_t K, float alpha, float beta, sfStream stream) {
  return bmm16_launch<__bf16>("sf_bmm_bf16", A, sAb, sAm, sAk, B, sBb, sBk, sBn, C, sCb, sCm, sCn, batch, M, N, K, alpha, beta, stream);
}

int sf_bmm_f16(const float* A, int64_t sAb, int64_t sAm, int64_t sAk, const float* B, int64_t sBb, int64_t sBk, int64_t sBn, float* C, int64_t sCb, int64_t sCm,
               int64_t sCn, int32_t batch, int32_t M, int32_t N, int32_t K, float alpha, float beta, sfStream stream) {
  return bmm16_launch<_Float16>("sf_bmm_f16", A, sAb, sAm, sAk, B, sBb, sBk, sBn, C, sCb, sCm, sCn, batch, M, N, K, alpha, beta, stream);
}

int sf_softmax_rows_fwd(const float* x, int64_t rows, int32_t L, float* y, sfStream stream) {
  SF_REQUIRE(x && y && rows >= 0 && L >= 1, "sf_softmax_rows_fwd: null pointer or empty rows");
  if (rows == 0) return 0;
  if (L <= 256) hipLaunchKernelGGL(softmax_fwd_wave_kernel, dim3((unsigned)((rows + 3) / 4 < 65536 ? (rows + 3) / 4 : 65536)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, L, y);
  else hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)(rows < 262144 ? rows : 262144)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, L, y);
  SF_CHECK_LAUNCH("softmax_rows_fwd");
  return 0;
}

int sf_softmax_rows_bwd(const float* g, const float* y, int64_t rows, int32_t L, float* dx, sfStream stream) {
  SF_REQUIRE(g && y && dx && rows >= 0 && L >= 1, "sf_softmax_rows_bwd: null pointer or empty rows");
  if (rows == 0) return 0;
  if (L <= 256) hipLaunchKernelGGL(softmax_bwd_wave_kernel, dim3((unsigned)((rows + 3) / 4 < 65536 ? (rows + 3) / 4 : 65536)), dim3(256), 0, (hipStream_t)stream, g, y, (long long)rows, L, dx);
  else hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)(rows < 262144 ? rows : 262144)), dim3(256), 0, (hipStream_t)stream, g, y, (long long)rows, L, dx);
  SF_CHECK_LAUNCH("softmax_rows_bwd");
  return 0;
}

}  // extern "C"
