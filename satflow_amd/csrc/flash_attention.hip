// Fused softmax(Q K^T) V for the self-attention layers of the DGMR / DVD-GAN discriminators (satflow/models/layers/Discriminator.py:104-126
// `SelfAttention`: energy = bmm(q, k^T), softmax(dim=-1), out = bmm(attention, v); the same in layers/Attention.py:173-223) in the 16-bit compute
// modes: operands rounded to bf16 / fp16 as torch.bmm's are under the reference's autocast (`precision: 16`), fp32 accumulation, fp32 softmax.
// The score matrix (N x N per frame: 64 x 64 positions -> 4096 x 4096 x 16 frames = 1 GB in fp32) never exists in memory: a workgroup owns 128
// queries, walks the keys in tiles of 64 and keeps running row maxima / sums (the usual online-softmax recurrence).
//
// Everything is computed TRANSPOSED so that a lane owns a QUERY (v_mfma_f32_32x32x16: D[row][col], lane = column):
//   S^T[j][i] = sum_d K[j][d] Q[i][d]          A = K tile rows (LDS, one ds_read_b128 per lane), B = this wave's 32 queries (registers, loaded once)
//   row statistics of query i                  in-lane over the 16 + 16 registers of the tile's two 32-key blocks, one xor-32 shuffle across the halves
//   O^T[e][i] += sum_j V[j][e] P[i][j]         B = P^T: its K index runs over keys in the order the S^T registers hold them (keys 4h + 8q + c of a
//                                              block: any order is fine as long as A uses the same one), so the exponentials are packed in place -
//                                              no lane ever hands a value to another; A = V^T read from the [key][channel] tile with the transposing
//                                              LDS read (ds_read_b64_tr_b16: four keys of this lane's channel per instruction)
// The backward pass (flash_attention_bwd_*) recomputes S^T from the saved log-sum-exp per query instead of reading saved probabilities.
#include "sf_common.h"

namespace {

typedef __bf16 fa_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fa_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 fa_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fa_f16x4 __attribute__((ext_vector_type(4)));
typedef float fa_f32x8 __attribute__((ext_vector_type(8)));

template <typename OT> struct fa_ops;
template <> struct fa_ops<__bf16> {
  typedef fa_bf16x8 v8; typedef fa_bf16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ v4 tr_read(unsigned a) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4*)(uintptr_t)a); }
};
template <> struct fa_ops<_Float16> {
  typedef fa_f16x8 v8; typedef fa_f16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ v4 tr_read(unsigned a) {  // (the read moves 16-bit patterns: the bf16 form of the builtin, re-typed)
    return __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) fa_bf16x4*)(uintptr_t)a));
  }
};

struct FlashParams {
  const float* q; const float* k; const float* v; int ldq, ldk, ldv;   // [batch][N][ld], fp32
  float* out; int ldo;                                                  // [batch][N][ldo]
  float* lse;                                                           // [batch][N]: row maximum + log(row sum) of the scaled scores
  int N, dqk, dv;                                                       // dqk in {16, 32}, dv a multiple of 32, <= 256, N a multiple of 128
  float scale;
  // backward
  const float* dout; int lddo;
  float* dq; float* dk; float* dvg; int lddq, lddk, lddv;
  float* delta;                                                         // [batch][N]: sum_e dout * out per query
};

constexpr int FA_KT = 64;         // keys per tile
// xor applied to the 64-byte channel groups of row `key` of a [key][channel] tile that is read with transposing loads (four consecutive keys per
// read): 512-byte rows and wider put the four keys on the same banks (xor by key & 3 within groups of four chunks), 128-byte rows the keys m and
// m + 2 (xor by (key >> 1) & 1), 64-byte rows none
template <int NT> __device__ __forceinline__ int fa_swz(int chunk, int key) {
  return NT >= 4 ? (chunk & ~3) | ((chunk & 3) ^ (key & 3)) : NT == 2 ? chunk ^ ((key >> 1) & 1) : chunk;
}
constexpr int FA_KPITCH = 80;     // bytes per K row in LDS (32 channels x 2 bytes = 64, padded: conflict-free ds_read_b128 over 32 rows)

// ---- forward -------------------------------------------------------------------------------------------------------------------------------
template <typename OT, int DV>
__global__ __launch_bounds__(256, 2) void flash_attention_fwd_kernel(const FlashParams p) {
  typedef typename fa_ops<OT>::v8 v8;
  typedef typename fa_ops<OT>::v4 v4;
  constexpr int NT = DV / 32;                 // 32-channel tiles of the value dimension
  constexpr int VPITCH = DV * 2;              // bytes per V row in LDS
  __shared__ __attribute__((aligned(1024))) char lds[FA_KT * FA_KPITCH + FA_KT * VPITCH];
  char* ldk = lds;
  char* ldv = lds + FA_KT * FA_KPITCH;
  const unsigned ldv0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldv;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int ks = p.dqk / 16;                  // k-steps of the score product (1 or 2)
  const float* qrow = p.q + ((long long)b * p.N + q0 + i) * p.ldq;

  // this wave's queries as the B operand of S^T: lane (i, h) holds Q[i][16 s + 8 h .. + 7]
  v8 qf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    fa_f32x8 f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (s < ks) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * h), c = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * h + 4);
      f = fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
    }
    qf[s] = __builtin_convertvector(f * p.scale, v8);
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // per-lane byte offset of the transposing V reads (see the header): key row = (lane >> 2) & 3 within a group of four, channel chunk by lane
  const int m4 = (lane >> 2) & 3;
  const int cbyte = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  for (int kt = 0; kt < p.N; kt += FA_KT) {
    __syncthreads();  // everybody is done with the previous tile
    // ---- stage the tile: K [64][dqk] and V [64][DV], fp32 -> 16 bit; V's 64-byte channel groups xor-ed with (key & 3) within groups of four ----
    {
      const int key = tid >> 2, part = tid & 3;          // 4 threads per key row: 8 channels each (dqk = 32), or the first two (dqk = 16)
      if (part * 8 < p.dqk) {
        const float* src = p.k + ((long long)b * p.N + kt + key) * p.ldk + part * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
        *reinterpret_cast<v8*>(ldk + key * FA_KPITCH + part * 16) = __builtin_convertvector(fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]}, v8);
      }
      for (int idx = tid; idx < FA_KT * (DV / 8); idx += 256) {
        const int vk = idx / (DV / 8), c8 = idx - vk * (DV / 8);   // key, group of 8 channels
        const float* src = p.v + ((long long)b * p.N + kt + vk) * p.ldv + c8 * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
        const int pchunk = fa_swz<NT>(c8 >> 2, vk);
        *reinterpret_cast<v8*>(ldv + vk * VPITCH + pchunk * 64 + (c8 & 3) * 16) = __builtin_convertvector(fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]}, v8);
      }
    }
    __syncthreads();

    // ---- scores of the tile's two 32-key blocks against this wave's queries ----
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < ks) {
          const v8 kf = *reinterpret_cast<const v8*>(ldk + (kb * 32 + i) * FA_KPITCH + (16 * s + 8 * h) * 2);
          st[kb] = fa_ops<OT>::mfma(kf, qf[s], st[kb]);
        }
    }
    // ---- online softmax for query i (both halves of the wave hold the same query, different keys) ----
    float tmax = st[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[kb][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[kb][r] = __expf(st[kb][r] - m_new); rs += st[kb][r]; }
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (__any(alpha != 1.f)) {  // (after the first tiles the running maximum rarely moves: wave-uniform skip)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
    }
    // ---- O^T += V^T P^T: K steps of 16 keys; step (kb, s2) takes registers 8 s2 .. 8 s2 + 7 of block kb = keys 16 s2 + 8 rho + 4 h + c ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        fa_f32x8 pf;
#pragma unroll
        for (int c = 0; c < 8; ++c) pf[c] = st[kb][8 * s2 + c];
        const v8 pb = __builtin_convertvector(pf, v8);
        const int key0 = kb * 32 + 16 * s2 + 4 * h + m4;   // + 8 rho
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int pchunk = fa_swz<NT>(t, key0);            // (key & 3, (key >> 1) & 1 are the same for both rho)
          const unsigned a0 = ldv0 + (unsigned)(key0 * VPITCH + pchunk * 64 + cbyte);
          const v4 lo = fa_ops<OT>::tr_read(a0), hi = fa_ops<OT>::tr_read(a0 + 8 * VPITCH);
          const v8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[t] = fa_ops<OT>::mfma(vf, pb, acc[t]);
        }
      }
  }
  // ---- normalise and store: lane (i, h) holds out[i][32 t + 4 h + 8 q + c] ----
  const float inv = 1.f / l_run;
  float* orow = p.out + ((long long)b * p.N + q0 + i) * p.ldo;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(orow + 32 * t + 4 * h + 8 * q) = f32x4{acc[t][4 * q] * inv, acc[t][4 * q + 1] * inv, acc[t][4 * q + 2] * inv, acc[t][4 * q + 3] * inv};
  if (h == 0 && p.lse) p.lse[(long long)b * p.N + q0 + i] = m_run + __logf(l_run);
}

template <typename OT>
int flash_fwd_launch(const FlashParams& p, int batch, hipStream_t st) {
  const dim3 grid(p.N / 128, batch), block(256);
  switch (p.dv) {
    case 32: hipLaunchKernelGGL((flash_attention_fwd_kernel<OT, 32>), grid, block, 0, st, p); break;
    case 64: hipLaunchKernelGGL((flash_attention_fwd_kernel<OT, 64>), grid, block, 0, st, p); break;
    case 128: hipLaunchKernelGGL((flash_attention_fwd_kernel<OT, 128>), grid, block, 0, st, p); break;
    case 256: hipLaunchKernelGGL((flash_attention_fwd_kernel<OT, 256>), grid, block, 0, st, p); break;
    default: sf_set_error("sf_flash_attention_fwd: value width %d not built (32, 64, 128, 256)", p.dv); return 1;
  }
  return 0;
}


// ---- backward ------------------------------------------------------------------------------------------------------------------------------
// dS = P o (dP - delta), dP = dO V^T, delta_i = sum_e dO[i][e] out[i][e];  dV = P^T dO,  dK = dS^T (scale q),  dQ = scale dS K.
// Two kernels, each recomputing the scores from the saved log-sum-exp (no atomics: every output has one owner):
//   flash_attention_bwd_dkdv_kernel: a workgroup owns 128 KEYS (lane = key column), walks the queries in tiles of 64;
//   flash_attention_bwd_dq_kernel:   a workgroup owns 128 QUERIES (lane = query column, as in the forward pass), walks the keys.
// The packing trick of the forward pass carries over: P / dS leave the score MFMA with a lane's 16 values spread over rows 4h + 8q + c, and that
// register order IS the K order of the next product's B operand; the matching A operand (dO^T, Q^T, K^T) comes out of the row-major tile through
// the transposing LDS read.  Tiles that are read both ways (rows as an A operand with K = channels, transposed with K = rows) use a pitch of
// 32 bytes beyond a multiple of 256: conflict-free transposing reads, two-way conflicts on the row reads.
constexpr int FA_QP = 96;   // bytes per row of a [row][32 channels] tile that is read both ways

__global__ __launch_bounds__(256) void flash_delta_kernel(const float* __restrict__ dout, int lddo, const float* __restrict__ out, int ldo, long long rows, int dv,
                                                          float* __restrict__ delta) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int e = lane * 4; e < dv; e += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(dout + row * lddo + e), b = *reinterpret_cast<const f32x4*>(out + row * ldo + e);
    s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) delta[row] = s;
}

// rows [r0, r0 + nrows) of a fp32 matrix -> a [row][32] 16-bit tile with FA_QP pitch, columns >= width zero, values scaled
template <typename OT>
__device__ __forceinline__ void fa_stage32(const float* __restrict__ src, int ld, int width, float scale, char* tile, int nrows, int tid) {
  typedef typename fa_ops<OT>::v8 v8;
  for (int idx = tid; idx < nrows * 4; idx += 256) {
    const int r = idx >> 2, part = idx & 3;
    fa_f32x8 f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (part * 8 < width) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + (long long)r * ld + part * 8), c = *reinterpret_cast<const f32x4*>(src + (long long)r * ld + part * 8 + 4);
      f = fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]} * scale;
    }
    *reinterpret_cast<v8*>(tile + r * FA_QP + part * 16) = __builtin_convertvector(f, v8);
  }
}
// rows of a fp32 [row][DV] matrix -> a 16-bit tile with the given pitch (no swizzle)
template <typename OT, int DV>
__device__ __forceinline__ void fa_stage_wide(const float* __restrict__ src, int ld, char* tile, int pitch, int nrows, int tid) {
  typedef typename fa_ops<OT>::v8 v8;
  for (int idx = tid; idx < nrows * (DV / 8); idx += 256) {
    const int r = idx / (DV / 8), c8 = idx - r * (DV / 8);
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + (long long)r * ld + c8 * 8), c = *reinterpret_cast<const f32x4*>(src + (long long)r * ld + c8 * 8 + 4);
    *reinterpret_cast<v8*>(tile + r * pitch + c8 * 16) = __builtin_convertvector(fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]}, v8);
  }
}
template <typename OT>
__device__ __forceinline__ typename fa_ops<OT>::v8 fa_tr8(unsigned addr, int pitch) {  // keys / queries row .. row + 3 and row + 8 .. row + 11 of this lane's channel
  typedef typename fa_ops<OT>::v4 v4;
  const v4 lo = fa_ops<OT>::tr_read(addr), hi = fa_ops<OT>::tr_read(addr + 8 * pitch);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <typename V8> __device__ __forceinline__ V8 fa_pack8(const f32x16& x, int s2) {
  fa_f32x8 f;
#pragma unroll
  for (int c = 0; c < 8; ++c) f[c] = x[8 * s2 + c];
  return __builtin_convertvector(f, V8);
}

template <typename OT, int DV>
__global__ __launch_bounds__(256, 1) void flash_attention_bwd_dkdv_kernel(const FlashParams p) {
  typedef typename fa_ops<OT>::v8 v8;
  constexpr int NT = DV / 32, VP = DV * 2 + 16, DP = DV * 2 + 32;
  __shared__ __attribute__((aligned(1024))) char lds[128 * VP + FA_KT * DP + FA_KT * FA_QP + 2 * FA_KT * 4];
  char* ldv = lds;                       // this workgroup's 128 value rows (row reads)
  char* ldd = lds + 128 * VP;            // dO tile (both ways)
  char* ldq = ldd + FA_KT * DP;          // (scale q) tile (both ways)
  float* lstat = reinterpret_cast<float*>(ldq + FA_KT * FA_QP);  // lse[64], delta[64]
  const unsigned ldd0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldd, ldq0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldq;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int k0 = blockIdx.x * 128;
  const int ks = p.dqk / 16;
  const int m4 = (lane >> 2) & 3, cbyte = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  // this wave's keys as the B operand of S = (scale q) k^T: lane (j, h) holds K[j][16 s + 8 h .. + 7]
  v8 kf[2];
  {
    const float* krow = p.k + ((long long)b * p.N + k0 + wave * 32 + j) * p.ldk;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      fa_f32x8 f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (s < ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(krow + 16 * s + 8 * h), c = *reinterpret_cast<const f32x4*>(krow + 16 * s + 8 * h + 4);
        f = fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      }
      kf[s] = __builtin_convertvector(f, v8);
    }
  }
  fa_stage_wide<OT, DV>(p.v + ((long long)b * p.N + k0) * p.ldv, p.ldv, ldv, VP, 128, tid);

  f32x16 dvt[NT], dkt;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dvt[t][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) dkt[r] = 0.f;

  for (int qt = 0; qt < p.N; qt += FA_KT) {
    __syncthreads();
    fa_stage32<OT>(p.q + ((long long)b * p.N + qt) * p.ldq, p.ldq, p.dqk, p.scale, ldq, FA_KT, tid);
    fa_stage_wide<OT, DV>(p.dout + ((long long)b * p.N + qt) * p.lddo, p.lddo, ldd, DP, FA_KT, tid);
    if (tid < FA_KT) lstat[tid] = p.lse[(long long)b * p.N + qt + tid];
    else if (tid < 2 * FA_KT) lstat[tid] = p.delta[(long long)b * p.N + qt + tid - FA_KT];
    __syncthreads();
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {
      // S[i][j]: A = (scale q) rows of the block, B = this wave's keys
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < ks) st = fa_ops<OT>::mfma(*reinterpret_cast<const v8*>(ldq + (qb * 32 + j) * FA_QP + (16 * s + 8 * h) * 2), kf[s], st);
      // dP[i][j]: A = dO rows, B = this wave's value rows (K = channels)
#pragma unroll
      for (int s = 0; s < DV / 16; ++s)
        dp = fa_ops<OT>::mfma(*reinterpret_cast<const v8*>(ldd + (qb * 32 + j) * DP + (16 * s + 8 * h) * 2),
                              *reinterpret_cast<const v8*>(ldv + (wave * 32 + j) * VP + (16 * s + 8 * h) * 2), dp);
      // P and dS for rows i = 4 h + 8 q + c of the block
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lstat + qb * 32 + 4 * h + 8 * q), d4 = *reinterpret_cast<const f32x4*>(lstat + FA_KT + qb * 32 + 4 * h + 8 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float pr = __expf(st[4 * q + c] - l4[c]);
          st[4 * q + c] = pr;
          dp[4 * q + c] = pr * (dp[4 * q + c] - d4[c]);
        }
      }
      // dV^T[e][j] += dO^T P,  dK^T[d][j] += (scale q)^T dS: K steps of 16 queries in register order
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const v8 pb = fa_pack8<v8>(st, s2), db = fa_pack8<v8>(dp, s2);
        const int row0 = qb * 32 + 16 * s2 + 4 * h + m4;
#pragma unroll
        for (int t = 0; t < NT; ++t) dvt[t] = fa_ops<OT>::mfma(fa_tr8<OT>(ldd0 + (unsigned)(row0 * DP + t * 64 + cbyte), DP), pb, dvt[t]);
        dkt = fa_ops<OT>::mfma(fa_tr8<OT>(ldq0 + (unsigned)(row0 * FA_QP + cbyte), FA_QP), db, dkt);
      }
    }
  }
  // lane (j, h) holds dV[j][32 t + 4 h + 8 q + c] and dK[j][4 h + 8 q + c]
  float* vrow = p.dvg + ((long long)b * p.N + k0 + wave * 32 + j) * p.lddv;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(vrow + 32 * t + 4 * h + 8 * q) = f32x4{dvt[t][4 * q], dvt[t][4 * q + 1], dvt[t][4 * q + 2], dvt[t][4 * q + 3]};
  float* krow = p.dk + ((long long)b * p.N + k0 + wave * 32 + j) * p.lddk;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (4 * h + 8 * q < p.dqk) *reinterpret_cast<f32x4*>(krow + 4 * h + 8 * q) = f32x4{dkt[4 * q], dkt[4 * q + 1], dkt[4 * q + 2], dkt[4 * q + 3]};
}

template <typename OT, int DV>
__global__ __launch_bounds__(256, 2) void flash_attention_bwd_dq_kernel(const FlashParams p) {
  typedef typename fa_ops<OT>::v8 v8;
  constexpr int VP = DV * 2 + 16;
  __shared__ __attribute__((aligned(1024))) char lds[FA_KT * FA_QP + FA_KT * VP];
  char* ldk = lds;                     // K tile (both ways)
  char* ldv = lds + FA_KT * FA_QP;     // V tile (row reads)
  const unsigned ldk0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ldk;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int ks = p.dqk / 16;
  const int m4 = (lane >> 2) & 3, cbyte = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const long long row = (long long)b * p.N + q0 + i;

  v8 qf[2], dof[DV / 16];
  {
    const float* qrow = p.q + row * p.ldq;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      fa_f32x8 f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (s < ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * h), c = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * h + 4);
        f = fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      }
      qf[s] = __builtin_convertvector(f * p.scale, v8);
    }
    const float* drow = p.dout + row * p.lddo;
#pragma unroll
    for (int s = 0; s < DV / 16; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(drow + 16 * s + 8 * h), c = *reinterpret_cast<const f32x4*>(drow + 16 * s + 8 * h + 4);
      dof[s] = __builtin_convertvector(fa_f32x8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]}, v8);
    }
  }
  const float lse_i = p.lse[row], delta_i = p.delta[row];
  f32x16 dqt;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqt[r] = 0.f;

  for (int kt = 0; kt < p.N; kt += FA_KT) {
    __syncthreads();
    fa_stage32<OT>(p.k + ((long long)b * p.N + kt) * p.ldk, p.ldk, p.dqk, 1.f, ldk, FA_KT, tid);
    fa_stage_wide<OT, DV>(p.v + ((long long)b * p.N + kt) * p.ldv, p.ldv, ldv, VP, FA_KT, tid);
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < ks) st = fa_ops<OT>::mfma(*reinterpret_cast<const v8*>(ldk + (kb * 32 + i) * FA_QP + (16 * s + 8 * h) * 2), qf[s], st);
#pragma unroll
      for (int s = 0; s < DV / 16; ++s) dp = fa_ops<OT>::mfma(*reinterpret_cast<const v8*>(ldv + (kb * 32 + i) * VP + (16 * s + 8 * h) * 2), dof[s], dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = __expf(st[r] - lse_i) * (dp[r] - delta_i) * p.scale;   // scale dS^T[j][i]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int row0 = kb * 32 + 16 * s2 + 4 * h + m4;
        dqt = fa_ops<OT>::mfma(fa_tr8<OT>(ldk0 + (unsigned)(row0 * FA_QP + cbyte), FA_QP), fa_pack8<v8>(dp, s2), dqt);
      }
    }
  }
  float* qrow = p.dq + row * p.lddq;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (4 * h + 8 * q < p.dqk) *reinterpret_cast<f32x4*>(qrow + 4 * h + 8 * q) = f32x4{dqt[4 * q], dqt[4 * q + 1], dqt[4 * q + 2], dqt[4 * q + 3]};
}

template <typename OT>
int flash_bwd_launch(const FlashParams& p, int batch, hipStream_t st) {
  const dim3 grid(p.N / 128, batch), block(256);
#define SF_FA_BWD(DVV)                                                                              \
  case DVV:                                                                                         \
    hipLaunchKernelGGL((flash_attention_bwd_dkdv_kernel<OT, DVV>), grid, block, 0, st, p);          \
    hipLaunchKernelGGL((flash_attention_bwd_dq_kernel<OT, DVV>), grid, block, 0, st, p);            \
    break;
  switch (p.dv) {
    SF_FA_BWD(32) SF_FA_BWD(64) SF_FA_BWD(128) SF_FA_BWD(256)
    default: sf_set_error("sf_flash_attention_bwd: value width %d not built (32, 64, 128, 256)", p.dv); return 1;
  }
#undef SF_FA_BWD
  return 0;
}

}  // namespace

extern "C" int sf_flash_attention_fwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, int32_t batch, int32_t n,
                                      int32_t dqk, int32_t dv, float scale, float* out, int32_t ldo, float* lse, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16 || dtype == SF_F16, "sf_flash_attention_fwd: dtype %d not built (the fused form exists for the 16-bit compute modes only)", dtype);
  SF_REQUIRE(q && k && v && out, "sf_flash_attention_fwd: null operand");
  SF_REQUIRE(batch >= 0 && n >= 0 && n % 128 == 0 && (dqk == 16 || dqk == 32) && dv >= 32 && dv % 32 == 0 && dv <= 256,
             "sf_flash_attention_fwd: n = %d (multiple of 128), dqk = %d (16 or 32), dv = %d (32, 64, 128, 256)", n, dqk, dv);
  SF_REQUIRE(ldq >= dqk && ldk >= dqk && ldv >= dv && ldo >= dv && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 &&
                 ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0,
             "sf_flash_attention_fwd: rows must be 16-byte aligned and at least as wide as the operand");
  if (batch == 0 || n == 0) return 0;
  FlashParams p{};
  p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.out = out; p.ldo = ldo; p.lse = lse;
  p.N = n; p.dqk = dqk; p.dv = dv; p.scale = scale;
  const int rc = dtype == SF_BF16 ? flash_fwd_launch<__bf16>(p, batch, (hipStream_t)stream) : flash_fwd_launch<_Float16>(p, batch, (hipStream_t)stream);
  if (rc) return rc;
  SF_CHECK_LAUNCH("flash_attention_fwd");
  return 0;
}

extern "C" int sf_flash_attention_bwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* out, int32_t ldo,
                                      const float* lse, const float* dout, int32_t lddo, int32_t batch, int32_t n, int32_t dqk, int32_t dv, float scale,
                                      float* dq, int32_t lddq, float* dk, int32_t lddk, float* dvg, int32_t lddv, float* delta, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16 || dtype == SF_F16, "sf_flash_attention_bwd: dtype %d not built (the fused form exists for the 16-bit compute modes only)", dtype);
  SF_REQUIRE(q && k && v && out && lse && dout && dq && dk && dvg && delta, "sf_flash_attention_bwd: null operand");
  SF_REQUIRE(batch >= 0 && n >= 0 && n % 128 == 0 && (dqk == 16 || dqk == 32) && dv >= 32 && dv % 32 == 0 && dv <= 256,
             "sf_flash_attention_bwd: n = %d (multiple of 128), dqk = %d (16 or 32), dv = %d (32, 64, 128, 256)", n, dqk, dv);
  SF_REQUIRE(ldq >= dqk && ldk >= dqk && ldv >= dv && ldo >= dv && lddo >= dv && lddq >= dqk && lddk >= dqk && lddv >= dv &&
                 (ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) % 4 == 0 &&
                 ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out) | ((uintptr_t)dout) | ((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dvg)) & 15) == 0,
             "sf_flash_attention_bwd: rows must be 16-byte aligned and at least as wide as the operand");
  if (batch == 0 || n == 0) return 0;
  FlashParams p{};
  p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.lse = const_cast<float*>(lse);
  p.N = n; p.dqk = dqk; p.dv = dv; p.scale = scale;
  p.dout = dout; p.lddo = lddo; p.dq = dq; p.dk = dk; p.dvg = dvg; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv; p.delta = delta;
  const long long rows = (long long)batch * n;
  hipLaunchKernelGGL(flash_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout, lddo, out, ldo, rows, dv, delta);
  const int rc = dtype == SF_BF16 ? flash_bwd_launch<__bf16>(p, batch, (hipStream_t)stream) : flash_bwd_launch<_Float16>(p, batch, (hipStream_t)stream);
  if (rc) return rc;
  SF_CHECK_LAUNCH("flash_attention_bwd");
  return 0;
}
