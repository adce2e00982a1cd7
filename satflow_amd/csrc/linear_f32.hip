// Pointwise (1x1) linear maps over NHWC pixels on the fp32 matrix cores: the q/kv/out projections
// of the axial attention (lucidrains SelfAttention.to_q/to_kv/to_out) and
// MetNet's Conv2d(hid, out, 1x1) head (A.6; reference call site satflow/models/pl_metnet.py:46-59).
//
//   fwd :  y[p][n]  = sum_k x[p][k] * W[n][k] + b[n]           (also the input gradient, with W^T)
//   wgrad: dW[n][k] = sum_p dy[p][n] * x[p][k],  db[n] = sum_p dy[p][n]
//
// These are tall-skinny GEMMs (P ~ 25k pixels, N, K <= 384): no LDS staging; every lane loads its
// MFMA operand element(s) straight from global memory (16-byte loads in fwd, coalesced 128-byte
// rows in wgrad), weights stay L2/L1 resident.  v_mfma_f32_32x32x2_f32, exact fp32.
#include <type_traits>

#include "sf_common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int NF>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, int xs, long long rows, int K,
                                                         const float* __restrict__ W, int N, const float* __restrict__ bias,
                                                         float* __restrict__ y, int ys, int yc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kh = lane >> 5;
  const long long row0 = (long long)blockIdx.x * 128 + wave * 32;
  const int n0 = blockIdx.y * 32 * NF;
  f32x16 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;
  // loads are unconditional from clamped rows and selected afterwards: a load under `if (row < rows)` is emitted as
  // load -> s_waitcnt vmcnt(0) -> use, one K step at a time (compiler lesson)
  const long long row = row0 + r;
  const bool row_ok = row < rows;
  const bool vec_out = (((uintptr_t)y) & 15) == 0 && ys % 4 == 0;
  const float* xr = x + (row_ok ? row : rows - 1) * xs + kh * 4;
  const float* wr[NF]; bool n_ok[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = n0 + nf * 32 + r;
    n_ok[nf] = n < N;
    wr[nf] = W + (long long)(n_ok[nf] ? n : N - 1) * K + kh * 4;
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int q = 0; q < K / 8; ++q) {
    f32x4 a = ld4(xr + q * 8);
    f32x4 b[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b[nf] = ld4(wr[nf] + q * 8);
    a = row_ok ? a : zero;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b[nf] = n_ok[nf] ? b[nf] : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[nf][j], a[j], acc[nf], 0, 0, 0);  // transposed product
  }
  // D[i][j]: i = output column within the fragment (8g + 4kh + c), j = this lane's row: 16-byte stores of column quads
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + nf * 32 + 8 * g + 4 * kh;
      if (row_ok && n < yc) {
        f32x4 v = {acc[nf][4 * g], acc[nf][4 * g + 1], acc[nf][4 * g + 2], acc[nf][4 * g + 3]};
        if (bias) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += (n + c < N) ? bias[n + c] : 0.f;
        }
        if (vec_out && n + 3 < yc) *reinterpret_cast<f32x4*>(y + row * ys + n) = v;
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c) if (n + c < yc) y[row * ys + n + c] = v[c];
        }
      }
    }
  }
}

// The same map with 16-BIT OPERANDS (bf16 or fp16, fp32 accumulate: what the reference's 16-bit autocast runs a 1x1 Conv2d / Linear in; the 1x1
// convolutions of the DGMR discriminators and attention blocks in the bf16 / f16 modes, Discriminator.py:36-60,186-190): tensors stay fp32 in memory, a
// lane converts its 8 floats per 16-K step (RNE) and ONE v_mfma_f32_32x32x16 replaces eight 32x32x2 fp32 steps - the kernel is then bound by its loads
// (65536 x 256 -> 256: 134 MB), not by the 157 TF/s fp32 matrix rate.  Same tiling, same transposed product, same epilogue as linear_fwd_kernel.
template <int NF, typename OT>
__global__ __launch_bounds__(256) void linear_fwd_lowp_kernel(const float* __restrict__ x, int xs, long long rows, int K,
                                                              const float* __restrict__ W, int N, const float* __restrict__ bias,
                                                              float* __restrict__ y, int ys, int yc) {
  typedef OT ot8 __attribute__((ext_vector_type(8)));
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kh = lane >> 5;
  const long long row0 = (long long)blockIdx.x * 128 + wave * 32;
  const int n0 = blockIdx.y * 32 * NF;
  f32x16 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;
  const long long row = row0 + r;
  const bool row_ok = row < rows;
  const bool vec_out = (((uintptr_t)y) & 15) == 0 && ys % 4 == 0;
  const float* xr = x + (row_ok ? row : rows - 1) * xs;
  const float* wr[NF]; bool n_ok[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = n0 + nf * 32 + r;
    n_ok[nf] = n < N;
    wr[nf] = W + (long long)(n_ok[nf] ? n : N - 1) * K;
  }
  auto cvt = [](const f32x4 lo, const f32x4 hi, bool ok) {
    const f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const f32x8 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_convertvector(ok ? v : z, ot8);
  };
#pragma unroll 2
  for (int q = 0; q < (K + 15) / 16; ++q) {
    // this lane's eight K values of the step: 16 q + 8 kh ..; K is a multiple of 8, so the half is whole or absent (clamped: unconditional loads)
    const bool k_ok = q * 16 + kh * 8 < K;
    const int ko = k_ok ? q * 16 + kh * 8 : 0;
    const f32x4 a0 = ld4(xr + ko), a1 = ld4(xr + ko + 4);
    f32x4 b0[NF], b1[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) { b0[nf] = ld4(wr[nf] + ko); b1[nf] = ld4(wr[nf] + ko + 4); }
    const ot8 av = cvt(a0, a1, row_ok && k_ok);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const ot8 bv = cvt(b0[nf], b1[nf], n_ok[nf] && k_ok);
      if constexpr (std::is_same<OT, _Float16>::value) acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc[nf], 0, 0, 0);
      else acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[nf], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + nf * 32 + 8 * g + 4 * kh;
      if (row_ok && n < yc) {
        f32x4 v = {acc[nf][4 * g], acc[nf][4 * g + 1], acc[nf][4 * g + 2], acc[nf][4 * g + 3]};
        if (bias) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += (n + c < N) ? bias[n + c] : 0.f;
        }
        if (vec_out && n + 3 < yc) *reinterpret_cast<f32x4*>(y + row * ys + n) = v;
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c) if (n + c < yc) y[row * ys + n + c] = v[c];
        }
      }
    }
  }
}

// Few rows (a latent vector through a fully connected layer: 2 x 240 -> 65536; an attention projection on a 4x4 map): the 128-row MFMA tile above
// is nearly empty and its W loads are one 16-byte piece per lane from 32 different rows.  Here a WAVE owns one output column n and a block of up to 8
// rows: W[n][:] is read once, coalesced (the whole layer is then one pass over W at HBM speed), the x rows come from L1/L2, the 64 partial dot products
// are combined with xor-shuffles (fixed order: deterministic).
__global__ __launch_bounds__(256) void linear_fwd_fewrows_kernel(const float* __restrict__ x, int xs, int rows, int K, const float* __restrict__ W, int N,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int ys, int yc) {
  const int lane = threadIdx.x & 63;
  const int rblocks = (rows + 7) / 8;
  const long long tasks = (long long)yc * rblocks;
  for (long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); task < tasks; task += (long long)gridDim.x * 4) {
    const int n = (int)(task / rblocks), r0 = (int)(task - (long long)n * rblocks) * 8;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    if (n < N) {
      const float* wr = W + (long long)n * K;
      for (int k = lane * 4; k < K; k += 256) {
        const f32x4 w = ld4(wr + k);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int rr = r0 + r < rows ? r0 + r : rows - 1;   // clamped: unconditional loads
          const f32x4 a = ld4(x + (long long)rr * xs + k);
          acc[r] += w[0] * a[0] + w[1] * a[1] + w[2] * a[2] + w[3] * a[3];
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[r] += __shfl_xor(acc[r], m);
    }
    const float b = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (lane == r && r0 + r < rows) y[(long long)(r0 + r) * ys + n] = acc[r] + b;   // pad lanes n >= N: zeros, as the MFMA kernel writes them
  }
}

// dW partial: grid (KS, ceil(N/128)); wave w owns n rows 32w..32w+31 of the block's 128, all K/32 column fragments.
template <int KF>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, int dys, int N, const float* __restrict__ x,
                                                           int xs, int K, long long rows, int KS, float* __restrict__ partial,
                                                           float* __restrict__ partial_db, int Npad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kh = lane >> 5;
  const int n = blockIdx.y * 128 + wave * 32 + r;
  f32x16 acc[KF];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[kf][i] = 0.f;
  float asum = 0.f;
  const long long chunk = ((rows + KS - 1) / KS + 1) & ~1LL;  // even
  const long long p0 = (long long)blockIdx.x * chunk;
  const long long p1 = p0 + chunk < rows ? p0 + chunk : rows;
  // unconditional loads from clamped addresses, selected afterwards, 4 pixel pairs in flight (see linear_fwd_kernel)
  const int nc = n < N ? n : N - 1;
  int kc[KF]; bool k_ok[KF];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf) { k_ok[kf] = kf * 32 + r < K; kc[kf] = k_ok[kf] ? kf * 32 + r : K - 1; }
  if (p0 < p1) {
    for (long long p = p0; p < p1; p += 8) {
      float a[4], b[4][KF]; bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long pp = p + 2 * u + kh;
        ok[u] = pp < p1;
        const long long pc = ok[u] ? pp : p1 - 1;
        a[u] = dy[pc * dys + nc];
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) b[u][kf] = x[pc * xs + kc[kf]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float av = (ok[u] && n < N) ? a[u] : 0.f;
        asum += av;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
          acc[kf] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, (ok[u] && k_ok[kf]) ? b[u][kf] : 0.f, acc[kf], 0, 0, 0);
      }
    }
  }
  const int Kpad = KF * 32;
#pragma unroll
  for (int kf = 0; kf < KF; ++kf)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int nn = blockIdx.y * 128 + wave * 32 + frag_row(reg, kh);
      partial[((size_t)blockIdx.x * Npad + nn) * Kpad + kf * 32 + r] = acc[kf][reg];
    }
  const float tot = asum + __shfl_xor(asum, 32);
  if (kh == 0) partial_db[(size_t)blockIdx.x * Npad + blockIdx.y * 128 + wave * 32 + r] = tot;
}

// Sum of the KS split-K slabs.  The slabs are small (Npad x Kpad <= a few thousand outputs) and many (up to 384): one thread
// per output walking all of them is 32 workgroups of pure load latency (72 us measured).  Here 8 threads share an output, each
// sums every 8th slab with 8 loads in flight, and the 8 partial sums are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_db, int KS,
                                                                  int Npad, int Kpad, int N, int K, float* __restrict__ dW,
                                                                  float* __restrict__ db) {
  __shared__ float red[8][32];
  const int lane_o = threadIdx.x & 31, part = threadIdx.x >> 5;
  const size_t gid = (size_t)blockIdx.x * 32 + lane_o;
  const size_t slab = (size_t)Npad * Kpad;
  // outputs [0, slab): weight-gradient entries; [slab, slab + Npad): bias-gradient entries (when db is requested)
  const bool is_w = gid < slab, is_b = !is_w && db && gid < slab + (size_t)Npad;
  const float* src = is_w ? partial + gid : partial_db + (gid - slab);
  const size_t pitch = is_w ? slab : (size_t)Npad;
  float s = 0.f;
  if (is_w || is_b) {
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = part;
    for (; i + 56 < KS; i += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] += src[(size_t)(i + 8 * u) * pitch];
    }
    for (; i < KS; i += 8) s8[0] += src[(size_t)i * pitch];
    s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  }
  red[part][lane_o] = s;
  __syncthreads();
  if (part == 0 && (is_w || is_b)) {
    const float tot = ((red[0][lane_o] + red[1][lane_o]) + (red[2][lane_o] + red[3][lane_o])) +
                      ((red[4][lane_o] + red[5][lane_o]) + (red[6][lane_o] + red[7][lane_o]));
    if (is_w) {
      const int k = gid % Kpad, n = gid / Kpad;
      if (n < N && k < K) dW[(size_t)n * K + k] = tot;
    } else if (gid - slab < (size_t)N) {
      db[gid - slab] = tot;
    }
  }
}

struct WPlan { int KS, nblk, Npad, KF, Kpad; size_t floats; };
WPlan wplan(int N, int K, long long rows) {
  WPlan pl;
  pl.nblk = (N + 127) / 128; pl.Npad = pl.nblk * 128;
  pl.KF = (K + 31) / 32; pl.Kpad = pl.KF * 32;
  long long ks = 512 / pl.nblk;
  if (ks > rows / 64) ks = rows / 64;
  if (ks < 1) ks = 1;
  pl.KS = (int)ks;
  pl.floats = (size_t)pl.KS * ((size_t)pl.Npad * pl.Kpad + pl.Npad);
  return pl;
}

}  // namespace

extern "C" {

int sf_linear_fwd(sfTensor x, int64_t rows, const float* W, int32_t N, const float* bias, sfTensor y, int32_t dtype,
                  sfStream stream) {
  SF_REQUIRE(dtype == SF_F32 || dtype == SF_BF16 || dtype == SF_F16, "sf_linear_fwd: dtype %d not built", dtype);
  SF_F32_ONLY(x, "sf_linear_fwd");
  SF_F32_ONLY(y, "sf_linear_fwd");
  SF_REQUIRE(x.c % 8 == 0 && x.stride % 4 == 0 && (((uintptr_t)x.ptr) & 15) == 0 && (((uintptr_t)W) & 15) == 0,
             "linear: K=%d must be a multiple of 8 and 16-byte aligned", x.c);
  SF_REQUIRE(N >= 1 && y.c >= 1, "linear: N=%d y.c=%d", N, y.c);
  if (rows == 0) return 0;
  const int lanes = y.c > N ? y.c : N;
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 64) {   // GEMV-shaped: one wave per (output column, block of 8 rows)
    const long long tasks = (long long)y.c * ((rows + 7) / 8);
    const long long blocks = (tasks + 3) / 4;
    hipLaunchKernelGGL(linear_fwd_fewrows_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, (const float*)x.ptr, x.stride, (int)rows,
                       x.c, W, N, bias, (float*)y.ptr, y.stride, y.c);
    SF_CHECK_LAUNCH("linear_fwd_fewrows");
    return 0;
  }
  int nf = (lanes + 31) / 32;
  if (nf > 4) nf = 4;
  // a square-ish map (1024 rows x 1024 columns) is 64 tiles of 128 x 128: narrower column blocks until the 256 CUs have a workgroup each
  while (nf > 1 && nf != 3 && ((rows + 127) / 128) * ((lanes + 32 * nf - 1) / (32 * nf)) < 256) nf >>= 1;
  dim3 grid((unsigned)((rows + 127) / 128), (lanes + 32 * nf - 1) / (32 * nf));
#define SF_LIN(NFV)                                                                                                         \
  hipLaunchKernelGGL((linear_fwd_kernel<NFV>), grid, dim3(256), 0, st, (const float*)x.ptr, x.stride, (long long)rows, x.c, W, N, \
                     bias, (float*)y.ptr, y.stride, y.c)
#define SF_LINL(NFV, OTV)                                                                                                                \
  hipLaunchKernelGGL((linear_fwd_lowp_kernel<NFV, OTV>), grid, dim3(256), 0, st, (const float*)x.ptr, x.stride, (long long)rows, x.c, W, N, \
                     bias, (float*)y.ptr, y.stride, y.c)
  if (dtype == SF_BF16) { switch (nf) { case 1: SF_LINL(1, __bf16); break; case 2: SF_LINL(2, __bf16); break; case 3: SF_LINL(3, __bf16); break; default: SF_LINL(4, __bf16); break; } }
  else if (dtype == SF_F16) { switch (nf) { case 1: SF_LINL(1, _Float16); break; case 2: SF_LINL(2, _Float16); break; case 3: SF_LINL(3, _Float16); break; default: SF_LINL(4, _Float16); break; } }
  else
  switch (nf) { case 1: SF_LIN(1); break; case 2: SF_LIN(2); break; case 3: SF_LIN(3); break; default: SF_LIN(4); break; }
#undef SF_LINL
#undef SF_LIN
  SF_CHECK_LAUNCH("linear_fwd");
  return 0;
}

size_t sf_linear_bwd_weight_workspace_bytes(int32_t N, int32_t K, int64_t rows) { return wplan(N, K, rows).floats * sizeof(float); }

int sf_linear_bwd_weight(sfTensor dy, int32_t N, sfTensor x, int64_t rows, float* dW, float* db, void* workspace,
                         size_t workspace_bytes, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_linear_bwd_weight: dtype %d not built", dtype);
  SF_F32_ONLY(dy, "sf_linear_bwd_weight");
  SF_F32_ONLY(x, "sf_linear_bwd_weight");
  const int K = x.c;
  SF_REQUIRE(K >= 1 && K <= 256 && N >= 1 && dy.c >= N, "linear wgrad: N=%d K=%d (K <= 256)", N, K);
  const WPlan pl = wplan(N, K, rows);
  SF_REQUIRE(workspace && workspace_bytes >= pl.floats * sizeof(float), "linear wgrad: workspace too small");
  float* partial = (float*)workspace;
  float* pdb = partial + (size_t)pl.KS * pl.Npad * pl.Kpad;
  dim3 grid(pl.KS, pl.nblk);
  hipStream_t st = (hipStream_t)stream;
#define SF_LW(KFV)                                                                                                               \
  hipLaunchKernelGGL((linear_wgrad_kernel<KFV>), grid, dim3(256), 0, st, (const float*)dy.ptr, dy.stride, N, (const float*)x.ptr, \
                     x.stride, K, (long long)rows, pl.KS, partial, pdb, pl.Npad)
  switch (pl.KF) {
    case 1: SF_LW(1); break; case 2: SF_LW(2); break; case 3: SF_LW(3); break; case 4: SF_LW(4); break;
    case 5: SF_LW(5); break; case 6: SF_LW(6); break; case 7: SF_LW(7); break; default: SF_LW(8); break;
  }
#undef SF_LW
  SF_CHECK_LAUNCH("linear_wgrad");
  const size_t slab = (size_t)pl.Npad * pl.Kpad;
  hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((slab + pl.Npad + 31) / 32)), dim3(256), 0, st, partial, pdb, pl.KS, pl.Npad,
                     pl.Kpad, N, K, dW, db);
  SF_CHECK_LAUNCH("linear_wgrad_reduce");
  return 0;
}

}  // extern "C"
