// HBM-bound kernels of the MetNet image encoder (upstream metnet MetNetPreprocessor / DownSampler,
// SURVEY Appendix A; reference call site satflow/models/pl_metnet.py:46-59,65):
//   - preprocessing: pixel-unshuffle(2) + centre crop, 2x2 means, concat -> NHWC frames
//   - 2x2/stride-2 max pooling, forward and backward (argmax recomputed, no index tensor)
//   - training-mode BatchNorm2d with statistics per lead-time group of frames:
//     reduce (fp64 accumulation across blocks) -> finalize (scale/shift, running stats) ->
//     apply; backward = reduce + apply.
// All NHWC, 16-byte accesses along the channel axis, one pass each.
#include <cstdlib>
#include <type_traits>

#include "sf_common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// activation storage dispatch: TA = float or __bf16 (sfTensor.dtype); arithmetic is fp32 either way
#define SF_DISPATCH_ACT(dt, ...)                       \
  do {                                                 \
    if ((dt) == SF_BF16) { using TA = __bf16; __VA_ARGS__; } \
    else { using TA = float; __VA_ARGS__; }            \
  } while (0)

// ---------------------------------------------------------------------------------------------
// preprocessing.  imgs[B][T][C][H][W] -> out frame j = t*B + b, [S][S][Cp], S = H/4 = W/4:
//   lanes [0, 4*sat)        sat channel c, sub-pixel (dh,dw) -> lane c*4 + dh*2 + dw, centre crop of the H/2 map
//   lanes [4*sat, 8*sat)    the same unshuffled channels, 2x2 mean
//   lanes [8*sat, 8*sat+C-sat) other channels: 2x2 mean of the raw image, centre crop
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ imgs, int B, int T, int C, int sat, int H, int W,
                                                         int S, TO* __restrict__ out, int oc, int os) {
  const long long total = (long long)B * T * S * S;
  // torchvision CenterCrop: int(round(d / 2.0)) with Python's round-half-to-even
  const int dT = H / 2 - S, dL = W / 2 - S;
  const int top = (dT & 1) ? ((dT / 2) & 1 ? dT / 2 + 1 : dT / 2) : dT / 2;
  const int left = (dL & 1) ? ((dL / 2) & 1 ? dL / 2 + 1 : dL / 2) : dL / 2;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int x = idx % S, y = (idx / S) % S;
    const long long j = idx / ((long long)S * S);
    const int b = j % B, t = j / B;
    const float* src = imgs + ((long long)b * T + t) * C * H * W;
    TO* dst = out + idx * os;
    // 4 satellite channels per round: all their loads (4 float4 rows of the 4x4 raw window + 2 float2 rows of the centre-crop
    // window each, channel index clamped) are issued before the first store - hipcc otherwise serialises load -> wait -> store
    // per channel, and vmcnt counts the stores too
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    // The two output quads of every satellite channel are kept in registers and stored back to back at the end (up to
    // LATE_SAT channels): interleaved with the next round's loads the 8-byte stores of one 192-byte pixel record reached
    // memory as scattered partial lines (291 -> 184 us for the BASELINE shape).
    constexpr int LATE_SAT = 12;
    const bool late = sat <= LATE_SAT;
    f32x4 res_c[LATE_SAT], res_m[LATE_SAT];
    for (int c0 = 0; c0 < sat; c0 += 4) {
      f32x4 R[4][4]; f32x2_t Ct[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* ch = src + (long long)(c0 + u < sat ? c0 + u : sat - 1) * H * W;
#pragma unroll
        for (int a = 0; a < 4; ++a) R[u][a] = *reinterpret_cast<const f32x4*>(ch + (long long)(4 * y + a) * W + 4 * x);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) Ct[u][dh] = *reinterpret_cast<const f32x2_t*>(ch + (long long)(2 * (y + top) + dh) * W + 2 * (x + left));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (c0 + u < sat) {
          const int c = c0 + u;
          f32x4 ctr, mean;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int dh = d >> 1, dw = d & 1;
            ctr[d] = Ct[u][dh][dw];
            mean[d] = ((R[u][dh][dw] + R[u][dh][2 + dw]) + (R[u][2 + dh][dw] + R[u][2 + dh][2 + dw])) * 0.25f;
          }
          if (late) {
#pragma unroll
            for (int k = 0; k < LATE_SAT; ++k)
              if (k == c) { res_c[k] = ctr; res_m[k] = mean; }
          } else {
            stv4(dst + c * 4, ctr);
            stv4(dst + 4 * sat + c * 4, mean);
          }
        }
      }
    }
    if (late) {
#pragma unroll
      for (int k = 0; k < LATE_SAT; ++k)
        if (k < sat) stv4(dst + k * 4, res_c[k]);
#pragma unroll
      for (int k = 0; k < LATE_SAT; ++k)
        if (k < sat) stv4(dst + 4 * sat + k * 4, res_m[k]);
    }
    for (int c = sat; c < C; ++c) {
      const float* ch = src + (long long)c * H * W;
      const int yy = 2 * (y + top), xx = 2 * (x + left);
      dst[8 * sat + (c - sat)] = (TO)(0.25f * (ch[(long long)yy * W + xx] + ch[(long long)yy * W + xx + 1] +
                                               ch[(long long)(yy + 1) * W + xx] + ch[(long long)(yy + 1) * W + xx + 1]));
    }
    for (int c = 8 * sat + (C - sat); c < oc; ++c) dst[c] = (TO)0.f;
  }
}

// Gradient of the preprocessing wrt the raw images (a gather: every raw pixel feeds at most two output lanes).
// One thread per 4 horizontally consecutive raw pixels (one 16-byte store); dimgs is written completely.
template <typename TO>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(const TO* __restrict__ dout, int os, int B, int T, int C, int sat, int H, int W,
                                                             int S, float* __restrict__ dimgs) {
  const int W4 = W / 4;
  const long long total = (long long)B * T * C * H * W4;
  const int dT = H / 2 - S, dL = W / 2 - S;
  const int top = (dT & 1) ? ((dT / 2) & 1 ? dT / 2 + 1 : dT / 2) : dT / 2;
  const int left = (dL & 1) ? ((dL / 2) & 1 ? dL / 2 + 1 : dL / 2) : dL / 2;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int x4 = idx % W4;
    long long r = idx / W4;
    const int Y = r % H; r /= H;
    const int c = r % C; r /= C;
    const int t = r % T, b = r / T;
    const long long frame = (long long)t * B + b;
    const TO* fr = dout + frame * S * S * os;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int yc = Y / 2 - top;  // row of the centre-cropped half-resolution map
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int X = 4 * x4 + k;
      const int xc = X / 2 - left;
      const bool in_crop = yc >= 0 && yc < S && xc >= 0 && xc < S;
      const int sub = (Y & 1) * 2 + (X & 1);
      float v = 0.f;
      if (c < sat) {
        v = 0.25f * (float)fr[((long long)(Y / 4) * S + x4) * os + 4 * sat + c * 4 + sub];
        if (in_crop) v += (float)fr[((long long)yc * S + xc) * os + c * 4 + sub];
      } else if (in_crop) {
        v = 0.25f * (float)fr[((long long)yc * S + xc) * os + 8 * sat + (c - sat)];
      }
      g[k] = v;
    }
    *reinterpret_cast<f32x4*>(dimgs + (((long long)b * T + t) * C + c) * H * W + (long long)Y * W + 4 * x4) = g;
  }
}

// ---------------------------------------------------------------------------------------------
// max pooling 2x2 stride 2
// ---------------------------------------------------------------------------------------------
struct OuterPerm { int L, T, B; };  // L == 0: identity.  pooled image of input image (l*T+t)*B+b is (t*L+l)*B+b
__device__ __forceinline__ long long perm_image(long long n, const OuterPerm& pm) {
  if (pm.L == 0) return n;
  const long long b = n % pm.B, t = (n / pm.B) % pm.T, l = n / ((long long)pm.B * pm.T);
  return (t * pm.L + l) * pm.B + b;
}

// DROP: the encoder-output dropouts (sf_dropout2's masks, indexed by the flat element index of the contiguous pooled tensor;
// npt = pooled images per timestep for the period-reduced index of the sequence-consistent mask) applied on the way out
// 8 channels per thread (one 16-byte access per bf16 tensor); the dropout masks are defined per channel quad: two per thread
__device__ __forceinline__ f32x8_t drop_scales8(const sfDrop& dr, unsigned long long g, unsigned long long g2) {
  const f32x4 a = sf_drop_scales(dr, g, g2), b = sf_drop_scales(dr, g + 1, g2 + 1);
  return f32x8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <typename TI, typename TO, bool DROP>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const TI* __restrict__ in, int is, long long N, int H, int W, int C,
                                                          TO* __restrict__ out, int os, const OuterPerm pm, const sfDrop dr, long long npt,
                                                          unsigned short* __restrict__ route) {
  const int Ho = H / 2, Wo = W / 2, q = C / 8;
  const long long total = N * Ho * Wo * q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (idx % q) * 8;
    const long long op = idx / q;
    const int xo = op % Wo, yo = (op / Wo) % Ho;
    const long long n = op / ((long long)Wo * Ho);
    const TI* p = in + ((n * H + 2 * yo) * W + 2 * xo) * is + c;
    f32x8_t m = ldv8(p);
    const f32x8_t v1 = ldv8(p + is), v2 = ldv8(p + (long long)W * is), v3 = ldv8(p + (long long)W * is + is);
    if (route) {  // which window element each channel took (2 bits per channel: first maximum in row-major order), [input window idx]
      unsigned rt = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int am = 0; float mm = m[j];
        if (v1[j] > mm) { mm = v1[j]; am = 1; }
        if (v2[j] > mm) { mm = v2[j]; am = 2; }
        if (v3[j] > mm) { mm = v3[j]; am = 3; }
        rt |= (unsigned)am << (2 * j);
      }
      route[idx] = (unsigned short)rt;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaxf(m[j], v1[j]), fmaxf(v2[j], v3[j]));
    const long long no = perm_image(n, pm);
    if constexpr (DROP) {
      const unsigned long long pixq = (unsigned long long)((yo * Wo + xo) * (C / 4) + c / 4), imgq = (unsigned long long)Ho * Wo * (C / 4);
      m = m * drop_scales8(dr, (unsigned long long)no * imgq + pixq, (unsigned long long)(no % npt) * imgq + pixq);
    }
    stv8(out + ((no * Ho + yo) * Wo + xo) * os + c, m);
  }
}

template <typename TI, typename TO, bool DROP>  // TI: input and its gradient, TO: pooled output's gradient
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const TI* __restrict__ in, int is, const TO* __restrict__ dout, int dos,
                                                          long long N, int H, int W, int C, TI* __restrict__ din, int dis,
                                                          const OuterPerm pm, const sfDrop dr, long long npt, const unsigned short* __restrict__ route,
                                                          TO* __restrict__ gmask, unsigned* __restrict__ amax) {
  // amax (nullable): raised to max |din| = the largest masked pooled gradient (every din value is one of them or zero): the SF_F32E kernels' scale word
  // of din (sfTensor::amax), zeroed by the caller; one atomic per wave at most
  const int Ho = H / 2, Wo = W / 2, q = C / 8;
  const long long total = N * Ho * Wo * q;
  unsigned vmax = 0u;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (idx % q) * 8;
    const long long op = idx / q;
    const int xo = op % Wo, yo = (op / Wo) % Ho;
    const long long n = op / ((long long)Wo * Ho);
    const long long base = (n * H + 2 * yo) * W + 2 * xo;
    f32x8_t v0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
    unsigned rt = 0;
    if (route) rt = route[idx];  // the forward pass recorded the routing: the input is not read again
    else {
      const TI* p = in + base * is + c;
      v0 = ldv8(p); v1 = ldv8(p + is); v2 = ldv8(p + (long long)W * is); v3 = ldv8(p + (long long)W * is + is);
    }
    const long long no = perm_image(n, pm);
    f32x8_t g = ldv8(dout + ((no * Ho + yo) * Wo + xo) * dos + c);
    if constexpr (DROP) {
      const unsigned long long pixq = (unsigned long long)((yo * Wo + xo) * (C / 4) + c / 4), imgq = (unsigned long long)Ho * Wo * (C / 4);
      g = g * drop_scales8(dr, (unsigned long long)no * imgq + pixq, (unsigned long long)(no % npt) * imgq + pixq);
    }
    // the masked pooled gradient itself (stored like dout): what the 2:4-sparse weight gradient builds its operand from instead of reading din
    if (gmask) stv8(gmask + ((no * Ho + yo) * Wo + xo) * dos + c, g);
    if (amax) {   // kernel-uniform
      float m = fabsf(g[0]);
#pragma unroll
      for (int j = 1; j < 8; ++j) m = fmaxf(m, fabsf(g[j]));
      const unsigned bits = __float_as_uint(m);
      vmax = bits > vmax ? bits : vmax;
    }
    f32x8_t g0, g1, g2, g3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // first maximum in row-major window order (the order torch's max_pool2d scans)
      int am = 0; float m = v0[j];
      if (v1[j] > m) { m = v1[j]; am = 1; }
      if (v2[j] > m) { m = v2[j]; am = 2; }
      if (v3[j] > m) { m = v3[j]; am = 3; }
      if (route) am = (rt >> (2 * j)) & 3;
      g0[j] = am == 0 ? g[j] : 0.f; g1[j] = am == 1 ? g[j] : 0.f; g2[j] = am == 2 ? g[j] : 0.f; g3[j] = am == 3 ? g[j] : 0.f;
    }
    TI* d = din + base * dis + c;
    stv8(d, g0); stv8(d + dis, g1); stv8(d + (long long)W * dis, g2); stv8(d + (long long)W * dis + dis, g3);
  }
  if (amax) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)vmax, off);
      vmax = o > vmax ? o : vmax;
    }
    if ((threadIdx.x & 63) == 0 && vmax > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, vmax);
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm: per (group, channel) reductions.  MODE 0: sum x, sum x^2.  MODE 1: sum dy, sum dy*xhat.
// Grid: (chunks, groups).  A block walks its pixel chunk with threads laid out [pixel][channel quad].
// ---------------------------------------------------------------------------------------------
template <int MODE, typename TA>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const TA* __restrict__ x, int xs, const TA* __restrict__ dy, int dys,
                                                        long long pix_per_group, int C, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, double* __restrict__ sums /*[G][2][C]*/) {
  extern __shared__ float red[];  // [2][rows][C]
  const int q = C / 8;  // 8 channels per thread (one 16-byte access per bf16 tensor)
  const int rows = 256 / q > 0 ? 256 / q : 1;  // pixel rows handled concurrently
  const int g = blockIdx.y;
  const int cq = threadIdx.x % q, row = threadIdx.x / q;
  const bool active = row < rows;
  f32x8_t s0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s1 = s0, mu = s0, rs = s0;
  if (MODE == 1 && active) { mu = ldv8(mean + (long long)g * C + cq * 8); rs = ldv8(rstd + (long long)g * C + cq * 8); }
  const long long chunk = (pix_per_group + gridDim.x - 1) / gridDim.x;
  const long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk < pix_per_group ? p0 + chunk : pix_per_group;
  if (active)
    for (long long p = p0 + row; p < p1; p += rows) {
      const long long gp = (long long)g * pix_per_group + p;
      const f32x8_t v = ldv8(x + gp * xs + cq * 8);
      if (MODE == 0) { s0 += v; s1 += v * v; }
      else { const f32x8_t d = ldv8(dy + gp * dys + cq * 8); s0 += d; s1 += d * ((v - mu) * rs); }
    }
  if (active) { stv8(red + (0 * rows + row) * C + cq * 8, s0); stv8(red + (1 * rows + row) * C + cq * 8, s1); }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i % C;
    float s = 0.f;
    for (int r2 = 0; r2 < rows; ++r2) s += red[(which * rows + r2) * C + c];
    atomicAdd(sums + ((long long)g * 2 + which) * C + c, (double)s);
  }
}

// sums[g][2][C] (fp64) from the convolution's per-tile records stats[tile][np][2] (fp32), tiles of group g = [g*tpg, (g+1)*tpg)
// 8 threads share a (group, channel): each walks every 8th tile record (sum and sum of squares in one 8-byte load, 8 loads in
// flight) and the 8 partial sums are combined through LDS in a fixed order.  (One thread per output walking all 384 tiles
// was 48 workgroups of pure load latency: 31 us.)
__global__ __launch_bounds__(256) void bn_sum_tiles_kernel(const float* __restrict__ stats, int tpg, int np, int C, double* __restrict__ sums) {
  __shared__ double red[8][32][2];
  const int lc = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + lc, g = blockIdx.y;
  double a0 = 0, a1 = 0;
  if (c < C) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t* p = reinterpret_cast<const f32x2_t*>(stats + ((size_t)g * tpg * np + c) * 2);
    double s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int t = part;
    for (; t + 56 < tpg; t += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x2_t v = p[(size_t)(t + 8 * u) * np];
        s1[u] += (double)v[0]; s2[u] += (double)v[1];
      }
    }
    for (; t < tpg; t += 8) {
      const f32x2_t v = p[(size_t)t * np];
      s1[0] += (double)v[0]; s2[0] += (double)v[1];
    }
    a0 = ((s1[0] + s1[1]) + (s1[2] + s1[3])) + ((s1[4] + s1[5]) + (s1[6] + s1[7]));
    a1 = ((s2[0] + s2[1]) + (s2[2] + s2[3])) + ((s2[4] + s2[5]) + (s2[6] + s2[7]));
  }
  red[part][lc][0] = a0; red[part][lc][1] = a1;
  __syncthreads();
  if (part < 2 && c < C) {  // part 0 -> sum, part 1 -> sum of squares
    double t = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][lc][part];
    sums[((size_t)g * 2 + part) * C + c] = t;
  }
}

// finalize forward statistics: mean/rstd per (group, channel), affine a = gamma*rstd, b = beta - mean*a;
// running stats updated group after group (the reference calls the module once per lead time, in order).
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int G, int C, int Creal, double count, float eps, float momentum,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ a, float* __restrict__ b,
                                   float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = 0.f, rv = 0.f;
  const bool real = c < Creal;
  if (real && running_mean) { rm = running_mean[c]; rv = running_var[c]; }
  for (int g = 0; g < G; ++g) {
    const double m = sums[((long long)g * 2 + 0) * C + c] / count;
    double var = sums[((long long)g * 2 + 1) * C + c] / count - m * m;
    if (var < 0) var = 0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[(long long)g * C + c] = (float)m;
    rstd[(long long)g * C + c] = r;
    const float ga = real ? gamma[c] : 0.f, be = real ? beta[c] : 0.f;
    a[(long long)g * C + c] = ga * r;
    b[(long long)g * C + c] = be - (float)m * ga * r;
    if (real && running_mean) {
      const double unbiased = count > 1 ? var * count / (count - 1) : var;
      // momentum < 0: cumulative moving average (nn.BatchNorm2d(momentum=None)); -momentum - 1 batches were tracked before group 0
      const float mom = momentum >= 0.f ? momentum : 1.f / (-momentum + (float)g);
      rm = (1.f - mom) * rm + mom * (float)m;
      rv = (1.f - mom) * rv + mom * (float)unbiased;
    }
  }
  if (real && running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

// eval mode: a, b from running statistics (one group)
__global__ void bn_eval_affine_kernel(int C, int Creal, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float* __restrict__ a, float* __restrict__ b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (c < Creal) {
    const float r = 1.f / sqrtf(running_var[c] + eps);
    a[c] = gamma[c] * r; b[c] = beta[c] - running_mean[c] * gamma[c] * r;
  } else { a[c] = 0.f; b[c] = 0.f; }
}

// eval-mode backward: the statistics are constants, so dx = gamma*rstd * dy; mean / rstd feed the MODE 1 reduction
// (dgamma = sum dy*xhat, dbeta = sum dy).  scratch = [mean | rstd | A | B | K] x C.
__global__ void bn_eval_bwd_prep_kernel(int C, int Creal, float eps, const float* __restrict__ gamma, const float* __restrict__ running_mean,
                                        const float* __restrict__ running_var, float* __restrict__ scratch) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const bool real = c < Creal;
  const float r = real ? 1.f / sqrtf(running_var[c] + eps) : 0.f;
  scratch[c] = real ? running_mean[c] : 0.f;
  scratch[C + c] = r;
  scratch[2 * C + c] = real ? gamma[c] * r : 0.f;
  scratch[3 * C + c] = 0.f;
  scratch[4 * C + c] = 0.f;
}

// y = x*a[g] + b[g]
template <typename TA>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TA* __restrict__ x, int xs, long long pixels, long long pix_per_group, int C,
                                                       const float* __restrict__ a, const float* __restrict__ b, TA* __restrict__ y, int ys) {
  const int q = C / 8;  // 8 channels per thread: one 16-byte access per bf16 tensor (C is a multiple of 16)
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long p = idx / q;
    const int c = (idx % q) * 8;
    const long long g = p / pix_per_group;
    stv8(y + p * ys + c, ldv8(x + p * xs + c) * ldv8(a + g * C + c) + ldv8(b + g * C + c));
  }
}

// dx = gamma*rstd * (dy - sum_dy/N - xhat * sum_dyxhat/N)  ==  A*dy + B*x + K  per (group, channel):
//   A = gamma*rstd,  B = -gamma*rstd^2*m2,  K = -gamma*rstd*m1 + gamma*rstd^2*m2*mean,   m1 = sum_dy/N, m2 = sum_dyxhat/N
__global__ void bn_bwd_coef_kernel(const double* __restrict__ sums, int G, int C, int Creal, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ coef /*[G][3][C]*/) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G * C) return;
  const int c = idx % C, g = idx / C;
  float A = 0.f, B = 0.f, K = 0.f;
  if (c < Creal) {
    const double m1 = sums[((size_t)g * 2 + 0) * C + c] / count, m2 = sums[((size_t)g * 2 + 1) * C + c] / count;
    const double ga = gamma[c], r = rstd[(size_t)g * C + c], mu = mean[(size_t)g * C + c];
    A = (float)(ga * r); B = (float)(-ga * r * r * m2); K = (float)(-ga * r * m1 + ga * r * r * m2 * mu);
  }
  coef[((size_t)g * 3 + 0) * C + c] = A; coef[((size_t)g * 3 + 1) * C + c] = B; coef[((size_t)g * 3 + 2) * C + c] = K;
}

template <typename TA>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const TA* __restrict__ x, int xs, const TA* __restrict__ dy, int dys,
                                                           long long pixels, long long pix_per_group, int C, const float* __restrict__ coef,
                                                           TA* __restrict__ dx, int dxs, unsigned* __restrict__ amax) {
  // amax (nullable; fp32 storage): raised to max |dx| by this launch - the SF_F32E kernels' scale word of dx (sfTensor::amax), zeroed by the caller;
  // non-negative floats order like unsigned integers, one atomic per wave at most and only while the word is below the wave's maximum
  const int q = C / 8;
  const long long total = pixels * q;
  unsigned vmax = 0u;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long p = idx / q;
    const int c = (idx % q) * 8;
    const float* k = coef + (p / pix_per_group) * 3 * C + c;
    const auto v = ldv8(k) * ldv8(dy + p * dys + c) + ldv8(k + C) * ldv8(x + p * xs + c) + ldv8(k + 2 * C);
    stv8(dx + p * dxs + c, v);
    if (amax) {   // kernel-uniform
      float m = fabsf(v[0]);
#pragma unroll
      for (int j = 1; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
      const unsigned bits = __float_as_uint(m);
      vmax = bits > vmax ? bits : vmax;
    }
  }
  if (amax) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)vmax, off);
      vmax = o > vmax ? o : vmax;
    }
    if ((threadIdx.x & 63) == 0 && vmax > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, vmax);
  }
}

// dgamma[c] (+)= sum_g sum_dyxhat, dbeta[c] (+)= sum_g sum_dy
__global__ void bn_param_grad_kernel(const double* __restrict__ sums, int G, int C, int Creal, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Creal) return;
  double sb = 0, sg = 0;
  for (int g = 0; g < G; ++g) { sb += sums[((long long)g * 2 + 0) * C + c]; sg += sums[((long long)g * 2 + 1) * C + c]; }
  dgamma[c] = (float)sg; dbeta[c] = (float)sb;
}

int grid_for(long long total) { return (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384); }
bool ok4(const sfTensor& t) {
  return t.ptr == nullptr || ((((uintptr_t)t.ptr) & (t.dtype == SF_BF16 ? 7 : 15)) == 0 && t.stride % 4 == 0 && t.c % 4 == 0 &&
                              (t.dtype == SF_F32 || t.dtype == SF_BF16));
}
// 8-channel accesses of the BatchNorm apply kernels: 16-byte aligned pixels in either storage type
bool ok8(const sfTensor& t) { return ok4(t) && (t.ptr == nullptr || ((((uintptr_t)t.ptr) & 15) == 0 && t.c % 8 == 0 && t.stride % 8 == 0)); }
bool same_dtype(const sfTensor& a, const sfTensor& b) { return a.ptr == nullptr || b.ptr == nullptr || a.dtype == b.dtype; }

}  // namespace

extern "C" {

int sf_metnet_preprocess_fwd(const float* imgs, int32_t B, int32_t T, int32_t C, int32_t sat, int32_t H, int32_t W, int32_t crop,
                             sfTensor out, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_metnet_preprocess_fwd: dtype %d not built", dtype);
  SF_REQUIRE(H % 4 == 0 && W % 4 == 0 && H / 4 == crop && W / 4 == crop,
             "preprocess: raw size %dx%d must be 4x input_size=%d (reference tests/test_models.py:46-49)", H, W, crop);
  SF_REQUIRE(sat >= 0 && sat <= C && out.c >= 8 * sat + (C - sat) && ok4(out), "preprocess: output lanes %d < %d", out.c, 8 * sat + C - sat);
  const long long total = (long long)B * T * crop * crop;
  if (total == 0) return 0;
  SF_DISPATCH_ACT(out.dtype, hipLaunchKernelGGL((preprocess_kernel<TA>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, imgs, B, T,
                                                C, sat, H, W, crop, (TA*)out.ptr, out.c, out.stride));
  SF_CHECK_LAUNCH("metnet_preprocess");
  return 0;
}

int sf_metnet_preprocess_bwd(sfTensor dout, int32_t B, int32_t T, int32_t C, int32_t sat, int32_t H, int32_t W, int32_t crop, float* dimgs,
                             int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_metnet_preprocess_bwd: dtype %d not built", dtype);
  SF_REQUIRE(H % 4 == 0 && W % 4 == 0 && H / 4 == crop && W / 4 == crop, "preprocess bwd: raw size %dx%d must be 4x input_size=%d", H, W, crop);
  SF_REQUIRE(sat >= 0 && sat <= C && dout.c >= 8 * sat + (C - sat) && ok4(dout) && (((uintptr_t)dimgs) & 15) == 0,
             "preprocess bwd: gradient lanes %d < %d or misaligned", dout.c, 8 * sat + C - sat);
  const long long total = (long long)B * T * C * H * (W / 4);
  if (total == 0) return 0;
  SF_DISPATCH_ACT(dout.dtype, hipLaunchKernelGGL((preprocess_bwd_kernel<TA>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                                 (const TA*)dout.ptr, dout.stride, B, T, C, sat, H, W, crop, dimgs));
  SF_CHECK_LAUNCH("metnet_preprocess_bwd");
  return 0;
}

static int maxpool_launch(bool bwd, sfTensor in, sfTensor dout_or_out, int64_t n, int32_t h, int32_t w, sfTensor din, int32_t perm_l, int32_t perm_t,
                          const sfDrop* drop, int64_t period, hipStream_t st, void* route = nullptr, void* gmask = nullptr) {
  sfTensor& out = dout_or_out;
  if (bwd && route && !in.ptr) { in = din; in.ptr = din.ptr; }  // routing recorded: the input tensor is only described (never read)
  SF_REQUIRE(!route || ((uintptr_t)route & 1) == 0, "maxpool2: route must be 2-byte aligned");
  SF_REQUIRE(h % 2 == 0 && w % 2 == 0 && in.c == out.c && ok8(in) && ok8(out) && (!bwd || (in.c == din.c && ok8(din))),
             "maxpool2: needs even H,W and matching channels (multiple of 8, 16-byte aligned pixels)");
  SF_REQUIRE(in.dtype == out.dtype || (in.dtype == SF_BF16 && out.dtype == SF_F32), "maxpool2: unsupported storage pair %d -> %d", in.dtype, out.dtype);
  SF_REQUIRE(!bwd || in.dtype == din.dtype, "maxpool2 bwd: din must be stored like the input");
  SF_REQUIRE(!bwd || !din.amax || (din.dtype == SF_F32 && ((uintptr_t)din.amax & 3) == 0), "maxpool2 bwd: din.amax goes with an fp32-stored din (4-byte aligned word)");
  OuterPerm pm{0, 0, 0};
  if (perm_l > 0) {
    SF_REQUIRE(perm_t > 0 && n % ((long long)perm_l * perm_t) == 0, "maxpool2: n=%lld not divisible by perm dims %d x %d", (long long)n, perm_l, perm_t);
    pm = OuterPerm{perm_l, perm_t, (int)(n / ((long long)perm_l * perm_t))};
  }
  const long long total = n * (h / 2) * (w / 2) * (in.c / 8);
  if (total == 0) return 0;
  sfDrop dr{};
  long long npt = 1;
  if (drop) {
    dr = *drop;
    const long long img = (long long)(h / 2) * (w / 2) * out.c;
    SF_REQUIRE(out.stride == out.c && period > 0 && period % img == 0, "maxpool2+dropout: the pooled tensor must be contiguous and the period (%lld) a whole number of images", (long long)period);
    npt = period / img;
  }
  const dim3 grid(grid_for(total)), block(256);
#define SF_MP(TI_, TO_, DROP_)                                                                                                                   \
  do {                                                                                                                                            \
    if (bwd) hipLaunchKernelGGL((maxpool_bwd_kernel<TI_, TO_, DROP_>), grid, block, 0, st, (const TI_*)in.ptr, in.stride, (const TO_*)out.ptr,   \
                                out.stride, (long long)n, h, w, in.c, (TI_*)din.ptr, din.stride, pm, dr, npt, (const unsigned short*)route,   \
                                (TO_*)gmask, (unsigned*)din.amax);                                                                               \
    else hipLaunchKernelGGL((maxpool_fwd_kernel<TI_, TO_, DROP_>), grid, block, 0, st, (const TI_*)in.ptr, in.stride, (long long)n, h, w, in.c,   \
                            (TO_*)out.ptr, out.stride, pm, dr, npt, (unsigned short*)route);                                                     \
  } while (0)
#define SF_MP_T(DROP_)                                         \
  do {                                                         \
    if (in.dtype == SF_F32) SF_MP(float, float, DROP_);        \
    else if (out.dtype == SF_BF16) SF_MP(__bf16, __bf16, DROP_); \
    else SF_MP(__bf16, float, DROP_);                          \
  } while (0)
  if (drop) SF_MP_T(true); else SF_MP_T(false);
#undef SF_MP_T
#undef SF_MP
  SF_CHECK_LAUNCH(bwd ? "maxpool2_bwd" : "maxpool2_fwd");
  return 0;
}

int sf_maxpool2_fwd(sfTensor in, int64_t n, int32_t h, int32_t w, sfTensor out, int32_t perm_l, int32_t perm_t, int32_t dtype,
                    sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_maxpool2_fwd: dtype %d not built", dtype);
  return maxpool_launch(false, in, out, n, h, w, sfTensor{}, perm_l, perm_t, nullptr, 0, (hipStream_t)stream);
}

int sf_maxpool2_bwd(sfTensor in, sfTensor dout, int64_t n, int32_t h, int32_t w, sfTensor din, int32_t perm_l, int32_t perm_t,
                    int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_maxpool2_bwd: dtype %d not built", dtype);
  return maxpool_launch(true, in, dout, n, h, w, din, perm_l, perm_t, nullptr, 0, (hipStream_t)stream);
}

static int check_drop(float p1, float p2) {
  SF_REQUIRE(p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f, "maxpool2+dropout: p1=%f p2=%f", p1, p2);
  return 0;
}

int sf_maxpool2_dropout_fwd(sfTensor in, int64_t n, int32_t h, int32_t w, sfTensor out, int32_t perm_l, int32_t perm_t, float p1, float p2,
                            int64_t period, uint64_t seed1, uint64_t seed2, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_maxpool2_dropout_fwd: dtype %d not built", dtype);
  if (int rc = check_drop(p1, p2)) return rc;
  const sfDrop d = sf_make_drop(p1, p2, seed1, seed2);
  return maxpool_launch(false, in, out, n, h, w, sfTensor{}, perm_l, perm_t, &d, period, (hipStream_t)stream);
}

int sf_maxpool2_dropout_bwd(sfTensor in, sfTensor dout, int64_t n, int32_t h, int32_t w, sfTensor din, int32_t perm_l, int32_t perm_t,
                            float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_maxpool2_dropout_bwd: dtype %d not built", dtype);
  if (int rc = check_drop(p1, p2)) return rc;
  const sfDrop d = sf_make_drop(p1, p2, seed1, seed2);
  return maxpool_launch(true, in, dout, n, h, w, din, perm_l, perm_t, &d, period, (hipStream_t)stream);
}

int sf_maxpool2_route_fwd(sfTensor in, int64_t n, int32_t h, int32_t w, sfTensor out, int32_t perm_l, int32_t perm_t, float p1, float p2,
                          int64_t period, uint64_t seed1, uint64_t seed2, void* route, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_maxpool2_route_fwd: dtype %d not built", dtype);
  SF_REQUIRE(route != nullptr, "sf_maxpool2_route_fwd: route must not be null");
  if (p1 == 0.f && p2 == 0.f) return maxpool_launch(false, in, out, n, h, w, sfTensor{}, perm_l, perm_t, nullptr, 0, (hipStream_t)stream, route);
  if (int rc = check_drop(p1, p2)) return rc;
  const sfDrop d = sf_make_drop(p1, p2, seed1, seed2);
  return maxpool_launch(false, in, out, n, h, w, sfTensor{}, perm_l, perm_t, &d, period, (hipStream_t)stream, route);
}

int sf_maxpool2_route_bwd(const void* route, sfTensor dout, int64_t n, int32_t h, int32_t w, sfTensor din, int32_t perm_l, int32_t perm_t, float p1,
                          float p2, int64_t period, uint64_t seed1, uint64_t seed2, void* masked_dout, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_maxpool2_route_bwd: dtype %d not built", dtype);
  SF_REQUIRE(route != nullptr && din.ptr != nullptr, "sf_maxpool2_route_bwd: route / din must not be null");
  SF_REQUIRE(!masked_dout || ((uintptr_t)masked_dout & 15) == 0, "sf_maxpool2_route_bwd: masked_dout must be 16-byte aligned");
  sfTensor none{};
  if (p1 == 0.f && p2 == 0.f) return maxpool_launch(true, none, dout, n, h, w, din, perm_l, perm_t, nullptr, 0, (hipStream_t)stream, (void*)route, masked_dout);
  if (int rc = check_drop(p1, p2)) return rc;
  const sfDrop d = sf_make_drop(p1, p2, seed1, seed2);
  return maxpool_launch(true, none, dout, n, h, w, din, perm_l, perm_t, &d, period, (hipStream_t)stream, (void*)route, masked_dout);
}

static int bn_reduce_launch(int mode, sfTensor x, sfTensor dy, int64_t pix_per_group, int32_t groups, const float* mean,
                            const float* rstd, double* sums, hipStream_t st) {
  const int C = x.c;
  SF_REQUIRE(C % 8 == 0 && C <= 2048 && ok8(x) && ok8(dy) && same_dtype(x, dy), "batchnorm: channels %d (multiple of 8, 16-byte aligned) / storage types", C);
  hipError_t e = sf_fill_async(sums, 0, sizeof(double) * 2 * C * groups, st);
  SF_REQUIRE(e == hipSuccess, "batchnorm: memset failed");
  const int q = C / 8, rows = 256 / q > 0 ? 256 / q : 1;
  long long chunks = pix_per_group / (rows * 16);
  if (chunks < 1) chunks = 1;
  if (chunks > 256) chunks = 256;
  const size_t shmem = sizeof(float) * 2 * rows * C;
  dim3 grid((unsigned)chunks, groups);
  if (mode == 0)
    SF_DISPATCH_ACT(x.dtype, hipLaunchKernelGGL((bn_reduce_kernel<0, TA>), grid, dim3(256), shmem, st, (const TA*)x.ptr, x.stride, (const TA*)nullptr, 0,
                                                (long long)pix_per_group, C, (const float*)nullptr, (const float*)nullptr, sums));
  else
    SF_DISPATCH_ACT(x.dtype, hipLaunchKernelGGL((bn_reduce_kernel<1, TA>), grid, dim3(256), shmem, st, (const TA*)x.ptr, x.stride, (const TA*)dy.ptr,
                                                dy.stride, (long long)pix_per_group, C, mean, rstd, sums));
  SF_CHECK_LAUNCH("bn_reduce");
  return 0;
}

static int bn_train_fwd_impl(sfTensor x, int64_t pix_per_group, int32_t groups, int32_t creal, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                             float* scale, float* shift, double* sums, const float* stats, int32_t tiles_per_group, int32_t stats_np,
                             sfTensor y, hipStream_t st) {
  // y.ptr == null: statistics, scale / shift and the running-stat update only (the apply is folded into the consuming convolution,
  // sf_conv3x3_fwd_folded)
  SF_REQUIRE(ok8(x) && creal <= x.c && (!y.ptr || (x.c == y.c && ok8(y) && x.dtype == y.dtype)), "batchnorm: channels (multiple of 8, 16-byte aligned) / storage type");
  if (stats) {
    SF_REQUIRE(tiles_per_group > 0 && stats_np >= x.c, "batchnorm: tiles_per_group=%d stats_np=%d", tiles_per_group, stats_np);
    hipLaunchKernelGGL(bn_sum_tiles_kernel, dim3((x.c + 31) / 32, groups), dim3(256), 0, st, stats, tiles_per_group, stats_np, x.c, sums);
    SF_CHECK_LAUNCH("bn_sum_tiles");
  } else {
    sfTensor none{nullptr, 0, 0, 0, 0, 0};
    if (int rc = bn_reduce_launch(0, x, none, pix_per_group, groups, nullptr, nullptr, sums, st)) return rc;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((x.c + 127) / 128), dim3(128), 0, st, sums, groups, x.c, creal, (double)pix_per_group, eps,
                     momentum, gamma, beta, mean, rstd, scale, shift, running_mean, running_var);
  SF_CHECK_LAUNCH("bn_finalize");
  if (!y.ptr) return 0;
  const long long pixels = pix_per_group * groups;
  SF_DISPATCH_ACT(x.dtype, hipLaunchKernelGGL((bn_apply_kernel<TA>), dim3(grid_for(pixels * (x.c / 8))), dim3(256), 0, st, (const TA*)x.ptr, x.stride,
                                              pixels, (long long)pix_per_group, x.c, (const float*)scale, (const float*)shift, (TA*)y.ptr, y.stride));
  SF_CHECK_LAUNCH("bn_apply");
  return 0;
}

int sf_batchnorm_train_fwd(sfTensor x, int64_t pix_per_group, int32_t groups, int32_t creal, const float* gamma, const float* beta,
                           float eps, float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                           float* scale, float* shift, double* sums, sfTensor y, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_batchnorm_train_fwd: dtype %d not built", dtype);
  return bn_train_fwd_impl(x, pix_per_group, groups, creal, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, sums,
                           nullptr, 0, 0, y, (hipStream_t)stream);
}

int sf_batchnorm_train_fwd_stats(sfTensor x, int64_t pix_per_group, int32_t groups, int32_t creal, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                                 float* scale, float* shift, double* sums, const float* stats, int32_t tiles_per_group, int32_t stats_np,
                                 sfTensor y, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_batchnorm_train_fwd_stats: dtype %d not built", dtype);
  SF_REQUIRE(stats != nullptr, "sf_batchnorm_train_fwd_stats: stats must not be null");
  return bn_train_fwd_impl(x, pix_per_group, groups, creal, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, sums,
                           stats, tiles_per_group, stats_np, y, (hipStream_t)stream);
}

int sf_batchnorm_eval_fwd(sfTensor x, int64_t pixels, int32_t creal, const float* gamma, const float* beta, float eps,
                          const float* running_mean, const float* running_var, float* scale, float* shift, sfTensor y,
                          int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_batchnorm_eval_fwd: dtype %d not built", dtype);
  SF_REQUIRE(x.c == y.c && ok8(x) && ok8(y) && creal <= x.c && x.dtype == y.dtype, "batchnorm eval: channels (multiple of 8, 16-byte aligned) / storage type");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((x.c + 127) / 128), dim3(128), 0, st, x.c, creal, eps, gamma, beta, running_mean,
                     running_var, scale, shift);
  SF_CHECK_LAUNCH("bn_eval_affine");
  SF_DISPATCH_ACT(x.dtype, hipLaunchKernelGGL((bn_apply_kernel<TA>), dim3(grid_for(pixels * (x.c / 8))), dim3(256), 0, st, (const TA*)x.ptr, x.stride,
                                              (long long)pixels, (long long)pixels, x.c, (const float*)scale, (const float*)shift, (TA*)y.ptr, y.stride));
  SF_CHECK_LAUNCH("bn_apply");
  return 0;
}

int sf_batchnorm_eval_bwd(sfTensor x, sfTensor dy, int64_t pixels, int32_t creal, const float* gamma, float eps, const float* running_mean,
                          const float* running_var, double* sums, float* scratch, sfTensor dx, float* dgamma, float* dbeta, int32_t dtype,
                          sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_batchnorm_eval_bwd: dtype %d not built", dtype);
  SF_REQUIRE(x.c == dy.c && x.c == dx.c && ok8(x) && ok8(dy) && ok8(dx) && x.dtype == dx.dtype && creal <= x.c,
             "batchnorm eval bwd: channels (multiple of 8, 16-byte aligned) / storage type");
  SF_REQUIRE(!dx.amax || (dx.dtype == SF_F32 && ((uintptr_t)dx.amax & 3) == 0), "batchnorm eval bwd: dx.amax goes with an fp32-stored dx (4-byte aligned word)");
  hipStream_t st = (hipStream_t)stream;
  const int C = x.c;
  hipLaunchKernelGGL(bn_eval_bwd_prep_kernel, dim3((C + 127) / 128), dim3(128), 0, st, C, creal, eps, gamma, running_mean, running_var, scratch);
  SF_CHECK_LAUNCH("bn_eval_bwd_prep");
  if (int rc = bn_reduce_launch(1, x, dy, pixels, 1, scratch, scratch + C, sums, st)) return rc;
  SF_DISPATCH_ACT(x.dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<TA>), dim3(grid_for(pixels * (C / 8))), dim3(256), 0, st, (const TA*)x.ptr, x.stride,
                                              (const TA*)dy.ptr, dy.stride, (long long)pixels, (long long)pixels, C, (const float*)(scratch + 2 * C),
                                              (TA*)dx.ptr, dx.stride, (unsigned*)dx.amax));
  SF_CHECK_LAUNCH("bn_bwd_apply");
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((creal + 127) / 128), dim3(128), 0, st, sums, 1, C, creal, dgamma, dbeta);
  SF_CHECK_LAUNCH("bn_param_grad");
  return 0;
}

int sf_batchnorm_train_bwd(sfTensor x, sfTensor dy, int64_t pix_per_group, int32_t groups, int32_t creal, const float* gamma,
                           const float* mean, const float* rstd, double* sums, float* coef, sfTensor dx, float* dgamma, float* dbeta,
                           int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_batchnorm_train_bwd: dtype %d not built", dtype);
  SF_REQUIRE(x.c == dy.c && x.c == dx.c && ok8(x) && ok8(dy) && ok8(dx) && x.dtype == dx.dtype, "batchnorm bwd: channels (multiple of 8, 16-byte aligned) / storage type");
  SF_REQUIRE(!dx.amax || (dx.dtype == SF_F32 && ((uintptr_t)dx.amax & 3) == 0), "batchnorm bwd: dx.amax goes with an fp32-stored dx (4-byte aligned word)");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = bn_reduce_launch(1, x, dy, pix_per_group, groups, mean, rstd, sums, st)) return rc;
  const long long pixels = pix_per_group * groups;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((groups * x.c + 255) / 256), dim3(256), 0, st, sums, groups, x.c, creal, (double)pix_per_group,
                     gamma, mean, rstd, coef);
  SF_CHECK_LAUNCH("bn_bwd_coef");
  SF_DISPATCH_ACT(x.dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<TA>), dim3(grid_for(pixels * (x.c / 8))), dim3(256), 0, st, (const TA*)x.ptr, x.stride,
                                              (const TA*)dy.ptr, dy.stride, pixels, (long long)pix_per_group, x.c, (const float*)coef, (TA*)dx.ptr,
                                              dx.stride, (unsigned*)dx.amax));
  SF_CHECK_LAUNCH("bn_bwd_apply");
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((creal + 127) / 128), dim3(128), 0, st, sums, groups, x.c, creal, dgamma, dbeta);
  SF_CHECK_LAUNCH("bn_param_grad");
  return 0;
}

int sf_batchnorm_train_bwd_coef(const double* sums, int64_t pix_per_group, int32_t groups, int32_t c, int32_t creal, const float* gamma,
                                const float* mean, const float* rstd, float* coef, float* dgamma, float* dbeta, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_batchnorm_train_bwd_coef: dtype %d not built", dtype);
  SF_REQUIRE(sums && gamma && mean && rstd && coef && dgamma && dbeta && groups >= 1 && creal <= c && c % 8 == 0 && pix_per_group > 0,
             "batchnorm bwd coef: null argument / groups=%d c=%d creal=%d", groups, c, creal);
  SF_REQUIRE(((uintptr_t)coef & 15) == 0, "batchnorm bwd coef: coef must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((groups * c + 255) / 256), dim3(256), 0, st, sums, groups, c, creal, (double)pix_per_group, gamma, mean,
                     rstd, coef);
  SF_CHECK_LAUNCH("bn_bwd_coef");
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((creal + 127) / 128), dim3(128), 0, st, sums, groups, c, creal, dgamma, dbeta);
  SF_CHECK_LAUNCH("bn_param_grad");
  return 0;
}

}  // extern "C"

// =============================================================================================
// Lead-time de-duplication of MetNet's first convolution.
//
// conv1 is linear, and the ConditionTime planes are constant one-hot images, so for lead time l
//   conv1([frame ; onehot_l]) = conv1_image(frame) + b + P_l,
//   P_l[y][x][co] = sum of the taps of W1[co][Cimg + l] whose input pixel lies inside the image,
// which takes only 9 values per (l, co): one per border class (top/interior/bottom x left/interior/right).
// The image part ("base") is computed ONCE per frame instead of once per (frame, lead time); the kernels
// below fuse "+ P_l" with the DownSampler's first 2x2 max-pool, forward and backward, so the per-lead-time
// full-resolution tensor is never materialised.
// =============================================================================================
namespace {

__device__ __forceinline__ int border_class(int y, int x, int H, int W) {
  const int ry = y == 0 ? 0 : (y == H - 1 ? 2 : 1), rx = x == 0 ? 0 : (x == W - 1 ? 2 : 1);
  return ry * 3 + rx;
}

// ptab[l][cls][c] from w1 [O][I][3][3]; columns Cimg + l.  Pad lanes c >= O are zero.
__global__ void leadbias_table_kernel(const float* __restrict__ w1, int O, int I, int cimg, int L, int Cp, float* __restrict__ ptab) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L * 9 * Cp) return;
  const int c = idx % Cp, cls = (idx / Cp) % 9, l = idx / (9 * Cp);
  float s = 0.f;
  if (c < O) {
    const int ry = cls / 3, rx = cls % 3;
    const float* w = w1 + ((size_t)c * I + cimg + l) * 9;
    for (int ky = 0; ky < 3; ++ky) {
      if ((ry == 0 && ky == 0) || (ry == 2 && ky == 2)) continue;  // input row y+ky-1 outside the image
      for (int kx = 0; kx < 3; ++kx) {
        if ((rx == 0 && kx == 0) || (rx == 2 && kx == 2)) continue;
        s += w[ky * 3 + kx];
      }
    }
  }
  ptab[idx] = s;
}

// pooled[(l*F + f)][yo][xo][c] = max over the 2x2 window of base[f] + ptab[l][class]
template <typename TA, bool PLDS>  // PLDS: border table staged in LDS (no global loads inside the lead-time loop, only stores)
__global__ __launch_bounds__(1024) void leadbias_pool_fwd_kernel(const TA* __restrict__ base, int bs, long long F, int H, int W, int C,
                                                                 int L, const float* __restrict__ ptab_g, TA* __restrict__ out, int os) {
  extern __shared__ float P[];
  if (PLDS) {
    for (int i = threadIdx.x; i < L * 9 * C; i += blockDim.x) P[i] = ptab_g[i];
    __syncthreads();
  }
  const float* ptab = PLDS ? P : ptab_g;
  const int Ho = H / 2, Wo = W / 2, q = C / 4;
  const long long total = F * Ho * Wo * q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (idx % q) * 4;
    const long long op = idx / q;
    const int xo = op % Wo, yo = (op / Wo) % Ho;
    const long long f = op / ((long long)Wo * Ho);
    const TA* p = base + ((f * H + 2 * yo) * W + 2 * xo) * bs + c;
    const f32x4 v0 = ldv4(p), v1 = ldv4(p + bs), v2 = ldv4(p + (long long)W * bs), v3 = ldv4(p + (long long)W * bs + bs);
    const int k0 = border_class(2 * yo, 2 * xo, H, W), k1 = border_class(2 * yo, 2 * xo + 1, H, W);
    const int k2 = border_class(2 * yo + 1, 2 * xo, H, W), k3 = border_class(2 * yo + 1, 2 * xo + 1, H, W);
#pragma unroll 4
    for (int l = 0; l < L; ++l) {
      const float* pt = ptab + (size_t)l * 9 * C + c;
      const f32x4 a0 = v0 + ld4(pt + k0 * C), a1 = v1 + ld4(pt + k1 * C), a2 = v2 + ld4(pt + k2 * C), a3 = v3 + ld4(pt + k3 * C);
      f32x4 m;
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a0[j], a1[j]), fmaxf(a2[j], a3[j]));
      stv4(out + (((l * F + f) * Ho + yo) * Wo + xo) * os + c, m);
    }
  }
}

// The same pooling that also emits what the BatchNorm behind it needs: per workgroup and lead time the sum and the sum of squares of
// the STORED outputs, stats[(l * gridDim.x + block)][C][2] - the layout sf_batchnorm_train_fwd_stats reads (group l owns gridDim.x
// "tiles").  A thread owns ONE channel quad for the whole kernel (its sums stay in registers, L <= LEAD_REG) and walks pooling
// windows; the window lanes of a workgroup are combined through LDS in lane order (deterministic).
constexpr int LEAD_REG_FWD = 12;
template <typename TA>
__global__ __launch_bounds__(512) void leadbias_pool_fwd_stats_kernel(const TA* __restrict__ base, int bs, long long F, int H, int W, int C, int L,
                                                                     const float* __restrict__ ptab_g, TA* __restrict__ out, int os,
                                                                     float* __restrict__ stats) {
  extern __shared__ float P[];  // [L][9][C] table | [L][C][2] sums
  const int nP = L * 9 * C;
  float* S = P + nP;
  for (int i = threadIdx.x; i < nP; i += blockDim.x) P[i] = ptab_g[i];
  for (int i = threadIdx.x; i < L * C * 2; i += blockDim.x) S[i] = 0.f;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2, q = C / 4;
  const int lanes = blockDim.x / q;
  const int cq = threadIdx.x % q, wl = threadIdx.x / q;
  const int c = cq * 4;
  const long long windows = F * Ho * Wo;
  f32x4 s1[LEAD_REG_FWD], s2[LEAD_REG_FWD];
#pragma unroll
  for (int l = 0; l < LEAD_REG_FWD; ++l) { s1[l] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[l] = s1[l]; }
  if (wl < lanes)
    for (long long op = (long long)blockIdx.x * lanes + wl; op < windows; op += (long long)gridDim.x * lanes) {
      const int xo = op % Wo, yo = (op / Wo) % Ho;
      const long long f = op / ((long long)Wo * Ho);
      const TA* p = base + ((f * H + 2 * yo) * W + 2 * xo) * bs + c;
      const f32x4 v0 = ldv4(p), v1 = ldv4(p + bs), v2 = ldv4(p + (long long)W * bs), v3 = ldv4(p + (long long)W * bs + bs);
      const int k0 = border_class(2 * yo, 2 * xo, H, W), k1 = border_class(2 * yo, 2 * xo + 1, H, W);
      const int k2 = border_class(2 * yo + 1, 2 * xo, H, W), k3 = border_class(2 * yo + 1, 2 * xo + 1, H, W);
#pragma unroll
      for (int l = 0; l < LEAD_REG_FWD; ++l) {
        if (l < L) {  // block-uniform
          const float* pt = P + (size_t)l * 9 * C + c;
          const f32x4 a0 = v0 + ld4(pt + k0 * C), a1 = v1 + ld4(pt + k1 * C), a2 = v2 + ld4(pt + k2 * C), a3 = v3 + ld4(pt + k3 * C);
          f32x4 m;
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a0[j], a1[j]), fmaxf(a2[j], a3[j]));
          TA* o = out + (((l * F + f) * Ho + yo) * Wo + xo) * os + c;
          stv4(o, m);
          const f32x4 r = rnd4<TA>(m);  // the stored values
          s1[l] += r; s2[l] += r * r;
        }
      }
    }
  // window lanes in order: lane w adds its sums in round w
  for (int w = 0; w < lanes; ++w) {
    if (wl == w) {
#pragma unroll
      for (int l = 0; l < LEAD_REG_FWD; ++l)
        if (l < L) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { S[((size_t)l * C + c + j) * 2] += s1[l][j]; S[((size_t)l * C + c + j) * 2 + 1] += s2[l][j]; }
        }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < L * C * 2; i += blockDim.x) {
    const int l = i / (C * 2), rest = i - l * C * 2;
    stats[((size_t)l * gridDim.x + blockIdx.x) * C * 2 + rest] = S[i];
  }
}

// dbase[f] = sum_l unpool(dpooled[l*F+f]);  class sums for the one-hot weight columns by a 2-stage reduce.
// Two kernels:
//  * the MAIN kernel (persistent grid) writes dbase for every window.  A thread owns ONE channel quad for the whole
//    kernel and walks pooling windows; its class sums cover the INTERIOR class only (class 4: ~90 % of all windows, all
//    four positions of the window share it), kept in registers for the first LEAD_REG lead times and flushed to LDS
//    once at the end - the loop body has no divergent branch and no atomics.  Slab: main_part[block][L][C];
//  * the BORDER kernel revisits only the windows that touch the image border, recomputes their argmax and adds the
//    gradient to the class of the winning position with LDS atomics (ds_add_f32).  Slab: border_part[block][L][9][C].
// Both request all lead times' gradients before using the first (12 independent HBM streams; a load under `l < L` is
// emitted as load + vmcnt(0) one by one, hence the clamped unconditional form) and read the border table from LDS
// when it fits next to the sums.
constexpr int LEAD_REG = 12;
constexpr int LEADBIAS_BLOCKS = 512, LEADBIAS_BORDER_BLOCKS = 256;  // border: 4 x 56 edge workgroups + 32 corner workgroups, one per CU (138 KB of LDS each at L = 12)

// border window r (0 .. 2 Wo + 2 (Ho - 2) - 1) of a frame: top row, bottom row, then left / right of the rows between (Ho, Wo >= 3)
__device__ __forceinline__ void border_window(int r, int Ho, int Wo, int& yo, int& xo) {
  if (r < Wo) { yo = 0; xo = r; }
  else if (r < 2 * Wo) { yo = Ho - 1; xo = r - Wo; }
  else { r -= 2 * Wo; yo = 1 + (r >> 1); xo = (r & 1) ? Wo - 1 : 0; }
}

__device__ __forceinline__ int argmax4(float a0, float a1, float a2, float a3) {
  int am = 0; float m = a0;  // first maximum in row-major window order, as max_pool2d
  if (a1 > m) { m = a1; am = 1; }
  if (a2 > m) { m = a2; am = 2; }
  if (a3 > m) { m = a3; am = 3; }
  return am;
}

template <typename TA, bool PLDS>
__global__ __launch_bounds__(512) void leadbias_pool_bwd_kernel(const TA* __restrict__ base, int bs, const TA* __restrict__ dout, int dos,
                                                               long long F, int H, int W, int C, int L, const float* __restrict__ ptab_g,
                                                               TA* __restrict__ dbase, int dbs, float* __restrict__ main_part) {
  extern __shared__ float S[];  // [L][C] interior-class sums (+ [L][9][C] table copy)
  const int nS = L * C, nP = L * 9 * C;
  for (int i = threadIdx.x; i < nS; i += blockDim.x) S[i] = 0.f;
  if (PLDS)
    for (int i = threadIdx.x; i < nP; i += blockDim.x) S[nS + i] = ptab_g[i];
  const float* ptab = PLDS ? S + nS : ptab_g;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2, q = C / 4;
  const int lanes = blockDim.x / q;  // windows in flight per block
  const int cq = threadIdx.x % q, wl = threadIdx.x / q;
  const int c = cq * 4;
  const long long windows = F * Ho * Wo;
  f32x4 s4[LEAD_REG];
#pragma unroll
  for (int l = 0; l < LEAD_REG; ++l) s4[l] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wl < lanes)
    for (long long op = (long long)blockIdx.x * lanes + wl; op < windows; op += (long long)gridDim.x * lanes) {
      const int xo = op % Wo, yo = (op / Wo) % Ho;
      const long long f = op / ((long long)Wo * Ho);
      const long long b0 = (f * H + 2 * yo) * W + 2 * xo;
      const TA* p = base + b0 * bs + c;
      const f32x4 v0 = ldv4(p), v1 = ldv4(p + bs), v2 = ldv4(p + (long long)W * bs), v3 = ldv4(p + (long long)W * bs + bs);
      const int k0 = border_class(2 * yo, 2 * xo, H, W), k1 = border_class(2 * yo, 2 * xo + 1, H, W);
      const int k2 = border_class(2 * yo + 1, 2 * xo, H, W), k3 = border_class(2 * yo + 1, 2 * xo + 1, H, W);
      const float interior = (k0 == 4 && k3 == 4) ? 1.f : 0.f;
      f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
      auto lead = [&](int l, f32x4 g) -> f32x4 {
        const float* pt = ptab + (size_t)l * 9 * C + c;
        const f32x4 a0 = v0 + ld4(pt + k0 * C), a1 = v1 + ld4(pt + k1 * C), a2 = v2 + ld4(pt + k2 * C), a3 = v3 + ld4(pt + k3 * C);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int am = argmax4(a0[j], a1[j], a2[j], a3[j]);
          g0[j] += am == 0 ? g[j] : 0.f; g1[j] += am == 1 ? g[j] : 0.f; g2[j] += am == 2 ? g[j] : 0.f; g3[j] += am == 3 ? g[j] : 0.f;
        }
        return g * interior;
      };
      f32x4 gl[LEAD_REG];
#pragma unroll
      for (int l = 0; l < LEAD_REG; ++l) gl[l] = ldv4(dout + ((((l < L ? l : 0) * F + f) * Ho + yo) * Wo + xo) * dos + c);
#pragma unroll
      for (int l = 0; l < LEAD_REG; ++l) s4[l] += lead(l < L ? l : 0, l < L ? gl[l] : f32x4{0.f, 0.f, 0.f, 0.f});
      for (int l = LEAD_REG; l < L; ++l) {
        const f32x4 gi = lead(l, ldv4(dout + (((l * F + f) * Ho + yo) * Wo + xo) * dos + c));
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(S + (size_t)l * C + c + j, gi[j]);
      }
      TA* d = dbase + b0 * dbs + c;
      stv4(d, g0); stv4(d + dbs, g1); stv4(d + (long long)W * dbs, g2); stv4(d + (long long)W * dbs + dbs, g3);
    }
#pragma unroll
  for (int l = 0; l < LEAD_REG; ++l)
    if (l < L && wl < lanes) {
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(S + (size_t)l * C + c + j, s4[l][j]);
    }
  __syncthreads();
  float* dstp = main_part + (size_t)blockIdx.x * nS;
  for (int i = threadIdx.x; i < nS; i += blockDim.x) dstp[i] = S[i];
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same two kernels with the INTERIOR windows (every position of the 2x2 window has border class 4: all but the outermost ring of
// pooled pixels, 88 % at 32x32) on a short path.  rocprofv3 counters of the kernels above (profiles/r04_leadbias_pmc_before.json) show them bound
// by vector ALU work, not by HBM: 27k / 50k VALU instructions per wave, the VALU of a SIMD busy 70 % / 80 % of the kernel - per window, lead time
// and channel 4 adds + 3 max (+ rounding, statistics) forward; 4 adds + an argmax + 4 routed accumulations backward.
//  * Forward: for an interior window all four positions take the SAME constant b_l, and x -> fl(x + b) is monotone, so
//        max_i fl(v_i + b_l) = fl(max_i v_i + b_l)    EXACTLY:
//    three max per channel ONCE, then one add per lead time.
//  * Backward: the winner of lead time l is the first maximum of fl(v_i + b_l).  It is the first maximum of v itself for EVERY l unless two
//    different v_i collide after the addition.  With m = max v and s = the largest v_i < m: fl(m + b) > fl(s + b) whenever m - s exceeds one
//    ulp at the magnitude of the sums, which (m - s) > 2^-21 (max(|m|, |s|) + max_l |b_l|) guarantees with a factor two to spare.  One
//    argmax per channel, the per-lead-time work is `G += g_l` (and the interior class sum); a wave in which ANY lane fails the test (needs
//    |v| < ~1e-5 |b|: a fraction of a percent of the waves) takes the general path for that window - a wave-uniform branch, no divergence.
// Border windows keep the general arithmetic: forward in a second loop of the same kernel; backward their input gradient is written by the border
// kernel, which visits them anyway for the class sums (inlined next to the short path the general body made hipcc spill 400-500 registers).
// A thread owns a channel OCTET (16-byte accesses for bf16 storage) instead of a quad; the forward kernel splits the lead times over two
// threads per octet so that its 2 x 6 x 8 statistics sums fit the registers.  Same results as the kernels above, bit for bit.
// ---------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x8_t ld8f(const float* p) { const f32x4 a = ld4(p), b = ld4(p + 4); return f32x8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
template <typename T> __device__ __forceinline__ f32x8_t rnd8(f32x8_t v);
template <> __device__ __forceinline__ f32x8_t rnd8<float>(f32x8_t v) { return v; }
template <> __device__ __forceinline__ f32x8_t rnd8<__bf16>(f32x8_t v) { return __builtin_convertvector(__builtin_convertvector(v, bf16x8_t), f32x8_t); }
__device__ __forceinline__ f32x8_t max8(f32x8_t a, f32x8_t b) {
  f32x8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = fmaxf(a[j], b[j]);
  return r;
}

constexpr int LB_LH = 6;  // lead times per thread of the forward kernel (two threads per channel octet)
template <typename TA>
__global__ __launch_bounds__(512) void leadbias_pool_fwd_stats2_kernel(const TA* __restrict__ base, int bs, long long F, int H, int W, int C, int L,
                                                                      const float* __restrict__ ptab_g, TA* __restrict__ out, int os,
                                                                      float* __restrict__ stats) {
  extern __shared__ float P[];  // [L][9][C] table | [L][C][2] sums
  const int nP = L * 9 * C;
  float* S = P + nP;
  for (int i = threadIdx.x; i < nP; i += blockDim.x) P[i] = ptab_g[i];
  for (int i = threadIdx.x; i < L * C * 2; i += blockDim.x) S[i] = 0.f;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2, o8 = C / 8, slots = 2 * o8;
  const int lanes = blockDim.x / slots;
  const int slot = threadIdx.x % slots, wl = threadIdx.x / slots;
  const int hh = slot / o8, c = (slot - hh * o8) * 8, l0 = hh * LB_LH;
  f32x8_t s1[LB_LH], s2[LB_LH];
#pragma unroll
  for (int i = 0; i < LB_LH; ++i) { s1[i] = f32x8_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; s2[i] = s1[i]; }
  auto emit = [&](int i, long long f, int yo, int xo, f32x8_t a) __attribute__((always_inline)) {
    stv8(out + ((((long long)(l0 + i) * F + f) * Ho + yo) * Wo + xo) * os + c, a);
    const f32x8_t r = rnd8<TA>(a);  // the stored values
    s1[i] += r; s2[i] += r * r;
  };
  if (wl < lanes) {
    // ---- interior windows ----
    const int Hi = Ho - 2, Wi = Wo - 2;
    const long long nint = F * Hi * Wi;
    // (window indices fit 32 bits - the launcher checks -: a 64-bit division is ~130 instructions with a branch, and three of them per window were a
    // third of this loop's instructions)
    for (unsigned op = blockIdx.x * lanes + wl; op < (unsigned)nint; op += gridDim.x * lanes) {
      const unsigned row = op / (unsigned)Wi;
      const int xo = 1 + (int)(op - row * Wi);
      const long long f = row / (unsigned)Hi;
      const int yo = 1 + (int)(row - (unsigned)f * Hi);
      const TA* p = base + ((f * H + 2 * yo) * W + 2 * xo) * bs + c;
      const f32x8_t m = max8(max8(ldv8(p), ldv8(p + bs)), max8(ldv8(p + (long long)W * bs), ldv8(p + (long long)W * bs + bs)));
#pragma unroll
      for (int i = 0; i < LB_LH; ++i)
        if (l0 + i < L) emit(i, f, yo, xo, m + ld8f(P + ((size_t)(l0 + i) * 9 + 4) * C + c));
    }
    // ---- border windows: the general arithmetic ----
    const int nb = 2 * Wo + 2 * (Ho - 2);
    const long long nbor = F * nb;
    for (unsigned op = blockIdx.x * lanes + wl; op < (unsigned)nbor; op += gridDim.x * lanes) {
      const long long f = op / (unsigned)nb;
      int yo, xo;
      border_window((int)(op - (unsigned)f * nb), Ho, Wo, yo, xo);
      const TA* p = base + ((f * H + 2 * yo) * W + 2 * xo) * bs + c;
      const f32x8_t v0 = ldv8(p), v1 = ldv8(p + bs), v2 = ldv8(p + (long long)W * bs), v3 = ldv8(p + (long long)W * bs + bs);
      const int k0 = border_class(2 * yo, 2 * xo, H, W), k1 = border_class(2 * yo, 2 * xo + 1, H, W);
      const int k2 = border_class(2 * yo + 1, 2 * xo, H, W), k3 = border_class(2 * yo + 1, 2 * xo + 1, H, W);
#pragma unroll
      for (int i = 0; i < LB_LH; ++i)
        if (l0 + i < L) {
          const float* pt = P + (size_t)(l0 + i) * 9 * C + c;
          emit(i, f, yo, xo, max8(max8(v0 + ld8f(pt + k0 * C), v1 + ld8f(pt + k1 * C)), max8(v2 + ld8f(pt + k2 * C), v3 + ld8f(pt + k3 * C))));
        }
    }
  }
  // window lanes in order: lane w adds its sums in round w (deterministic)
  for (int w = 0; w < lanes; ++w) {
    if (wl == w) {
#pragma unroll
      for (int i = 0; i < LB_LH; ++i)
        if (l0 + i < L) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { S[((size_t)(l0 + i) * C + c + j) * 2] += s1[i][j]; S[((size_t)(l0 + i) * C + c + j) * 2 + 1] += s2[i][j]; }
        }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < L * C * 2; i += blockDim.x) {
    const int l = i / (C * 2), rest = i - l * C * 2;
    stats[((size_t)l * gridDim.x + blockIdx.x) * C * 2 + rest] = S[i];
  }
}

template <typename TA, bool PLDS>
__global__ __launch_bounds__(512) void leadbias_pool_bwd2_kernel(const TA* __restrict__ base, int bs, const TA* __restrict__ dout, int dos,
                                                                long long F, int H, int W, int C, int L, const float* __restrict__ ptab_g,
                                                                TA* __restrict__ dbase, int dbs, float* __restrict__ main_part) {
  extern __shared__ float S[];  // [L][C] interior-class sums | [C] max_l |b_l| of the interior class (+ [L][9][C] table copy)
  const int nS = L * C, nP = L * 9 * C;
  float* Bc = S + nS;
  for (int i = threadIdx.x; i < nS; i += blockDim.x) S[i] = 0.f;
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    float b = 0.f;
    for (int l = 0; l < L; ++l) b = fmaxf(b, fabsf(ptab_g[((size_t)l * 9 + 4) * C + i]));
    Bc[i] = b;
  }
  if (PLDS)
    for (int i = threadIdx.x; i < nP; i += blockDim.x) S[nS + C + i] = ptab_g[i];
  const float* ptab = PLDS ? S + nS + C : ptab_g;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2, o8 = C / 8;
  const int lanes = blockDim.x / o8;  // windows in flight per block
  const int oc = threadIdx.x % o8, wl = threadIdx.x / o8;
  const int c = oc * 8;
  const bool active = wl < lanes;
  typedef typename std::conditional<std::is_same<TA, float>::value, f32x8_t, bf16x8_t>::type raw8;  // an octet as stored
  auto ldraw = [](const TA* q) __attribute__((always_inline)) -> raw8 {
    if constexpr (std::is_same<TA, float>::value) return ld8f(q);
    else return *reinterpret_cast<const bf16x8_t*>(q);
  };
  f32x8_t s4[LEAD_REG];
  const f32x8_t zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < LEAD_REG; ++l) s4[l] = zero8;
  const long long lead_stride = F * Ho * Wo * dos;  // elements between two lead times of dout
  // ---- interior windows ----
  {
    const int Hi = Ho - 2, Wi = Wo - 2;
    const long long nint = F * Hi * Wi;
    const long long rounds = (nint + (long long)gridDim.x * lanes - 1) / ((long long)gridDim.x * lanes);  // every wave runs every round (wave votes inside)
    for (long long rd = 0; rd < rounds; ++rd) {
      const long long op = (rd * gridDim.x + blockIdx.x) * lanes + wl;
      const bool on = active && op < nint;
      const unsigned opc = on ? (unsigned)op : 0u;  // (32-bit window indices: see the forward kernel)
      const unsigned row = opc / (unsigned)Wi;
      const int xo = 1 + (int)(opc - row * Wi);
      const long long f = row / (unsigned)Hi;
      const int yo = 1 + (int)(row - (unsigned)f * Hi);
      const long long b0 = (f * H + 2 * yo) * W + 2 * xo;
#ifdef SF_EXP_LB_NOBASE
      const TA* p = base + (b0 & 0xffff) * bs + c;
#else
      const TA* p = base + b0 * bs + c;
#endif
      const raw8 r0 = ldraw(p), r1 = ldraw(p + bs), r2 = ldraw(p + (long long)W * bs), r3 = ldraw(p + (long long)W * bs + bs);
      raw8 gr[LEAD_REG];
      // a scalar base per lead time + ONE 32-bit element offset per window (a lead time's tensor is below 2^31 elements, the launcher checks): the
      // 64-bit multiply-add chains in front of every load were a fifth of this loop's instructions
      const unsigned goff = ((((unsigned)f * Ho + yo) * Wo + xo) * (unsigned)dos) + c;
#pragma unroll
      for (int l = 0; l < LEAD_REG; ++l)  // all lead times requested before the first use; clamped index: unconditional loads
#ifdef SF_EXP_LB_NODOUT   // (ablation builds only: tools/ablate_leadbias.sh)
        gr[l] = ldraw(dout + (long long)(l < L ? l : 0) * lead_stride + (goff & 0xffffu));
#else
        gr[l] = ldraw(dout + (long long)(l < L ? l : 0) * lead_stride + goff);
#endif
      // first maximum of v, the largest value below it, and the collision test
      int am[8];
      bool safe = true;
      {
      const f32x8_t v0 = __builtin_convertvector(r0, f32x8_t), v1 = __builtin_convertvector(r1, f32x8_t), v2 = __builtin_convertvector(r2, f32x8_t),
                    v3 = __builtin_convertvector(r3, f32x8_t);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float m = v0[j]; int a = 0;
        if (v1[j] > m) { m = v1[j]; a = 1; }
        if (v2[j] > m) { m = v2[j]; a = 2; }
        if (v3[j] > m) { m = v3[j]; a = 3; }
        float sc = -INFINITY;
        sc = v0[j] < m ? fmaxf(sc, v0[j]) : sc; sc = v1[j] < m ? fmaxf(sc, v1[j]) : sc;
        sc = v2[j] < m ? fmaxf(sc, v2[j]) : sc; sc = v3[j] < m ? fmaxf(sc, v3[j]) : sc;
        am[j] = a;
        const float mag = fmaxf(fabsf(m), sc == -INFINITY ? 0.f : fabsf(sc)) + Bc[c + j];
        safe = safe && (sc == -INFINITY || (m - sc) > mag * 4.76837158203125e-07f);  // 2^-21
      }
      }
      f32x8_t G = zero8;
#pragma unroll
      for (int l = 0; l < LEAD_REG; ++l) {
        const f32x8_t g = l < L ? __builtin_convertvector(gr[l], f32x8_t) : zero8;
        G += g;
        if (on) s4[l] += g;  // (off only in a workgroup's last round: no per-element select)
      }
      if (__all(safe || !on)) {
        if (on) {
          f32x8_t gp[4];
#pragma unroll
          for (int j = 0; j < 8; ++j) { gp[0][j] = am[j] == 0 ? G[j] : 0.f; gp[1][j] = am[j] == 1 ? G[j] : 0.f; gp[2][j] = am[j] == 2 ? G[j] : 0.f; gp[3][j] = am[j] == 3 ? G[j] : 0.f; }
#ifdef SF_EXP_LB_NOSTORE
          TA* d = dbase + (b0 & 0xffff) * dbs + c;
#else
          TA* d = dbase + b0 * dbs + c;
#endif
          stv8(d, gp[0]); stv8(d + dbs, gp[1]); stv8(d + (long long)W * dbs, gp[2]); stv8(d + (long long)W * dbs + dbs, gp[3]);
        }
      } else if (on) {
        // a lane of this wave failed the collision test (a fraction of a percent of the waves): the general arithmetic for this window, lead time by
        // lead time in a rolled loop (all four positions have class 4) - slow, rare, and small in registers
        const f32x8_t v0 = __builtin_convertvector(r0, f32x8_t), v1 = __builtin_convertvector(r1, f32x8_t), v2 = __builtin_convertvector(r2, f32x8_t),
                      v3 = __builtin_convertvector(r3, f32x8_t);
        f32x8_t gp[4] = {zero8, zero8, zero8, zero8};
#pragma unroll 1
        for (int l = 0; l < L; ++l) {
          const f32x8_t g = __builtin_convertvector(ldraw(dout + ((((long long)l * F + f) * Ho + yo) * Wo + xo) * dos + c), f32x8_t);
          const f32x8_t bl = ld8f(ptab + ((size_t)l * 9 + 4) * C + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int a = argmax4(v0[j] + bl[j], v1[j] + bl[j], v2[j] + bl[j], v3[j] + bl[j]);
            gp[0][j] += a == 0 ? g[j] : 0.f; gp[1][j] += a == 1 ? g[j] : 0.f; gp[2][j] += a == 2 ? g[j] : 0.f; gp[3][j] += a == 3 ? g[j] : 0.f;
          }
        }
        TA* d = dbase + b0 * dbs + c;
        stv8(d, gp[0]); stv8(d + dbs, gp[1]); stv8(d + (long long)W * dbs, gp[2]); stv8(d + (long long)W * dbs + dbs, gp[3]);
      }
    }
  }
#pragma unroll
  for (int l = 0; l < LEAD_REG; ++l)
    if (l < L && active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(S + (size_t)l * C + c + j, s4[l][j]);
    }
  __syncthreads();
  float* dstp = main_part + (size_t)blockIdx.x * nS;
  for (int i = threadIdx.x; i < nS; i += blockDim.x) dstp[i] = S[i];
}

// Border windows.  Blocks take roles: 4 x EDGE blocks (top / bottom / left / right edge without the corners: the two
// positions on the image border share one class, the two inner ones are class 4, so two register sums per lead time
// suffice and the loop is atomics-free like the main kernel's) and the last LEADBIAS_CORNER_BLOCKS blocks for the corner
// windows (4 classes; LDS atomics) - or for every border window when the image is too small to have plain edges.
constexpr int LEADBIAS_BORDER_THREADS = 512;  // two waves per SIMD (227 VGPRs): twice the windows in flight per CU of a kernel that is a chain of dependent loads (round 6: 256 -> 512)
constexpr int LEADBIAS_CORNER_BLOCKS = 32;  // (8 were the tail of the kernel: 290 us of per-window serial loads)
// dbase (nullable, round 4): also write the input gradient of the windows visited here - the main kernel then covers the interior windows only
// (leadbias_pool_bwd2_kernel).
template <typename TA, bool PLDS>
__global__ __launch_bounds__(LEADBIAS_BORDER_THREADS) void leadbias_border_kernel(const TA* __restrict__ base, int bs, const TA* __restrict__ dout, int dos,
                                                             long long F, int H, int W, int C, int L, const float* __restrict__ ptab_g,
                                                             float* __restrict__ border_part, TA* __restrict__ dbase, int dbs) {
  extern __shared__ float S[];  // [L][9][C] (+ table copy)
  const int nS = L * 9 * C;
  for (int i = threadIdx.x; i < nS; i += blockDim.x) S[i] = 0.f;
  if (PLDS)
    for (int i = threadIdx.x; i < nS; i += blockDim.x) S[nS + i] = ptab_g[i];
  const float* ptab = PLDS ? S + nS : ptab_g;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2, q = C / 4;
  const int lanes = blockDim.x / q;
  const int cq = threadIdx.x % q, wl = threadIdx.x / q;
  const int c = cq * 4;
  const bool edges = Ho >= 3 && Wo >= 3 && L <= LEAD_REG;
  const int edge_blocks = ((int)gridDim.x - LEADBIAS_CORNER_BLOCKS) / 4;  // per edge type
  const int role = (int)blockIdx.x < 4 * edge_blocks ? (int)blockIdx.x / edge_blocks : 4;
  auto window = [&](long long f, int yo, int xo, f32x4 (&v)[4], int (&k)[4]) {
    const TA* p = base + ((f * H + 2 * yo) * W + 2 * xo) * bs + c;
    v[0] = ldv4(p); v[1] = ldv4(p + bs); v[2] = ldv4(p + (long long)W * bs); v[3] = ldv4(p + (long long)W * bs + bs);
    k[0] = border_class(2 * yo, 2 * xo, H, W); k[1] = border_class(2 * yo, 2 * xo + 1, H, W);
    k[2] = border_class(2 * yo + 1, 2 * xo, H, W); k[3] = border_class(2 * yo + 1, 2 * xo + 1, H, W);
  };
  if (role < 4) {
    if (edges && wl < lanes) {
      const int n_t = role < 2 ? Wo - 2 : Ho - 2;                       // edge windows of this type per frame
      const int outer_mask = role == 0 ? 0x3 : role == 1 ? 0xC : role == 2 ? 0x5 : 0xA;  // positions on the image border
      const int ko = role == 0 ? 1 : role == 1 ? 7 : role == 2 ? 3 : 5;  // their class; the inner positions are class 4
      f32x4 so[LEAD_REG], si[LEAD_REG];
#pragma unroll
      for (int l = 0; l < LEAD_REG; ++l) so[l] = si[l] = f32x4{0.f, 0.f, 0.f, 0.f};
      const long long total = F * n_t;
      for (long long e = (long long)(blockIdx.x - role * edge_blocks) * lanes + wl; e < total; e += (long long)edge_blocks * lanes) {
        const long long f = e / n_t;
        const int i = 1 + (int)(e - f * n_t);
        const int yo = role == 0 ? 0 : role == 1 ? Ho - 1 : i, xo = role == 2 ? 0 : role == 3 ? Wo - 1 : i;
        f32x4 v[4]; int k[4];
        window(f, yo, xo, v, k);
        f32x4 gl[LEAD_REG];
#pragma unroll
        for (int l = 0; l < LEAD_REG; ++l) gl[l] = ldv4(dout + ((((l < L ? l : 0) * F + f) * Ho + yo) * Wo + xo) * dos + c);
        f32x4 gp0 = {0.f, 0.f, 0.f, 0.f}, gp1 = gp0, gp2 = gp0, gp3 = gp0;
#pragma unroll
        for (int l = 0; l < LEAD_REG; ++l) {
          const float* pt = ptab + (size_t)(l < L ? l : 0) * 9 * C + c;
          const f32x4 a0 = v[0] + ld4(pt + k[0] * C), a1 = v[1] + ld4(pt + k[1] * C), a2 = v[2] + ld4(pt + k[2] * C), a3 = v[3] + ld4(pt + k[3] * C);
          const f32x4 g = l < L ? gl[l] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int am = argmax4(a0[j], a1[j], a2[j], a3[j]);
            const bool outer = (outer_mask >> am) & 1;
            so[l][j] += outer ? g[j] : 0.f; si[l][j] += outer ? 0.f : g[j];
            if (dbase) { gp0[j] += am == 0 ? g[j] : 0.f; gp1[j] += am == 1 ? g[j] : 0.f; gp2[j] += am == 2 ? g[j] : 0.f; gp3[j] += am == 3 ? g[j] : 0.f; }
          }
        }
        if (dbase) {
          TA* d = dbase + ((f * H + 2 * yo) * W + 2 * xo) * dbs + c;
          stv4(d, gp0); stv4(d + dbs, gp1); stv4(d + (long long)W * dbs, gp2); stv4(d + (long long)W * dbs + dbs, gp3);
        }
      }
#pragma unroll
      for (int l = 0; l < LEAD_REG; ++l)
        if (l < L) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            atomicAdd(S + ((size_t)l * 9 + ko) * C + c + j, so[l][j]);
            atomicAdd(S + ((size_t)l * 9 + 4) * C + c + j, si[l][j]);
          }
        }
    }
  } else if (wl < lanes) {
    // corners (or, without plain edges, every border window: top row, bottom row, then left / right of the rows between)
    const int per_frame = edges ? 4 : (Ho > 1 ? 2 * Wo : Wo) + (Ho > 2 ? (Wo > 1 ? 2 : 1) * (Ho - 2) : 0);
    const long long total = F * per_frame;
    const int cb = (int)blockIdx.x - 4 * edge_blocks, ncb = (int)gridDim.x - 4 * edge_blocks;
    for (long long bw = (long long)cb * lanes + wl; bw < total; bw += (long long)ncb * lanes) {
      const long long f = bw / per_frame;
      int r = (int)(bw - f * per_frame), yo, xo;
      if (edges) { yo = (r & 2) ? Ho - 1 : 0; xo = (r & 1) ? Wo - 1 : 0; }
      else if (r < Wo) { yo = 0; xo = r; }
      else if (Ho > 1 && r < 2 * Wo) { yo = Ho - 1; xo = r - Wo; }
      else { r -= 2 * Wo; const int per_row = Wo > 1 ? 2 : 1; yo = 1 + r / per_row; xo = (r % per_row) ? Wo - 1 : 0; }
      f32x4 v[4]; int k[4];
      window(f, yo, xo, v, k);
      f32x4 gl[LEAD_REG];  // the first LEAD_REG lead times' gradients requested together (clamped index: unconditional loads)
#pragma unroll
      for (int l = 0; l < LEAD_REG; ++l) gl[l] = ldv4(dout + ((((l < L ? l : 0) * F + f) * Ho + yo) * Wo + xo) * dos + c);
      f32x4 gq[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      for (int l = 0; l < L; ++l) {
        const float* pt = ptab + (size_t)l * 9 * C + c;
        const f32x4 a0 = v[0] + ld4(pt + k[0] * C), a1 = v[1] + ld4(pt + k[1] * C), a2 = v[2] + ld4(pt + k[2] * C), a3 = v[3] + ld4(pt + k[3] * C);
        f32x4 g;
        if (l < LEAD_REG) {
          g = gl[0];
#pragma unroll
          for (int u = 1; u < LEAD_REG; ++u) g = l == u ? gl[u] : g;
        } else g = ldv4(dout + (((l * F + f) * Ho + yo) * Wo + xo) * dos + c);
        float* Sl = S + (size_t)l * 9 * C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int am = argmax4(a0[j], a1[j], a2[j], a3[j]);
          atomicAdd(Sl + k[am] * C + j, g[j]);
          gq[0][j] += am == 0 ? g[j] : 0.f; gq[1][j] += am == 1 ? g[j] : 0.f; gq[2][j] += am == 2 ? g[j] : 0.f; gq[3][j] += am == 3 ? g[j] : 0.f;
        }
      }
      if (dbase) {
        TA* d = dbase + ((f * H + 2 * yo) * W + 2 * xo) * dbs + c;
        stv4(d, gq[0]); stv4(d + dbs, gq[1]); stv4(d + (long long)W * dbs, gq[2]); stv4(d + (long long)W * dbs + dbs, gq[3]);
      }
    }
  }
  __syncthreads();
  float* dstp = border_part + (size_t)blockIdx.x * nS;
  for (int i = threadIdx.x; i < nS; i += blockDim.x) dstp[i] = S[i];
}

// stage 1: cls_sum[l][cls][c] = sum of the border slabs (+ the main slabs for the interior class).  grid (ceil(n/64), 1),
// block (64, 8): 8 slab groups per element, combined through LDS.
__global__ __launch_bounds__(512) void leadbias_reduce_kernel(const float* __restrict__ main_part, int nmain, const float* __restrict__ border_part,
                                                             int nborder, int L, int C, float* __restrict__ cls_sum) {
  __shared__ float red[8][64];
  const int n = L * 9 * C;
  const int i = blockIdx.x * 64 + threadIdx.x, grp = threadIdx.y;
  float a = 0.f;
  if (i < n) {
    float a4[4] = {0.f, 0.f, 0.f, 0.f};  // 4 loads in flight per chain (fixed order: deterministic)
    int b = grp;
    for (; b + 24 < nborder; b += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a4[u] += border_part[(size_t)(b + 8 * u) * n + i];
    }
    for (; b < nborder; b += 8) a4[0] += border_part[(size_t)b * n + i];
    const int c = i % C, cls = (i / C) % 9, l = i / (9 * C);
    if (cls == 4) {
      b = grp;
      for (; b + 24 < nmain; b += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] += main_part[((size_t)(b + 8 * u) * L + l) * C + c];
      }
      for (; b < nmain; b += 8) a4[0] += main_part[((size_t)b * L + l) * C + c];
    }
    a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  red[grp][threadIdx.x] = a;
  __syncthreads();
  if (grp == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
    cls_sum[i] = t;
  }
}

// stage 2: dw1[co][cimg + l][ky][kx] = sum over the border classes for which that tap lies inside the image
__global__ void leadbias_wgrad_kernel(const float* __restrict__ cls_sum, int L, int C, int O, int I, int cimg, float* __restrict__ dw1) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L * O) return;
  const int co = idx % O, l = idx / O;
  float s[9];
  for (int cls = 0; cls < 9; ++cls) s[cls] = cls_sum[((size_t)l * 9 + cls) * C + co];
  float* w = dw1 + ((size_t)co * I + cimg + l) * 9;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      float a = 0.f;
      for (int ry = 0; ry < 3; ++ry) {
        if ((ry == 0 && ky == 0) || (ry == 2 && ky == 2)) continue;
        for (int rx = 0; rx < 3; ++rx) {
          if ((rx == 0 && kx == 0) || (rx == 2 && kx == 2)) continue;
          a += s[ry * 3 + rx];
        }
      }
      w[ky * 3 + kx] = a;
    }
}


}  // namespace

extern "C" {

size_t sf_leadtime_pool_workspace_floats(int32_t L, int32_t C) { return (size_t)L * 9 * C * (2 + LEADBIAS_BORDER_BLOCKS) + (size_t)L * C * LEADBIAS_BLOCKS; }

int sf_leadtime_pool_fwd(sfTensor base, int64_t frames, int32_t h, int32_t w, const float* w1, int32_t O, int32_t I, int32_t cimg,
                         int32_t L, float* workspace, sfTensor out, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_leadtime_pool_fwd: dtype %d not built", dtype);
  SF_REQUIRE(h % 2 == 0 && w % 2 == 0 && h >= 2 && w >= 2 && base.c == out.c && ok4(base) && ok4(out) && base.dtype == out.dtype && O <= base.c && cimg + L <= I,
             "leadtime_pool: shapes (h=%d w=%d C=%d O=%d I=%d cimg=%d L=%d)", h, w, base.c, O, I, cimg, L);
  hipStream_t st = (hipStream_t)stream;
  const int C = base.c, nt = L * 9 * C;
  hipLaunchKernelGGL(leadbias_table_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, w1, O, I, cimg, L, C, workspace);
  SF_CHECK_LAUNCH("leadbias_table");
  const long long total = frames * (h / 2) * (w / 2) * (C / 4);
  if (total == 0) return 0;
  const size_t tab_bytes = (size_t)nt * sizeof(float);
  const int blocks = (int)((total + 1023) / 1024 < 512 ? (total + 1023) / 1024 : 512);
  if (tab_bytes <= 76 * 1024) {  // two 1024-thread workgroups per CU keep their copies of the table
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)leadbias_pool_fwd_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
      (void)hipFuncSetAttribute((const void*)leadbias_pool_fwd_kernel<__bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
      attr_set = true;
    }
    SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_fwd_kernel<TA, true>), dim3(blocks), dim3(1024), tab_bytes, st, (const TA*)base.ptr,
                                                   base.stride, (long long)frames, h, w, C, L, (const float*)workspace, (TA*)out.ptr, out.stride));
  } else {
    SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_fwd_kernel<TA, false>), dim3(blocks), dim3(1024), 0, st, (const TA*)base.ptr,
                                                   base.stride, (long long)frames, h, w, C, L, (const float*)workspace, (TA*)out.ptr, out.stride));
  }
  SF_CHECK_LAUNCH("leadbias_pool_fwd");
  return 0;
}

int32_t sf_leadtime_pool_stats_tiles(void) { return LEADBIAS_BLOCKS; }

int sf_leadtime_pool_fwd_stats(sfTensor base, int64_t frames, int32_t h, int32_t w, const float* w1, int32_t O, int32_t I, int32_t cimg,
                               int32_t L, float* workspace, sfTensor out, float* stats, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_leadtime_pool_fwd_stats: dtype %d not built", dtype);
  SF_REQUIRE(h % 2 == 0 && w % 2 == 0 && h >= 2 && w >= 2 && base.c == out.c && ok4(base) && ok4(out) && base.dtype == out.dtype && O <= base.c && cimg + L <= I,
             "leadtime_pool: shapes (h=%d w=%d C=%d O=%d I=%d cimg=%d L=%d)", h, w, base.c, O, I, cimg, L);
  SF_REQUIRE(stats != nullptr && L <= LEAD_REG_FWD && base.c / 4 <= 512, "leadtime_pool_fwd_stats: stats null, L=%d > %d or too many channels", L, LEAD_REG_FWD);
  hipStream_t st = (hipStream_t)stream;
  const int C = base.c, nt = L * 9 * C;
  const size_t lds = ((size_t)nt + (size_t)L * C * 2) * sizeof(float);
  SF_REQUIRE(lds <= 160 * 1024, "leadtime_pool_fwd_stats: L*11*C floats exceed LDS");
  hipLaunchKernelGGL(leadbias_table_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, w1, O, I, cimg, L, C, workspace);
  SF_CHECK_LAUNCH("leadbias_table");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)leadbias_pool_fwd_stats_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_pool_fwd_stats_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  static const bool old_kernels = getenv("SF_LEADBIAS_V1") != nullptr;  // A/B switch: the kernels of rounds 1-3
  if (!old_kernels && h >= 6 && w >= 6 && C % 8 == 0 && 2 * (C / 8) <= 512 && L <= 2 * LB_LH && frames * (h / 2) * (w / 2) < 0x7fffffffll) {  // interior windows on the short path
    static bool attr2_set = false;
    if (!attr2_set) {
      (void)hipFuncSetAttribute((const void*)leadbias_pool_fwd_stats2_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)leadbias_pool_fwd_stats2_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr2_set = true;
    }
    SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_fwd_stats2_kernel<TA>), dim3(LEADBIAS_BLOCKS), dim3(512), lds, st, (const TA*)base.ptr,
                                                   base.stride, (long long)frames, h, w, C, L, (const float*)workspace, (TA*)out.ptr, out.stride, stats));
    SF_CHECK_LAUNCH("leadbias_pool_fwd_stats2");
    return 0;
  }
  SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_fwd_stats_kernel<TA>), dim3(LEADBIAS_BLOCKS), dim3(512), lds, st, (const TA*)base.ptr,
                                                 base.stride, (long long)frames, h, w, C, L, (const float*)workspace, (TA*)out.ptr, out.stride, stats));
  SF_CHECK_LAUNCH("leadbias_pool_fwd_stats");
  return 0;
}

int sf_leadtime_pool_bwd(sfTensor base, sfTensor dout, int64_t frames, int32_t h, int32_t w, const float* w1, int32_t O, int32_t I,
                         int32_t cimg, int32_t L, float* workspace, sfTensor dbase, float* dw1, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_leadtime_pool_bwd: dtype %d not built", dtype);
  SF_REQUIRE(h % 2 == 0 && w % 2 == 0 && base.c == dout.c && base.c == dbase.c && ok4(base) && ok4(dout) && ok4(dbase) &&
                 base.dtype == dout.dtype && base.dtype == dbase.dtype, "leadtime_pool bwd: shapes / storage types");
  hipStream_t st = (hipStream_t)stream;
  const int C = base.c, nt = L * 9 * C;
  SF_REQUIRE((size_t)nt * sizeof(float) <= 160 * 1024, "leadtime_pool bwd: L*9*C floats exceed LDS");
  hipLaunchKernelGGL(leadbias_table_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, w1, O, I, cimg, L, C, workspace);
  SF_CHECK_LAUNCH("leadbias_table");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd_kernel<__bf16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd_kernel<__bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_border_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_border_kernel<__bf16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_border_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)leadbias_border_kernel<__bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // workspace: [table nt][border slabs 128 x nt][main slabs 512 x L*C][cls_sum nt]
  float* border_part = workspace + nt;
  float* main_part = border_part + (size_t)LEADBIAS_BORDER_BLOCKS * nt;
  float* cls_sum = main_part + (size_t)LEADBIAS_BLOCKS * L * C;
  const size_t f4 = sizeof(float), lds_main = (size_t)L * C * f4, lds_border = (size_t)nt * f4, lds_tab = (size_t)nt * f4;
#define SF_LB_ARGS (const TA*)base.ptr, base.stride, (const TA*)dout.ptr, dout.stride, (long long)frames, h, w, C, L, (const float*)workspace
  static const bool old_kernels = getenv("SF_LEADBIAS_V1") != nullptr;  // A/B switch: the main kernel of rounds 1-3
  bool border_dbase = false;
  if (!old_kernels && base.dtype == SF_BF16 && h >= 6 && w >= 6 && C % 8 == 0 && C / 8 <= 512 && L <= LEAD_REG && frames * (h / 2) * (w / 2) * dout.stride < 0x7fffffffll) {  // interior windows on the short path (bf16 storage:
                                                                                                               // with fp32 octets the kernel spills 60 registers)
    static bool attr2_set = false;
    if (!attr2_set) {
      (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd2_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd2_kernel<__bf16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd2_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)leadbias_pool_bwd2_kernel<__bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr2_set = true;
    }
    const size_t lds2 = lds_main + (size_t)C * f4;
    if (lds2 + lds_tab <= 160 * 1024)
      SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_bwd2_kernel<TA, true>), dim3(LEADBIAS_BLOCKS), dim3(512), lds2 + lds_tab, st,
                                                     SF_LB_ARGS, (TA*)dbase.ptr, dbase.stride, main_part));
    else
      SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_bwd2_kernel<TA, false>), dim3(LEADBIAS_BLOCKS), dim3(512), lds2, st,
                                                     SF_LB_ARGS, (TA*)dbase.ptr, dbase.stride, main_part));
    SF_CHECK_LAUNCH("leadbias_pool_bwd2");
    border_dbase = true;  // the border ring's input gradient comes from the border kernel
  } else if (lds_main + lds_tab <= 160 * 1024)
    SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_bwd_kernel<TA, true>), dim3(LEADBIAS_BLOCKS), dim3(512), lds_main + lds_tab, st,
                                                   SF_LB_ARGS, (TA*)dbase.ptr, dbase.stride, main_part));
  else
    SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_pool_bwd_kernel<TA, false>), dim3(LEADBIAS_BLOCKS), dim3(512), lds_main, st,
                                                   SF_LB_ARGS, (TA*)dbase.ptr, dbase.stride, main_part));
  SF_CHECK_LAUNCH("leadbias_pool_bwd");
  if (lds_border + lds_tab <= 160 * 1024)
    SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_border_kernel<TA, true>), dim3(LEADBIAS_BORDER_BLOCKS), dim3(LEADBIAS_BORDER_THREADS), lds_border + lds_tab, st,
                                                   SF_LB_ARGS, border_part, border_dbase ? (TA*)dbase.ptr : (TA*)nullptr, dbase.stride));
  else
    SF_DISPATCH_ACT(base.dtype, hipLaunchKernelGGL((leadbias_border_kernel<TA, false>), dim3(LEADBIAS_BORDER_BLOCKS), dim3(LEADBIAS_BORDER_THREADS), lds_border, st,
                                                   SF_LB_ARGS, border_part, border_dbase ? (TA*)dbase.ptr : (TA*)nullptr, dbase.stride));
#undef SF_LB_ARGS
  SF_CHECK_LAUNCH("leadbias_border");
  hipLaunchKernelGGL(leadbias_reduce_kernel, dim3((nt + 63) / 64), dim3(64, 8), 0, st, main_part, LEADBIAS_BLOCKS, border_part, LEADBIAS_BORDER_BLOCKS, L,
                     C, cls_sum);
  SF_CHECK_LAUNCH("leadbias_reduce");
  hipLaunchKernelGGL(leadbias_wgrad_kernel, dim3((L * O + 127) / 128), dim3(128), 0, st, cls_sum, L, C, O, I, cimg, dw1);
  SF_CHECK_LAUNCH("leadbias_wgrad");
  return 0;
}

}  // extern "C"
