// General 2-D convolution (any kernel size / stride / zero padding) on the exact-fp32 matrix cores, forward, input gradient and
// weight gradient.  Used by the PatchGAN discriminator of CloudGAN (reference satflow/models/gan/discriminators.py:139-223:
// 4x4 convolutions with stride 2 and 1, padding 1) - the caller-side network of SURVEY 8f-2 around the ConvLSTM generator.
//
// Implicit GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32: the parity gate is rtol 1e-4), operands gathered straight from global
// memory - these layers are 1-2 % of the generator's FLOPs, so there is no LDS staging:
//   forward   M = output pixels, N = cout, K = (ky, kx, ci)   out[p][co] = sum x[in(p, ky, kx)][ci] * w[co][ci][ky][kx] (+ bias, LeakyReLU)
//   bwd data  M = input pixels,  N = cin,  K = (ky, kx, co)   dx[q][ci]  = sum dy[out(q, ky, kx)][co] * w[co][ci][ky][kx]
//   bwd wgt   M = cout, N = cin, K = output pixels, one (ky, kx) tap per workgroup column, split-K slabs + deterministic reduce
// Activations NHWC fp32 with explicit pixel strides (channel counts padded to a multiple of 8 by the caller); weights in the
// reference's OIHW layout, read directly (no packing).
#include "sf_common.h"

namespace {

struct GConvParams {
  const float* x; int xs;          // input  [N][H][W][xs], cin real channels (lanes >= cin are never read as non-zero weights)
  const float* w;                  // [cout][cin][kh][kw]
  const float* bias;               // [cout] or null
  float* y; int ys; int yc;        // output [N][OH][OW][ys]; yc = channel lanes the caller owns in a pixel (<= ys: a strided view keeps its neighbours)
  int N, H, W, OH, OW, cin, cout, kh, kw, stride, pad;
  float slope;                     // fused LeakyReLU negative slope (1 = identity)
};

__device__ __forceinline__ float pick4(const f32x4& a, const f32x4& b, int k) {  // element k of the 8 values [a | b], k compile-time
  return k < 4 ? a[k] : b[k - 4];
}

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gconv_fwd_kernel(const GConvParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const long long npix = (long long)p.N * p.OH * p.OW;
  const long long pm = ((long long)blockIdx.x * 4 + wave) * 32 + i;  // this lane's output pixel (A row)
  const int co = blockIdx.y * 32 + i;                                  // this lane's output channel (B column)
  const bool pv = pm < npix;
  const long long pc = pv ? pm : 0;
  const int ox = pc % p.OW, oy = (pc / p.OW) % p.OH, n = pc / ((long long)p.OW * p.OH);
  const int cin8 = (p.cin + 7) / 8 * 8;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int ky = 0; ky < p.kh; ++ky)
    for (int kx = 0; kx < p.kw; ++kx) {
      const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
      const bool ok = pv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float* xp = p.x + ((long long)(n * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * p.xs;
      const float* wp = p.w + (long long)(co < p.cout ? co : 0) * p.cin * p.kh * p.kw + ky * p.kw + kx;
      for (int c0 = 0; c0 < cin8; c0 += 8) {
        f32x4 a0 = *reinterpret_cast<const f32x4*>(xp + c0), a1 = *reinterpret_cast<const f32x4*>(xp + c0 + 4);
        if (!ok) { a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int ci = c0 + 2 * m + h;
          const float b = (co < p.cout && ci < p.cin) ? wp[(long long)ci * p.kh * p.kw] : 0.f;
          const float a = h ? pick4(a0, a1, 2 * m + 1) : pick4(a0, a1, 2 * m);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
    }
  // D[row = pixel][col = channel]: lane holds channel `co`, rows frag_row(reg, h)
  if (co < p.yc) {
    const float bv = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long q = ((long long)blockIdx.x * 4 + wave) * 32 + frag_row(r, h);
      if (q < npix) {
        float v = co < p.cout ? acc[r] + bv : 0.f;  // pad lanes are written as zeros
        v = v > 0.f ? v : v * p.slope;
        p.y[q * p.ys + co] = v;
      }
    }
  }
}

// ---- input gradient --------------------------------------------------------------------------------------------------
// here p.x = dy [N][OH][OW][xs] (cout channels), p.y = dx [N][H][W][ys] (cin channels), p.w as in the forward
__global__ __launch_bounds__(256) void gconv_bwd_data_kernel(const GConvParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const long long npix = (long long)p.N * p.H * p.W;
  const long long pm = ((long long)blockIdx.x * 4 + wave) * 32 + i;  // this lane's INPUT pixel
  const int ci = blockIdx.y * 32 + i;
  const bool pv = pm < npix;
  const long long pc = pv ? pm : 0;
  const int ix = pc % p.W, iy = (pc / p.W) % p.H, n = pc / ((long long)p.W * p.H);
  const int co8 = (p.cout + 7) / 8 * 8;
  const long long wstride = (long long)p.cin * p.kh * p.kw;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int ky = 0; ky < p.kh; ++ky)
    for (int kx = 0; kx < p.kw; ++kx) {
      const int ty = iy + p.pad - ky, tx = ix + p.pad - kx;
      const bool ok = pv && ty >= 0 && tx >= 0 && ty % p.stride == 0 && tx % p.stride == 0 && ty / p.stride < p.OH && tx / p.stride < p.OW;
      const float* gp = p.x + ((long long)(n * p.OH + (ok ? ty / p.stride : 0)) * p.OW + (ok ? tx / p.stride : 0)) * p.xs;
      const float* wp = p.w + (long long)(ci < p.cin ? ci : 0) * p.kh * p.kw + ky * p.kw + kx;
      for (int c0 = 0; c0 < co8; c0 += 8) {
        f32x4 a0 = *reinterpret_cast<const f32x4*>(gp + c0), a1 = *reinterpret_cast<const f32x4*>(gp + c0 + 4);
        if (!ok) { a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int co = c0 + 2 * m + h;
          const float b = (ci < p.cin && co < p.cout) ? wp[co * wstride] : 0.f;
          const float a = h ? pick4(a0, a1, 2 * m + 1) : pick4(a0, a1, 2 * m);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
    }
  if (ci < p.yc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long q = ((long long)blockIdx.x * 4 + wave) * 32 + frag_row(r, h);
      if (q < npix) p.y[q * p.ys + ci] = ci < p.cin ? acc[r] : 0.f;
    }
  }
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
// grid (co tiles * ci tiles, kh * kw taps, K slices); 4 waves split the slice's pixels, LDS reduce, partial[ks][tap][co][ci]
struct GWgradParams {
  const float* x; int xs; const float* dy; int ds;
  int N, H, W, OH, OW, cin, cout, kh, kw, stride, pad, KS, cit;
  float* partial;
};

__global__ __launch_bounds__(256) void gconv_bwd_weight_kernel(const GWgradParams p) {
  __shared__ float red[4][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int cot = blockIdx.x / p.cit, cit = blockIdx.x % p.cit;
  const int tap = blockIdx.y, ky = tap / p.kw, kx = tap % p.kw, ks = blockIdx.z;
  const int co = cot * 32 + i, ci = cit * 32 + i;
  const long long npix = (long long)p.N * p.OH * p.OW;
  const long long per = (npix + p.KS - 1) / p.KS;
  const long long k0 = (long long)ks * per, k1 = k0 + per < npix ? k0 + per : npix;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // this wave takes the pixel pairs k0 + 2 * (wave + 4 j); a lane's k index within the pair is its half-wave (uniform trip count)
  for (long long kp = k0 + 2 * wave; kp < k1; kp += 8) {
    const long long k = kp + h;
    const bool kv = k < k1;
    const long long kc = kv ? k : 0;
    const int ox = kc % p.OW, oy = (kc / p.OW) % p.OH, n = kc / ((long long)p.OW * p.OH);
    const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
    const bool ok = kv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const float a = (kv && co < p.cout) ? p.dy[kc * p.ds + co] : 0.f;
    const float b = (ok && ci < p.cin) ? p.x[((long long)(n * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * p.xs + ci] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // D[row = co][col = ci]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][frag_row(r, h)][i] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += 256) {
    const int ro = e >> 5, cl = e & 31;
    const float s = (red[0][ro][cl] + red[1][ro][cl]) + (red[2][ro][cl] + red[3][ro][cl]);
    const int oc = cot * 32 + ro, ic = cit * 32 + cl;
    if (oc < p.cout && ic < p.cin) p.partial[(((long long)ks * p.kh * p.kw + tap) * p.cout + oc) * p.cin + ic] = s;
  }
}

__global__ void gconv_wgrad_reduce_kernel(const float* __restrict__ partial, int KS, int taps, int cout, int cin, float* __restrict__ dw, int accumulate) {
  const long long slab = (long long)taps * cout * cin;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= slab) return;
  const int ic = e % cin, oc = (e / cin) % cout, tap = e / ((long long)cin * cout);
  float s = 0.f;
  for (int k = 0; k < KS; ++k) s += partial[(long long)k * slab + e];
  float* d = dw + ((long long)oc * cin + ic) * taps + tap;
  *d = accumulate ? *d + s : s;
}

// column sums over pixels (bias gradient): out[c] (+)= sum_p x[p][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int xs, long long pixels, int C, float* __restrict__ out, int accumulate) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0;
  for (long long q = threadIdx.x; q < pixels; q += 256) s += (double)x[q * xs + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && c < C) out[c] = accumulate ? out[c] + (float)red[0] : (float)red[0];
}

// y = x > 0 ? x : slope * x (forward);  dx = dy * (y > 0 ? 1 : slope) (backward, from the OUTPUT's sign: slope > 0 keeps the sign)
__global__ __launch_bounds__(256) void leaky_kernel(const float* __restrict__ x, const float* __restrict__ ref, long long n4, float slope, float* __restrict__ y) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[idx];
    const f32x4 s = ref ? reinterpret_cast<const f32x4*>(ref)[idx] : v;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = s[j] > 0.f ? v[j] : v[j] * slope;
    reinterpret_cast<f32x4*>(y)[idx] = o;
  }
}

// dx = dy * y * (1 - y): the sigmoid's backward from its OUTPUT (the head convolutions' fused sigmoid epilogue), in the order torch evaluates it
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n4, float* __restrict__ dx) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
    const f32x4 g = reinterpret_cast<const f32x4*>(dy)[idx], v = reinterpret_cast<const f32x4*>(y)[idx];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (g[j] * v[j]) * (1.0f - v[j]);
    reinterpret_cast<f32x4*>(dx)[idx] = o;
  }
}

bool okt(const sfTensor& t) { return t.ptr && t.dtype == SF_F32 && ((uintptr_t)t.ptr & 15) == 0 && t.stride % 8 == 0 && t.c % 8 == 0; }

}  // namespace

extern "C" {

int sf_conv2d_fwd(sfTensor x, int32_t n, int32_t h, int32_t w, const float* weight, const float* bias, int32_t cin, int32_t cout, int32_t kh,
                  int32_t kw, int32_t stride, int32_t pad, float leaky_slope, sfTensor y, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_conv2d_fwd: exact-fp32 kernel only (dtype %d)", dtype);
  SF_REQUIRE(okt(x) && okt(y) && weight, "sf_conv2d_fwd: fp32 NHWC tensors with 16-byte aligned pixels, channel counts padded to 8");
  SF_REQUIRE(kh >= 1 && kw >= 1 && stride >= 1 && pad >= 0 && cin >= 1 && cout >= 1 && x.c >= (cin + 7) / 8 * 8 && y.c >= cout, "sf_conv2d_fwd: geometry");
  const int oh = (h + 2 * pad - kh) / stride + 1, ow = (w + 2 * pad - kw) / stride + 1;
  SF_REQUIRE(oh >= 1 && ow >= 1, "sf_conv2d_fwd: empty output (%dx%d)", oh, ow);
  GConvParams p{(const float*)x.ptr, x.stride, weight, bias, (float*)y.ptr, y.stride, y.c, n, h, w, oh, ow, cin, cout, kh, kw, stride, pad, leaky_slope};
  const long long npix = (long long)n * oh * ow;
  if (npix == 0) return 0;
  hipLaunchKernelGGL(gconv_fwd_kernel, dim3((unsigned)((npix + 127) / 128), (y.c + 31) / 32), dim3(256), 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("conv2d_fwd");
  return 0;
}

int sf_conv2d_bwd_data(sfTensor dy, int32_t n, int32_t h, int32_t w, const float* weight, int32_t cin, int32_t cout, int32_t kh, int32_t kw,
                       int32_t stride, int32_t pad, sfTensor dx, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_conv2d_bwd_data: exact-fp32 kernel only (dtype %d)", dtype);
  SF_REQUIRE(okt(dy) && okt(dx) && weight && dy.c >= (cout + 7) / 8 * 8 && dx.c >= cin, "sf_conv2d_bwd_data: tensors");
  const int oh = (h + 2 * pad - kh) / stride + 1, ow = (w + 2 * pad - kw) / stride + 1;
  GConvParams p{(const float*)dy.ptr, dy.stride, weight, nullptr, (float*)dx.ptr, dx.stride, dx.c, n, h, w, oh, ow, cin, cout, kh, kw, stride, pad, 1.f};
  const long long npix = (long long)n * h * w;
  if (npix == 0) return 0;
  hipLaunchKernelGGL(gconv_bwd_data_kernel, dim3((unsigned)((npix + 127) / 128), (dx.c + 31) / 32), dim3(256), 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("conv2d_bwd_data");
  return 0;
}

static int gconv_ks(long long npix, int blocks) {
  long long ks = 1024 / (blocks > 0 ? blocks : 1);
  if (ks < 1) ks = 1;
  if (ks > 64) ks = 64;
  const long long maxks = (npix + 255) / 256;  // at least 256 pixels per slice
  if (ks > maxks) ks = maxks < 1 ? 1 : maxks;
  return (int)ks;
}

size_t sf_conv2d_bwd_weight_workspace_bytes(int32_t n, int32_t oh, int32_t ow, int32_t cin, int32_t cout, int32_t kh, int32_t kw) {
  const int blocks = ((cout + 31) / 32) * ((cin + 31) / 32) * kh * kw;
  return (size_t)gconv_ks((long long)n * oh * ow, blocks) * kh * kw * cout * cin * sizeof(float);
}

int sf_conv2d_bwd_weight(sfTensor x, sfTensor dy, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw,
                         int32_t stride, int32_t pad, float* dw, float* db, int32_t accumulate, void* workspace, size_t workspace_bytes,
                         int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_conv2d_bwd_weight: exact-fp32 kernel only (dtype %d)", dtype);
  SF_REQUIRE(okt(x) && okt(dy) && dw && x.c >= cin && dy.c >= cout, "sf_conv2d_bwd_weight: tensors");
  const int oh = (h + 2 * pad - kh) / stride + 1, ow = (w + 2 * pad - kw) / stride + 1;
  const long long npix = (long long)n * oh * ow;
  GWgradParams p{(const float*)x.ptr, x.stride, (const float*)dy.ptr, dy.stride, n, h, w, oh, ow, cin, cout, kh, kw, stride, pad, 1, (cin + 31) / 32, (float*)workspace};
  const int blocks = ((cout + 31) / 32) * p.cit * kh * kw;
  p.KS = gconv_ks(npix, blocks);
  SF_REQUIRE(workspace && workspace_bytes >= (size_t)p.KS * kh * kw * cout * cin * sizeof(float), "sf_conv2d_bwd_weight: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gconv_bwd_weight_kernel, dim3(((cout + 31) / 32) * p.cit, kh * kw, p.KS), dim3(256), 0, st, p);
  SF_CHECK_LAUNCH("conv2d_bwd_weight");
  const long long slab = (long long)kh * kw * cout * cin;
  hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, p.partial, p.KS, kh * kw, cout, cin, dw, accumulate);
  SF_CHECK_LAUNCH("conv2d_wgrad_reduce");
  if (db) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cout), dim3(256), 0, st, (const float*)dy.ptr, dy.stride, npix, cout, db, accumulate);
    SF_CHECK_LAUNCH("conv2d_bias_grad");
  }
  return 0;
}

int sf_leaky_relu(const float* x, const float* sign_ref, int64_t n, float slope, float* y, sfStream stream) {
  SF_REQUIRE(n % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!sign_ref || ((uintptr_t)sign_ref & 15) == 0), "sf_leaky_relu: 16-byte aligned, n %% 4 == 0");
  if (n == 0) return 0;
  const long long n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(leaky_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, sign_ref, n4, slope, y);
  SF_CHECK_LAUNCH("leaky_relu");
  return 0;
}

int sf_sigmoid_bwd(const float* dy, const float* y, int64_t n, float* dx, sfStream stream) {
  SF_REQUIRE(dy && y && dx && n >= 0 && n % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)dx & 15) == 0,
             "sf_sigmoid_bwd: 16-byte aligned, n %% 4 == 0");
  if (n == 0) return 0;
  const long long n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, y, n4, dx);
  SF_CHECK_LAUNCH("sigmoid_bwd");
  return 0;
}

}  // extern "C"
