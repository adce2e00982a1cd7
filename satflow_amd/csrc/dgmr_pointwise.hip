// HBM-bound stages of the in-tree DGMR / DVD-GAN style networks: everything of
//   satflow/models/layers/Normalization.py (SpectralNorm :10-62, ConditionalNorm :65-85),
//   satflow/models/layers/GResBlock.py:55-99, satflow/models/layers/Discriminator.py:200-314 / :368-478,
//   satflow/models/layers/Generator.py:75-131
// that is not a convolution (those run on the MFMA kernels: sf_conv3x3_*, sf_conv2d_*, sf_linear_*):
//   spectral normalisation (power iteration, sigma, W / sigma and its backward), 2x2 (x2 frames) sum pooling and nearest expansion
//   with a scale (avg_pool2d / avg_pool3d / F.interpolate and their backward passes), the three temporal taps of a Conv3d as a
//   channel stack, conditional BatchNorm -> ReLU -> up-sampling in one pass, ReLU + sum over pixels (discriminator head),
//   gamma * attention + x, tanh, and the gate arithmetic of the generator's ConvGRU.
// All activations NHWC fp32 with 16-byte channel quads.
#include "sf_common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
#define kZero4 (f32x4{0.f, 0.f, 0.f, 0.f})

inline int grid_of(long long work, int cap = 16384) {
  long long b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline bool okd(const sfTensor& t, int lanes) {  // dense fp32 tensor of `lanes` channels, 16-byte aligned pixels
  return t.ptr && t.dtype == SF_F32 && ((uintptr_t)t.ptr & 15) == 0 && t.c == lanes && t.stride == lanes && lanes % 4 == 0;
}
inline bool oks(const sfTensor& t, int lanes) {  // strided view: at least `lanes` channels
  return t.ptr && t.dtype == SF_F32 && ((uintptr_t)t.ptr & 15) == 0 && t.c >= lanes && t.stride >= t.c && t.stride % 4 == 0 && lanes % 4 == 0;
}

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads; result in every thread
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return s;
}

// ---- spectral normalisation (Normalization.py:19-31) ------------------------------------------------------------------
// vraw[j] = sum_i W[i][j] u[i]
// grid (column blocks, row slices): slice r sums its rows into part[r][j]; sn_sum_slices_kernel adds the slices in a fixed order
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ W, int h, int w, const float* __restrict__ u, float* __restrict__ part) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= w) return;
  const int per = (h + gridDim.y - 1) / gridDim.y, i0 = blockIdx.y * per, i1 = i0 + per < h ? i0 + per : h;
  float s = 0.f;
  for (int i = i0; i < i1; ++i) s = __builtin_fmaf(W[(size_t)i * w + j], u[i], s);
  part[(size_t)blockIdx.y * w + j] = s;
}
__global__ __launch_bounds__(256) void sn_sum_slices_kernel(const float* __restrict__ part, int slices, int w, float* __restrict__ vraw) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= w) return;
  float s = 0.f;
  for (int r = 0; r < slices; ++r) s += part[(size_t)r * w + j];
  vraw[j] = s;
}
// v = vraw / (|vraw| + eps) (block 0 stores it);  uraw[i] = sum_j W[i][j] v[j]   - one workgroup per row
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ W, int h, int w, const float* __restrict__ vraw, float* __restrict__ v,
                                                     float* __restrict__ uraw) {
  __shared__ float red[4];
  float q = 0.f;
  for (int j = threadIdx.x; j < w; j += 256) q = __builtin_fmaf(vraw[j], vraw[j], q);
  const float den = sqrtf(block_sum(q, red)) + 1e-12f;
  const int i = blockIdx.x;
  float s = 0.f;
  for (int j = threadIdx.x; j < w; j += 256) {
    const float vj = vraw[j] / den;
    if (i == 0) v[j] = vj;
    s = __builtin_fmaf(W[(size_t)i * w + j], vj, s);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) uraw[i] = s;
}
// u = uraw / (|uraw| + eps) (in place into the caller's u; only between power iterations)
__global__ __launch_bounds__(256) void sn_norm_u_kernel(const float* __restrict__ uraw, int h, float* __restrict__ u) {
  __shared__ float red[4];
  float q = 0.f;
  for (int i = threadIdx.x; i < h; i += 256) q = __builtin_fmaf(uraw[i], uraw[i], q);
  const float den = sqrtf(block_sum(q, red)) + 1e-12f;
  for (int i = threadIdx.x; i < h; i += 256) u[i] = uraw[i] / den;
}
// u = uraw / (|uraw| + eps), sigma = u . uraw ( = u . (W v) ), Wout = W / sigma
__global__ __launch_bounds__(256) void sn_finish_kernel(const float* __restrict__ W, int h, int w, const float* __restrict__ uraw, float* __restrict__ u,
                                                        float* __restrict__ sigma_out, float* __restrict__ Wout) {
  __shared__ float red[4];
  float q = 0.f;
  for (int i = threadIdx.x; i < h; i += 256) q = __builtin_fmaf(uraw[i], uraw[i], q);
  const float den = sqrtf(block_sum(q, red)) + 1e-12f;
  float d = 0.f;
  for (int i = threadIdx.x; i < h; i += 256) {
    const float ui = uraw[i] / den;
    if (blockIdx.x == 0) u[i] = ui;
    d = __builtin_fmaf(ui, uraw[i], d);
  }
  const float sigma = block_sum(d, red);
  if (blockIdx.x == 0 && threadIdx.x == 0) *sigma_out = sigma;
  const long long n = (long long)h * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) Wout[e] = W[e] / sigma;
}
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) s = __builtin_fmaf(a[e], b[e], s);
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// dWbar = g / sigma - (<g, Wbar> / sigma^2) u v^T
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v,
                                                     const float* __restrict__ sigma, const float* __restrict__ partial, int nparts, int h, int w,
                                                     float* __restrict__ dW) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
  const float dot = block_sum(s, red);
  const float sg = *sigma;
  const float c = dot / (sg * sg);
  const long long n = (long long)h * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / w), j = (int)(e - (long long)i * w);
    dW[e] = g[e] / sg - c * u[i] * v[j];
  }
}

// ---- 2x2 (x tp frames) sum pooling / nearest expansion ---------------------------------------------------------------
// y[o][yy][xx] = scale * sum over the 2x2 window (and tp frames) of x + addend;  image o = k * nb + b reads frames tp*k .. tp*k + tp-1
__global__ __launch_bounds__(256) void pool2_kernel(const float* __restrict__ x, int xs, long long n_out, int oh, int ow, int tp, int nb, float scale,
                                                    const float* __restrict__ add, int as, float* __restrict__ y, int ys, int q) {
  const long long total = n_out * oh * ow * q;
  const int h = oh * 2, w = ow * 2;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int xx = (int)(r % ow); r /= ow;
    const int yy = (int)(r % oh); const long long o = r / oh;
    const long long k = o / nb, b = o - k * nb;
    f32x4 s = kZero4;
    for (int f = 0; f < tp; ++f) {
      const float* base = x + ((((k * tp + f) * nb + b) * h + 2 * yy) * (long long)w + 2 * xx) * xs + c;
      s += (ld4(base) + ld4(base + xs)) + (ld4(base + (long long)w * xs) + ld4(base + (long long)w * xs + xs));
    }
    s *= scale;
    const long long pix = (o * oh + yy) * ow + xx;
    if (add) s += ld4(add + pix * as + c);
    st4(y + pix * ys + c, s);
  }
}
// y[image][2 yy + a][2 xx + b] = scale * x[image'][yy][xx], every output element once
__global__ __launch_bounds__(256) void expand2_kernel(const float* __restrict__ x, int xs, long long n_in, int h, int w, int tp, int nb, float scale,
                                                      float* __restrict__ y, int ys, int q) {
  const int oh = 2 * h, ow = 2 * w;
  const long long total = n_in * tp * oh * ow * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int ox = (int)(r % ow); r /= ow;
    const int oy = (int)(r % oh); const long long o = r / oh;  // output image = (k * tp + f) * nb + b
    const long long kf = o / nb, b = o - kf * nb, k = kf / tp;
    const f32x4 v = ld4(x + (((k * nb + b) * h + (oy >> 1)) * (long long)w + (ox >> 1)) * xs + c) * scale;
    st4(y + ((o * oh + oy) * ow + ox) * ys + c, v);
  }
}

// ---- Conv3d temporal taps as a channel stack -------------------------------------------------------------------------
// y[t][b][p][dt * C + c] = x[t + dt - 1][b][p][c] (zero outside 0 <= t + dt - 1 < T);  per = nb * pixels per image
__global__ __launch_bounds__(256) void tstack3_fwd_kernel(const float* __restrict__ x, int T, long long per, int q, float* __restrict__ y) {
  const long long total = (long long)T * per * 3 * q;
  const int C = q * 4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int dt = (int)(r % 3); r /= 3;
    const long long p = r % per; const int t = (int)(r / per);
    const int ts = t + dt - 1;
    const f32x4 v = (ts >= 0 && ts < T) ? ld4(x + ((long long)ts * per + p) * C + c) : kZero4;
    st4(y + ((long long)t * per + p) * 3 * C + dt * C + c, v);
  }
}
// gx[t][..][c] = gy[t + 1][..][c] + gy[t][..][C + c] + gy[t - 1][..][2 C + c]
__global__ __launch_bounds__(256) void tstack3_bwd_kernel(const float* __restrict__ gy, int T, long long per, int q, float* __restrict__ gx) {
  const long long total = (long long)T * per * q;
  const int C = q * 4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    const long long r = idx / q;
    const long long p = r % per; const int t = (int)(r / per);
    f32x4 s = ld4(gy + ((long long)t * per + p) * 3 * C + C + c);
    if (t + 1 < T) s += ld4(gy + ((long long)(t + 1) * per + p) * 3 * C + c);
    if (t >= 1) s += ld4(gy + ((long long)(t - 1) * per + p) * 3 * C + 2 * C + c);
    st4(gx + r * C + c, s);
  }
}


// ---- a 5x5 'same' convolution as ONE 3x3 convolution ------------------------------------------------------------------
// The 5x5 kernel is covered by four 3x3 tiles at offsets {0, 2}^2 (the shared middle row / column zeroed in the second tile), each
// reading the input shifted by (2 ty - 1, 2 tx - 1).  On a domain padded by 2 on every side no tap of an interior output touches
// the 3x3 kernel's own zero padding, so the shifted copies can be stacked as channels of ONE tensor:
//   ys[n][q][s * C + c] = x[n][q - 2 + d_s][c]  (zero outside the image),  q in [0, H + 4) x [0, W + 4),  d_s = (2 ty - 1, 2 tx - 1)
// and the 5x5 result is the interior of conv3x3(ys, W3).  pad_shift4_bwd is the adjoint; crop / zero-pad kernels cut the border.
__global__ __launch_bounds__(256) void pad_shift4_fwd_kernel(const float* __restrict__ x, long long n, int h, int w, int q, float* __restrict__ y) {
  const int H4 = h + 4, W4 = w + 4, C = q * 4;
  const long long total = n * H4 * W4 * 4 * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int s = (int)(r % 4); r /= 4;
    const int xx = (int)(r % W4); r /= W4;
    const int yy = (int)(r % H4); const long long img = r / H4;
    const int sy = yy - 2 + 2 * (s >> 1) - 1, sx = xx - 2 + 2 * (s & 1) - 1;
    const f32x4 v = (sy >= 0 && sy < h && sx >= 0 && sx < w) ? ld4(x + ((img * h + sy) * (long long)w + sx) * C + c) : kZero4;
    st4(y + ((img * H4 + yy) * (long long)W4 + xx) * 4 * C + s * C + c, v);
  }
}
__global__ __launch_bounds__(256) void pad_shift4_bwd_kernel(const float* __restrict__ gy, long long n, int h, int w, int q, float* __restrict__ gx) {
  const int H4 = h + 4, W4 = w + 4, C = q * 4;
  const long long total = n * h * w * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int sx = (int)(r % w); r /= w;
    const int sy = (int)(r % h); const long long img = r / h;
    f32x4 a = kZero4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // the padded position that read this pixel through shift s (always inside the padded domain)
      const int yy = sy + 2 - (2 * (s >> 1) - 1), xx = sx + 2 - (2 * (s & 1) - 1);
      a += ld4(gy + ((img * H4 + yy) * (long long)W4 + xx) * 4 * C + s * C + c);
    }
    st4(gx + idx * 4, a);
  }
}
// Weight of the ONE 3x3 convolution that equals a 5x5 convolution over the four shifted copies (pad_shift4 above): tile (ty, tx) = copy 2 ty + tx holds
// the 5x5 taps (2 ty + ky, 2 tx + kx); the middle row / column of the 5x5 kernel is covered twice and belongs to the upper / left tile.
//   w3[o][(2 ty + tx) * lanes + i][ky][kx] = w5[o][i][2 ty + ky][2 tx + kx]   (0 for i >= I and for the masked duplicates)
// w5 may be a column slice of a wider weight: row pitch `so` elements.  The backward pass is the inverse gather (every 5x5 tap has one owner).
__global__ __launch_bounds__(256) void regroup5_fwd_kernel(const float* __restrict__ w5, long long so, int O, int I, int lanes, float* __restrict__ w3) {
  const long long total = (long long)O * 4 * lanes * 9;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int k = (int)(idx % 9); long long r = idx / 9;
    const int i = (int)(r % lanes); r /= lanes;
    const int t = (int)(r % 4); const long long o = r / 4;
    const int ty = t >> 1, tx = t & 1, ky = k / 3, kx = k - 3 * ky;
    const bool live = i < I && !(ty && ky == 0) && !(tx && kx == 0);
    w3[idx] = live ? w5[o * so + (long long)i * 25 + (2 * ty + ky) * 5 + 2 * tx + kx] : 0.f;
  }
}
__global__ __launch_bounds__(256) void regroup5_bwd_kernel(const float* __restrict__ g3, int O, int I, int lanes, float* __restrict__ g5) {
  const long long total = (long long)O * I * 25;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int x = (int)(idx % 5); long long r = idx / 5;
    const int y = (int)(r % 5); r /= 5;
    const int i = (int)(r % I); const long long o = r / I;
    const int ty = y > 2 ? 1 : 0, tx = x > 2 ? 1 : 0;   // rows / columns 0..2 belong to the upper / left tile
    g5[idx] = g3[((o * 4 + 2 * ty + tx) * lanes + i) * 9 + (y - 2 * ty) * 3 + (x - 2 * tx)];
  }
}
// ---- the same 5x5 'same' convolution as ONE 3x3 convolution on the HALF-RESOLUTION domain (the ConvGRU sequence path) ---------------
// Fold the 2x2 pixel blocks of input AND output into channels (a pure permutation, sf_space_to_depth2):
//   xs[n][Y][X][(2 py + px) * C + c] = x[n][2 Y + py][2 X + px][c]
// A 5x5 tap (dy, dx) in [-2, 2]^2 of output phase (py, px) reads block (Y + ey, X + ex), phase (qy, qx) with py + dy = 2 ey + qy: ey in {-1, 0, 1},
// i.e. a 3x3 'same' convolution from 4C to 4O channels whose zero padding IS the 5x5 convolution's.  Against the four-shifted-copies route above:
// the same 36 / 25 tap overhead, but no domain padded by 2 (68 x 68 pixels in 16 x 32 tiles run at 60 %), no copy written four times per call and
// no crop - and since every pointwise stage of a recurrent cell is layout-blind the WHOLE sequence stays in this layout (two permutations per cell).
//   w3[(g * 4 + po) * hp + o][pi * lanes + i][ey + 1][ex + 1] = w5[g * hp + o][i][dy + 2][dx + 2],  dy = 2 ey + qy - py (0 where |dy| > 2, i >= I)
// (rows: `hp` per gate block, gate-major then output phase: the z | r halves of a GRU's gate tensor stay halves).
__global__ __launch_bounds__(256) void s2d2_kernel(const float* __restrict__ x, long long n, int H2, int W2, int q, float* __restrict__ y, int inverse) {
  const int C = q * 4;
  const long long total = n * H2 * W2 * 4 * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int s = (int)(r % 4); r /= 4;
    const int X = (int)(r % W2); r /= W2;
    const int Y = (int)(r % H2); const long long img = r / H2;
    const long long full = ((img * (2 * H2) + 2 * Y + (s >> 1)) * (long long)(2 * W2) + 2 * X + (s & 1)) * C + c;
    if (inverse) st4(y + full, ld4(x + idx * 4));
    else st4(y + idx * 4, ld4(x + full));
  }
}
__global__ __launch_bounds__(256) void regroup5_s2d_fwd_kernel(const float* __restrict__ w5, long long so, int R, int hp, int I, int lanes, float* __restrict__ w3) {
  const long long total = (long long)R * 4 * 4 * lanes * 9;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int k = (int)(idx % 9); long long r = idx / 9;
    const int i = (int)(r % lanes); r /= lanes;
    const int pi = (int)(r % 4); r /= 4;
    const int o = (int)(r % hp); r /= hp;
    const int po = (int)(r % 4); const long long g = r / 4;
    const int dy = 2 * (k / 3 - 1) + (pi >> 1) - (po >> 1), dx = 2 * (k % 3 - 1) + (pi & 1) - (po & 1);
    const bool live = i < I && dy >= -2 && dy <= 2 && dx >= -2 && dx <= 2;
    w3[idx] = live ? w5[(g * hp + o) * so + (long long)i * 25 + (dy + 2) * 5 + dx + 2] : 0.f;
  }
}
__global__ __launch_bounds__(256) void regroup5_s2d_bwd_kernel(const float* __restrict__ g3, int R, int hp, int I, int lanes, float* __restrict__ g5) {
  const long long total = (long long)R * I * 25;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int x = (int)(idx % 5); long long r = idx / 5;
    const int y = (int)(r % 5); r /= 5;
    const int i = (int)(r % I); const long long row = r / I;
    const long long g = row / hp; const int o = (int)(row % hp);
    float a = 0.f;
#pragma unroll
    for (int po = 0; po < 4; ++po) {  // every output phase reads this tap once: py + dy = 2 ey + qy
      const int ty = (po >> 1) + y - 2, tx = (po & 1) + x - 2;
      const int ey = (ty + 2) / 2 - 1, ex = (tx + 2) / 2 - 1;   // floor(t / 2) for t in [-2, 3]
      const int pi = 2 * (ty - 2 * ey) + (tx - 2 * ex);
      a += g3[((((g * 4 + po) * hp + o) * 4 + pi) * lanes + i) * 9 + (ey + 1) * 3 + ex + 1];
    }
    g5[idx] = a;
  }
}
// 4x4 stride-2 convolution with padding 1 (the PatchGAN discriminator's down-sampling layers, gan/discriminators.py:166-197) on the 3x3 kernels:
// pad by 1 and fold 2x2 pixel blocks into channels,
//   ys[n][Y][X][(2 dy + dx) * C + c] = x[n][2 Y + dy - 1][2 X + dx - 1][c]  (zero outside the image),  Y in [0, h/2 + 1), X in [0, w/2 + 1)
// then out[Y][X] = sum_{a,b in {0,1}} W'[a][b] . ys[Y + a][X + b] with W'[a][b][(dy, dx, c)] = W[c][2a + dy][2b + dx]: a 2x2 convolution, i.e. the
// 3x3 "same" convolution whose taps (1 + a, 1 + b) hold W' and whose other five taps are zero; the first h/2 x w/2 outputs are the result.
__global__ __launch_bounds__(256) void pad_s2d_fwd_kernel(const float* __restrict__ x, long long n, int h, int w, int q, float* __restrict__ y) {
  const int H2 = h / 2 + 1, W2 = w / 2 + 1, C = q * 4;
  const long long total = n * H2 * W2 * 4 * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int s = (int)(r % 4); r /= 4;
    const int X = (int)(r % W2); r /= W2;
    const int Y = (int)(r % H2); const long long img = r / H2;
    const int sy = 2 * Y + (s >> 1) - 1, sx = 2 * X + (s & 1) - 1;
    const f32x4 v = (sy >= 0 && sy < h && sx >= 0 && sx < w) ? ld4(x + ((img * h + sy) * (long long)w + sx) * C + c) : kZero4;
    st4(y + ((img * H2 + Y) * (long long)W2 + X) * 4 * C + s * C + c, v);
  }
}
__global__ __launch_bounds__(256) void pad_s2d_bwd_kernel(const float* __restrict__ gy, long long n, int h, int w, int q, float* __restrict__ gx) {
  const int H2 = h / 2 + 1, W2 = w / 2 + 1, C = q * 4;
  const long long total = n * h * w * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int sx = (int)(r % w); r /= w;
    const int sy = (int)(r % h); const long long img = r / h;
    const int Y = (sy + 1) >> 1, X = (sx + 1) >> 1, s = ((sy + 1) & 1) * 2 + ((sx + 1) & 1);   // every pixel sits in exactly one block
    st4(gx + idx * 4, ld4(gy + ((img * H2 + Y) * (long long)W2 + X) * 4 * C + s * C + c));
  }
}
// y[n][yy][xx] = x[n][yy + b][xx + b] (crop) or, with pad != 0, y = x inside and 0 on a border of width b (the adjoint)
__global__ __launch_bounds__(256) void border_kernel(const float* __restrict__ x, long long n, int h, int w, int b, int q, int pad, float* __restrict__ y) {
  const int C = q * 4;
  if (!pad) {  // x [h + 2b][w + 2b] -> y [h][w]
    const long long total = n * h * w * q;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
      const int c = (int)(idx % q) * 4;
      long long r = idx / q;
      const int xx = (int)(r % w); r /= w;
      const int yy = (int)(r % h); const long long img = r / h;
      st4(y + idx * 4, ld4(x + ((img * (h + 2 * b) + yy + b) * (long long)(w + 2 * b) + xx + b) * C + c));
    }
  } else {     // x [h][w] -> y [h + 2b][w + 2b]
    const int H2 = h + 2 * b, W2 = w + 2 * b;
    const long long total = n * H2 * W2 * q;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
      const int c = (int)(idx % q) * 4;
      long long r = idx / q;
      const int xx = (int)(r % W2) - b; r /= W2;
      const int yy = (int)(r % H2) - b; const long long img = r / H2;
      const f32x4 v = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? ld4(x + ((img * h + yy) * (long long)w + xx) * C + c) : kZero4;
      st4(y + idx * 4, v);
    }
  }
}


// ---- nn.MaxPool3d over NHWC tokens [B][D0][D1][D2][C] (Attention.py:50,127: kernel 2 or (2,1,1), stride pooling_factor) --------
struct Pool3 { int d0, d1, d2, o0, o1, o2, k0, k1, k2, s0, s1, s2; };
__global__ __launch_bounds__(256) void maxpool3_fwd_kernel(const float* __restrict__ x, long long B, Pool3 g, int q, float* __restrict__ y) {
  const int C = q * 4;
  const long long total = B * g.o0 * g.o1 * g.o2 * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int c2 = (int)(r % g.o2); r /= g.o2;
    const int c1 = (int)(r % g.o1); r /= g.o1;
    const int c0 = (int)(r % g.o0); const long long b = r / g.o0;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int a0 = 0; a0 < g.k0; ++a0)
      for (int a1 = 0; a1 < g.k1; ++a1)
        for (int a2 = 0; a2 < g.k2; ++a2) {
          const f32x4 v = ld4(x + ((((b * g.d0 + c0 * g.s0 + a0) * g.d1 + c1 * g.s1 + a1) * (long long)g.d2) + c2 * g.s2 + a2) * C + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
        }
    st4(y + idx * 4, m);
  }
}
// gradient to the FIRST maximum of every window in scan order (as torch); windows may overlap only when stride < kernel, which the
// callers never use (stride >= kernel is required), so every input element belongs to at most one window: no atomics
__global__ __launch_bounds__(256) void maxpool3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, long long B, Pool3 g, int q,
                                                           float* __restrict__ gx) {
  const int C = q * 4;
  const long long total = B * g.d0 * g.d1 * g.d2 * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int i2 = (int)(r % g.d2); r /= g.d2;
    const int i1 = (int)(r % g.d1); r /= g.d1;
    const int i0 = (int)(r % g.d0); const long long b = r / g.d0;
    const int c0 = i0 / g.s0, c1 = i1 / g.s1, c2 = i2 / g.s2;
    const int a0 = i0 - c0 * g.s0, a1 = i1 - c1 * g.s1, a2 = i2 - c2 * g.s2;
    f32x4 o = kZero4;
    if (c0 < g.o0 && c1 < g.o1 && c2 < g.o2 && a0 < g.k0 && a1 < g.k1 && a2 < g.k2) {
      const f32x4 mine = ld4(x + idx * 4);
      bool first[4] = {true, true, true, true};
      for (int b0 = 0; b0 < g.k0; ++b0)
        for (int b1 = 0; b1 < g.k1; ++b1)
          for (int b2 = 0; b2 < g.k2; ++b2) {
            if (b0 == a0 && b1 == a1 && b2 == a2) continue;
            const f32x4 v = ld4(x + ((((b * g.d0 + c0 * g.s0 + b0) * g.d1 + c1 * g.s1 + b1) * (long long)g.d2) + c2 * g.s2 + b2) * C + c);
            const bool before = (b0 * g.k1 + b1) * g.k2 + b2 < (a0 * g.k1 + a1) * g.k2 + a2;
#pragma unroll
            for (int j = 0; j < 4; ++j) first[j] = first[j] && (before ? v[j] < mine[j] : v[j] <= mine[j]);
          }
      const f32x4 gv = ld4(gy + ((((b * g.o0 + c0) * g.o1 + c1) * (long long)g.o2) + c2) * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = first[j] ? gv[j] : 0.f;
    }
    st4(gx + idx * 4, o);
  }
}

// ---- conditional BatchNorm (+ ReLU, + nearest x2) --------------------------------------------------------------------
// y = act(gamma[n][c] * (x - mean[c]) * rstd[c] + beta[n][c]), stored once or to the 2x2 block of the up-sampled map
__global__ __launch_bounds__(256) void film_fwd_kernel(const float* __restrict__ x, long long n, int h, int w, int q, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ embed, int creal, int relu, int up,
                                                       float* __restrict__ y) {
  const long long total = n * h * w * q;
  const int C = q * 4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    long long r = idx / q;
    const int xx = (int)(r % w); r /= w;
    const int yy = (int)(r % h); const long long img = r / h;
    const f32x4 v = ld4(x + idx * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cj = c + j;
      float z = 0.f;
      if (cj < creal) {
        const float xh = (v[j] - mean[cj]) * rstd[cj];
        z = embed[img * 2 * creal + cj] * xh + embed[img * 2 * creal + creal + cj];
        if (relu) z = fmaxf(z, 0.f);
      }
      o[j] = z;
    }
    if (!up) st4(y + idx * 4, o);
    else {
      float* d = y + (((img * 2 * h + 2 * yy) * (long long)(2 * w)) + 2 * xx) * C + c;
      st4(d, o); st4(d + C, o); st4(d + (long long)2 * w * C, o); st4(d + (long long)2 * w * C + C, o);
    }
  }
}
// dz = mask * (gy, summed over the 2x2 block when up);  dxhat = dz * gamma;  per (image, channel): dgamma = sum dz * xhat, dbeta = sum dz
// grid (slices, n, ceil(q / 16)); thread = (pixel lane 0..15, channel quad 0..15)
__global__ __launch_bounds__(256) void film_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, int h, int w, int q,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ embed,
                                                       int creal, int relu, int up, float* __restrict__ dxhat, float* __restrict__ partial) {
  __shared__ float red[2][16][16][4];
  const int C = q * 4;
  const int ql = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int qq = blockIdx.z * 16 + ql;
  const long long img = blockIdx.y;
  const int slices = gridDim.x, sl = blockIdx.x;
  const long long pixels = (long long)h * w;
  const long long per = (pixels + slices - 1) / slices, p0 = sl * per, p1 = p0 + per < pixels ? p0 + per : pixels;
  f32x4 ag = kZero4, ab = kZero4;
  if (qq < q) {
    const int c = qq * 4;
    float gm[4], bt[4], mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cj = c + j;
      const bool live = cj < creal;
      gm[j] = live ? embed[img * 2 * creal + cj] : 0.f; bt[j] = live ? embed[img * 2 * creal + creal + cj] : 0.f;
      mu[j] = live ? mean[cj] : 0.f; rs[j] = live ? rstd[cj] : 0.f;
    }
    for (long long p = p0 + pl; p < p1; p += 16) {
      const long long e = (img * pixels + p) * C + c;
      const f32x4 v = ld4(x + e);
      f32x4 g;
      if (!up) g = ld4(gy + e);
      else {
        const int yy = (int)(p / w), xx = (int)(p - (long long)yy * w);
        const float* s = gy + (((img * 2 * h + 2 * yy) * (long long)(2 * w)) + 2 * xx) * C + c;
        g = (ld4(s) + ld4(s + C)) + (ld4(s + (long long)2 * w * C) + ld4(s + (long long)2 * w * C + C));
      }
      f32x4 d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[j] - mu[j]) * rs[j];
        const float z = gm[j] * xh + bt[j];
        const float dz = (c + j < creal && (!relu || z > 0.f)) ? g[j] : 0.f;
        ag[j] = __builtin_fmaf(dz, xh, ag[j]); ab[j] += dz;
        d[j] = dz * gm[j];
      }
      st4(dxhat + e, d);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][pl][ql][j] = ag[j]; red[1][pl][ql][j] = ab[j]; }
  __syncthreads();
  if (threadIdx.x < 128) {  // (which, quad, j): sum the 16 pixel lanes
    const int which = threadIdx.x >> 6, ql2 = (threadIdx.x >> 2) & 15, j = threadIdx.x & 3;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[which][k][ql2][j];
    const int cj = (blockIdx.z * 16 + ql2) * 4 + j;
    if (cj < creal) partial[((long long)sl * gridDim.y + img) * 2 * creal + which * creal + cj] = s;
  }
}
__global__ void sum_slices_kernel(const float* __restrict__ partial, int slices, long long n, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int k = 0; k < slices; ++k) s += partial[(long long)k * n + e];
  out[e] = s;
}

// ---- ReLU + sum over the pixels of an image (discriminator head, Discriminator.py:286-293) -----------------------------
// grid (n, ceil(q / 16)); thread = (pixel lane, channel quad)
__global__ __launch_bounds__(256) void relu_sum_fwd_kernel(const float* __restrict__ x, long long pixels, int q, float* __restrict__ out) {
  __shared__ float red[16][16][4];
  const int C = q * 4, ql = threadIdx.x & 15, pl = threadIdx.x >> 4, qq = blockIdx.y * 16 + ql;
  const long long img = blockIdx.x;
  f32x4 a = kZero4;
  if (qq < q)
    for (long long p = pl; p < pixels; p += 16) {
      const f32x4 v = ld4(x + (img * pixels + p) * C + qq * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += fmaxf(v[j], 0.f);
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[pl][ql][j] = a[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int ql2 = threadIdx.x >> 2, j = threadIdx.x & 3;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][ql2][j];
    const int cj = (blockIdx.y * 16 + ql2) * 4 + j;
    if (cj < C) out[img * C + cj] = s;
  }
}
__global__ __launch_bounds__(256) void relu_sum_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, long long n, long long pixels, int q,
                                                           float* __restrict__ gx) {
  const long long total = n * pixels * q;
  const int C = q * 4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % q) * 4;
    const long long img = idx / q / pixels;
    const f32x4 v = ld4(x + idx * 4), gv = ld4(g + img * C + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.f ? gv[j] : 0.f;
    st4(gx + idx * 4, o);
  }
}

// ---- y = gamma * o + x (SelfAttention residual, Discriminator.py:125) and plain sums ----------------------------------
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ o, const float* __restrict__ x, const float* __restrict__ gamma, float alpha,
                                                   long long n4, float* __restrict__ y) {
  const float g = gamma ? *gamma : alpha;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (long long)gridDim.x * 256) {
    f32x4 v = ld4(o + idx * 4) * g;
    if (x) v += ld4(x + idx * 4);
    st4(y + idx * 4, v);
  }
}
__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = s;
}

// ---- tanh -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tanh_kernel(const float* __restrict__ x, const float* __restrict__ yref, long long n4, float* __restrict__ out) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (long long)gridDim.x * 256) {
    const f32x4 v = ld4(x + idx * 4);
    f32x4 o;
    if (!yref) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = sf_tanh(v[j]);
    } else {  // backward: x = dy, yref = y
      const f32x4 yv = ld4(yref + idx * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = v[j] * (1.f - yv[j] * yv[j]);
    }
    st4(out + idx * 4, o);
  }
}

// ---- gate arithmetic of the generator's ConvGRU -----------------------------------------------------------------------
// gates: z = sig(gx_z + gh_z), r = sig(gx_r + gh_r), rh = r * h      (gx, gh: [.., 2*H] = z | r pre-activation parts)
__global__ __launch_bounds__(256) void dvdgru_gates_fwd_kernel(const float* __restrict__ gx, int sx, const float* __restrict__ gh, int sh,
                                                               const float* __restrict__ h, int hs, long long pixels, int H, float* __restrict__ zr,
                                                               float* __restrict__ rh) {
  const int q = H >> 2;
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long pix = idx / q;
    const int c = (int)(idx - pix * q) * 4;
    f32x4 az = ld4(gx + pix * sx + c), ar = ld4(gx + pix * sx + H + c);
    if (gh) { az += ld4(gh + pix * sh + c); ar += ld4(gh + pix * sh + H + c); }
    const f32x4 hv = h ? ld4(h + pix * hs + c) : kZero4;
    f32x4 z, r, o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { z[j] = sf_sigmoid(az[j]); r[j] = sf_sigmoid(ar[j]); o[j] = r[j] * hv[j]; }
    st4(zr + pix * 2 * H + c, z); st4(zr + pix * 2 * H + H + c, r);
    st4(rh + pix * H + c, o);
  }
}
// in: dz (gradient wrt the gate z, nullable), drh (wrt r*h, nullable), saved zr, h  ->  dpre [.., 2H] (z | r pre-activations), dh (+= nothing: written)
__global__ __launch_bounds__(256) void dvdgru_gates_bwd_kernel(const float* __restrict__ dz, int dzs, const float* __restrict__ drh, const float* __restrict__ zr,
                                                               const float* __restrict__ h, int hs, long long pixels, int H, float* __restrict__ dpre,
                                                               float* __restrict__ dh) {
  const int q = H >> 2;
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long pix = idx / q;
    const int c = (int)(idx - pix * q) * 4;
    const f32x4 z = ld4(zr + pix * 2 * H + c), r = ld4(zr + pix * 2 * H + H + c);
    const f32x4 gz = dz ? ld4(dz + pix * dzs + c) : kZero4, gr = drh ? ld4(drh + pix * H + c) : kZero4;
    const f32x4 hv = h ? ld4(h + pix * hs + c) : kZero4;
    f32x4 pz, pr, o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pz[j] = gz[j] * z[j] * (1.f - z[j]);
      pr[j] = gr[j] * hv[j] * r[j] * (1.f - r[j]);
      o[j] = gr[j] * r[j];
    }
    st4(dpre + pix * 2 * H + c, pz); st4(dpre + pix * 2 * H + H + c, pr);
    if (dh) st4(dh + pix * H + c, o);
  }
}
// out: n = tanh(gx_o + gh_o), h' = h (1 - z) + n z
__global__ __launch_bounds__(256) void dvdgru_out_fwd_kernel(const float* __restrict__ gx, int sx, const float* __restrict__ gh, int sh,
                                                             const float* __restrict__ zr, const float* __restrict__ h, int hs, long long pixels, int H,
                                                             float* __restrict__ cand, float* __restrict__ hn) {
  const int q = H >> 2;
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long pix = idx / q;
    const int c = (int)(idx - pix * q) * 4;
    f32x4 a = ld4(gx + pix * sx + c);
    if (gh) a += ld4(gh + pix * sh + c);
    const f32x4 z = ld4(zr + pix * 2 * H + c);
    const f32x4 hv = h ? ld4(h + pix * hs + c) : kZero4;
    f32x4 n, o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { n[j] = sf_tanh(a[j]); o[j] = hv[j] * (1.f - z[j]) + n[j] * z[j]; }
    if (cand) st4(cand + pix * H + c, n);
    st4(hn + pix * H + c, o);
  }
}
// in: dh', saved cand, zr, h  ->  da (wrt the candidate's pre-activation), dz (wrt the gate z), dh (direct path dh' (1 - z))
__global__ __launch_bounds__(256) void dvdgru_out_bwd_kernel(const float* __restrict__ dhn, const float* __restrict__ cand, const float* __restrict__ zr,
                                                             const float* __restrict__ h, int hs, long long pixels, int H, float* __restrict__ da,
                                                             float* __restrict__ dz, int dzs, float* __restrict__ dh) {
  const int q = H >> 2;
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long pix = idx / q;
    const int c = (int)(idx - pix * q) * 4;
    const f32x4 g = ld4(dhn + pix * H + c), n = ld4(cand + pix * H + c), z = ld4(zr + pix * 2 * H + c);
    const f32x4 hv = h ? ld4(h + pix * hs + c) : kZero4;
    f32x4 a, gz, gh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = g[j] * z[j] * (1.f - n[j] * n[j]);
      gz[j] = g[j] * (n[j] - hv[j]);
      gh[j] = g[j] * (1.f - z[j]);
    }
    st4(da + pix * H + c, a); st4(dz + pix * dzs + c, gz);
    if (dzs >= 2 * H) st4(dz + pix * dzs + H + c, kZero4);   // dz as the gradient of the whole z | r tensor: the r half does not reach h' here
    if (dh) st4(dh + pix * H + c, gh);
  }
}

}  // namespace

extern "C" {

static int sn_slices(int height) { const int s = height / 16; return s < 1 ? 1 : (s > 32 ? 32 : s); }
size_t sf_spectral_norm_workspace_floats(int32_t height, int32_t width) { return (size_t)width + height + (size_t)sn_slices(height) * width; }

int sf_spectral_norm_fwd(const float* w_bar, int32_t height, int32_t width, float* u, float* v, int32_t power_iterations, float* w_out, float* sigma,
                         float* workspace, sfStream stream) {
  SF_REQUIRE(w_bar && u && v && w_out && sigma && workspace && height >= 1 && width >= 1 && power_iterations >= 1,
             "sf_spectral_norm_fwd: null pointer or empty matrix (%d x %d, %d iterations)", height, width, power_iterations);
  hipStream_t st = (hipStream_t)stream;
  float* vraw = workspace;
  float* uraw = workspace + width;
  float* part = uraw + height;
  const int slices = sn_slices(height);
  for (int it = 0; it < power_iterations; ++it) {
    if (slices == 1) hipLaunchKernelGGL(sn_wt_u_kernel, dim3((width + 255) / 256, 1), dim3(256), 0, st, w_bar, height, width, (const float*)u, vraw);
    else {
      hipLaunchKernelGGL(sn_wt_u_kernel, dim3((width + 255) / 256, slices), dim3(256), 0, st, w_bar, height, width, (const float*)u, part);
      hipLaunchKernelGGL(sn_sum_slices_kernel, dim3((width + 255) / 256), dim3(256), 0, st, (const float*)part, slices, width, vraw);
    }
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(height), dim3(256), 0, st, w_bar, height, width, (const float*)vraw, v, uraw);
    if (it + 1 < power_iterations) hipLaunchKernelGGL(sn_norm_u_kernel, dim3(1), dim3(256), 0, st, (const float*)uraw, height, u);
  }
  hipLaunchKernelGGL(sn_finish_kernel, dim3(grid_of((long long)height * width, 1024)), dim3(256), 0, st, w_bar, height, width, (const float*)uraw, u, sigma, w_out);
  SF_CHECK_LAUNCH("spectral_norm_fwd");
  return 0;
}

int sf_spectral_norm_bwd(const float* g, const float* w_bar, const float* u, const float* v, const float* sigma, int32_t height, int32_t width,
                         float* dw_bar, float* workspace, sfStream stream) {
  SF_REQUIRE(g && w_bar && u && v && sigma && dw_bar && workspace && height >= 1 && width >= 1, "sf_spectral_norm_bwd: null pointer or empty matrix");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)height * width;
  const int parts = grid_of(n, 512);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(parts), dim3(256), 0, st, g, w_bar, n, workspace);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid_of(n, 1024)), dim3(256), 0, st, g, u, v, sigma, (const float*)workspace, parts, height, width, dw_bar);
  SF_CHECK_LAUNCH("spectral_norm_bwd");
  return 0;
}

int sf_pool2(sfTensor x, int64_t n_out, int32_t oh, int32_t ow, int32_t tpool, int32_t nb, float scale, sfTensor addend, sfTensor y, sfStream stream) {
  SF_REQUIRE(tpool == 1 || tpool == 2, "sf_pool2: tpool %d (1 or 2)", tpool);
  SF_REQUIRE(oks(x, y.c) && oks(y, y.c) && (!addend.ptr || oks(addend, y.c)) && oh >= 1 && ow >= 1 && nb >= 1 && n_out % nb == 0,
             "sf_pool2: fp32 NHWC tensors with 16-byte pixels, n_out a multiple of nb");
  if (n_out <= 0) return 0;
  const int q = y.c / 4;
  hipLaunchKernelGGL(pool2_kernel, dim3(grid_of(n_out * oh * ow * q)), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, x.stride, (long long)n_out, oh, ow,
                     tpool, nb, scale, (const float*)addend.ptr, addend.stride, (float*)y.ptr, y.stride, q);
  SF_CHECK_LAUNCH("pool2");
  return 0;
}

int sf_expand2(sfTensor x, int64_t n_in, int32_t h, int32_t w, int32_t texp, int32_t nb, float scale, sfTensor y, sfStream stream) {
  SF_REQUIRE(texp == 1 || texp == 2, "sf_expand2: texp %d (1 or 2)", texp);
  SF_REQUIRE(oks(x, y.c) && oks(y, y.c) && h >= 1 && w >= 1 && nb >= 1 && n_in % nb == 0, "sf_expand2: fp32 NHWC tensors with 16-byte pixels, n_in a multiple of nb");
  if (n_in <= 0) return 0;
  const int q = y.c / 4;
  hipLaunchKernelGGL(expand2_kernel, dim3(grid_of(n_in * texp * 4 * h * w * q)), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, x.stride, (long long)n_in, h, w,
                     texp, nb, scale, (float*)y.ptr, y.stride, q);
  SF_CHECK_LAUNCH("expand2");
  return 0;
}

int sf_time_stack3_fwd(sfTensor x, int32_t T, int64_t pixels_per_frame, sfTensor y, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(y, 3 * x.c) && T >= 1, "sf_time_stack3_fwd: dense fp32 x [T][per][C], y [T][per][3C]");
  if (pixels_per_frame <= 0) return 0;
  hipLaunchKernelGGL(tstack3_fwd_kernel, dim3(grid_of((long long)T * pixels_per_frame * 3 * (x.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, T,
                     (long long)pixels_per_frame, x.c / 4, (float*)y.ptr);
  SF_CHECK_LAUNCH("time_stack3_fwd");
  return 0;
}

int sf_time_stack3_bwd(sfTensor gy, int32_t T, int64_t pixels_per_frame, sfTensor gx, sfStream stream) {
  SF_REQUIRE(gx.ptr && okd(gx, gx.c) && okd(gy, 3 * gx.c) && T >= 1, "sf_time_stack3_bwd: dense fp32 gy [T][per][3C], gx [T][per][C]");
  if (pixels_per_frame <= 0) return 0;
  hipLaunchKernelGGL(tstack3_bwd_kernel, dim3(grid_of((long long)T * pixels_per_frame * (gx.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)gy.ptr, T,
                     (long long)pixels_per_frame, gx.c / 4, (float*)gx.ptr);
  SF_CHECK_LAUNCH("time_stack3_bwd");
  return 0;
}

int sf_pad_shift_stack4_fwd(sfTensor x, int64_t n, int32_t h, int32_t w, sfTensor y, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(y, 4 * x.c) && h >= 1 && w >= 1, "sf_pad_shift_stack4_fwd: dense fp32 x [n][h][w][C], y [n][h+4][w+4][4C]");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pad_shift4_fwd_kernel, dim3(grid_of(n * (h + 4) * (w + 4) * x.c)), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (long long)n, h, w,
                     x.c / 4, (float*)y.ptr);
  SF_CHECK_LAUNCH("pad_shift_stack4_fwd");
  return 0;
}

int sf_pad_shift_stack4_bwd(sfTensor gy, int64_t n, int32_t h, int32_t w, sfTensor gx, sfStream stream) {
  SF_REQUIRE(gx.ptr && okd(gx, gx.c) && okd(gy, 4 * gx.c) && h >= 1 && w >= 1, "sf_pad_shift_stack4_bwd: dense fp32 gy [n][h+4][w+4][4C], gx [n][h][w][C]");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pad_shift4_bwd_kernel, dim3(grid_of(n * h * w * (gx.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)gy.ptr, (long long)n, h, w,
                     gx.c / 4, (float*)gx.ptr);
  SF_CHECK_LAUNCH("pad_shift_stack4_bwd");
  return 0;
}

int sf_regroup5x5_fwd(const float* w5, int64_t row_pitch, int32_t O, int32_t I, int32_t lanes, float* w3, sfStream stream) {
  SF_REQUIRE(w5 && w3 && O >= 1 && I >= 1 && lanes >= I && row_pitch >= (int64_t)I * 25, "sf_regroup5x5_fwd: null pointer, lanes (%d) < I (%d) or row pitch too small", lanes, I);
  hipLaunchKernelGGL(regroup5_fwd_kernel, dim3(grid_of((long long)O * 4 * lanes * 9)), dim3(256), 0, (hipStream_t)stream, w5, (long long)row_pitch, O, I, lanes, w3);
  SF_CHECK_LAUNCH("regroup5x5_fwd");
  return 0;
}
int sf_regroup5x5_bwd(const float* g3, int32_t O, int32_t I, int32_t lanes, float* g5, sfStream stream) {
  SF_REQUIRE(g3 && g5 && O >= 1 && I >= 1 && lanes >= I, "sf_regroup5x5_bwd: null pointer or lanes (%d) < I (%d)", lanes, I);
  hipLaunchKernelGGL(regroup5_bwd_kernel, dim3(grid_of((long long)O * I * 25)), dim3(256), 0, (hipStream_t)stream, g3, O, I, lanes, g5);
  SF_CHECK_LAUNCH("regroup5x5_bwd");
  return 0;
}

int sf_space_to_depth2(sfTensor x, int64_t n, int32_t h, int32_t w, int32_t inverse, sfTensor y, sfStream stream) {
  const sfTensor& full = inverse ? y : x;
  const sfTensor& half = inverse ? x : y;
  SF_REQUIRE(x.ptr && y.ptr && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 && okd(full, full.c) && okd(half, 4 * full.c),
             "sf_space_to_depth2: dense fp32 [n][h][w][C] <-> [n][h/2][w/2][4C], h and w even (h %d w %d)", h, w);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(s2d2_kernel, dim3(grid_of(n * h * w * (full.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (long long)n, h / 2, w / 2,
                     full.c / 4, (float*)y.ptr, inverse);
  SF_CHECK_LAUNCH("space_to_depth2");
  return 0;
}

int sf_regroup5x5_s2d_fwd(const float* w5, int64_t row_pitch, int32_t R, int32_t hp, int32_t I, int32_t lanes, float* w3, sfStream stream) {
  SF_REQUIRE(w5 && w3 && R >= 1 && hp >= 1 && R % hp == 0 && I >= 1 && lanes >= I && row_pitch >= (int64_t)I * 25,
             "sf_regroup5x5_s2d_fwd: null pointer, rows (%d) not a multiple of the gate block (%d), lanes (%d) < I (%d) or row pitch too small", R, hp, lanes, I);
  hipLaunchKernelGGL(regroup5_s2d_fwd_kernel, dim3(grid_of((long long)R * 16 * lanes * 9)), dim3(256), 0, (hipStream_t)stream, w5, (long long)row_pitch, R, hp, I, lanes, w3);
  SF_CHECK_LAUNCH("regroup5x5_s2d_fwd");
  return 0;
}
int sf_regroup5x5_s2d_bwd(const float* g3, int32_t R, int32_t hp, int32_t I, int32_t lanes, float* g5, sfStream stream) {
  SF_REQUIRE(g3 && g5 && R >= 1 && hp >= 1 && R % hp == 0 && I >= 1 && lanes >= I, "sf_regroup5x5_s2d_bwd: null pointer, rows (%d) not a multiple of the gate block (%d) or lanes (%d) < I (%d)", R, hp, lanes, I);
  hipLaunchKernelGGL(regroup5_s2d_bwd_kernel, dim3(grid_of((long long)R * I * 25)), dim3(256), 0, (hipStream_t)stream, g3, R, hp, I, lanes, g5);
  SF_CHECK_LAUNCH("regroup5x5_s2d_bwd");
  return 0;
}

int sf_pad_s2d_fwd(sfTensor x, int64_t n, int32_t h, int32_t w, sfTensor y, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(y, 4 * x.c) && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0,
             "sf_pad_s2d_fwd: dense fp32 x [n][h][w][C] with even h, w; y [n][h/2+1][w/2+1][4C]");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pad_s2d_fwd_kernel, dim3(grid_of(n * (h / 2 + 1) * (w / 2 + 1) * x.c)), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (long long)n, h, w,
                     x.c / 4, (float*)y.ptr);
  SF_CHECK_LAUNCH("pad_s2d_fwd");
  return 0;
}

int sf_pad_s2d_bwd(sfTensor gy, int64_t n, int32_t h, int32_t w, sfTensor gx, sfStream stream) {
  SF_REQUIRE(gx.ptr && okd(gx, gx.c) && okd(gy, 4 * gx.c) && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0,
             "sf_pad_s2d_bwd: dense fp32 gy [n][h/2+1][w/2+1][4C], gx [n][h][w][C] with even h, w");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pad_s2d_bwd_kernel, dim3(grid_of(n * h * w * (gx.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)gy.ptr, (long long)n, h, w,
                     gx.c / 4, (float*)gx.ptr);
  SF_CHECK_LAUNCH("pad_s2d_bwd");
  return 0;
}

int sf_border(sfTensor x, int64_t n, int32_t h, int32_t w, int32_t border, int32_t pad, sfTensor y, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(y, x.c) && h >= 1 && w >= 1 && border >= 1, "sf_border: dense fp32 tensors of equal channel count, border >= 1");
  if (n <= 0) return 0;
  const long long work = pad ? n * (h + 2 * border) * (w + 2 * border) * (x.c / 4) : n * h * w * (x.c / 4);
  hipLaunchKernelGGL(border_kernel, dim3(grid_of(work)), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (long long)n, h, w, border, x.c / 4, pad, (float*)y.ptr);
  SF_CHECK_LAUNCH("border");
  return 0;
}

static int pool3_geom(int32_t d0, int32_t d1, int32_t d2, int32_t k0, int32_t k1, int32_t k2, int32_t s0, int32_t s1, int32_t s2, Pool3* g) {
  if (d0 < 1 || d1 < 1 || d2 < 1 || k0 < 1 || k1 < 1 || k2 < 1 || s0 < k0 || s1 < k1 || s2 < k2 || k0 > d0 || k1 > d1 || k2 > d2) return 1;
  *g = Pool3{d0, d1, d2, (d0 - k0) / s0 + 1, (d1 - k1) / s1 + 1, (d2 - k2) / s2 + 1, k0, k1, k2, s0, s1, s2};
  return 0;
}
int sf_maxpool3d_fwd(sfTensor x, int64_t batch, int32_t d0, int32_t d1, int32_t d2, int32_t k0, int32_t k1, int32_t k2, int32_t s0, int32_t s1, int32_t s2,
                     sfTensor y, sfStream stream) {
  Pool3 g;
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(y, x.c) && pool3_geom(d0, d1, d2, k0, k1, k2, s0, s1, s2, &g) == 0,
             "sf_maxpool3d_fwd: dense fp32 tokens, window <= extent, stride >= window");
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(grid_of(batch * g.o0 * g.o1 * g.o2 * (x.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (long long)batch, g,
                     x.c / 4, (float*)y.ptr);
  SF_CHECK_LAUNCH("maxpool3d_fwd");
  return 0;
}
int sf_maxpool3d_bwd(sfTensor x, sfTensor gy, int64_t batch, int32_t d0, int32_t d1, int32_t d2, int32_t k0, int32_t k1, int32_t k2, int32_t s0, int32_t s1,
                     int32_t s2, sfTensor gx, sfStream stream) {
  Pool3 g;
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(gy, x.c) && okd(gx, x.c) && pool3_geom(d0, d1, d2, k0, k1, k2, s0, s1, s2, &g) == 0,
             "sf_maxpool3d_bwd: dense fp32 tokens, window <= extent, stride >= window");
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(grid_of(batch * d0 * d1 * d2 * (x.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (const float*)gy.ptr,
                     (long long)batch, g, x.c / 4, (float*)gx.ptr);
  SF_CHECK_LAUNCH("maxpool3d_bwd");
  return 0;
}

int sf_film_act_fwd(sfTensor x, int64_t n, int32_t h, int32_t w, const float* mean, const float* rstd, const float* embed, int32_t creal, int32_t relu,
                    int32_t up, sfTensor y, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(y, x.c) && mean && rstd && embed && creal >= 1 && creal <= x.c && h >= 1 && w >= 1,
             "sf_film_act_fwd: dense fp32 x / y, statistics and embedding [n][2*creal]");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(film_fwd_kernel, dim3(grid_of(n * h * w * (x.c / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (long long)n, h, w, x.c / 4, mean,
                     rstd, embed, creal, relu, up, (float*)y.ptr);
  SF_CHECK_LAUNCH("film_act_fwd");
  return 0;
}

static int film_slices(long long n, int h, int w, int q) {
  long long blocks = n * ((q + 15) / 16);
  long long s = (1024 + blocks - 1) / blocks;
  const long long maxs = ((long long)h * w + 255) / 256;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}
size_t sf_film_act_bwd_workspace_floats(int64_t n, int32_t h, int32_t w, int32_t c, int32_t creal) {
  return (size_t)film_slices(n, h, w, c / 4) * n * 2 * creal;
}
int sf_film_act_bwd(sfTensor gy, sfTensor x, int64_t n, int32_t h, int32_t w, const float* mean, const float* rstd, const float* embed, int32_t creal,
                    int32_t relu, int32_t up, sfTensor dxhat, float* dembed, float* workspace, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(gy, x.c) && okd(dxhat, x.c) && mean && rstd && embed && dembed && workspace && creal >= 1 && creal <= x.c,
             "sf_film_act_bwd: dense fp32 tensors, statistics, embedding and workspace");
  if (n <= 0) return 0;
  const int q = x.c / 4, slices = film_slices(n, h, w, q);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(film_bwd_kernel, dim3(slices, (unsigned)n, (q + 15) / 16), dim3(256), 0, st, (const float*)gy.ptr, (const float*)x.ptr, h, w, q, mean, rstd, embed,
                     creal, relu, up, (float*)dxhat.ptr, workspace);
  const long long ne = n * 2 * creal;
  hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, (const float*)workspace, slices, ne, dembed);
  SF_CHECK_LAUNCH("film_act_bwd");
  return 0;
}

int sf_relu_sum_pixels_fwd(sfTensor x, int64_t n, int64_t pixels, float* out, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && out, "sf_relu_sum_pixels_fwd: dense fp32 x, out [n][C]");
  if (n <= 0) return 0;
  const int q = x.c / 4;
  hipLaunchKernelGGL(relu_sum_fwd_kernel, dim3((unsigned)n, (q + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)x.ptr, (long long)pixels, q, out);
  SF_CHECK_LAUNCH("relu_sum_pixels_fwd");
  return 0;
}

int sf_relu_sum_pixels_bwd(const float* g, sfTensor x, int64_t n, int64_t pixels, sfTensor gx, sfStream stream) {
  SF_REQUIRE(x.ptr && okd(x, x.c) && okd(gx, x.c) && g, "sf_relu_sum_pixels_bwd: dense fp32 tensors");
  if (n <= 0 || pixels <= 0) return 0;
  hipLaunchKernelGGL(relu_sum_bwd_kernel, dim3(grid_of(n * pixels * (x.c / 4))), dim3(256), 0, (hipStream_t)stream, g, (const float*)x.ptr, (long long)n,
                     (long long)pixels, x.c / 4, (float*)gx.ptr);
  SF_CHECK_LAUNCH("relu_sum_pixels_bwd");
  return 0;
}

int sf_axpy(const float* o, const float* x, const float* gamma_dev, float alpha, int64_t n, float* y, sfStream stream) {
  SF_REQUIRE(o && y && n % 4 == 0 && (((uintptr_t)o | (uintptr_t)y | (uintptr_t)x) & 15) == 0, "sf_axpy: 16-byte aligned fp32 arrays, n %% 4 == 0");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_of(n / 4)), dim3(256), 0, (hipStream_t)stream, o, x, gamma_dev, alpha, (long long)(n / 4), y);
  SF_CHECK_LAUNCH("axpy");
  return 0;
}

int sf_dot(const float* a, const float* b, int64_t n, float* out, float* workspace, sfStream stream) {
  SF_REQUIRE(a && b && out && workspace && n >= 0, "sf_dot: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int parts = grid_of(n, 512);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(parts), dim3(256), 0, st, a, b, (long long)n, workspace);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, parts, out);
  SF_CHECK_LAUNCH("dot");
  return 0;
}

int sf_tanh(const float* x, const float* y_for_backward, int64_t n, float* out, sfStream stream) {
  SF_REQUIRE(x && out && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)y_for_backward) & 15) == 0, "sf_tanh: 16-byte aligned fp32 arrays, n %% 4 == 0");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(tanh_kernel, dim3(grid_of(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y_for_backward, (long long)(n / 4), out);
  SF_CHECK_LAUNCH("tanh");
  return 0;
}

int sf_dvdgru_gates_fwd(sfTensor gx, sfTensor gh, sfTensor h, int64_t pixels, int32_t hidp, sfTensor zr, sfTensor rh, sfStream stream) {
  SF_REQUIRE(hidp > 0 && hidp % 4 == 0 && oks(gx, 2 * hidp) && (!gh.ptr || oks(gh, 2 * hidp)) && (!h.ptr || oks(h, hidp)) && okd(zr, 2 * hidp) && okd(rh, hidp),
             "sf_dvdgru_gates_fwd: gx / gh [.., 2*hidp], h [.., hidp], dense zr / rh (fp32, 16-byte pixels)");
  if (pixels <= 0) return 0;
  hipLaunchKernelGGL(dvdgru_gates_fwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)gx.ptr, gx.stride,
                     (const float*)gh.ptr, gh.stride, (const float*)h.ptr, h.stride, (long long)pixels, hidp, (float*)zr.ptr, (float*)rh.ptr);
  SF_CHECK_LAUNCH("dvdgru_gates_fwd");
  return 0;
}

int sf_dvdgru_gates_bwd(sfTensor dz, sfTensor drh, sfTensor zr, sfTensor h, int64_t pixels, int32_t hidp, sfTensor dpre, sfTensor dh, sfStream stream) {
  SF_REQUIRE(hidp > 0 && hidp % 4 == 0 && (!dz.ptr || oks(dz, hidp)) && (!drh.ptr || okd(drh, hidp)) && okd(zr, 2 * hidp) && (!h.ptr || oks(h, hidp)) &&
                 okd(dpre, 2 * hidp) && (!dh.ptr || okd(dh, hidp)), "sf_dvdgru_gates_bwd: fp32 tensors, dense except dz / h (row-strided)");
  if (pixels <= 0) return 0;
  hipLaunchKernelGGL(dvdgru_gates_bwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)dz.ptr, dz.stride, (const float*)drh.ptr,
                     (const float*)zr.ptr, (const float*)h.ptr, h.stride, (long long)pixels, hidp, (float*)dpre.ptr, (float*)dh.ptr);
  SF_CHECK_LAUNCH("dvdgru_gates_bwd");
  return 0;
}

int sf_dvdgru_out_fwd(sfTensor gx, sfTensor gh, sfTensor zr, sfTensor h, int64_t pixels, int32_t hidp, sfTensor cand, sfTensor h_new, sfStream stream) {
  SF_REQUIRE(hidp > 0 && hidp % 4 == 0 && oks(gx, hidp) && (!gh.ptr || oks(gh, hidp)) && okd(zr, 2 * hidp) && (!h.ptr || oks(h, hidp)) &&
                 (!cand.ptr || okd(cand, hidp)) && okd(h_new, hidp), "sf_dvdgru_out_fwd: gx / gh [.., hidp], dense zr, cand, h_new (fp32)");
  if (pixels <= 0) return 0;
  hipLaunchKernelGGL(dvdgru_out_fwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)gx.ptr, gx.stride,
                     (const float*)gh.ptr, gh.stride, (const float*)zr.ptr, (const float*)h.ptr, h.stride, (long long)pixels, hidp, (float*)cand.ptr, (float*)h_new.ptr);
  SF_CHECK_LAUNCH("dvdgru_out_fwd");
  return 0;
}

int sf_dvdgru_out_bwd(sfTensor dh_new, sfTensor cand, sfTensor zr, sfTensor h, int64_t pixels, int32_t hidp, sfTensor da, sfTensor dz, sfTensor dh,
                      sfStream stream) {
  SF_REQUIRE(hidp > 0 && hidp % 4 == 0 && okd(dh_new, hidp) && okd(cand, hidp) && okd(zr, 2 * hidp) && (!h.ptr || oks(h, hidp)) && okd(da, hidp) && (okd(dz, hidp) || okd(dz, 2 * hidp)) &&
                 (!dh.ptr || okd(dh, hidp)), "sf_dvdgru_out_bwd: dense fp32 tensors (dz: hidp lanes, or 2*hidp = the gradient of z | r with a zero r half)");
  if (pixels <= 0) return 0;
  hipLaunchKernelGGL(dvdgru_out_bwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)dh_new.ptr, (const float*)cand.ptr,
                     (const float*)zr.ptr, (const float*)h.ptr, h.stride, (long long)pixels, hidp, (float*)da.ptr, (float*)dz.ptr, dz.stride, (float*)dh.ptr);
  SF_CHECK_LAUNCH("dvdgru_out_bwd");
  return 0;
}

}  // extern "C"
