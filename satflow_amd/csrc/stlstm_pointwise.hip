// Pointwise halves of the ST-LSTM cell with memory decoupling (PredRNN v2), SURVEY 8f-4:
// reference satflow/models/layers/SpatioTemporalLSTMCell_memory_decoupling.py:110-138 (layer_norm=False).  The four 3x3
// convolutions and the 1x1 convolution of the cell run on the MFMA kernels (sf_conv3x3_fwd, sf_linear_fwd); what is left are two
// HBM-bound streaming stages, forward and backward:
//   gates (:114-132):  from gx = conv_x(x) [i f g i' f' g' o], gh = conv_h(h) [i f g o], gm = conv_m(m) [i' f' g'], c, m:
//       i = sig(i_x + i_h), f = sig(f_x + f_h + 1), g = tanh(g_x + g_h), delta_c = i g, c' = f c + delta_c,
//       i' = sig(i_x' + i_m), f' = sig(f_x' + f_m + 1), g' = tanh(g_x' + g_m), delta_m = i' g', m' = f' m + delta_m,
//       pre_o = o_x + o_h;  c', m' are written twice: as tensors of their own (second source pair of conv_o) and side by side as
//       mem = [c' | m'] (the 1x1 convolution's input)
//   out (:135-136):  o = sig(pre_o + conv_o(mem)), h' = o tanh(conv_last(mem)).
// All tensors NHWC fp32, gate-major channel blocks of hidp (padded hidden channels); 16-byte accesses.
#include "sf_common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

struct StGatesParams {
  const float* gx; const float* gh; const float* gm; int s_gx, s_gh, s_gm;
  const float* c; const float* m; int s_c, s_m;
  float* c_new; float* m_new; float* mem; float* delta_c; float* delta_m; float* pre_o; float* gates;  // gates [.., 6*hidp]: i f g i' f' g'
  long long pixels; int hidp; float forget_bias;
};

__global__ __launch_bounds__(256) void stlstm_gates_fwd_kernel(const StGatesParams p) {
  const int q = p.hidp >> 2, H = p.hidp;
  const long long total = p.pixels * q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long pix = idx / q;
    const int ch = (int)(idx - pix * q) * 4;
    const float* gx = p.gx + pix * p.s_gx + ch;
    const float* gh = p.gh + pix * p.s_gh + ch;
    const float* gm = p.gm + pix * p.s_gm + ch;
    const f32x4 ix = ld4(gx), fx = ld4(gx + H), gxg = ld4(gx + 2 * H), ixp = ld4(gx + 3 * H), fxp = ld4(gx + 4 * H), gxp = ld4(gx + 5 * H), ox = ld4(gx + 6 * H);
    const f32x4 ih = ld4(gh), fh = ld4(gh + H), ghg = ld4(gh + 2 * H), oh = ld4(gh + 3 * H);
    const f32x4 im = ld4(gm), fm = ld4(gm + H), gmg = ld4(gm + 2 * H);
    const f32x4 c = ld4(p.c + pix * p.s_c + ch), m = ld4(p.m + pix * p.s_m + ch);
    f32x4 i, f, g, ip, fp, gp, dc, dm, cn, mn, po;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      i[j] = sf_sigmoid(ix[j] + ih[j]); f[j] = sf_sigmoid(fx[j] + fh[j] + p.forget_bias); g[j] = sf_tanh(gxg[j] + ghg[j]);
      dc[j] = i[j] * g[j]; cn[j] = f[j] * c[j] + dc[j];
      ip[j] = sf_sigmoid(ixp[j] + im[j]); fp[j] = sf_sigmoid(fxp[j] + fm[j] + p.forget_bias); gp[j] = sf_tanh(gxp[j] + gmg[j]);
      dm[j] = ip[j] * gp[j]; mn[j] = fp[j] * m[j] + dm[j];
      po[j] = ox[j] + oh[j];
    }
    st4(p.c_new + pix * H + ch, cn); st4(p.m_new + pix * H + ch, mn);
    st4(p.mem + pix * 2 * H + ch, cn); st4(p.mem + pix * 2 * H + H + ch, mn);
    st4(p.delta_c + pix * H + ch, dc); st4(p.delta_m + pix * H + ch, dm); st4(p.pre_o + pix * H + ch, po);
    if (p.gates) {
      float* gs = p.gates + pix * 6 * H + ch;
      st4(gs, i); st4(gs + H, f); st4(gs + 2 * H, g); st4(gs + 3 * H, ip); st4(gs + 4 * H, fp); st4(gs + 5 * H, gp);
    }
  }
}

struct StGatesBwdParams {
  const float* d_cnew; const float* d_mnew; const float* d_mem; const float* d_dc; const float* d_dm; const float* d_po;  // each nullable
  const float* gates; const float* c; const float* m; int s_c, s_m;
  float* dgx; float* dgh; float* dgm; float* dc; float* dm;
  long long pixels; int hidp;
};

__global__ __launch_bounds__(256) void stlstm_gates_bwd_kernel(const StGatesBwdParams p) {
  const int q = p.hidp >> 2, H = p.hidp;
  const long long total = p.pixels * q;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long pix = idx / q;
    const int ch = (int)(idx - pix * q) * 4;
    // gradients wrt c', m' arrive through their own tensors and through mem
    f32x4 Dc = p.d_cnew ? ld4(p.d_cnew + pix * H + ch) : z4, Dm = p.d_mnew ? ld4(p.d_mnew + pix * H + ch) : z4;
    if (p.d_mem) { Dc += ld4(p.d_mem + pix * 2 * H + ch); Dm += ld4(p.d_mem + pix * 2 * H + H + ch); }
    const f32x4 Ddc = p.d_dc ? ld4(p.d_dc + pix * H + ch) : z4, Ddm = p.d_dm ? ld4(p.d_dm + pix * H + ch) : z4;
    const f32x4 Dpo = p.d_po ? ld4(p.d_po + pix * H + ch) : z4;
    const float* gs = p.gates + pix * 6 * H + ch;
    const f32x4 i = ld4(gs), f = ld4(gs + H), g = ld4(gs + 2 * H), ip = ld4(gs + 3 * H), fp = ld4(gs + 4 * H), gp = ld4(gs + 5 * H);
    const f32x4 c = ld4(p.c + pix * p.s_c + ch), m = ld4(p.m + pix * p.s_m + ch);
    f32x4 ai, af, ag, aip, afp, agp, dc, dm;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dig = Dc[j] + Ddc[j];  // d(i * g): through c' and through delta_c
      ai[j] = dig * g[j] * i[j] * (1.f - i[j]);
      ag[j] = dig * i[j] * (1.f - g[j] * g[j]);
      af[j] = Dc[j] * c[j] * f[j] * (1.f - f[j]);
      dc[j] = Dc[j] * f[j];
      const float digp = Dm[j] + Ddm[j];
      aip[j] = digp * gp[j] * ip[j] * (1.f - ip[j]);
      agp[j] = digp * ip[j] * (1.f - gp[j] * gp[j]);
      afp[j] = Dm[j] * m[j] * fp[j] * (1.f - fp[j]);
      dm[j] = Dm[j] * fp[j];
    }
    float* a = p.dgx + pix * 7 * H + ch;
    st4(a, ai); st4(a + H, af); st4(a + 2 * H, ag); st4(a + 3 * H, aip); st4(a + 4 * H, afp); st4(a + 5 * H, agp); st4(a + 6 * H, Dpo);
    float* b = p.dgh + pix * 4 * H + ch;
    st4(b, ai); st4(b + H, af); st4(b + 2 * H, ag); st4(b + 3 * H, Dpo);
    float* d = p.dgm + pix * 3 * H + ch;
    st4(d, aip); st4(d + H, afp); st4(d + 2 * H, agp);
    st4(p.dc + pix * H + ch, dc); st4(p.dm + pix * H + ch, dm);
  }
}

// out stage: o = sig(pre_o + co), t = tanh(last), h' = o t; saved [o | t] for the backward pass
__global__ __launch_bounds__(256) void stlstm_out_fwd_kernel(const float* __restrict__ pre_o, const float* __restrict__ co, int s_co,
                                                            const float* __restrict__ last, int s_last, long long pixels, int hidp,
                                                            float* __restrict__ h_new, float* __restrict__ saved) {
  const int q = hidp >> 2;
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long pix = idx / q;
    const int ch = (int)(idx - pix * q) * 4;
    const f32x4 a = ld4(pre_o + pix * hidp + ch) + ld4(co + pix * s_co + ch), l = ld4(last + pix * s_last + ch);
    f32x4 o, t, h;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = sf_sigmoid(a[j]); t[j] = sf_tanh(l[j]); h[j] = o[j] * t[j]; }
    st4(h_new + pix * hidp + ch, h);
    if (saved) { st4(saved + pix * 2 * hidp + ch, o); st4(saved + pix * 2 * hidp + hidp + ch, t); }
  }
}

// d_a = dh t o (1 - o) (gradient wrt pre_o and wrt conv_o's output alike), d_last = dh o (1 - t^2)
__global__ __launch_bounds__(256) void stlstm_out_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ saved, long long pixels, int hidp,
                                                            float* __restrict__ d_a, float* __restrict__ d_last) {
  const int q = hidp >> 2;
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long pix = idx / q;
    const int ch = (int)(idx - pix * q) * 4;
    const f32x4 g = ld4(dh + pix * hidp + ch), o = ld4(saved + pix * 2 * hidp + ch), t = ld4(saved + pix * 2 * hidp + hidp + ch);
    f32x4 da, dl;
#pragma unroll
    for (int j = 0; j < 4; ++j) { da[j] = g[j] * t[j] * o[j] * (1.f - o[j]); dl[j] = g[j] * o[j] * (1.f - t[j] * t[j]); }
    st4(d_a + pix * hidp + ch, da); st4(d_last + pix * hidp + ch, dl);
  }
}


// ---- nn.LayerNorm([C', W, W]) behind the cell's convolutions (layer_norm=True, reference :20-62) ---------------------------
// Activations: NHWC with GATE-MAJOR PADDED lanes: lane l = g * hidp + j is channel g * hid + j of the reference (j < hid), pad lanes
// are zero and take no part.  Statistics per sample over all real channels and pixels; the affine parameters keep the reference's
// [C'][pixels] (CHW) layout.  Per sample the sums are taken by LN_SLICES workgroups in double precision; one thread per sample then adds the slices in slice order and leaves the
// two finished per-sample values in slice 0 (ln_finalize_kernel) - every consumer reads those two values, not the 32 slices.
#define LN_SLICES 32
struct LnGeom { int gates, hid, hidp; long long pixels; };
__device__ __forceinline__ int ln_channel(const LnGeom& g, int lane) {  // real channel of a lane or -1
  const int gt = lane / g.hidp, j = lane - gt * g.hidp;
  return j < g.hid ? gt * g.hid + j : -1;
}
// partial[n][slice][2] = (sum a, sum b): mode 0: a = x, b = x^2; mode 1 (backward): a = dy*gamma, b = dy*gamma*xhat
__global__ __launch_bounds__(256) void ln_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                      const double* __restrict__ fwd_partial, float eps, LnGeom g, double* __restrict__ partial) {
  __shared__ double red[2][4];
  const int lanes = g.gates * g.hidp;
  const long long n = blockIdx.y, per = (g.pixels + LN_SLICES - 1) / LN_SLICES, p0 = (long long)blockIdx.x * per, p1 = p0 + per < g.pixels ? p0 + per : g.pixels;
  float mean = 0.f, rstd = 0.f;
  if (dy) { mean = (float)fwd_partial[n * LN_SLICES * 2]; rstd = (float)fwd_partial[n * LN_SLICES * 2 + 1]; }  // finished by ln_finalize_kernel
  double a = 0, b = 0;
  const long long work = (p1 > p0 ? p1 - p0 : 0) * lanes;
  for (long long e = threadIdx.x; e < work; e += 256) {
    const long long p = p0 + e / lanes;
    const int lane = (int)(e % lanes), c = ln_channel(g, lane);
    if (c < 0) continue;
    const float v = x[(n * g.pixels + p) * lanes + lane];
    if (!dy) { a += v; b += (double)v * v; }
    else {
      const float d = dy[(n * g.pixels + p) * lanes + lane] * gamma[(long long)c * g.pixels + p];
      a += d; b += (double)d * ((v - mean) * rstd);
    }
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[(n * LN_SLICES + blockIdx.x) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[(n * LN_SLICES + blockIdx.x) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
// One thread per sample: slices added in slice order; slice 0 then holds (mean, rstd) [mode 0] or the two backward means [mode 1], as floats widened to
// double.  The slices 1.. keep their sums (nobody reads them afterwards).
__global__ __launch_bounds__(64) void ln_finalize_kernel(double* __restrict__ partial, long long n_samples, double cnt, float eps, int mode) {
  const long long n = (long long)blockIdx.x * 64 + threadIdx.x;
  if (n >= n_samples) return;
  double s0 = 0, s1 = 0;
  for (int k = 0; k < LN_SLICES; ++k) { s0 += partial[(n * LN_SLICES + k) * 2]; s1 += partial[(n * LN_SLICES + k) * 2 + 1]; }
  float a, b;
  if (mode == 0) {
    const double mu = s0 / cnt, var = s1 / cnt - mu * mu;
    a = (float)mu; b = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps));
  } else { a = (float)(s0 / cnt); b = (float)(s1 / cnt); }
  partial[n * LN_SLICES * 2] = (double)a; partial[n * LN_SLICES * 2 + 1] = (double)b;
}
__device__ __forceinline__ void ln_stats(const double* partial, long long n, float& a, float& b) {
  a = (float)partial[n * LN_SLICES * 2]; b = (float)partial[n * LN_SLICES * 2 + 1];
}
// y = (x - mean_n) * rstd_n * gamma[c][p] + beta[c][p]  (pad lanes: 0)
__global__ __launch_bounds__(256) void ln_apply_kernel(const float* __restrict__ x, const double* __restrict__ partial, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, LnGeom g, long long n_samples, float* __restrict__ y) {
  const int lanes = g.gates * g.hidp;
  const long long total = n_samples * g.pixels * lanes;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int lane = (int)(e % lanes), c = ln_channel(g, lane);
    const long long pp = e / lanes, p = pp % g.pixels, n = pp / g.pixels;
    float o = 0.f;
    if (c >= 0) {
      float mean, rstd;
      ln_stats(partial, n, mean, rstd);
      o = (x[e] - mean) * rstd * gamma[(long long)c * g.pixels + p] + beta[(long long)c * g.pixels + p];
    }
    y[e] = o;
  }
}
// dx = rstd_n * (dxhat - mean_n(dxhat) - xhat * mean_n(dxhat * xhat)), dxhat = dy * gamma
__global__ __launch_bounds__(256) void ln_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const double* __restrict__ fpart,
                                                           const double* __restrict__ bpart, const float* __restrict__ gamma, float eps, LnGeom g,
                                                           long long n_samples, float* __restrict__ dx) {
  const int lanes = g.gates * g.hidp;
  const long long total = n_samples * g.pixels * lanes;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int lane = (int)(e % lanes), c = ln_channel(g, lane);
    const long long pp = e / lanes, p = pp % g.pixels, n = pp / g.pixels;
    float o = 0.f;
    if (c >= 0) {
      float mean, rstd, m0, m1;
      ln_stats(fpart, n, mean, rstd);
      ln_stats(bpart, n, m0, m1);
      const float xh = (x[e] - mean) * rstd;
      o = rstd * (dy[e] * gamma[(long long)c * g.pixels + p] - m0 - xh * m1);
    }
    dx[e] = o;
  }
}
// dgamma[c][p] = sum_n dy * xhat, dbeta[c][p] = sum_n dy: one thread per (pixel, lane), coalesced over lanes
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ dy, const double* __restrict__ fpart, float eps,
                                                            LnGeom g, long long n_samples, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int lanes = g.gates * g.hidp;
  const long long total = g.pixels * lanes;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int lane = (int)(e % lanes), c = ln_channel(g, lane);
    if (c < 0) continue;
    const long long p = e / lanes;
    float sg = 0.f, sb = 0.f;
    for (long long n = 0; n < n_samples; ++n) {
      float mean, rstd;
      ln_stats(fpart, n, mean, rstd);
      const float d = dy[n * total + e];
      sg = __builtin_fmaf(d, (x[n * total + e] - mean) * rstd, sg); sb += d;
    }
    dgamma[(long long)c * g.pixels + p] = sg; dbeta[(long long)c * g.pixels + p] = sb;
  }
}

bool okf(const sfTensor& t, int c) {  // fp32, 16-byte aligned pixels, at least c lanes
  return t.ptr && t.dtype == SF_F32 && (((uintptr_t)t.ptr) & 15) == 0 && t.stride % 4 == 0 && t.c >= c;
}
bool okf0(const sfTensor& t, int c) { return !t.ptr || okf(t, c); }
int grid_of(long long total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

}  // namespace

extern "C" {

int sf_stlstm_gates_fwd(sfTensor gx, sfTensor gh, sfTensor gm, sfTensor c, sfTensor m, int64_t pixels, int32_t hidp, float forget_bias,
                        sfTensor c_new, sfTensor m_new, sfTensor mem, sfTensor delta_c, sfTensor delta_m, sfTensor pre_o, sfTensor gates,
                        int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_stlstm_gates_fwd: dtype %d not built", dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && hidp > 0, "stlstm gates: hidp=%d", hidp);
  SF_REQUIRE(okf(gx, 7 * hidp) && okf(gh, 4 * hidp) && okf(gm, 3 * hidp) && okf(c, hidp) && okf(m, hidp), "stlstm gates: gx [..,7*hidp], gh [..,4*hidp], gm [..,3*hidp], c, m [..,hidp] fp32, 16-byte aligned");
  SF_REQUIRE(okf(c_new, hidp) && okf(m_new, hidp) && okf(mem, 2 * hidp) && okf(delta_c, hidp) && okf(delta_m, hidp) && okf(pre_o, hidp) && okf0(gates, 6 * hidp) &&
                 c_new.stride == hidp && m_new.stride == hidp && mem.stride == 2 * hidp && delta_c.stride == hidp && delta_m.stride == hidp &&
                 pre_o.stride == hidp && (!gates.ptr || gates.stride == 6 * hidp),
             "stlstm gates: outputs must be dense fp32 tensors of hidp (mem: 2*hidp, gates: 6*hidp) lanes");
  if (pixels <= 0) return 0;
  StGatesParams p{};
  p.gx = (const float*)gx.ptr; p.gh = (const float*)gh.ptr; p.gm = (const float*)gm.ptr; p.s_gx = gx.stride; p.s_gh = gh.stride; p.s_gm = gm.stride;
  p.c = (const float*)c.ptr; p.m = (const float*)m.ptr; p.s_c = c.stride; p.s_m = m.stride;
  p.c_new = (float*)c_new.ptr; p.m_new = (float*)m_new.ptr; p.mem = (float*)mem.ptr; p.delta_c = (float*)delta_c.ptr; p.delta_m = (float*)delta_m.ptr;
  p.pre_o = (float*)pre_o.ptr; p.gates = (float*)gates.ptr;
  p.pixels = pixels; p.hidp = hidp; p.forget_bias = forget_bias;
  hipLaunchKernelGGL(stlstm_gates_fwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("stlstm_gates_fwd");
  return 0;
}

int sf_stlstm_gates_bwd(sfTensor d_c_new, sfTensor d_m_new, sfTensor d_mem, sfTensor d_delta_c, sfTensor d_delta_m, sfTensor d_pre_o, sfTensor gates,
                        sfTensor c, sfTensor m, int64_t pixels, int32_t hidp, sfTensor dgx, sfTensor dgh, sfTensor dgm, sfTensor dc, sfTensor dm,
                        int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_stlstm_gates_bwd: dtype %d not built", dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && hidp > 0, "stlstm gates bwd: hidp=%d", hidp);
  auto dense = [&](const sfTensor& t, int lanes) { return !t.ptr || (okf(t, lanes) && t.stride == lanes); };
  SF_REQUIRE(dense(d_c_new, hidp) && dense(d_m_new, hidp) && dense(d_mem, 2 * hidp) && dense(d_delta_c, hidp) && dense(d_delta_m, hidp) && dense(d_pre_o, hidp),
             "stlstm gates bwd: incoming gradients must be dense fp32 tensors (or null)");
  SF_REQUIRE(okf(gates, 6 * hidp) && gates.stride == 6 * hidp && okf(c, hidp) && okf(m, hidp), "stlstm gates bwd: gates [..,6*hidp] dense, c, m [..,hidp]");
  SF_REQUIRE(okf(dgx, 7 * hidp) && dgx.stride == 7 * hidp && okf(dgh, 4 * hidp) && dgh.stride == 4 * hidp && okf(dgm, 3 * hidp) && dgm.stride == 3 * hidp &&
                 okf(dc, hidp) && dc.stride == hidp && okf(dm, hidp) && dm.stride == hidp, "stlstm gates bwd: outputs must be dense fp32 tensors");
  if (pixels <= 0) return 0;
  StGatesBwdParams p{};
  p.d_cnew = (const float*)d_c_new.ptr; p.d_mnew = (const float*)d_m_new.ptr; p.d_mem = (const float*)d_mem.ptr;
  p.d_dc = (const float*)d_delta_c.ptr; p.d_dm = (const float*)d_delta_m.ptr; p.d_po = (const float*)d_pre_o.ptr;
  p.gates = (const float*)gates.ptr; p.c = (const float*)c.ptr; p.m = (const float*)m.ptr; p.s_c = c.stride; p.s_m = m.stride;
  p.dgx = (float*)dgx.ptr; p.dgh = (float*)dgh.ptr; p.dgm = (float*)dgm.ptr; p.dc = (float*)dc.ptr; p.dm = (float*)dm.ptr;
  p.pixels = pixels; p.hidp = hidp;
  hipLaunchKernelGGL(stlstm_gates_bwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("stlstm_gates_bwd");
  return 0;
}

static bool ln_ok(const sfTensor& t, int lanes) { return t.ptr && t.dtype == SF_F32 && t.c == lanes && t.stride == lanes; }

int sf_layernorm_chw_fwd(sfTensor x, int64_t n, int64_t pixels, int32_t gates, int32_t hid, int32_t hidp, const float* gamma, const float* beta, float eps,
                         double* partial, sfTensor y, sfStream stream) {
  SF_REQUIRE(gates >= 1 && hid >= 1 && hidp >= hid && ln_ok(x, gates * hidp) && ln_ok(y, gates * hidp) && gamma && beta && partial,
             "sf_layernorm_chw_fwd: dense fp32 x / y of gates * hidp lanes, affine parameters, partial sums buffer");
  if (n <= 0 || pixels <= 0) return 0;
  const LnGeom g{gates, hid, hidp, (long long)pixels};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ln_sums_kernel, dim3(LN_SLICES, (unsigned)n), dim3(256), 0, st, (const float*)x.ptr, (const float*)nullptr, (const float*)nullptr,
                     (const double*)nullptr, eps, g, partial);
  const double cnt = (double)pixels * gates * hid;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, partial, (long long)n, cnt, eps, 0);
  hipLaunchKernelGGL(ln_apply_kernel, dim3(grid_of(n * pixels * gates * hidp)), dim3(256), 0, st, (const float*)x.ptr, (const double*)partial, gamma, beta, eps, g,
                     (long long)n, (float*)y.ptr);
  SF_CHECK_LAUNCH("layernorm_chw_fwd");
  return 0;
}

int sf_layernorm_chw_bwd(sfTensor x, sfTensor dy, int64_t n, int64_t pixels, int32_t gates, int32_t hid, int32_t hidp, const float* gamma, float eps,
                         const double* partial, double* bwd_partial, sfTensor dx, float* dgamma, float* dbeta, sfStream stream) {
  SF_REQUIRE(gates >= 1 && hid >= 1 && hidp >= hid && ln_ok(x, gates * hidp) && ln_ok(dy, gates * hidp) && ln_ok(dx, gates * hidp) && gamma && partial && bwd_partial &&
                 dgamma && dbeta, "sf_layernorm_chw_bwd: dense fp32 tensors of gates * hidp lanes, gamma, the forward's partial sums, scratch, parameter gradients");
  if (n <= 0 || pixels <= 0) return 0;
  const LnGeom g{gates, hid, hidp, (long long)pixels};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ln_sums_kernel, dim3(LN_SLICES, (unsigned)n), dim3(256), 0, st, (const float*)x.ptr, (const float*)dy.ptr, gamma, partial, eps, g, bwd_partial);
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, bwd_partial, (long long)n, (double)pixels * gates * hid, eps, 1);
  hipLaunchKernelGGL(ln_bwd_apply_kernel, dim3(grid_of(n * pixels * gates * hidp)), dim3(256), 0, st, (const float*)x.ptr, (const float*)dy.ptr, partial,
                     (const double*)bwd_partial, gamma, eps, g, (long long)n, (float*)dx.ptr);
  hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(grid_of(pixels * gates * hidp)), dim3(256), 0, st, (const float*)x.ptr, (const float*)dy.ptr, partial, eps, g,
                     (long long)n, dgamma, dbeta);
  SF_CHECK_LAUNCH("layernorm_chw_bwd");
  return 0;
}

int sf_stlstm_out_fwd(sfTensor pre_o, sfTensor conv_o, sfTensor last, int64_t pixels, int32_t hidp, sfTensor h_new, sfTensor saved, int32_t dtype,
                      sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_stlstm_out_fwd: dtype %d not built", dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && hidp > 0, "stlstm out: hidp=%d", hidp);
  SF_REQUIRE(okf(pre_o, hidp) && pre_o.stride == hidp && okf(conv_o, hidp) && okf(last, hidp) && okf(h_new, hidp) && h_new.stride == hidp &&
                 (!saved.ptr || (okf(saved, 2 * hidp) && saved.stride == 2 * hidp)),
             "stlstm out: pre_o, h_new dense [..,hidp]; conv_o, last [..,>=hidp]; saved dense [..,2*hidp] or null (fp32, 16-byte aligned)");
  if (pixels <= 0) return 0;
  hipLaunchKernelGGL(stlstm_out_fwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)pre_o.ptr,
                     (const float*)conv_o.ptr, conv_o.stride, (const float*)last.ptr, last.stride, (long long)pixels, hidp, (float*)h_new.ptr, (float*)saved.ptr);
  SF_CHECK_LAUNCH("stlstm_out_fwd");
  return 0;
}

int sf_stlstm_out_bwd(sfTensor dh, sfTensor saved, int64_t pixels, int32_t hidp, sfTensor d_a, sfTensor d_last, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_stlstm_out_bwd: dtype %d not built", dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && hidp > 0, "stlstm out bwd: hidp=%d", hidp);
  SF_REQUIRE(okf(dh, hidp) && dh.stride == hidp && okf(saved, 2 * hidp) && saved.stride == 2 * hidp && okf(d_a, hidp) && d_a.stride == hidp && okf(d_last, hidp) &&
                 d_last.stride == hidp, "stlstm out bwd: dense fp32 tensors, 16-byte aligned");
  if (pixels <= 0) return 0;
  hipLaunchKernelGGL(stlstm_out_bwd_kernel, dim3(grid_of(pixels * (hidp / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)dh.ptr, (const float*)saved.ptr,
                     (long long)pixels, hidp, (float*)d_a.ptr, (float*)d_last.ptr);
  SF_CHECK_LAUNCH("stlstm_out_bwd");
  return 0;
}

}  // extern "C"
