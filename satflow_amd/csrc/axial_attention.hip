// Axial attention core (lucidrains AxialAttention(dim, dim_index=1, heads=8, num_dimensions=2) as used
// by metnet.MetNet.temporal_agg; reference call site satflow/models/pl_metnet.py:46-59).
//
// Input  qkv [img][H][W][6*hidp] = [q0|k0|v0|q1|k1|v1]: projections for axis 0 (attend along H, one
//        sequence per column) and axis 1 (along W, one sequence per row); channel = head*e + j.
// Output att [img][H][W][2*hidp] = [axis-0 result | axis-1 result] (heads merged); the caller's
// out-projection over the concatenation IS the sum of the two axes' to_out (sum_axial_out).
//
// Sequences are <= 32 long and e = hid/heads is tiny, so a whole softmax row lives in one lane's
// registers: one thread per (pixel, axis, head); no LDS, no cross-lane reduction, no atomics.  The
// backward is gather-form as well (each thread recomputes the softmax rows of its line), so it is
// deterministic.  Work is ~10 MFLOP per image: latency-bound by construction.
#include "sf_common.h"

namespace {

constexpr int MAXL = 32;  // max sequence length along an axis
constexpr int MAXE = 16;  // max head dim

struct AttnParams {
  const float* qkv; int qs;
  float* att; int as;
  const float* datt; int das;
  float* dqkv; int dqs;
  long long nimg; int H, W, hid, hidp, heads;
  float scale;
};

struct Line { long long base; int len, step, pos; };  // pixel index = base + i*step, own position pos

__device__ __forceinline__ Line line_of(long long img, int y, int x, int axis, int H, int W) {
  Line l;
  if (axis == 0) { l.base = img * H * W + x; l.len = H; l.step = W; l.pos = y; }
  else           { l.base = (img * H + y) * (long long)W; l.len = W; l.step = 1; l.pos = x; }
  return l;
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  const int e = p.hid / p.heads;
  const long long total = p.nimg * p.H * p.W * 2 * p.heads;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    // head fastest, then axis, then pixel: neighbouring lanes share the line's k/v cache lines
    const int head = idx % p.heads;
    const int axis = (idx / p.heads) % 2;
    const long long pix = idx / (2 * p.heads);
    const int x = pix % p.W, y = (pix / p.W) % p.H;
    const long long img = pix / ((long long)p.W * p.H);
    const Line ln = line_of(img, y, x, axis, p.H, p.W);
    const int off = axis * 3 * p.hidp + head * e;
    float q[MAXE];
    for (int j = 0; j < e; ++j) q[j] = p.qkv[pix * p.qs + off + j] * p.scale;
    float s[MAXL];
    float m = -INFINITY;
    for (int i = 0; i < ln.len; ++i) {
      const float* k = p.qkv + (ln.base + (long long)i * ln.step) * p.qs + off + p.hidp;
      float d = 0.f;
      for (int j = 0; j < e; ++j) d += q[j] * k[j];
      s[i] = d; m = fmaxf(m, d);
    }
    float z = 0.f;
    for (int i = 0; i < ln.len; ++i) { s[i] = expf(s[i] - m); z += s[i]; }
    const float inv = 1.f / z;
    float o[MAXE];
    for (int j = 0; j < e; ++j) o[j] = 0.f;
    for (int i = 0; i < ln.len; ++i) {
      const float* v = p.qkv + (ln.base + (long long)i * ln.step) * p.qs + off + 2 * p.hidp;
      const float w = s[i] * inv;
      for (int j = 0; j < e; ++j) o[j] += w * v[j];
    }
    float* out = p.att + pix * p.as + axis * p.hidp + head * e;
    for (int j = 0; j < e; ++j) out[j] = o[j];
  }
}

// Square LxL maps with head dim E (MetNet: 16x16, E = 8 or 4): sequence length and head dim are compile-time constants, so
// every loop is unrolled, k / v rows come in as float4 and the loads of a whole line are in flight together (the generic
// kernel above walks a runtime-length loop of scalar loads: 205 us for 96 images against ~40 MB of traffic).
template <int L, int E>
__global__ __launch_bounds__(256) void attn_fwd_sq_kernel(const AttnParams p) {
  constexpr int Q = E / 4;
  const long long total = p.nimg * L * L * 2 * p.heads;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int head = idx % p.heads;
  const int axis = (idx / p.heads) % 2;
  const long long pix = idx / (2 * p.heads);
  const int x = pix % L, y = (pix / L) % L;
  const long long img = pix / (L * L);
  const long long base = axis == 0 ? img * L * L + x : (img * L + y) * (long long)L;
  const int step = axis == 0 ? L : 1;
  const int off = axis * 3 * p.hidp + head * E;
  f32x4 q[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) q[j] = *reinterpret_cast<const f32x4*>(p.qkv + pix * p.qs + off + 4 * j) * p.scale;
  float s[L];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const float* k = p.qkv + (base + (long long)i * step) * p.qs + off + p.hidp;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(k + 4 * j);
      d += q[j][0] * kv[0] + q[j][1] * kv[1] + q[j][2] * kv[2] + q[j][3] * kv[3];
    }
    s[i] = d; m = fmaxf(m, d);
  }
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < L; ++i) { s[i] = expf(s[i] - m); z += s[i]; }
  const float inv = 1.f / z;
  f32x4 o[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const float* v = p.qkv + (base + (long long)i * step) * p.qs + off + 2 * p.hidp;
    const float w = s[i] * inv;
#pragma unroll
    for (int j = 0; j < Q; ++j) o[j] += *reinterpret_cast<const f32x4*>(v + 4 * j) * w;
  }
  float* out = p.att + pix * p.as + axis * p.hidp + head * E;
#pragma unroll
  for (int j = 0; j < Q; ++j) *reinterpret_cast<f32x4*>(out + 4 * j) = o[j];
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnParams p) {
  const int e = p.hid / p.heads;
  const long long total = p.nimg * p.H * p.W * 2 * p.heads;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int head = idx % p.heads;
    const int axis = (idx / p.heads) % 2;
    const long long pix = idx / (2 * p.heads);
    const int x = pix % p.W, y = (pix / p.W) % p.H;
    const long long img = pix / ((long long)p.W * p.H);
    const Line ln = line_of(img, y, x, axis, p.H, p.W);
    const int off = axis * 3 * p.hidp + head * e;
    const int doff = axis * p.hidp + head * e;
    float dq[MAXE], dk[MAXE], dv[MAXE];
    for (int j = 0; j < e; ++j) dq[j] = dk[j] = dv[j] = 0.f;
    // every query row r of this line: its softmax, then the pieces that land on this pixel
    for (int rq = 0; rq < ln.len; ++rq) {
      const long long rp = ln.base + (long long)rq * ln.step;
      float q[MAXE], g[MAXE];
      for (int j = 0; j < e; ++j) { q[j] = p.qkv[rp * p.qs + off + j]; g[j] = p.datt[rp * p.das + doff + j]; }
      float s[MAXL], dP[MAXL];
      float m = -INFINITY;
      for (int i = 0; i < ln.len; ++i) {
        const float* kv = p.qkv + (ln.base + (long long)i * ln.step) * p.qs + off;
        float d = 0.f, dp = 0.f;
        for (int j = 0; j < e; ++j) { d += q[j] * kv[p.hidp + j]; dp += g[j] * kv[2 * p.hidp + j]; }
        s[i] = d * p.scale; dP[i] = dp; m = fmaxf(m, s[i]);
      }
      float z = 0.f;
      for (int i = 0; i < ln.len; ++i) { s[i] = expf(s[i] - m); z += s[i]; }
      const float inv = 1.f / z;
      float spd = 0.f;
      for (int i = 0; i < ln.len; ++i) { s[i] *= inv; spd += s[i] * dP[i]; }
      // as key/value position `pos` of query row rq
      const float pj = s[ln.pos];
      const float dsj = pj * (dP[ln.pos] - spd) * p.scale;
      for (int j = 0; j < e; ++j) { dv[j] += pj * g[j]; dk[j] += dsj * q[j]; }
      // as the query itself
      if (rq == ln.pos) {
        for (int i = 0; i < ln.len; ++i) {
          const float ds = s[i] * (dP[i] - spd) * p.scale;
          const float* k = p.qkv + (ln.base + (long long)i * ln.step) * p.qs + off + p.hidp;
          for (int j = 0; j < e; ++j) dq[j] += ds * k[j];
        }
      }
    }
    float* d = p.dqkv + pix * p.dqs + off;
    for (int j = 0; j < e; ++j) { d[j] = dq[j]; d[p.hidp + j] = dk[j]; d[2 * p.hidp + j] = dv[j]; }
  }
}


// Backward, block per (image, axis, head): one thread per pixel.  Each thread computes ITS query row's
// softmax P and dS once, parks both rows in LDS, and after a barrier gathers the columns it needs as key/value
// position - L x less arithmetic than the gather-from-scratch kernel above (which stays as the fallback for
// images with more than 1024 pixels).  Deterministic (no atomics).
__global__ void attn_bwd_block_kernel(const AttnParams p) {
  extern __shared__ float sm[];  // P rows [pix][LS] then dS rows [pix][LS]
  const int e = p.hid / p.heads;
  const int npix = p.H * p.W;
  const int LS = (p.H > p.W ? p.H : p.W) + 1;
  float* Pl = sm;
  float* Dl = sm + (size_t)npix * LS;
  const long long img = blockIdx.x;
  const int t = threadIdx.x;
  const bool live = t < npix;
  const int x = live ? t % p.W : 0, y = live ? t / p.W : 0;
  const long long pix = img * npix + t;
  {
    const int ah = blockIdx.y;  // (axis, head) pairs are independent: 2*heads workgroups per image instead of a serial loop
    const int axis = ah / p.heads, head = ah % p.heads;
    const Line ln = line_of(img, y, x, axis, p.H, p.W);
    const int off = axis * 3 * p.hidp + head * e;
    const int doff = axis * p.hidp + head * e;
    if (live) {
      float q[MAXE], g[MAXE];
      for (int j = 0; j < e; ++j) { q[j] = p.qkv[pix * p.qs + off + j]; g[j] = p.datt[pix * p.das + doff + j]; }
      float s[MAXL], dP[MAXL];
      float m = -INFINITY;
      for (int i = 0; i < ln.len; ++i) {
        const float* kv = p.qkv + (ln.base + (long long)i * ln.step) * p.qs + off;
        float d = 0.f, dp = 0.f;
        for (int j = 0; j < e; ++j) { d += q[j] * kv[p.hidp + j]; dp += g[j] * kv[2 * p.hidp + j]; }
        s[i] = d * p.scale; dP[i] = dp; m = fmaxf(m, s[i]);
      }
      float z = 0.f;
      for (int i = 0; i < ln.len; ++i) { s[i] = expf(s[i] - m); z += s[i]; }
      const float inv = 1.f / z;
      float spd = 0.f;
      for (int i = 0; i < ln.len; ++i) { s[i] *= inv; spd += s[i] * dP[i]; }
      float dq[MAXE];
      for (int j = 0; j < e; ++j) dq[j] = 0.f;
      for (int i = 0; i < ln.len; ++i) {
        const float ds = s[i] * (dP[i] - spd) * p.scale;
        Pl[t * LS + i] = s[i];
        Dl[t * LS + i] = ds;
        const float* k = p.qkv + (ln.base + (long long)i * ln.step) * p.qs + off + p.hidp;
        for (int j = 0; j < e; ++j) dq[j] += ds * k[j];
      }
      float* d = p.dqkv + pix * p.dqs + off;
      for (int j = 0; j < e; ++j) d[j] = dq[j];
    }
    __syncthreads();
    if (live) {
      float dk[MAXE], dv[MAXE];
      for (int j = 0; j < e; ++j) dk[j] = dv[j] = 0.f;
      for (int rq = 0; rq < ln.len; ++rq) {
        const long long rp = ln.base + (long long)rq * ln.step;      // global pixel of query row rq
        const int rt = (int)(rp - img * npix);                        // its thread / LDS row
        const float pj = Pl[rt * LS + ln.pos], dsj = Dl[rt * LS + ln.pos];
        const float* qr = p.qkv + rp * p.qs + off;
        const float* gr = p.datt + rp * p.das + doff;
        for (int j = 0; j < e; ++j) { dv[j] += pj * gr[j]; dk[j] += dsj * qr[j]; }
      }
      float* d = p.dqkv + pix * p.dqs + off;
      for (int j = 0; j < e; ++j) { d[p.hidp + j] = dk[j]; d[2 * p.hidp + j] = dv[j]; }
    }
  }
}

// The same block-per-(image, axis, head) backward for square LxL maps with head dim E: unrolled, float4 loads.
template <int L, int E>
__global__ __launch_bounds__(L * L) void attn_bwd_sq_kernel(const AttnParams p) {
  constexpr int Q = E / 4, LS = L + 1, NP = L * L;
  __shared__ float Pl[NP * LS], Dl[NP * LS];
  const long long img = blockIdx.x;
  const int t = threadIdx.x;
  const int x = t % L, y = t / L;
  const long long pix = img * NP + t;
  const int ah = blockIdx.y;
  const int axis = ah / p.heads, head = ah % p.heads;
  const long long base = axis == 0 ? img * NP + x : (img * L + y) * (long long)L;
  const int step = axis == 0 ? L : 1, pos = axis == 0 ? y : x;
  const int off = axis * 3 * p.hidp + head * E;
  const int doff = axis * p.hidp + head * E;
  f32x4 q[Q], g[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    q[j] = *reinterpret_cast<const f32x4*>(p.qkv + pix * p.qs + off + 4 * j);
    g[j] = *reinterpret_cast<const f32x4*>(p.datt + pix * p.das + doff + 4 * j);
  }
  float s[L], dP[L];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const float* kv = p.qkv + (base + (long long)i * step) * p.qs + off;
    float d = 0.f, dp = 0.f;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const f32x4 k = *reinterpret_cast<const f32x4*>(kv + p.hidp + 4 * j), v = *reinterpret_cast<const f32x4*>(kv + 2 * p.hidp + 4 * j);
      d += q[j][0] * k[0] + q[j][1] * k[1] + q[j][2] * k[2] + q[j][3] * k[3];
      dp += g[j][0] * v[0] + g[j][1] * v[1] + g[j][2] * v[2] + g[j][3] * v[3];
    }
    s[i] = d * p.scale; dP[i] = dp; m = fmaxf(m, s[i]);
  }
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < L; ++i) { s[i] = expf(s[i] - m); z += s[i]; }
  const float inv = 1.f / z;
  float spd = 0.f;
#pragma unroll
  for (int i = 0; i < L; ++i) { s[i] *= inv; spd += s[i] * dP[i]; }
  f32x4 dq[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) dq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const float ds = s[i] * (dP[i] - spd) * p.scale;
    Pl[t * LS + i] = s[i];
    Dl[t * LS + i] = ds;
    const float* k = p.qkv + (base + (long long)i * step) * p.qs + off + p.hidp;
#pragma unroll
    for (int j = 0; j < Q; ++j) dq[j] += *reinterpret_cast<const f32x4*>(k + 4 * j) * ds;
  }
  float* d = p.dqkv + pix * p.dqs + off;
#pragma unroll
  for (int j = 0; j < Q; ++j) *reinterpret_cast<f32x4*>(d + 4 * j) = dq[j];
  __syncthreads();
  f32x4 dk[Q], dv[Q];
#pragma unroll
  for (int j = 0; j < Q; ++j) dk[j] = dv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rq = 0; rq < L; ++rq) {
    const long long rp = base + (long long)rq * step;
    const int rt = (int)(rp - img * NP);
    const float pj = Pl[rt * LS + pos], dsj = Dl[rt * LS + pos];
    const float* qr = p.qkv + rp * p.qs + off;
    const float* gr = p.datt + rp * p.das + doff;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      dv[j] += *reinterpret_cast<const f32x4*>(gr + 4 * j) * pj;
      dk[j] += *reinterpret_cast<const f32x4*>(qr + 4 * j) * dsj;
    }
  }
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    *reinterpret_cast<f32x4*>(d + p.hidp + 4 * j) = dk[j];
    *reinterpret_cast<f32x4*>(d + 2 * p.hidp + 4 * j) = dv[j];
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same core on the matrix cores (16 x 16 maps, head dim 8: MetNet's temporal_agg; exact-fp32 `v_mfma_f32_16x16x4_f32`).
// One wave per (line, head): the line's 16 positions are the 16 rows / columns of the MFMA tile.
//   S^T[j][i] = sum_e K[j][e] Q[i][e]      2 MFMAs (e = 0..3, 4..7): lane (i = lane % 16, g = lane / 16) then holds the scores of QUERY i against
//                                          keys 4g .. 4g+3 ("query-major" layout: A operand lane (j, kk) = K[j][kk], B operand lane (i, kk) = Q[i][kk])
//   softmax over the keys of a query       in-lane over the 4 registers, across the 4 lane groups by two xor-shuffles (16, 32)
//   O[i][e]   = sum_j P[i][j] V[j][e]      4 MFMAs with the K index permuted (step r uses keys r, 4+r, 8+r, 12+r): the A operand of step r is
//                                          simply register r of P - no data movement; B operand lane (e, g) = V[4g + r][e] (e < 8, else 0)
// The backward also needs P and dS with a KEY per lane (dK = dS^T Q, dV = P^T dO): the same two products with the operands swapped land in that
// layout directly (2 + 2 more MFMAs; the row statistics come over by ds_bpermute) - 20 MFMAs per (line, head), no LDS memory, no atomics.
// Block = (image, axis, quarter of the lines), one line per wave, its 8 heads back to back.  sf_axial_attention_core_fwd/bwd, reference call site satflow/models/pl_metnet.py:46-59.
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct AttnLine {  // per-lane addressing of one line: pixel of position `pos` = base + pos * step
  long long base; int step;
};
__device__ __forceinline__ float shfl_xor_f(float v, int m) { return __shfl_xor(v, m); }

// A wave first pulls its line's rows - 16 positions x [q | k | v] = 192 floats each - into LDS with 16-byte loads (all in flight at once), pitch 196
// floats: the per-head operand reads ([pos][column]: 16 positions x 4 columns, or 4 positions x 16 columns) then hit 64 different banks.  The
// backward writes dq / dk / dv of a head over that head's q / k / v columns (nobody reads them again) and the whole line leaves with 16-byte stores.
constexpr int AT_PITCH = 196, AT_OPITCH = 68;

__device__ __forceinline__ void attn_load_rows(const float* __restrict__ src, long long base, int step, int stride, int col0, int ncol4, int lane,
                                               float* lds_rows, int pitch) {
  // rows of ncol4 float4 starting at column col0, 16 positions
  for (int idx = lane; idx < 16 * ncol4; idx += 64) {
    const int pos = idx / ncol4, c4 = idx - pos * ncol4;
    *reinterpret_cast<f32x4v*>(lds_rows + pos * pitch + 4 * c4) = *reinterpret_cast<const f32x4v*>(src + (base + (long long)pos * step) * stride + col0 + 4 * c4);
  }
}
__device__ __forceinline__ void attn_store_rows(float* __restrict__ dst, long long base, int step, int stride, int col0, int ncol4, int lane,
                                                const float* lds_rows, int pitch) {
  for (int idx = lane; idx < 16 * ncol4; idx += 64) {
    const int pos = idx / ncol4, c4 = idx - pos * ncol4;
    *reinterpret_cast<f32x4v*>(dst + (base + (long long)pos * step) * stride + col0 + 4 * c4) = *reinterpret_cast<const f32x4v*>(lds_rows + pos * pitch + 4 * c4);
  }
}

__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const AttnParams p) {
  constexpr int L = 16, E = 8;
  __shared__ __attribute__((aligned(16))) float rows[4][L * AT_PITCH], outs[4][L * AT_OPITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long img = blockIdx.x >> 3;
  const int axis = (blockIdx.x >> 2) & 1;
  const int pos = lane & 15, g = lane >> 4;
  const int line = 4 * (blockIdx.x & 3) + wave;  // one line per wave: 16 x 2 x nimg waves, the eight heads of a line back to back
  // axis 0: attend along H, the line is column `line`; axis 1: along W, the line is row `line`
  const long long base = axis == 0 ? img * L * L + line : (img * L + line) * (long long)L;
  const int step = axis == 0 ? L : 1;
  float* R = rows[wave];
  float* O = outs[wave];
  attn_load_rows(p.qkv, base, step, p.qs, axis * 3 * p.hidp, 48, lane, R, AT_PITCH);
  __syncthreads();
#pragma unroll 2
  for (int head = 0; head < 8; ++head) {
    const int ho = head * E;
    const float* rp = R + pos * AT_PITCH + ho + g;
    const float q0 = rp[0], q1 = rp[4], k0 = rp[64], k1 = rp[68];
    float vv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) vv[r] = pos < E ? R[(4 * g + r) * AT_PITCH + 128 + ho + pos] : 0.f;
    f32x4v st = {0.f, 0.f, 0.f, 0.f};
    st = __builtin_amdgcn_mfma_f32_16x16x4f32(k0, q0, st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, q1, st, 0, 0, 0);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] *= p.scale; m = fmaxf(m, st[r]); }
    m = fmaxf(m, shfl_xor_f(m, 16)); m = fmaxf(m, shfl_xor_f(m, 32));
    float z = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] = expf(st[r] - m); z += st[r]; }
    z += shfl_xor_f(z, 16); z += shfl_xor_f(z, 32);
    const float inv = 1.f / z;
    f32x4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(st[r] * inv, vv[r], o, 0, 0, 0);
    if (pos < E) {
#pragma unroll
      for (int r = 0; r < 4; ++r) O[(4 * g + r) * AT_OPITCH + ho + pos] = o[r];
    }
  }
  __syncthreads();
  attn_store_rows(p.att, base, step, p.as, axis * p.hidp, 16, lane, O, AT_OPITCH);
}

__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const AttnParams p) {
  constexpr int L = 16, E = 8;
  __shared__ __attribute__((aligned(16))) float rows[4][L * AT_PITCH], grows[4][L * AT_OPITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long img = blockIdx.x >> 3;
  const int axis = (blockIdx.x >> 2) & 1;
  const int pos = lane & 15, g = lane >> 4;
  const int line = 4 * (blockIdx.x & 3) + wave;
  const long long base = axis == 0 ? img * L * L + line : (img * L + line) * (long long)L;
  const int step = axis == 0 ? L : 1;
  float* R = rows[wave];
  float* G = grows[wave];
  attn_load_rows(p.qkv, base, step, p.qs, axis * 3 * p.hidp, 48, lane, R, AT_PITCH);
  attn_load_rows(p.datt, base, step, p.das, axis * p.hidp, 16, lane, G, AT_OPITCH);
  __syncthreads();
#pragma unroll 2
  for (int head = 0; head < 8; ++head) {
    const int ho = head * E;
    // operands with this lane's own position (lane (pos, kk = g)): e = g and g + 4
    const float* rp = R + pos * AT_PITCH + ho + g;
    const float q0 = rp[0], q1 = rp[4], k0 = rp[64], k1 = rp[68], v0 = rp[128], v1 = rp[132];
    const float d0 = G[pos * AT_OPITCH + ho + g], d1 = G[pos * AT_OPITCH + ho + g + 4];
    // operands with positions 4g + r and channel e = pos (B operands of the K = position products)
    float kb[4], qb[4], db[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* rr = R + (4 * g + r) * AT_PITCH + ho + pos;
      qb[r] = pos < E ? rr[0] : 0.f;
      kb[r] = pos < E ? rr[64] : 0.f;
      db[r] = pos < E ? G[(4 * g + r) * AT_OPITCH + ho + pos] : 0.f;
    }
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
    // query-major: lane (i, g) <- X[i][4g + r]
    f32x4v st = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, q1, __builtin_amdgcn_mfma_f32_16x16x4f32(k0, q0, zero, 0, 0, 0), 0, 0, 0);
    f32x4v dpt = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, d1, __builtin_amdgcn_mfma_f32_16x16x4f32(v0, d0, zero, 0, 0, 0), 0, 0, 0);
    // key-major: lane (j, g) <- X[4g + r][j]
    f32x4v sn = __builtin_amdgcn_mfma_f32_16x16x4f32(q1, k1, __builtin_amdgcn_mfma_f32_16x16x4f32(q0, k0, zero, 0, 0, 0), 0, 0, 0);
    f32x4v dpn = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, v1, __builtin_amdgcn_mfma_f32_16x16x4f32(d0, v0, zero, 0, 0, 0), 0, 0, 0);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] *= p.scale; m = fmaxf(m, st[r]); }
    m = fmaxf(m, shfl_xor_f(m, 16)); m = fmaxf(m, shfl_xor_f(m, 32));
    float z = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] = expf(st[r] - m); z += st[r]; }
    z += shfl_xor_f(z, 16); z += shfl_xor_f(z, 32);
    const float inv = 1.f / z;
    float spd = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] *= inv; spd += st[r] * dpt[r]; }
    spd += shfl_xor_f(spd, 16); spd += shfl_xor_f(spd, 32);
    f32x4v dst;  // dS, query-major
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = st[r] * (dpt[r] - spd) * p.scale;
    // the statistics of query 4g + r, for the key-major layout (any lane of that query holds them: take lane 4g + r)
    f32x4v pn, dsn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int src = 4 * g + r;
      const float mi = __shfl(m, src), ii = __shfl(inv, src), si = __shfl(spd, src);
      const float pr = expf(sn[r] * p.scale - mi) * ii;
      pn[r] = pr;
      dsn[r] = pr * (dpn[r] - si) * p.scale;
    }
    f32x4v dq = zero, dk = zero, dv = zero;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dq = __builtin_amdgcn_mfma_f32_16x16x4f32(dst[r], kb[r], dq, 0, 0, 0);   // dQ[i][e] = sum_j dS[i][j] K[j][e]
      dk = __builtin_amdgcn_mfma_f32_16x16x4f32(dsn[r], qb[r], dk, 0, 0, 0);   // dK[j][e] = sum_i dS[i][j] Q[i][e]
      dv = __builtin_amdgcn_mfma_f32_16x16x4f32(pn[r], db[r], dv, 0, 0, 0);    // dV[j][e] = sum_i P[i][j] dO[i][e]
    }
    if (pos < E) {  // over this head's q / k / v columns: every read of them is above
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* d = R + (4 * g + r) * AT_PITCH + ho + pos;
        d[0] = dq[r]; d[64] = dk[r]; d[128] = dv[r];
      }
    }
  }
  __syncthreads();
  attn_store_rows(p.dqkv, base, step, p.dqs, axis * 3 * p.hidp, 48, lane, R, AT_PITCH);
}

int check(const AttnParams& p, sfTensor qkv, int c_other, const char* what) {
  if (p.hid % p.heads != 0 || p.hid / p.heads > MAXE || p.H > MAXL || p.W > MAXL || p.hid > p.hidp || qkv.c < 6 * p.hidp || c_other < 2 * p.hidp) {
    sf_set_error("%s: unsupported shape hid=%d heads=%d H=%d W=%d (need hid%%heads==0, hid/heads<=%d, H,W<=%d)", what, p.hid, p.heads, p.H,
                 p.W, MAXE, MAXL);
    return 1;
  }
  return 0;
}

}  // namespace

extern "C" {

int sf_axial_attention_core_fwd(sfTensor qkv, int64_t nimg, int32_t h, int32_t w, int32_t hid, int32_t hidp, int32_t heads, sfTensor att,
                                int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_axial_attention_core_fwd: dtype %d not built", dtype);
  SF_F32_ONLY(qkv, "sf_axial_attention_core_fwd");
  SF_F32_ONLY(att, "sf_axial_attention_core_fwd");
  AttnParams p{};
  p.qkv = (const float*)qkv.ptr; p.qs = qkv.stride; p.att = (float*)att.ptr; p.as = att.stride;
  p.nimg = nimg; p.H = h; p.W = w; p.hid = hid; p.hidp = hidp; p.heads = heads;
  if (check(p, qkv, att.c, "axial_attention fwd")) return 1;
  p.scale = 1.0f / sqrtf((float)(hid / heads));
  const long long total = nimg * h * w * 2 * heads;
  if (total == 0) return 0;
  // pad lanes of att (hid..hidp) are never written by the kernel: zero them once
  if (hid < hidp) SF_REQUIRE(sf_fill_async(att.ptr, 0, (size_t)nimg * h * w * att.stride * sizeof(float), (hipStream_t)stream) == hipSuccess, "memset");
  const int e = hid / heads;
  const bool vec = h == w && (((uintptr_t)qkv.ptr | (uintptr_t)att.ptr) & 15) == 0 && qkv.stride % 4 == 0 && att.stride % 4 == 0 && hidp % 4 == 0;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  static const bool no_mfma = getenv("SF_ATTN_NO_MFMA") != nullptr;  // A/B switch: the thread-per-row kernels
  if (!no_mfma && vec && h == 16 && w == 16 && e == 8 && heads == 8 && hidp == 64) hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3((unsigned)(8 * nimg)), block, 0, (hipStream_t)stream, p);
  else if (vec && h == 16 && e == 8) hipLaunchKernelGGL((attn_fwd_sq_kernel<16, 8>), grid, block, 0, (hipStream_t)stream, p);
  else if (vec && h == 16 && e == 4) hipLaunchKernelGGL((attn_fwd_sq_kernel<16, 4>), grid, block, 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("axial_attention_fwd");
  return 0;
}

int sf_axial_attention_core_bwd(sfTensor qkv, sfTensor datt, int64_t nimg, int32_t h, int32_t w, int32_t hid, int32_t hidp, int32_t heads,
                                sfTensor dqkv, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_axial_attention_core_bwd: dtype %d not built", dtype);
  SF_F32_ONLY(qkv, "sf_axial_attention_core_bwd");
  SF_F32_ONLY(datt, "sf_axial_attention_core_bwd");
  SF_F32_ONLY(dqkv, "sf_axial_attention_core_bwd");
  AttnParams p{};
  p.qkv = (const float*)qkv.ptr; p.qs = qkv.stride; p.datt = (const float*)datt.ptr; p.das = datt.stride;
  p.dqkv = (float*)dqkv.ptr; p.dqs = dqkv.stride;
  p.nimg = nimg; p.H = h; p.W = w; p.hid = hid; p.hidp = hidp; p.heads = heads;
  if (check(p, qkv, datt.c, "axial_attention bwd")) return 1;
  SF_REQUIRE(dqkv.c >= 6 * hidp, "axial_attention bwd: dqkv lanes");
  p.scale = 1.0f / sqrtf((float)(hid / heads));
  const long long total = nimg * h * w * 2 * heads;
  if (total == 0) return 0;
  if (hid < hidp) SF_REQUIRE(sf_fill_async(dqkv.ptr, 0, (size_t)nimg * h * w * dqkv.stride * sizeof(float), (hipStream_t)stream) == hipSuccess, "memset");
  const int npix = h * w;
  const int e = hid / heads;
  const bool vec = h == w && (((uintptr_t)qkv.ptr | (uintptr_t)datt.ptr | (uintptr_t)dqkv.ptr) & 15) == 0 && qkv.stride % 4 == 0 &&
                   datt.stride % 4 == 0 && dqkv.stride % 4 == 0 && hidp % 4 == 0;
  static const bool no_mfma = getenv("SF_ATTN_NO_MFMA") != nullptr;
  if (!no_mfma && vec && h == 16 && w == 16 && e == 8 && heads == 8 && hidp == 64) {
    hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3((unsigned)(8 * nimg)), dim3(256), 0, (hipStream_t)stream, p);
  } else if (vec && h == 16 && (e == 8 || e == 4)) {
    const dim3 grid((unsigned)nimg, 2 * heads), block(256);
    if (e == 8) hipLaunchKernelGGL((attn_bwd_sq_kernel<16, 8>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_bwd_sq_kernel<16, 4>), grid, block, 0, (hipStream_t)stream, p);
  } else if (npix <= 1024) {
    const int threads = (npix + 63) / 64 * 64;
    const int LS = (h > w ? h : w) + 1;
    const size_t shmem = (size_t)2 * npix * LS * sizeof(float);
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_bwd_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(attn_bwd_block_kernel, dim3((unsigned)nimg, 2 * heads), dim3(threads), shmem, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  }
  SF_CHECK_LAUNCH("axial_attention_bwd");
  return 0;
}

}  // extern "C"
