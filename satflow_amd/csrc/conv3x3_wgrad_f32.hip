// Weight / bias gradient of the 3x3 same convolution on the fp32 matrix cores.
//
//   dW[tap][co][ci] = sum over pixels  dout[pixel][co] * in[pixel + tap][ci]
//
// GEMM view: M = co, N = ci, K = pixels (one GEMM per tap, sharing the A operand).
// A workgroup (4 waves) owns a 128(co) x 32(ci) x 9(tap) slab of dW; wave w owns co rows
// 32w..32w+31 and keeps 9 accumulator tiles (144 VGPRs).  It walks its share of the K range in
// 4x16-pixel tiles: the dout tile [64 px][128 co] and the input halo tile [6x18 px][32 ci] are
// staged in LDS in their native NHWC order, which is already the layout the f32 MFMA wants
// (operand element per lane = one float, lanes along the channel axis -> conflict-free
// ds_read_b32); per K-step of 2 pixels: 1 A read + 9 shifted B reads feed 9 MFMAs.
// Split-K partial slabs go to a workspace and are reduced (deterministically, no atomics) by a
// second kernel that also scatters to the reference's OIHW gradient layout and sums the bias
// gradient, which the main kernel gets for free from the A operand it already holds.
#include "wgrad_common.h"

namespace {

using namespace sfwgrad;

constexpr int KT_H = 4;                            // K tile height (pixels); width KT_W = 16
constexpr int KT_PIX = KT_H * KT_W;                // 64
constexpr int HALO_H = KT_H + 2, HALO_W = KT_W + 2;
constexpr int PA = CO_T;                           // LDS floats per dout pixel
constexpr int PB = CI_T;                           // LDS floats per input pixel

__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const WgradParams p) {
  __shared__ __attribute__((aligned(16))) float lds[KT_PIX * PA + HALO_H * HALO_W * PB];
  float* lds_a = lds;
  float* lds_b = lds + KT_PIX * PA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, kh = lane >> 5;
  const int ks = blockIdx.x, cot = blockIdx.y, cit = blockIdx.z;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float asum = 0.f;

  for (int tile = ks; tile < p.ntiles; tile += p.KS) {
    int t = tile;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int x0 = tx * KT_W, y0 = ty * KT_H;
    __syncthreads();
    // dout tile: 64 px x 32 float4
    for (int pc = tid; pc < KT_PIX * (CO_T / 4); pc += 256) {
      const int pix = pc / (CO_T / 4), c4 = (pc % (CO_T / 4)) * 4;
      const int gy = y0 + (pix >> 4), gx = x0 + (pix & 15);
      const int co = cot * CO_T + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy < p.H && gx < p.W && co < p.dc)
        v = *reinterpret_cast<const f32x4*>(p.dout + ((size_t)(n * p.H + gy) * p.W + gx) * p.ds + co);
      *reinterpret_cast<f32x4*>(lds_a + pix * PA + c4) = v;
    }
    // input halo tile: 108 px x 8 float4
    for (int pc = tid; pc < HALO_H * HALO_W * (CI_T / 4); pc += 256) {
      const int pix = pc / (CI_T / 4), c4 = (pc % (CI_T / 4)) * 4;
      const int iy = pix / HALO_W, ix = pix - iy * HALO_W;
      const int gy = y0 + iy - 1, gx = x0 + ix - 1;
      const int kc = cit * CI_T + c4;  // channel in the concatenated padded K space
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
        if (kc < p.c0) {
          int ns = n / p.idiv0; if (p.imod0) ns %= p.imod0;
          v = *reinterpret_cast<const f32x4*>(p.src0 + ((size_t)(ns * p.H + gy) * p.W + gx) * p.s0 + kc);
        } else if (kc - p.c0 < p.c1) {
          int ns = n / p.idiv1; if (p.imod1) ns %= p.imod1;
          v = *reinterpret_cast<const f32x4*>(p.src1 + ((size_t)(ns * p.H + gy) * p.W + gx) * p.s1 + (kc - p.c0));
        }
      }
      *reinterpret_cast<f32x4*>(lds_b + pix * PB + c4) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int kstep = 0; kstep < KT_PIX / 2; ++kstep) {
      const int px = 2 * kstep + kh;  // pixel of this lane's k index
      const float a = lds_a[px * PA + 32 * wave + r];
      asum += a;
      const float* bb = lds_b + ((px >> 4) * HALO_W + (px & 15)) * PB + r;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float b = bb[((tap / 3) * HALO_W + (tap % 3)) * PB];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
      }
    }
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
  if (cit == 0) {
    const float tot = asum + __shfl_xor(asum, 32);
    if (kh == 0) p.partial_db[(size_t)ks * p.NpT + cot * CO_T + 32 * wave + r] = tot;
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_db, int KS, int NpT,
                                    int KpT, int Np, int Kp, const int* __restrict__ nmap, const int* __restrict__ kmap, int I,
                                    float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const size_t slab = (size_t)9 * NpT * KpT;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < slab) {
    const int ci = gid % KpT;
    const int co = (gid / KpT) % NpT;
    const int tap = gid / ((size_t)KpT * NpT);
    if (co < Np && ci < Kp) {
      const int o = nmap[co], i = kmap[ci];
      if (o >= 0 && i >= 0) {
        // 8 independent partial sums: 8 slab reads in flight instead of a load -> wait -> add chain (fixed order: deterministic)
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= KS; k += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) s8[u] += partial[(size_t)(k + u) * slab + gid];
        }
        for (; k < KS; ++k) s8[0] += partial[(size_t)k * slab + gid];
        const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        float* d = dw + ((size_t)o * I + i) * 9 + tap;
        *d = accumulate ? *d + s : s;
      }
    }
  }
  if (db && gid < (size_t)NpT) {
    const int co = (int)gid;
    if (co < Np && nmap[co] >= 0) {
      // same 8-chain form as above: a serial walk over KS slabs by this one workgroup was the tail of the whole kernel
      float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 8 <= KS; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] += partial_db[(size_t)(k + u) * NpT + co];
      }
      for (; k < KS; ++k) t8[0] += partial_db[(size_t)k * NpT + co];
      const float s = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
      float* d = db + nmap[co];
      *d = accumulate ? *d + s : s;
    }
  }
}

// ---- folded BatchNorm (sf_conv3x3_bwd_weight_folded): dW = sum_g scale_g (.) dWraw_g + shift_g (x) V_g ----
// V_g[tap][co] = sum of dout over the group's pixels whose tap neighbour lies inside the image = total - excluded border row -
// excluded border column + their corner.  The border sums are taken here from the bf16 dout (about 12 % of the tensor at 32x32),
// the group totals come from the main kernel's per-segment bias sums.
constexpr int BORDER_CATS = 8;  // top row, bottom row, left column, right column, corners TL, TR, BL, BR
constexpr int BORDER_IMGS = 2;  // images per workgroup (their loads are independent: both in flight)
__global__ __launch_bounds__(256) void fold_border_sums_kernel(const __bf16* __restrict__ dout, int ds, int Np, int H, int W, int imgs_per_group,
                                                              int chunks, float* __restrict__ bpart) {
  // block = (chunk of BORDER_IMGS images, group); thread = (pixel lane, channel octet); bpart[group][chunk][cat][Np]
  extern __shared__ float red[];  // [pixel lanes][8 cats][Np]
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  const int oct = Np / 8, plx = 256 / oct;  // Np <= 2048 (launcher)
  const int o = threadIdx.x % oct, pl = threadIdx.x / oct;
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int nb = 2 * W + 2 * (H - 2);  // border pixels of an image
  float acc[BORDER_CATS][8];
#pragma unroll
  for (int c = 0; c < BORDER_CATS; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[c][k] = 0.f;
  const int i0 = chunk * BORDER_IMGS;
  const size_t img_e = (size_t)H * W * ds;
  const __bf16* base = dout + (size_t)(g * imgs_per_group + i0) * img_e + 8 * o;
  const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pl < plx)
#pragma unroll 2
    for (int b = pl; b < nb; b += plx) {
      int y, x;
      if (b < W) { y = 0; x = b; }
      else if (b < 2 * W) { y = H - 1; x = b - W; }
      else { const int q = b - 2 * W; y = 1 + (q >> 1); x = (q & 1) ? W - 1 : 0; }
      bf16x8_t v[BORDER_IMGS];
#pragma unroll
      for (int im = 0; im < BORDER_IMGS; ++im)
        v[im] = i0 + im < imgs_per_group ? *reinterpret_cast<const bf16x8_t*>(base + im * img_e + (size_t)(y * W + x) * ds) : zero;
      const float top = y == 0 ? 1.f : 0.f, bot = y == H - 1 ? 1.f : 0.f, lef = x == 0 ? 1.f : 0.f, rig = x == W - 1 ? 1.f : 0.f;
      const float m[BORDER_CATS] = {top, bot, lef, rig, top * lef, top * rig, bot * lef, bot * rig};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float f = 0.f;
#pragma unroll
        for (int im = 0; im < BORDER_IMGS; ++im) f += (float)v[im][k];
#pragma unroll
        for (int c = 0; c < BORDER_CATS; ++c) acc[c][k] = __builtin_fmaf(m[c], f, acc[c][k]);
      }
    }
  // fixed-order reduction over the pixel lanes
  if (pl < plx)
#pragma unroll
    for (int c = 0; c < BORDER_CATS; ++c)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[((size_t)pl * BORDER_CATS + c) * Np + 8 * o + k] = acc[c][k];
  __syncthreads();
  for (int e = threadIdx.x; e < BORDER_CATS * Np; e += 256) {
    float s = 0.f;
    for (int q = 0; q < plx; ++q) s += red[(size_t)q * BORDER_CATS * Np + e];
    bpart[((size_t)g * chunks + chunk) * BORDER_CATS * Np + e] = s;
  }
}

// V_g[tap][co] from the group totals (per-segment bias sums of the main kernel) and the border sums; also the slot table of the
// reduction: slot (ks, seg) of the grouped main kernel belongs to group (ks * per_slice) / tpg + seg, unused slots get -1.
// block = (32 channels, group), 8 parts x 32 channels: the parts split the slices / chunks, fixed-order combine through LDS.
__global__ __launch_bounds__(256) void fold_v_kernel(const float* __restrict__ partial_db, int KS, int maxseg, int per_slice, int ntiles, int tpg,
                                                    int NpT, const float* __restrict__ bpart, int chunks, int Np, int groups,
                                                    float* __restrict__ V, int* __restrict__ slot_group) {
  __shared__ float red[8][BORDER_CATS + 1][32];
  const int lc = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int co = blockIdx.x * 32 + lc, g = blockIdx.y;
  if (blockIdx.x == 0 && g == 0)
    for (int s = threadIdx.x; s < KS * maxseg; s += 256) {
      const int ks = s / maxseg, seg = s - ks * maxseg, b = ks * per_slice;
      int gg = -1;
      if (b < ntiles) {
        const int e = (b + per_slice < ntiles ? b + per_slice : ntiles) - 1;
        if (seg <= e / tpg - b / tpg) gg = b / tpg + seg;
      }
      slot_group[s] = gg;
    }
  float S = 0.f;
  for (int ks = part; ks < KS; ks += 8) {
    const int b = ks * per_slice;
    if (b >= ntiles) break;
    const int e = (b + per_slice < ntiles ? b + per_slice : ntiles) - 1;
    const int g0 = b / tpg, g1 = e / tpg;
    if (g >= g0 && g <= g1) S += partial_db[((size_t)ks * maxseg + (g - g0)) * NpT + co];
  }
  red[part][BORDER_CATS][lc] = S;
  {
    float s[BORDER_CATS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (co < Np)
      for (int q = part; q < chunks; q += 8) {
        const float* bp = bpart + ((size_t)g * chunks + q) * BORDER_CATS * Np + co;
#pragma unroll
        for (int c = 0; c < BORDER_CATS; ++c) s[c] += bp[(size_t)c * Np];  // eight independent loads per trip
      }
#pragma unroll
    for (int c = 0; c < BORDER_CATS; ++c) red[part][c][lc] = s[c];
  }
  __syncthreads();
  if (part != 0) return;
  float B[BORDER_CATS + 1];
#pragma unroll
  for (int c = 0; c <= BORDER_CATS; ++c) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][c][lc];
    B[c] = s;
  }
  S = B[BORDER_CATS];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      // tap (ky, kx) reads the input at (y + ky - 1, x + kx - 1): excluded are the top row for ky == 0, the bottom row for ky == 2, ...
      float v = S;
      if (ky == 0) v -= B[0];
      if (ky == 2) v -= B[1];
      if (kx == 0) v -= B[2];
      if (kx == 2) v -= B[3];
      if (ky == 0 && kx == 0) v += B[4];
      if (ky == 0 && kx == 2) v += B[5];
      if (ky == 2 && kx == 0) v += B[6];
      if (ky == 2 && kx == 2) v += B[7];
      V[((size_t)g * 9 + ky * 3 + kx) * NpT + co] = v;
    }
}

// block = one (tap, co) row of the slab, threads over ci: slot table, V and the row index are block-uniform (scalar loads)
__global__ __launch_bounds__(256) void wgrad_reduce_folded_kernel(const float* __restrict__ partial, const float* __restrict__ partial_db, int slots,
                                                                 const int* __restrict__ slot_group, int groups, int NpT, int KpT, int Np, int Kp,
                                                                 const int* __restrict__ nmap, const int* __restrict__ kmap, int I,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ V, float* __restrict__ dw, float* __restrict__ db,
                                                                 int accumulate) {
  const size_t slab = (size_t)9 * NpT * KpT;
  const int row = blockIdx.x;  // tap * NpT + co
  const int co = row % NpT, tap = row / NpT;
  const int o = co < Np ? nmap[co] : -1;
  if (o >= 0)
    for (int ci = threadIdx.x; ci < Kp; ci += 256) {
      const int i = kmap[ci];
      if (i < 0) continue;
      const size_t gid = (size_t)row * KpT + ci;
      // fixed order: slots ascending over eight chains
      float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int s = 0;
      for (; s + 8 <= slots; s += 8) {
        // all eight slab reads requested before the first use: unconditional loads (an unused slot's slab is allocated, its garbage is
        // discarded by a select, never multiplied) - a load under `if (g >= 0)` is emitted as load -> wait -> use, one at a time
        float pv[8], sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int g = slot_group[s + u];
          pv[u] = partial[(size_t)(s + u) * slab + gid];
          sc[u] = scale[(size_t)(g >= 0 ? g : 0) * Kp + ci];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int g = slot_group[s + u];
          s8[u] = __builtin_fmaf(g >= 0 ? sc[u] : 0.f, g >= 0 ? pv[u] : 0.f, s8[u]);
        }
      }
      for (; s < slots; ++s) {
        const int g = slot_group[s];
        if (g >= 0) s8[0] = __builtin_fmaf(scale[(size_t)g * Kp + ci], partial[(size_t)s * slab + gid], s8[0]);
      }
      float r = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
      float t = 0.f;
      for (int g = 0; g < groups; ++g) t = __builtin_fmaf(shift[(size_t)g * Kp + ci], V[((size_t)g * 9 + tap) * NpT + co], t);
      r += t;
      float* d = dw + ((size_t)o * I + i) * 9 + tap;
      *d = accumulate ? *d + r : r;
    }
  if (db && tap == 0 && o >= 0 && threadIdx.x == 0) {
    float r = 0.f;
    for (int s = 0; s < slots; ++s)
      if (slot_group[s] >= 0) r += partial_db[(size_t)s * NpT + co];
    float* d = db + o;
    *d = accumulate ? *d + r : r;
  }
}

// ---- the BatchNorm backward's two reductions from the weight gradient's own partial results (no pass over the activation gradient) ----
// With dn = conv^T(dout, W) the gradient entering the folded BatchNorm:
//   sum_px dn[ci]         = sum_{tap, co} W[co][ci][tap] * V_g[tap][co]
//   sum_px dn[ci] * x[ci] = sum_{tap, co} W[co][ci][tap] * dWraw_g[co][ci][tap]      (dWraw_g = the group's raw weight gradient)
// Wt[tap][co][ci]: the weights in the slab layout (zero at pad lanes), so that the products run over coalesced rows.
__global__ void fold_wt_kernel(const float* __restrict__ w, int I, const int* __restrict__ nmap, const int* __restrict__ kmap, int Np, int Kp,
                               int NpT, int KpT, float* __restrict__ wt) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)9 * NpT * KpT) return;
  const int ci = gid % KpT, co = (gid / KpT) % NpT, tap = gid / ((size_t)KpT * NpT);
  float v = 0.f;
  if (co < Np && ci < Kp) {
    const int o = nmap[co], i = kmap[ci];
    if (o >= 0 && i >= 0) v = w[((size_t)o * I + i) * 9 + tap];
  }
  wt[gid] = v;
}

constexpr int S2_ROWPARTS = 8;
// block = (slot or slots + group, 64-lane ci tile, row part):
//   T[slot][part][ci]          = sum over the part's rows of Wt[row][ci] * partial[slot][row][ci]
//   T[slots + group][part][ci] = sum over the part's rows of Wt[row][ci] * V[group][row]
__global__ __launch_bounds__(256) void fold_bn_s2_kernel(const float* __restrict__ partial, const int* __restrict__ slot_group, int slots,
                                                        const float* __restrict__ wt, const float* __restrict__ V, int NpT, int KpT,
                                                        double* __restrict__ T) {
  __shared__ double red[4][64];
  const int slot = blockIdx.x, cit = blockIdx.y, part = blockIdx.z;
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int ci = cit * 64 + lc;
  const int rows = 9 * NpT, per = (rows + S2_ROWPARTS - 1) / S2_ROWPARTS;
  const int r0 = part * per, r1 = r0 + per < rows ? r0 + per : rows;
  double acc = 0.0;
  float a4[4] = {0.f, 0.f, 0.f, 0.f};  // short fp32 chains (16 terms each), combined in double
  int u = 0;
  if (slot >= slots) {  // block-uniform: the V part of group slot - slots
    const float* vg = V + (size_t)(slot - slots) * rows;
    for (int r = r0 + rl; r < r1; r += 4, ++u) {
      a4[u & 3] = __builtin_fmaf(wt[(size_t)r * KpT + ci], vg[r], a4[u & 3]);
      if ((u & 63) == 63) { acc += ((double)a4[0] + (double)a4[1]) + ((double)a4[2] + (double)a4[3]); a4[0] = a4[1] = a4[2] = a4[3] = 0.f; }
    }
  } else if (slot_group[slot] >= 0) {
    const float* ps = partial + (size_t)slot * rows * KpT;
    int r = r0 + rl;
    for (; r + 28 < r1; r += 32) {  // eight rows' loads in flight
      float wv[8], pv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { wv[k] = wt[(size_t)(r + 4 * k) * KpT + ci]; pv[k] = ps[(size_t)(r + 4 * k) * KpT + ci]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) a4[k & 3] = __builtin_fmaf(wv[k], pv[k], a4[k & 3]);
      u += 8;
      if ((u & 63) == 0) { acc += ((double)a4[0] + (double)a4[1]) + ((double)a4[2] + (double)a4[3]); a4[0] = a4[1] = a4[2] = a4[3] = 0.f; }
    }
    for (; r < r1; r += 4, ++u) a4[u & 3] = __builtin_fmaf(wt[(size_t)r * KpT + ci], ps[(size_t)r * KpT + ci], a4[u & 3]);
  }
  acc += ((double)a4[0] + (double)a4[1]) + ((double)a4[2] + (double)a4[3]);
  red[rl][lc] = acc;
  __syncthreads();
  if (rl == 0) T[((size_t)slot * S2_ROWPARTS + part) * KpT + ci] = (red[0][lc] + red[1][lc]) + (red[2][lc] + red[3][lc]);
}

// thread = (group, ci): sums[g][0][ci] = sum dn, sums[g][1][ci] = sum dn * xhat = rstd * (sum dn * x - mean * sum dn)
__global__ void fold_bn_sums_kernel(const double* __restrict__ T, const int* __restrict__ slot_group, int slots, int groups, int KpT, int C,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, double* __restrict__ sums) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= groups * C) return;
  const int g = id / C, ci = id - g * C;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int q = 0; q < S2_ROWPARTS; ++q) s1 += T[((size_t)(slots + g) * S2_ROWPARTS + q) * KpT + ci];
  for (int s = 0; s < slots; ++s)
    if (slot_group[s] == g) {
#pragma unroll
      for (int q = 0; q < S2_ROWPARTS; ++q) s2 += T[((size_t)s * S2_ROWPARTS + q) * KpT + ci];
    }
  sums[((size_t)g * 2 + 0) * C + ci] = s1;
  sums[((size_t)g * 2 + 1) * C + ci] = (double)rstd[(size_t)g * C + ci] * (s2 - (double)mean[(size_t)g * C + ci] * s1);
}

struct FoldLayout { int chunks; size_t bpart_off, v_off, slot_off, wt_off, t_off, total_floats; };
inline FoldLayout fold_layout(const Plan& pl, int Np, int n, int groups) {
  FoldLayout f;
  f.chunks = (n / groups + BORDER_IMGS - 1) / BORDER_IMGS;
  f.bpart_off = pl.ws_floats;
  f.v_off = f.bpart_off + (size_t)groups * f.chunks * BORDER_CATS * Np;
  f.slot_off = f.v_off + (size_t)groups * 9 * pl.cot * DMA_CO_T;
  f.wt_off = (f.slot_off + (size_t)pl.KS * pl.maxseg + 3) / 4 * 4;
  f.t_off = f.wt_off + (size_t)9 * pl.cot * DMA_CO_T * pl.cit * DMA_CI_T;                       // doubles from here: keep 8-byte alignment
  f.t_off = (f.t_off + 1) / 2 * 2;
  f.total_floats = f.t_off + 2 * ((size_t)pl.KS * pl.maxseg + groups) * S2_ROWPARTS * pl.cit * DMA_CI_T;
  return f;
}

}  // namespace

extern "C" {

size_t sf_conv3x3_bwd_weight_folded_workspace_bytes(int32_t Np, int32_t Kp, int32_t n, int32_t h, int32_t w, int32_t groups) {
  if (groups < 1 || n % groups) return 0;
  return fold_layout(sf_wgrad_bf16_dma_plan(Np, Kp, n, h, w, groups, 1), Np, n, groups).total_floats * sizeof(float);
}

static int bwd_weight_folded_impl(sfTensor src, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap, const int32_t* kmap,
                                  int32_t O, int32_t I, const float* scale, const float* shift, int32_t groups, float* dw, float* db,
                                  int32_t accumulate, const float* weight, const float* mean, const float* rstd, double* bn_sums,
                                  void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream, int sparse24, sfTensor pooled = sfTensor{},
                                  const void* route = nullptr, int32_t perm_l = 0, int32_t perm_t = 0);

int sf_conv3x3_bwd_weight_folded(sfTensor src, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap, const int32_t* kmap,
                                 int32_t O, int32_t I, const float* scale, const float* shift, int32_t groups, float* dw, float* db,
                                 int32_t accumulate, const float* weight, const float* mean, const float* rstd, double* bn_sums,
                                 void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream) {
  return bwd_weight_folded_impl(src, dout, n, h, w, nmap, kmap, O, I, scale, shift, groups, dw, db, accumulate, weight, mean, rstd, bn_sums, workspace,
                                workspace_bytes, dtype, stream, 0);
}

int32_t sf_conv3x3_bwd_weight_folded_sparse24_supported(int32_t Np, int32_t Kp, int32_t n, int32_t h, int32_t w, int32_t groups) {
  static const bool off = getenv("SF_NO_WGRAD_SPARSE") != nullptr;   // A/B switch
  if (off || groups < 1 || n <= 0 || n % groups || h < 2 || w < 2 || (w & 1) || (h & 1)) return 0;
  const Plan pl = sf_wgrad_bf16_dma_plan(Np, Kp, n, h, w, groups, 1);
  return pl.edge_mode == 0 ? 1 : 0;
}

int sf_conv3x3_bwd_weight_folded_sparse24(sfTensor src, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap, const int32_t* kmap,
                                          int32_t O, int32_t I, const float* scale, const float* shift, int32_t groups, float* dw, float* db,
                                          int32_t accumulate, const float* weight, const float* mean, const float* rstd, double* bn_sums,
                                          sfTensor pooled_dout, const void* route, int32_t perm_l, int32_t perm_t,
                                          void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream) {
  SF_REQUIRE(sf_conv3x3_bwd_weight_folded_sparse24_supported(dout.c, src.c, n, h, w, groups),
             "sf_conv3x3_bwd_weight_folded_sparse24: shape not taken by the 2:4-sparse path (ask sf_conv3x3_bwd_weight_folded_sparse24_supported)");
  static const bool no_pooled = getenv("SF_NO_WGRAD_POOLED") != nullptr;   // A/B switch: build the sparse operand from dout even when the pooled form is given
  const bool pooled = pooled_dout.ptr && !no_pooled && h % 4 == 0 && w % 16 == 0 && dout.c % 128 == 0;
  return bwd_weight_folded_impl(src, dout, n, h, w, nmap, kmap, O, I, scale, shift, groups, dw, db, accumulate, weight, mean, rstd, bn_sums, workspace,
                                workspace_bytes, dtype, stream, pooled ? 2 : 1, pooled_dout, route, perm_l, perm_t);
}

static int bwd_weight_folded_impl(sfTensor src, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap, const int32_t* kmap,
                                  int32_t O, int32_t I, const float* scale, const float* shift, int32_t groups, float* dw, float* db,
                                  int32_t accumulate, const float* weight, const float* mean, const float* rstd, double* bn_sums,
                                  void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream, int sparse24, sfTensor pooled,
                                  const void* route, int32_t perm_l, int32_t perm_t) {
  SF_REQUIRE(dtype == SF_BF16 && src.ptr && dout.ptr && src.dtype == SF_BF16 && dout.dtype == SF_BF16,
             "sf_conv3x3_bwd_weight_folded: bf16-stored tensors and the SF_BF16 kernels only");
  SF_REQUIRE(groups >= 1 && n % groups == 0 && h >= 2 && w >= 2, "bwd_weight_folded: n=%d must split into %d groups of whole images, h, w >= 2", n, groups);
  SF_REQUIRE(src.c % SF_CPAD == 0 && dout.c % 8 == 0 && dout.c <= 2048 && src.idiv <= 1 && src.imod <= 0, "bwd_weight_folded: channel padding / no image remap");
  SF_REQUIRE(scale && shift && dw, "bwd_weight_folded: null argument");
  SF_REQUIRE(!bn_sums || (weight && mean && rstd && ((uintptr_t)workspace & 7) == 0), "bwd_weight_folded: bn_sums needs weight, mean, rstd (and an 8-byte aligned workspace)");
  const int Np = dout.c, Kp = src.c;
  // the pooled sparse operand takes K tiles of 8 rows (two tiles per DMA stage: wgrad_pooled8_body) when the image height allows and the workspace the
  // caller sized for the 4-row plan holds that plan's slabs too (SF_WGRAD_TR8=0: A/B switch back to 4-row tiles)
  static const bool no_tr8 = getenv("SF_WGRAD_TR8") != nullptr && getenv("SF_WGRAD_TR8")[0] == '0';
  Plan pl = sf_wgrad_bf16_dma_plan(Np, Kp, n, h, w, groups, 1);
  if (sparse24 == 2 && !no_tr8 && h % 8 == 0) {
    const Plan p8 = sf_wgrad_bf16_dma_plan(Np, Kp, n, h, w, groups, 1, 8);
    if (p8.edge_mode == 0 && fold_layout(p8, Np, n, groups).total_floats * sizeof(float) <= workspace_bytes) pl = p8;
  }
  const FoldLayout fl = fold_layout(pl, Np, n, groups);
  SF_REQUIRE(workspace && workspace_bytes >= fl.total_floats * sizeof(float), "bwd_weight_folded: workspace too small (%zu < %zu)", workspace_bytes,
             fl.total_floats * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  WgradParams p{};
  p.bf = 1; p.bf_dout = 1;
  p.src0 = (const float*)src.ptr; p.c0 = src.c; p.s0 = src.stride;
  p.idiv0 = p.idiv1 = 1;
  p.dout = (const float*)dout.ptr; p.dc = dout.c; p.ds = dout.stride;
  p.N = n; p.H = h; p.W = w;
  p.sparse24 = sparse24;
  if (sparse24 == 2) {
    SF_REQUIRE(pooled.ptr && route && pooled.dtype == SF_BF16 && pooled.c == dout.c && pooled.idiv <= 1 && pooled.imod <= 0,
               "bwd_weight_folded_sparse24: the pooled gradient must be a bf16 tensor of dout's lanes, with the routing record");
    SF_REQUIRE((perm_l == 0 && perm_t == 0) || (perm_l > 0 && perm_t > 0 && n % (perm_l * perm_t) == 0),
               "bwd_weight_folded_sparse24: n=%d not divisible by the pooling's permutation %d x %d", n, perm_l, perm_t);
    p.pool_g = pooled.ptr; p.pool_s = pooled.stride; p.pool_route = (const unsigned short*)route;
    p.pool_L = perm_l; p.pool_T = perm_t; p.pool_B = perm_l > 0 ? n / (perm_l * perm_t) : 0;
  }
  if (int rc = sf_launch_wgrad_bf16_dma(p, pl, (float*)workspace, st)) return rc;
  float* bpart = (float*)workspace + fl.bpart_off;
  float* V = (float*)workspace + fl.v_off;
  {
    const int oct = Np / 8, plx = 256 / oct > 0 ? 256 / oct : 1;
    SF_REQUIRE(oct <= 256, "bwd_weight_folded: Np=%d too wide", Np);
    const size_t shmem = sizeof(float) * plx * BORDER_CATS * Np;
    hipLaunchKernelGGL(fold_border_sums_kernel, dim3(fl.chunks, groups), dim3(256), shmem, st, (const __bf16*)dout.ptr, dout.stride, Np, h, w, n / groups,
                       fl.chunks, bpart);
    SF_CHECK_LAUNCH("fold_border_sums");
  }
  const int per_slice = (pl.ntiles + pl.KS - 1) / pl.KS;
  int* slot_group = (int*)((float*)workspace + fl.slot_off);
  hipLaunchKernelGGL(fold_v_kernel, dim3(p.NpT / 32, groups), dim3(256), 0, st, p.partial_db, pl.KS, pl.maxseg, per_slice, pl.ntiles, pl.tpg,
                     p.NpT, bpart, fl.chunks, Np, groups, V, slot_group);
  SF_CHECK_LAUNCH("fold_v");
  hipLaunchKernelGGL(wgrad_reduce_folded_kernel, dim3(9 * p.NpT), dim3(256), 0, st, p.partial, p.partial_db,
                     pl.KS * pl.maxseg, slot_group, groups, p.NpT, p.KpT, Np, Kp, nmap, kmap, I, scale, shift, V, dw, db, accumulate);
  SF_CHECK_LAUNCH("wgrad_reduce_folded");
  if (bn_sums) {
    float* wt = (float*)workspace + fl.wt_off;
    double* T = (double*)((float*)workspace + fl.t_off);
    const int slots = pl.KS * pl.maxseg;
    const size_t slab = (size_t)9 * p.NpT * p.KpT;
    hipLaunchKernelGGL(fold_wt_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, weight, I, nmap, kmap, Np, Kp, p.NpT, p.KpT, wt);
    SF_CHECK_LAUNCH("fold_wt");
    hipLaunchKernelGGL(fold_bn_s2_kernel, dim3(slots + groups, p.KpT / 64, S2_ROWPARTS), dim3(256), 0, st, p.partial, slot_group, slots, wt, V, p.NpT,
                       p.KpT, T);
    SF_CHECK_LAUNCH("fold_bn_s2");
    hipLaunchKernelGGL(fold_bn_sums_kernel, dim3((groups * Kp + 255) / 256), dim3(256), 0, st, T, slot_group, slots, groups, p.KpT, Kp, mean, rstd, bn_sums);
    SF_CHECK_LAUNCH("fold_bn_sums");
  }
  (void)O;
  return 0;
}

size_t sf_conv3x3_bwd_weight_workspace_bytes(int32_t Np, int32_t Kp, int32_t n, int32_t h, int32_t w) {
  // the loader-wave bf16 variant uses taller K tiles, i.e. never more tiles / a larger KS than this plan; the all-bf16-storage
  // variant has its own plan
  const size_t a = make_plan(Np, Kp, n, h, w, KT_H).ws_floats, b = sf_wgrad_bf16_dma_plan(Np, Kp, n, h, w).ws_floats,
               c = sf_wgrad_bf16_dma_plan(Np, Kp, n, h, w, 0, 1).ws_floats;  // (the edge-slab plans may cut more K slices; mode 2's are a subset of mode 1's)
  return (a > b ? (a > c ? a : c) : (b > c ? b : c)) * sizeof(float);
}

int sf_conv3x3_bwd_weight(sfTensor src0, sfTensor src1, sfTensor dout, int32_t n, int32_t h, int32_t w,
                          const int32_t* nmap, const int32_t* kmap, int32_t O, int32_t I, float* dw, float* db,
                          int32_t accumulate, void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32 || dtype == SF_BF16 || dtype == SF_F16 || dtype == SF_F32E, "sf_conv3x3_bwd_weight: dtype %d not built", dtype);
  SF_REQUIRE(src0.c % SF_CPAD == 0 && src1.c % SF_CPAD == 0 && dout.c % 4 == 0, "bwd_weight: channel padding");
  SF_REQUIRE(src0.ptr || src0.c == 0, "bwd_weight: src0 null");
  SF_REQUIRE(src1.ptr || src1.c == 0, "bwd_weight: src1 null with c=%d (pass c=0)", src1.c);
  const int Np = dout.c, Kp = src0.c + src1.c;
  const bool all_bf16 = dtype == SF_BF16 && dout.dtype == SF_BF16 && (!src0.ptr || src0.dtype == SF_BF16) && (!src1.ptr || src1.dtype == SF_BF16) &&
                        (src0.ptr || src1.ptr);
  // 1: one input tensor; 2: two, the first a whole number of 64-channel tiles wide (every tile of the K space has one source)
  const int one_src = (src0.ptr && src0.c > 0) != (src1.ptr && src1.c > 0) ? 1 : (src0.c % DMA_CI_T == 0 ? 2 : 0);
  const Plan pl = all_bf16 ? sf_wgrad_bf16_dma_plan(Np, Kp, n, h, w, 0, one_src) : make_plan(Np, Kp, n, h, w, (dtype == SF_BF16 || dtype == SF_F16 || dtype == SF_F32E) ? 8 : KT_H);
  SF_REQUIRE(workspace && workspace_bytes >= pl.ws_floats * sizeof(float), "bwd_weight: workspace too small (%zu < %zu)",
             workspace_bytes, pl.ws_floats * sizeof(float));
  WgradParams p{};
  {
    const int nsrc = (src0.ptr != nullptr) + (src1.ptr != nullptr);
    const int nbf = (src0.ptr && src0.dtype == SF_BF16) + (src1.ptr && src1.dtype == SF_BF16);
    SF_REQUIRE(nbf == 0 || nbf == nsrc, "bwd_weight: src0 and src1 must share one storage type");
    p.bf = nbf != 0; p.bf_dout = dout.dtype == SF_BF16;
    SF_REQUIRE(!(p.bf || p.bf_dout) || dtype == SF_BF16, "bwd_weight: bf16-stored tensors need the SF_BF16 kernel");
    SF_REQUIRE(!p.bf || p.bf_dout, "bwd_weight: bf16-stored inputs need a bf16-stored output gradient");
  }
  p.src0 = (const float*)src0.ptr; p.src1 = (const float*)src1.ptr;
  p.c0 = src0.c; p.c1 = src1.c; p.s0 = src0.stride; p.s1 = src1.stride;
  p.idiv0 = src0.idiv > 1 ? src0.idiv : 1; p.imod0 = src0.imod > 0 ? src0.imod : 0;
  p.idiv1 = src1.idiv > 1 ? src1.idiv : 1; p.imod1 = src1.imod > 0 ? src1.imod : 0;
  p.dout = (const float*)dout.ptr; p.dc = dout.c; p.ds = dout.stride;
  p.N = n; p.H = h; p.W = w; p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.KS = pl.KS;
  p.NpT = pl.cot * CO_T; p.KpT = pl.cit * CI_T;
  p.partial = (float*)workspace;
  p.partial_db = p.partial + (size_t)pl.KS * 9 * p.NpT * p.KpT;
  if (all_bf16) {
    if (int rc = sf_launch_wgrad_bf16_dma(p, pl, (float*)workspace, (hipStream_t)stream)) return rc;
  } else if (dtype == SF_BF16) {
    if (int rc = sf_launch_wgrad_bf16(p, pl, (hipStream_t)stream)) return rc;
  } else if (dtype == SF_F16) {
    if (int rc = sf_launch_wgrad_f16(p, pl, (hipStream_t)stream)) return rc;
  } else if (dtype == SF_F32E) {   // three fp16 products per fp32 product; dout (a gradient) scaled through its amax word, if it carries one
    p.amax_dout = dout.amax;
    if (int rc = sf_launch_wgrad_f32e(p, pl, (hipStream_t)stream)) return rc;
  } else {
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3(pl.KS, pl.cot, pl.cit), dim3(256), 0, (hipStream_t)stream, p);
    SF_CHECK_LAUNCH("wgrad_f32");
  }
  const size_t slab = (size_t)9 * p.NpT * p.KpT;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     p.partial, p.partial_db, pl.KS, p.NpT, p.KpT, Np, Kp, nmap, kmap, I, dw, db, accumulate);
  SF_CHECK_LAUNCH("wgrad_reduce");
  (void)O;
  return 0;
}

// Weight gradient of sf_conv5x5_fwd: dW3 [O][4 * lanes][3][3] of the 3x3 convolution over the four shifted views of x (sf_regroup5x5_bwd takes it to the 5x5
// weight), x read in place.  16-bit operand kernels, fp32-stored x / dout, x.c a multiple of 32.
size_t sf_conv5x5_bwd_weight_workspace_bytes(int32_t Np, int32_t xc, int32_t n, int32_t h, int32_t w) { return make_plan(Np, 4 * xc, n, h, w, 8).ws_floats * sizeof(float); }

int sf_conv5x5_bwd_weight(sfTensor x, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap, const int32_t* kmap, int32_t O, int32_t I, float* dw,
                          float* db, int32_t accumulate, void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16 || dtype == SF_F16, "sf_conv5x5_bwd_weight: the 16-bit operand kernels only (dtype %d)", dtype);
  SF_REQUIRE(x.ptr && x.dtype == SF_F32 && x.idiv <= 1 && x.imod <= 0 && x.c >= CI_T && x.c % CI_T == 0 && dout.ptr && dout.dtype == SF_F32 && dout.c % 4 == 0,
             "sf_conv5x5_bwd_weight: fp32-stored x (lanes a multiple of %d: %d) and dout", CI_T, x.c);
  const int Np = dout.c, Kp = 4 * x.c;
  const Plan pl = make_plan(Np, Kp, n, h, w, 8);
  SF_REQUIRE(workspace && workspace_bytes >= pl.ws_floats * sizeof(float), "sf_conv5x5_bwd_weight: workspace too small (%zu < %zu)", workspace_bytes, pl.ws_floats * sizeof(float));
  WgradParams p{};
  p.src0 = (const float*)x.ptr; p.c0 = Kp; p.s0 = x.stride; p.idiv0 = 1; p.idiv1 = 1;
  p.shift4 = x.c;
  p.dout = (const float*)dout.ptr; p.dc = dout.c; p.ds = dout.stride;
  p.N = n; p.H = h; p.W = w; p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.KS = pl.KS;
  p.NpT = pl.cot * CO_T; p.KpT = pl.cit * CI_T;
  p.partial = (float*)workspace;
  p.partial_db = p.partial + (size_t)pl.KS * 9 * p.NpT * p.KpT;
  if (int rc = (dtype == SF_BF16 ? sf_launch_wgrad_bf16 : sf_launch_wgrad_f16)(p, pl, (hipStream_t)stream)) return rc;
  const size_t slab = (size_t)9 * p.NpT * p.KpT;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p.partial, p.partial_db, pl.KS, p.NpT, p.KpT, Np, Kp,
                     nmap, kmap, I, dw, db, accumulate);
  SF_CHECK_LAUNCH("conv5x5 wgrad_reduce");
  (void)O;
  return 0;
}

}  // extern "C"
