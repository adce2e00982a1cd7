// 3x3 "same" convolution as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32),
// NHWC, with fused epilogues (bias / sigmoid / ConvLSTM gates + state update).
//
// GEMM view: M = output pixels, N = output channels, K = 9 taps x input channels.
// One workgroup (256 threads = 4 waves) owns a 16x16 pixel tile of one image and 32*NF output
// channels.  Wave w owns pixel rows 4w..4w+3 as two 32-row M fragments (2 image rows x 16 px
// each) and all NF N fragments: 2*NF accumulator tiles of 32x32 (16 VGPRs each).
//
// K loop: input channels in chunks of 16.  Per chunk the 18x18 halo tile is staged once in LDS
// (80-byte pixel pitch: 16 floats + 4 pad, so 16 consecutive pixels hit 16 distinct 16-byte
// bank slots for ds_read_b128) and reused by all 9 taps; the tap's 32*NF x 16 weight slab is
// double-buffered in LDS from the pre-packed image (sf_conv3x3_pack_weights), the next slab's
// global loads being issued before the current tap's 16*NF MFMAs (64 cycles each).
//
// Exact fp32: the MFMA is a k-ordered fmaf chain (no reduced precision anywhere), which is what
// lets this path meet the reference's rtol 1e-4 / atol 1e-5 through an 18-step recurrence.
#include "conv_common.h"

namespace {

using namespace sfconv;

constexpr int TILE = 16;          // output tile edge (pixels)
constexpr int HALO = TILE + 2;    // staged tile edge
constexpr int PITCH = KC + 4;     // LDS floats per pixel / per weight row (80 B)

template <int NF, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_kernel(const ConvParams p) {
  constexpr int NB = 32 * NF;
  __shared__ __attribute__((aligned(16))) float lds[HALO * HALO * PITCH + 2 * NB * PITCH];
  float* lds_in = lds;
  float* lds_w = lds + HALO * HALO * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, kh = lane >> 5;
  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (private L2 each).  The N blocks of one
  // pixel tile re-read the same input halo tile, so they are given consecutive slots on ONE XCD: the second read
  // is an L2 hit instead of a second trip to HBM.  Pure speed choice; falls back when the tile count is ragged.
  int tile, nb;
  if (gridDim.x % 8 == 0 && gridDim.y > 1) {
    const int id = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = id & 7, j = id >> 3;
    tile = (j / (int)gridDim.y) * 8 + xcd;
    nb = j % (int)gridDim.y;
  } else { tile = blockIdx.x; nb = blockIdx.y; }
  const int tx = tile % p.tiles_x; tile /= p.tiles_x;
  const int ty = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int x0 = tx * TILE, y0 = ty * TILE;

  f32x16 acc[2][NF];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

  // chunks actually present (a NULL source contributes zeros: skip its chunks)
  const int ch0 = p.src0 ? p.c0 / KC : 0;
  const int ch1 = p.src1 ? p.c1 / KC : 0;
  const int c0_chunks = p.c0 / KC;
  const int nsteps = (ch0 + ch1) * 9;

  // weight slab prefetch registers: NB rows x 4 pieces of 16 B = NB*4 pieces over 256 threads
  constexpr int WPIECES = (NB * 4 + 255) / 256;
  f32x4 wreg[WPIECES];

  auto chunk_of = [&](int s) {  // step -> packed chunk index
    const int ci = s / 9;
    return ci < ch0 ? ci : c0_chunks + (ci - ch0);
  };
  auto load_w = [&](int s) {
    const int tap = s % 9;
    const float* wsrc = (const float*)p.wp + (((size_t)nb * p.chunks_total + chunk_of(s)) * 9 + tap) * (NB * KC);
#pragma unroll
    for (int j = 0; j < WPIECES; ++j) {
      const int pc = tid + j * 256;
      if (pc < NB * 4) wreg[j] = *reinterpret_cast<const f32x4*>(wsrc + pc * 4);
    }
  };
  auto store_w = [&](int buf) {
    float* wdst = lds_w + buf * NB * PITCH;
#pragma unroll
    for (int j = 0; j < WPIECES; ++j) {
      const int pc = tid + j * 256;
      if (pc < NB * 4) *reinterpret_cast<f32x4*>(wdst + (pc >> 2) * PITCH + (pc & 3) * 4) = wreg[j];
    }
  };

  if (nsteps > 0) load_w(0);

  for (int s = 0; s < nsteps; ++s) {
    const int tap = s % 9;
    if (tap == 0) {
      // ---- stage the halo tile of this channel chunk ----
      const int ci = s / 9;
      const float* src; int cbase, stride;
      int ns;
      if (ci < ch0) { src = p.src0; cbase = ci * KC; stride = p.s0; ns = n / p.idiv0; if (p.imod0) ns %= p.imod0; }
      else          { src = p.src1; cbase = (ci - ch0) * KC; stride = p.s1; ns = n / p.idiv1; if (p.imod1) ns %= p.imod1; }
      __syncthreads();  // everyone is done reading the previous chunk's tile / weight buffers
      for (int pc = tid; pc < HALO * HALO * 4; pc += 256) {
        const int pix = pc >> 2, piece = pc & 3;
        const int iy = pix / HALO, ix = pix - iy * HALO;
        const int gy = y0 + iy - 1, gx = x0 + ix - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
          v = *reinterpret_cast<const f32x4*>(src + ((size_t)(ns * p.H + gy) * p.W + gx) * stride + cbase + piece * 4);
        *reinterpret_cast<f32x4*>(lds_in + pix * PITCH + piece * 4) = v;
      }
    }
    store_w(s & 1);
    __syncthreads();
    if (s + 1 < nsteps) load_w(s + 1);

    const int ky = tap / 3, kx = tap - ky * 3;
    const float* wbuf = lds_w + (s & 1) * NB * PITCH;
    const float* abase = lds_in + ((4 * wave + (r >> 4) + ky) * HALO + (r & 15) + kx) * PITCH + kh * 4;
    const float* bbase = wbuf + r * PITCH + kh * 4;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 a[2], b[NF];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) a[mf] = *reinterpret_cast<const f32x4*>(abase + mf * 2 * HALO * PITCH + q * 8);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) b[nf] = *reinterpret_cast<const f32x4*>(bbase + nf * 32 * PITCH + q * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mf][j], b[nf][j], acc[mf][nf], 0, 0, 0);
    }
  }

  conv_epilogue<NF, EPI>(acc, p, n, nb, y0, x0, wave, r, kh);
}

// ---- weight repack ---------------------------------------------------------------------------
__global__ void pack_weights_f32_kernel(const float* __restrict__ w, int O, int I, const int* __restrict__ nmap, int Np,
                                        const int* __restrict__ kmap, int Kp, int NB, int transpose,
                                        float* __restrict__ packed, const float* __restrict__ bias,
                                        float* __restrict__ bias_packed) {
  const size_t total = (size_t)Np * Kp * 9;
  const int chunks = Kp / KC;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    // e -> [nblk][chunk][tap][row][k16]
    size_t t = e;
    const int k16 = t % KC; t /= KC;
    const int row = t % NB; t /= NB;
    const int tap = t % 9; t /= 9;
    const int chunk = t % chunks;
    const int nblk = t / chunks;
    const int nn = nmap[nblk * NB + row];
    const int kk = kmap[chunk * KC + k16];
    float v = 0.f;
    if (nn >= 0 && kk >= 0) v = transpose ? w[((size_t)kk * I + nn) * 9 + (8 - tap)] : w[((size_t)nn * I + kk) * 9 + tap];
    packed[e] = v;
  }
  if (bias_packed && blockIdx.x == 0)
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
      const int nn = nmap[i];
      bias_packed[i] = (bias && nn >= 0) ? bias[nn] : 0.f;
    }
}

// Folded BatchNorm: the constant part of conv(a*x + b) per border class.  An output pixel in row class rc (0 top, 1 interior,
// 2 bottom) and column class cc sees the taps whose input pixel lies inside the image; the shift b reaches it only through those.
// The shift goes through the SAME rounded weights as x does - conv(a*x + b) = conv_{W*a}(x + b/a), What = bf16(W*a) as packed:
//   table[g][rc*3+cc][n] = bias[n] + sum over valid taps, k  What[g][n][k][tap] * (shift[g][k] / scale[g][k])
// With exact weights for the shift the two halves would not cancel: sum_px What*x - W*a*mean*N leaves (What - W*a) * mean(x), an
// offset common to all pixels of a channel (about 2^-9 * mean/std of the output's std: invisible per pixel, but it IS the error of
// the next BatchNorm's running mean).  scale == 0 or tiny (gamma ~ 0, shift / scale not finite): the channel contributes w * shift exactly.
// One workgroup per output lane n: its weight row (I x 9 floats, contiguous in OIHW), the scales and the ratios shift / scale go to
// LDS once; thread (g, tap) forms the per-tap sum over k (fp32, fixed order), thread (g, class) adds the valid taps.
__global__ __launch_bounds__(256) void fold_bias_table_kernel(const float* __restrict__ w, int I, const int* __restrict__ nmap, int Np,
                                                             const int* __restrict__ kmap, int Kp, const float* __restrict__ bias,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, int groups,
                                                             float* __restrict__ table) {
  extern __shared__ float sh[];  // [I * 9] weight row | [groups * Kp] scale | [groups * Kp] shift / scale (or shift) | [groups * 9] tap sums
  float* wrow = sh;
  float* sa = sh + I * 9;
  float* sc = sa + groups * Kp;
  float* tsum = sc + groups * Kp;
  const int n = blockIdx.x, nn = nmap[n];
  if (nn >= 0)
    for (int e = threadIdx.x; e < I * 9; e += 256) wrow[e] = w[(size_t)nn * I * 9 + e];
  for (int e = threadIdx.x; e < groups * Kp; e += 256) {
    const float a = scale[e], b = shift[e];
    // a pruned / decayed BatchNorm channel (|gamma * rstd| tiny but nonzero): b / a would overflow while bf16(w * a) flushes to
    // zero - inf * 0 = NaN in every output pixel of the group.  Such a channel takes the exact w * shift branch like scale == 0.
    const float ratio = b / a;
    const bool tiny = !(fabsf(a) >= 1e-18f) || !(fabsf(ratio) <= 3.0e38f);
    sa[e] = tiny ? 0.f : a; sc[e] = tiny ? b : ratio;
  }
  __syncthreads();
  for (int gt = threadIdx.x; gt < groups * 9; gt += 256) {
    const int g = gt / 9, tap = gt - g * 9;
    float t = 0.f;
    if (nn >= 0)
      for (int k = 0; k < Kp; ++k) {
        const int kk = kmap[k];
        if (kk < 0) continue;
        const float wv = wrow[kk * 9 + tap], a = sa[g * Kp + k];
        t = __builtin_fmaf(a != 0.f ? (float)(__bf16)(wv * a) : wv, sc[g * Kp + k], t);
      }
    tsum[gt] = t;
  }
  __syncthreads();
  const float b0 = (bias && nn >= 0) ? bias[nn] : 0.f;
  for (int gc = threadIdx.x; gc < groups * 9; gc += 256) {
    const int g = gc / 9, cls = gc - g * 9, rc = cls / 3, cc = cls - rc * 3;
    float s = b0;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const bool valid = !(rc == 0 && ky == 0) && !(rc == 2 && ky == 2) && !(cc == 0 && kx == 0) && !(cc == 2 && kx == 2);
        if (valid) s += tsum[g * 9 + ky * 3 + kx];
      }
    table[((size_t)g * 9 + cls) * Np + n] = s;
  }
}

template <int EPI>
int launch_conv(const ConvParams& p, int nf, int nblk, hipStream_t st) {
  dim3 grid(p.tiles_x * p.tiles_y * p.N, nblk), block(256);
  switch (nf) {
    case 1: hipLaunchKernelGGL((conv3x3_f32_kernel<1, EPI>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((conv3x3_f32_kernel<2, EPI>), grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL((conv3x3_f32_kernel<3, EPI>), grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL((conv3x3_f32_kernel<4, EPI>), grid, block, 0, st, p); break;
    case 5: hipLaunchKernelGGL((conv3x3_f32_kernel<5, EPI>), grid, block, 0, st, p); break;
    default: sf_set_error("conv3x3: unsupported nf=%d (1..5)", nf); return 1;
  }
  SF_CHECK_LAUNCH("conv3x3_f32");
  return 0;
}

// split-K (sf_conv3x3_fwd_splitk): out[pix][c] = bias[c] + sum_z part[z][pix][c], channel quads, fp32 or bf16 output
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, long long slab, int splits, int np, long long pixels,
                                                            const float* __restrict__ bias, void* __restrict__ out, int out_c, int out_s, int out_bf) {
  const int q = out_c >> 2;
  const long long total = pixels * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long pix = idx / q;
    const int c = (int)(idx - pix * q) * 4;
    f32x4 a = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* src = part + pix * np + c;
    for (int z = 0; z < splits; ++z) a += *reinterpret_cast<const f32x4*>(src + z * slab);
    if (out_bf) {
      bf16x4 b = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
      *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + pix * out_s + c) = b;
    } else {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + pix * out_s + c) = a;
    }
  }
}

// How many channel slices a plain bf16 launch should be cut into (1 = do not split).  Few small images with many input channels - a recurrent
// cell's state convolution on 2 x 20 x 20 pixels with 2048 -> 1024 channels is 32 workgroups each streaming 4.7 MB of weights through its LDS.
int splitk_plan(int n, int h, int w, int Np, int nf, int Kp, int* split_c) {
  static const bool off = getenv("SF_NO_SPLITK") != nullptr;
  *split_c = 0;
  if (off || nf != 4 || Kp % KC != 0 || h < 1 || w < 1 || n < 1) return 1;
  const int th = h > 16 ? 32 : 16;
  if (h <= 16 && w <= 16 && n >= 512) return 1;   // the two-images-per-workgroup kernel takes these
  const long long wgs = (long long)((w + 15) / 16) * ((h + th - 1) / th) * n * (Np / 128);
  const int chunks = Kp / KC;
  if (wgs > 128 || chunks < 8) return 1;
  int s = (int)(256 / wgs);
  if (s > 8) s = 8;
  if (s > chunks / 4) s = chunks / 4;
  if (s < 2) return 1;
  const int per = (chunks + s - 1) / s;
  *split_c = per * KC;
  return (chunks + per - 1) / per;
}

}  // namespace

extern "C" {

size_t sf_conv3x3_packed_elems(int32_t Np, int32_t Kp) { return (size_t)Np * Kp * 9; }

int sf_conv3x3_pack_weights(const float* w, int32_t O, int32_t I, const int32_t* nmap, int32_t Np, const int32_t* kmap,
                            int32_t Kp, int32_t nf, int32_t transpose, void* packed, const float* bias,
                            float* bias_packed, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32 || dtype == SF_BF16 || dtype == SF_F16 || dtype == SF_F32E, "sf_conv3x3_pack_weights: dtype %d not built", dtype);
  SF_REQUIRE(nf >= 1 && nf <= 5 && Np % (32 * nf) == 0, "pack: Np=%d must be a multiple of 32*nf (nf=%d)", Np, nf);
  SF_REQUIRE(Kp % KC == 0, "pack: Kp=%d must be a multiple of %d", Kp, KC);
  if (dtype == SF_BF16 || dtype == SF_F16 || dtype == SF_F32E) {
    // (SF_F32E: `packed` holds 3 * Kp / 16 virtual chunks - the fp16 parts [lo'(w), hi(w)] of every real chunk, then hi(w) of all - 27 * Np * Kp halves)
    (dtype == SF_BF16 ? sf_pack_weights_bf16 : dtype == SF_F16 ? sf_pack_weights_f16 : sf_pack_weights_f32e)(w, O, I, nmap, Np, kmap, Kp, 32 * nf, transpose, packed, bias, bias_packed, (hipStream_t)stream, nullptr, 1);
    SF_CHECK_LAUNCH("pack_weights (16-bit)");
    return 0;
  }
  const size_t total = (size_t)Np * Kp * 9;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_weights_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, O, I, nmap, Np, kmap,
                     Kp, 32 * nf, transpose, (float*)packed, bias, bias_packed);
  SF_CHECK_LAUNCH("pack_weights");
  return 0;
}

static int conv3x3_fwd_impl(sfTensor src0, sfTensor src1, int32_t n, int32_t h, int32_t w, const void* wpacked,
                            const float* bias_packed, int32_t Np, int32_t nf, int32_t epilogue, sfTensor out, float* stats, int32_t dtype,
                            sfStream stream, int32_t fold_groups = 0) {
  SF_REQUIRE(dtype == SF_F32 || dtype == SF_BF16 || dtype == SF_F16 || dtype == SF_F32E, "sf_conv3x3_fwd: dtype %d not built", dtype);
  if (check_src(src0, "conv3x3 src0") || check_src(src1, "conv3x3 src1")) return 1;
  SF_REQUIRE(nf >= 1 && nf <= 5 && Np % (32 * nf) == 0 && out.c <= Np, "conv3x3: bad Np=%d nf=%d out.c=%d", Np, nf, out.c);
  ConvParams p{};
  p.src0 = (const float*)src0.ptr; p.src1 = (const float*)src1.ptr;
  p.c0 = src0.c; p.c1 = src1.c; p.s0 = src0.stride; p.s1 = src1.stride;
  set_remap(p, src0, src1);
  p.N = n; p.H = h; p.W = w; p.tiles_x = (w + TILE - 1) / TILE; p.tiles_y = (h + TILE - 1) / TILE;
  p.wp = (const float*)wpacked; p.bias = bias_packed; p.chunks_total = (src0.c + src1.c) / KC;
  p.out = (float*)out.ptr; p.out_c = out.c; p.out_s = out.stride;
  const int nblk = Np / (32 * nf);
  p.bf0 = src0.ptr && src0.dtype == SF_BF16; p.bf1 = src1.ptr && src1.dtype == SF_BF16; p.out_bf = out.dtype == SF_BF16;
  SF_REQUIRE(dtype == SF_BF16 || !(p.bf0 || p.bf1 || p.out_bf), "conv3x3: bf16-stored tensors need the SF_BF16 kernel");
  SF_REQUIRE((dtype != SF_F16 && dtype != SF_F32E) || !fold_groups, "conv3x3: the folded-BatchNorm launches are SF_BF16 only");
  SF_REQUIRE(!stats || ((dtype == SF_BF16 || dtype == SF_F32E) && epilogue == SF_EPI_LINEAR), "conv3x3: output statistics need the SF_BF16 / SF_F32E kernels with the linear epilogue");
  p.stats = stats; p.stats_np = Np;
  if (fold_groups) {  // bias_packed is the [groups][9][Np] border-class table, wpacked holds one image per group
    p.bias = nullptr; p.bias_tab = bias_packed; p.np = Np;
    p.wgroup = n / fold_groups; p.wgroup_bytes = (long long)Np * (src0.c + src1.c) * 9 * 2;
  }
  if (dtype == SF_F32E) {   // three fp16 products per fp32 product: the packed image has three virtual chunks per real one; src0 may carry a gradient's amax word
    p.chunks_total *= 3;
    p.amax0 = src0.amax;
    SF_REQUIRE(!src1.amax, "conv3x3 (SF_F32E): only src0 takes an amax word (the output gradient of an input-gradient launch)");
  }
  if (dtype == SF_BF16 || dtype == SF_F16 || dtype == SF_F32E) {
    SF_REQUIRE(epilogue == SF_EPI_LINEAR || epilogue == SF_EPI_SIGMOID, "conv3x3: unknown epilogue %d", epilogue);
    // the SF_BF16 kernel stores 16-byte channel quads (fp32) / octets (bf16) per pixel
    SF_REQUIRE(out.ptr && ((uintptr_t)out.ptr & 15) == 0 && out.stride % (p.out_bf ? 8 : 4) == 0 && out.c % 8 == 0,
               "conv3x3: the SF_BF16 kernel needs a 16-byte aligned output (pointer, stride %d, channels %d)", out.stride, out.c);
    SF_REQUIRE(!bias_packed || ((uintptr_t)bias_packed & 15) == 0, "conv3x3: bias_packed must be 16-byte aligned");
    return (dtype == SF_BF16 ? sf_launch_conv_bf16 : dtype == SF_F16 ? sf_launch_conv_f16 : sf_launch_conv_f32e)(p, nf, nblk, epilogue == SF_EPI_LINEAR ? EPI_LINEAR : EPI_SIGMOID, (hipStream_t)stream);
  }
  if (epilogue == SF_EPI_LINEAR) return launch_conv<EPI_LINEAR>(p, nf, nblk, (hipStream_t)stream);
  if (epilogue == SF_EPI_SIGMOID) return launch_conv<EPI_SIGMOID>(p, nf, nblk, (hipStream_t)stream);
  sf_set_error("conv3x3: unknown epilogue %d", epilogue);
  return 1;
}

int sf_conv3x3_fwd(sfTensor src0, sfTensor src1, int32_t n, int32_t h, int32_t w, const void* wpacked,
                   const float* bias_packed, int32_t Np, int32_t nf, int32_t epilogue, sfTensor out, int32_t dtype,
                   sfStream stream) {
  return conv3x3_fwd_impl(src0, src1, n, h, w, wpacked, bias_packed, Np, nf, epilogue, out, nullptr, dtype, stream);
}

size_t sf_conv3x3_fwd_splitk_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t Np, int32_t nf, int32_t Kp, int32_t dtype) {
  int split_c;
  if (dtype != SF_BF16 && dtype != SF_F16) return 0;
  const int s = splitk_plan(n, h, w, Np, nf, Kp, &split_c);
  return s > 1 ? (size_t)s * n * h * w * Np * sizeof(float) : 0;
}

int sf_conv3x3_fwd_splitk(sfTensor src, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_packed, int32_t Np, int32_t nf,
                          sfTensor out, void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16 || dtype == SF_F16, "sf_conv3x3_fwd_splitk: the 16-bit operand kernels only (dtype %d)", dtype);
  SF_REQUIRE(dtype == SF_BF16 || (src.dtype == SF_F32 && out.dtype == SF_F32), "sf_conv3x3_fwd_splitk: SF_F16 kernels take fp32-stored tensors");
  if (check_src(src, "conv3x3 split-K src")) return 1;
  SF_REQUIRE(src.ptr && src.idiv <= 1 && src.imod <= 0, "sf_conv3x3_fwd_splitk: one plain source (no image remap)");
  SF_REQUIRE(nf == 4 && Np % 128 == 0 && out.c <= Np && out.ptr && ((uintptr_t)out.ptr & 15) == 0 && out.c % 8 == 0 && out.stride % 8 == 0,
             "sf_conv3x3_fwd_splitk: nf=4 launches with 16-byte aligned outputs (Np=%d nf=%d out.c=%d)", Np, nf, out.c);
  int split_c;
  const int splits = splitk_plan(n, h, w, Np, nf, src.c, &split_c);
  SF_REQUIRE(splits > 1, "sf_conv3x3_fwd_splitk: this shape is not split (sf_conv3x3_fwd_splitk_workspace_bytes returned 0)");
  const size_t slab = (size_t)n * h * w * Np;
  SF_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= slab * splits * sizeof(float), "sf_conv3x3_fwd_splitk: workspace too small");
  SF_REQUIRE(!bias_packed || ((uintptr_t)bias_packed & 15) == 0, "conv3x3: bias_packed must be 16-byte aligned");
  ConvParams p{};
  p.src0 = (const float*)src.ptr; p.c0 = src.c; p.s0 = src.stride;
  p.idiv0 = 1; p.idiv1 = 1;
  p.N = n; p.H = h; p.W = w; p.tiles_x = (w + TILE - 1) / TILE; p.tiles_y = (h + TILE - 1) / TILE;
  p.wp = wpacked; p.bias = nullptr; p.chunks_total = src.c / KC;
  p.out = (float*)workspace; p.out_c = Np; p.out_s = Np; p.out_bf = 0;
  p.bf0 = src.dtype == SF_BF16;
  p.stats_np = Np;
  p.split_c = split_c; p.split_out = (long long)slab;
  if (int rc = (dtype == SF_BF16 ? sf_launch_conv_bf16 : sf_launch_conv_f16)(p, nf, Np / 128, EPI_LINEAR, (hipStream_t)stream)) return rc;
  const long long pixels = (long long)n * h * w;
  const long long quads = pixels * (out.c / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, (long long)slab, splits, Np, pixels, bias_packed, out.ptr, out.c, out.stride, (int)(out.dtype == SF_BF16));
  SF_CHECK_LAUNCH("conv3x3 split-K reduce");
  return 0;
}

// A 5x5 'same' convolution as ONE 3x3 convolution over four shifted VIEWS of x (ConvParams::shift4): the weights of sf_regroup5x5_fwd (packed for
// Kp = 4 * x.c), no padded domain, no copies.  16-bit operand kernels, fp32-stored x.  Few small images with many channels are split over the
// virtual channel axis like sf_conv3x3_fwd_splitk (workspace from sf_conv5x5_fwd_workspace_bytes; 0 = not split).
size_t sf_conv5x5_fwd_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t Np, int32_t nf, int32_t xc, int32_t dtype) {
  int split_c;
  if (dtype != SF_BF16 && dtype != SF_F16) return 0;
  const int s = splitk_plan(n, h, w, Np, nf, 4 * xc, &split_c);
  return s > 1 ? (size_t)s * n * h * w * Np * sizeof(float) : 0;
}

int sf_conv5x5_fwd(sfTensor x, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_packed, int32_t Np, int32_t nf, sfTensor out,
                   void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16 || dtype == SF_F16, "sf_conv5x5_fwd: the 16-bit operand kernels only (dtype %d)", dtype);
  if (check_src(x, "conv5x5 x")) return 1;
  SF_REQUIRE(x.ptr && x.dtype == SF_F32 && x.idiv <= 1 && x.imod <= 0 && x.c >= KC, "sf_conv5x5_fwd: one plain fp32-stored source of at least %d lanes", KC);
  SF_REQUIRE(nf >= 1 && nf <= 4 && Np % (32 * nf) == 0 && out.c <= Np, "sf_conv5x5_fwd: bad Np=%d nf=%d out.c=%d (the NF = 5 kernels have no shifted-view loader)", Np, nf, out.c);
  SF_REQUIRE(out.ptr && out.dtype == SF_F32 && ((uintptr_t)out.ptr & 15) == 0 && out.stride % 4 == 0 && out.c % 8 == 0,
             "sf_conv5x5_fwd: a 16-byte aligned fp32-stored output (stride %d, channels %d)", out.stride, out.c);
  SF_REQUIRE(!bias_packed || ((uintptr_t)bias_packed & 15) == 0, "sf_conv5x5_fwd: bias_packed must be 16-byte aligned");
  ConvParams p{};
  p.src0 = (const float*)x.ptr; p.c0 = 4 * x.c; p.s0 = x.stride;
  p.idiv0 = 1; p.idiv1 = 1;
  p.N = n; p.H = h; p.W = w; p.tiles_x = (w + TILE - 1) / TILE; p.tiles_y = (h + TILE - 1) / TILE;
  p.wp = (const float*)wpacked; p.chunks_total = 4 * x.c / KC;
  p.shift4 = x.c / KC; p.chunk0 = 0;
  p.stats_np = Np;
  int split_c = 0;
  const int splits = splitk_plan(n, h, w, Np, nf, 4 * x.c, &split_c);
  const size_t slab = (size_t)n * h * w * Np;
  if (splits > 1 && workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= slab * splits * sizeof(float)) {
    p.bias = nullptr;
    p.out = (float*)workspace; p.out_c = Np; p.out_s = Np;
    p.split_c = split_c; p.split_out = (long long)slab;
    if (int rc = (dtype == SF_BF16 ? sf_launch_conv_bf16 : sf_launch_conv_f16)(p, nf, Np / 128, EPI_LINEAR, (hipStream_t)stream)) return rc;
    const long long pixels = (long long)n * h * w, quads = pixels * (out.c / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, (long long)slab, splits, Np, pixels, bias_packed, out.ptr, out.c, out.stride, 0);
    SF_CHECK_LAUNCH("conv5x5 split-K reduce");
    return 0;
  }
  p.bias = bias_packed;
  p.out = (float*)out.ptr; p.out_c = out.c; p.out_s = out.stride;
  return (dtype == SF_BF16 ? sf_launch_conv_bf16 : sf_launch_conv_f16)(p, nf, Np / (32 * nf), EPI_LINEAR, (hipStream_t)stream);
}

int32_t sf_conv3x3_stats_tiles(int32_t h, int32_t w) { return sf_conv_bf16_tiles(h, w); }

int sf_conv3x3_fwd_stats(sfTensor src0, sfTensor src1, int32_t n, int32_t h, int32_t w, const void* wpacked,
                         const float* bias_packed, int32_t Np, int32_t nf, sfTensor out, float* stats, int32_t dtype,
                         sfStream stream) {
  SF_REQUIRE(stats != nullptr, "sf_conv3x3_fwd_stats: stats must not be null");
  return conv3x3_fwd_impl(src0, src1, n, h, w, wpacked, bias_packed, Np, nf, SF_EPI_LINEAR, out, stats, dtype, stream);
}

int sf_conv3x3_fold_pack(const float* w, int32_t O, int32_t I, const int32_t* nmap, int32_t Np, const int32_t* kmap, int32_t Kp,
                         int32_t nf, const float* bias, const float* scale, const float* shift, int32_t groups, void* packed,
                         float* bias_tab, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16, "sf_conv3x3_fold_pack: dtype %d not built (SF_BF16 kernels only)", dtype);
  SF_REQUIRE(nf >= 1 && nf <= 5 && Np % (32 * nf) == 0 && Kp % KC == 0 && groups >= 1, "fold_pack: Np=%d nf=%d Kp=%d groups=%d", Np, nf, Kp, groups);
  SF_REQUIRE(w && scale && shift && packed && bias_tab && ((uintptr_t)bias_tab & 15) == 0, "fold_pack: null / misaligned argument");
  sf_pack_weights_bf16(w, O, I, nmap, Np, kmap, Kp, 32 * nf, 0, packed, nullptr, nullptr, (hipStream_t)stream, scale, groups);
  SF_CHECK_LAUNCH("pack_weights_bf16 (grouped)");
  const size_t tab_lds = ((size_t)(I + groups) * 9 + 2 * (size_t)groups * Kp) * sizeof(float);
  SF_REQUIRE(tab_lds <= 64 * 1024, "fold_pack: I=%d, Kp=%d, groups=%d exceed the table kernel's LDS (%zu bytes)", I, Kp, groups, tab_lds);
  hipLaunchKernelGGL(fold_bias_table_kernel, dim3(Np), dim3(256), tab_lds, (hipStream_t)stream, w, I, nmap, Np, kmap,
                     Kp, bias, scale, shift, groups, bias_tab);
  SF_CHECK_LAUNCH("fold_bias_table");
  return 0;
}

int sf_conv3x3_fwd_folded(sfTensor src, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_tab, int32_t Np, int32_t nf,
                          int32_t groups, sfTensor out, float* stats, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16, "sf_conv3x3_fwd_folded: dtype %d not built (SF_BF16 kernels only)", dtype);
  SF_REQUIRE(groups >= 1 && n % groups == 0 && h >= 2 && w >= 2, "fwd_folded: n=%d must split into %d groups of whole images, h, w >= 2", n, groups);
  SF_REQUIRE(bias_tab && ((uintptr_t)bias_tab & 15) == 0, "fwd_folded: bias_tab null / not 16-byte aligned");
  // images of at most 16x16 pixels could take the two-images-per-workgroup kernel, which has one weight stream: not with per-group weights
  SF_REQUIRE(h > 16 || stats || nf < 4 || n < 512, "fwd_folded: this shape dispatches to the two-image tile kernel (no grouped weights)");
  sfTensor none{nullptr, 0, 0, 0, 0, 0};
  return conv3x3_fwd_impl(src, none, n, h, w, wpacked, bias_tab, Np, nf, SF_EPI_LINEAR, out, stats, dtype, stream, groups);
}

// conv3x3 (folded BatchNorm in front) + MaxPool2d(2) in one launch: the one-wave-per-SIMD kernel with the pooling in its epilogue
// (conv3x3_bf16_persist4.hip, MODE 3).  The convolution's own output is never written.
static void pool_params(ConvParams& p, sfTensor src, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_tab, int32_t Np, int32_t groups,
                        sfTensor pooled, int32_t perm_l, int32_t perm_t, void* route) {
  p.src0 = (const float*)src.ptr; p.c0 = src.c; p.s0 = src.stride; p.bf0 = src.dtype == SF_BF16;
  p.idiv0 = src.idiv > 0 ? src.idiv : 1; p.imod0 = src.imod; p.idiv1 = 1;
  p.N = n; p.H = h; p.W = w;
  p.wp = wpacked; p.chunks_total = src.c / KC;
  p.out = nullptr; p.out_c = pooled.c; p.out_s = pooled.stride; p.out_bf = 1;   // (out_c / out_s describe the channel lanes; nothing is stored through `out`)
  p.bias_tab = bias_tab; p.np = Np; p.stats_np = Np;
  p.wgroup = n / groups; p.wgroup_bytes = (long long)Np * src.c * 9 * 2;
  p.pool_out = pooled.ptr; p.pool_s = pooled.stride; p.pool_route = (unsigned short*)route;
  p.pool_L = perm_l > 0 ? perm_l : 0; p.pool_T = perm_l > 0 ? perm_t : 0; p.pool_B = perm_l > 0 && perm_t > 0 ? n / (perm_l * perm_t) : 0;
}

int32_t sf_conv3x3_fwd_folded_pool_supported(int32_t n, int32_t h, int32_t w, int32_t Np, int32_t nf, int32_t cin, int32_t cout, int32_t groups) {
  if (n <= 0 || groups < 1 || n % groups || h < 2 || w < 2 || cin % KC || nf != 4 || Np % 128 || cout % 64 || cout > Np) return 0;
  ConvParams p{};
  sfTensor src{(void*)16, cin, cin, 0, 0, SF_BF16}, pooled{(void*)16, cout, cout, 0, 0, SF_BF16};
  pool_params(p, src, n, h, w, (const void*)16, (const float*)16, Np, groups, pooled, 0, 0, (void*)16);
  return sf_conv_bf16_persist_ok(p, EPI_LINEAR, nf) && sf_conv_bf16_persist4_ok(p, nf) ? 1 : 0;
}

int sf_conv3x3_fwd_folded_pool(sfTensor src, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_tab, int32_t Np, int32_t nf,
                               int32_t groups, sfTensor pooled, int32_t perm_l, int32_t perm_t, void* route, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16, "sf_conv3x3_fwd_folded_pool: dtype %d not built (SF_BF16 kernels only)", dtype);
  if (check_src(src, "conv3x3 folded+pool src")) return 1;
  SF_REQUIRE(src.ptr && src.dtype == SF_BF16 && src.idiv <= 1 && src.imod <= 0, "fwd_folded_pool: one bf16-stored source without image remap");
  SF_REQUIRE(pooled.ptr && pooled.dtype == SF_BF16 && ((uintptr_t)pooled.ptr & 15) == 0 && pooled.stride % 8 == 0 && pooled.c <= Np,
             "fwd_folded_pool: bf16-stored pooled tensor with 16-byte aligned pixels, at most Np = %d channels", Np);
  SF_REQUIRE(route && ((uintptr_t)route & 7) == 0, "fwd_folded_pool: route null / not 8-byte aligned");
  SF_REQUIRE(bias_tab && ((uintptr_t)bias_tab & 15) == 0, "fwd_folded_pool: bias_tab null / not 16-byte aligned");
  SF_REQUIRE(perm_l <= 0 || (perm_t > 0 && n % (perm_l * perm_t) == 0), "fwd_folded_pool: n=%d not divisible by perm dims %d x %d", n, perm_l, perm_t);
  SF_REQUIRE(sf_conv3x3_fwd_folded_pool_supported(n, h, w, Np, nf, src.c, pooled.c, groups),
             "fwd_folded_pool: shape not taken by the pooled-epilogue kernel (n=%d h=%d w=%d Np=%d nf=%d cin=%d cout=%d groups=%d): ask "
             "sf_conv3x3_fwd_folded_pool_supported first and run sf_conv3x3_fwd_folded + sf_maxpool2_route_fwd otherwise", n, h, w, Np, nf, src.c, pooled.c, groups);
  ConvParams p{};
  pool_params(p, src, n, h, w, wpacked, bias_tab, Np, groups, pooled, perm_l, perm_t, route);
  return sf_launch_conv_bf16_persist4(p, Np / 128, (hipStream_t)stream);
}

int sf_conv3x3_bwd_data_bn(sfTensor dout, int32_t n, int32_t h, int32_t w, const void* wpacked, int32_t Np, int32_t nf, sfTensor x,
                           const float* coef, int32_t groups, sfTensor dx, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16, "sf_conv3x3_bwd_data_bn: dtype %d not built (SF_BF16 kernels only)", dtype);
  if (check_src(dout, "conv3x3 bwd_data_bn dout")) return 1;
  SF_REQUIRE(groups >= 1 && n % groups == 0, "bwd_data_bn: n=%d must split into %d groups of whole images", n, groups);
  SF_REQUIRE(nf >= 1 && nf <= 5 && Np % (32 * nf) == 0 && dx.c <= Np, "bwd_data_bn: bad Np=%d nf=%d dx.c=%d", Np, nf, dx.c);
  SF_REQUIRE(dout.ptr && dout.dtype == SF_BF16 && x.ptr && x.dtype == SF_BF16 && dx.ptr && dx.dtype == SF_BF16 && x.c == dx.c && x.c % 8 == 0 &&
                 x.stride % 8 == 0 && ((uintptr_t)x.ptr & 15) == 0 && dx.stride % 8 == 0 && ((uintptr_t)dx.ptr & 15) == 0,
             "bwd_data_bn: bf16-stored dout, x and dx with matching channel lanes (multiples of 8) and aligned pixels");
  SF_REQUIRE(coef && ((uintptr_t)coef & 15) == 0, "bwd_data_bn: coef null / not 16-byte aligned");
  // images of at most 16x16 pixels could take the two-images-per-workgroup kernel: one group per tile there is not guaranteed
  SF_REQUIRE(h > 16 || nf < 4 || n < 512, "bwd_data_bn: this shape dispatches to the two-image tile kernel");
  ConvParams p{};
  p.src0 = (const float*)dout.ptr; p.c0 = dout.c; p.s0 = dout.stride; p.bf0 = 1;
  p.idiv0 = p.idiv1 = 1;
  p.N = n; p.H = h; p.W = w;
  p.wp = (const float*)wpacked; p.chunks_total = dout.c / KC;
  p.out = (float*)dx.ptr; p.out_c = dx.c; p.out_s = dx.stride; p.out_bf = 1;
  p.bnb_coef = coef; p.bnb_x = x.ptr; p.bnb_xs = x.stride; p.bnb_group = n / groups; p.bnb_c = x.c;
  return sf_launch_conv_bf16(p, nf, Np / (32 * nf), EPI_LINEAR, (hipStream_t)stream);
}

int sf_convlstm_cell_fwd(sfTensor x, sfTensor h_prev, sfTensor c_prev, int32_t n, int32_t h, int32_t w,
                         const void* wpacked, const float* bias_packed, int32_t hidp, sfTensor h_out, sfTensor c_out,
                         sfTensor gates, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32 || dtype == SF_BF16 || dtype == SF_F32E, "sf_convlstm_cell_fwd: dtype %d not built", dtype);
  if (check_src(x, "convlstm x") || check_src(h_prev, "convlstm h_prev")) return 1;
  SF_F32_ONLY(c_prev, "sf_convlstm_cell_fwd"); SF_F32_ONLY(c_out, "sf_convlstm_cell_fwd");
  if (dtype != SF_BF16) {  // bf16-stored x / hidden states: SF_BF16 kernel only
    SF_F32_ONLY(x, "sf_convlstm_cell_fwd"); SF_F32_ONLY(h_prev, "sf_convlstm_cell_fwd"); SF_F32_ONLY(h_out, "sf_convlstm_cell_fwd");
  }
  if (dtype == SF_F32E) SF_F32_ONLY(gates, "sf_convlstm_cell_fwd (SF_F32E)");
  SF_REQUIRE(h_out.dtype == SF_F32 || h_out.dtype == SF_BF16, "sf_convlstm_cell_fwd: h_out storage type %d", h_out.dtype);
  SF_REQUIRE(!gates.ptr || gates.dtype == SF_F32 || gates.dtype == SF_BF16, "sf_convlstm_cell_fwd: gates storage type %d", gates.dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && h_prev.c == hidp, "convlstm: hidp=%d h_prev.c=%d", hidp, h_prev.c);
  SF_REQUIRE(x.ptr && h_out.ptr && c_out.ptr, "convlstm: x, h_out, c_out must be non-null");
  ConvParams p{};
  p.src0 = (const float*)x.ptr; p.src1 = (const float*)h_prev.ptr;
  p.c0 = x.c; p.c1 = h_prev.c; p.s0 = x.stride; p.s1 = h_prev.stride;
  set_remap(p, x, h_prev);
  p.N = n; p.H = h; p.W = w; p.tiles_x = (w + TILE - 1) / TILE; p.tiles_y = (h + TILE - 1) / TILE;
  p.wp = (const float*)wpacked; p.bias = bias_packed; p.chunks_total = (x.c + h_prev.c) / KC;
  p.c_prev = (const float*)c_prev.ptr; p.cprev_s = c_prev.stride;
  p.c_out = (float*)c_out.ptr; p.cout_s = c_out.stride;
  p.h_out = (float*)h_out.ptr; p.hout_s = h_out.stride;
  p.gates = (float*)gates.ptr; p.gates_s = gates.stride; p.gates_bf = gates.ptr && gates.dtype == SF_BF16;
  p.hout_bf = h_out.dtype == SF_BF16;
  p.bf0 = x.dtype == SF_BF16; p.bf1 = h_prev.ptr && h_prev.dtype == SF_BF16;
  p.hidp = hidp;
  const int nblk = (hidp + 31) / 32;
  if (dtype == SF_BF16 || dtype == SF_F32E) {  // pixel-per-lane epilogue: states and gates move as 16-byte quads / octets
    auto quad = [](const sfTensor& t) { return !t.ptr || (((uintptr_t)t.ptr & 15) == 0 && t.stride % (t.dtype == SF_BF16 ? 8 : 4) == 0); };
    SF_REQUIRE(quad(c_prev) && quad(h_out) && quad(c_out) && quad(gates), "sf_convlstm_cell_fwd: states / gates need 16-byte aligned pixels");
    if (dtype == SF_F32E) { p.chunks_total *= 3; return sf_launch_conv_f32e(p, 4, nblk, EPI_LSTM, (hipStream_t)stream); }
    return sf_launch_conv_bf16(p, 4, nblk, EPI_LSTM, (hipStream_t)stream);
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.N, nblk), block(256);
  hipLaunchKernelGGL((conv3x3_f32_kernel<4, EPI_LSTM>), grid, block, 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("convlstm_cell_fwd");
  return 0;
}

int sf_convgru_step_fwd(sfTensor gx, sfTensor h_prev, int32_t n, int32_t h, int32_t w, const void* wpacked,
                        const float* bias_packed, int32_t hidp, sfTensor h_out, sfTensor gates, int32_t dtype,
                        sfStream stream) {
  SF_REQUIRE(dtype == SF_F32 || dtype == SF_BF16 || dtype == SF_F32E, "sf_convgru_step_fwd: dtype %d not built", dtype);
  if (check_src(h_prev, "convgru h_prev")) return 1;
  SF_F32_ONLY(gx, "sf_convgru_step_fwd"); SF_F32_ONLY(h_prev, "sf_convgru_step_fwd"); SF_F32_ONLY(h_out, "sf_convgru_step_fwd");
  SF_REQUIRE(!gates.ptr || gates.dtype == SF_F32 || (gates.dtype == SF_BF16 && dtype == SF_BF16), "sf_convgru_step_fwd: gates storage type %d", gates.dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && h_prev.c == hidp && gx.c == 3 * hidp, "convgru: hidp=%d h_prev.c=%d gx.c=%d", hidp, h_prev.c, gx.c);
  SF_REQUIRE(gx.ptr && h_out.ptr, "convgru: gx and h_out must be non-null");
  ConvParams p{};
  p.src0 = (const float*)h_prev.ptr; p.c0 = h_prev.c; p.s0 = h_prev.stride;
  p.idiv0 = p.idiv1 = 1;
  p.N = n; p.H = h; p.W = w; p.tiles_x = (w + TILE - 1) / TILE; p.tiles_y = (h + TILE - 1) / TILE;
  p.wp = (const float*)wpacked; p.bias = bias_packed; p.chunks_total = h_prev.c / KC;
  p.h_out = (float*)h_out.ptr; p.hout_s = h_out.stride;
  p.gates = (float*)gates.ptr; p.gates_s = gates.stride; p.gates_bf = gates.ptr && gates.dtype == SF_BF16;
  p.gx = (const float*)gx.ptr; p.gx_s = gx.stride;
  p.h_prev = (const float*)h_prev.ptr; p.hprev_s = h_prev.stride;
  p.hidp = hidp;
  const int nblk = (hidp + 31) / 32;
  if (dtype == SF_BF16) return sf_launch_conv_bf16(p, 3, nblk, EPI_GRU, (hipStream_t)stream);
  if (dtype == SF_F32E) { p.chunks_total *= 3; return sf_launch_conv_f32e(p, 3, nblk, EPI_GRU, (hipStream_t)stream); }
  dim3 grid(p.tiles_x * p.tiles_y * p.N, nblk), block(256);
  hipLaunchKernelGGL((conv3x3_f32_kernel<3, EPI_GRU>), grid, block, 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("convgru_step_fwd");
  return 0;
}

}  // extern "C"
