// 3x3 "same" convolution as an implicit GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16),
// fp32 activations in HBM (NHWC), bf16 operands, fp32 accumulate, fp32 outputs - the arithmetic of
// torch.autocast(bfloat16) around the reference's Conv2d, with the epilogues of conv_common.h.
// Sources / the linear output may also be STORED as bf16 (sfTensor.dtype, the MetNet encoder's "bf16a" mode):
// the staging then copies 16-byte pieces unconverted and the epilogue rounds once (RNE) on the way out.
//
// Workgroup = WAVES waves (8 for large images: a 32x16-pixel tile; 4: 16x16), each wave owns 4 tile
// rows = two 32-pixel M fragments and all NF 32-channel N fragments (2*NF accumulator tiles).
// K loop over 16-channel chunks; per chunk ALL 9 taps are resident in LDS:
//   * weights: the packed bf16 image is already the LDS image (32-byte rows, 16-byte halves swapped
//     on rows with bit 3 set so that a ds_read_b128 lane group hits 16 distinct bank slots); it is
//     streamed by LDS-DMA (global_load_lds_dwordx4, no VGPRs) into the other half of a double buffer
//     while the current chunk computes;
//   * input halo tile: fp32 global loads issued before the chunk's MFMAs, converted to bf16
//     (v_cvt_pk_bf16_f32, RNE) and written after them into the other input buffer (halves swapped on
//     odd halo rows - conflict-free for the 2-rows-per-fragment access pattern).
// One barrier per chunk; 18*NF MFMAs (32 cycles each) per wave between barriers.
#include <cstdlib>

#include "conv_common.h"

// The 16-bit OPERAND type is a compile-time choice of this translation unit: __bf16 here, _Float16 when the file is included by conv3x3_f16.hip
// (SF_OPERAND_F16: the `precision: 16` of the reference's configs/trainer/half.yaml:33 - fp16 MFMA operands, fp32 accumulate, fp32 storage).  The f16
// build renames the three launch entry points (sf_launch_conv_f16 / sf_conv_f16_tiles / sf_pack_weights_f16) and never takes the persistent kernels.
// SF_SPLIT3 (with SF_OPERAND_F16; conv3x3_f32e.hip): the SF_F32E compute mode - fp32-equivalent products from THREE fp16 products.  The K loop runs over 3 x
// the chunks: phase A, per real chunk c, the virtual chunks 2c = hi(x) * lo'(w) and 2c + 1 = lo'(x) * hi(w) (lo' = the residual after the fp16 rounding,
// scaled by 2^11 so that it has the operand's own exponent range); then the accumulators are multiplied by 2^-11 (exact) and phase B adds hi(x) * hi(w).
// The packed weight image holds the virtual chunks in that order (pack kernel below); a real chunk's fp32 halo tile is loaded once for its two phase-A
// chunks.  fp32-stored tensors only; the recurrent epilogues are allowed (the fused ConvLSTM cell is the pinned hot path of the parity mode).
#ifdef SF_SPLIT3
#ifndef SF_OPERAND_F16
#error "SF_SPLIT3 is built on the fp16 operand type"
#endif
#define SF_OP_T _Float16
#define SF_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define sf_launch_conv_bf16 sf_launch_conv_f32e
#define sf_conv_bf16_tiles sf_conv_f32e_tiles
#define sf_pack_weights_bf16 sf_pack_weights_f32e
#elif defined(SF_OPERAND_F16)
#define SF_OP_T _Float16
#define SF_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define sf_launch_conv_bf16 sf_launch_conv_f16
#define sf_conv_bf16_tiles sf_conv_f16_tiles
#define sf_pack_weights_bf16 sf_pack_weights_f16
#else
#define SF_OP_T __bf16
#define SF_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

namespace {

using namespace sfconv;

typedef SF_OP_T bf16x8 __attribute__((ext_vector_type(8)));  // eight operands of the translation unit's 16-bit type
typedef float f32x8 __attribute__((ext_vector_type(8)));

#ifdef SF_SPLIT3
constexpr float SPLIT_UP = 2048.f, SPLIT_DOWN = 1.f / 2048.f;   // 2^11: the low part's scale
// part of v (already multiplied by the tensor's power-of-two scale) that a virtual chunk multiplies: hi = fp16(v), or lo' = fp16((v - hi) * 2^11)
__device__ __forceinline__ bf16x8 split_part(f32x8 v, bool lo) {
  const bf16x8 hi = __builtin_convertvector(v, bf16x8);
  if (!lo) return hi;
  return __builtin_convertvector((v - __builtin_convertvector(hi, f32x8)) * SPLIT_UP, bf16x8);
}
// power-of-two scale that puts a tensor's largest magnitude (device word `amax`, sf_amax) at 2^14; 1 without a word or for an all-zero tensor.
// Returns (scale, 1 / scale) - both exact.
__device__ __forceinline__ void split_scale(const float* amax, float& s, float& inv) {
  s = inv = 1.f;
  if (!amax) return;
  const unsigned bits = __builtin_bit_cast(unsigned, *amax);
  const int e = (int)((bits >> 23) & 0xffu) - 127;   // floor(log2 amax) of a normal value
  if ((bits & 0x7fffffffu) == 0u) return;
  int k = 14 - e;
  k = k > 126 ? 126 : (k < -126 ? -126 : k);
  s = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
  inv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
}
#endif

constexpr int HALO_W = TILE_W + 2;  // 18
constexpr int PIX_B = 32;           // bytes per pixel / per weight row in LDS (16 bf16)

// LDS-DMA through a buffer descriptor, issued from inline asm: wave-uniform descriptor + scalar byte offset (chunk / tile
// position) + a per-lane byte offset that is CONSTANT for the whole kernel (an out-of-image piece carries an out-of-range offset:
// the hardware range check then writes zeros to its LDS slot, tools/ubench/buf_lds_oob.hip).  Hidden from hipcc on purpose: it
// would otherwise put `s_waitcnt vmcnt(0)` in front of LDS reads that might alias a pending DMA (seen at mid-chunk in this
// kernel) - the chunk loop waits for its DMA itself, once, at the top of the next chunk.  M0 (the LDS destination) is saved and
// restored: other code of the same kernel may use compiler-issued LDS-DMA.
constexpr unsigned DMA_SENT = 0x80000000u;  // >= any descriptor's num_records (the launcher checks tensor bytes < 2^31)
__device__ __forceinline__ void bufdma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);  // wave-uniform by construction; keeps it in an SGPR where hipcc cannot prove that
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff) : "memory");
}

// DUAL (8 waves, images of at most 16x16 pixels): the 32x16 tile is TWO consecutive images, waves 0-3 on the first and
// 4-7 on the second, each image with its own halo rows in LDS (2 x 18 rows) - small images keep the 8-wave workgroup's
// weight reuse and occupancy instead of dropping to the 4-wave 16x16 kernel.
// WS (4 waves): ONE weight buffer instead of two - 58 KB of LDS, so TWO workgroups share a CU (one wave of each per SIMD).  A
// workgroup then waits for its next chunk's weights after every chunk (the DMA is issued behind an end-of-chunk barrier), but the
// partner workgroup's MFMAs, prologue and epilogue run in those gaps: for the fused LSTM cell, whose epilogue moves 5 state / gate
// tensors per tile, that overlap is worth more than the double buffer (ConvLSTM cell).
// SPLITK: grid z = slice of the input channels (ConvParams::split_c): the same kernel on a shifted source / weight / output pointer.
// SHIFT: src0 is four displaced views of one fp32-stored tensor (ConvParams::shift4 - a 5x5 convolution, sf_conv5x5_fwd): the halo loader applies the view's
// pixel shift, and the taps the view does not own (first row of the lower views, first column of the right ones: sf_regroup5x5_fwd masks those weights) are
// loaded but not multiplied - 25 instead of 36 taps' worth of MFMAs over the four views.  Its own instantiations: the plain kernels keep their code.
template <int WAVES, int NF, int EPI, bool DUAL = false, bool TR = false, bool BNB = false, bool WS = false, bool SPLITK = false, bool SHIFT = false>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 8 || WS) ? 2 : 1) void conv3x3_bf16_kernel(const ConvParams p_in) {
  static_assert(!SHIFT || (EPI == EPI_LINEAR && TR && !BNB && !WS && NF != 5), "shifted views: plain linear launches, NF <= 4");
  static_assert(!WS || (WAVES == 4 && !DUAL), "the single-weight-buffer variant is a 4-wave layout");
  static_assert(!SPLITK || (EPI == EPI_LINEAR && !DUAL && TR && !BNB && !WS), "split-K: plain linear launches only");
  ConvParams p_split;
  if constexpr (SPLITK) {
    const int z = blockIdx.z;
    p_split = p_in;
    const int cbeg = z * p_in.split_c;
    if constexpr (SHIFT) p_split.chunk0 = cbeg / KC;   // shifted views: the slice starts at a virtual chunk, the tensor pointer stays
    else p_split.src0 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p_in.src0) + (size_t)cbeg * (p_in.bf0 ? 2 : 4));
    p_split.c0 = p_in.c0 - cbeg < p_in.split_c ? p_in.c0 - cbeg : p_in.split_c;
    p_split.wp = reinterpret_cast<const char*>(p_in.wp) + (size_t)(cbeg / KC) * (9 * 32 * NF * PIX_B);
    p_split.out = p_in.out + (size_t)z * p_in.split_out;
  }
  const ConvParams& p = SPLITK ? p_split : p_in;
  constexpr int WBUFS = WS ? 1 : 2;
  static_assert(!DUAL || WAVES == 8, "dual-image tiles are an 8-wave layout");
  constexpr int NB = 32 * NF;
  constexpr int THREADS = WAVES * 64;
  constexpr int TH = 4 * WAVES;            // tile rows
  constexpr int HALO_H = DUAL ? 36 : TH + 2;
  constexpr int IN_B = HALO_H * HALO_W * PIX_B;
  constexpr int W_B = 9 * NB * PIX_B;
  constexpr int PIECES = HALO_H * HALO_W * 2;                 // 16-byte bf16 pieces of the halo tile
  constexpr int NPIECE = (PIECES + THREADS - 1) / THREADS;
  __shared__ __attribute__((aligned(1024))) char lds[WBUFS * W_B + 2 * IN_B];
  __shared__ __attribute__((aligned(16))) float lds_coef[BNB ? 3 * NB : 4];  // BatchNorm-backward epilogue: (A, B, K) of this N block
  // NF = 5 (160 accumulators, 256 registers all in use): the per-piece DMA words live in LDS, [piece][thread], and are read back at issue time - an LDS
  // read waits on lgkmcnt; a register that spills waits on vmcnt(0), i.e. on the DMA pieces just issued (see in_pk below)
  constexpr bool PK_LDS = NF == 5 && !DUAL;
  __shared__ unsigned lds_pk[PK_LDS ? 5 * THREADS : 1];
  char* lds_w = lds;
  char* lds_in = lds + WBUFS * W_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int tx = DUAL ? 0 : tile % p.tiles_x; if (!DUAL) tile /= p.tiles_x;
  const int ty = DUAL ? 0 : tile % p.tiles_y;
  const int n = DUAL ? 2 * tile : tile / p.tiles_y;  // (first) image of the tile
  const int x0 = tx * TILE_W, y0 = ty * TH;
  const int img = DUAL ? wave >> 2 : 0, wl = DUAL ? wave & 3 : wave;  // this wave's image within the tile / its 4-row band
  const int n_w = n + img;
  // halo-tile pixel -> source pixel; false outside the image (or past the last image of a dual tile)
  auto halo = [&](int pix, int& ni, int& gy, int& gx, int& iy) -> bool {
    iy = pix / HALO_W;
    const int ix = pix - iy * HALO_W;
    const int sel = DUAL ? (iy >= 18 ? 1 : 0) : 0;
    ni = n + sel; gy = y0 + (iy - 18 * sel) - 1; gx = x0 + ix - 1;
    return ni < p.N && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
  };

  if constexpr (BNB) {  // staged now, read in the epilogue: the K loop's barriers order the two, the epilogue pays nothing for it
    const float* co = p.bnb_coef + (size_t)(n / p.bnb_group) * 3 * p.bnb_c;
    for (int i = tid; i < 3 * NB; i += THREADS) {
      const int c = nb * NB + i % NB;
      lds_coef[i] = c < p.out_c ? co[(size_t)(i / NB) * p.bnb_c + c] : 0.f;
    }
  }

  f32x16 acc[2][NF];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;
  // Fused LSTM cell: the accumulators START at the gate biases (lane = pixel, register 4g + c of fragment q = channel 8g + 4kh + c of gate q): the
  // loads run under the first chunk's staging and the epilogue begins with the cell-state loads instead of sixteen bias loads.
#ifdef SF_SPLIT3
  float sx, sx_inv;                        // the source's power-of-two scale (gradient operands, ConvParams::amax0) and its inverse
  split_scale(p.amax0, sx, sx_inv);
  const float acc0_scale = SPLIT_UP * sx;  // accumulators that START at a bias live through the 2^-11 rescale and the final 1 / sx
#endif
  if constexpr (EPI == EPI_LSTM) {
    static_assert(TR, "the LSTM epilogue is the transposed one");
    if (p.bias) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + nb * NB + 32 * nf + 8 * g + 4 * kh);
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
#pragma unroll
#ifdef SF_SPLIT3
            for (int c = 0; c < 4; ++c) acc[mf][nf][4 * g + c] = b[c] * acc0_scale;
#else
            for (int c = 0; c < 4; ++c) acc[mf][nf][4 * g + c] = b[c];
#endif
        }
    }
  }
  // Folded BatchNorm (sf_conv3x3_fwd_folded): the accumulators START at the group's bias of each pixel's border class (the epilogue
  // then adds no bias) - here, before the K loop, nothing else is live yet.
  if constexpr (EPI == EPI_LINEAR && !DUAL) {
    if (p.bias_tab) {
      const float* tab = p.bias_tab + (size_t)(p.wgroup ? n / p.wgroup : 0) * 9 * p.np + nb * NB;
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
        if constexpr (TR) {  // lane = pixel r of the fragment; register 4g + c = channel 8g + 4kh + c
          const float* t = tab + (size_t)border_cls(y0 + 4 * wave + 2 * mf + (r >> 4), x0 + (r & 15), p.H, p.W) * p.np + 4 * kh;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 b = *reinterpret_cast<const f32x4*>(t + nf * 32 + 8 * g);
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[mf][nf][4 * g + c] = b[c];
            }
        } else {  // lane = channel r; register = pixel frag_row(reg, kh) of the fragment
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int rr = frag_row(reg, kh);
            const float* t = tab + (size_t)border_cls(y0 + 4 * wave + 2 * mf + (rr >> 4), x0 + (rr & 15), p.H, p.W) * p.np + r;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf][reg] = t[nf * 32];
          }
        }
    }
  }

  const int ch0 = p.src0 ? p.c0 / KC : 0;
  const int ch1 = p.src1 ? p.c1 / KC : 0;
  const int c0_chunks = p.c0 / KC;
#ifdef SF_SPLIT3
  const int nreal = ch0 + ch1;          // real 16-channel chunks of the sources
  const int nch = 3 * nreal;            // virtual chunks of the K loop
  const int wreal = p.chunks_total / 3; // real chunks of the packed image (sources that are absent at t = 0 keep their weight chunks)
  auto v_real = [&](int cv) { return cv < 2 * nreal ? cv >> 1 : cv - 2 * nreal; };
  auto v_xlo = [&](int cv) { return cv < 2 * nreal && (cv & 1); };   // the chunk multiplies lo'(x) (with hi(w)); every other chunk hi(x)
#else
  const int nch = ch0 + ch1;
#endif

  f32x8 inreg[NPIECE];
  int staged_bf = 0;  // storage type of the chunk held in inreg (block-uniform)

  // ---- bf16-stored sources, single-image tiles: everything about a halo piece except the chunk's channel offset is a block
  // constant - source image (the remap division happens ONCE here, not per piece and chunk), per-lane byte offset and validity.
  // Descriptor = the source image, started one image row + one pixel early so that the halo origin has a non-negative offset.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000), rs_in1 = rs_in0, rs_w = rs_in0;
  // ONE packed word per piece - halo pixel offset iy * W + ix in bits 0..29, the logical 16-byte half in bit 30, all ones = outside the image - from which the
  // byte offset of either source is two VALU operations at issue time.  (Rounds 1-4 kept a ready offset per SOURCE: ten registers in the NF = 5 kernel, which
  // spilled exactly those - and a scratch reload between two DMA pieces is a vector-memory operation whose `s_waitcnt vmcnt(0)` waits for the pieces just
  // issued to LAND: the staging block of every chunk stalled four times for a full memory latency.  Found in round 5 through the same effect in
  // conv3x3_wgrad_bf16_dma.hip.)
  unsigned in_pk[NPIECE];
  unsigned so_in0 = 0, so_in1 = 0;
  if constexpr (!DUAL) {
    auto desc = [&](const float* src, int stride, int idiv, int imod, unsigned& so) {
      int ns = n / idiv; if (imod) ns %= imod;
      const long long pxb = 2ll * stride, lead = (long long)(p.W + 1) * pxb, img = (long long)p.H * p.W * pxb;
      so = (unsigned)((y0 * p.W + x0) * (int)pxb);
      return __builtin_amdgcn_make_buffer_rsrc((void*)(src ? (const char*)src + ns * img - lead : nullptr), 0, src ? (int)(img + 2 * lead) : 0, 0x00020000);
    };
    rs_in0 = desc(p.bf0 ? p.src0 : nullptr, p.s0, p.idiv0, p.imod0, so_in0);
    rs_in1 = desc(p.bf1 ? p.src1 : nullptr, p.s1, p.idiv1, p.imod1, so_in1);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int pc = tid + j * THREADS, pix = pc >> 1;
      const int iy = pix / HALO_W, ix = pix - iy * HALO_W;
      const int gy = y0 + iy - 1, gx = x0 + ix - 1;
      const bool ok = pc < PIECES && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int half = (pc & 1) ^ (iy & 1);  // the DMA writes lane-linearly: physical half pc & 1 fetches the logical half (bank swizzle)
      in_pk[j] = ok ? (unsigned)(iy * p.W + ix) | ((unsigned)half << 30) : 0xffffffffu;
      if constexpr (PK_LDS) { static_assert(!PK_LDS || NPIECE <= 5, "lds_pk holds five pieces per thread"); lds_pk[j * THREADS + tid] = in_pk[j]; }
    }
  }
  // weights: descriptor over this N block's packed image; the chunk and the 1 KiB piece go into the scalar offset
  const long long wgrp = (!DUAL && p.wgroup) ? (long long)(n / p.wgroup) * p.wgroup_bytes : 0;  // grouped weights: this image's packed image
  rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wp + wgrp + (size_t)nb * p.chunks_total * W_B), 0, p.chunks_total * W_B, 0x00020000);

  auto issue_weights = [&](int ci, int buf) {
#ifdef SF_SPLIT3
    const int rc = v_real(ci), rchunk = rc < ch0 ? rc : c0_chunks + (rc - ch0);
    const int chunk = ci < 2 * nreal ? 2 * rchunk + (ci & 1) : 2 * wreal + rchunk;   // packed order: [lo'(w) c, hi(w) c] per real chunk, then hi(w) of all
#else
    const int chunk = ci < ch0 ? ci : c0_chunks + (ci - ch0);
#endif
    const unsigned dst = lds0 + buf * W_B;
    for (int i = wave; i < 9 * NF; i += WAVES) bufdma16(lane * 16, rs_w, (unsigned)(chunk * W_B + i * 1024), dst + i * 1024);
  };
  auto load_input = [&](int ci) {  // fp32-stored source: registers now, bf16 conversion + ds_write after the chunk's MFMAs
#ifdef SF_SPLIT3
    if (v_xlo(ci)) return;   // the second phase-A chunk of a real chunk: its fp32 values are the ones already in `inreg`
    ci = v_real(ci);
#endif
    const float* src; int cbase, stride, idiv, imod;
    if (ci < ch0) { src = p.src0; cbase = ci * KC; stride = p.s0; idiv = p.idiv0; imod = p.imod0; }
    else          { src = p.src1; cbase = (ci - ch0) * KC; stride = p.s1; idiv = p.idiv1; imod = p.imod1; }
    int sdy = 0, sdx = 0;
    if constexpr (SHIFT) {  // four shifted views of src0 (ConvParams::shift4): this chunk's view and its pixel shift
      const int cg = ci + p.chunk0, sv = cg / p.shift4;
      cbase = (cg - sv * p.shift4) * KC; sdy = 2 * (sv >> 1) - 1; sdx = 2 * (sv & 1) - 1;
    }
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int pc = tid + j * THREADS;
      int ni, gy, gx, iy;
      bool ok = halo(pc >> 1, ni, gy, gx, iy) && pc < PIECES;
      if constexpr (SHIFT) { gy += sdy; gx += sdx; ok = pc < PIECES && ni < p.N && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W; }
      f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (ok) {
        int ns = ni / idiv; if (imod) ns %= imod;
        v = *reinterpret_cast<const f32x8*>(src + ((size_t)(ns * p.H + gy) * p.W + gx) * stride + cbase + (pc & 1) * 8);
      }
      inreg[j] = v;
    }
  };
  auto store_input = [&](int buf, bool lo_part = false) {
    if (staged_bf) return;  // bf16-stored sources went straight to LDS (dma_input)
    char* dst = lds_in + buf * IN_B;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int pc = tid + j * THREADS;
      const int pix = pc >> 1, half = pc & 1;
      const int iy = pix / HALO_W;
#ifdef SF_SPLIT3
      if (pc < PIECES) *reinterpret_cast<bf16x8*>(dst + pix * PIX_B + 16 * (half ^ (iy & 1))) = split_part(inreg[j] * sx, lo_part);
#else
      if (pc < PIECES) *reinterpret_cast<bf16x8*>(dst + pix * PIX_B + 16 * (half ^ (iy & 1))) = __builtin_convertvector(inreg[j], bf16x8);
#endif
    }
  };
  // bf16-STORED sources need no conversion: their halo tile goes HBM -> LDS by LDS-DMA like the weights (no staging
  // registers, no ds_write, nothing to wait for before the next barrier's vmcnt(0)).  The DMA writes lane-linearly, so
  // piece pc = (pixel, physical half) lands in 16-byte slot pc and fetches the LOGICAL half (physical ^ row parity,
  // the bank swizzle of the register path).  Halo pixels outside the image are never written by the DMA (lanes
  // masked off): their slots are zeroed once per block, below.
  auto piece_off = [](unsigned pk, int stride) __attribute__((always_inline)) {
    return pk == 0xffffffffu ? DMA_SENT : (pk & 0x3fffffffu) * (unsigned)(2 * stride) + ((pk >> 30) << 4);
  };
  auto piece_word = [&](int j) __attribute__((always_inline)) {
    if constexpr (PK_LDS) {   // (thread index rebuilt from the wave number and the exec-mask count: nothing to keep in a register, nothing to spill)
      typedef __attribute__((address_space(3))) const unsigned* lds_u32;
      const unsigned t = (unsigned)wave * 64u + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds_pk;
      return *(lds_u32)(uintptr_t)(base + (unsigned)(j * THREADS) * 4u + t * 4u);
    } else return in_pk[j];
  };
  auto dma_input = [&](int ci, int buf) {
    if constexpr (!DUAL) {
      const unsigned dst = lds0 + (unsigned)(WBUFS * W_B + buf * IN_B + wave * 1024);
      // (a uniform branch per source rather than selects: the descriptor must stay in SGPRs)
      if (ci < ch0) {
        const unsigned so = so_in0 + (unsigned)(ci * KC * 2);
#pragma unroll
        for (int j = 0; j < NPIECE; ++j)
          if (tid + j * THREADS < PIECES) bufdma16(piece_off(piece_word(j), p.s0), rs_in0, so, dst + j * WAVES * 1024);  // lanes past the tile are masked off: a zero-filling lane there would write into the other buffer
      } else {
        const unsigned so = so_in1 + (unsigned)((ci - ch0) * KC * 2);
#pragma unroll
        for (int j = 0; j < NPIECE; ++j)
          if (tid + j * THREADS < PIECES) bufdma16(piece_off(piece_word(j), p.s1), rs_in1, so, dst + j * WAVES * 1024);
      }
      return;
    }
    const float* src; int cbase, stride, idiv, imod;
    if (ci < ch0) { src = p.src0; cbase = ci * KC; stride = p.s0; idiv = p.idiv0; imod = p.imod0; }
    else          { src = p.src1; cbase = (ci - ch0) * KC; stride = p.s1; idiv = p.idiv1; imod = p.imod1; }
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int pc = tid + j * THREADS;  // = (wave + j*WAVES) * 64 + lane: one wave-instruction fills 64 consecutive slots
      int ni, gy, gx, iy;
      if (halo(pc >> 1, ni, gy, gx, iy) && pc < PIECES) {
        int ns = ni / idiv; if (imod) ns %= imod;
        const __bf16* img_p = reinterpret_cast<const __bf16*>(src) + (size_t)ns * p.H * p.W * stride + cbase;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(img_p + (size_t)(gy * p.W + gx) * stride + 8 * ((pc & 1) ^ (iy & 1))),
            (__attribute__((address_space(3))) void*)(lds_in + buf * IN_B + (wave + j * WAVES) * 1024), 16, 0, 0);
      }
    }
  };
  auto stage_input = [&](int ci) {  // issue side of the next chunk's input staging
#ifdef SF_SPLIT3
    (void)dma_input;
    load_input(ci);   // fp32-stored sources only (the launcher refuses bf16 storage)
#else
    staged_bf = ci < ch0 ? p.bf0 : p.bf1;
    if (staged_bf) dma_input(ci, ci & 1);
    else load_input(ci);
#endif
  };
  if (DUAL && (p.bf0 || p.bf1)) {  // compiler-issued DMA path (lanes masked off): zero the out-of-image halo slots of both buffers once
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int pc = tid + j * THREADS;
      int ni, gy, gx, iy;
      if (!halo(pc >> 1, ni, gy, gx, iy) && pc < PIECES) {
        *reinterpret_cast<f32x4*>(lds_in + pc * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds_in + IN_B + pc * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }

  if (nch > 0) {
    issue_weights(0, 0);
    stage_input(0);
    store_input(0);
  }

  // per-lane LDS offsets
  const int rowpar = (r >> 4) & 1;
  const int a_lane = (((DUAL ? 18 * img : 0) + 4 * wl + (r >> 4)) * HALO_W + (r & 15)) * PIX_B;
  const int a_half_even = 16 * (kh ^ rowpar), a_half_odd = 16 * (kh ^ rowpar ^ 1);  // by parity of ky
  const int b_lane = r * PIX_B + 16 * (kh ^ ((r >> 3) & 1));

  for (int ci = 0; ci < nch; ++ci) {
    const int cur = ci & 1;
    // this chunk's weight DMA has landed (LDS-DMA is not covered by the barrier) AND this wave's ds_write of the staged input has left the LDS queue.
    // The second half is not a formality: hipcc 7.2 drops the LDS wait of __syncthreads()'s release fence when it can (it assumes the LDS serves all
    // waves in one order), and in the SF_SPLIT3 build it did so on the path loop tail -> this barrier.  A wave of another SIMD then read a 16-byte piece
    // of the staged tile before the write arrived: one stale piece in ~1000 launches of the f32e ConvGRU step, found as a once-in-25-runs parity failure
    // (round 6; tools/scan_barrier_waits.py + tests/test_host_cpu.py check the ISA of every kernel for the pattern).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // Staging of the NEXT chunk (weight DMA + input DMA / loads) is issued at a different tap by the two waves that
    // share a SIMD (wave w and w + 4 of an 8-wave workgroup): a wave stalls for a few hundred cycles while it issues its
    // ~7 LDS-DMA pieces, and when both partners do that at the same moment the SIMD's matrix pipe idles.  Out of phase,
    // the partner's MFMAs cover the stall (measured 256->256@32x32: 2.56 -> 2.42 ms; splitting by wave parity instead,
    // which pairs waves of different SIMDs, gains nothing; spreading the pieces over the taps loses the gain).
#ifdef SF_SPLIT3
    if (ci == 2 * nreal) {   // phase A -> phase B: the correction terms were accumulated at 2^11 times their value
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mf][nf][i] *= SPLIT_DOWN;
    }
#endif
    const bool stage_late = WAVES == 8 && wave >= 4;
    auto stage_next = [&]() {
#ifdef SF_EXP_NOSTAGE   // ablation (tools/ablate_lstm_cell.sh): no staging of the next chunk - MFMAs on whatever the LDS holds
      return;
#endif
      if (ci + 1 < nch) {
        if constexpr (!WS) issue_weights(ci + 1, cur ^ 1);
        stage_input(ci + 1);
      }
    };
    const char* inb = lds_in + cur * IN_B + a_lane;
    const char* wb = lds_w + (WS ? 0 : cur) * W_B + b_lane;
    auto load_tap = [&](int tap, bf16x8 (&a)[2], bf16x8 (&b)[NF]) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
        a[mf] = *reinterpret_cast<const bf16x8*>(inb + ((2 * mf + ky) * HALO_W + kx) * PIX_B + ((ky & 1) ? a_half_odd : a_half_even));
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) b[nf] = *reinterpret_cast<const bf16x8*>(wb + (tap * NB + nf * 32) * PIX_B);
    };
    if constexpr (NF == 5) {
      // NF = 5 (160 accumulators): the weight fragments are SINGLE-buffered - fragment nf of the next tap is read into the same registers right
      // behind the two MFMAs that consumed it (operands are read at issue), eight MFMAs ahead of its next use; only the two pixel fragments
      // are double-buffered.  20 registers less than two full operand sets: the kernel spilled 36 registers INSIDE the tap loop (1.6 GB of
      // scratch reloads per launch of the 256 -> 160 input gradient: 3.74 GB read against 2.1 GB of operands); the 16 that are still spilled
      // are DMA offsets reloaded once per chunk in the staging blocks.  256 -> 160 @32x32 x 2304: 1.56 -> 1.44 ms.  (Single-buffering the pixel
      // fragments too: same spills, 2 % slower.  Dropping the never-taken fp32 staging path from the BatchNorm-backward variant: the
      // compiler then spills 128 accumulator registers around the epilogue instead, 1.93 ms.)
      bf16x8 fa[2][2], fb1[NF];
      auto load_a = [&](int tap, bf16x8 (&a)[2]) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
          a[mf] = *reinterpret_cast<const bf16x8*>(inb + ((2 * mf + ky) * HALO_W + kx) * PIX_B + ((ky & 1) ? a_half_odd : a_half_even));
      };
      auto load_b = [&](int tap, int nf) { return *reinterpret_cast<const bf16x8*>(wb + (tap * NB + nf * 32) * PIX_B); };
      load_a(0, fa[0]);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) fb1[nf] = load_b(0, nf);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_a(tap + 1, fa[(tap + 1) & 1]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
            acc[mf][nf] = TR ? SF_MFMA_32X32X16(fb1[nf], fa[tap & 1][mf], acc[mf][nf])
                             : SF_MFMA_32X32X16(fa[tap & 1][mf], fb1[nf], acc[mf][nf]);
          if (tap + 1 < 9) fb1[nf] = load_b(tap + 1, nf);
        }
        if (tap + 1 < 9) {   // 2 MFMAs, then reads and MFMAs alternate: every read sits behind the MFMAs that free its registers
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 0 || tap == 5) {
          if (stage_late == (tap == 5)) stage_next();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    bf16x8 fa[2][2], fb[2][NF];
    load_tap(0, fa[0], fb[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
      bool dead = false;
      if constexpr (SHIFT) {  // a tap this chunk's view does not own (block-uniform): operands read, products skipped
        const int sv = (ci + p.chunk0) / p.shift4;
        dead = ((sv >> 1) && tap / 3 == 0) || ((sv & 1) && tap % 3 == 0);
      }
      if (!dead)
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#ifdef SF_EXP_NOMFMA   // ablation: the fragment reads stay (one cheap use each), the matrix instructions go
          acc[mf][nf][0] += (float)fb[tap & 1][nf][0] + (float)fa[tap & 1][mf][0];
#else
          acc[mf][nf] = TR ? SF_MFMA_32X32X16(fb[tap & 1][nf], fa[tap & 1][mf], acc[mf][nf])
                           : SF_MFMA_32X32X16(fa[tap & 1][mf], fb[tap & 1][nf], acc[mf][nf]);
#endif
      // scheduling: one LDS read (the next tap's operands) after each of the first MFMAs of this tap, the remaining
      // MFMAs behind them - a clump of 2+NF reads between two MFMA groups measured 2 % (NF=4) to 15 % (NF=5) slower
      if (tap + 1 < 9) {
        constexpr int READS = 2 + NF, MFMAS = 2 * NF, PAIRS = READS < MFMAS ? READS : MFMAS;
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (MFMAS > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, MFMAS - PAIRS, 0);
        if constexpr (READS > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, READS - PAIRS, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tap == 0 || tap == 5) {
        if (stage_late == (tap == 5)) stage_next();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
#ifdef SF_SPLIT3
    if (ci + 1 < nch) store_input(cur ^ 1, v_xlo(ci + 1));
#else
    if (ci + 1 < nch) store_input(cur ^ 1);  // other buffer: last read in chunk ci-1, every wave is past this chunk's barrier
#endif
    if constexpr (WS) {
      if (ci + 1 < nch) {
        __syncthreads();              // every wave has read its last weight fragment of this chunk
        issue_weights(ci + 1, 0);     // lands while the partner workgroup on this CU computes
      }
    }
  }

#ifdef SF_SPLIT3
  if (p.amax0) {   // block-uniform: the gradient operand's power-of-two scale leaves the accumulators (exact)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][nf][i] *= sx_inv;
  }
#endif
  // optional per-tile BatchNorm statistics of the stored outputs (linear epilogue): lane sums -> LDS (the operand
  // buffers are free once every wave has left the K loop) -> one [32*NF][2] record per workgroup
  float* lds_stats = nullptr;
  if constexpr (EPI == EPI_LINEAR && !DUAL) {
    if (p.stats) {
      __syncthreads();
      lds_stats = reinterpret_cast<float*>(lds);
      for (int i = tid; i < 2 * NB; i += THREADS) lds_stats[i] = 0.f;
      __syncthreads();
    }
  }
  if constexpr (TR && BNB) {
    conv_epilogue_tr_bnb<NF>(acc, p, n, nb, y0, x0, wave, r, kh, lds_coef);
    return;
  } else if constexpr (TR) {
#ifdef SF_EXP_NOEPI   // ablation: no epilogue (one never-taken store keeps the accumulators alive)
    {
      float keep = 0.f;
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int i = 0; i < 16; ++i) keep += acc[mf][nf][i];
      if (keep == 12345.678f) p.c_out[0] = keep;
    }
    return;
#endif
    if (!DUAL || n_w < p.N) conv_epilogue_tr<NF, EPI>(acc, p, n_w, nb, y0, x0, wl, r, kh);
    return;
  } else {
    if (!DUAL || n_w < p.N) conv_epilogue<NF, EPI>(acc, p, n_w, nb, y0, x0, wl, r, kh, lds_stats);
  }
  if constexpr (EPI == EPI_LINEAR && !DUAL) {
    if (lds_stats) {
      __syncthreads();
      const size_t tile_lin = (size_t)(n * p.tiles_y + ty) * p.tiles_x + tx;
      for (int i = tid; i < 2 * NB; i += THREADS)
        p.stats[(tile_lin * p.stats_np + nb * NB + (i % NB)) * 2 + i / NB] = lds_stats[i];
    }
  }
}

// ---- weight repack (bf16 LDS image) ----------------------------------------------------------
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, int O, int I, const int* __restrict__ nmap, int Np,
                                         const int* __restrict__ kmap, int Kp, int NB, int transpose, SF_OP_T* __restrict__ packed,
                                         const float* __restrict__ bias, float* __restrict__ bias_packed, const float* __restrict__ kscale, int groups) {
#ifdef SF_SPLIT3
  const int rchunks = Kp / KC, chunks = 3 * rchunks;   // virtual chunks: [lo'(w) c, hi(w) c] per real chunk c, then hi(w) of every chunk
  const size_t image = (size_t)Np * Kp * 27, total = image * groups;
#else
  const size_t image = (size_t)Np * Kp * 9, total = image * groups;
  const int chunks = Kp / KC;
#endif
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    // logical element e -> [group][nblk][chunk][tap][row][k16]
    const int grp = (int)(e / image);
    size_t t = e - grp * image;
    const int k16 = t % KC; t /= KC;
    const int row = t % NB; t /= NB;
    const int tap = t % 9; t /= 9;
    const int chunk = t % chunks;
    const int nblk = t / chunks;
    const int nn = nmap[nblk * NB + row];
#ifdef SF_SPLIT3
    const int rchunk = chunk < 2 * rchunks ? chunk >> 1 : chunk - 2 * rchunks;
    const bool lo_part = chunk < 2 * rchunks && !(chunk & 1);
    const int kk = kmap[rchunk * KC + k16];
#else
    const int kk = kmap[chunk * KC + k16];
#endif
    float v = 0.f;
    if (nn >= 0 && kk >= 0) v = transpose ? w[((size_t)kk * I + nn) * 9 + (8 - tap)] : w[((size_t)nn * I + kk) * 9 + tap];
    if (kscale) v *= kscale[(size_t)grp * Kp + chunk * KC + k16];
    // physical position: the two 8-element halves of a row are swapped on rows with bit 3 set
    const int half = (k16 >> 3) ^ ((row >> 3) & 1);
    const size_t base = e - k16;
#ifdef SF_SPLIT3
    if (lo_part) v = (v - (float)(SF_OP_T)v) * SPLIT_UP;
#endif
    packed[base + half * 8 + (k16 & 7)] = (SF_OP_T)v;
  }
  if (bias_packed && blockIdx.x == 0)
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
      const int nn = nmap[i];
      bias_packed[i] = (bias && nn >= 0) ? bias[nn] : 0.f;
    }
}

// the shifted-view instantiation of a plain linear launch (ConvParams::shift4 set); false = not such a launch
template <int WAVES, int NFV, int EPI, bool DUAL>
bool launch_shifted(const ConvParams& p, dim3 grid, dim3 block, hipStream_t st) {
  if constexpr (EPI == EPI_LINEAR && NFV != 5) {
    if (p.shift4) {
      hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, NFV, EPI, DUAL, true, false, false, false, true>), grid, block, 0, st, p);
      return true;
    }
  }
  return false;
}

template <int WAVES, int EPI, bool DUAL = false>
int launch_w(const ConvParams& p0, int nf, int nblk, hipStream_t st) {
  ConvParams p = p0;
  constexpr int TH = 4 * WAVES;
  p.tiles_x = DUAL ? 1 : (p.W + TILE_W - 1) / TILE_W;
  p.tiles_y = DUAL ? 1 : (p.H + TH - 1) / TH;
  dim3 grid(DUAL ? (p.N + 1) / 2 : p.tiles_x * p.tiles_y * p.N, nblk), block(WAVES * 64);
  if constexpr (EPI == EPI_LSTM) {
    if (nf != 4) { sf_set_error("bf16 conv: LSTM epilogue needs nf=4"); return 1; }
    if constexpr (WAVES == 4 && !DUAL) {
      // single weight buffer, 58 KB of LDS, two workgroups per CU: one's epilogue (224 KB of stores per item: ~10 us of the CU's memory pipe) runs
      // under the other's K loop.  Default for the large launches of the training shapes since round 4 (cfg 2: 813 -> 819 samples/s, three A/B
      // pairs on one box); SF_LSTM_WS=1 forces it, SF_LSTM_W8=1 restores the 8-wave double-buffered kernel.
      static const bool ws_env = getenv("SF_LSTM_WS") != nullptr, w8 = getenv("SF_LSTM_W8") != nullptr;
      const bool ws = ws_env || (!w8 && (long long)grid.x * grid.y >= 1024);
      if (ws) hipLaunchKernelGGL((conv3x3_bf16_kernel<4, 4, EPI, false, true, false, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, 4, EPI, DUAL, true>), grid, block, 0, st, p);
    } else
    hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, 4, EPI, DUAL, true>), grid, block, 0, st, p);
  } else if constexpr (EPI == EPI_GRU) {
    if (nf != 3) { sf_set_error("bf16 conv: GRU epilogue needs nf=3"); return 1; }
    // channel-per-lane epilogue: on these few small workgroups its coalesced 4-byte state reads beat the 16-byte
    // pixel-per-lane form (22.8 vs 24.9 us per step)
    hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, 3, EPI, DUAL, false>), grid, block, 0, st, p);
  } else {
    if constexpr (EPI == EPI_LINEAR && !DUAL) {
      if (p.split_c > 0) {   // split-K (sf_conv3x3_fwd_splitk planned it): grid z = channel slices
        if (nf != 4 || p.stats || p.bnb_coef || p.bias_tab || p.src1) { sf_set_error("bf16 conv: split-K takes plain nf=4 single-source launches"); return 1; }
        grid.z = (p.c0 + p.split_c - 1) / p.split_c;
        if (p.shift4) hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, 4, EPI, false, true, false, false, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, 4, EPI, false, true, false, false, true>), grid, block, 0, st, p);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) { sf_set_error("conv3x3_bf16 (split-K): launch failed: %s", hipGetErrorString(es)); return 2; }
        return 0;
      }
    }
    // without BatchNorm statistics the product is computed transposed (pixel-per-lane epilogue with 16-byte stores)
    const bool tr = p.stats == nullptr;
#define SF_CONV_CASE(NFV)                                                                                              \
  case NFV:                                                                                                            \
    if (tr) {                                                                                                          \
      if constexpr (EPI == EPI_LINEAR && !DUAL) {                                                                      \
        if (p.bnb_coef) hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, NFV, EPI, DUAL, true, true>), grid, block, 0, st, p); \
        else if (!launch_shifted<WAVES, NFV, EPI, DUAL>(p, grid, block, st))                                           \
          hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, NFV, EPI, DUAL, true>), grid, block, 0, st, p);               \
      } else if (!launch_shifted<WAVES, NFV, EPI, DUAL>(p, grid, block, st))                                           \
        hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, NFV, EPI, DUAL, true>), grid, block, 0, st, p);                 \
    }                                                                                                                  \
    else if constexpr (EPI == EPI_LINEAR && !DUAL) hipLaunchKernelGGL((conv3x3_bf16_kernel<WAVES, NFV, EPI, DUAL, false>), grid, block, 0, st, p); \
    else { sf_set_error("bf16 conv: statistics need the linear epilogue on single-image tiles"); return 1; }          \
    break;
    switch (nf) {
      SF_CONV_CASE(1) SF_CONV_CASE(2) SF_CONV_CASE(3) SF_CONV_CASE(4) SF_CONV_CASE(5)
      default: sf_set_error("bf16 conv: unsupported nf=%d", nf); return 1;
    }
#undef SF_CONV_CASE
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error("conv3x3_bf16: launch failed: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

template <int EPI>
int launch_e(const ConvParams& p, int nf, int nblk, hipStream_t st) {
  // 8-wave 32x16 tiles when the image has more than 16 rows.  Images of at most 16x16 pixels: 4-wave 16x16 tiles, except
  // where that kernel fits only one workgroup per CU (NF >= 4: 94 KB LDS) and there are enough images to fill the chip
  // with 8-wave workgroups holding two images each (measured 192->256 @16x16 x 2304: 811 -> 574 us; NF = 3 and the
  // 96-image ConvGRU steps are faster on the 4-wave kernel)
  if constexpr (EPI == EPI_LSTM) {  // the fused cell on 16x16 tiles, two workgroups per CU, when that gives at least 1024 workgroups (see launch_w)
    static const bool w4 = getenv("SF_LSTM_W4") != nullptr, w8 = getenv("SF_LSTM_W8") != nullptr;
    const long long wgs16 = (long long)((p.W + TILE_W - 1) / TILE_W) * ((p.H + 15) / 16) * p.N * nblk;
    if (w4 || (!w8 && p.H > 16 && wgs16 >= 1024)) return launch_w<4, EPI>(p, nf, nblk, st);
  }
  if (p.H > 16) return launch_w<8, EPI>(p, nf, nblk, st);
  // round 5: NF = 3 with a bf16-STORED source (the x-part of MetNet's ConvGRU, 256 -> 192 on 2304 maps: LDS-DMA operands) takes the two-image kernel too:
  // 601 -> 525 us, two A/B pairs in one call (tools/probe_gru_xpart.py; SF_CONV_NO_DUAL_NF3=1: back); fp32-stored NF = 3 launches stay on the 4-wave kernel
  static const bool no_dual3 = getenv("SF_CONV_NO_DUAL_NF3") != nullptr;
  if (p.W <= 16 && !p.stats && (nf >= 4 || (nf == 3 && p.bf0 && !p.src1 && !no_dual3)) && p.N >= 512) return launch_w<8, EPI, true>(p, nf, nblk, st);
  return launch_w<4, EPI>(p, nf, nblk, st);
}

}  // namespace

int sf_conv_bf16_tiles(int h, int w) {
  const int th = h > 16 ? 32 : 16;
  return ((w + TILE_W - 1) / TILE_W) * ((h + th - 1) / th);
}

int sf_launch_conv_bf16(const sfconv::ConvParams& p, int nf, int nblk, int epi, hipStream_t st) {
  // large single-source bf16-stored launches: one persistent workgroup per CU (SF_NO_PERSIST_CONV=1: A/B switch)
#ifndef SF_OPERAND_F16
  static const bool no_persist = getenv("SF_NO_PERSIST_CONV") != nullptr;
  if (!no_persist && !p.split_c && sf_conv_bf16_persist_ok(p, epi, nf)) return sf_launch_conv_bf16_persist(p, nf, nblk, st);
#elif defined(SF_SPLIT3)
  if (p.bf0 || p.bf1 || p.out_bf || p.gates_bf || p.hout_bf || p.bias_tab || p.bnb_coef || p.split_c || p.shift4 || p.chunks_total % 3) {
    sf_set_error("f32e conv: fp32-stored tensors; no folded BatchNorm / split-K / shifted views");
    return 1;
  }
#else
  if (p.bf0 || p.bf1 || p.out_bf || p.stats || p.bias_tab || p.bnb_coef || epi == EPI_LSTM || epi == EPI_GRU) {
    sf_set_error("f16 conv: fp32-stored tensors, linear / sigmoid epilogue only");
    return 1;
  }
#endif
  switch (epi) {
    case EPI_LINEAR: return launch_e<EPI_LINEAR>(p, nf, nblk, st);
    case EPI_SIGMOID: return launch_e<EPI_SIGMOID>(p, nf, nblk, st);
    case EPI_LSTM: return launch_e<EPI_LSTM>(p, nf, nblk, st);
    case EPI_GRU: return launch_e<EPI_GRU>(p, nf, nblk, st);
  }
  sf_set_error("bf16 conv: unknown epilogue %d", epi);
  return 1;
}

void sf_pack_weights_bf16(const float* w, int O, int I, const int* nmap, int Np, const int* kmap, int Kp, int NB, int transpose,
                          void* packed, const float* bias, float* bias_packed, hipStream_t st, const float* kscale, int groups) {
#ifdef SF_SPLIT3
  const size_t total = (size_t)Np * Kp * 27 * groups;
#else
  const size_t total = (size_t)Np * Kp * 9 * groups;
#endif
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks), dim3(256), 0, st, w, O, I, nmap, Np, kmap, Kp, NB, transpose,
                     (SF_OP_T*)packed, bias, bias_packed, kscale, groups);
}
