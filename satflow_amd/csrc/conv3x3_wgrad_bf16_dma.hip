// Weight gradient of the 3x3 same convolution on the bf16 matrix cores for bf16-STORED tensors ("bf16a" mode):
//
//   dW[tap][co][ci] = sum over pixels  dout[pixel][co] * in[pixel + tap][ci],   fp32 accumulate
//
// GEMM view: M = co, N = ci, K = pixels.  The MFMA wants 8 consecutive K values (pixels) of ONE channel per lane while
// HBM holds NHWC (channel-contiguous).  Round 1 transposed in registers with dedicated loader waves (conv3x3_wgrad_bf16.hip:
// 45 % of the MFMA peak, loaders and compute waves never overlapped fully).  Here nobody transposes:
//   * the NHWC tiles go HBM -> LDS unchanged by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write),
//     four K tiles deep (a 128 KiB ring), with counted `s_waitcnt vmcnt` and raw `s_barrier`: two tiles stay in flight
//     across every barrier;
//   * the MFMA operands are read with `ds_read_b64_tr_b16`: a 16-lane group hands in the addresses of four pixels x
//     16 channels and every lane receives the four pixels of ITS channel - the transpose is done by the LDS read.  A
//     horizontal tap shift is just another pixel address: no shifted copies, no v_alignbit, no halo side array.
// Workgroup = 8 compute waves (2 per SIMD), slab 128 (co) x 64 (ci) x 9 taps: wave w owns co fragment w & 3 and ci half
// w >> 2, i.e. 9 accumulator tiles (144 VGPRs).  K tile = 4 rows x 16 pixels; per tile a wave reads 4 dout fragments and
// 18 input fragments (6 halo rows x 3 horizontal shifts, each feeding up to 3 vertical taps) for 36 MFMAs.
// LDS image of a stage (32 KiB): dout [64 px][256 B] with the four 64-byte channel quarters of a pixel xor-ed with
// (pixel & 3); input halo [6 x 18 px][128 B] with the two 64-byte halves xor-ed with (x >> 1) & 1 - both make the
// four pixels of a transposing read fall on disjoint banks (the swizzle is applied to the DMA's SOURCE address).
// Out-of-image / out-of-range pieces are fetched from a 16-byte zero page instead of being masked off, so every wave
// issues exactly 4 DMA instructions per tile and the counted waits stay exact.
// The bias gradient is summed from the dout fragments (VALU, one wave in eight).  Split-K partial slabs in the shared
// layout; `wgrad_reduce_kernel` (conv3x3_wgrad_f32.hip) finishes.
#include <type_traits>

#include "wgrad_common.h"

namespace {

using namespace sfwgrad;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));

constexpr int TR = 4, TW = 16;             // K tile: 4 rows x 16 pixels
constexpr int HR = TR + 2, HW = TW + 2;    // halo tile 6 x 18
constexpr int NS_REG = 4;                  // ring depth (tiles) of the regular slab: 3, 4 and 5 measured alike (the DMA's latency is covered)
constexpr int LDS_BYTES = NS_REG * 32768;  // regular: 4 stages x (16 KiB dout + 16 KiB input); wide: 3 x 40 KiB
#ifndef SF_EXP_NS_POOLED
#define SF_EXP_NS_POOLED 6
#endif
constexpr int NS_POOLED = SF_EXP_NS_POOLED;               // pooled sparse operand: stages of 8 KiB + 16 KiB
constexpr int LDS_BYTES_POOLED = NS_POOLED * 24576;
constexpr int THREADS = 512;

__device__ __forceinline__ bf16x4_t tr_read(unsigned lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(uintptr_t)lds_addr);
}
// LDS-DMA issued from inline asm: hipcc must not know about it - it would drain every pending DMA (`s_waitcnt vmcnt(0)`) in front
// of each LDS read that might alias, i.e. in front of every fragment read, and the ring would never have a tile in flight
// (cdna_hip_programming.md 5.7).  M0 (the wave-uniform LDS destination) is written in the same statement and restored.
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ bf16x8 cat8(bf16x4_t a, bf16x4_t b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// LDS-DMA through a buffer descriptor: wave-uniform descriptor + scalar offset (the tile position) + a per-lane byte offset that is
// CONSTANT for the whole kernel; a lane whose piece lies outside the image gets an out-of-range offset and the hardware
// range check writes zeros to its LDS slot (tools/ubench/buf_lds_oob.hip) - one v_cndmask per instruction instead of a
// 64-bit address per lane.  M0 is not used by anything else in this kernel (no compiler-issued LDS-DMA), so it is not saved.
__device__ __forceinline__ void bufdma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_dst) {
  // (readfirstlane: the values are wave-uniform, but the compiler may hold them in vector registers)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(lds_dst)),
               "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}

// wave-uniform pointer the compiler may have computed on the vector ALU -> scalar registers (inline asm "s" operands are not legalised)
__device__ __forceinline__ void* uniform_ptr(const void* q) {
  const uintptr_t v = (uintptr_t)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (void*)(((uintptr_t)hi << 32) | lo);
}

// FAST: every 64-channel input tile of the block comes from ONE source (single source, or c0 a multiple of 64): DMA through
// buffer descriptors with kernel-constant lane offsets.  Otherwise (the ConvLSTM's 16-lane x source in front of h) per-lane
// 64-bit addresses.
// GROUPED (folded BatchNorm, sf_conv3x3_bwd_weight_folded): the images come in groups of p.tpg tiles whose raw gradients are
// scaled differently afterwards, so a slice that crosses a group boundary stores its accumulators there (segment 0, 1, ...
// of the slice: partial[ks * maxseg + seg]) and starts again from zero.
// GEO (1, 2: FAST only): slab geometries for half-empty edge tiles.  With the regular geometry (0: 128 co x 64 ci) the waves of a dead half multiply
// zeros.  GEO 1 - the LAST ci tile holds at most 32 channels (160 = 64 + 64 + 32 input channels: conv2 of MetNet's DownSampler): a workgroup takes the
// live halves of TWO co tiles, slab 256 co x 32 ci, wave w owns co fragment w of the 8; stage = 64 pixels x 512 bytes of dout + halo x 64 bytes of
// input.  GEO 2 - the LAST co tile holds at most 64 channels (192 = 128 + 64 output channels: the ConvGRU's gates; 160 = 128 + 32: conv1): a workgroup
// takes the live halves of TWO ci tiles, slab 64 co x 128 ci, wave w owns co fragment w & 1 and ci fragment w >> 1; stage = 64 pixels x 128 bytes of
// dout + halo x 256 bytes of input.  Both: 40 KiB per stage, three stages, five DMA instructions per wave and tile instead of four.  Every workgroup
// of the launch then does the same MFMA work on live operands, and the plan cuts the K slices shorter.
// SPARSE (GEO 0; round 5): dout is the gradient behind a 2x2 max-pooling - one non-zero per window and channel, i.e. at most two in any four consecutive
// pixels of a row - and goes in as the SPARSE operand of v_smfmac_f32_32x32x32_bf16 (operand layout and index encoding established on hardware:
// tools/ubench/smfmac_probe.hip; 1.9x the dense stream on random data).  One sparse instruction multiplies TWO dout rows (K = 32 pixels): the dense
// fragment of a row (a lane holds 8 consecutive pixels = two four-groups of its channel) is compressed in registers - per pixel pair the non-zero value and
// its position bit - into 4 values + 2 index nibbles, rows r and r + 2 share an instruction; the dense operand is the halo rows (a, a + 2) read by the two
// lane halves (the lane's 16 pixels of ITS row: four transposing reads).  18 instead of 36 matrix instructions per tile, the same products (zeros are
// skipped), 48 instead of 36 fragment reads.  A dout that breaks the structure would be multiplied WRONGLY: the entry point is a separate one.
template <bool FAST, bool GROUPED, int GEO, int SPARSE = 0>   // SPARSE: 0 dense, 1 dout as the 2:4-sparse operand, 2 the same operand from the POOLED gradient + routing record
__device__ __forceinline__ void wgrad_dma_body(const WgradParams& p, const int per_slice, const char* __restrict__ zero, char* lds, const int ks,
                                               const int cot, const int cit) {
  static_assert(GEO == 0 || FAST, "the edge geometries use the descriptor DMA path");
  constexpr bool WIDE = GEO != 0;
  constexpr int A_PIX = GEO == 1 ? 4 * DMA_CO_T : GEO == 2 ? DMA_CO_T : 2 * DMA_CO_T;   // bytes per dout pixel in a stage (256 / 64 / 128 channels)
  constexpr int B_PIX = GEO == 1 ? DMA_CI_T : GEO == 2 ? 4 * DMA_CI_T : 2 * DMA_CI_T;   // bytes per input pixel (32 / 128 / 64 channels)
  // (SPARSE 2: the pooled gradient's 4 KB + 512 B of codes + the dummy pieces of waves 5-7 in 8 KB)
  constexpr int A_BYTES = SPARSE == 2 ? 8192 : TR * TW * A_PIX;                         // 32768 / 8192 / 16384
  constexpr int B_BYTES = GEO == 1 ? 8192 : GEO == 2 ? 32768 : 16384;                   // 108 px x B_PIX, padded to whole 8-block rounds
  constexpr int STAGE = A_BYTES + B_BYTES;
  // ring depth.  SPARSE 2: SIX stages of 24 KB - a tile of the sparse kernel is half the matrix work of a dense one, so the same DMA latency spans twice
  // as many tiles: with four stages (two tiles in flight behind the rendezvous) the fetch rate was bytes-in-flight / latency, 0.35 ms of the call
  constexpr int NS = WIDE ? 3 : SPARSE == 2 ? ::NS_POOLED : ::NS_REG;
  constexpr bool POOLED = SPARSE == 2;
  static_assert(!POOLED || (FAST && GEO == 0), "the pooled operand is built for the regular slabs of the descriptor DMA path");
  // DMA instructions per wave and tile: 2 + 2 / 4 + 1 / 1 + 4; POOLED: ONE for the sparse operand's sources (waves 0-3: the pooled gradient's 4 KB,
  // wave 4: 512 bytes of routing codes, waves 5-7: an all-out-of-range instruction that only keeps the counted waits uniform) + 2
  constexpr int NA = POOLED ? 1 : A_BYTES / 1024 / 8, NBK = B_BYTES / 1024 / 8, NDMA = NA + NBK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = GEO == 2 ? wave & 1 : wave & 3;                 // co fragment of this wave (within its co tile)
  const int wc = GEO == 1 ? 0 : GEO == 2 ? wave >> 1 : wave >> 2;  // ci fragment
  const int cotw = GEO == 1 ? wave >> 2 : 0;                     // GEO 1: which of the two co tiles
  // a slice owns a CONTIGUOUS run of tiles: vertically adjacent tiles re-read two halo rows, which are then L2 hits
  const int t_begin = ks * per_slice;
  const int t_end = t_begin + per_slice < p.ntiles ? t_begin + per_slice : p.ntiles;
  const int my_tiles = t_end > t_begin ? t_end - t_begin : 0;

  // ---- per-lane constants of the DMA pieces this wave issues: dout blocks {wave, wave + 8}, input blocks {wave, wave + 8} ----
  // dout block a = pixels 4a .. 4a+3 (row a / 4, x = 4 * (a % 4) + lane / 16); piece c = lane % 16 sits in physical quarter
  // c / 4 and carries logical quarter (c / 4) ^ (pixel & 3)
  const int a_px = lane >> 4;  // (regular geometry; the wide one has its own per-block constants in the fast path below)
  const int a_ch = cot * DMA_CO_T + ((((lane >> 2) & 3) ^ a_px) * 4 + (lane & 3)) * 8;
  const bool a_chok = a_ch < p.dc;
  const long long a_pxb = 2ll * p.ds;  // bytes per dout pixel in HBM
  // input block b = halo pixels 8b .. 8b+7 (hp = 8b + lane / 8 -> hy = hp / 18, hx = hp % 18); piece c = lane % 8 sits in
  // physical half c / 4 and carries logical half (c / 4) ^ ((hx >> 1) & 1)
  int b_hy[2], b_hx[2], b_off[2];
  bool b_ok[2], b_s1[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int hp = 8 * (wave + 8 * u) + (lane >> 3);
    b_hy[u] = hp / HW; b_hx[u] = hp % HW;
    const int kc = cit * DMA_CI_T + (((((lane >> 2) & 1) ^ ((b_hx[u] >> 1) & 1)) * 4) + (lane & 3)) * 8;  // concatenated padded K space
    b_s1[u] = kc >= p.c0;
    b_ok[u] = hp < HR * HW && (kc < p.c0 ? p.src0 != nullptr : (kc - p.c0 < p.c1 && p.src1 != nullptr));
    const int s = b_s1[u] ? p.s1 : p.s0, ch = b_s1[u] ? kc - p.c0 : kc;
    b_off[u] = 2 * ((b_hy[u] * p.W + b_hx[u]) * s + ch);
  }

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  // Tiles are issued strictly in order, one `issue()` per tile: the tile coordinates advance incrementally (no divisions in the loop).
  int nx_i = 0, nx_stage = 0, nx_tx, nx_ty, nx_n;
  {
    int t = t_begin < p.ntiles ? t_begin : 0;
    nx_tx = t % p.tiles_x; t /= p.tiles_x;
    nx_ty = t % p.tiles_y;
    nx_n = t / p.tiles_y;
  }
  const bool remap = p.idiv0 > 1 || p.imod0 > 0 || p.idiv1 > 1 || p.imod1 > 0;  // block-uniform
  const long long a_img = (long long)p.H * p.W * a_pxb, b_img0 = (long long)p.H * p.W * 2 * p.s0, b_img1 = (long long)p.H * p.W * 2 * p.s1;
  auto issue_slow = [&]() {  // DMA of this slice's next tile into ring stage nx_i % NS (tiles past the end fetch zeros: the counts stay exact)
    const bool live = nx_i < my_tiles;
    const int n = nx_n, x0 = nx_tx * TW, y0 = nx_ty * TR;
    const unsigned stage = lds0 + (unsigned)(nx_stage * STAGE);
    const char* abase = (const char*)p.dout + n * a_img + (y0 * p.W + x0) * (int)a_pxb;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int a = wave + 8 * u, row = a >> 2, x = 4 * (a & 3) + a_px;
      const bool ok = live && a_chok && y0 + row < p.H && x0 + x < p.W;
      const char* g = ok ? abase + ((row * p.W + x) * (int)a_pxb + 2 * a_ch) : zero;
      glds16(g, stage + a * 1024);
    }
    int ns0 = n, ns1 = n;
    if (remap) {
      ns0 = n / p.idiv0; if (p.imod0) ns0 %= p.imod0;
      ns1 = n / p.idiv1; if (p.imod1) ns1 %= p.imod1;
    }
    const int org = (y0 - 1) * p.W + (x0 - 1);  // halo origin (may lie before the image: only valid lanes dereference)
    const char* bbase0 = (const char*)p.src0 + ns0 * b_img0 + (long long)org * (2 * p.s0);
    const char* bbase1 = (const char*)p.src1 + ns1 * b_img1 + (long long)org * (2 * p.s1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int gy = y0 - 1 + b_hy[u], gx = x0 - 1 + b_hx[u];
      const bool ok = live && b_ok[u] && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const char* g = ok ? (b_s1[u] ? bbase1 : bbase0) + b_off[u] : zero;
      glds16(g, stage + A_BYTES + (wave + 8 * u) * 1024);
    }
    ++nx_i;
    nx_stage = nx_stage + 1 == NS ? 0 : nx_stage + 1;
    if (++nx_tx == p.tiles_x) { nx_tx = 0; if (++nx_ty == p.tiles_y) { nx_ty = 0; ++nx_n; } }
  };

  // ---- fast path: kernel-constant lane offsets + per-lane validity masks over the 16 tile classes ----
  // tile class = (ty == 0) | (ty == tiles_y - 1) << 1 | (tx == 0) << 2 | (tx == tiles_x - 1) << 3: the validity of a piece depends
  // on the tile only through its class (first / last tile row and column: image borders and ragged edges)
  constexpr unsigned SENT = 0x80000000u;  // >= any descriptor's num_records (the launcher checks image bytes < 2^31)
  unsigned fb_off[NBK], fb_mask[NBK];
  const bool from0 = cit * DMA_CI_T < p.c0;  // block-uniform; exact when FAST
  const float* bsrc = from0 ? p.src0 : p.src1;
  const int bs = from0 ? p.s0 : p.s1, bidiv = from0 ? p.idiv0 : p.idiv1, bimod = from0 ? p.imod0 : p.imod1;
  // dout block a = pixels PPB * a .. of the tile (regular: 4 pixels of 256 bytes, piece lane % 16 in physical quarter (lane / 4) % 4 carrying logical
  // quarter ^ (pixel & 3); wide: 2 pixels of 512 bytes, piece lane % 32 in physical eighth, the xor on its low two bits).  Row and first column of a
  // block are WAVE constants (scalar offset of the DMA), the lane adds its pixel within the block and its channel piece: one per-lane offset for all
  // of a wave's blocks, validity from two comparisons per block instead of a class mask per block (registers: the wide body holds four blocks)
  constexpr int PPB = 1024 / A_PIX;          // pixels per dout block: 4 / 2 / 8
  const int a_pl = lane / (64 / PPB);
  unsigned fa_voff = 0;
  if constexpr (FAST) {
    const int P = PPB * wave + a_pl;         // (pixel & 3) of this lane's pixel is the same in every block of the wave (blocks are 8 apart)
    const int c16 = lane & (A_PIX / 16 - 1), pq = c16 >> 2;
    // 64-byte channel groups of a pixel are xor-ed so that the four pixels of a transposing read fall on disjoint banks: with 256 / 512 bytes per
    // pixel by (pixel & 3) on the low two bits, with 128 bytes per pixel (two groups; pixels m and m + 2 share banks) by (pixel >> 1) & 1
    const int lq = GEO == 2 ? pq ^ ((P >> 1) & 1) : (pq & 4) | ((pq & 3) ^ (P & 3));
    const int ch = cot * DMA_CO_T + lq * 32 + (c16 & 3) * 8;
    fa_voff = ch < p.dc ? (unsigned)((a_pl * p.ds + ch) * 2) : 0x80000000u;
#pragma unroll
    for (int u = 0; u < NBK; ++u) {
      // input block b: regular = halo pixels 8b .. 8b+7 (128 bytes each), piece lane % 8 in physical half carrying logical half ^ ((hx >> 1) & 1);
      // wide = halo pixels 16b .. 16b+15 (64 bytes each), piece lane % 4, no swizzle (four consecutive pixels are one 256-byte bank row)
      const int b = wave + 8 * u;
      constexpr int PPBB = 1024 / B_PIX;     // halo pixels per input block: 8 / 16 / 4
      const int hp = PPBB * b + lane / (64 / PPBB);
      const int hy = hp / HW, hx = hp % HW;
      const int cb = lane & (B_PIX / 16 - 1), pqb = cb >> 2;
      const int lqb = GEO == 1 ? 0 : GEO == 2 ? pqb ^ (hx & 3) : pqb ^ ((hx >> 1) & 1);  // (64 bytes per pixel: four consecutive pixels are one bank row)
      const int kc = cit * DMA_CI_T + lqb * 32 + (cb & 3) * 8;
      const int ch = from0 ? kc : kc - p.c0;
      const bool chok = hp < HR * HW && bsrc != nullptr && ch < (from0 ? p.c0 : p.c1);
      fb_off[u] = (unsigned)(((hy * p.W + hx) * bs + ch) * 2);
      fb_mask[u] = 0;
      for (int cls = 0; cls < 16; ++cls) {
        const int y0 = (cls & 2) ? (p.tiles_y - 1) * TR : ((cls & 1) ? 0 : TR), x0 = (cls & 8) ? (p.tiles_x - 1) * TW : ((cls & 4) ? 0 : TW);
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        if (chok && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) fb_mask[u] |= 1u << cls;
      }
    }
  }
  const unsigned b_pxb = 2u * (unsigned)bs;
  const long long b_img = (long long)p.H * p.W * b_pxb;
  // POOLED: this lane's piece of the tile's 2 x 8 pooled pixels.  Waves 0-3: piece P = 64 wave + lane = pooled pixel P / 16 (row P / 128, column (P / 16) % 8),
  // 16-byte channel piece P % 16 in physical quarter (P % 16) / 4 carrying logical quarter ^ (pixel & 3) - the dense layout's swizzle, [pixel][256 B] at the
  // stage's origin; wave 4, lanes 0-31: routing codes of pooled pixel lane / 2, octets 8 (lane % 2) .. + 7 of this co tile, [pixel][32 B] at + 4096
  const int Hp = p.H >> 1, Wp = p.W >> 1;
  [[maybe_unused]] unsigned pa_voff = SENT;
  [[maybe_unused]] const long long g_img = (long long)Hp * Wp * p.pool_s * 2, r_img = (long long)Hp * Wp * (p.dc >> 3) * 2;
  if constexpr (POOLED) {
    if (wave < 4) {
      const int P = 64 * wave + lane, pp = P >> 4, c16 = P & 15;
      const int ch = cot * DMA_CO_T + (((c16 >> 2) ^ (pp & 3)) * 32) + (c16 & 3) * 8;
      if (ch < p.dc) pa_voff = (unsigned)((((pp >> 3) * Wp + (pp & 7)) * p.pool_s + ch) * 2);
    } else if (wave == 4 && lane < 32) {
      const int pp = lane >> 1, oct = cot * (DMA_CO_T / 8) + (lane & 1) * 8;
      if (oct * 8 < p.dc) pa_voff = (unsigned)((((pp >> 3) * Wp + (pp & 7)) * (p.dc >> 3) + oct) * 2);
    }
  }
  auto pooled_image = [&](int n) {   // the pooling's outer permutation of the image index
    if (p.pool_L == 0) return n;
    const int b = n % p.pool_B, t = (n / p.pool_B) % p.pool_T, l = n / (p.pool_B * p.pool_T);
    return (t * p.pool_L + l) * p.pool_B + b;
  };
  [[maybe_unused]] int nx_ng = POOLED ? pooled_image(nx_n) : 0;
  auto issue_fast = [&]() {
    const bool live = nx_i < my_tiles;
    const int n = nx_n, px0 = (nx_ty * TR) * p.W + nx_tx * TW;
    const unsigned cls = (nx_ty == 0 ? 1u : 0u) | (nx_ty == p.tiles_y - 1 ? 2u : 0u) | (nx_tx == 0 ? 4u : 0u) | (nx_tx == p.tiles_x - 1 ? 8u : 0u);
    const unsigned sel = live ? 1u << cls : 0u;  // dead tiles (past the slice's end) fetch zeros: the DMA counts stay exact
    const unsigned stage = lds0 + (unsigned)(nx_stage * STAGE);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.dout + n * a_img), 0, __builtin_amdgcn_readfirstlane((int)a_img), 0x00020000);
    const unsigned soa = (unsigned)px0 * (unsigned)a_pxb;
    const int rows_left = p.H - nx_ty * TR, cols_left = p.W - nx_tx * TW;
    if constexpr (POOLED) {
      const unsigned pix0 = (unsigned)((nx_ty * (TR / 2)) * Wp + nx_tx * (TW / 2));
      const __amdgpu_buffer_rsrc_t rsg = wave < 4
          ? __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.pool_g + nx_ng * g_img), 0, __builtin_amdgcn_readfirstlane((int)g_img), 0x00020000)
          : __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.pool_route + n * r_img), 0, __builtin_amdgcn_readfirstlane((int)r_img), 0x00020000);
      bufdma16(live ? pa_voff : SENT, rsg, pix0 * (unsigned)(wave < 4 ? p.pool_s * 2 : (p.dc >> 3) * 2), stage + wave * 1024);
    } else
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int a = wave + 8 * u, row_a = (PPB * a) >> 4, x_a = (PPB * a) & 15;
      const bool ok = live && row_a < rows_left && x_a + a_pl < cols_left;
#ifdef SF_EXP_WG_NOADMA   // timing experiment: no dout traffic (zero-filled pieces)
      bufdma16(ok && false ? fa_voff : SENT, rsa, soa + (unsigned)((row_a * p.W + x_a) * (int)a_pxb), stage + a * 1024);
#else
      bufdma16(ok ? fa_voff : SENT, rsa, soa + (unsigned)((row_a * p.W + x_a) * (int)a_pxb), stage + a * 1024);
#endif
    }
    int ns = n;
    if (remap) { ns = n / bidiv; if (bimod) ns %= bimod; }
    // descriptor starts one image row + one pixel BEFORE the image so that the halo origin has a non-negative offset
    const long long lead = (long long)(p.W + 1) * b_pxb;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)bsrc + ns * b_img - lead), 0, __builtin_amdgcn_readfirstlane((int)(b_img + 2 * lead)), 0x00020000);
    const unsigned sob = (unsigned)px0 * b_pxb;
#pragma unroll
#ifdef SF_EXP_WG_NOBDMA   // timing experiment: no input traffic
    for (int u = 0; u < NBK; ++u) bufdma16(SENT, rsb, sob, stage + A_BYTES + (wave + 8 * u) * 1024);
#else
    for (int u = 0; u < NBK; ++u) bufdma16((fb_mask[u] & sel) ? fb_off[u] : SENT, rsb, sob, stage + A_BYTES + (wave + 8 * u) * 1024);
#endif
    ++nx_i;
    nx_stage = nx_stage + 1 == NS ? 0 : nx_stage + 1;
    if (++nx_tx == p.tiles_x) { nx_tx = 0; if (++nx_ty == p.tiles_y) { nx_ty = 0; ++nx_n; if constexpr (POOLED) nx_ng = pooled_image(nx_n); } }
  };
  auto issue = [&]() {
    if constexpr (FAST) issue_fast(); else issue_slow();
  };
  (void)a_px; (void)a_chok; (void)b_ok; (void)b_s1; (void)b_off; (void)b_hy; (void)b_hx;

  // ---- per-lane LDS read addresses (transposing reads: lane l of a 16-lane group supplies pixel (l >> 2) & 3, 8-byte chunk l & 3) ----
  const int m = (lane >> 2) & 3, khalf = lane >> 5;
  const int cbyte = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const unsigned a_base = lds0 + (8 * khalf + m) * A_PIX + ((GEO == 2 ? wq ^ ((m >> 1) & 1) : cotw * 4 + (wq ^ m)) * 64) + cbyte;  // + row * TW * A_PIX + r * 4 * A_PIX
  unsigned b_base[3];                                                               // + hrow * 2304 + r * 512
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
    b_base[kx] = lds0 + A_BYTES + (8 * khalf + m + kx) * B_PIX + (GEO == 1 ? 0 : GEO == 2 ? (wc ^ ((m + kx) & 3)) * 64 : (wc ^ (((m + kx) >> 1) & 1)) * 64) + cbyte;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  const bool want_bias = cit == 0 && wc == 0;  // wave-uniform (wide: every wave has its own co fragment)

  // prologue: NS - 1 tiles in flight, tile 0 visible
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NS - 2)) : "memory");
  __builtin_amdgcn_s_barrier();

  // fragment reads of ring stage st (base & 0x3ffff: tells the compiler the LDS address is non-negative so that it folds the row /
  // half offsets into the instructions' immediate offset fields instead of recomputing an address per read)
  int rd_stage = 0;  // ring stage of the tile whose fragments are requested next
  auto stage_addr = [&](unsigned& aA, unsigned (&aB)[3]) {
    const unsigned so = (unsigned)(rd_stage * STAGE);
    rd_stage = rd_stage + 1 == NS ? 0 : rd_stage + 1;
    aA = (a_base + so) & 0x3ffffu;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aB[kx] = (b_base[kx] + so) & 0x3ffffu;
  };
  auto load_a = [&](unsigned aA, bf16x8 (&arow)[TR]) {
#pragma unroll
    for (int row = 0; row < TR; ++row) arow[row] = cat8(tr_read(aA + row * (TW * A_PIX)), tr_read(aA + row * (TW * A_PIX) + 4 * A_PIX));
  };
  auto load_b = [&](const unsigned (&aB)[3], int hrow, bf16x8 (&b)[3]) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) b[kx] = cat8(tr_read(aB[kx] + hrow * (HW * B_PIX)), tr_read(aB[kx] + hrow * (HW * B_PIX) + 4 * B_PIX));
  };
  auto mfma_row = [&](int hrow, const bf16x8 (&arow)[TR], const bf16x8 (&b)[3]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int row = hrow - ky;
      if (row < 0 || row >= TR) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(arow[row], b[kx], acc[ky * 3 + kx], 0, 0, 0);
    }
  };

  // ---- SPARSE: operands of the 2:4 instruction ----
  struct SpA { bf16x8 a02, a13; int i02, i13; float bias; };   // dout rows (0, 2) and (1, 3) of a tile, compressed; their index words; the tile's bias sum
  // one dense row fragment (8 consecutive pixels of this lane's channel = 4 pixel pairs) -> 4 values + one index byte: per pair the non-zero value (a pair
  // holds at most one; -0 counts as zero) and its position - slot 2q: position 0 / 1 of four-group q, slot 2q + 1: position 2 / 3
  auto compress24 = [](const bf16x8& row, unsigned& v0, unsigned& v1, unsigned& nib2) __attribute__((always_inline)) {
    const u32x4 d = __builtin_bit_cast(u32x4, row);
    unsigned val[4], pos[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool lo = (d[q] & 0x7fffu) != 0u;
      val[q] = lo ? d[q] : d[q] >> 16;
      pos[q] = lo ? 0u : 1u;
    }
    v0 = __builtin_amdgcn_perm(val[1], val[0], 0x05040100u);
    v1 = __builtin_amdgcn_perm(val[3], val[2], 0x05040100u);
    nib2 = pos[0] | ((2u + pos[1]) << 2) | (pos[2] << 4) | ((2u + pos[3]) << 6);
  };
  // per-lane read addresses of the sparse instruction's DENSE operand: lane half kb reads the 16 pixels of halo row a + 2 kb (four transposing reads)
  unsigned bs_base[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
    bs_base[kx] = lds0 + A_BYTES + (m + kx) * B_PIX + (GEO == 0 ? (wc ^ (((m + kx) >> 1) & 1)) * 64 : 0) + cbyte + khalf * (2 * HW * B_PIX);
  auto load_b16 = [&](const unsigned (&aBS)[3], int a, bf16x16 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned ad = aBS[kx] + a * (HW * B_PIX);
      const bf16x8 lo = cat8(tr_read(ad), tr_read(ad + 4 * B_PIX)), hi = cat8(tr_read(ad + 8 * B_PIX), tr_read(ad + 12 * B_PIX));
      b[kx] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    }
  };
  // steps 2 and 3: the operand of halo rows (a, a + 2) from the operand of rows (a - 2, a) already in registers - its upper half (row a) moves to the lower
  // lanes (v_permlane32_swap of a register with itself leaves the upper half's value in both halves of the second result), only the UPPER lanes read LDS (row
  // a + 2): the wave's 64-lane read becomes a 32-lane one, and the tile's fragment reads go from 56 to 44 wave-equivalents
  auto shift_b16 = [&](const unsigned (&aBS)[3], int a, bf16x16 (&b)[3]) __attribute__((always_inline)) {
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      u32x8 w = __builtin_bit_cast(u32x8, b[kx]);
#pragma unroll
      for (int d = 0; d < 8; ++d) w[d] = __builtin_amdgcn_permlane32_swap(w[d], w[d], false, false)[1];
      if (khalf) {   // (aBS carries the upper lanes' + 2 rows already)
        const unsigned ad = aBS[kx] + a * (HW * B_PIX);
        const bf16x8 lo = cat8(tr_read(ad), tr_read(ad + 4 * B_PIX)), hi = cat8(tr_read(ad + 8 * B_PIX), tr_read(ad + 12 * B_PIX));
        w = __builtin_bit_cast(u32x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
      }
      b[kx] = __builtin_bit_cast(bf16x16, w);
    }
  };
  auto stage_addr_s = [&](unsigned& aA, unsigned (&aBS)[3]) {
    const unsigned so = (unsigned)(rd_stage * STAGE);
    rd_stage = rd_stage + 1 == NS ? 0 : rd_stage + 1;
    aA = (a_base + so) & 0x3ffffu;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aBS[kx] = (bs_base[kx] + so) & 0x3ffffu;
  };

  // The two waves of a SIMD (w, w + 4) issue the next tile's DMA at different points of the tile so that one's MFMAs cover
  // the other's issue stall (see conv3x3_bf16.hip).
  const bool stage_late = wave >= 4;
  bf16x8 arow_a[TR], arow_b[TR], bq0[3], bq1[3];
  unsigned aA, aB[3];
  if constexpr (!SPARSE) {
    stage_addr(aA, aB);
    load_a(aA, arow_a);
    load_b(aB, 0, bq0);
  }

  // One tile: 36 MFMAs in two halves around the ring's rendezvous.  `cur` holds this tile's dout fragments (read during the
  // previous tile's second half), `nxt` receives the next tile's; halo row h + 1 is requested before row h's MFMAs.
  // The rendezvous (counted vmcnt: this wave's pieces of tile i + 1 have landed, one tile stays in flight; barrier: everybody's
  // have, and everybody is done with tile i - 1, whose stage the DMA of tile i + NS - 1 may now overwrite) sits in the MIDDLE
  // of the tile: behind it the wave still has half a tile of MFMAs whose operands are in registers or in this tile's stage, and
  // the next tile's first fragments are requested under them - no wave starts a tile with an empty matrix pipe.
  auto tile = [&](bf16x8 (&cur)[TR], bf16x8 (&nxt)[TR]) {
    if (want_bias) {
#pragma unroll
      for (int row = 0; row < TR; ++row) {
        const u32x4 d = __builtin_bit_cast(u32x4, cur[row]);
#pragma unroll
        for (int k = 0; k < 4; ++k) bsum += __builtin_bit_cast(float, d[k] << 16) + __builtin_bit_cast(float, d[k] & 0xffff0000u);
      }
    }
    load_b(aB, 1, bq1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(0, cur, bq0);
    __builtin_amdgcn_sched_barrier(0);
    load_b(aB, 2, bq0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(1, cur, bq1);
    __builtin_amdgcn_sched_barrier(0);
    load_b(aB, 3, bq1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(2, cur, bq0);
    __builtin_amdgcn_sched_barrier(0);
    load_b(aB, 4, bq0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NS - 3)) : "memory");
    __builtin_amdgcn_s_barrier();
    if (!stage_late) issue();
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(3, cur, bq1);
    __builtin_amdgcn_sched_barrier(0);
    load_b(aB, 5, bq1);
    unsigned nA, nB[3];
    stage_addr(nA, nB);
    load_a(nA, nxt);
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(4, cur, bq0);
    __builtin_amdgcn_sched_barrier(0);
    load_b(nB, 0, bq0);
    if (stage_late) issue();
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(5, cur, bq1);
    __builtin_amdgcn_sched_barrier(0);
    aA = nA;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aB[kx] = nB[kx];
  };
  // SPARSE tile: step a = 0 .. 3 reads the dense operand of halo rows (a, a + 2) at the three horizontal shifts and multiplies dout rows (0, 2) at vertical
  // tap ky = a (a <= 2) and rows (1, 3) at ky = a - 1 (a >= 1): 3 + 6 + 6 + 3 instructions; the ring's rendezvous sits between steps 1 and 2 as in the
  // dense tile; the next tile's dout rows are read and compressed, and its step-0 operands read, under steps 2 and 3.
  [[maybe_unused]] SpA spa, spb;
  [[maybe_unused]] bf16x16 sq0[3], sq1[3];
  [[maybe_unused]] unsigned aBS[3];
  auto smf = [&](const bf16x8& a, int idx, int ky, const bf16x16 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b[kx], acc[ky * 3 + kx], idx, 0, 0);
  };
  auto compress_rows = [&](const bf16x8 (&rows)[TR], SpA& sp) __attribute__((always_inline)) {
    float bs_ = 0.f;
    unsigned v[TR][2], nb[TR];
#pragma unroll
    for (int row = 0; row < TR; ++row) {
      if (want_bias) {
        const u32x4 d = __builtin_bit_cast(u32x4, rows[row]);
#pragma unroll
        for (int k = 0; k < 4; ++k) bs_ += __builtin_bit_cast(float, d[k] << 16) + __builtin_bit_cast(float, d[k] & 0xffff0000u);
      }
      compress24(rows[row], v[row][0], v[row][1], nb[row]);
    }
    sp.a02 = __builtin_bit_cast(bf16x8, u32x4{v[0][0], v[0][1], v[2][0], v[2][1]});
    sp.a13 = __builtin_bit_cast(bf16x8, u32x4{v[1][0], v[1][1], v[3][0], v[3][1]});
    sp.i02 = (int)(nb[0] | (nb[2] << 8));
    sp.i13 = (int)(nb[1] | (nb[3] << 8));
    sp.bias = bs_;
  };
  // POOLED: the same two operands from the pooled gradient and the routing codes.  A lane (channel i of its fragment, half h) needs the tile's pooled
  // pixels 4h .. 4h + 3 of both pooled rows: one transposing read each (4 values = 4 windows of the lane's channel) and the 8 windows' code words (the
  // lane's octet; lanes of an octet read the same address).  Window (pr, pc) with code 2 dy + dx puts g at dense pixel (2 pr + dy, 2 pc + dx): it is the
  // (possibly zero) survivor of pixel PAIR pc in dense row 2 pr + dy - so slot s of K block h (dense row dy) / block 2 + h (dense row 2 + dy) holds
  // dy == code.dy ? g : 0, and the pair's position bit is dx for both dense rows: rows (0, 2) and (1, 3) share ONE index word.
  const unsigned ap_base = lds0 + (4 * khalf + m) * 256 + ((wq ^ m) * 64) + cbyte;                       // + pooled row * 2048
  const unsigned cp_base = lds0 + 4096 + (4 * khalf) * 32 + (4 * wq + ((lane & 31) >> 3)) * 2;           // + (pooled row * 8 + s) * 32
  const unsigned csh = 2u * (lane & 7);
  auto load_pooled = [&](unsigned so, SpA& sp) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) const unsigned short* lds_u16;
    const unsigned ag = (ap_base + so) & 0x3ffffu, ac = (cp_base + so) & 0x3ffffu;
    const u32x2 g0 = __builtin_bit_cast(u32x2, tr_read(ag)), g1 = __builtin_bit_cast(u32x2, tr_read(ag + 2048));
    unsigned code[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) code[s] = ((unsigned)*(lds_u16)(uintptr_t)(ac + ((s >> 2) * 8 + (s & 3)) * 32) >> csh) & 3u;
    const unsigned gd[4] = {g0[0], g0[1], g1[0], g1[1]};   // slots (0, 1), (2, 3) of pooled row 0, (4, 5), (6, 7) of pooled row 1
    unsigned v02[4], v13[4], idx = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned ce = code[2 * q], co = code[2 * q + 1];
      const unsigned keep1 = ((ce & 2u) ? 0x0000ffffu : 0u) | ((co & 2u) ? 0xffff0000u : 0u);   // halves whose window routed to its LOWER row (dy = 1)
      v13[q] = gd[q] & keep1;
      v02[q] = gd[q] & ~keep1;
      idx |= ((ce & 1u) | ((2u + (co & 1u)) << 2)) << (4 * q);
    }
    sp.a02 = __builtin_bit_cast(bf16x8, u32x4{v02[0], v02[1], v02[2], v02[3]});
    sp.a13 = __builtin_bit_cast(bf16x8, u32x4{v13[0], v13[1], v13[2], v13[3]});
    sp.i02 = sp.i13 = (int)idx;
    float bs_ = 0.f;
    if (want_bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bs_ += __builtin_bit_cast(float, gd[q] << 16) + __builtin_bit_cast(float, gd[q] & 0xffff0000u);
    }
    sp.bias = bs_;
  };
#ifdef SF_EXP_WG_NOSB   // experiment: leave the interleaving of fragment reads, compression and matrix instructions to the compiler
#define SF_SPARSE_SB
#else
#define SF_SPARSE_SB __builtin_amdgcn_sched_barrier(0)
#endif
  auto tile_s = [&](SpA& cur, SpA& nxt) {
    bsum += cur.bias;
    load_b16(aBS, 1, sq1);
    SF_SPARSE_SB;
    smf(cur.a02, cur.i02, 0, sq0);
    SF_SPARSE_SB;
#ifdef SF_EXP_WG_SHIFT   // experiment: rows (a, a + 2) from rows (a - 2, a) in registers + a half-wave read
    shift_b16(aBS, 2, sq0);
#else
    load_b16(aBS, 2, sq0);
#endif
    SF_SPARSE_SB;
    smf(cur.a02, cur.i02, 1, sq1);
    smf(cur.a13, cur.i13, 0, sq1);
    SF_SPARSE_SB;
#ifdef SF_EXP_WG_SHIFT
    shift_b16(aBS, 3, sq1);
#else
    load_b16(aBS, 3, sq1);
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NS - 3)) : "memory");
#ifndef SF_EXP_WG_NOBAR   // timing experiment (results WRONG): what does the per-tile rendezvous cost?
    __builtin_amdgcn_s_barrier();
#endif
    if (!stage_late) issue();
    // the next tile's dout rows are requested NOW (its stage is visible behind the barrier) and compressed under the six instructions of step 2
    unsigned nA, nBS[3];
    const unsigned nso = (unsigned)(rd_stage * STAGE);
    stage_addr_s(nA, nBS);
    [[maybe_unused]] bf16x8 rows[TR];
    if constexpr (!POOLED) load_a(nA, rows);
    SF_SPARSE_SB;
    smf(cur.a02, cur.i02, 2, sq0);
    smf(cur.a13, cur.i13, 1, sq0);
    if constexpr (POOLED) load_pooled(nso, nxt); else compress_rows(rows, nxt);
    SF_SPARSE_SB;
    load_b16(nBS, 0, sq0);
    if (stage_late) issue();
    SF_SPARSE_SB;
    smf(cur.a13, cur.i13, 2, sq1);
    SF_SPARSE_SB;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aBS[kx] = nBS[kx];
  };
  // partial[slot][tap][co][ci], slot = ks (or ks * maxseg + segment)
  const int r = lane & 31, kh = lane >> 5;
  auto store_partial = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = (cot + cotw) * DMA_CO_T + 32 * wq + frag_row(reg, kh);
        const int ci = cit * DMA_CI_T + 32 * wc + r;
        p.partial[(((size_t)slot * 9 + tap) * p.NpT + co) * p.KpT + ci] = acc[tap][reg];
        // the tile's dead half: nobody else writes it
        if constexpr (GEO == 1) p.partial[(((size_t)slot * 9 + tap) * p.NpT + co) * p.KpT + ci + 32] = 0.f;
        if constexpr (GEO == 2) p.partial[(((size_t)slot * 9 + tap) * p.NpT + co + 64) * p.KpT + ci] = 0.f;
      }
    if (want_bias) {
      const float tot = bsum + __shfl_xor(bsum, 32);
      if (kh == 0) p.partial_db[(size_t)slot * p.NpT + (cot + cotw) * DMA_CO_T + 32 * wq + r] = tot;
    }
  };
  int seg = 0, next_flush = 0x7fffffff;
  if constexpr (GROUPED) next_flush = p.tpg - t_begin % p.tpg;  // tiles until this slice's first group boundary
  auto group_boundary = [&](int i) __attribute__((always_inline)) {  // block-uniform
    if constexpr (GROUPED) {
      if (__builtin_expect(i == next_flush, 0)) {
        store_partial(ks * p.maxseg + seg);
        ++seg; next_flush += p.tpg;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
        bsum = 0.f;
        // stores and loads retire out of order with each other: with stores pending the counted waits of the ring would no longer
        // mean "tile landed" - drain everything once (the ring refills within a tile)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  };
  if constexpr (SPARSE) {
    {
      unsigned sA;
      const unsigned so0 = (unsigned)(rd_stage * STAGE);
      stage_addr_s(sA, aBS);
      if constexpr (POOLED) load_pooled(so0, spa);
      else {
        bf16x8 rows0[TR];
        load_a(sA, rows0);
        compress_rows(rows0, spa);
      }
      load_b16(aBS, 0, sq0);
    }
    for (int i = 0; i < my_tiles; i += 2) {
      group_boundary(i);
      tile_s(spa, spb);
      if (i + 1 < my_tiles) { group_boundary(i + 1); tile_s(spb, spa); }
    }
  } else
  for (int i = 0; i < my_tiles; i += 2) {
    group_boundary(i);
    tile(arow_a, arow_b);
    if (i + 1 < my_tiles) { group_boundary(i + 1); tile(arow_b, arow_a); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the (dummy) tail may land after the block has retired
  store_partial(GROUPED ? ks * p.maxseg + seg : ks);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// POOLED sparse operand on 8-ROW K tiles (SPARSE 2 in pairs; round 5).  Timing builds and SQ counters of the kernels above say the whole family is
// bound by the bytes it pulls through the LDS-DMA path (~4 TB/s whatever the matrix work; profiles/r05_wgrad_sparse_ab.txt), and a 4 x 16 tile
// fetches a 6 x 18 halo of the input: 1.69x its pixels.  Here a stage holds TWO vertically adjacent tiles - 4 pooled rows of the gradient + codes
// (9 KB) and a 10 x 18 input halo (22.5 KB; 1.41x) - fetched by 2 + 3 DMA instructions per wave, and the two halves run the SPARSE-2 tile code on
// their rows of it: 31.5 KB instead of 36 per 128 pixels, one rendezvous per 36 matrix instructions instead of per 18.  Image heights in whole 8-row
// tiles (the plan counts tiles of 8 rows: Plan::tr); everything else - slab, waves, operand construction, partial slabs - as SPARSE 2.
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int TR8 = 8, HR8 = TR8 + 2;
constexpr int P8_A_BYTES = 10240, P8_B_BYTES = 24576, P8_STAGE = P8_A_BYTES + P8_B_BYTES, P8_NS = 4;   // [g 8 KB | codes 1 KB | dump 1 KB][halo 24 KB]
constexpr int LDS_BYTES_P8 = P8_NS * P8_STAGE;   // 136 KB
constexpr int LDS_CONST_P8 = THREADS * 32;       // + 16 KB: every lane's eight DMA constants (offsets, class masks) - in registers they pushed the kernel into scratch
                                                 // INSIDE the tile loop, and a scratch reload is a vector memory operation: it breaks the ring's counted waits
__device__ __forceinline__ void wgrad_pooled8_body(const WgradParams& p, const int per_slice, char* lds, const int ks, const int cot, const int cit) {
  constexpr int B_PIX = 2 * DMA_CI_T;
  constexpr int A_BYTES = P8_A_BYTES, STAGE = P8_STAGE, NS = P8_NS;
  constexpr int NA = 2, NBK = 3, NDMA = NA + NBK;
  constexpr unsigned SENT = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, wc = wave >> 2;
  const int t_begin = ks * per_slice;
  const int t_end = t_begin + per_slice < p.ntiles ? t_begin + per_slice : p.ntiles;
  const int my_tiles = t_end > t_begin ? t_end - t_begin : 0;   // (tiles of 8 rows)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  int nx_i = 0, nx_stage = 0, nx_tx, nx_ty, nx_n;
  {
    int t = t_begin < p.ntiles ? t_begin : 0;
    nx_tx = t % p.tiles_x; t /= p.tiles_x;
    nx_ty = t % p.tiles_y;
    nx_n = t / p.tiles_y;
  }
  const int Hp = p.H >> 1, Wp = p.W >> 1;
  const long long g_img = (long long)Hp * Wp * p.pool_s * 2, r_img = (long long)Hp * Wp * (p.dc >> 3) * 2;
  // pooled gradient: piece P = 64 wave + lane of the tile's 4 x 8 pooled pixels (pixel P / 16: row P / 128, column (P / 16) % 8), [pixel][256 B] at the stage's origin,
  // the 64-byte channel groups xor-ed with (pixel & 3); routing codes: wave 0, lane = (pooled pixel lane / 2, octets 8 (lane % 2) ..), [pixel][32 B] at + 8192
  unsigned pa_voff = SENT, pc_voff = SENT;
  {
    const int P = 64 * wave + lane, pp = P >> 4, c16 = P & 15;
    const int ch = cot * DMA_CO_T + (((c16 >> 2) ^ (pp & 3)) * 32) + (c16 & 3) * 8;
    if (ch < p.dc) pa_voff = (unsigned)((((pp >> 3) * Wp + (pp & 7)) * p.pool_s + ch) * 2);
    if (wave == 0) {
      const int pq = lane >> 1, oct = cot * (DMA_CO_T / 8) + (lane & 1) * 8;
      if (oct * 8 < p.dc) pc_voff = (unsigned)((((pq >> 3) * Wp + (pq & 7)) * (p.dc >> 3) + oct) * 2);
    }
  }
  const unsigned pc_dst = wave == 0 ? 8192u : 9216u;   // (the other waves' all-out-of-range instruction zero-fills a dump block)
  unsigned fb_off[NBK], fb_mask[NBK];
  const int bs = p.s0;
#pragma unroll
  for (int u = 0; u < NBK; ++u) {
    const int b = wave + 8 * u;
    const int hp = 8 * b + (lane >> 3);
    const int hy = hp / HW, hx = hp % HW;
    const int cb = lane & 7, pqb = cb >> 2;
    const int kc = cit * DMA_CI_T + ((pqb ^ ((hx >> 1) & 1)) * 32) + (cb & 3) * 8;
    const bool chok = hp < HR8 * HW && kc < p.c0;
    fb_off[u] = (unsigned)(((hy * p.W + hx) * bs + kc) * 2);
    fb_mask[u] = 0;
    for (int cls = 0; cls < 16; ++cls) {
      const int y0 = (cls & 2) ? (p.tiles_y - 1) * TR8 : ((cls & 1) ? 0 : TR8), x0 = (cls & 8) ? (p.tiles_x - 1) * TW : ((cls & 4) ? 0 : TW);
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      if (chok && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) fb_mask[u] |= 1u << cls;
    }
  }
  const unsigned b_pxb = 2u * (unsigned)bs;
  const long long b_img = (long long)p.H * p.W * b_pxb;
  {
    typedef __attribute__((address_space(3))) u32x4* lds_w4;
    const unsigned ca = lds0 + (unsigned)(LDS_BYTES_P8 + tid * 32);
    *(lds_w4)(uintptr_t)ca = u32x4{pa_voff, pc_voff, fb_off[0], fb_mask[0]};
    *(lds_w4)(uintptr_t)(ca + 16) = u32x4{fb_off[1], fb_mask[1], fb_off[2], fb_mask[2]};
  }
  auto pooled_image = [&](int n) {
    if (p.pool_L == 0) return n;
    const int b = n % p.pool_B, t = (n / p.pool_B) % p.pool_T, l = n / (p.pool_B * p.pool_T);
    return (t * p.pool_L + l) * p.pool_B + b;
  };
  int nx_ng = pooled_image(nx_n);
  const bool stage_late = wave >= 4;
  auto issue = [&]() {
    typedef __attribute__((address_space(3))) const u32x4* lds_c4;   // (an LDS read, ordered by lgkmcnt: never a vector-memory operation)
    // (the lane id is taken again from the exec-mask count: two VALU operations the compiler can repeat anywhere instead of keeping - or spilling - a register)
    const unsigned lane_now = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned ca = lds0 + (unsigned)LDS_BYTES_P8 + ((unsigned)wave * 64u + lane_now) * 32u;
    const u32x4 c0 = *(lds_c4)(uintptr_t)ca, c1 = *(lds_c4)(uintptr_t)(ca + 16);
    const unsigned pa_voff = c0[0], pc_voff = c0[1];
    const unsigned fb_off[NBK] = {c0[2], c1[0], c1[2]}, fb_mask[NBK] = {c0[3], c1[1], c1[3]};
    const bool live = nx_i < my_tiles;
    const int n = nx_n, px0 = (nx_ty * TR8) * p.W + nx_tx * TW;
    const unsigned cls = (nx_ty == 0 ? 1u : 0u) | (nx_ty == p.tiles_y - 1 ? 2u : 0u) | (nx_tx == 0 ? 4u : 0u) | (nx_tx == p.tiles_x - 1 ? 8u : 0u);
    const unsigned sel = live ? 1u << cls : 0u;
    const unsigned stage = lds0 + (unsigned)(nx_stage * STAGE);
    const unsigned pix0 = (unsigned)((nx_ty * (TR8 / 2)) * Wp + nx_tx * (TW / 2));
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.pool_g + nx_ng * g_img), 0, __builtin_amdgcn_readfirstlane((int)g_img), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.pool_route + n * r_img), 0, __builtin_amdgcn_readfirstlane((int)r_img), 0x00020000);
    bufdma16(live ? pa_voff : SENT, rsg, pix0 * (unsigned)(p.pool_s * 2), stage + wave * 1024);
    bufdma16(live ? pc_voff : SENT, rsr, pix0 * (unsigned)((p.dc >> 3) * 2), stage + pc_dst);
    const long long lead = (long long)(p.W + 1) * b_pxb;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.src0 + n * b_img - lead), 0, __builtin_amdgcn_readfirstlane((int)(b_img + 2 * lead)), 0x00020000);
    const unsigned sob = (unsigned)px0 * b_pxb;
#pragma unroll
    for (int u = 0; u < NBK; ++u) bufdma16((fb_mask[u] & sel) ? fb_off[u] : SENT, rsb, sob, stage + A_BYTES + (wave + 8 * u) * 1024);
    ++nx_i;
    nx_stage = nx_stage + 1 == NS ? 0 : nx_stage + 1;
    if (++nx_tx == p.tiles_x) { nx_tx = 0; if (++nx_ty == p.tiles_y) { nx_ty = 0; ++nx_n; nx_ng = pooled_image(nx_n); } }
  };

  // ---- per-lane LDS read addresses (half 1 of a stage: + 4096 in the pooled gradient, + 512 in the codes, + 4 halo rows) ----
  const int m = (lane >> 2) & 3, khalf = lane >> 5;
  const int cbyte = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  unsigned bs_base[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
    bs_base[kx] = lds0 + A_BYTES + (m + kx) * B_PIX + ((wc ^ (((m + kx) >> 1) & 1)) * 64) + cbyte + khalf * (2 * HW * B_PIX);
  const unsigned ap_base = lds0 + (4 * khalf + m) * 256 + ((wq ^ m) * 64) + cbyte;
  const unsigned cp_base = lds0 + 8192 + (4 * khalf) * 32 + (4 * wq + ((lane & 31) >> 3)) * 2;
  const unsigned csh = 2u * (lane & 7);
  constexpr unsigned HALF_A = 4096, HALF_C = 512, HALF_B = 4 * HW * B_PIX;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  const bool want_bias = cit == 0 && wc == 0;

#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NS - 2)) : "memory");
  __builtin_amdgcn_s_barrier();

  struct SpA { bf16x8 a02, a13; int idx; float bias; };
  auto load_pooled = [&](unsigned so, unsigned half, SpA& sp) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) const unsigned short* lds_u16;
    const unsigned ag = (ap_base + so + half * HALF_A) & 0x3ffffu, ac = (cp_base + so + half * HALF_C) & 0x3ffffu;
    const u32x2 g0 = __builtin_bit_cast(u32x2, tr_read(ag)), g1 = __builtin_bit_cast(u32x2, tr_read(ag + 2048));
    unsigned code[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) code[s] = ((unsigned)*(lds_u16)(uintptr_t)(ac + ((s >> 2) * 8 + (s & 3)) * 32) >> csh) & 3u;
    const unsigned gd[4] = {g0[0], g0[1], g1[0], g1[1]};
    unsigned v02[4], v13[4], idx = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned ce = code[2 * q], co = code[2 * q + 1];
      const unsigned keep1 = ((ce & 2u) ? 0x0000ffffu : 0u) | ((co & 2u) ? 0xffff0000u : 0u);
      v13[q] = gd[q] & keep1;
      v02[q] = gd[q] & ~keep1;
      idx |= ((ce & 1u) | ((2u + (co & 1u)) << 2)) << (4 * q);
    }
    sp.a02 = __builtin_bit_cast(bf16x8, u32x4{v02[0], v02[1], v02[2], v02[3]});
    sp.a13 = __builtin_bit_cast(bf16x8, u32x4{v13[0], v13[1], v13[2], v13[3]});
    sp.idx = (int)idx;
    float bs_ = 0.f;
    if (want_bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bs_ += __builtin_bit_cast(float, gd[q] << 16) + __builtin_bit_cast(float, gd[q] & 0xffff0000u);
    }
    sp.bias = bs_;
  };
  auto load_b16 = [&](const unsigned (&aBS)[3], int a, bf16x16 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned ad = aBS[kx] + a * (HW * B_PIX);
      const bf16x8 lo = cat8(tr_read(ad), tr_read(ad + 4 * B_PIX)), hi = cat8(tr_read(ad + 8 * B_PIX), tr_read(ad + 12 * B_PIX));
      b[kx] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    }
  };
  auto smf = [&](const bf16x8& a, int idx, int ky, const bf16x16 (&b)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b[kx], acc[ky * 3 + kx], idx, 0, 0);
  };
  SpA spa, spb;
  bf16x16 sq0[3], sq1[3];
  unsigned aBS[3];      // fragment addresses of the half being multiplied
  unsigned cur_so = 0;  // its stage
  int rd_stage = 0;
  auto half_addr = [&](unsigned so, unsigned half, unsigned (&a)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) a[kx] = (bs_base[kx] + so + half * HALF_B) & 0x3ffffu;
  };
  // one half (4 rows) of a stage; HALF 0 hands over to half 1 of the same stage, HALF 1 holds the ring's rendezvous and hands over to the next stage
  auto half_s = [&](auto HALF, SpA& cur, SpA& nxt) __attribute__((always_inline)) {
    constexpr int H = decltype(HALF)::value;
    bsum += cur.bias;
    load_b16(aBS, 1, sq1);
    __builtin_amdgcn_sched_barrier(0);
    smf(cur.a02, cur.idx, 0, sq0);
    __builtin_amdgcn_sched_barrier(0);
    load_b16(aBS, 2, sq0);
    __builtin_amdgcn_sched_barrier(0);
    smf(cur.a02, cur.idx, 1, sq1);
    smf(cur.a13, cur.idx, 0, sq1);
    __builtin_amdgcn_sched_barrier(0);
    load_b16(aBS, 3, sq1);
    unsigned nso;
    if constexpr (H == 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NS - 3)) : "memory");
      __builtin_amdgcn_s_barrier();
      if (!stage_late) issue();
      rd_stage = rd_stage + 1 == NS ? 0 : rd_stage + 1;
      nso = (unsigned)(rd_stage * STAGE);
    } else nso = cur_so;
    unsigned nBS[3];
    half_addr(nso, H == 1 ? 0u : 1u, nBS);
    __builtin_amdgcn_sched_barrier(0);
    smf(cur.a02, cur.idx, 2, sq0);
    smf(cur.a13, cur.idx, 1, sq0);
    load_pooled(nso, H == 1 ? 0u : 1u, nxt);
    __builtin_amdgcn_sched_barrier(0);
    load_b16(nBS, 0, sq0);
    if constexpr (H == 1) { if (stage_late) issue(); }
    __builtin_amdgcn_sched_barrier(0);
    smf(cur.a13, cur.idx, 2, sq1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aBS[kx] = nBS[kx];
    cur_so = nso;
  };
  const int r = lane & 31, kh = lane >> 5;
  auto store_partial = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = cot * DMA_CO_T + 32 * wq + frag_row(reg, kh);
        const int ci = cit * DMA_CI_T + 32 * wc + r;
        p.partial[(((size_t)slot * 9 + tap) * p.NpT + co) * p.KpT + ci] = acc[tap][reg];
      }
    if (want_bias) {
      const float tot = bsum + __shfl_xor(bsum, 32);
      if (kh == 0) p.partial_db[(size_t)slot * p.NpT + cot * DMA_CO_T + 32 * wq + r] = tot;
    }
  };
  int seg = 0, next_flush = p.tpg - t_begin % p.tpg;
  auto group_boundary = [&](int i) __attribute__((always_inline)) {
    if (__builtin_expect(i == next_flush, 0)) {
      store_partial(ks * p.maxseg + seg);
      ++seg; next_flush += p.tpg;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
      bsum = 0.f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  half_addr(0u, 0u, aBS);
  load_pooled(0u, 0u, spa);
  load_b16(aBS, 0, sq0);
  for (int i = 0; i < my_tiles; ++i) {
    group_boundary(i);
    half_s(std::integral_constant<int, 0>{}, spa, spb);
    half_s(std::integral_constant<int, 1>{}, spb, spa);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_partial(ks * p.maxseg + seg);
}

__global__ __launch_bounds__(THREADS, 2) void wgrad_pooled8_kernel(const WgradParams p, const int per_slice, const int xcd_groups, const int cot_n) {
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES_P8 + LDS_CONST_P8];
  const int units = gridDim.y;
  int ks, unit;
  {
    const int id = blockIdx.x + gridDim.x * blockIdx.y;
    if (xcd_groups) {
      const int x = id % 8, j = id / 8;
      ks = (j / units) * 8 + x;
      unit = j % units;
    } else { ks = blockIdx.x; unit = blockIdx.y; }
  }
  wgrad_pooled8_body(p, per_slice, lds, ks, unit % cot_n, unit / cot_n);
}

// Workgroup -> (K slice, slab).  Grid = (KS, units); `units` = the regular (co tile, ci tile) slabs followed by the edge slabs (edge_mode 1: one GEO-1
// slab per PAIR of co tiles of the last ci tile; edge_mode 2: one GEO-2 slab per PAIR of ci tiles of the last co tile, a leftover ci tile stays
// regular).  The slabs of one K slice are consecutive on one XCD (blocks are dealt round-robin to the 8 XCDs), so a slice's tiles are shared
// through that XCD's L2 (pure speed choice).
template <bool FAST, bool GROUPED = false, int SPARSE = 0>
__global__ __launch_bounds__(THREADS, 2) void wgrad_bf16_dma_kernel(const WgradParams p, const int per_slice, const char* __restrict__ zero,
                                                                     const int xcd_groups, const int cot_n, const int cit_n, const int edge_mode) {
  __shared__ __attribute__((aligned(1024))) char lds[SPARSE == 2 ? LDS_BYTES_POOLED : LDS_BYTES];
  const int units = gridDim.y;
  int ks, unit;
  {
    const int id = blockIdx.x + gridDim.x * blockIdx.y;
    if (xcd_groups) {
      const int x = id % 8, j = id / 8;
      ks = (j / units) * 8 + x;
      unit = j % units;
    } else { ks = blockIdx.x; unit = blockIdx.y; }
  }
  int geo = 0, cot = unit % cot_n, cit = unit / cot_n;  // (block-uniform)
  if (FAST && edge_mode == 1) {
    const int regular = cot_n * (cit_n - 1);
    if (unit >= regular) { geo = 1; cot = 2 * (unit - regular); cit = cit_n - 1; }
  } else if (FAST && edge_mode == 2) {
    const int regular = (cot_n - 1) * cit_n, pairs = cit_n / 2;
    if (unit < regular) { cot = unit % (cot_n - 1); cit = unit / (cot_n - 1); }
    else if (unit < regular + pairs) { geo = 2; cot = cot_n - 1; cit = 2 * (unit - regular); }
    else { cot = cot_n - 1; cit = cit_n - 1; }  // the odd ci tile out
  }
  if constexpr (FAST) {
    if (geo == 1) { wgrad_dma_body<true, GROUPED, 1>(p, per_slice, zero, lds, ks, cot, cit); return; }
    if (geo == 2) { wgrad_dma_body<true, GROUPED, 2>(p, per_slice, zero, lds, ks, cot, cit); return; }
  }
  wgrad_dma_body<FAST, GROUPED, 0, SPARSE>(p, per_slice, zero, lds, ks, cot, cit);
}

}  // namespace

sfwgrad::Plan sf_wgrad_bf16_dma_plan(int Np, int Kp, int n, int h, int w, int groups, int single_source, int tr) {
  using namespace sfwgrad;
  Plan pl;
  pl.tr = tr == TR8 ? TR8 : TR;   // K-tile height: 4 rows; 8 for the pooled sparse operand on tile pairs (wgrad_pooled8_body)
  pl.tiles_x = (w + TW - 1) / TW;
  pl.tiles_y = (h + pl.tr - 1) / pl.tr;
  pl.ntiles = pl.tiles_x * pl.tiles_y * n;
  pl.cot = (Np + DMA_CO_T - 1) / DMA_CO_T;
  pl.cit = (Kp + DMA_CI_T - 1) / DMA_CI_T;
  // the last ci tile holds at most 32 channels, the co tiles pair up, one input tensor: wide slabs for that tile (see wgrad_dma_body)
  static const bool no_wide = getenv("SF_NO_WGRAD_WIDE") != nullptr;  // A/B switch
  const int tail = Kp - (pl.cit - 1) * DMA_CI_T, ctail = Np - (pl.cot - 1) * DMA_CO_T;
  pl.wide_pairs = 0; pl.edge_mode = 0;
  pl.units = pl.cot * pl.cit;
  if (!no_wide && single_source) {  // 1: one input tensor; 2: two, the first a whole number of ci tiles wide (every tile has one source)
    if (tail > 0 && tail <= 32 && pl.cot % 2 == 0) {            // GEO 1: half-empty last ci tile
      pl.edge_mode = 1; pl.wide_pairs = pl.cot / 2;
      pl.units = pl.cot * (pl.cit - 1) + pl.wide_pairs;
    } else if (single_source == 1 && ctail > 0 && ctail <= 64 && pl.cit >= 2) {  // GEO 2: half-empty last co tile (a pair of ci tiles = one source)
      pl.edge_mode = 2; pl.wide_pairs = pl.cit / 2;
      pl.units = (pl.cot - 1) * pl.cit + pl.wide_pairs + (pl.cit & 1);
    }
  }
  // one workgroup per CU (128 KiB of LDS): as many K slices as fill the 256 CUs once, in whole groups of 8 (one per XCD)
  int want = 256 / pl.units;
  want = want / 8 * 8;
  // (a weight of thousands of slabs fills the chip unsplit: no 8 partial copies of a 300 MB gradient - the floor of make_plan, wgrad_common.h)
  int floor_ks = 2048 / pl.units;
  floor_ks = groups > 0 ? 8 : floor_ks < 1 ? 1 : floor_ks > 8 ? 8 : floor_ks;
  if (want < floor_ks) want = floor_ks;
  const int per_min = 6;  // a slice shorter than the ring + pipeline fill is all prologue
  int ks = want;
  while (ks > 8 && (pl.ntiles + ks - 1) / ks < per_min) ks -= 8;
  if (ks > pl.ntiles) ks = pl.ntiles > 0 ? pl.ntiles : 1;
  pl.KS = ks;
  pl.tpg = 0; pl.maxseg = 1;
  if (groups > 0) {  // segments a slice can need: the groups it touches
    pl.tpg = pl.ntiles / groups;
    const int per = (pl.ntiles + ks - 1) / ks;
    for (int k = 0; k < ks && pl.tpg > 0; ++k) {
      const int b = k * per, e = (b + per < pl.ntiles ? b + per : pl.ntiles) - 1;
      if (e >= b && e / pl.tpg - b / pl.tpg + 1 > pl.maxseg) pl.maxseg = e / pl.tpg - b / pl.tpg + 1;
    }
  }
  pl.ws_floats = (size_t)pl.KS * pl.maxseg * ((size_t)9 * pl.cot * DMA_CO_T * pl.cit * DMA_CI_T + (size_t)pl.cot * DMA_CO_T) + 64;  // + the zero page
  return pl;
}

int sf_launch_wgrad_bf16_dma(sfwgrad::WgradParams& p, const sfwgrad::Plan& pl, float* workspace, hipStream_t st) {
  using namespace sfwgrad;
  if ((((uintptr_t)p.dout) & 15) || p.ds % 8 || p.dc % 8 || (p.src0 && ((((uintptr_t)p.src0) & 15) || p.s0 % 8 || p.c0 % 8)) ||
      (p.src1 && ((((uintptr_t)p.src1) & 15) || p.s1 % 8 || p.c1 % 8))) {
    sf_set_error("wgrad_bf16_dma: bf16 tensors need 16-byte aligned pixels (strides and channel counts multiples of 8)");
    return 1;
  }
  {
    const long long px = (long long)p.H * p.W + p.W + 1;
    const long long smax = p.ds > p.s0 ? (p.ds > p.s1 ? p.ds : p.s1) : (p.s0 > p.s1 ? p.s0 : p.s1);
    if (px * smax * 2 >= (1ll << 31)) { sf_set_error("wgrad_bf16_dma: one image of a tensor must be smaller than 2 GiB"); return 1; }
  }
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.KS = pl.KS;
  p.NpT = pl.cot * DMA_CO_T; p.KpT = pl.cit * DMA_CI_T;
  // workspace: [zero page 64 floats][partial][partial_db]
  if (sf_fill_async(workspace, 0, 64 * sizeof(float), st) != hipSuccess) { sf_set_error("wgrad_bf16_dma: memset failed"); return 2; }
  p.partial = workspace + 64;
  p.partial_db = p.partial + (size_t)pl.KS * pl.maxseg * 9 * p.NpT * p.KpT;
  p.tpg = pl.tpg; p.maxseg = pl.maxseg;
  const int per_slice = (pl.ntiles + pl.KS - 1) / pl.KS;
  const int xcd_groups = (pl.KS % 8 == 0) ? 1 : 0;
  const bool fast = !(p.src0 && p.src1 && p.c1 > 0) || p.c0 % DMA_CI_T == 0;
  if (pl.wide_pairs && (!fast || (pl.edge_mode == 2 && p.src0 && p.src1 && p.c1 > 0))) { sf_set_error("wgrad_bf16_dma: the plan's edge slabs need tile-aligned sources"); return 1; }
  const dim3 grid(pl.KS, pl.units);
  if (p.sparse24 && (!fast || pl.edge_mode != 0 || (p.W & 1))) { sf_set_error("wgrad_bf16_dma: the 2:4-sparse path takes regular slabs of a single-source launch with an even image width"); return 1; }
  if (pl.tpg > 0) {
    if (!fast) { sf_set_error("wgrad_bf16_dma: grouped slices need a single input source"); return 1; }
    if (p.sparse24 == 2) {
      if (!p.pool_g || !p.pool_route || (p.H % TR) || (p.W % TW) || (p.dc % 8) || (p.pool_s % 8) || (p.pool_L > 0 && (p.pool_T <= 0 || p.pool_B <= 0 || p.N != p.pool_L * p.pool_T * p.pool_B)) ||
          (long long)(p.H / 2) * (p.W / 2) * p.pool_s * 2 >= (1ll << 31)) {
        sf_set_error("wgrad_bf16_dma: the pooled sparse operand needs the pooled gradient + routing record, whole 4 x 16 tiles and a permutation that covers n");
        return 1;
      }
      if (pl.tr == TR8) {
        if (p.H % TR8) { sf_set_error("wgrad_bf16_dma: the 8-row plan needs image heights in whole 8-row tiles"); return 1; }
        hipLaunchKernelGGL(wgrad_pooled8_kernel, grid, dim3(THREADS), 0, st, p, per_slice, xcd_groups, pl.cot);
      } else
      hipLaunchKernelGGL((wgrad_bf16_dma_kernel<true, true, 2>), grid, dim3(THREADS), 0, st, p, per_slice, (const char*)workspace, xcd_groups, pl.cot, pl.cit, pl.edge_mode);
    } else if (p.sparse24) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<true, true, 1>), grid, dim3(THREADS), 0, st, p, per_slice, (const char*)workspace, xcd_groups, pl.cot, pl.cit, pl.edge_mode);
    else hipLaunchKernelGGL((wgrad_bf16_dma_kernel<true, true>), grid, dim3(THREADS), 0, st, p, per_slice, (const char*)workspace, xcd_groups, pl.cot, pl.cit, pl.edge_mode);
  } else if (p.sparse24) { sf_set_error("wgrad_bf16_dma: the 2:4-sparse path is built for the grouped (folded BatchNorm) launches"); return 1; }
  else if (fast) hipLaunchKernelGGL(wgrad_bf16_dma_kernel<true>, grid, dim3(THREADS), 0, st, p, per_slice, (const char*)workspace, xcd_groups, pl.cot, pl.cit, pl.edge_mode);
  else hipLaunchKernelGGL(wgrad_bf16_dma_kernel<false>, grid, dim3(THREADS), 0, st, p, per_slice, (const char*)workspace, xcd_groups, pl.cot, pl.cit, 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error("wgrad_bf16_dma: launch failed: %s", hipGetErrorString(e)); return 2; }
  return 0;
}
