// ONE-WAVE-PER-SIMD persistent bf16 3x3 convolution (round 4): the MetNet encoder's large launches (conv2 / conv3 / conv4 forward,
// 256 -> 256 @32x32 x 2304 images and alike) on four 512-register waves per CU instead of eight 256-register ones.
//
// Why: in the 8-wave kernels every wave owns 2 pixel x 4 channel fragments (8 MFMAs per 6 fragment
// reads) and the two waves of a SIMD run the same mixed stream - MFMAs, LDS reads, LDS-DMA issue, chunk barrier - so whenever both
// are not issuing MFMAs the matrix pipe idles (measured: pipe busy ~61 %), and between two items nothing covers the epilogue.  Here
//   * a wave owns 4 pixel x 4 channel fragments (256 accumulator registers): 16 MFMAs per 8 fragment reads, 0.5 reads per MFMA
//     instead of 0.75, and the single wave of a SIMD issues its MFMAs back to back with at most a few other instructions per gap;
//   * NOTHING in the K loop waits: the weights stream through a THREE-stage LDS ring (the DMA of chunk g + 2 is issued in the
//     middle of chunk g, behind the only barrier of the chunk, which is also where every wave has waited - counted `vmcnt` - for
//     its own pieces of chunk g + 1); the input halo rows are PRIVATE to the wave that reads them (10 rows of 18 pixels, two
//     stages; 2 of 10 rows are fetched twice) so they need no barrier at all, only the issuing wave's own counted wait; the first
//     tap's fragments of chunk g + 1 are read during the last tap of chunk g;
//   * the EPILOGUE OF ITEM k RUNS INSIDE THE FIRST TAP OF ITEM k + 1: fragment by fragment the accumulators are rounded, paired
//     (`v_permlane32_swap`) and stored as 16-byte channel octets - through a buffer descriptor with out-of-range offsets for
//     pixels / channels outside the tensor, so that every wave issues exactly 32 stores per item and the counted waits stay exact -
//     and the fragment's first MFMA of the next item is issued right behind them with the folded-BatchNorm bias (from an LDS copy
//     of the 9-class table, DMA-fetched two chunks ahead) as its C operand.  The stores drain under the following taps' MFMAs;
//     no `vmcnt(0)` anywhere between the prologue and the last item.
// Same arithmetic as conv3x3_bf16_kernel<8, 4, EPI_LINEAR, false, true>: same products, same K order (chunks ascending, taps
// ascending), accumulators started at the border-class bias: results are bit-identical (tests/test_conv_persist_gpu.py).
// Scope: NF = 4 (128-channel N blocks), at least 3 K chunks, ONE bf16-stored source without image remap, bf16-stored output, linear
// epilogue with no per-channel bias (plain, or grouped weights + border-class bias table = the folded BatchNorm).
// Three instantiations: MODE 0 as above; MODE 1 adds the BatchNorm statistics of the stored values (per-tile sums, reduce-scatter over the half-waves,
// partials through each wave's free private input stage); MODE 2 is the input gradient with the BatchNorm BACKWARD in the epilogue
// (out = A * acc + B * x + K: x read in the store's 64-byte runs with hand-counted waits, coefficients in the bias table's LDS slot).
// MODE 3 (round 5) is conv4 of the MetNet encoder WITH THE 2x2 MAX-POOLING BEHIND IT IN THE EPILOGUE (sf_conv3x3_fwd_folded_pool).  The lane <-> pixel
// map of the MFMA's pixel operand is WINDOW-MAJOR there: lane r of fragment mf = 2 dy + dx holds pixel (2 wy + dy, 2 wx + dx) of the wave's 8 x 16 band
// (wy = r >> 3, wx = r & 7), so the four pixels of a pooling window are four REGISTERS of one lane: the maximum and the 2-bit routing code
// (sf_maxpool2_route_fwd's: first maximum of the STORED bf16 values in row-major order) are in-lane arithmetic, no value changes lanes, and an item
// stores 8 + 2 instead of 32 KiB-sized pieces per wave.  The same map shares pixel fragments between taps twice over - fragment (dy, dx) of tap
// (ky, kx) is rows 2 wy + dy + ky, columns 2 wx + dx + kx: 16 distinct fragments per chunk instead of 27 - with the private halo rows stored
// even columns first, odd columns second (the DMA's lane-linear LDS order is free: a fragment then reads 8 consecutive 32-byte pixels per window row).
// (The item switch is: epilogue of item k - 32 stores per wave -, accumulators of item k + 1 from the LDS table, first tap; deferring half of the stores
// under the next item's K loop was built and measured: no gain, see the note in front of `epilogue`.)
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace sfconv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int HALO_W = TILE_W + 2;  // 18
constexpr int PIX_B = 32;
constexpr unsigned DMA_SENT = 0x80000000u;  // >= any descriptor's num_records
// Cache policy of the epilogue's streams (the stored result, MODE 2's read of x): nobody reads them again inside the launch, while every 128-byte line of the
// INPUT is needed by four consecutive chunks, by the tile's other N block and by the neighbouring waves' halos - `nt` (aux bit 1) keeps the streams from
// pushing those lines out of the XCD's 4 MiB L2.  -DSF_W4_NT=0: the A/B build.
#ifndef SF_W4_NT
#define SF_W4_NT 1
#endif
constexpr int ST_AUX = SF_W4_NT >= 2 ? 2 : 0;   // (1: the read of x only)

// LDS-DMA hidden from hipcc (conv3x3_bf16.hip explains why).  M0 is declared clobbered instead of being saved and restored around every
// piece: this kernel issues ~15 pieces per 144 MFMAs from the only wave of its SIMD, every scalar instruction next to them counts.
__device__ __forceinline__ void bufdma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_dst) {
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               : : "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff) : "memory");  // (M0 is not declared: hipcc treats it as reserved and never keeps a value in it across statements here)
}
// The K loop's pieces: M0 is written ONCE per group of up to four pieces whose LDS destinations are 1 KiB apart, the piece inside the group is the
// instruction's immediate offset (added to the LDS address AND to the memory address: the caller's offsets are built for that).  5 instead of 15 M0 writes +
// wait states per chunk: 6440 -> 6390 shader cycles per chunk, bit-identical results (round 6, profiles/r06_w4_m0.txt - which also records the trap on the way: a
// timing build with NO M0 writes ran 11 % faster, but because every piece then lands on one LDS address and the matrix operands never change: less power,
// more clock).  M0 is not written by anything else between the pieces of a group (checked on the ISA).
__device__ __forceinline__ void dma_group(unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
template <int IMM>
__device__ __forceinline__ void bufdma16_imm(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" : : "v"(voff), "s"(rs), "s"(soff), "i"(IMM) : "memory");
}
__device__ __forceinline__ void bufdma16_q(int q, unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff) {  // q: compile-time after unrolling
  if (q == 0) bufdma16_imm<0>(voff, rs, soff); else if (q == 1) bufdma16_imm<1024>(voff, rs, soff);
  else if (q == 2) bufdma16_imm<2048>(voff, rs, soff); else bufdma16_imm<3072>(voff, rs, soff);
}
__device__ __forceinline__ void* uniform_ptr(const void* q) {  // inline asm "s" operands are not legalised
  const uintptr_t v = (uintptr_t)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (void*)(((uintptr_t)hi << 32) | lo);
}
__device__ __forceinline__ unsigned pk2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// One accumulator element AGPR -> VGPR.  As an asm operand constrained to "a" the accumulators stay in the AGPR half through the
// loop-header phis; read by plain VALU code hipcc gives the PHIS the VGPR class and copies all 256 accumulators across at every
// loop header (264 v_accvgpr_read + 75 spilled registers, scratch reloads - and their vmcnt(0) - inside the chunk loop).
__device__ __forceinline__ float acc_read(float x) {
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(x));
  return v;
}
#define SF_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// MODE 3's pooling arithmetic: a bare v_max_f32, and `w = 2 w + bit` with the bit taken per lane from a wave mask (v_addc_co_u32 with the mask as carry-in)
__device__ __forceinline__ float vmax(float a, float b) {
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ unsigned shift_in(unsigned w, unsigned long long mask) {
  unsigned o;
  asm("v_addc_co_u32 %0, vcc, %1, %1, %2" : "=v"(o) : "v"(w), "s"(mask) : "vcc");
  return o;
}
// A 16-byte buffer load hipcc does not see (MODE 2: the BatchNorm input read in the epilogue).  Issued through the builtin, hipcc would have to drain
// `vmcnt` to 0 at every use: the epilogue's stores share the counter and LLVM does not assume loads and stores retire in order (they do:
// tools/ubench/vmcnt_order.hip).  The wait is counted by hand and TIED to the loaded registers, so no use can move above it.
__device__ __forceinline__ u32x4_t bufload16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff) {
  u32x4_t v;
#if SF_W4_NT
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=&v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
#else
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
#endif
  return v;
}
template <int N>
__device__ __forceinline__ void wait_loaded4(u32x4_t& a, u32x4_t& b, u32x4_t& c, u32x4_t& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N) : "memory");
}
// One level of a reduce-scatter over the 32 lanes of a half-wave: 2 * M values per lane in, M out - the lane keeps the half its bit M selects and adds
// the partner's (lane ^ M) copy of that half.  After the levels 16, 8, 4, 2, 1 lane r holds entry r of the 32, summed over the 32 lanes.
template <int M>
__device__ __forceinline__ void reduce_scatter_level(float (&L)[32], int r) {
  const bool bit = (r & M) != 0;
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const float keep = bit ? L[M + k] : L[k], send = bit ? L[k] : L[M + k];
    L[k] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x1F | (M << 10)));
  }
}

constexpr int NF = 4, MFR = 4, NB = 32 * NF;         // fragments per wave: 4 pixel x 4 channel; 128-channel N block
constexpr int W_B = 9 * NB * PIX_B;                   // 36864: one weight chunk
constexpr int WST = 3;                                // weight ring stages
constexpr int PROWS = 10;                             // private halo rows per wave: 8 + 2
constexpr int PIN_B = PROWS * HALO_W * PIX_B;         // 5760 per wave and stage
constexpr int PPIECES = PROWS * HALO_W * 2;           // 360 16-byte pieces
constexpr int NPJ = (PPIECES + 63) / 64;              // 6 DMA instructions per wave and chunk
constexpr int TAB_B = 10 * NB * 4;                    // border-class bias table: 9 classes (+ 1 row of padding) x 128 fp32
constexpr int IN0 = WST * W_B, TAB0 = IN0 + 4 * 2 * PIN_B;

// WIN: window-major lane <-> pixel map of the pixel fragments (always with MODE 3; for MODE 1 a launch-time choice, SF_CONV_W4_WIN): 16 instead of
// 27 pixel-fragment reads per chunk, same products in the same order, the same 64-byte store runs (a store register then covers two window rows).
template <int MODE, bool WIN = (MODE == 3)>
__global__ __launch_bounds__(256, 1) void conv3x3_bf16_persist4_kernel(const ConvParams p, const int items, const int nblk, const int pair_order) {
  __shared__ __attribute__((aligned(1024))) char lds[TAB0 + TAB_B];  // 110592 + 46080 + 5120 = 161792 of 163840

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int nch = p.c0 / KC;
  const int tiles_total = items / nblk;
  const bool xcd_order = tiles_total % 8 == 0 && nblk > 1;  // the N blocks of a tile on ONE XCD (conv3x3_bf16.hip)
  const int grid = gridDim.x;
  // items of this workgroup (the launcher keeps grid <= items).  pair_order (SF_CONV_W4_PAIR=1, an A/B switch): a workgroup takes ALL N blocks of a
  // tile back to back (the second pass over the tile's input comes out of the L2 it has just been through) instead of one N block of it
  const int K = pair_order ? nblk * ((tiles_total - (int)blockIdx.x + grid - 1) / grid) : (items - (int)blockIdx.x + grid - 1) / grid;

  // ---- work items: a handful of scalars (everything else is derived where it is used) + the per-lane input offsets ----
  struct Item {
    int valid, n, grp, nb, x0, y0;
    unsigned in_mask;  // per lane: bit j = input piece j of this lane lies inside the image
  };
  // LDS slot of a private halo row -> halo column.  MODE 3 stores a row's even columns first, then its odd columns (window-major fragments read 8
  // consecutive pixels per window row), and swizzles the 16-byte halves by the parity of the window row instead of the pixel row.
  auto slot_col = [](int s) __attribute__((always_inline)) { return WIN ? (s < 9 ? 2 * s : 2 * (s - 9) + 1) : s; };
  auto half_flip = [](int iy) __attribute__((always_inline)) { return WIN ? (iy >> 1) & 1 : iy & 1; };
  // per-lane byte offsets of the six input pieces relative to the wave's halo origin (the same for every item; validity is the item's mask)
  unsigned in_base_off[NPJ];
#pragma unroll
  for (int j = 0; j < NPJ; ++j) {
    const int pc = lane + j * 64, pix = pc >> 1;
    const int iy = pix / HALO_W, ix = slot_col(pix - iy * HALO_W);
    const int half = (pc & 1) ^ half_flip(iy);  // the DMA writes lane-linearly: physical half pc & 1 holds the logical half (bank swizzle)
    // (minus the piece's immediate offset inside its M0 group, see dma_group: a piece j >= 1 starts in halo row >= j, i.e. at least j rows of W * s0 * 2 >= 1024
    // bytes in - the launcher's contract)
    in_base_off[j] = (unsigned)(((iy * p.W + ix) * p.s0 + 8 * half) * 2) - (unsigned)((j & 3) * 1024);
  }
  auto setup = [&](int k, Item& it) __attribute__((always_inline)) {
    it.valid = k < K;
    const int w = it.valid ? (int)blockIdx.x + k * grid : (int)blockIdx.x;  // past the end: some valid item (its descriptors get 0 records)
    int tile;
    if (pair_order) { const int kk = it.valid ? k : 0; tile = (int)blockIdx.x + (kk / nblk) * grid; it.nb = kk % nblk; }
    else if (xcd_order) { const int xcd = w & 7, j = w >> 3; tile = (j / nblk) * 8 + xcd; it.nb = j % nblk; }
    else { tile = w % tiles_total; it.nb = w / tiles_total; }
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    it.n = tile / p.tiles_y;
    it.grp = p.wgroup ? it.n / p.wgroup : 0;
    it.x0 = tx * TILE_W; it.y0 = ty * 32;
    int l = lane;
    asm volatile("" : "+v"(l));  // recomputed per call: hoisted out of the loops these per-lane pixel coordinates are 18 live-through (spilled) registers
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
      const int pc = l + j * 64, pix = pc >> 1;
      const int iy = pix / HALO_W, ix = slot_col(pix - iy * HALO_W);
      const int gy = it.y0 + 8 * wave + iy - 1, gx = it.x0 + ix - 1;
      const bool ok = pc < PPIECES && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      m |= ok ? 1u << j : 0u;
    }
    it.in_mask = m;
  };
  const int pxb = 2 * p.s0;
  const int in_bytes = __builtin_amdgcn_readfirstlane((int)((long long)pxb * ((long long)p.H * p.W + 2 * (p.W + 1))));
  const int w_bytes = __builtin_amdgcn_readfirstlane(p.chunks_total * W_B);
  const int tab_bytes = __builtin_amdgcn_readfirstlane(p.bias_tab ? 9 * p.np * 4 : 0);
  const int out_bytes = __builtin_amdgcn_readfirstlane(p.H * p.W * p.out_s * 2);
  const long long img_in = (long long)p.H * p.W * pxb, lead_in = (long long)(p.W + 1) * pxb;

  // descriptors (wave-uniform by construction; `valid` = 0 gives 0 records: every piece is zero-filled, nothing is read)
  auto rs_input = [&](const Item& it) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.src0 + it.n * img_in - lead_in), 0, __builtin_amdgcn_readfirstlane(it.valid ? in_bytes : 0), 0x00020000);
  };
  auto rs_weights = [&](const Item& it) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.wp + (long long)it.grp * p.wgroup_bytes + (size_t)it.nb * p.chunks_total * W_B), 0,
                                             __builtin_amdgcn_readfirstlane(it.valid ? w_bytes : 0), 0x00020000);
  };
  auto rs_table = [&](const Item& it) __attribute__((always_inline)) {
    if constexpr (MODE == 2) {  // the (A, B, K) rows of the item's BatchNorm group, from this N block's first channel on ([3][bnb_c] floats per group)
      const int g = it.n / p.bnb_group;
      return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)(p.bnb_coef + (size_t)g * 3 * p.bnb_c + it.nb * NB)), 0,
                                               __builtin_amdgcn_readfirstlane(it.valid ? (3 * p.bnb_c - it.nb * NB) * 4 : 0), 0x00020000);
    }
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)(p.bias_tab + (size_t)it.grp * 9 * p.np + it.nb * NB)), 0,
                                             __builtin_amdgcn_readfirstlane(it.valid ? tab_bytes : 0), 0x00020000);
  };
  auto rs_output = [&](int n) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.out + (size_t)n * p.H * p.W * p.out_s * 2), 0, out_bytes, 0x00020000);
  };
  auto rs_bnb_x = [&](int n) __attribute__((always_inline)) {  // MODE 2: the BatchNorm's input, laid out like the output (the launcher checks)
    return __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.bnb_x + (size_t)n * p.H * p.W * p.out_s * 2), 0, out_bytes, 0x00020000);
  };
  // one input piece (j: 0..5) of chunk ci of item `it` into this wave's private stage
  auto dma_in = [&](const Item& it, __amdgpu_buffer_rsrc_t rs, int ci, int stage, int j) __attribute__((always_inline)) {
    const unsigned dst = lds0 + (unsigned)(IN0 + (wave * 2 + stage) * PIN_B + j * 1024);
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((it.y0 + 8 * wave) * p.W + it.x0) * pxb + ci * KC * 2));
    if (j + 1 < NPJ || lane < PPIECES - (NPJ - 1) * 64) bufdma16((it.in_mask >> j) & 1u ? in_base_off[j] + (unsigned)((j & 3) * 1024) : DMA_SENT, rs, so, dst);  // the last instruction: 40 lanes (the rest is masked off)
  };
  // one weight piece (i: 0..8 -> 1 KiB piece wave + 4 i) of chunk ci (the prologue's assignment; the K loop gives a wave nine CONSECUTIVE pieces)
  auto dma_w = [&](__amdgpu_buffer_rsrc_t rs, int ci, int stage, int i) __attribute__((always_inline)) {
    const int pc = wave + 4 * i;
    bufdma16(lane * 16, rs, __builtin_amdgcn_readfirstlane((unsigned)(ci * W_B + pc * 1024)), lds0 + (unsigned)(stage * W_B + pc * 1024));
  };
  // the bias table of item `it`: two pieces per wave (1 KiB = two classes; pieces past the table repeat piece 4 = {class 8, class 8})
  auto dma_tab = [&](__amdgpu_buffer_rsrc_t rs, int i) __attribute__((always_inline)) {
    int pp = wave + 4 * i; pp = pp < 4 ? pp : 4;
    constexpr int CLS_MAX = MODE == 2 ? 2 : 8;  // MODE 2: the three coefficient rows land as [3][NB] floats at TAB0 (piece 1 = {K, K}; later pieces repeat it)
    int cls = 2 * pp + (lane >> 5); cls = cls < CLS_MAX ? cls : CLS_MAX;
    bufdma16((unsigned)(cls * (MODE == 2 ? p.bnb_c : p.np) * 4 + (lane & 31) * 16), rs, 0u, lds0 + (unsigned)(TAB0 + pp * 1024));
  };

  // ---- per-lane constants of the fragment reads (as conv3x3_bf16.hip; the halo rows are wave-local) ----
  const int rowpar = (r >> 4) & 1;
  const int a_lane = ((r >> 4) * HALO_W + (r & 15)) * PIX_B;
  const int a_half_even = 16 * (kh ^ rowpar), a_half_odd = 16 * (kh ^ rowpar ^ 1);
  const int b_lane = r * PIX_B + 16 * (kh ^ ((r >> 3) & 1));
  // Pixel fragments are SHARED between the taps of a column: fragment mf of tap (ky = 2, kx) is rows 2 mf + 2, 2 mf + 3 of the halo = fragment mf + 1 of
  // tap (0, kx).  The three fragments mf = 1 .. 3 of a top-row tap are read into `keep[kx]` and used again six taps later, whose own read is then the
  // one new fragment (rows 8, 9): 63 fragment reads per chunk instead of 72 (the reads are paid in clock: the chip is power-limited and a build without
  // them runs at 2.04 instead of 1.80 GHz, profiles/r04_conv_w4_ablation.txt).  Same operands, same order of the products: results unchanged bit for bit.
#ifdef SF_EXP_W4_NOREUSE   // ablation build: every tap reads its four pixel fragments
  constexpr bool REUSE = false;
#else
  constexpr bool REUSE = true;
#endif
  bf16x8 fa[3][MFR], fb[3][NF], keep[3][MFR - 1];
  // MODE 3, window-major pixel fragments: F[a][b] = pixels (2 wy + a, 2 wx + b) of the halo, a, b = 0 .. 3; fragment mf = 2 dy + dx of tap (ky, kx) is
  // F[dy + ky][dx + kx].  F[a][b] is first needed at tap 3 max(a - 1, 0) + max(b - 1, 0) and last at tap 3 min(a, 2) + min(b, 2): one set of 16 serves
  // the whole chunk loop - a fragment is re-read (for the next chunk) one tap before its first use, always after its last use in this chunk.
  bf16x8 F[4][4];
  const int wwy = r >> 3, wwx = r & 7;
  const int p_lane = (2 * wwy * HALO_W + wwx) * PIX_B;
  const int p_half0 = 16 * (kh ^ (wwy & 1)), p_half1 = 16 * (kh ^ (wwy & 1) ^ 1);   // by the parity of the halo's window row (2 wy + a) >> 1 = wy + (a >> 1)
  auto pool_first_use = [](int a, int b) { return 3 * (a > 1 ? a - 1 : 0) + (b > 1 ? b - 1 : 0); };
  auto load_tap = [&](int sw, int si, int tap) __attribute__((always_inline)) {  // operands of `tap` into set tap % 3 (and keep[kx])
    const int ky = tap / 3, kx = tap % 3, set = tap % 3;
    const char* inb = lds + IN0 + (wave * 2 + si) * PIN_B + a_lane;
    const char* wb = lds + sw * W_B + b_lane;
    if constexpr (WIN) {
      const char* pb = lds + IN0 + (wave * 2 + si) * PIN_B + p_lane;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (pool_first_use(a, b) == tap)
            F[a][b] = *reinterpret_cast<const bf16x8*>(pb + (a * HALO_W + (b & 1) * 9 + (b >> 1)) * PIX_B + ((a >> 1) ? p_half1 : p_half0));
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) fb[set][nf] = *reinterpret_cast<const bf16x8*>(wb + (tap * NB + nf * 32) * PIX_B);
      return;
    }
#pragma unroll
    for (int mf = 0; mf < MFR; ++mf) {
      if (REUSE && ky == 2 && mf + 1 < MFR) continue;  // in keep[kx][mf] since tap kx
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(inb + ((2 * mf + ky) * HALO_W + kx) * PIX_B + ((ky & 1) ? a_half_odd : a_half_even));
      if (REUSE && ky == 0 && mf >= 1) keep[kx][mf - 1] = v; else fa[set][mf] = v;
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) fb[set][nf] = *reinterpret_cast<const bf16x8*>(wb + (tap * NB + nf * 32) * PIX_B);
  };
  auto frag_a = [&](int tap, int mf) __attribute__((always_inline)) -> const bf16x8& {
    const int ky = tap / 3, kx = tap % 3;
    if constexpr (WIN) return F[(mf >> 1) + ky][(mf & 1) + kx];
    if (REUSE && ky == 0 && mf >= 1) return keep[kx][mf - 1];
    if (REUSE && ky == 2 && mf + 1 < MFR) return keep[kx][mf];
    return fa[tap % 3][mf];
  };
  auto tap_reads = [](int tap) {
    if (WIN) return NF + (tap == 0 ? 4 : (tap == 4 || tap == 5 || tap == 7 || tap == 8) ? 1 : 2);
    return NF + ((REUSE && tap / 3 == 2) ? 1 : MFR);
  };

  // ---- epilogue of one fragment: 32 channels x 32 pixels -> two 16-byte stores per lane (always issued) ----
  // A lane pair (r, kh = 0 / 1) holds 2 x 16 bytes of its pixel per octet: stored as they are, an instruction writes 32-byte runs 512 bytes apart, and the
  // L2 takes a request per run.  One v_permlane16_swap per dword turns the two octets of a lane pair into the two pixel ROWS of the fragment: the
  // four 16-lane rows of a register then hold four consecutive 16-byte pieces of one pixel - 64-byte runs, half the requests (the same exchange as in
  // convgru_seq.hip).  Lane (rho = lane >> 4, i = lane & 15) addresses pixel (row j of the fragment, i) for register j and piece
  // 2 * (rho & 1) + (rho >> 1) of the fragment's 32 channels.
  const int rho = lane >> 4, piece = 2 * (rho & 1) + (rho >> 1);
  auto epi_frag = [&](const f32x16& a, __amdgpu_buffer_rsrc_t rs_out, unsigned voff0, unsigned voff1, int nb_item, int nf, float* s1 = nullptr,
                      float* s2 = nullptr, bool own_ok = true) __attribute__((always_inline)) {
    const int cb = nb_item * NB + nf * 32;
    u32x4_t oc[2];
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      // (the one-item kernel adds a zero bias vector here, which turns an accumulator that is exactly -0 into +0: the two kernels agree
      // in every VALUE - torch.equal - and differ in the sign bit of such zeros; 256 additions per item are not worth that bit)
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = acc_read(a[4 * g + i]);
      const unsigned ax = pk2(v[0], v[1]), ay = pk2(v[2], v[3]);
      const unsigned bx = pk2(v[4], v[5]), by = pk2(v[6], v[7]);
      if constexpr (MODE == 1) {  // BatchNorm statistics of the STORED (rounded) values of this lane's own pixel: channels 8g + 4kh + c and 8(g+1) + 4kh + c
        const unsigned pk4[4] = {ax, ay, bx, by};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = own_ok ? __builtin_bit_cast(float, pk4[i] << 16) : 0.f, hi = own_ok ? __builtin_bit_cast(float, pk4[i] & 0xffff0000u) : 0.f;
          s1[4 * g + 2 * i] += lo; s2[4 * g + 2 * i] = __builtin_fmaf(lo, lo, s2[4 * g + 2 * i]);
          s1[4 * g + 2 * i + 1] += hi; s2[4 * g + 2 * i + 1] = __builtin_fmaf(hi, hi, s2[4 * g + 2 * i + 1]);
        }
      }
      const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      oc[g >> 1] = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
    }
    u32x4_t r0, r1;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const auto sw = __builtin_amdgcn_permlane16_swap(oc[0][d], oc[1][d], false, false);
      r0[d] = sw[0]; r1[d] = sw[1];
    }
    // the fragment's position goes into the SCALAR offset (an out-of-range one past out_c - a multiple of 32 here: voff + soff never wraps back)
    const unsigned soff = cb < p.out_c ? (unsigned)(nf * 32 * 2) : 0x40000000u;
    __builtin_amdgcn_raw_buffer_store_b128(r0, rs_out, voff0, soff, ST_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(r1, rs_out, voff1, soff, ST_AUX);
  };
  // byte offset of this lane's piece of pixel (row j of fragment mf, lane & 15) inside the item's output image (+ this N block's first channel), or the sentinel
  auto out_voff = [&](const Item& it, int mf, int j) __attribute__((always_inline)) -> unsigned {
    // (window-major: store register j of fragment mf = (dy, dx) holds the windows of rows 2 j + ((lane >> 3) & 1), column lane & 7)
    const int py = WIN ? it.y0 + 8 * wave + 2 * (2 * j + ((lane >> 3) & 1)) + (mf >> 1) : it.y0 + 8 * wave + 2 * mf + j;
    const int px = WIN ? it.x0 + 2 * (lane & 7) + (mf & 1) : it.x0 + (lane & 15);
    return (py < p.H && px < p.W) ? (unsigned)(((py * p.W + px) * p.out_s + it.nb * NB + 8 * piece) * 2) : DMA_SENT;
  };

#ifdef SF_EXP_W4_CLK
  const unsigned long long clk0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
#endif
  Item cur, nxt;
  setup(0, cur);
  nxt = cur; nxt.valid = 0;  // (set up for real inside the first chunk)

  // ---- prologue: chunks 0 and 1 of the weights, chunk 0 of the input, the first bias table ----
  {
    const __amdgpu_buffer_rsrc_t rw = rs_weights(cur), ri = rs_input(cur), rt = rs_table(cur);
#pragma unroll
    for (int i = 0; i < 9; ++i) dma_w(rw, 0, 0, i);
#pragma unroll
    for (int i = 0; i < 9; ++i) dma_w(rw, 1, 1, i);
#pragma unroll
    for (int j = 0; j < NPJ; ++j) dma_in(cur, ri, 0, 0, j);
    dma_tab(rt, 0); dma_tab(rt, 1);
  }
  SF_VMCNT(0);
  __builtin_amdgcn_s_barrier();

  f32x16 acc[MFR][NF];
  // accumulators of an item START at the border-class bias of each lane's pixel (the folded BatchNorm; zeros without a table), read from the LDS copy
  auto init_acc = [&](const Item& it) __attribute__((always_inline)) {
    if constexpr (MODE == 2) {  // (the table region holds the epilogue's coefficients, not a bias)
#pragma unroll
      for (int mf = 0; mf < MFR; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;
      return;
    }
#pragma unroll
    for (int mf = 0; mf < MFR; ++mf) {
      const int cls = WIN ? border_cls(it.y0 + 8 * wave + 2 * wwy + (mf >> 1), it.x0 + 2 * wwx + (mf & 1), p.H, p.W)
                                : border_cls(it.y0 + 8 * wave + 2 * mf + (r >> 4), it.x0 + (r & 15), p.H, p.W);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const char* t = lds + TAB0 + cls * (NB * 4) + (nf * 32 + 4 * kh) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(t + q * 32);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[mf][nf][4 * q + c] = b[c];
        }
      }
    }
  };
  // (Measured and removed, round 4: half of an item's stores held back in 64 spare registers and issued one per tap under the next item's first two
  // chunks - or two per tap under its first chunk.  2.167 against 2.128 ms same box, 6990 against 6560 cycles per chunk: a 1 KiB store blocks the only wave
  // of its SIMD for ~400 cycles wherever it is issued - the CU's ~10 B/clk store path shared by four waves that store at the same time - so spreading the
  // stores moves the idle matrix-pipe time, it does not hide it; profiles/r04_conv_w4_ablation.txt.)
  auto epilogue = [&](const Item& it, int free_stage) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs_out = rs_output(it.n);
    if constexpr (MODE == 1) {
      // ---- with the BatchNorm statistics of the tile (sf_conv3x3_fwd_stats semantics: per-tile sum and sum of squares of the stored values) ----
      // channel fragments outermost: 16 + 16 running sums per lane over its four pixels, then a reduce-scatter over the 32 lanes of a half-wave
      // (ds_swizzle xor 16 .. 1: 31 exchanges instead of 160 for a plain butterfly of every value) that leaves entry r of
      // [sum of the 16 channels | sum of squares] in lane r; the four waves' partial sums meet in each wave's FREE private input stage
      // (nobody else touches it), one workgroup barrier, 256 threads add the four partials in wave order and store the tile's row,
      // one more barrier before the stages are filled again.
      unsigned vo0[MFR], vo1[MFR];
      unsigned okm = 0;
#pragma unroll
      for (int mf = 0; mf < MFR; ++mf) {
        vo0[mf] = out_voff(it, mf, 0); vo1[mf] = out_voff(it, mf, 1);
        const int py = WIN ? it.y0 + 8 * wave + 2 * wwy + (mf >> 1) : it.y0 + 8 * wave + 2 * mf + (r >> 4);
        const int px = WIN ? it.x0 + 2 * wwx + (mf & 1) : it.x0 + (r & 15);
        okm |= (py < p.H && px < p.W) ? 1u << mf : 0u;
      }
      float* part = reinterpret_cast<float*>(lds + IN0 + (wave * 2 + free_stage) * PIN_B);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        float L[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) L[i] = 0.f;
#pragma unroll
        for (int mf = 0; mf < MFR; ++mf) {
          epi_frag(acc[mf][nf], rs_out, vo0[mf], vo1[mf], it.nb, nf, L, L + 16, (okm >> mf) & 1u);
          __builtin_amdgcn_sched_barrier(0);
        }
        reduce_scatter_level<16>(L, r); reduce_scatter_level<8>(L, r); reduce_scatter_level<4>(L, r); reduce_scatter_level<2>(L, r);
        reduce_scatter_level<1>(L, r);
        part[(nf * 2 + kh) * 32 + r] = L[0];
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      {
        const int c = tid & 127, which = tid >> 7;
        const int cc = c & 31, idx = (((c >> 5) * 2 + ((cc >> 2) & 1)) * 32 + which * 16 + 4 * (cc >> 3) + (cc & 3));
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += *reinterpret_cast<const float*>(lds + IN0 + (w * 2 + free_stage) * PIN_B + idx * 4);
        const size_t tile_lin = (size_t)(it.n * p.tiles_y + it.y0 / 32) * p.tiles_x + it.x0 / TILE_W;
        if (it.nb * NB + c < p.stats_np) p.stats[(tile_lin * p.stats_np + it.nb * NB + c) * 2 + which] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      return;
    }
    if constexpr (MODE == 2) {
      // ---- BatchNorm backward in the epilogue (sf_conv3x3_bwd_data_bn): out = A * acc + B * x + K, x = the BatchNorm's input at the output's own
      // pixels and channels.  x is read in the 64-byte runs the result is stored in (two 16-byte loads per fragment) and taken
      // back to the accumulator layout by the store path's two exchanges in reverse (both are involutions); loads run two fragments ahead of
      // their use (24 registers), every wave issues exactly 32 loads and 32 stores per item and waits by count.  PIXEL fragments outermost, as in the
      // plain epilogue: the four channel fragments of a pixel row are the four 64-byte quarters of two 128-byte lines and must be touched back to back -
      // channel fragments outermost (one read of the coefficients per fragment column instead of per fragment) measured 5.1 GB read + 1.56 GB written per
      // launch against 3.1 + 1.2 algorithmic: half-line writes and the second half of every x line fetched again.
      const __amdgpu_buffer_rsrc_t rs_x = rs_bnb_x(it.n);
      unsigned vo0[MFR], vo1[MFR];
#pragma unroll
      for (int mf = 0; mf < MFR; ++mf) { vo0[mf] = out_voff(it, mf, 0); vo1[mf] = out_voff(it, mf, 1); }
      // The LOADS address x in WHOLE 128-byte lines: a load instruction covers 8 pixels x one line = the 64-byte quarters of TWO neighbouring channel
      // fragments (lane L = pixel L / 8 [+ 8 for the second instruction of the row], 16-byte piece L % 8 of the line), four instructions per pair of
      // fragments - as many as before.  History of this read (PMC, against a build without it, -DSF_EXP_W4_NOX): in the store's lane order (a
      // quarter-wave = one piece of 16 different pixels) 2.93 GB fetched per launch for this 1.21 GB tensor - four passes per instruction that each
      // touch 16 lines, not merged on their way to memory; adjacent lanes on one 64-byte run: 2.53 GB - the two halves of a line belong to neighbouring
      // fragments and were requested back to back by two instructions, both missed; whole lines (round 6): see profiles/r06_w4_bnb_lines.txt.
      // From line order to the store order: one DPP move per dword gathers a FRAGMENT's 64 pieces from the two instructions of a row (lanes with
      // L % 8 >= 4 take the other instruction's lane L - 4 for the even fragment, lanes with L % 8 < 4 its lane L + 4 for the odd one: row_shr:4 /
      // row_shl:4 under a bank mask), then the ds_bpermute per dword as before (source lane 8 (pixel % 8) + 4 (pixel / 8 [even] or 1 - pixel / 8 [odd])
      // + piece; no LDS memory involved).
      unsigned lo0[MFR], lo1[MFR];
#pragma unroll
      for (int mf = 0; mf < MFR; ++mf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int py = it.y0 + 8 * wave + 2 * mf + j, px = it.x0 + (lane >> 3);
          (j ? lo1[mf] : lo0[mf]) = (py < p.H && px < p.W) ? (unsigned)(((py * p.W + px) * p.out_s + it.nb * NB + 8 * (lane & 7)) * 2) : DMA_SENT;
        }
      // the second instruction of a row: 8 pixels to the right (a sentinel stays one: DMA_SENT + the step is still past every descriptor's range)
      const bool right_ok = it.x0 + 8 + (lane >> 3) < p.W;
      const unsigned right_step = (unsigned)(8 * p.out_s * 2);
      auto right = [&](unsigned o) __attribute__((always_inline)) { return right_ok ? o + right_step : DMA_SENT; };
      const int px16 = lane & 15;
      const int bp_even = (8 * (px16 & 7) + 4 * (px16 >> 3) + piece) * 4, bp_odd = (8 * (px16 & 7) + 4 * (1 - (px16 >> 3)) + piece) * 4;
      constexpr int NPAIR = MFR * NF / 2;
      u32x4_t xq[2][2][2];   // [pair & 1][row j][left / right instruction]
      auto request = [&](int q) __attribute__((always_inline)) {
        const int mf = q / (NF / 2), nfp = q % (NF / 2);
        const unsigned soff = it.nb * NB + nfp * 64 < p.out_c ? (unsigned)(nfp * 64 * 2) : 0x40000000u;
#ifdef SF_EXP_W4_NOX   // ablation build (tools/ablate_w4.sh): no read of x - what the input tiles alone cost in this mode
        for (int j = 0; j < 2; ++j) for (int h = 0; h < 2; ++h) xq[q & 1][j][h] = u32x4_t{soff, 0u, 0u, 0u};
        return;
#endif
        xq[q & 1][0][0] = bufload16(lo0[mf], rs_x, soff);
        xq[q & 1][0][1] = bufload16(right(lo0[mf]), rs_x, soff);
        xq[q & 1][1][0] = bufload16(lo1[mf], rs_x, soff);
        xq[q & 1][1][1] = bufload16(right(lo1[mf]), rs_x, soff);
      };
      request(0);
#pragma unroll
      for (int mf = 0; mf < MFR; ++mf) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int f = mf * NF + nf, q = f >> 1;
          f32x4 cA[4], cB[4], cK[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const char* lc = lds + TAB0 + (nf * 32 + 8 * g + 4 * kh) * 4;
            cA[g] = *reinterpret_cast<const f32x4*>(lc); cB[g] = *reinterpret_cast<const f32x4*>(lc + NB * 4); cK[g] = *reinterpret_cast<const f32x4*>(lc + 2 * NB * 4);
          }
          u32x4_t (&xp)[2][2] = xq[q & 1];
          if ((f & 1) == 0) {
            if (q + 1 < NPAIR) request(q + 1);
            // younger than this pair's four loads: the four loads of the next pair and the four stores of the previous one
            if (q == 0 || q == NPAIR - 1) wait_loaded4<4>(xp[0][0], xp[0][1], xp[1][0], xp[1][1]); else wait_loaded4<8>(xp[0][0], xp[0][1], xp[1][0], xp[1][1]);
          }
          // line order -> fragment (DPP) -> store order (ds_bpermute) -> accumulator layout: rows back (v_permlane16_swap), then quads back (v_permlane32_swap)
          u32x4_t oc[2];
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            int g0, g1;
            if ((f & 1) == 0) {
              g0 = __builtin_amdgcn_update_dpp((int)xp[0][0][d], (int)xp[0][1][d], 0x114, 0xF, 0xA, false);   // row_shr:4 into lanes 4-7, 12-15 of every row
              g1 = __builtin_amdgcn_update_dpp((int)xp[1][0][d], (int)xp[1][1][d], 0x114, 0xF, 0xA, false);
            } else {
              g0 = __builtin_amdgcn_update_dpp((int)xp[0][0][d], (int)xp[0][1][d], 0x104, 0xF, 0x5, false);   // row_shl:4 into lanes 0-3, 8-11
              g1 = __builtin_amdgcn_update_dpp((int)xp[1][0][d], (int)xp[1][1][d], 0x104, 0xF, 0x5, false);
            }
            const int bp = (f & 1) ? bp_odd : bp_even;
            const unsigned s0 = (unsigned)__builtin_amdgcn_ds_bpermute(bp, g0), s1 = (unsigned)__builtin_amdgcn_ds_bpermute(bp, g1);
            const auto sw = __builtin_amdgcn_permlane16_swap(s0, s1, false, false);
            oc[0][d] = sw[0]; oc[1][d] = sw[1];
          }
          const f32x16& a = acc[mf][nf];
          u32x4_t so[2];
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const auto qx = __builtin_amdgcn_permlane32_swap(oc[g >> 1][0], oc[g >> 1][2], false, false);   // (ax, bx): channels 0, 1 of octets g, g + 1
            const auto qy = __builtin_amdgcn_permlane32_swap(oc[g >> 1][1], oc[g >> 1][3], false, false);   // (ay, by): channels 2, 3
            float v[8];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
              const unsigned w0 = qx[gg], w1 = qy[gg];
              const float xv[4] = {__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xffff0000u),
                                   __builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
#pragma unroll
              for (int c = 0; c < 4; ++c)
                v[4 * gg + c] = __builtin_fmaf(cA[g + gg][c], acc_read(a[4 * (g + gg) + c]), __builtin_fmaf(cB[g + gg][c], xv[c], cK[g + gg][c]));
            }
            const unsigned ax = pk2(v[0], v[1]), ay = pk2(v[2], v[3]);
            const unsigned bx = pk2(v[4], v[5]), by = pk2(v[6], v[7]);
            const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            so[g >> 1] = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
          }
          u32x4_t r0, r1;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const auto sw = __builtin_amdgcn_permlane16_swap(so[0][d], so[1][d], false, false);
            r0[d] = sw[0]; r1[d] = sw[1];
          }
          const unsigned soff = it.nb * NB + nf * 32 < p.out_c ? (unsigned)(nf * 32 * 2) : 0x40000000u;
          __builtin_amdgcn_raw_buffer_store_b128(r0, rs_out, vo0[mf], soff, ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(r1, rs_out, vo1[mf], soff, ST_AUX);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
    if constexpr (MODE == 3) {
      // ---- 2x2 max-pooling in the epilogue (sf_conv3x3_fwd_folded_pool): the four fragments of a channel block are the four pixels of this lane's
      // window.  Values are rounded to bf16 FIRST (the unfused path pools the stored tensor: ties between values that differ only below bf16
      // precision must go to the first window element), the maximum is taken as fmax(fmax(v0, v1), fmax(v2, v3)) like maxpool_fwd_kernel, the code is
      // the first maximum in row-major order: c01 = v1 > v0, c23 = v3 > v2, bottom = max23 > max01, code = bottom ? 2 + c23 : c01.
      // Stores: per channel block two 16-byte pieces per lane exactly as the plain epilogue's (quads paired over the half-waves, then rows of 16
      // lanes exchanged so that a register holds 64-byte runs) - a register then belongs to window rows wy = 0, 1 (r0) or 2, 3 (r1) - and the routing
      // words: a lane holds one byte of each of the block's four 16-bit words, the half-wave partner the other; 8 + 2 stores per wave, always issued.
      const int Ho = p.H >> 1, Wo = p.W >> 1;
      long long n_out = it.n;
      if (p.pool_L > 0) { const long long b = it.n % p.pool_B, t = (it.n / p.pool_B) % p.pool_T, l = it.n / ((long long)p.pool_B * p.pool_T); n_out = (t * p.pool_L + l) * p.pool_B + b; }
      const int pool_bytes = __builtin_amdgcn_readfirstlane(Ho * Wo * p.pool_s * 2);
      const int q8 = p.out_c >> 3;   // routing words per pooled pixel
      const __amdgpu_buffer_rsrc_t rs_pool = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.pool_out + (size_t)n_out * Ho * Wo * p.pool_s * 2), 0, pool_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_route = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)p.pool_route + (size_t)it.n * Ho * Wo * q8 * 2), 0,
                                                                                __builtin_amdgcn_readfirstlane(Ho * Wo * q8 * 2), 0x00020000);
      unsigned pv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // register j of the store path: window rows 2 j + ((lane >> 3) & 1), window column lane & 7, piece as in the plain epilogue
        const int pyo = (it.y0 >> 1) + 4 * wave + 2 * j + ((lane >> 3) & 1), pxo = (it.x0 >> 1) + (lane & 7);
        pv[j] = (pyo < Ho && pxo < Wo) ? (unsigned)(((pyo * Wo + pxo) * p.pool_s + it.nb * NB + 8 * piece) * 2) : DMA_SENT;
      }
      const int ryo = (it.y0 >> 1) + 4 * wave + wwy, rxo = (it.x0 >> 1) + wwx;   // this lane's own window
      const unsigned rv_off = (ryo < Ho && rxo < Wo) ? (unsigned)(((ryo * Wo + rxo) * q8 + it.nb * (NB / 8)) * 2 + 8 * kh) : DMA_SENT;
      unsigned rd[NF][2];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        unsigned P[8], Wc = 0;
#pragma unroll
        for (int jj = 7; jj >= 0; --jj) {   // channel pairs (i = 2 jj, 2 jj + 1 of the lane's 16), high to low: the code word is shifted in
          float rv[4][2];
#pragma unroll
          for (int mf = 0; mf < MFR; ++mf) {
            const unsigned pk = pk2(acc_read(acc[mf][nf][2 * jj]), acc_read(acc[mf][nf][2 * jj + 1]));
            rv[mf][0] = __builtin_bit_cast(float, pk << 16); rv[mf][1] = __builtin_bit_cast(float, pk & 0xffff0000u);
          }
          float mx[2];
#pragma unroll
          for (int e = 1; e >= 0; --e) {
            // (v_max_f32 through asm: the operands come out of bit operations, and hipcc would canonicalise each of them - one more v_max apiece -
            // before an fmaxf; the comparisons as wave masks, the code bits combined on the scalar unit and shifted into the word through the carry)
            const float m01 = vmax(rv[0][e], rv[1][e]), m23 = vmax(rv[2][e], rv[3][e]);
            const unsigned long long c01 = __builtin_amdgcn_ballot_w64(rv[1][e] > rv[0][e]), c23 = __builtin_amdgcn_ballot_w64(rv[3][e] > rv[2][e]);
            const unsigned long long bot = __builtin_amdgcn_ballot_w64(m23 > m01);
            mx[e] = vmax(m01, m23);
            Wc = shift_in(shift_in(Wc, bot), (c23 & bot) | (c01 & ~bot));
          }
          P[jj] = (__builtin_bit_cast(unsigned, mx[0]) >> 16) | (__builtin_bit_cast(unsigned, mx[1]) & 0xffff0000u);
        }
        // pooled values: quads -> octets over the half-waves, rows of 16 lanes -> 64-byte runs (as epi_frag)
        u32x4_t oc[2];
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          const auto sx = __builtin_amdgcn_permlane32_swap(P[2 * g], P[2 * g + 2], false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(P[2 * g + 1], P[2 * g + 3], false, false);
          oc[g >> 1] = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
        }
        u32x4_t r0, r1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const auto sw16 = __builtin_amdgcn_permlane16_swap(oc[0][d], oc[1][d], false, false);
          r0[d] = sw16[0]; r1[d] = sw16[1];
        }
        const unsigned soff = it.nb * NB + nf * 32 < p.out_c ? (unsigned)(nf * 32 * 2) : 0x40000000u;
        __builtin_amdgcn_raw_buffer_store_b128(r0, rs_pool, pv[0], soff, ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(r1, rs_pool, pv[1], soff, ST_AUX);
        // routing words of this block's four octets: word g = byte g of the kh = 0 lane | byte g of the kh = 1 lane << 8
        const auto wsw = __builtin_amdgcn_permlane32_swap(Wc, Wc, false, false);   // [0]: the kh = 0 lane's word, [1]: the kh = 1 lane's (in both lanes)
        rd[nf][0] = __builtin_amdgcn_perm(wsw[1], wsw[0], 0x05010400u);
        rd[nf][1] = __builtin_amdgcn_perm(wsw[1], wsw[0], 0x07030602u);
        __builtin_amdgcn_sched_barrier(0);  // (one channel block's temporaries at a time)
      }
      // 8 bytes of routing per channel block: blocks 0, 2 leave from the kh = 0 lane, blocks 1, 3 from the kh = 1 lane (its offset carries + 8)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t v = kh ? u32x2_t{rd[2 * q + 1][0], rd[2 * q + 1][1]} : u32x2_t{rd[2 * q][0], rd[2 * q][1]};
        const unsigned soff = it.nb * NB + q * 64 < p.out_c ? (unsigned)(q * 16) : 0x40000000u;
        __builtin_amdgcn_raw_buffer_store_b64(v, rs_route, rv_off, soff, ST_AUX);
      }
      return;
    }
#pragma unroll
    for (int mf = 0; mf < MFR; ++mf) {
      const unsigned voff0 = out_voff(it, mf, 0), voff1 = out_voff(it, mf, 1);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        epi_frag(acc[mf][nf], rs_out, voff0, voff1, it.nb, nf);
        __builtin_amdgcn_sched_barrier(0);  // (one fragment's temporaries at a time)
      }
    }
  };
  init_acc(cur);

  // fragment sets rotate modulo 3 (tap t uses set t % 3, the next tap's operands are read into set (t + 1) % 3): nine taps per chunk, so the
  // rotation is the same in every chunk and the next chunk's first tap lands in set 0; only two sets are live at any time
  load_tap(0, 0, 0);

#ifdef SF_EXP_W4_SPVALU
  unsigned spd[4] = {(unsigned)lane, (unsigned)tid, 0x07060302u, 0x05040100u};
#endif
  int sw = 0;    // weight ring stage of the current chunk
  int gpar = 0;  // parity of the chunk counter: the private input stage
  // K >= 1 items of nch >= 3 chunks (the launcher's contract; do-while loops: no empty-loop join points for the 256 accumulators)
  int k = 0;
  do {
    int ci = 0;
    do {
      const bool first = ci == 0;  // the previous item's 32 epilogue stores are in flight
      const bool last = ci + 1 == nch, last2 = ci + 2 >= nch;
      const int sw1 = sw == 2 ? 0 : sw + 1, sw2 = sw == 0 ? 2 : sw - 1;  // stages of chunks g + 1, g + 2
      const int si = gpar, si1 = gpar ^ 1;
      // everything the chunk's 15 DMA pieces need, computed ONCE here (scalar code in front of the first tap's MFMAs, which the scheduler
      // spreads between them): descriptors, scalar offsets, LDS destinations; a piece is then `s_mov m0` + the load, issued BETWEEN two MFMAs
      const Item& it1 = last ? nxt : cur;
      const Item& it2 = last2 ? nxt : cur;
      const __amdgpu_buffer_rsrc_t ri = rs_input(it1);
      const __amdgpu_buffer_rsrc_t rw = rs_weights(it2);
      const unsigned so_in = __builtin_amdgcn_readfirstlane((unsigned)(((it1.y0 + 8 * wave) * p.W + it1.x0) * pxb + (last ? 0 : ci + 1) * KC * 2));
      const unsigned so_w = __builtin_amdgcn_readfirstlane((unsigned)((last2 ? ci + 2 - nch : ci + 2) * W_B + wave * 9 * 1024));   // this wave's nine consecutive pieces
      const unsigned dst_in = lds0 + (unsigned)(IN0 + (wave * 2 + si1) * PIN_B);
      const unsigned dst_w = lds0 + (unsigned)(sw2 * W_B + wave * 9 * 1024);
      const unsigned m1 = it1.in_mask;
      // (-DSF_EXP_W4_*: ablation builds of tools/ablate_w4.sh, never part of the shipped library)
      auto piece_in = [&](int j) __attribute__((always_inline)) {
#ifdef SF_EXP_W4_NODMA
        return;
#endif
        const unsigned voff = (m1 >> j) & 1u ? in_base_off[j] : DMA_SENT;
        if ((j & 3) == 0) dma_group(dst_in + j * 1024);   // pieces 0 .. 3 | 4, 5
        if (j + 1 < NPJ || lane < PPIECES - (NPJ - 1) * 64) bufdma16_q(j & 3, voff, ri, so_in);  // the last instruction: 40 lanes
      };
      auto piece_w = [&](int i) __attribute__((always_inline)) {
#ifdef SF_EXP_W4_NODMA
        return;
#endif
        if ((i & 3) == 0) dma_group(dst_w + i * 1024);    // pieces 0 .. 3 | 4 .. 7 | 8
        bufdma16_q(i & 3, lane * 16, rw, so_w + (i & ~3) * 1024);
      };

#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#ifdef SF_EXP_W4_NOREAD
        if (tap == 8 && ci + 100 == nch) {
#else
        if (tap == 8) {
#endif
          // this wave's input pieces of chunk g + 1 have landed: everything younger (the 9 weight pieces of chunk g + 2 and, in the second-to-last
          // chunk of an item, the two table pieces behind them) may stay in flight
#ifndef SF_EXP_W4_NOSYNC
          if (ci + 2 == nch) SF_VMCNT(11); else SF_VMCNT(9);
#endif
          __builtin_amdgcn_sched_barrier(0);
          load_tap(sw1, si1, 0);  // first tap of the next chunk
        } else {
#ifndef SF_EXP_W4_NOREAD
          load_tap(sw, si, tap + 1);
#else
          if (ci + 100 == nch) load_tap(sw, si, tap + 1);
#endif
        }
        // MFMAs 0 .. 7 with one fragment read behind each | DMA piece | MFMAs 8 .. 11 | DMA piece | MFMAs 12 .. 15
        auto mfmas = [&](int lo, int hi) __attribute__((always_inline)) {
#ifdef SF_EXP_W4_SPVALU   // estimate build (below): the vector-ALU work of building 2:4-compressed operands, ~384 instructions per chunk and wave
          asm volatile("v_perm_b32 %0, %0, %1, %2\n\tv_and_b32 %1, %1, %3\n\tv_perm_b32 %2, %2, %3, %0\n\tv_or_b32 %3, %3, %1\n\t"
                       "v_perm_b32 %0, %0, %1, %2\n\tv_and_b32 %1, %1, %3\n\tv_perm_b32 %2, %2, %3, %0\n\tv_or_b32 %3, %3, %1\n\t"
                       "v_perm_b32 %0, %0, %1, %2\n\tv_and_b32 %1, %1, %3\n\tv_perm_b32 %2, %2, %3, %0\n\tv_or_b32 %3, %3, %1\n\t"
                       "v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %3" : "+v"(spd[0]), "+v"(spd[1]), "+v"(spd[2]), "+v"(spd[3]));
#endif
#ifdef SF_EXP_W4_SPEST    // estimate build (tools/ablate_w4.sh, timing only, results wrong): 6 instead of 9 matrix instructions per (pixel, channel) fragment pair and
          if (tap % 3 == 2) return;   // chunk - what the 2:4-sparse instruction would issue for a gradient behind a 2x2 max-pool
#endif
#pragma unroll
          for (int f = lo; f < hi; ++f) {
            const int mf = f / NF, nf = f % NF;
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tap % 3][nf], frag_a(tap, mf), acc[mf][nf], 0, 0, 0);
          }
        };
        mfmas(0, 8);
#pragma unroll
        for (int q = 0; q < tap_reads(tap == 8 ? 0 : tap + 1); ++q) {  // one read of the NEXT tap's operands behind each of the first MFMAs
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // taps 0 .. 2: input of chunk g + 1 (other private stage); taps 3 .. 7: weights of chunk g + 2 (stage sw2, free behind the rendezvous)
        if (tap < 3) piece_in(2 * tap); else if (tap < 8) piece_w(2 * (tap - 3));
        __builtin_amdgcn_sched_barrier(0);
        mfmas(8, 12);
        __builtin_amdgcn_sched_barrier(0);
        if (tap < 3) piece_in(2 * tap + 1); else if (tap < 7) piece_w(2 * (tap - 3) + 1);
        if (tap == 7 && ci + 2 == nch) {  // the next item's bias table (read when its accumulators are initialised)
          const __amdgpu_buffer_rsrc_t rt = rs_table(MODE == 2 ? cur : nxt);  // (MODE 2: THIS item's coefficients, read by its epilogue)
          dma_tab(rt, 0); dma_tab(rt, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(12, 16);
        __builtin_amdgcn_sched_barrier(0);


        if (tap == 2) {
          // rendezvous of the chunk: this wave's weight pieces of chunk g + 1 (and a table behind them) have landed - only this chunk's six input
          // pieces (and, in an item's first chunk, the previous item's 32 stores in front of them) are younger -, then everybody's; stage sw2
          // (chunk g - 1) is free behind the barrier
#ifndef SF_EXP_W4_NOSYNC
          // (MODE 2: 32 loads + 32 stores + 6 pieces are younger - the counter's field ends at 63; the seven oldest of them are long complete)
          if (first) { if constexpr (MODE == 1) SF_VMCNT(39); else if constexpr (MODE == 2) SF_VMCNT(63); else if constexpr (MODE == 3) SF_VMCNT(16); else SF_VMCNT(38); } else SF_VMCNT(6);   // (MODE 3: 8 + 2 stores + 6 pieces)
          __builtin_amdgcn_s_barrier();
#endif
#ifdef SF_EXP_W4_STAGN   // experiment (tools/ablate_w4.sh): behind the chunk's rendezvous wave w idles w x STAGN x 16 cycles, so that the four waves - which run
          // the same code at the same matrix-instruction pace from here to the next rendezvous - reach the CU's one address unit with their LDS-DMA pieces
          // one after the other instead of together
#define SF_W4_NOPS(n) for (int q_ = 0; q_ < (n); ++q_) asm volatile("s_nop 15")
          if (wave >= 1) { _Pragma("unroll") SF_W4_NOPS(SF_EXP_W4_STAGN); }
          if (wave >= 2) { _Pragma("unroll") SF_W4_NOPS(SF_EXP_W4_STAGN); }
          if (wave >= 3) { _Pragma("unroll") SF_W4_NOPS(SF_EXP_W4_STAGN); }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if (tap == 4 && first) {  // the item after this one (needed from the top of the second-to-last chunk on: nch >= 3)
          setup(k + 1, nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      sw = sw1; gpar ^= 1;
    } while (++ci < nch);
    // ---- item switch: this item's epilogue (32 stores per wave, always), the next item's accumulators ----
#ifndef SF_EXP_W4_NOEPI
    epilogue(cur, gpar ^ 1);
    cur = nxt;
    init_acc(cur);
#else
    if (k + 100 == K) epilogue(cur, gpar ^ 1);
    cur = nxt;
#endif
  } while (++k < K);
#ifdef SF_EXP_W4_SPVALU
  if ((spd[0] ^ spd[1] ^ spd[2] ^ spd[3]) == 0x12345u && p.N < 0) reinterpret_cast<unsigned*>(p.out)[0] = spd[0];   // (never true: keeps the dummy chain alive)
#endif
#ifdef SF_EXP_W4_CLK
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
    const unsigned long long c = __builtin_readcyclecounter() - clk0, w = wall_clock64() - wall0;
    printf("wg %d: %llu shader cycles in %llu ticks of 10 ns = %.3f GHz; %.0f cycles per chunk\n", (int)blockIdx.x, c, w, (double)c / (double)w / 10.0, (double)c / (K * nch));
  }
#endif
}

}  // namespace

// Does this launch qualify?  (sf_conv_bf16_persist_ok has already established: linear epilogue, single bf16-stored source without
// image remap, bf16 output, 32-row tiles, >= 1024 tiles.)
bool sf_conv_bf16_persist4_ok(const sfconv::ConvParams& p, int nf) {
  static const bool off = getenv("SF_NO_CONV_W4") != nullptr;
  static const bool no_stats = getenv("SF_NO_CONV_W4_STATS") != nullptr;  // A/B switch: statistics on the 8-wave persistent kernel
  static const bool no_bnb = getenv("SF_NO_CONV_W4_BNB") != nullptr;      // A/B switch: the BatchNorm-backward epilogue on the 8-wave one-item kernel
  if (off || nf != 4 || (p.stats && no_stats) || p.bias || p.c0 / sfconv::KC < 3 || p.src1 || p.out_c % 32) return false;
  if ((long long)p.W * p.s0 * 2 < 1024) return false;   // an image row of the input is at least 1 KiB: the input pieces' immediate offsets (dma_group) never exceed a lane's own offset
  // pooled epilogue: whole windows per tile (even H, W), no statistics / BatchNorm backward, routing words of whole 64-channel pairs
  if (p.pool_out && (p.stats || p.bnb_coef || (p.H & 1) || (p.W & 1) || p.out_c % 64 || !p.pool_route || p.pool_s % 8 ||
                     (long long)(p.H / 2) * (p.W / 2) * p.pool_s * 2 >= 0x7fffffffll)) return false;
  // BatchNorm-backward epilogue: x is read at the output's own offsets
  if (p.bnb_coef && (no_bnb || p.stats || p.bias_tab || p.wgroup || p.bnb_xs != p.out_s || p.bnb_c < p.out_c || p.bnb_group < 1)) return false;
  if ((long long)p.H * p.W * p.out_s * 2 >= 0x7fffffffll || (long long)p.H * p.W * p.s0 * 2 >= 0x7fffffffll) return false;
  return true;
}

int sf_launch_conv_bf16_persist4(const sfconv::ConvParams& p0, int nblk, hipStream_t st) {
  sfconv::ConvParams p = p0;
  p.tiles_x = (p.W + sfconv::TILE_W - 1) / sfconv::TILE_W;
  p.tiles_y = (p.H + 31) / 32;
  const int items = p.tiles_x * p.tiles_y * p.N * nblk;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { sf_set_error("conv3x3_bf16_persist4: device query failed"); return 2; }
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  static const bool pair_env = getenv("SF_CONV_W4_PAIR") != nullptr;
  const int pair = pair_env && nblk > 1 ? 1 : 0;
  const int tiles = items / nblk;
  const int grid = pair ? (tiles < cus ? tiles : cus) : (items < cus ? items : cus);
  // window-major pixel fragments outside the pooled mode (round 5; bit-identical results): the statistics launches take them by default (-0.8 .. -1.0 %:
  // 2.255 -> 2.236 ms at 256 -> 256, 1.536 -> 1.520 at 160 -> 256, two A/B pairs on one box; SF_CONV_W4_WIN=0: the A/B switch back).  The plain
  // launches do not: that instantiation spilled 68 bytes per lane (the fragment set next to the pixel-fragment-outermost epilogue) and ran 2.25 -> 2.49 ms;
  // the MetNet step has no plain launch of this kernel since conv4 took the pooled mode, so it is not instantiated.
  static const char* win_env = getenv("SF_CONV_W4_WIN");
  static const bool win_stats = !win_env || win_env[0] != '0';
  if (p.pool_out) hipLaunchKernelGGL((conv3x3_bf16_persist4_kernel<3>), dim3(grid), dim3(256), 0, st, p, items, nblk, pair);
  else if (p.bnb_coef) hipLaunchKernelGGL((conv3x3_bf16_persist4_kernel<2>), dim3(grid), dim3(256), 0, st, p, items, nblk, pair);
  else if (p.stats) {
    if (win_stats) hipLaunchKernelGGL((conv3x3_bf16_persist4_kernel<1, true>), dim3(grid), dim3(256), 0, st, p, items, nblk, pair);
    else hipLaunchKernelGGL((conv3x3_bf16_persist4_kernel<1>), dim3(grid), dim3(256), 0, st, p, items, nblk, pair);
  } else hipLaunchKernelGGL((conv3x3_bf16_persist4_kernel<0>), dim3(grid), dim3(256), 0, st, p, items, nblk, pair);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error("conv3x3_bf16_persist4: launch failed: %s", hipGetErrorString(e)); return 2; }
  return 0;
}
