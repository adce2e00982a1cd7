// PERSISTENT variant of the bf16 3x3 convolution (conv3x3_bf16.hip) for the large launches of the MetNet encoder:
// one workgroup per CU walks its share of the (pixel tile, N block) work items instead of one workgroup per item.
//
// Why: with 113 KB of LDS there is one workgroup per CU, so nothing covers an item's fixed costs - workgroup launch, descriptor
// set-up, the latency of the first K chunk's DMA, the store tail of the epilogue.  Measured by repeating the K loop inside the
// one-item kernel (tools/ubench note in DESIGN.md): 256->256 @32x32 x 2304 takes 2.45 ms of which the K loops are 1.91 - 15 us of
// fixed cost per 53 us item.  Here the NEXT item's first chunk (weights + input halo tile) is requested during the LAST chunk of
// the current item, so it lands under that chunk's MFMAs and the epilogue, and there is one launch per CU.
//
// Same arithmetic, same K order, same epilogues (conv_common.h) as conv3x3_bf16_kernel<8, NF, EPI_LINEAR, false, TR, BNB>: results
// are bit-identical.  Scope: 8 waves (32x16 tiles, H > 16), ONE bf16-stored source, linear epilogue - plain / grouped weights with
// border-class bias (folded BatchNorm) / statistics (the BatchNorm-backward epilogue and NF = 5 stay on the one-item kernel: registers).
#include "conv_common.h"

namespace {

using namespace sfconv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HALO_W = TILE_W + 2;  // 18
constexpr int PIX_B = 32;
constexpr unsigned DMA_SENT = 0x80000000u;

__device__ __forceinline__ void bufdma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff) : "memory");
}
__device__ __forceinline__ void* uniform_ptr(const void* q) {  // inline asm "s" operands are not legalised
  const uintptr_t v = (uintptr_t)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (void*)(((uintptr_t)hi << 32) | lo);
}

enum { PM_TR = 0, PM_STATS = 1, PM_BNB = 2 };  // epilogue: transposed linear / channel-per-lane with statistics / transposed + BatchNorm backward

template <int NF, int MODE>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16_persist_kernel(const ConvParams p, const int items, const int nblk) {
  constexpr int WAVES = 8, THREADS = 512, NB = 32 * NF, TH = 32;
  constexpr int HALO_H = TH + 2;
  constexpr int IN_B = HALO_H * HALO_W * PIX_B;
  constexpr int W_B = 9 * NB * PIX_B;
  constexpr int PIECES = HALO_H * HALO_W * 2;
  constexpr int NPIECE = (PIECES + THREADS - 1) / THREADS;
  __shared__ __attribute__((aligned(1024))) char lds[2 * W_B + 2 * IN_B];
  __shared__ __attribute__((aligned(16))) float lds_small[MODE == PM_BNB ? 3 * NB : 2 * NB];  // BatchNorm-backward coefficients / statistics
  char* lds_w = lds;
  char* lds_in = lds + 2 * W_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int nch = p.c0 / KC;
  const int tiles_total = items / nblk;
  const bool xcd_order = tiles_total % 8 == 0 && nblk > 1;  // the N blocks of a tile on ONE XCD (conv3x3_bf16.hip)

  struct Item {
    int n, nb, x0, y0, tx, ty;
    const char* in_base; const char* w_base;  // descriptor bases (the descriptors are formed at the DMA: they must sit in SGPRs)
    unsigned so_in;
    unsigned in_off[NPIECE];
  };
  auto setup = [&](int w, Item& it) __attribute__((always_inline)) {
    int tile;
    if (xcd_order) { const int xcd = w & 7, j = w >> 3; tile = (j / nblk) * 8 + xcd; it.nb = j % nblk; }
    else { tile = w % tiles_total; it.nb = w / tiles_total; }
    it.tx = tile % p.tiles_x; tile /= p.tiles_x;
    it.ty = tile % p.tiles_y;
    it.n = tile / p.tiles_y;
    it.x0 = it.tx * TILE_W; it.y0 = it.ty * TH;
    const long long pxb = 2ll * p.s0, lead = (long long)(p.W + 1) * pxb, img = (long long)p.H * p.W * pxb;
    it.so_in = (unsigned)((it.y0 * p.W + it.x0) * (int)pxb);
    it.in_base = (const char*)p.src0 + it.n * img - lead;
    const long long wgrp = p.wgroup ? (long long)(it.n / p.wgroup) * p.wgroup_bytes : 0;
    it.w_base = (const char*)p.wp + wgrp + (size_t)it.nb * p.chunks_total * W_B;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int pc = tid + j * THREADS, pix = pc >> 1;
      const int iy = pix / HALO_W, ix = pix - iy * HALO_W;
      const int gy = it.y0 + iy - 1, gx = it.x0 + ix - 1;
      const bool ok = pc < PIECES && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const int half = (pc & 1) ^ (iy & 1);
      it.in_off[j] = ok ? (unsigned)(((iy * p.W + ix) * p.s0 + 8 * half) * 2) : DMA_SENT;
    }
  };
  // stage chunk ci of item `it` into buffer `buf`: weights (9 * NF pieces of 1 KiB over the waves) + the input halo tile
  const int in_bytes = __builtin_amdgcn_readfirstlane((int)(2ll * p.s0 * ((long long)p.H * p.W + 2 * (p.W + 1))));
  const int w_bytes = __builtin_amdgcn_readfirstlane(p.chunks_total * W_B);
  auto stage = [&](const Item& it, int ci, int buf) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(it.w_base), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(it.in_base), 0, in_bytes, 0x00020000);
    const unsigned wdst = lds0 + buf * W_B;
    for (int i = wave; i < 9 * NF; i += WAVES) bufdma16(lane * 16, rs_w, (unsigned)(ci * W_B + i * 1024), wdst + i * 1024);
    const unsigned dst = lds0 + (unsigned)(2 * W_B + buf * IN_B + wave * 1024);
    const unsigned so = it.so_in + (unsigned)(ci * KC * 2);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j)
      if (tid + j * THREADS < PIECES) bufdma16(it.in_off[j], rs_in, __builtin_amdgcn_readfirstlane(so), dst + j * WAVES * 1024);  // lanes past the tile are masked off
  };

  const int rowpar = (r >> 4) & 1;
  const int a_lane = ((4 * wave + (r >> 4)) * HALO_W + (r & 15)) * PIX_B;
  const int a_half_even = 16 * (kh ^ rowpar), a_half_odd = 16 * (kh ^ rowpar ^ 1);
  const int b_lane = r * PIX_B + 16 * (kh ^ ((r >> 3) & 1));

  Item cur;
  int w = blockIdx.x;
  if (w >= items || nch <= 0) return;
  setup(w, cur);
  int g = 0;  // chunks processed by this workgroup so far: chunk g lives in buffer g & 1
  stage(cur, 0, 0);

  for (;;) {
    const int wn = w + gridDim.x;
    const bool has_next = wn < items;
    const int n = cur.n, nb = cur.nb, x0 = cur.x0, y0 = cur.y0;

    if constexpr (MODE == PM_BNB || MODE == PM_STATS) {
      // the previous item's epilogue has read lds_small: every wave must be past it before it is rewritten
      __syncthreads();
      if constexpr (MODE == PM_BNB) {
        const float* co = p.bnb_coef + (size_t)(n / p.bnb_group) * 3 * p.bnb_c;
        for (int i = tid; i < 3 * NB; i += THREADS) {
          const int c = nb * NB + i % NB;
          lds_small[i] = c < p.out_c ? co[(size_t)(i / NB) * p.bnb_c + c] : 0.f;
        }
      } else {
        for (int i = tid; i < 2 * NB; i += THREADS) lds_small[i] = 0.f;
      }
    }

    f32x16 acc[2][NF];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;
    if (p.bias_tab) {  // folded BatchNorm: the accumulators start at the group's bias of each pixel's border class
      const float* tab = p.bias_tab + (size_t)(p.wgroup ? n / p.wgroup : 0) * 9 * p.np + nb * NB;
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
        if constexpr (MODE != PM_STATS) {
          const float* t = tab + (size_t)border_cls(y0 + 4 * wave + 2 * mf + (r >> 4), x0 + (r & 15), p.H, p.W) * p.np + 4 * kh;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const f32x4 b = *reinterpret_cast<const f32x4*>(t + nf * 32 + 8 * gq);
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[mf][nf][4 * gq + c] = b[c];
            }
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int rr = frag_row(reg, kh);
            const float* t = tab + (size_t)border_cls(y0 + 4 * wave + 2 * mf + (rr >> 4), x0 + (rr & 15), p.H, p.W) * p.np + r;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf][reg] = t[nf * 32];
          }
        }
    }

    for (int ci = 0; ci < nch; ++ci, ++g) {
      const int cbuf = g & 1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this chunk's DMA has landed (and, at ci = 0, the previous item's stores have left)
      __syncthreads();
      const bool stage_late = wave >= 4;  // the two waves of a SIMD stage at different taps (conv3x3_bf16.hip)
      auto stage_next = [&]() __attribute__((always_inline)) {
        if (ci + 1 < nch) stage(cur, ci + 1, cbuf ^ 1);
        else if (has_next) {  // the next item's first chunk arrives under this chunk and the epilogue
          Item nxt;           // (set up here and again after the epilogue: nothing of it stays live across the epilogue's registers)
          setup(wn, nxt);
          stage(nxt, 0, cbuf ^ 1);
        }
      };
      const char* inb = lds_in + cbuf * IN_B + a_lane;
      const char* wb = lds_w + cbuf * W_B + b_lane;
      auto load_tap = [&](int tap, bf16x8 (&a)[2], bf16x8 (&b)[NF]) __attribute__((always_inline)) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
          a[mf] = *reinterpret_cast<const bf16x8*>(inb + ((2 * mf + ky) * HALO_W + kx) * PIX_B + ((ky & 1) ? a_half_odd : a_half_even));
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) b[nf] = *reinterpret_cast<const bf16x8*>(wb + (tap * NB + nf * 32) * PIX_B);
      };
      bf16x8 fa[2][2], fb[2][NF];
      load_tap(0, fa[0], fb[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            acc[mf][nf] = MODE != PM_STATS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tap & 1][nf], fa[tap & 1][mf], acc[mf][nf], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tap & 1][mf], fb[tap & 1][nf], acc[mf][nf], 0, 0, 0);
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

    // ---- epilogue of the current item (the operand buffers are NOT free here: the next item's first chunk is landing in one) ----
    if constexpr (MODE == PM_BNB) {
      conv_epilogue_tr_bnb<NF, 1>(acc, p, n, nb, y0, x0, wave, r, kh, lds_small);  // (x quads per M fragment: the next item's state holds registers)
    } else if constexpr (MODE == PM_TR) {
      conv_epilogue_tr<NF, EPI_LINEAR>(acc, p, n, nb, y0, x0, wave, r, kh);
    } else {
      conv_epilogue<NF, EPI_LINEAR>(acc, p, n, nb, y0, x0, wave, r, kh, lds_small);
      __syncthreads();
      const size_t tile_lin = (size_t)(n * p.tiles_y + cur.ty) * p.tiles_x + cur.tx;
      for (int i = tid; i < 2 * NB; i += THREADS) p.stats[(tile_lin * p.stats_np + nb * NB + (i % NB)) * 2 + i / NB] = lds_small[i];
    }
    if (!has_next) break;
    w = wn;
    setup(w, cur);
  }
}

template <int MODE>
int launch_mode(const ConvParams& p, int nf, int nblk, int items, int grid, hipStream_t st) {
#define SF_PCASE(NFV) \
  case NFV: hipLaunchKernelGGL((conv3x3_bf16_persist_kernel<NFV, MODE>), dim3(grid), dim3(512), 0, st, p, items, nblk); break;
  switch (nf) {
    SF_PCASE(1) SF_PCASE(2) SF_PCASE(3) SF_PCASE(4)  // (NF = 5 is never dispatched - sf_conv_bf16_persist_ok - and its statistics variant spilled 256 bytes)
    default: sf_set_error("bf16 conv (persistent): unsupported nf=%d", nf); return 1;
  }
#undef SF_PCASE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error("conv3x3_bf16_persist: launch failed: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

}  // namespace

// Does this launch qualify (and pay: at least a few items per workgroup)?  Linear epilogue, single bf16-stored source without image
// remap, 8-wave tiles (H > 16).
bool sf_conv_bf16_persist_ok(const sfconv::ConvParams& p, int epi, int nf) {
  if (nf > 4) return false;  // (NF = 5 spills)
  if (epi != sfconv::EPI_LINEAR || p.H <= 16 || !p.src0 || !p.bf0 || p.src1 || p.idiv0 > 1 || p.imod0 > 0 || !p.out_bf) return false;
  // the BatchNorm-backward epilogue spills next to the persistent loop's state (340 bytes at NF = 4): measured in round 4, 2.88 ms against the
  // one-item kernel's 2.38 ms for the 256 -> 256 input gradient - one-item kernel
  // (the one-wave-per-SIMD kernel has the registers for it: conv3x3_bf16_persist4.hip, MODE 2)
  if (p.bnb_coef && !sf_conv_bf16_persist4_ok(p, nf)) return false;
  const int tiles = ((p.W + sfconv::TILE_W - 1) / sfconv::TILE_W) * ((p.H + 31) / 32) * p.N;
  static const char* mt = getenv("SF_PERSIST_MIN_TILES");   // experiment switch (round 5): the tile count from which the persistent kernels take a launch
  static const int min_tiles = mt ? atoi(mt) : 4 * 256;
  return tiles >= min_tiles;
}

int sf_launch_conv_bf16_persist(const sfconv::ConvParams& p0, int nf, int nblk, hipStream_t st) {
  sfconv::ConvParams p = p0;
  p.tiles_x = (p.W + sfconv::TILE_W - 1) / sfconv::TILE_W;
  p.tiles_y = (p.H + 31) / 32;
  const int items = p.tiles_x * p.tiles_y * p.N * nblk;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { sf_set_error("conv3x3_bf16_persist: device query failed"); return 2; }
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int grid = items < cus ? items : cus;
  if (sf_conv_bf16_persist4_ok(p, nf)) return sf_launch_conv_bf16_persist4(p, nblk, st);  // one wave per SIMD (conv3x3_bf16_persist4.hip)
  if (p.stats) return launch_mode<PM_STATS>(p, nf, nblk, items, grid, st);
  return launch_mode<PM_TR>(p, nf, nblk, items, grid, st);
}
