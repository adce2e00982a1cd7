// Weight gradient of the 3x3 same convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16):
//
//   dW[tap][co][ci] = sum over pixels  bf16(dout[pixel][co]) * bf16(in[pixel + tap][ci]),   fp32 accumulate
//
// GEMM view: M = co, N = ci, K = pixels.  The MFMA wants 8 consecutive K values per lane, i.e. 8
// consecutive PIXELS of one channel, while HBM holds NHWC fp32 (channel-contiguous).
//
// Workgroup = 8 waves with two roles (wave-uniform branch), one 128(co) x 32(ci) x 9(tap) slab of dW:
//   * 4 LOADER waves transpose in registers: a lane loads float4 (4 channels) of 8..9 consecutive pixels
//     of one image row (coalesced along the channel axis across lanes), converts to bf16 and writes one
//     16-byte run of 8 pixels per channel into channel-major LDS tiles of the NEXT K tile
//     (A: dout [co][8 rows][16 px];  B: in [ci][10 halo rows][16 px] + one halo pixel left/right);
//   * 4 COMPUTE waves (one per SIMD, co fragment w) run the CURRENT K tile halo-row by halo-row: one B read
//     feeds up to 9 MFMAs (3 taps ky x 3 taps kx) against a 3-row register window of dout fragments; the two
//     horizontally shifted taps are built from the aligned 16-byte run plus one neighbour dword with v_alignbit
//     - no shifted LDS copies.
// LDS layout (found by exhaustive search over pitch / xor patterns): the 4 channels of a quad sit in
// adjacent 16-byte slots, slot index xor-ed with (quad>>1)&3, quad pitch = 68 (A) / 84 (B) slots:
// conflict-free for the loaders' ds_write_b128 (8-lane groups along the quad axis) AND for the compute
// waves' ds_read_b128 lane groups.  Two LDS buffers, one barrier per K tile.
// Split-K partial slabs in the fp32 kernel's layout; the shared reduce kernel finishes (the bias gradient
// is summed in fp32 by the loaders from the unrounded values).
// BF = true: the tensors are already stored as bf16 - the loaders fetch 8-byte channel quads and transpose
// with v_perm_b32 (one per LDS dword) instead of converting.
#include <type_traits>

#include "wgrad_common.h"

namespace {

using namespace sfwgrad;

// operand type of the translation unit: __bf16, or _Float16 when included by conv3x3_wgrad_f16.hip (SF_OPERAND_F16; fp32-stored tensors only)
// SF_SPLIT3 (conv3x3_wgrad_f32e.hip; SF_F32E compute mode): fp32-equivalent products from three fp16 products.  Every K tile (pixel tile) is staged
// THREE times as virtual tiles - phase A: (hi(dout), lo'(in)) and (lo'(dout), hi(in)) of each real tile (loaded once, converted twice), the compute waves
// then scale their accumulators by 2^-11, phase B: (hi(dout), hi(in)); lo' = fp16((v - hi) * 2^11).  dout is a gradient: it is multiplied by the power of
// two that puts WgradParams::amax_dout at 2^14 before the split, the accumulators by the inverse before they are stored.  fp32-stored tensors only.
#ifdef SF_SPLIT3
#ifndef SF_OPERAND_F16
#error "SF_SPLIT3 is built on the fp16 operand type"
#endif
#define SF_OP_T _Float16
#define SF_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define sf_launch_wgrad_bf16 sf_launch_wgrad_f32e
#elif defined(SF_OPERAND_F16)
#define SF_OP_T _Float16
#define SF_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define sf_launch_wgrad_bf16 sf_launch_wgrad_f16
#else
#define SF_OP_T __bf16
#define SF_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef SF_OP_T bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KR = 8;                          // K-tile rows
constexpr int HR = KR + 2;                     // halo rows
constexpr int A_S = 68 * 16;                   // bytes per channel quad of the dout tile (1088)
constexpr int B_S = 84 * 16;                   // bytes per channel quad of the input tile (1344)
constexpr int H_P = HR * 8 + 4;                // bytes per channel of the halo-pixel array (84)
constexpr int A_BYTES = (CO_T / 4) * A_S;      // 34816
constexpr int B_BYTES = (CI_T / 4) * B_S;      // 10752
constexpr int H_BYTES = CI_T * H_P;            // 2688
constexpr int BUF = A_BYTES + B_BYTES + H_BYTES;  // 48256
constexpr int THREADS = 512, LOADERS = 256;

__device__ __forceinline__ int quad_slot(int cq, int c) { return (c ^ ((cq >> 1) & 3)) * 16; }

__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  f32x8 v = {a0, a1, a2, a3, a4, a5, a6, a7};
  return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ float F(unsigned bits) { return __builtin_bit_cast(float, bits); }
#ifdef SF_SPLIT3
constexpr float SPLIT_UP = 2048.f, SPLIT_DOWN = 1.f / 2048.f;
// the fp16 part of v * scale a virtual tile multiplies: hi = fp16(v s), or lo' = fp16((v s - hi) * 2^11)
__device__ __forceinline__ float part_f(float v, float scale, bool lo) {
  v *= scale;
  const float hi = (float)(SF_OP_T)v;
  return lo ? (v - hi) * SPLIT_UP : v;
}
__device__ __forceinline__ void split_scale(const float* amax, float& s, float& inv) {   // as in conv3x3_bf16.hip
  s = inv = 1.f;
  if (!amax) return;
  const unsigned bits = __builtin_bit_cast(unsigned, *amax);
  const int e = (int)((bits >> 23) & 0xffu) - 127;
  if ((bits & 0x7fffffffu) == 0u) return;
  int k = 14 - e;
  k = k > 126 ? 126 : (k < -126 ? -126 : k);
  s = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
  inv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
}
#endif
__device__ __forceinline__ unsigned bf16_bits(float x) {  // the 16 bits of x in the translation unit's operand type
  SF_OP_T b = (SF_OP_T)x;
  return (unsigned)__builtin_bit_cast(unsigned short, b);
}
// bf16 storage: pixel j of a channel quad is {ch0 | ch1 << 16, ch2 | ch3 << 16}.  One LDS dword of channel c holds pixels
// (2k, 2k+1): v_perm_b32 picks the low (even c) or high (odd c) halves of the two pixels' dwords.
__device__ __forceinline__ unsigned pair_bf(u32x2 even_px, u32x2 odd_px, int c) {
  const unsigned lo = (c & 2) ? even_px[1] : even_px[0], hi = (c & 2) ? odd_px[1] : odd_px[0];
  return __builtin_amdgcn_perm(hi, lo, (c & 1) ? 0x07060302u : 0x05040100u);
}
__device__ __forceinline__ unsigned chan_bits(u32x2 px, int c) {  // bf16 bits of channel c in the low half
  const unsigned d = (c & 2) ? px[1] : px[0];
  return (c & 1) ? d >> 16 : d & 0xffffu;
}

// ABF / BBF: storage type of dout (the A operand) / of the input sources (the B operand).  (true, false) is the ConvLSTM's
// "bf16a" combination: bf16-stored gate gradients against fp32 inputs and hidden states.
template <bool ABF, bool BBF, bool MIXED>
__global__ __launch_bounds__(THREADS, 2) void wgrad_bf16_kernel(const WgradParams p, const int xcd_groups) {
  using VA = u32x4;                                    // fp32 dout: one pixel's channel quad (raw bits); bf16 dout uses octets
  using VT = std::conditional_t<BBF, u32x2, u32x4>;   // input source: one pixel's channel quad as loaded (raw bits)
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, kh = lane >> 5;

  // block id -> (ks, cot, cit): the (cot, cit) combinations of one K slice are consecutive on one XCD
  // (blocks are dealt round-robin to the 8 XCDs), so the slice's tiles are shared through that XCD's L2.
  const int combos = gridDim.y * gridDim.z;
  int ks, cot, cit;
  {
    const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (xcd_groups) {
      const int x = id % 8, j = id / 8;
      ks = (j / combos) * 8 + x;
      const int combo = j % combos;
      cot = combo % gridDim.y; cit = combo / gridDim.y;
    } else { ks = blockIdx.x; cot = blockIdx.y; cit = blockIdx.z; }
  }
#ifdef SF_SPLIT3
  const int real_tiles = ks < p.ntiles ? (p.ntiles - ks + p.KS - 1) / p.KS : 0;
  const int my_tiles = 3 * real_tiles;   // virtual tiles: 2 t, 2 t + 1 = phase A of real tile t; 2 real_tiles + t = phase B
  float sa, sa_inv;                      // dout's power-of-two scale
  split_scale(p.amax_dout, sa, sa_inv);
  auto v_real = [&](int i) { return i < 2 * real_tiles ? i >> 1 : i - 2 * real_tiles; };
#else
  const int my_tiles = ks < p.ntiles ? (p.ntiles - ks + p.KS - 1) / p.KS : 0;
#endif

  if (wave >= 4) {
    // =========================== loader waves ===========================
    // The loaders' load -> transpose -> LDS-store chain sets the pace of a K tile; on their SIMD they win the issue arbitration
    // against the compute wave's MFMA / ds_read stream (measured 256->256: 2.85 -> 2.58 ms; the opposite priority: no change).
    __builtin_amdgcn_s_setprio(3);
    const int lt = tid - LOADERS;
    const int a_cq = lt % 32, b_cq = lt % 8;
    const int kcb = cit * CI_T + b_cq * 4;  // input channel (concatenated padded K space)
    const int aco = cot * CO_T + a_cq * 4;
    const bool a_ok = aco < p.dc;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // Register-staged software pipeline: global loads stay in flight across the barrier, i.e. during the compute waves' work.
    // Loads are raw BUFFER loads: one descriptor per tensor image (scalar, rebuilt per tile), a per-lane byte offset that
    // is constant for the whole kernel, and the tile / pixel position in the scalar offset - no per-load 64-bit VALU
    // address arithmetic (that, not bandwidth, set the loaders' pace).  Out-of-image pixels get an out-of-range
    // voffset and come back as zeros from the hardware range check, so the loads are unconditional and countable
    // (s_waitcnt vmcnt(N)): BF (8-byte quads, half the registers) keeps two tiles in flight.
    const bool b_item = lt < 8 * HR * 2;
    const int brest = lt / 8, bhalf = brest & 1, bhrow = brest >> 1;
    constexpr unsigned SENT = 0x80000000u;  // >= any descriptor's num_records (host checks image bytes < 2^31)
    constexpr int ESZ = BBF ? 2 : 4, ESZ_A = ABF ? 2 : 4;
    const unsigned a_px = (unsigned)p.ds * ESZ_A;
    unsigned a_const[2]; int a_row[2], a_half[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rest = (lt + u * LOADERS) / 32;
      a_half[u] = rest & 1; a_row[u] = rest >> 1;
      a_const[u] = (unsigned)((a_row[u] * p.W + 8 * a_half[u]) * p.ds + aco) * ESZ_A;
    }
    // The block's 32 input channels come from ONE source unless c0 is not a multiple of 32 (MIXED kernels: the ConvLSTM's
    // 16-lane x source).  Plain kernels pick the source with scalar selects - no branch near the loads; MIXED kernels
    // load from both descriptors (a lane's foreign source gets the out-of-range offset, a missing source a
    // zero-length descriptor) and OR the two results when the tile is written to LDS.
    const bool from0 = cit * CI_T < p.c0;  // block-uniform; exact for plain kernels
    const bool lane0 = b_item && kcb < p.c0, lane1 = b_item && kcb >= p.c0 && kcb - p.c0 < p.c1;
    const unsigned b_px0 = (unsigned)p.s0 * ESZ, b_px1 = (unsigned)p.s1 * ESZ;
    // four shifted views of src0 (WgradParams::shift4; fp32 tensors, plain kernel): this block's view, its pixel shift, the channel inside the view
    constexpr bool SHIFT_OK = !ABF && !BBF && !MIXED;
    const int sview = (SHIFT_OK && p.shift4) ? (cit * CI_T) / p.shift4 : 0;
    const int sdy = (SHIFT_OK && p.shift4) ? 2 * (sview >> 1) - 1 : 0, sdx = (SHIFT_OK && p.shift4) ? 2 * (sview & 1) - 1 : 0;
    const int kreal = (SHIFT_OK && p.shift4) ? kcb - sview * p.shift4 : kcb;
    const unsigned b_const0 = (unsigned)((bhrow * p.W + 8 * bhalf) * p.s0 + kreal) * ESZ;
    const unsigned b_const1 = (unsigned)((bhrow * p.W + 8 * bhalf) * p.s1 + (kcb - p.c0)) * ESZ;
    // bf16 storage: dout is fetched as 16-byte channel OCTETS, one item (octet 16, row 8, half 2) per loader thread - half
    // the load instructions of the quad form for the same registers; fp32 storage keeps two quad items per thread.
    struct Stage { VA va[ABF ? 1 : 2][ABF ? 1 : 8]; u32x4 va8[ABF ? 8 : 1]; VT vb[10], vb1[MIXED ? 10 : 1]; };
    const int o_oct = lt % 16, o_rest = lt / 16, o_half = o_rest & 1, o_row = o_rest >> 1;
    const int oco = cot * CO_T + o_oct * 8;
    const unsigned o_const = (unsigned)((o_row * p.W + 8 * o_half) * p.ds + oco) * ESZ_A;
    float bsum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto ld = [](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) -> VT {
      if constexpr (BBF) return __builtin_bit_cast(VT, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
      else return __builtin_bit_cast(VT, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    // descriptor of image `ns` of a source, starting one image row + one pixel BEFORE the image so that lane offsets are
    // non-negative (nothing is read there); a null source gives a zero-length descriptor (every load returns zeros)
    auto halo_srd = [&](const float* src, int n, int idiv, int imod, unsigned px) {
      int ns = n / idiv; if (imod) ns %= imod;
      // (shifted views: two rows + two pixels of lead, so that the view displaced by (-1, -1) still has non-negative offsets)
      const int lead = (SHIFT_OK && p.shift4) ? 2 * (p.W + 1) : p.W + 1;
      const char* base = (const char*)src + ((long long)ns * p.H * p.W - lead) * px;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(src ? base : nullptr), 0, src ? (int)((unsigned)(p.H * p.W + lead) * px) : 0, 0x00020000);
    };
    auto load_tile = [&](int i, Stage& s) {
#ifdef SF_SPLIT3
      if (i < 2 * real_tiles && (i & 1)) return;   // second phase-A tile of a real tile: the registers already hold it
      i = v_real(i);
#endif
      int t = ks + i * p.KS;
      const int tx = t % p.tiles_x; t /= p.tiles_x;
      const int ty = t % p.tiles_y;
      const int n = t / p.tiles_y;
      const int x0 = tx * KT_W, y0 = ty * KR;
      const unsigned tile_px = (unsigned)(y0 * p.W + x0);
      // dout tile: items (cq 32, row 8, half 2) -> item = lt and lt + 256
      {
        const char* base = (const char*)p.dout + (long long)n * p.H * p.W * a_px;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((unsigned)(p.H * p.W) * a_px), 0x00020000);
        const unsigned so = tile_px * a_px;
        if constexpr (ABF) {
          const int lim = (oco < p.dc && y0 + o_row < p.H) ? p.W - x0 - 8 * o_half : 0;  // pixels j < lim are inside the image
#pragma unroll
          for (int j = 0; j < 8; ++j)
            s.va8[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, j < lim ? o_const : SENT, so + j * a_px, 0));
        } else {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int lim = (a_ok && y0 + a_row[u] < p.H) ? p.W - x0 - 8 * a_half[u] : 0;  // pixels j < lim are inside the image
#pragma unroll
            for (int j = 0; j < 8; ++j)
              s.va[u][j] = __builtin_bit_cast(VA, __builtin_amdgcn_raw_buffer_load_b128(rs, j < lim ? a_const[u] : SENT, so + j * a_px, 0));
          }
        }
      }
      // input halo tile: items (cq 8, halo row 10, half 2) = 160 items; pixels x0-1+8*half .. +9
      const int gy = y0 + bhrow - 1 + sdy, gx0 = x0 - 1 + 8 * bhalf + sdx;
      const unsigned shift_px = (SHIFT_OK && p.shift4) ? (unsigned)((sdy + 1) * p.W + sdx + 1) : 0u;  // the displacement, on top of the longer lead
      const bool rok = gy >= 0 && gy < p.H;
      const int lo = rok ? -gx0 : 99, hi = rok ? p.W - gx0 : 0;  // pixels lo <= j < hi are inside the image
      if constexpr (MIXED) {
        const __amdgpu_buffer_rsrc_t rs0 = halo_srd(p.src0, n, p.idiv0, p.imod0, b_px0), rs1 = halo_srd(p.src1, n, p.idiv1, p.imod1, b_px1);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const bool in = j >= lo && j < hi;
          s.vb[j] = ld(rs0, (lane0 && in) ? b_const0 : SENT, (tile_px + j) * b_px0);
          s.vb1[j] = ld(rs1, (lane1 && in) ? b_const1 : SENT, (tile_px + j) * b_px1);
        }
      } else {
        const unsigned px = from0 ? b_px0 : b_px1, bc = from0 ? b_const0 : b_const1;
        const bool lane = from0 ? lane0 : lane1;
        const __amdgpu_buffer_rsrc_t rs = halo_srd(from0 ? p.src0 : p.src1, n, from0 ? p.idiv0 : p.idiv1, from0 ? p.imod0 : p.imod1, px);
#pragma unroll
        for (int j = 0; j < 10; ++j) s.vb[j] = ld(rs, (lane && j >= lo && j < hi) ? bc : SENT, (tile_px + shift_px + j) * px);
      }
    };
    auto store_tile = [&](int i, Stage& s) {
#ifdef SF_SPLIT3
      const bool phase_b = i >= 2 * real_tiles;
      const bool a_lo = !phase_b && (i & 1), b_lo = !phase_b && !(i & 1);   // (hi(dout), lo'(in)), (lo'(dout), hi(in)), then (hi, hi)
#endif
      char* la = lds + (i & 1) * BUF;
      char* lb = la + A_BYTES;
      char* lh = lb + B_BYTES;
      if constexpr (ABF) {
        const u32x4 (&v8)[8] = s.va8;
        if (cit == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
              const unsigned d = v8[j][c8 >> 1];
              bsum8[c8] += __builtin_bit_cast(float, (c8 & 1) ? (d & 0xffff0000u) : (d << 16));
            }
        }
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
          const int quad = 2 * o_oct + (c8 >> 2), d = c8 >> 1;
          const unsigned sel = (c8 & 1) ? 0x07060302u : 0x05040100u;
          char* dst = la + quad * A_S + (o_row * 2 + o_half) * 64 + quad_slot(quad, c8 & 3);
          *reinterpret_cast<u32x4*>(dst) = u32x4{__builtin_amdgcn_perm(v8[1][d], v8[0][d], sel), __builtin_amdgcn_perm(v8[3][d], v8[2][d], sel),
                                                 __builtin_amdgcn_perm(v8[5][d], v8[4][d], sel), __builtin_amdgcn_perm(v8[7][d], v8[6][d], sel)};
        }
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int rest = (lt + u * LOADERS) / 32, half = rest & 1, row = rest >> 1;
          char* dst = la + a_cq * A_S + (row * 2 + half) * 64;
          const VA (&va)[8] = s.va[u];
#ifdef SF_SPLIT3
          if (cit == 0 && phase_b) {  // the bias gradient (fp32 sum of the unrounded values) once per real tile
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += __builtin_bit_cast(f32x4, va[j]);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
            *reinterpret_cast<bf16x8*>(dst + quad_slot(a_cq, c)) =
                pack8(part_f(F(va[0][c]), sa, a_lo), part_f(F(va[1][c]), sa, a_lo), part_f(F(va[2][c]), sa, a_lo), part_f(F(va[3][c]), sa, a_lo),
                      part_f(F(va[4][c]), sa, a_lo), part_f(F(va[5][c]), sa, a_lo), part_f(F(va[6][c]), sa, a_lo), part_f(F(va[7][c]), sa, a_lo));
#else
          if (cit == 0) {  // block-uniform: only the first ci tile's blocks report the bias gradient
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += __builtin_bit_cast(f32x4, va[j]);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
            *reinterpret_cast<bf16x8*>(dst + quad_slot(a_cq, c)) =
                pack8(F(va[0][c]), F(va[1][c]), F(va[2][c]), F(va[3][c]), F(va[4][c]), F(va[5][c]), F(va[6][c]), F(va[7][c]));
#endif
        }
      }
      if (b_item) {
        char* dst = lb + b_cq * B_S + (bhrow * 2 + bhalf) * 64;
        VT vb[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          if constexpr (MIXED) vb[j] = s.vb[j] | s.vb1[j];  // a lane belongs to one source; the other load returned zeros
          else vb[j] = s.vb[j];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // halo pixels: left neighbour in the HIGH half of its dword, right neighbour in the LOW half
          unsigned* hp = reinterpret_cast<unsigned*>(lh + (b_cq * 4 + c) * H_P + bhrow * 8);
          if constexpr (BBF) {
            *reinterpret_cast<u32x4*>(dst + quad_slot(b_cq, c)) = u32x4{pair_bf(vb[1], vb[2], c), pair_bf(vb[3], vb[4], c),
                                                                        pair_bf(vb[5], vb[6], c), pair_bf(vb[7], vb[8], c)};
            hp[bhalf] = bhalf ? chan_bits(vb[9], c) : chan_bits(vb[0], c) << 16;
          } else {
#ifdef SF_SPLIT3
            *reinterpret_cast<bf16x8*>(dst + quad_slot(b_cq, c)) =
                pack8(part_f(F(vb[1][c]), 1.f, b_lo), part_f(F(vb[2][c]), 1.f, b_lo), part_f(F(vb[3][c]), 1.f, b_lo), part_f(F(vb[4][c]), 1.f, b_lo),
                      part_f(F(vb[5][c]), 1.f, b_lo), part_f(F(vb[6][c]), 1.f, b_lo), part_f(F(vb[7][c]), 1.f, b_lo), part_f(F(vb[8][c]), 1.f, b_lo));
            hp[bhalf] = bhalf ? bf16_bits(part_f(F(vb[9][c]), 1.f, b_lo)) : bf16_bits(part_f(F(vb[0][c]), 1.f, b_lo)) << 16;
#else
            *reinterpret_cast<bf16x8*>(dst + quad_slot(b_cq, c)) =
                pack8(F(vb[1][c]), F(vb[2][c]), F(vb[3][c]), F(vb[4][c]), F(vb[5][c]), F(vb[6][c]), F(vb[7][c]), F(vb[8][c]));
            hp[bhalf] = bhalf ? bf16_bits(F(vb[9][c])) : bf16_bits(F(vb[0][c])) << 16;
#endif
          }
        }
      }
    };
    // my_tiles + 1 barriers in every variant (the compute waves count the same)
    if constexpr (ABF && !MIXED) {  // octet dout + quad inputs: two stages fit the register budget for either input type
      Stage s0, s1;
      if (my_tiles > 0) load_tile(0, s0);
      if (my_tiles > 1) load_tile(1, s1);
      for (int i = 0; i <= my_tiles; i += 2) {
        if (i < my_tiles) store_tile(i, s0);
        if (i + 2 < my_tiles) load_tile(i + 2, s0);
        __syncthreads();
        if (i + 1 <= my_tiles) {
          if (i + 1 < my_tiles) store_tile(i + 1, s1);
          if (i + 3 < my_tiles) load_tile(i + 3, s1);
          __syncthreads();
        }
      }
    } else {
      Stage s0;
      if (my_tiles > 0) load_tile(0, s0);
      for (int i = 0; i <= my_tiles; ++i) {
        if (i < my_tiles) store_tile(i, s0);
        if (i + 1 < my_tiles) load_tile(i + 1, s0);
        __syncthreads();
      }
    }
    // bias gradient: 8 loader threads share a channel quad
    if (cit == 0) {
      float* red = reinterpret_cast<float*>(lds);
      if constexpr (ABF) {  // [rest 16][channel 128]
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) red[o_rest * CO_T + o_oct * 8 + c8] = bsum8[c8];
      } else {
        *reinterpret_cast<f32x4*>(red + lt * 4) = bsum;
      }
    }
    __syncthreads();
  } else {
    // =========================== compute waves ===========================
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int co_l = 32 * wave + r;  // co within the tile
    const int a_off = (co_l >> 2) * A_S + quad_slot(co_l >> 2, co_l & 3) + kh * 64;
    const int b_q = (r >> 2) * B_S + quad_slot(r >> 2, r & 3);
    const int h_off = r * H_P;

    // neighbour dwords of the 16-byte run: kh = 0 -> left halo dword / first dword of the upper half;
    // kh = 1 -> last dword of the lower half / right halo dword.  One per-lane address + per-lane row pitch each.
    const int l_off = kh ? A_BYTES + b_q + 12 : A_BYTES + B_BYTES + h_off, l_pitch = kh ? 128 : 8;
    const int r_off = kh ? A_BYTES + B_BYTES + h_off + 4 : A_BYTES + b_q + 64, r_pitch = kh ? 8 : 128;

    // shifted views (WgradParams::shift4): view (ty, tx) = 2 ty + tx of the 5x5 kernel's four 3x3 tiles owns no taps in its first row when ty = 1 and none
    // in its first column when tx = 1 (sf_regroup5x5_fwd masks them: the 5x5 kernel's middle row / column belongs to the upper / left tile) - their
    // products are skipped, the accumulators stay zero: 25 instead of 36 taps' worth of MFMAs over the four views
    const int view_c = (!ABF && !BBF && !MIXED && p.shift4) ? (cit * CI_T) / p.shift4 : 0;
    const bool dead_ky0 = (view_c >> 1) != 0, dead_kx0 = (view_c & 1) != 0;
    __syncthreads();  // tile 0 staged
    for (int i = 0; i < my_tiles; ++i) {
#ifdef SF_SPLIT3
      if (i == 2 * real_tiles) {   // phase A -> phase B: the correction terms were accumulated at 2^11 times their value
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[t][k] *= SPLIT_DOWN;
      }
#endif
      const char* la = lds + (i & 1) * BUF;
      const char* lb = la + A_BYTES;
      // Halo-row major: input row hrow is read and shifted ONCE and feeds the taps ky = 0..2 of the output rows
      // hrow, hrow-1, hrow-2, whose dout fragments sit in a 3-row register window (72 MFMAs per tile either way,
      // but 10 instead of 24 B-row reads / shift sequences).
      // Software pipeline over the (fully unrolled) halo rows: the LDS reads of row hrow+1 are requested before the MFMAs of
      // row hrow.  Written row by row (read, shift, 9 MFMAs) hipcc keeps each row's four reads next to their first use,
      // followed at once by `s_waitcnt lgkmcnt`, and with ONE compute wave per SIMD every such wait is idle matrix-pipe
      // time (256->256: 2.53 -> 2.45 ms; forcing a one-read-per-MFMA interleave with sched_group_barrier on top of this,
      // or a third stage for the shifts, gained nothing more).  The dout window holds 4 rows: 3 in use + the one in flight.
      bf16x8 arow[4];
      u32x4 mid_n; unsigned left_n, right_n;
      auto load_row = [&](int hrow) {
        if (hrow < KR) arow[hrow & 3] = *reinterpret_cast<const bf16x8*>(la + a_off + hrow * 128);
        mid_n = *reinterpret_cast<const u32x4*>(lb + b_q + hrow * 128 + kh * 64);
        left_n = *reinterpret_cast<const unsigned*>(la + l_off + hrow * l_pitch);
        right_n = *reinterpret_cast<const unsigned*>(la + r_off + hrow * r_pitch);
      };
      load_row(0);
#pragma unroll
      for (int hrow = 0; hrow < HR; ++hrow) {
        const u32x4 mid = mid_n;
        const unsigned left = left_n, right = right_n;
        if (hrow + 1 < HR) load_row(hrow + 1);
        u32x4 b0, b2;
        b0[0] = __builtin_amdgcn_alignbit(mid[0], left, 16);
        b0[1] = __builtin_amdgcn_alignbit(mid[1], mid[0], 16);
        b0[2] = __builtin_amdgcn_alignbit(mid[2], mid[1], 16);
        b0[3] = __builtin_amdgcn_alignbit(mid[3], mid[2], 16);
        b2[0] = b0[1];
        b2[1] = b0[2];
        b2[2] = b0[3];
        b2[3] = __builtin_amdgcn_alignbit(right, mid[3], 16);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int row = hrow - ky;
          if (row < 0 || row >= KR) continue;
          if (ky == 0 && dead_ky0) continue;
          const bf16x8 a = arow[row & 3];
          acc[ky * 3 + 1] = SF_MFMA_32X32X16(a, __builtin_bit_cast(bf16x8, mid), acc[ky * 3 + 1]);
          if (!dead_kx0) acc[ky * 3 + 0] = SF_MFMA_32X32X16(a, __builtin_bit_cast(bf16x8, b0), acc[ky * 3 + 0]);
          acc[ky * 3 + 2] = SF_MFMA_32X32X16(a, __builtin_bit_cast(bf16x8, b2), acc[ky * 3 + 2]);
        }
      }
      __syncthreads();
    }
#ifdef SF_SPLIT3
    if (p.amax_dout) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] *= sa_inv;
    }
#endif
    // partial[ks][tap][co][ci]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = cot * CO_T + 32 * wave + frag_row(reg, kh);
        const int ci = cit * CI_T + r;
        p.partial[(((size_t)ks * 9 + tap) * p.NpT + co) * p.KpT + ci] = acc[tap][reg];
      }
    __syncthreads();  // loaders have parked their bias sums
    if (cit == 0 && tid < 128) {
      const float* red = reinterpret_cast<const float*>(lds);
      const int cq = tid / 4, c = tid % 4;
      float s = 0.f;
      if constexpr (ABF) {
        for (int k = 0; k < LOADERS / 16; ++k) s += red[k * CO_T + tid];
      } else {
        for (int k = 0; k < LOADERS / 32; ++k) s += red[(k * 32 + cq) * 4 + c];
      }
      p.partial_db[(size_t)ks * p.NpT + cot * CO_T + tid] = s;
    }
  }
}

}  // namespace

int sf_launch_wgrad_bf16(const sfwgrad::WgradParams& p, const sfwgrad::Plan& pl, hipStream_t st) {
  if ((((uintptr_t)p.dout) & 15) || p.ds % 4 || p.dc % 4 || (p.src0 && ((((uintptr_t)p.src0) & 15) || p.s0 % 4)) ||
      (p.src1 && ((((uintptr_t)p.src1) & 15) || p.s1 % 4))) {
    sf_set_error("wgrad_bf16: tensors must be 16-byte aligned with 4-aligned strides");
    return 1;
  }
  {
    const long long esz = 4, px = (long long)p.H * p.W + p.W + 1;  // bound for either storage type
    const long long smax = p.ds > p.s0 ? (p.ds > p.s1 ? p.ds : p.s1) : (p.s0 > p.s1 ? p.s0 : p.s1);
    if (px * smax * esz >= (1ll << 31)) { sf_set_error("wgrad_bf16: one image of a tensor must be smaller than 2 GiB"); return 1; }
  }
#ifdef SF_SPLIT3
  if (p.bf || p.bf_dout || p.shift4) { sf_set_error("wgrad_f32e: fp32-stored tensors, no shifted views"); return 1; }
#endif
  const int xcd_groups = (pl.KS % 8 == 0) ? 1 : 0;
  const bool mixed = p.src0 && p.src1 && p.c1 > 0 && p.c0 % CI_T != 0;  // some block's 32 input channels straddle the two sources
  const dim3 grid(pl.KS, pl.cot, pl.cit), block(THREADS);
#define SF_WG(A_, B_)                                                                                             \
  do {                                                                                                            \
    if (mixed) hipLaunchKernelGGL((wgrad_bf16_kernel<A_, B_, true>), grid, block, 0, st, p, xcd_groups);          \
    else hipLaunchKernelGGL((wgrad_bf16_kernel<A_, B_, false>), grid, block, 0, st, p, xcd_groups);               \
  } while (0)
  if (p.bf_dout && p.bf) SF_WG(true, true);
  else if (p.bf_dout) SF_WG(true, false);
  else if (!p.bf) SF_WG(false, false);
  else { sf_set_error("wgrad_bf16: bf16-stored inputs with an fp32-stored output gradient are not built"); return 1; }
#undef SF_WG
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { sf_set_error("wgrad_bf16: launch failed: %s", hipGetErrorString(e)); return 2; }
  return 0;
}
