// Persistent ConvGRU: the recurrent half of ALL timesteps of a sequence in ONE launch, hidden state resident on chip.
//
// Replaces the per-step launches of sf_convgru_step_fwd (upstream metnet ConvGRU / ConvGRUCell; reference
// call site satflow/models/pl_metnet.py:46-59): for t = 0 .. T-1
//   [z_h | r_h | h2] = conv3x3(h_{t-1}) (+ bias on h2),  z = sig(z_x + z_h), r = sig(r_x + r_h),
//   n = tanh(n_x + r * h2),  h_t = (1 - z) * n + z * h_{t-1},            gx_t = [z_x | r_x | n_x] precomputed for all steps.
// One workgroup owns one image (maps of at most 16x16 pixels: MetNet's 16x16) for the whole sequence:
//   * the fp32 state h lives in REGISTERS (the lane that computes an element of h_t is the lane that needs it at t + 1) and is
//     never read back from memory; its bf16 image - the MFMA operand of the next step - lives in LDS as a halo tile whose
//     border stays zero ("same" padding);
//   * the recurrent weights (the packed image of sf_conv3x3_pack_weights with the GRU map, nf = 3) stream from L2 through a
//     two-stage LDS ring by LDS-DMA, one 16-channel K chunk at a time, both N blocks of the workgroup per stage;
//   * gx_t is requested at the start of step t and consumed by its epilogue (a whole K loop later);
//   * per step the only HBM traffic is gx_t in, h_t (and the saved gates, for the backward pass) out.
// Arithmetic is that of the per-step kernel (same K order, same bf16 rounding of h, same epilogue formulas): with fp32-stored
// gx the states are bit-identical to 24 launches of sf_convgru_step_fwd.
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace sfconv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int HALO = 18;                      // 16 + 2
constexpr int PIX_B = 32;                     // bytes per pixel / weight row in LDS (one 16-channel chunk)
constexpr int CHUNK_B = HALO * HALO * PIX_B;  // 10368: one K chunk of the state tile
constexpr int NBG = 96;                       // rows of one N block: z | r | h2 of 32 hidden channels
constexpr int W_B = 9 * NBG * PIX_B;          // 27648: weights of one (N block, chunk)
constexpr int PIECES_W = W_B / 1024;          // 27 DMA pieces of 1 KiB

struct GruSeqParams {
  const void* gx; int gx_s, gx_bf;        // [T][n][H][W][3*hidp] (fp32 or bf16)
  const float* h0; int h0_s;              // initial state [n][H][W][hidp] or null (zeros)
  float* hs; int hs_s;                    // all states [T][n][H][W][hidp]
  void* gates; int gates_s, gates_bf;     // saved z | r | n | h2, [T][n][H][W][4*hidp], or null
  const void* wp; const float* bias;      // packed weights [nblk][chunks][9][96][16] bf16; bias [nblk*96] or null
  int T, n, H, W, hidp, chunks;
};

// LDS-DMA hidden from hipcc (see conv3x3_bf16.hip): wave-uniform descriptor + scalar offset + constant per-lane offset.
__device__ __forceinline__ void bufdma16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff) : "memory");
}

__device__ __forceinline__ unsigned pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// NBLK: 32-channel N blocks of the hidden state (1: hidp <= 32, 2: hidp <= 64); 4 waves per N block, wave = (4-row band, N block).
template <int NBLK, bool GXBF>
__global__ __launch_bounds__(256 * NBLK, NBLK) void convgru_seq_fwd_kernel(const GruSeqParams p) {
  constexpr int WAVES = 4 * NBLK, THREADS = 64 * WAVES;
  constexpr int STAGE_B = NBLK * W_B;
  constexpr int MAXCH = 2 * NBLK;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_B + MAXCH * CHUNK_B];
  char* lds_h = lds + 2 * STAGE_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wl = wave & 3, nbk = wave >> 2;
  const int r = lane & 31, kh = lane >> 5;
  const int img = blockIdx.x;
  const int chunks = p.chunks;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  // zero the state tile (its halo border stays zero for the whole sequence)
  for (int i = tid; i < MAXCH * CHUNK_B / 16; i += THREADS) *reinterpret_cast<f32x4*>(lds_h + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- weights: one descriptor over the whole packed image; stage (t, ci) = chunk ci of every N block of this workgroup ----
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, NBLK * chunks * W_B, 0x00020000);
  auto issue_weights = [&](int ci, int buf) {
    for (int i = wave; i < NBLK * PIECES_W; i += WAVES) {
      const int nb = i / PIECES_W, pc = i - nb * PIECES_W;
      bufdma16(lane * 16, rs_w, (unsigned)((nb * chunks + ci) * W_B + pc * 1024), lds0 + (unsigned)(buf * STAGE_B + i * 1024));
    }
  };

  // ---- this lane's two pixels (one per M fragment) and its 16 hidden channels: quads at hb + 8g, g = 0..3 ----
  const int hb = nbk * 32 + 4 * kh;
  int py[2], px[2]; bool ok[2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) { py[mf] = 4 * wl + 2 * mf + (r >> 4); px[mf] = r & 15; ok[mf] = py[mf] < p.H && px[mf] < p.W; }
  const long long img_px = (long long)p.H * p.W;

  // state registers
  f32x4 hst[2][4];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      hst[mf][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.h0 && ok[mf] && hb + 8 * g < p.hidp)
        hst[mf][g] = *reinterpret_cast<const f32x4*>(p.h0 + ((long long)img * img_px + py[mf] * p.W + px[mf]) * p.h0_s + hb + 8 * g);
    }
  // bf16 image of a state into the LDS tile: after the half-wave swap a lane holds the channel octets hb8 + 16*(g/2) .. +7
  auto write_state_tile = [&]() {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const unsigned ax = pk(hst[mf][g][0], hst[mf][g][1]), ay = pk(hst[mf][g][2], hst[mf][g][3]);
        const unsigned bx = pk(hst[mf][g + 1][0], hst[mf][g + 1][1]), by = pk(hst[mf][g + 1][2], hst[mf][g + 1][3]);
        const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        const int ch = nbk * 32 + 8 * (g + kh);  // first channel of this lane's octet
        if (ok[mf] && ch < p.hidp) {
          const int iy = py[mf] + 1, ix = px[mf] + 1;
          char* dst = lds_h + (ch >> 4) * CHUNK_B + (iy * HALO + ix) * PIX_B + 16 * (((ch >> 3) & 1) ^ (iy & 1));
          *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
        }
      }
  };
  __syncthreads();  // the zero fill is complete
  if (p.h0) write_state_tile();

  // per-lane LDS offsets of the fragment reads (as in conv3x3_bf16.hip)
  const int rowpar = (r >> 4) & 1;
  const int a_lane = ((4 * wl + (r >> 4)) * HALO + (r & 15)) * PIX_B;
  const int a_half_even = 16 * (kh ^ rowpar), a_half_odd = 16 * (kh ^ rowpar ^ 1);
  const int b_lane = nbk * W_B + r * PIX_B + 16 * (kh ^ ((r >> 3) & 1));

  if (p.T > 0 && chunks > 0) issue_weights(0, 0);

  using GXV = typename std::conditional<GXBF, bf16x4, f32x4>::type;
  for (int t = 0; t < p.T; ++t) {
    f32x16 acc[2][3];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][g][i] = 0.f;

    // this step's x-part: bf16-stored, it is requested now and consumed by the epilogue a whole K loop later (48 registers);
    // fp32-stored, it would take 96 registers over the K loop and is fetched by the epilogue instead
    GXV gxv[2][3][4];
    const long long pix_t = ((long long)t * p.n + img) * img_px;
    auto load_gx = [&](int mf) {
    {
      const long long pix = pix_t + (ok[mf] ? py[mf] * p.W + px[mf] : 0);  // clamped: loads are unconditional
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int hq = hb + 8 * g < p.hidp ? hb + 8 * g : 0;
          if constexpr (GXBF) gxv[mf][q][g] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.gx) + pix * p.gx_s + q * p.hidp + hq);
          else gxv[mf][q][g] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.gx) + pix * p.gx_s + q * p.hidp + hq);
        }
    }
    };
    if constexpr (GXBF) { load_gx(0); load_gx(1); }

    for (int ci = 0; ci < chunks; ++ci) {
      const int it = t * chunks + ci, cur = it & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this chunk's weights have landed (and this step's x-part, last step's stores)
      __syncthreads();                                   // ... everybody's; the previous step's state tile writes are visible
      // the two waves of a SIMD (w, w + 4) issue the next chunk's DMA at different taps: one's MFMAs cover the other's issue stall
      const bool stage_late = NBLK == 2 && wave >= 4;
      const bool more = it + 1 < p.T * chunks;
      const int nci = ci + 1 < chunks ? ci + 1 : 0;
      if (more && !stage_late) issue_weights(nci, cur ^ 1);
      const char* inb = lds_h + ci * CHUNK_B + a_lane;
      const char* wb = lds + cur * STAGE_B + b_lane;
      auto load_tap = [&](int tap, bf16x8 (&a)[2], bf16x8 (&b)[3]) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
          a[mf] = *reinterpret_cast<const bf16x8*>(inb + ((2 * mf + ky) * HALO + kx) * PIX_B + ((ky & 1) ? a_half_odd : a_half_even));
#pragma unroll
        for (int g = 0; g < 3; ++g) b[g] = *reinterpret_cast<const bf16x8*>(wb + (tap * NBG + g * 32) * PIX_B);
      };
      bf16x8 fa[2][2], fb[2][3];
      load_tap(0, fa[0], fb[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int g = 0; g < 3; ++g)  // transposed product: D[channel][pixel] - a lane owns one pixel and channel quads
            acc[mf][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tap & 1][g], fa[tap & 1][mf], acc[mf][g], 0, 0, 0);
        if (tap == 3 && more && stage_late) issue_weights(nci, cur ^ 1);
      }
    }
    __syncthreads();  // every wave is done reading the state tile of step t - 1

    // ---- epilogue: gates, new state (registers), outputs, bf16 image of the new state into the tile ----
    f32x4 b2[4];  // bias of the candidate's h-part (z / r biases ride on the x-part)
#pragma unroll
    for (int g = 0; g < 4; ++g) b2[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nbk * NBG + 64 + 8 * g + 4 * kh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      if constexpr (!GXBF) load_gx(mf);
      const long long pix = pix_t + (ok[mf] ? py[mf] * p.W + px[mf] : 0);
      f32x4 zz[4], rr[4], nn[4], hh2[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gz = __builtin_convertvector(gxv[mf][0][g], f32x4), gr = __builtin_convertvector(gxv[mf][1][g], f32x4),
                    gn = __builtin_convertvector(gxv[mf][2][g], f32x4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float z = sf_sigmoid(acc[mf][0][4 * g + c] + gz[c]);
          const float rg = sf_sigmoid(acc[mf][1][4 * g + c] + gr[c]);
          const float h2 = acc[mf][2][4 * g + c] + b2[g][c];
          const float cand = sf_tanh(sf_gru_cand_arg(gn[c], rg, h2));
          zz[g][c] = z; rr[g][c] = rg; nn[g][c] = cand; hh2[g][c] = h2;
          hst[mf][g][c] = sf_gru_blend(z, cand, hst[mf][g][c]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (ok[mf] && hb + 8 * g < p.hidp) *reinterpret_cast<f32x4*>(p.hs + pix * p.hs_s + hb + 8 * g) = hst[mf][g];
      if (p.gates) {
        if (p.gates_bf) {
          __bf16* gp = reinterpret_cast<__bf16*>(p.gates) + pix * p.gates_s + nbk * 32 + 8 * kh;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4* v = q == 0 ? zz : q == 1 ? rr : q == 2 ? nn : hh2;
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
              const unsigned ax = pk(v[g][0], v[g][1]), ay = pk(v[g][2], v[g][3]);
              const unsigned bx = pk(v[g + 1][0], v[g + 1][1]), by = pk(v[g + 1][2], v[g + 1][3]);
              const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
              const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
              if (ok[mf] && nbk * 32 + 8 * (g + kh) < p.hidp) *reinterpret_cast<u32x4_t*>(gp + q * p.hidp + 8 * g) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            }
          }
        } else {
          float* gp = reinterpret_cast<float*>(p.gates) + pix * p.gates_s + hb;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (ok[mf] && hb + 8 * g < p.hidp) {
              *reinterpret_cast<f32x4*>(gp + 8 * g) = zz[g];
              *reinterpret_cast<f32x4*>(gp + p.hidp + 8 * g) = rr[g];
              *reinterpret_cast<f32x4*>(gp + 2 * p.hidp + 8 * g) = nn[g];
              *reinterpret_cast<f32x4*>(gp + 3 * p.hidp + 8 * g) = hh2[g];
            }
        }
      }
    }
    write_state_tile();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" int sf_convgru_seq_fwd(sfTensor gx, sfTensor h0, int32_t T, int32_t n, int32_t h, int32_t w, const void* wpacked,
                                  const float* bias_packed, int32_t hidp, sfTensor hs, sfTensor gates, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16, "sf_convgru_seq_fwd: the persistent sequence kernel is built for the SF_BF16 kernels (got dtype %d)", dtype);
  SF_REQUIRE(h >= 1 && w >= 1 && h <= 16 && w <= 16, "sf_convgru_seq_fwd: one workgroup owns a whole map: H, W <= 16 (got %dx%d)", h, w);
  SF_REQUIRE(hidp % SF_CPAD == 0 && hidp >= 16 && hidp <= 64, "sf_convgru_seq_fwd: hidp=%d (16..64)", hidp);
  SF_REQUIRE(gx.ptr && hs.ptr && wpacked && gx.c == 3 * hidp && hs.c == hidp && hs.dtype == SF_F32, "sf_convgru_seq_fwd: gx [..,3*hidp], hs [..,hidp] fp32");
  SF_REQUIRE((gx.dtype == SF_F32 || gx.dtype == SF_BF16) && ((uintptr_t)gx.ptr & 15) == 0 && gx.stride % (gx.dtype == SF_BF16 ? 8 : 4) == 0,
             "sf_convgru_seq_fwd: gx needs 16-byte aligned pixels");
  SF_REQUIRE(((uintptr_t)hs.ptr & 15) == 0 && hs.stride % 4 == 0 && (!h0.ptr || (h0.dtype == SF_F32 && ((uintptr_t)h0.ptr & 15) == 0 && h0.stride % 4 == 0 && h0.c == hidp)),
             "sf_convgru_seq_fwd: hs / h0 must be fp32 with 16-byte aligned pixels");
  SF_REQUIRE(!gates.ptr || ((gates.dtype == SF_F32 || gates.dtype == SF_BF16) && gates.c == 4 * hidp && ((uintptr_t)gates.ptr & 15) == 0 &&
                            gates.stride % (gates.dtype == SF_BF16 ? 8 : 4) == 0), "sf_convgru_seq_fwd: gates [..,4*hidp], 16-byte aligned pixels");
  if (T <= 0 || n <= 0) return 0;
  GruSeqParams p{};
  p.gx = gx.ptr; p.gx_s = gx.stride; p.gx_bf = gx.dtype == SF_BF16;
  p.h0 = (const float*)h0.ptr; p.h0_s = h0.stride;
  p.hs = (float*)hs.ptr; p.hs_s = hs.stride;
  p.gates = gates.ptr; p.gates_s = gates.stride; p.gates_bf = gates.ptr && gates.dtype == SF_BF16;
  p.wp = wpacked; p.bias = bias_packed;
  p.T = T; p.n = n; p.H = h; p.W = w; p.hidp = hidp; p.chunks = hidp / 16;
  const int nblk = (hidp + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
#define SF_GRU_SEQ(NB_, BF_) hipLaunchKernelGGL((convgru_seq_fwd_kernel<NB_, BF_>), dim3(n), dim3(256 * NB_), 0, st, p)
  if (nblk == 1) { if (p.gx_bf) SF_GRU_SEQ(1, true); else SF_GRU_SEQ(1, false); }
  else { if (p.gx_bf) SF_GRU_SEQ(2, true); else SF_GRU_SEQ(2, false); }
#undef SF_GRU_SEQ
  SF_CHECK_LAUNCH("convgru_seq_fwd");
  return 0;
}
