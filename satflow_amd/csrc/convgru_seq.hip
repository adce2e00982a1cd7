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
#include <cstdlib>
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
  unsigned long long* mbox;               // SPLIT kernel: boundary-row mailbox (see convgru_seq_fwd_kernel), zeroed before the launch
  unsigned spin_limit; int mute_half;     // polls before a receiver gives up; test hook: this half never sends (sf_convgru_seq_debug)
};

// SPLIT: boundary-row hand-off between the two workgroups of an image.  One row of the bf16 state = chunks x 16 pixels x 2 octets x
// 4 dwords; every dword travels as an 8-byte {tag = epoch, value} granule written by ONE write-through store - the data is the flag
// (cdna_hip_programming.md Guideline 16, form R2): no fence, no separate flag.  Slots alternate with the epoch's parity: a sender
// can be at most one state ahead of its receiver.
// Workspace layout (u64 words): [0] STICKY error word (zeroed by the caller when it allocates the workspace, only ever OR-ed into by the
// kernels, read by the host mirror: satflow_amd.device_errors()), [1] ticket counter, [2 ..] the granule slots.  Words 1.. are zeroed
// by the library before every launch.
// Residency: nothing is assumed.  A workgroup takes a TICKET when it starts (ticket k -> map k / 2, half k % 2), so partners are
// always two workgroups that the hardware has actually started, in start order: the holder of the highest issued ticket K either has
// its partner running (K odd) or waits for ticket K + 1, which the next free slot receives while every other pair keeps making
// progress.  The launch completes whenever at least two workgroups can be resident at a time - whatever else holds CUs (an RCCL
// kernel on the exchange stream, another process, a CU mask).  The spin is bounded anyway; a receiver that gives up sets the error
// word AND poisons its workgroup's state with NaN, so that a failed hand-off can never pass as a result.
constexpr int MB_HDR = 2;                                    // error word + ticket counter
constexpr int MB_ROW = 4 * 16 * 2 * 4;                       // granules of one boundary row (4 chunks max)
__host__ __device__ constexpr long long mbox_slot(long long img, int half, int parity) { return MB_HDR + ((img * 2 + half) * 2 + parity) * MB_ROW; }
constexpr unsigned MB_SPIN_LIMIT = 1u << 22;                 // default polls before a receiver gives up (seconds; then: error word + NaN state)
// The product library has NO process-wide state here: spin bound and muted half are constants.  tests/native builds this file a second time
// with -DSF_TEST_HOOKS (libsatflow_hip_hooks.so, never shipped): a shorter spin and a half that never sends, to exercise the failure path.
#ifdef SF_TEST_HOOKS
int g_spin_limit = (int)MB_SPIN_LIMIT, g_mute_half = -1;
#else
constexpr int g_spin_limit = (int)MB_SPIN_LIMIT, g_mute_half = -1;
#endif

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
// SPLIT (NBLK = 2): TWO workgroups per image, each owning 8 rows (192 workgroups instead of 96 on the 256 CUs for MetNet); a wave
// then has ONE M fragment (32 pixels), wave = (4-row band, M fragment of the band, N block): still 8 waves, two per SIMD.  After
// every step a workgroup hands the bf16 image of its boundary row to its partner through the mailbox and reads the partner's
// into its halo row; everything else (K order, roundings, epilogue formulas, the lane that owns an element) is unchanged, so the
// states stay bit-identical to the per-step kernel.
// F4 (SPLIT, GXBF): the counted-wait time loop for hidp = 64 with bf16 gates (see chunk4) - its own instantiation, chosen by the launcher.
template <int NBLK, bool GXBF, bool SPLIT = false, bool F4 = false>
__global__ __launch_bounds__(256 * NBLK, NBLK) void convgru_seq_fwd_kernel(const GruSeqParams p) {
  static_assert(!SPLIT || NBLK == 2, "the split kernel is the 8-wave layout");
  static_assert(!F4 || (SPLIT && GXBF), "the counted-wait loop is the split kernel with a bf16 x-part");
  constexpr int MFW = SPLIT ? 1 : 2;   // M fragments per wave
  constexpr int WAVES = 4 * NBLK, THREADS = 64 * WAVES;
  constexpr int STAGE_B = NBLK * W_B;
  constexpr int MAXCH = 2 * NBLK;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_B + MAXCH * CHUNK_B];
  char* lds_h = lds + 2 * STAGE_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbk = wave >> 2;
  const int r = lane & 31, kh = lane >> 5;
  __shared__ unsigned s_ctl[2];  // SPLIT: [0] this workgroup's ticket, [1] "a hand-off timed out"
  if constexpr (SPLIT) {
    if (tid == 0) { s_ctl[0] = atomicAdd(reinterpret_cast<unsigned*>(p.mbox + 1), 1u); s_ctl[1] = 0u; }
    __syncthreads();
  }
  const int ticket = SPLIT ? __builtin_amdgcn_readfirstlane((int)s_ctl[0]) : (int)blockIdx.x;
  const int img = SPLIT ? ticket >> 1 : ticket;
  const int half = SPLIT ? ticket & 1 : 0;
  // first tile row of this wave's fragments: 4-row band wl of the map (SPLIT: band (wave & 3) >> 1 of this half, fragment wave & 1)
  const int wrow = SPLIT ? 8 * half + 4 * ((wave & 3) >> 1) + 2 * (wave & 1) : 4 * (wave & 3);
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
  int py[MFW], px[MFW]; bool ok[MFW];
#pragma unroll
  for (int mf = 0; mf < MFW; ++mf) { py[mf] = wrow + 2 * mf + (r >> 4); px[mf] = r & 15; ok[mf] = py[mf] < p.H && px[mf] < p.W; }
  const long long img_px = (long long)p.H * p.W;
  // SPLIT: the row this workgroup sends (its last / first row) and the halo row it receives (the partner's first / last row)
  const int send_row = half == 0 ? 7 : 8, recv_row = half == 0 ? 8 : 7;
  const bool has_partner = SPLIT && p.H > 8;
  unsigned mb_failed = 0;

  // state registers
  f32x4 hst[MFW][4];
#pragma unroll
  for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      hst[mf][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.h0 && ok[mf] && hb + 8 * g < p.hidp)
        hst[mf][g] = *reinterpret_cast<const f32x4*>(p.h0 + ((long long)img * img_px + py[mf] * p.W + px[mf]) * p.h0_s + hb + 8 * g);
    }
  // bf16 image of a state into the LDS tile: after the half-wave swap a lane holds the channel octets hb8 + 16*(g/2) .. +7
  auto write_state_tile = [&](unsigned epoch) {
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
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
          if constexpr (SPLIT) {
#ifndef SF_EXP_GRU_NOSEND
            if (has_partner && py[mf] == send_row && half != p.mute_half)
#else
            if (false)
#endif
            {  // the partner's halo row: four {epoch, dword} granules, write-through
              unsigned long long* g8 = p.mbox + mbox_slot(img, half, epoch & 1) + (ch >> 4) * 128 + px[mf] * 8 + ((ch >> 3) & 1) * 4;
              const unsigned long long tag = (unsigned long long)epoch << 32;
              __hip_atomic_store(g8 + 0, tag | sx[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(g8 + 1, tag | sy[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(g8 + 2, tag | sx[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(g8 + 3, tag | sy[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
  };
  // SPLIT: wave 0 sweeps the partner's granules of `epoch` (8 per lane) until every tag matches, then writes them into the halo row
  auto receive_row = [&](unsigned epoch) {
#ifdef SF_EXP_GRU_NOPOLL
    return;
#endif
    if constexpr (SPLIT) {
      if (has_partner && wave == 0) {
        const unsigned long long* src = p.mbox + mbox_slot(img, half ^ 1, epoch & 1);
        unsigned v[8];
        bool need[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int gi = lane + 64 * k;                       // granule = (chunk, pixel, octet, dword)
          need[k] = (gi >> 7) < chunks && ((gi >> 3) & 15) < p.W;
        }
        for (unsigned spins = 0;; ++spins) {
          bool all = true;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned long long x = __hip_atomic_load(src + lane + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[k] = (unsigned)x;
            all = all && (!need[k] || (unsigned)(x >> 32) == epoch);
          }
          if (__all(all)) break;
          if (spins >= p.spin_limit) { mb_failed = 1; s_ctl[1] = 1u; break; }   // wave-uniform (spins is): never hang the GPU
          __builtin_amdgcn_s_sleep(2);
        }
        const int iy = recv_row + 1;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int gi = lane + 64 * k, ck = gi >> 7, pxl = (gi >> 3) & 15, oh = (gi >> 2) & 1, dw = gi & 3;
          if (need[k]) *reinterpret_cast<unsigned*>(lds_h + ck * CHUNK_B + (iy * HALO + pxl + 1) * PIX_B + 16 * (oh ^ (iy & 1)) + 4 * dw) = v[k];
        }
      }
    }
  };
#ifdef SF_EXP_GRU_STAGGER   // timing experiment: maps start up to one step apart, so that the chip does not see every workgroup's traffic at the same moment
  if constexpr (SPLIT) {
    for (int i = 0; i < (img % 16) * SF_EXP_GRU_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
  }
#endif
  __syncthreads();  // the zero fill is complete
  if (p.h0) { write_state_tile(1u); receive_row(1u); }

  // per-lane LDS offsets of the fragment reads (as in conv3x3_bf16.hip)
  const int rowpar = (r >> 4) & 1;
  const int a_lane = ((wrow + (r >> 4)) * HALO + (r & 15)) * PIX_B;
  const int a_half_even = 16 * (kh ^ rowpar), a_half_odd = 16 * (kh ^ rowpar ^ 1);
  const int b_lane = nbk * W_B + r * PIX_B + 16 * (kh ^ ((r >> 3) & 1));

  if (p.T > 0 && chunks > 0) issue_weights(0, 0);

  // bias of the candidate's h-part (z / r biases ride on the x-part): loaded ONCE - inside the time loop the DMA statements' memory clobbers keep
  // the compiler from hoisting it, and every step's epilogue then began with an exposed L2 round trip
  f32x4 b2[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) b2[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nbk * NBG + 64 + 8 * g + 4 * kh) : f32x4{0.f, 0.f, 0.f, 0.f};

  using GXV = typename std::conditional<GXBF, bf16x4, f32x4>::type;
  // SPLIT with bf16 gates: a step's outputs (state + saved gates) are STORED ONE STEP LATE, in the middle of the next step's last K chunk.  Stored right
  // after the epilogue they were the youngest vector-memory operations at the next step's first `s_waitcnt vmcnt(0)` (needed for that chunk's weights),
  // and the whole workgroup sat out their round trip to memory every step.  The state is still in `hst` until the next epilogue; the gates wait as 32
  // packed dwords (`pend`).  The outputs of the last step leave after the time loop.
  constexpr int NPEND = SPLIT ? 4 : 1;
  u32x4_t pend[NPEND][2];
  const bool defer = SPLIT && p.gates && p.gates_bf;
  // the counted-wait path of the time loop (see chunk4): hidp = 64, bf16 x-part, bf16 gates, outputs addressable through 32-bit buffer offsets
  const long long out_px = (long long)p.T * p.n * img_px;
  constexpr bool fast4 = F4;  // launcher: hidp == 64, bf16 gates present, out_px * stride * element size < 2^31 for both outputs
  const bool sender = has_partner && half != p.mute_half && (wrow == send_row || wrow + 1 == send_row);  // this wave issues the 8 hand-off stores
  const __amdgpu_buffer_rsrc_t rs_hs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hs, 0, (int)(fast4 ? out_px * p.hs_s * 4 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)p.gates, 0, (int)(fast4 ? out_px * p.gates_s * 2 : 0), 0x00020000);
  // Row-pair exchange (SPLIT): a lane pair (r, kh = 0 / 1) holds 2 x 16 bytes of its pixel per register pair, so a store instruction wrote 32-byte
  // runs - and the L2 takes one request per run: the 590 k partial-line requests per step WERE the store cost (tools/ablate_gru.sh).  After one
  // v_permlane16_swap per dword the four 16-lane rows of a register hold four consecutive 16-byte pieces of ONE pixel (rows 0 / 1 of the fragment
  // alternate between the two result registers): 64-byte runs, half the requests.  Lane (row rho = lane >> 4, i = lane & 15) then addresses pixel
  // (wrow + j, i) for result j and piece 2 * (rho & 1) + (rho >> 1) of the run.
  const int rho = lane >> 4, piece = 2 * (rho & 1) + (rho >> 1);
  bool okj[2]; long long pixj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    okj[j] = (wrow + j) < p.H && (lane & 15) < p.W;
    pixj[j] = (long long)img * img_px + (okj[j] ? (wrow + j) * p.W + (lane & 15) : 0);
  }
  auto rows16 = [&](const u32x4_t& a, const u32x4_t& b, u32x4_t& r0, u32x4_t& r1) {  // (A, B) of a lane pair -> the two per-row registers
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const auto sw = __builtin_amdgcn_permlane16_swap(a[d], b[d], false, false);
      r0[d] = sw[0]; r1[d] = sw[1];
    }
  };
  auto store_pending = [&](int ts) {
#ifdef SF_EXP_GRU_NOSTORE
#pragma unroll
    for (int q = 0; q < NPEND; ++q) { asm volatile("" ::"v"(pend[q][0]), "v"(pend[q][1])); }
    return;
#endif
    if constexpr (SPLIT) {
      const long long step = (long long)ts * p.n * img_px;
#pragma unroll
      for (int gp2 = 0; gp2 < 4; gp2 += 2) {  // state quads (g, g + 1): 16 channels = 64 bytes per pixel
        u32x4_t r0, r1;
        rows16(__builtin_bit_cast(u32x4_t, hst[0][gp2]), __builtin_bit_cast(u32x4_t, hst[0][gp2 + 1]), r0, r1);
        const int ch = nbk * 32 + 8 * gp2 + 4 * piece;
        if (okj[0] && ch < p.hidp) *reinterpret_cast<u32x4_t*>(p.hs + (step + pixj[0]) * p.hs_s + ch) = r0;
        if (okj[1] && ch < p.hidp) *reinterpret_cast<u32x4_t*>(p.hs + (step + pixj[1]) * p.hs_s + ch) = r1;
      }
      const int chg = nbk * 32 + 8 * piece;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (okj[j] && chg < p.hidp)
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<__bf16*>(p.gates) + (step + pixj[j]) * p.gates_s + q * p.hidp + chg) = pend[q][j];
    }
  };
  u32x4_t gxr[3][2];  // SPLIT: the x-part as loaded (F4: of the NEXT step from the last chunk on)
#ifdef SF_EXP_GRU_CLK
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, stamp = wall_clock64();
#define SF_GRU_STAMP(i) { const unsigned long long now_ = wall_clock64(); ph[i] += now_ - stamp; stamp = now_; }
#else
#define SF_GRU_STAMP(i) {}
#endif
  for (int t = 0; t < p.T; ++t) {
    f32x16 acc[MFW][3];
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][g][i] = 0.f;

    // this step's x-part: bf16-stored, it is requested now and consumed by the epilogue a whole K loop later (48 registers);
    // fp32-stored, it would take 96 registers over the K loop and is fetched by the epilogue instead
    GXV gxv[MFW][3][4];
    const long long pix_t = ((long long)t * p.n + img) * img_px;
    auto load_gx = [&](int mf) {
#ifdef SF_EXP_GRU_NOGX
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int c = 0; c < 4; ++c) gxv[mf][q][g][c] = 0;
      return;
#endif
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
    if constexpr (GXBF && !SPLIT) {
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf) load_gx(mf);
    }
    // SPLIT, bf16 x-part: the mirror image of the stores - six 16-byte loads per lane in 64-byte runs per pixel (rows 0 / 1 of the fragment per
    // register), un-swapped by v_permlane16_swap into the octets of the lane pair's own pixel and by v_permlane32_swap into this lane's channel
    // quads (twelve 8-byte loads in 16-byte runs before: twice the instructions, four times the requests)
    auto load_gx_rows = [&]() {
#ifdef SF_EXP_GRU_NOGX
#pragma unroll
      for (int q = 0; q < 3; ++q) { gxr[q][0] = u32x4_t{0, 0, 0, 0}; gxr[q][1] = u32x4_t{0, 0, 0, 0}; }
      return;
#endif
      const int chg = nbk * 32 + 8 * piece < p.hidp ? nbk * 32 + 8 * piece : 0;  // clamped: loads are unconditional
      const long long step = (long long)t * p.n * img_px;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          gxr[q][j] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const __bf16*>(p.gx) + (step + pixj[j]) * p.gx_s + q * p.hidp + chg);
    };
    auto unpack_gx = [&]() {
      if constexpr (GXBF && SPLIT) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          u32x4_t oc[2];  // octets of this lane's pixel: channels 8 * kh .. and 16 + 8 * kh ..
          rows16(gxr[q][0], gxr[q][1], oc[0], oc[1]);
#pragma unroll
          for (int o = 0; o < 2; ++o) {  // octet (low quad, high quad) of the lane pair -> quads 2o and 2o + 1 of this lane
            const auto s0 = __builtin_amdgcn_permlane32_swap(oc[o][0], oc[o][2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(oc[o][1], oc[o][3], false, false);
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            gxv[0][q][2 * o] = __builtin_bit_cast(bf16x4, u32x2_t{s0[0], s1[0]});
            gxv[0][q][2 * o + 1] = __builtin_bit_cast(bf16x4, u32x2_t{s0[1], s1[1]});
          }
        }
      }
    };
    // SPLIT: the partner's boundary row of the state this step reads (epoch t + 1: the state after step t - 1)
    if (t > 0) receive_row((unsigned)t + 1u);
    SF_GRU_STAMP(0)   // top of the step: accumulator clear + the partner's row

    // one K chunk; LAST = the step's last chunk, peeled out of the loop in the SPLIT kernel (straight-line code: see below)
    auto chunk = [&](int ci, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      const int it = t * chunks + ci, cur = it & 1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this chunk's weights have landed (and everything older)
      __syncthreads();                                   // ... everybody's; the previous step's state tile writes are visible
      if (ci == 0) SF_GRU_STAMP(1)                       // wait + barrier of the first chunk
      else if (LAST) SF_GRU_STAMP(2)                     // chunks 0 .. n-2 and the last chunk's wait
      if constexpr (SPLIT && LAST) {
        // this step's x-part was requested a chunk ago and the wait above covered it: consumed HERE (the compiler's own wait for these loads lands
        // where the counter is already zero), or its conservative wait at the epilogue's first use would also wait for the stores that follow
        if constexpr (GXBF) unpack_gx();
        // last step's outputs: the next vmcnt(0) is a whole chunk of MFMAs, the epilogue and the hand-off away
        if (defer && t > 0) store_pending(t - 1);
      }
      // the two waves of a SIMD (w, w + 4) issue the next chunk's DMA at different taps: one's MFMAs cover the other's issue stall
      const bool stage_late = NBLK == 2 && wave >= 4;
      const bool more = it + 1 < p.T * chunks;
      const int nci = ci + 1 < chunks ? ci + 1 : 0;
      if (more && !stage_late) issue_weights(nci, cur ^ 1);
      if constexpr (GXBF && SPLIT && !LAST) {
        // x-part of this step: requested behind the last-but-one wait, complete at the last one (under a chunk of MFMAs).  Requested at the top of
        // the step it was the youngest operation at the first chunk's wait: an exposed round trip to HBM per step.
        if (ci == chunks - 2) load_gx_rows();
      }
      const char* inb = lds_h + ci * CHUNK_B + a_lane;
      const char* wb = lds + cur * STAGE_B + b_lane;
      auto load_tap = [&](int tap, bf16x8 (&a)[MFW], bf16x8 (&b)[3]) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int mf = 0; mf < MFW; ++mf)
          a[mf] = *reinterpret_cast<const bf16x8*>(inb + ((2 * mf + ky) * HALO + kx) * PIX_B + ((ky & 1) ? a_half_odd : a_half_even));
#pragma unroll
        for (int g = 0; g < 3; ++g) b[g] = *reinterpret_cast<const bf16x8*>(wb + (tap * NBG + g * 32) * PIX_B);
      };
      bf16x8 fa[2][MFW], fb[2][3];
      load_tap(0, fa[0], fb[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
#pragma unroll
        for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
          for (int g = 0; g < 3; ++g) {  // transposed product: D[channel][pixel] - a lane owns one pixel and channel quads
#ifdef SF_EXP_GRU_NOMFMA
            if (tap > 0) { asm volatile("" ::"v"(fb[tap & 1][g]), "v"(fa[tap & 1][mf])); continue; }
#endif
            acc[mf][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tap & 1][g], fa[tap & 1][mf], acc[mf][g], 0, 0, 0);
          }
        if (tap == 3 && more && stage_late) issue_weights(nci, cur ^ 1);
      }
    };
    // ---- hidp = 64 (four K chunks) with bf16 x-part and bf16 gates: MetNet's configuration.  NOTHING here waits for a store or for the x-part:
    // the 12 output stores of the previous step and the 6 x-part loads of this one are spread over the K loop (a store per tap at taps 5..7 of every
    // chunk, a load pair at tap 4 of chunks 0..2), each behind the chunk's weight DMA, and every chunk wait is COUNTED: it leaves exactly the
    // operations issued after the awaited weights in flight (loads and stores retire from vmcnt in order: tools/ubench/vmcnt_order.hip).  Issued in
    // one burst, the 96 KB of stores per workgroup - 18 MB from the 192 workgroups at the same moment - held the issuing waves for 3.8 us per step
    // (the HBM write rate), wherever in the step the burst sat (tools/ablate_gru.sh, CLK variants).  All stores are buffer stores that are ALWAYS
    // issued (invalid lanes / the first step: an offset outside the descriptor), so the counts are static.
    auto load_gx4 = [&](int ts, int lo = 0, int hi = 6) {  // loads lo .. hi-1 of the six (gate q = i >> 1, fragment row j = i & 1)
      const int chg = nbk * 32 + 8 * piece;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (2 * q + j < lo || 2 * q + j >= hi) continue;
#ifdef SF_EXP_GRU_NOGX   // timing only: the same instructions on one hot line per workgroup (no HBM traffic)
          gxr[q][j] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const __bf16*>(p.gx) + (long long)img * img_px * p.gx_s + (lane & 7) * 8);
#else
          gxr[q][j] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const __bf16*>(p.gx) + ((long long)ts * p.n * img_px + pixj[j]) * p.gx_s + q * 64 + chg);
#endif
        }
    };
    if constexpr (F4) {
      if (t == 0) load_gx4(0);
    }
    auto chunk4 = [&](auto ci_tag) {
      if constexpr (F4) {
        constexpr int CI = decltype(ci_tag)::value;
        const int it = t * 4 + CI, cur = it & 1;
        if constexpr (CI == 0) {  // younger than this chunk's weights: the last chunk's 6 loads + 3 stores and the epilogue's hand-off (8 granule stores)
          if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (sender) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // the previous chunk's stores (chunk 1: the x-part loads are older - a K loop old)
        __syncthreads();
        if (CI == 0) SF_GRU_STAMP(1)
        else if (CI == 3) SF_GRU_STAMP(2)
        if constexpr (CI == 3) unpack_gx();
        const bool stage_late = wave >= 4;
        const bool more = it + 1 < p.T * 4;
        constexpr int nci = (CI + 1) & 3;
        if (more && !stage_late) issue_weights(nci, cur ^ 1);
        const char* inb = lds_h + CI * CHUNK_B + a_lane;
        const char* wb = lds + cur * STAGE_B + b_lane;
        auto load_tap = [&](int tap, bf16x8& a, bf16x8 (&b)[3]) {
          const int ky = tap / 3, kx = tap % 3;
          a = *reinterpret_cast<const bf16x8*>(inb + (ky * HALO + kx) * PIX_B + ((ky & 1) ? a_half_odd : a_half_even));
#pragma unroll
          for (int g = 0; g < 3; ++g) b[g] = *reinterpret_cast<const bf16x8*>(wb + (tap * NBG + g * 32) * PIX_B);
        };
        bf16x8 fa[2], fb[2][3];
        load_tap(0, fa[0], fb[0]);
        u32x4_t hr0, hr1;  // state quads after the row exchange (chunk 0: quads 0 / 1, and 2 / 3 for its third store; chunk 1: quads 2 / 3)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
#pragma unroll
          for (int g = 0; g < 3; ++g) {
#ifdef SF_EXP_GRU_NOMFMA
            if (tap > 0) { asm volatile("" ::"v"(fb[tap & 1][g]), "v"(fa[tap & 1])); continue; }
#endif
            acc[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tap & 1][g], fa[tap & 1], acc[0][g], 0, 0, 0);
          }
          if (tap == 3 && more && stage_late) issue_weights(nci, cur ^ 1);

          if (tap >= 5 && tap <= 7) {  // store k of the previous step's twelve
            const int k = 3 * CI + tap - 5;
            const unsigned off_bad = 0x80000000u;  // beyond both descriptors' ranges (checked by the launcher)
            const long long stepm = (long long)(t - 1) * p.n * img_px;
            if (k < 4) {
              const int gp2 = k & 2, j = k & 1;
              if (k == 0 || k == 2 || k == 3) rows16(__builtin_bit_cast(u32x4_t, hst[0][gp2]), __builtin_bit_cast(u32x4_t, hst[0][gp2 + 1]), hr0, hr1);
              const int ch = nbk * 32 + 8 * gp2 + 4 * piece;
#ifdef SF_EXP_GRU_NOSTORE   // timing only: the same instructions, every lane outside the descriptor's range (no traffic)
              const unsigned off = off_bad | (unsigned)ch;
#else
              const unsigned off = t > 0 && okj[j] ? (unsigned)(((stepm + pixj[j]) * p.hs_s + ch) * 4) : off_bad;
#endif
              __builtin_amdgcn_raw_buffer_store_b128(j ? hr1 : hr0, rs_hs, (int)off, 0, 0);
            } else {
              const int q = (k - 4) >> 1, j = (k - 4) & 1;
#ifdef SF_EXP_GRU_NOSTORE
              const unsigned off = off_bad | (unsigned)q;
#else
              const unsigned off = t > 0 && okj[j] ? (unsigned)(((stepm + pixj[j]) * p.gates_s + q * 64 + nbk * 32 + 8 * piece) * 2) : off_bad;
#endif
              __builtin_amdgcn_raw_buffer_store_b128(pend[q][j], rs_g, (int)off, 0, 0);
            }
          }
        }
      }
    };
    if constexpr (SPLIT) {
      if constexpr (fast4) {
        chunk4(std::integral_constant<int, 0>{}); chunk4(std::integral_constant<int, 1>{});
        chunk4(std::integral_constant<int, 2>{}); chunk4(std::integral_constant<int, 3>{});
      } else {
        if constexpr (GXBF) {
          if (chunks < 2) load_gx_rows();  // a single chunk: no earlier wait to hide behind
        }
        for (int ci = 0; ci + 1 < chunks; ++ci) chunk(ci, std::false_type{});
        chunk(chunks - 1, std::true_type{});
      }
    } else {
      for (int ci = 0; ci < chunks; ++ci) chunk(ci, std::false_type{});
    }
    __syncthreads();  // every wave is done reading the state tile of step t - 1
    SF_GRU_STAMP(3)   // the last chunk (with the deferred stores and the prefetch) + the barrier behind the K loop

    // ---- epilogue: gates, new state (registers), outputs, bf16 image of the new state into the tile ----
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) {
      if constexpr (!GXBF) load_gx(mf);
      const long long pix = pix_t + (ok[mf] ? py[mf] * p.W + px[mf] : 0);
      f32x4 zz[4], rr[4], nn[4], hh2[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gz = __builtin_convertvector(gxv[mf][0][g], f32x4), gr = __builtin_convertvector(gxv[mf][1][g], f32x4),
                    gn = __builtin_convertvector(gxv[mf][2][g], f32x4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#ifdef SF_EXP_GRU_NOTRANS
          const float z = 0.5f * (acc[mf][0][4 * g + c] + gz[c]);
          const float rg = 0.5f * (acc[mf][1][4 * g + c] + gr[c]);
          const float h2 = acc[mf][2][4 * g + c] + b2[g][c];
          const float cand = 0.25f * sf_gru_cand_arg(gn[c], rg, h2);
#else
          const float z = sf_sigmoid(acc[mf][0][4 * g + c] + gz[c]);
          const float rg = sf_sigmoid(acc[mf][1][4 * g + c] + gr[c]);
          const float h2 = acc[mf][2][4 * g + c] + b2[g][c];
          const float cand = sf_tanh(sf_gru_cand_arg(gn[c], rg, h2));
#endif
          zz[g][c] = z; rr[g][c] = rg; nn[g][c] = cand; hh2[g][c] = h2;
          hst[mf][g][c] = sf_gru_blend(z, cand, hst[mf][g][c]);
        }
        if constexpr (F4) {
          // the NEXT step's x-part (this step's was unpacked at the top of the last chunk: gxr is free), its six loads spread through the gate
          // arithmetic - the CU's memory pipe is idle here, and by the time the hand-off stores are issued below they have been taken; the last step
          // reloads its own (always issued: the counts are static)
          const int ts = t + 1 < p.T ? t + 1 : t;
          if (g == 0) load_gx4(ts, 0, 2);
          else if (g == 1) load_gx4(ts, 2, 3);
          else if (g == 2) load_gx4(ts, 3, 5);
          else load_gx4(ts, 5, 6);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (SPLIT) {  // a hand-off of this workgroup timed out (flag set before one of the chunk barriers above): NaN from here on
        if (s_ctl[1]) {
#pragma unroll
          for (int g = 0; g < 4; ++g) hst[mf][g] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        }
      }
      // SPLIT (one M fragment per wave): the new state's bf16 image - and with it the boundary row for the partner workgroup - leaves
      // BEFORE this step's outputs are stored: the hand-off's latency then runs under the stores
      if constexpr (SPLIT) write_state_tile((unsigned)t + 2u);
      if constexpr (SPLIT) {
        if (defer) {  // pack the gates; they and the state are stored during the next step (or after the loop)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4* v = q == 0 ? zz : q == 1 ? rr : q == 2 ? nn : hh2;
            u32x4_t oc[2];
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
              const unsigned ax = pk(v[g][0], v[g][1]), ay = pk(v[g][2], v[g][3]);
              const unsigned bx = pk(v[g + 1][0], v[g + 1][1]), by = pk(v[g + 1][2], v[g + 1][3]);
              const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
              const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
              oc[g >> 1] = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            }
            rows16(oc[0], oc[1], pend[q][0], pend[q][1]);
          }
          continue;
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
    if constexpr (!SPLIT) write_state_tile((unsigned)t + 2u);
    SF_GRU_STAMP(4)   // epilogue
  }
#ifdef SF_EXP_GRU_CLK
  if (SPLIT && lane == 0 && (ticket == 100 || ticket == 101) && (wave == 0 || wave == 5))
    printf("ticket %d wave %d: per step (ns) top+poll %.0f | chunk-0 wait %.0f | chunks 0..n-2 %.0f | last chunk %.0f | epilogue %.0f | sum %.0f\n", ticket, wave,
           10.0 * ph[0] / p.T, 10.0 * ph[1] / p.T, 10.0 * ph[2] / p.T, 10.0 * ph[3] / p.T, 10.0 * ph[4] / p.T, 10.0 * (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / p.T);
#endif
  if constexpr (SPLIT) {
    if (defer && p.T > 0) store_pending(p.T - 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (SPLIT) {
    if (mb_failed && lane == 0) atomicOr(reinterpret_cast<unsigned*>(p.mbox), 1u);  // the sticky error word
  }
}


// =============================================================================================
// Persistent BACKWARD of the same time loop (sf_convgru_seq_bwd): for t = T-1 .. 0
//   dh_t   = [g_seq_t] + [g_last at t = T-1] + dd_{t+1} + conv3x3^T(dgh_{t+1}, Wh)
//   (az, ar, an, d2, dd) = gate backward(dh_t; saved z, r, n, h2 of step t; h_{t-1})          (sf_gru_bwd)
//   dgx_t = [az | ar | an],  dgh_t = [az | ar | d2]                                          -> HBM (bf16), read by the batched convolutions
// One workgroup owns one image.  The carried gradient (dd and the convolution result) never leaves registers: the lane that
// receives an element of conv^T(dgh_{t+1}) from the MFMA is the lane that runs the gate backward of that element.  dgh_{t+1} as the
// convolution's operand lives in LDS as bf16 [K chunk][pixel][16 channels] (no halo: taps that fall outside the image read a
// zero pixel kept behind every chunk); the transposed-flipped recurrent weights (sf_conv3x3_pack_weights, transpose 1, N = hidp,
// K = 3 * hidp) stream from L2 through a two-stage ring by LDS-DMA.  The saved gates and h_{t-1} of the NEXT step to process are
// requested before the K loop and arrive under it.
// 8 waves: wave = (4-row band, M fragment): one 32-pixel fragment x all hidp channels per wave; K order (chunks ascending, taps
// ascending) and products are those of the per-step convolution, the gate arithmetic is sf_gru_bwd: dgx / dgh are bit-identical
// to the per-step kernels (sf_convgru_bwd_gates + sf_conv3x3_fwd), tests/test_convgru_seq_gpu.py.
struct GruSeqBwdParams {
  const float* g_seq; int gs_s;        // [T][n][H][W][hidp] fp32 or null
  const float* g_last; int gl_s;       // [n][H][W][hidp] fp32 or null
  const __bf16* gates; int gates_s;    // saved z | r | n | h2, [T][n][H][W][4*hidp] bf16
  const float* hs; int hs_s;           // states [T][n][H][W][hidp] fp32 (h_{t-1} = hs[t-1], zeros at t = 0)
  __bf16* dgx; int dgx_s;              // [T][n][H][W][3*hidp] bf16
  __bf16* dgh; int dgh_s;
  const void* wp;                      // packed transposed weights [chunks = 3*hidp/16][9][hidp][16] bf16
  int T, n, H, W, hidp;
  unsigned long long* mbox;            // SPLIT kernel: boundary-row mailbox, zeroed before the launch
  unsigned spin_limit; int mute_half;  // as in GruSeqParams
};
constexpr int MBB_ROW = 12 * 16 * 2 * 4;  // granules of one boundary row of dgh (3 * 64 channels = 12 K chunks)
__host__ __device__ constexpr long long mbox_bwd_slot(long long img, int half, int parity) { return MB_HDR + ((img * 2 + half) * 2 + parity) * MBB_ROW; }

constexpr int BP_CHUNK_B = (256 + 1) * PIX_B;  // 256 pixels + one zero pixel per K chunk

// NFR: 32-channel fragments of the hidden state (hidp = 32 * NFR); MFW: M fragments (32 pixels) per wave - 8 / MFW waves.
// MFW = 2 (4 waves, one per SIMD, 512 registers each): a tap costs 2 + NFR fragment reads for 2 * NFR MFMAs instead of 1 + NFR for
// NFR; the element-to-lane map stays that of the MFMA result.  The launcher uses MFW = 1 (see there).
// SPLIT (NFR = 2, MFW = 1): two workgroups per image, 8 rows each; wave = (4-row band, M fragment of the band, channel fragment): a
// wave owns ONE 32-pixel fragment x ONE 32-channel fragment (NFW = 1), still 8 waves.  After the gate backward of a step the bf16
// dgh values of the workgroup's boundary row go to the partner through the mailbox (12 K chunks x 16 pixels x 2 octets x 4 dwords
// = 1536 granules) and the partner's row lands in the tile before the K loop; everything else is unchanged (bit-identical results).
// CW (SPLIT): counted waits - the step's 12 output stores and the next step's 24 loads are spread over the K loop (four per chunk, each behind the
// chunk's weight DMA), no chunk wait covers them (see the forward kernel's chunk4 for the measurements behind this).
template <int NFR, int MFW, bool SPLIT = false, bool CW = false>
__global__ __launch_bounds__(512 / MFW, 2 / MFW) void convgru_seq_bwd_kernel(const GruSeqBwdParams p) {
  static_assert(!SPLIT || (NFR == 2 && MFW == 1), "the split kernel is the hidp = 64, one-fragment-per-wave layout");
  static_assert(!CW || SPLIT, "counted waits: the split kernel");
  constexpr int NFW = SPLIT ? 1 : NFR;  // channel fragments per wave
  constexpr int HID = 32 * NFR, CHUNKS = 3 * HID / 16;
  constexpr int WB = 9 * HID * PIX_B;          // weights of one chunk
  constexpr int PIECES = WB / 1024;            // 9 * NFR
  constexpr int WAVES = 8 / MFW, THREADS = 64 * WAVES;
  constexpr int RING = 3;                      // weight stages: a chunk's DMA is issued two chunks ahead of its use
  __shared__ __attribute__((aligned(1024))) char lds[RING * WB + CHUNKS * BP_CHUNK_B];
  char* lds_t = lds + RING * WB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, kh = lane >> 5;
  __shared__ unsigned s_ctl[2];  // SPLIT: [0] this workgroup's ticket, [1] "a hand-off timed out" (see the forward kernel)
  if constexpr (SPLIT) {
    if (tid == 0) { s_ctl[0] = atomicAdd(reinterpret_cast<unsigned*>(p.mbox + 1), 1u); s_ctl[1] = 0u; }
    __syncthreads();
  }
  const int ticket = SPLIT ? __builtin_amdgcn_readfirstlane((int)s_ctl[0]) : (int)blockIdx.x;
  const int img = SPLIT ? ticket >> 1 : ticket;
  const int half = SPLIT ? ticket & 1 : 0;
  // first row of this wave's M fragment(s): band wl = wave & 3, fragments mf0 .. of it (SPLIT: band (wave & 3) >> 1 of this half, fragment wave & 1)
  const int wrow = SPLIT ? 8 * half + 4 * ((wave & 3) >> 1) + 2 * (wave & 1) : 4 * (wave & 3) + 2 * (MFW == 2 ? 0 : wave >> 2);
  const int nf0 = SPLIT ? wave >> 2 : 0;  // first channel fragment of this wave
  const int send_row = half == 0 ? 7 : 8, recv_row = half == 0 ? 8 : 7;
  const bool has_partner = SPLIT && p.H > 8;
  unsigned mb_failed = 0;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  for (int i = tid; i < CHUNKS * BP_CHUNK_B / 16; i += THREADS) *reinterpret_cast<f32x4*>(lds_t + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, CHUNKS * WB, 0x00020000);
  auto issue_weights = [&](int ci, int buf) {
    for (int i = wave; i < PIECES; i += WAVES) bufdma16(lane * 16, rs_w, (unsigned)(ci * WB + i * 1024), lds0 + (unsigned)(buf * WB + i * 1024));
  };

  // this lane's pixels (one per M fragment) and its channel quads nf * 32 + 8g + 4kh (the transposed product's accumulator layout)
  const long long img_px = (long long)p.H * p.W;
  const long long step_px = (long long)p.n * img_px;
  const int cq = 4 * kh;
  int py[MFW], px[MFW];
  bool ok[MFW];
  long long pix_i[MFW];  // + t * n * img_px; clamped: loads are unconditional
  int a_off[MFW][9];     // A-operand read offsets per tap (within a chunk): the source pixel of tap (ky, kx), or the chunk's zero pixel
#pragma unroll
  for (int m = 0; m < MFW; ++m) {
    py[m] = wrow + 2 * m + (r >> 4); px[m] = r & 15;
    ok[m] = py[m] < p.H && px[m] < p.W;
    pix_i[m] = (long long)img * img_px + (ok[m] ? py[m] * p.W + px[m] : 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int sy = py[m] + tap / 3 - 1, sx = px[m] + tap % 3 - 1;
      const bool in = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
      a_off[m][tap] = in ? (sy * 16 + sx) * PIX_B + 16 * (kh ^ (sy & 1)) : 256 * PIX_B + 16 * kh;
    }
  }
  const int b_lane = r * PIX_B + 16 * (kh ^ ((r >> 3) & 1));

  __syncthreads();  // zero fill complete
  const int total_chunks = (p.T - 1) * CHUNKS;  // K chunks of the whole sequence (T - 1 convolutions)
  if (total_chunks > 0) issue_weights(0, 0);
  if (total_chunks > 1) issue_weights(1 % CHUNKS, 1);
  const bool many = wave < PIECES - WAVES * (PIECES / WAVES);  // this wave issues one more DMA piece per chunk than the others (uniform)

  // gradient wrt h_t carried into the gate backward (this lane's 16 * NFR elements per M fragment)
  f32x4 dh[MFW][NFW][4];
#pragma unroll
  for (int m = 0; m < MFW; ++m) {
    const long long pt = (long long)(p.T - 1) * step_px + pix_i[m];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = (nf0 + nf) * 32 + 8 * g + cq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p.g_seq) v = *reinterpret_cast<const f32x4*>(p.g_seq + pt * p.gs_s + ch);
        if (p.g_last) v += *reinterpret_cast<const f32x4*>(p.g_last + pix_i[m] * p.gl_s + ch);
        dh[m][nf][g] = v;
      }
  }
  // saved gates and previous state of the step about to be processed
  bf16x4 gv[MFW][4][NFW][4];
  f32x4 hp[MFW][NFW][4];
  auto request = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MFW; ++m) {
      const long long pt = (long long)t * step_px + pix_i[m];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            gv[m][q][nf][g] = *reinterpret_cast<const bf16x4*>(p.gates + pt * p.gates_s + q * p.hidp + (nf0 + nf) * 32 + 8 * g + cq);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          hp[m][nf][g] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (t > 0) hp[m][nf][g] = *reinterpret_cast<const f32x4*>(p.hs + (pt - step_px) * p.hs_s + (nf0 + nf) * 32 + 8 * g + cq);
        }
    }
  };
  if constexpr (!CW) request(p.T - 1);
  // ---- CW: state of the spread stores / loads ----
  // rows of the fragment per register after v_permlane16_swap (64-byte runs per pixel, see the forward kernel): lane (rho = lane >> 4, i = lane & 15)
  // addresses pixel (wrow + j, i) for register j and 16-byte piece 2 * (rho & 1) + (rho >> 1) of the wave's 32 channels
  const int rho = lane >> 4, piece = 2 * (rho & 1) + (rho >> 1);
  bool okj[2]; long long pixj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    okj[j] = (wrow + j) < p.H && (lane & 15) < p.W;
    pixj[j] = (long long)img * img_px + (okj[j] ? (wrow + j) * p.W + (lane & 15) : 0);
  }
  u32x4_t pendb[4][2];   // az, ar, an, d2 of the step just processed, per fragment row
  f32x4 gsv[4];          // g_seq of the step about to be processed (CW: loaded inside the K loop)
  const long long out_px = (long long)p.T * step_px;
  const __amdgpu_buffer_rsrc_t rs_dgx = __builtin_amdgcn_make_buffer_rsrc((void*)p.dgx, 0, (int)(CW ? out_px * p.dgx_s * 2 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dgh = __builtin_amdgcn_make_buffer_rsrc((void*)p.dgh, 0, (int)(CW ? out_px * p.dgh_s * 2 : 0), 0x00020000);
  const bool sender = has_partner && half != p.mute_half && (wrow == send_row || wrow + 1 == send_row);  // this wave issues the 24 hand-off stores
  const __amdgpu_buffer_rsrc_t rs_gates = __builtin_amdgcn_make_buffer_rsrc((void*)p.gates, 0, (int)(CW ? out_px * p.gates_s * 2 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.hs, 0, (int)(CW ? out_px * p.hs_s * 4 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_gs = __builtin_amdgcn_make_buffer_rsrc((void*)p.g_seq, 0, (int)(CW && p.g_seq ? out_px * p.gs_s * 4 : 0), 0x00020000);
  unsigned pxl32 = (unsigned)pix_i[0];   // this lane's pixel (within the step), laundered per chunk in the K loop
  unsigned pxj32[2] = {(unsigned)pixj[0], (unsigned)pixj[1]};
  // store k (0..11) of step ts: dgx = [az | ar | an], dgh = [az | ar | d2]
  auto store_k = [&](int k, int ts) {
    const int qs = k < 8 ? k >> 2 : (k < 10 ? 2 : 3), j = k & 1;
    const bool to_dgh = k < 8 ? ((k >> 1) & 1) : k >= 10;
    const int qd = qs == 3 ? 2 : qs;
    const int chg = nf0 * 32 + 8 * piece;
    const int stride = to_dgh ? p.dgh_s : p.dgx_s;
    const unsigned bad = 0x80000000u;
#ifdef SF_EXP_GRU_NOSTORE
    const unsigned off = bad | (unsigned)k;
    (void)chg; (void)qd;
#else
    const unsigned off = okj[j] ? (unsigned)((pxj32[j] * stride + qd * HID + chg) * 2) : bad;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(pendb[qs][j], to_dgh ? rs_dgh : rs_dgx, (int)off, (int)(unsigned)((long long)ts * step_px * stride * 2), 0);
  };
  // load k (0..15) for step ts (clamped at 0; always issued): 0..7 the saved gates as loaded - gate k >> 1, fragment row k & 1, 16 bytes per lane in
  // 64-byte runs per pixel (the mirror image of the stores; unpacked by the gate backward) -, 8..11 h_{ts-1}, 12..15 g_seq_ts
  u32x4_t gvr[4][2];
  auto load_k = [&](int k, int ts) {
    const int tc = ts > 0 ? ts : 0;
    const int g = k & 3;
#ifdef SF_EXP_GRU_NOGX   // timing only: one hot line per workgroup
    const long long hot = (long long)img * img_px;
    if (k < 8) gvr[k >> 1][k & 1] = *reinterpret_cast<const u32x4_t*>(p.gates + hot * p.gates_s + (lane & 7) * 8);
    else if (k < 12) hp[0][0][g] = *reinterpret_cast<const f32x4*>(p.hs + hot * p.hs_s + (lane & 15) * 4);
    else gsv[g] = *reinterpret_cast<const f32x4*>(p.hs + hot * p.hs_s + (lane & 15) * 4);
    (void)tc;
#else
    // descriptor + per-lane 32-bit offset rebuilt from the (per-chunk laundered) pixel index + scalar step offset: no per-lane addresses kept across the loop
    const unsigned so_g = (unsigned)(tc * step_px * p.gates_s * 2), so_h = (unsigned)((tc > 0 ? tc - 1 : 0) * step_px * p.hs_s * 4);
    const unsigned cl = (unsigned)(nf0 * 32 + cq);
    if (k < 8) gvr[k >> 1][k & 1] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_gates, (int)((pxj32[k & 1] * p.gates_s + (k >> 1) * HID + nf0 * 32 + 8 * piece) * 2), (int)so_g, 0));
    else if (k < 12) hp[0][0][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_hsb, (int)((pxl32 * p.hs_s + cl + 8 * g) * 4), (int)so_h, 0));
    else if (p.g_seq) gsv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_gs, (int)((pxl32 * p.gs_s + cl + 8 * g) * 4), (int)(unsigned)(tc * step_px * p.gs_s * 4), 0));
    else gsv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_hsb, (int)((pxl32 * p.hs_s + cl + 8 * g) * 4), (int)so_h, 0));
#endif
  };
  if constexpr (CW) {
#pragma unroll
    for (int k = 0; k < 12; ++k) load_k(k, p.T - 1);  // the first step's gates and previous state (its g_seq is part of dh already)
  }
#ifdef SF_EXP_GRU_CLK
  unsigned long long bph[6] = {0, 0, 0, 0, 0, 0}, bstamp = wall_clock64();
#define SF_GRUB_STAMP(i) { const unsigned long long now_ = wall_clock64(); bph[i] += now_ - bstamp; bstamp = now_; }
#else
#define SF_GRUB_STAMP(i) {}
#endif

  for (int t = p.T - 1; t >= 0; --t) {
    // ---- gate backward of step t ----
    f32x4 dd[MFW][NFW][4];
    if constexpr (CW) {
      // the same arithmetic, two channel quads (one octet per gate) at a time: at most 40 fp32 results are live before they are packed (all 80 at
      // once pushed the 256-register budget of two waves per SIMD into scratch once the packed octets have to survive into the K loop)
      u32x4_t octs[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // saved gates: rows of the fragment -> octets of this lane pair's pixel -> this lane's channel quads
        u32x4_t oc[2];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const auto sw = __builtin_amdgcn_permlane16_swap(gvr[q][0][d], gvr[q][1][d], false, false);
          oc[0][d] = sw[0]; oc[1][d] = sw[1];
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(oc[o][0], oc[o][2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(oc[o][1], oc[o][3], false, false);
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          gv[0][q][0][2 * o] = __builtin_bit_cast(bf16x4, u32x2_t{s0[0], s1[0]});
          gv[0][q][0][2 * o + 1] = __builtin_bit_cast(bf16x4, u32x2_t{s0[1], s1[1]});
        }
      }
#pragma unroll
      for (int gp = 0; gp < 4; gp += 2) {
        f32x4 o4[4][2];  // [az, ar, an, d2][quad of the pair]
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const int g = gp + gi;
          const f32x4 z = __builtin_convertvector(gv[0][0][0][g], f32x4), rr = __builtin_convertvector(gv[0][1][0][g], f32x4),
                      nn = __builtin_convertvector(gv[0][2][0][g], f32x4), h2 = __builtin_convertvector(gv[0][3][0][g], f32x4);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const sfGruBwd o = sf_gru_bwd(dh[0][0][g][c], z[c], rr[c], nn[c], h2[c], t == 0 ? 0.f : hp[0][0][g][c]);
            o4[0][gi][c] = o.az; o4[1][gi][c] = o.ar; o4[2][gi][c] = o.an; o4[3][gi][c] = o.d2; dd[0][0][g][c] = o.dd;
          }
        }
        const int ch = nf0 * 32 + 8 * (gp + kh);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned ax = pk(o4[q][0][0], o4[q][0][1]), ay = pk(o4[q][0][2], o4[q][0][3]);
          const unsigned bx = pk(o4[q][1][0], o4[q][1][1]), by = pk(o4[q][1][2], o4[q][1][3]);
          const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          const u32x4_t oct = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
          octs[q][gp >> 1] = oct;
          if (ok[0] && q != 2 && t > 0) {  // dgh = [az | ar | d2]: the next convolution's operand, and the partner's rows next to this one
            const int k = (q == 3 ? 2 : q) * HID + ch;
            *reinterpret_cast<u32x4_t*>(lds_t + (k >> 4) * BP_CHUNK_B + (py[0] * 16 + px[0]) * PIX_B + 16 * (((ch >> 3) & 1) ^ (py[0] & 1))) = oct;
            if (has_partner && py[0] == send_row && half != p.mute_half) {
              const unsigned epoch = (unsigned)(p.T - t);
              unsigned long long* g8 = p.mbox + mbox_bwd_slot(img, half, epoch & 1) + (k >> 4) * 128 + px[0] * 8 + ((ch >> 3) & 1) * 4;
              const unsigned long long tag = (unsigned long long)epoch << 32;
              __hip_atomic_store(g8 + 0, tag | oct[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(g8 + 1, tag | oct[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(g8 + 2, tag | oct[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(g8 + 3, tag | oct[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)  // the two octets of the lane pair -> the two fragment rows (64-byte runs)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const auto sw = __builtin_amdgcn_permlane16_swap(octs[q][0][d], octs[q][1][d], false, false);
          pendb[q][0][d] = sw[0]; pendb[q][1][d] = sw[1];
        }
      // the 12 stores leave NOW: they drain while the hand-off and the first chunk's wait keep the memory pipe idle anyway (the K loop's share of the
      // CU's ~10 B/clk is taken by the 24 loads of the next step); the counted waits of chunks 0 and 1 leave them in flight
      // (the polling wave's stores wait until its poll is over: the poll's own loads could only return behind them; and nobody's stores enter the CU's
      // memory pipe before every wave's hand-off stores have - the partner workgroup is waiting for exactly those)
#ifndef SF_EXP_GRU_NOSENDBAR
      if (has_partner && t > 0) __syncthreads();
#endif
      if (!(has_partner && wave == 0) || t == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) store_k(k, t);
      }
    } else
#pragma unroll
    for (int m = 0; m < MFW; ++m) {
      const long long pt = (long long)t * step_px + pix_i[m];
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) {
        f32x4 az[4], ar[4], an[4], d2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 z = __builtin_convertvector(gv[m][0][nf][g], f32x4), rr = __builtin_convertvector(gv[m][1][nf][g], f32x4),
                      nn = __builtin_convertvector(gv[m][2][nf][g], f32x4), h2 = __builtin_convertvector(gv[m][3][nf][g], f32x4);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const sfGruBwd o = sf_gru_bwd(dh[m][nf][g][c], z[c], rr[c], nn[c], h2[c], hp[m][nf][g][c]);
            az[g][c] = o.az; ar[g][c] = o.ar; an[g][c] = o.an; d2[g][c] = o.d2; dd[m][nf][g][c] = o.dd;
          }
        }
        // octets after the half-wave swap: this lane then holds channels nf*32 + 8*(g + kh) .. +7 for g = 0, 2
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // q: 0 az, 1 ar, 2 an (dgx only), 3 d2 (dgh only)
          const f32x4* v = q == 0 ? az : q == 1 ? ar : q == 2 ? an : d2;
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const unsigned ax = pk(v[g][0], v[g][1]), ay = pk(v[g][2], v[g][3]);
            const unsigned bx = pk(v[g + 1][0], v[g + 1][1]), by = pk(v[g + 1][2], v[g + 1][3]);
            const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            const u32x4_t oct = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            const int ch = (nf0 + nf) * 32 + 8 * (g + kh);
            if (ok[m]) {
              if (q != 3) *reinterpret_cast<u32x4_t*>(p.dgx + pt * p.dgx_s + q * p.hidp + ch) = oct;
              if (q != 2) {
                const int qh = q == 3 ? 2 : q;  // dgh = [az | ar | d2]
                *reinterpret_cast<u32x4_t*>(p.dgh + pt * p.dgh_s + qh * p.hidp + ch) = oct;
                if (t > 0) {  // the next convolution's operand
                  const int k = qh * HID + ch;
                  *reinterpret_cast<u32x4_t*>(lds_t + (k >> 4) * BP_CHUNK_B + (py[m] * 16 + px[m]) * PIX_B + 16 * (((ch >> 3) & 1) ^ (py[m] & 1))) = oct;
                  if constexpr (SPLIT) {
                    if (has_partner && py[m] == send_row && half != p.mute_half) {  // ... and the partner's, for its rows next to this one
                      const unsigned epoch = (unsigned)(p.T - t);
                      unsigned long long* g8 = p.mbox + mbox_bwd_slot(img, half, epoch & 1) + (k >> 4) * 128 + px[m] * 8 + ((ch >> 3) & 1) * 4;
                      const unsigned long long tag = (unsigned long long)epoch << 32;
                      __hip_atomic_store(g8 + 0, tag | oct[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                      __hip_atomic_store(g8 + 1, tag | oct[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                      __hip_atomic_store(g8 + 2, tag | oct[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                      __hip_atomic_store(g8 + 3, tag | oct[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                  }
                }
              }
            }
          }
        }
      }
    }
    SF_GRUB_STAMP(0)  // gate backward, stores, tile writes, hand-off stores
    if (t == 0) break;
    if constexpr (!CW) request(t - 1);  // arrives under the K loop
    if constexpr (SPLIT) {
      if (has_partner && wave == 0) {  // the partner's boundary row of dgh_t: 1536 granules, 24 per lane, swept until every tag matches
        const unsigned epoch = (unsigned)(p.T - t);
        const unsigned long long* src = p.mbox + mbox_bwd_slot(img, half ^ 1, epoch & 1);
        constexpr int PER = MBB_ROW / 64;
        unsigned v[PER];
        // the lane id through an empty asm: the 24 LDS addresses below are recomputed per step (a v_add each) instead of being kept - and spilled -
        // across the time loop
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        for (unsigned spins = 0;; ++spins) {
          bool all = true;
#pragma unroll
          for (int kk = 0; kk < PER; ++kk) {
            const int gi = lane_l + 64 * kk;
            const unsigned long long x = __hip_atomic_load(src + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[kk] = (unsigned)x;
            all = all && (((gi >> 3) & 15) >= p.W || (unsigned)(x >> 32) == epoch);
          }
          if (__all(all)) break;
          if (spins >= p.spin_limit) { mb_failed = 1; s_ctl[1] = 1u; break; }
          __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int kk = 0; kk < PER; ++kk) {
          const int gi = lane_l + 64 * kk, ck = gi >> 7, pxl = (gi >> 3) & 15, oh = (gi >> 2) & 1, dw = gi & 3;
          if (pxl < p.W) *reinterpret_cast<unsigned*>(lds_t + ck * BP_CHUNK_B + (recv_row * 16 + pxl) * PIX_B + 16 * (oh ^ (recv_row & 1)) + 4 * dw) = v[kk];
        }
      }
    }

    if constexpr (CW) {
      if (has_partner && wave == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) store_k(k, t);
      }
    }
    SF_GRUB_STAMP(1)  // request + the partner's row
    // ---- carry = conv3x3^T(dgh_t, Wh): D[channel][pixel], K = 3 * hidp ----
    f32x16 acc[MFW][NFW];
#pragma unroll
    for (int m = 0; m < MFW; ++m)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][nf][i] = 0.f;
    // CW: the K loop unrolled, every wait counted.  A CU moves ~10 bytes per clock to and from HBM (MI355X_MICROARCH.md) - 224 KB per step here,
    // 10 us - so its memory pipe has to be busy all the time, never with a burst that the next wait sits out: the step's 12 stores are issued by the
    // gate backward and drain during the hand-off, the 24 loads of the next step to process go two per chunk (taps 4 and 7, behind the chunk's
    // weight DMA) through the whole K loop.  At chunk ci the awaited weights (chunk `it`) were issued two chunks ago; younger than them are the
    // loads of chunks ci-2 and ci-1, this wave's pieces of chunk it+1 and - for ci < 2 - the gate backward's 12 stores and its hand-off stores (24
    // in a sending wave): exactly those may stay in flight.
    auto chunk_cw = [&](auto ci_tag) {
      if constexpr (CW) {
        constexpr int CI = decltype(ci_tag)::value;
        const int it = (p.T - 1 - t) * CHUNKS + CI, cur = it % RING;
        asm volatile("" : "+v"(pxl32), "+v"(pxj32[0]), "+v"(pxj32[1]));  // offsets derived from these are rebuilt per chunk, not kept (and spilled)
        // loads per chunk: two in chunks 0..3 (taps 4 and 7), one in chunks 4..11 (tap 4); S2 = those of the two previous chunks (+ this step's 12 stores)
        constexpr int SC[14] = {1, 1, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1};  // s(ci - 2), s(ci - 1) at [ci], [ci + 1]
        constexpr int S2 = SC[CI] + SC[CI + 1] + (CI < 2 ? 12 : 0);
        constexpr int PW = PIECES / WAVES;
        if (CI < 2 && t == p.T - 1) {  // the first step processed: no spread operations behind it yet - the plain count (waits for more, never less)
          if (many) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        } else if (it + 1 < total_chunks) {
          const int extra = (many ? 1 : 0) + 2 * (CI < 2 && sender ? 1 : 0);
          if (extra == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + S2) : "memory");
          else if (extra == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + S2 + 1) : "memory");
          else if (extra == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + S2 + 24) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + S2 + 25) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (CI == 0) SF_GRUB_STAMP(2)
        const bool more = it + 2 < total_chunks;
        const bool stage_late = wave >= 4;
        const int nci = (CI + 2) % CHUNKS, nbuf = (it + 2) % RING;
        if (more && !stage_late) issue_weights(nci, nbuf);
        const char* inb = lds_t + CI * BP_CHUNK_B;
        const char* wb = lds + cur * WB + b_lane;
        bf16x8 fa[2], fb[2];
        auto load_tap = [&](int tap, bf16x8& a, bf16x8& b) __attribute__((always_inline)) {
          a = *reinterpret_cast<const bf16x8*>(inb + a_off[0][tap]);
          b = *reinterpret_cast<const bf16x8*>(wb + (tap * HID + nf0 * 32) * PIX_B);
        };
        load_tap(0, fa[0], fb[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tap & 1], fa[tap & 1], acc[0][0], 0, 0, 0);
          if (tap == 3 && more && stage_late) issue_weights(nci, nbuf);
          if (tap == 4 || (tap == 7 && CI < 4)) {
            const int i = CI < 4 ? 2 * CI + (tap == 7) : CI + 4;  // issue order: g_seq (needed first, by the carry sum), the saved gates, h_{t-1}
            load_k(i < 4 ? 12 + i : i - 4, t - 1);
          }
        }
      }
    };
    if constexpr (CW) {
      chunk_cw(std::integral_constant<int, 0>{}); chunk_cw(std::integral_constant<int, 1>{}); chunk_cw(std::integral_constant<int, 2>{});
      chunk_cw(std::integral_constant<int, 3>{}); chunk_cw(std::integral_constant<int, 4>{}); chunk_cw(std::integral_constant<int, 5>{});
      chunk_cw(std::integral_constant<int, 6>{}); chunk_cw(std::integral_constant<int, 7>{}); chunk_cw(std::integral_constant<int, 8>{});
      chunk_cw(std::integral_constant<int, 9>{}); chunk_cw(std::integral_constant<int, 10>{}); chunk_cw(std::integral_constant<int, 11>{});
    } else
    for (int ci = 0; ci < CHUNKS; ++ci) {
      const int it = (p.T - 1 - t) * CHUNKS + ci, cur = it % RING;
      // chunk `it` has landed; chunk it + 1 (this wave's pieces of it: the youngest DMA) may stay in flight.  Counted wait: loads
      // return in order, so "at most the younger chunk's pieces outstanding" implies this chunk is complete (pending stores or
      // the gate prefetch can only make the wait longer, never shorter).
      if (it + 1 < total_chunks) {
        if (many) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES / WAVES + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES / WAVES) : "memory");
      } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();  // ... everybody's; the tile writes of this step are visible; stage (it + 2) % RING (chunk it - 1) is free
      if (ci == 0) SF_GRUB_STAMP(2)  // first chunk's wait + barrier
      const bool more = it + 2 < total_chunks;
      const bool stage_late = MFW == 1 && wave >= 4;  // two waves per SIMD: they issue at different taps
      const int nci = (ci + 2) % CHUNKS, nbuf = (it + 2) % RING;
      if (more && !stage_late) issue_weights(nci, nbuf);
      const char* inb = lds_t + ci * BP_CHUNK_B;
      const char* wb = lds + cur * WB + b_lane;
      bf16x8 fa[2][MFW], fb[2][NFW];
      auto load_tap = [&](int tap, bf16x8 (&a)[MFW], bf16x8 (&b)[NFW]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MFW; ++m) a[m] = *reinterpret_cast<const bf16x8*>(inb + a_off[m][tap]);
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) b[nf] = *reinterpret_cast<const bf16x8*>(wb + (tap * HID + (nf0 + nf) * 32) * PIX_B);
      };
      load_tap(0, fa[0], fb[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_tap(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
#pragma unroll
        for (int m = 0; m < MFW; ++m)
#pragma unroll
          for (int nf = 0; nf < NFW; ++nf)
            acc[m][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tap & 1][nf], fa[tap & 1][m], acc[m][nf], 0, 0, 0);
        if (tap == 3 && more && stage_late) issue_weights(nci, nbuf);
      }
    }
    __syncthreads();  // every wave is done reading the tile of step t
    SF_GRUB_STAMP(3)  // K loop

    // dh_{t-1} = [g_seq_{t-1}] + dd_t + carry   (summation order of the per-step path: (g_seq + direct) + carry)
#pragma unroll
    for (int m = 0; m < MFW; ++m) {
      const long long ptm = (long long)(t - 1) * step_px + pix_i[m];
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 base = dd[m][nf][g];
          if constexpr (CW) { if (p.g_seq) base = gsv[g] + dd[m][nf][g]; }
          else if (p.g_seq) base = *reinterpret_cast<const f32x4*>(p.g_seq + ptm * p.gs_s + (nf0 + nf) * 32 + 8 * g + cq) + dd[m][nf][g];
#pragma unroll
          for (int c = 0; c < 4; ++c) dh[m][nf][g][c] = base[c] + acc[m][nf][4 * g + c];
          if constexpr (SPLIT) {  // a hand-off timed out (flag set before the K loop's barriers): every gradient from here on is NaN
            if (s_ctl[1]) dh[m][nf][g] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
          }
        }
    }
    SF_GRUB_STAMP(4)  // carry sum
  }
#ifdef SF_EXP_GRU_CLK
  if (SPLIT && lane == 0 && ticket == 100 && (wave == 0 || wave == 5))
    printf("bwd ticket %d wave %d: per step (ns) gate backward + stores %.0f | request + poll %.0f | chunk-0 wait %.0f | K loop %.0f | carry sum %.0f | sum %.0f\n", ticket, wave,
           10.0 * bph[0] / p.T, 10.0 * bph[1] / p.T, 10.0 * bph[2] / p.T, 10.0 * bph[3] / p.T, 10.0 * bph[4] / p.T, 10.0 * (bph[0] + bph[1] + bph[2] + bph[3] + bph[4]) / p.T);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (SPLIT) {
    if (mb_failed && lane == 0) atomicOr(reinterpret_cast<unsigned*>(p.mbox), 1u);
  }
}

}  // namespace

// mailbox of the two-workgroups-per-image kernel: 2 directions x 2 parities x one boundary row of granules per image + an error word
#ifdef SF_TEST_HOOKS
extern "C" void sf_convgru_seq_debug(int32_t spin_limit, int32_t mute_half) {
  g_spin_limit = spin_limit > 0 ? spin_limit : (int)MB_SPIN_LIMIT;
  g_mute_half = mute_half;
}
#endif

extern "C" size_t sf_convgru_seq_fwd_workspace_bytes(int32_t n, int32_t h, int32_t hidp) {
  if (n <= 0 || h <= 8 || hidp <= 32) return 0;
  return (size_t)mbox_slot(n, 0, 0) * sizeof(unsigned long long);
}

extern "C" int sf_convgru_seq_fwd(sfTensor gx, sfTensor h0, int32_t T, int32_t n, int32_t h, int32_t w, const void* wpacked,
                                  const float* bias_packed, int32_t hidp, sfTensor hs, sfTensor gates, void* workspace, size_t workspace_bytes,
                                  int32_t dtype, sfStream stream) {
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
  // two workgroups per image (8 rows each, boundary rows through the mailbox) when the map has more than 8 rows and both N blocks
  // are in use.  2n <= number of CUs is a SPEED heuristic only (one round of workgroups): correctness does not depend on residency,
  // partners are assigned by start-order tickets (see the mailbox comment)
  static const bool no_split = getenv("SF_GRU_NO_SPLIT") != nullptr;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
  }
  const size_t need = sf_convgru_seq_fwd_workspace_bytes(n, h, hidp);
  const bool have_ws = workspace && need > 0 && workspace_bytes >= need;
  // ticket counter + granule slots; word 0 (the sticky error word) belongs to the caller
  if (have_ws && sf_fill_async((char*)workspace + 8, 0, need - 8, st) != hipSuccess) { sf_set_error("sf_convgru_seq_fwd: mailbox memset failed"); return 2; }
  if (!no_split && nblk == 2 && h > 8 && 2 * n <= cus && have_ws) {
    p.mbox = (unsigned long long*)workspace;
    p.spin_limit = (unsigned)g_spin_limit; p.mute_half = g_mute_half;
    const long long out_px = (long long)T * n * h * w;
    static const bool no_f4 = getenv("SF_GRU_NO_F4") != nullptr;  // A/B switch: the vmcnt(0) time loop with the outputs stored in one burst
    const bool f4 = !no_f4 && p.gx_bf && p.gates_bf && hidp == 64 && out_px * p.hs_s * 4 < (1ll << 31) && out_px * p.gates_s * 2 < (1ll << 31);
    if (f4) hipLaunchKernelGGL((convgru_seq_fwd_kernel<2, true, true, true>), dim3(2 * n), dim3(512), 0, st, p);
    else if (p.gx_bf) hipLaunchKernelGGL((convgru_seq_fwd_kernel<2, true, true>), dim3(2 * n), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((convgru_seq_fwd_kernel<2, false, true>), dim3(2 * n), dim3(512), 0, st, p);
    SF_CHECK_LAUNCH("convgru_seq_fwd (split)");
    return 0;
  }
#define SF_GRU_SEQ(NB_, BF_) hipLaunchKernelGGL((convgru_seq_fwd_kernel<NB_, BF_>), dim3(n), dim3(256 * NB_), 0, st, p)
  if (nblk == 1) { if (p.gx_bf) SF_GRU_SEQ(1, true); else SF_GRU_SEQ(1, false); }
  else { if (p.gx_bf) SF_GRU_SEQ(2, true); else SF_GRU_SEQ(2, false); }
#undef SF_GRU_SEQ
  SF_CHECK_LAUNCH("convgru_seq_fwd");
  return 0;
}

extern "C" size_t sf_convgru_seq_bwd_workspace_bytes(int32_t n, int32_t h, int32_t hidp) {
  if (n <= 0 || h <= 8 || hidp != 64) return 0;
  return (size_t)mbox_bwd_slot(n, 0, 0) * sizeof(unsigned long long);
}

extern "C" int sf_convgru_seq_bwd(sfTensor g_seq, sfTensor g_last, sfTensor gates, sfTensor hs, int32_t T, int32_t n, int32_t h, int32_t w,
                                  const void* wpacked_t, int32_t hidp, sfTensor dgx, sfTensor dgh, void* workspace, size_t workspace_bytes,
                                  int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_BF16, "sf_convgru_seq_bwd: the persistent sequence kernel is built for the SF_BF16 kernels (got dtype %d)", dtype);
  SF_REQUIRE(h >= 1 && w >= 1 && h <= 16 && w <= 16, "sf_convgru_seq_bwd: one workgroup owns a whole map: H, W <= 16 (got %dx%d)", h, w);
  SF_REQUIRE(hidp == 32 || hidp == 64, "sf_convgru_seq_bwd: hidp=%d (32 or 64)", hidp);
  SF_REQUIRE(gates.ptr && gates.dtype == SF_BF16 && gates.c == 4 * hidp && ((uintptr_t)gates.ptr & 7) == 0 && gates.stride % 4 == 0,
             "sf_convgru_seq_bwd: gates [..,4*hidp] bf16-stored, 8-byte aligned channel quads");
  SF_REQUIRE(hs.ptr && hs.dtype == SF_F32 && hs.c == hidp && ((uintptr_t)hs.ptr & 15) == 0 && hs.stride % 4 == 0, "sf_convgru_seq_bwd: hs [..,hidp] fp32, 16-byte aligned pixels");
  auto grad_ok = [&](const sfTensor& t) { return !t.ptr || (t.dtype == SF_F32 && t.c == hidp && ((uintptr_t)t.ptr & 15) == 0 && t.stride % 4 == 0); };
  SF_REQUIRE(grad_ok(g_seq) && grad_ok(g_last), "sf_convgru_seq_bwd: g_seq / g_last [..,hidp] fp32, 16-byte aligned pixels (or null)");
  auto out_ok = [&](const sfTensor& t) { return t.ptr && t.dtype == SF_BF16 && t.c == 3 * hidp && ((uintptr_t)t.ptr & 15) == 0 && t.stride % 8 == 0; };
  SF_REQUIRE(out_ok(dgx) && out_ok(dgh) && wpacked_t, "sf_convgru_seq_bwd: dgx / dgh [..,3*hidp] bf16-stored, 16-byte aligned pixels; weights non-null");
  if (T <= 0 || n <= 0) return 0;
  GruSeqBwdParams p{};
  p.g_seq = (const float*)g_seq.ptr; p.gs_s = g_seq.stride;
  p.g_last = (const float*)g_last.ptr; p.gl_s = g_last.stride;
  p.gates = (const __bf16*)gates.ptr; p.gates_s = gates.stride;
  p.hs = (const float*)hs.ptr; p.hs_s = hs.stride;
  p.dgx = (__bf16*)dgx.ptr; p.dgx_s = dgx.stride;
  p.dgh = (__bf16*)dgh.ptr; p.dgh_s = dgh.stride;
  p.wp = wpacked_t;
  p.T = T; p.n = n; p.H = h; p.W = w; p.hidp = hidp;
  hipStream_t st = (hipStream_t)stream;
  // (4 waves with two M fragments each - fewer fragment reads per MFMA - measured SLOWER, 695 vs 580 us: one wave per SIMD has
  // nobody to cover its LDS waits and its gate arithmetic)
  static const bool no_split = getenv("SF_GRU_NO_SPLIT") != nullptr;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
  }
  const size_t need = sf_convgru_seq_bwd_workspace_bytes(n, h, hidp);
  const bool have_ws = workspace && need > 0 && workspace_bytes >= need;
  if (have_ws && sf_fill_async((char*)workspace + 8, 0, need - 8, st) != hipSuccess) { sf_set_error("sf_convgru_seq_bwd: mailbox memset failed"); return 2; }
  if (!no_split && hidp == 64 && h > 8 && 2 * n <= cus && have_ws) {  // two workgroups per map (see sf_convgru_seq_fwd)
    p.mbox = (unsigned long long*)workspace;
    p.spin_limit = (unsigned)g_spin_limit; p.mute_half = g_mute_half;
    const long long out_px = (long long)T * n * h * w;
    static const bool no_cw = getenv("SF_GRU_NO_F4") != nullptr;  // A/B switch, as in the forward launcher
    if (!no_cw && out_px * p.dgx_s * 2 < (1ll << 31) && out_px * p.dgh_s * 2 < (1ll << 31) && out_px * p.gates_s * 2 < (1ll << 31) && out_px * p.hs_s * 4 < (1ll << 31) &&
        (!p.g_seq || out_px * p.gs_s * 4 < (1ll << 31)))
      hipLaunchKernelGGL((convgru_seq_bwd_kernel<2, 1, true, true>), dim3(2 * n), dim3(512), 0, st, p);
    else
      hipLaunchKernelGGL((convgru_seq_bwd_kernel<2, 1, true>), dim3(2 * n), dim3(512), 0, st, p);
    SF_CHECK_LAUNCH("convgru_seq_bwd (split)");
    return 0;
  }
  if (hidp == 64) hipLaunchKernelGGL((convgru_seq_bwd_kernel<2, 1>), dim3(n), dim3(512), 0, st, p);
  else hipLaunchKernelGGL((convgru_seq_bwd_kernel<1, 1>), dim3(n), dim3(512), 0, st, p);
  SF_CHECK_LAUNCH("convgru_seq_bwd");
  return 0;
}
