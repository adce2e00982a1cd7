// Shared by the fp32 and bf16 implicit-GEMM convolution kernels: parameter block, fused epilogues
// (the MFMA C/D fragment layout is dtype independent on gfx950), host-side parameter helpers.
#pragma once
#include "sf_common.h"

namespace sfconv {

constexpr int TILE_W = 16;        // output tile width (pixels); tile height = 4 rows per wave
constexpr int KC = 16;            // input channels per K chunk

enum { EPI_LINEAR = 0, EPI_SIGMOID = 1, EPI_LSTM = 2, EPI_GRU = 3 };

struct ConvParams {
  const float* src0; const float* src1;
  int c0, c1, s0, s1;
  int idiv0, imod0, idiv1, imod1;  // image-index remap of the sources (see sfTensor)
  int N, H, W, tiles_x, tiles_y;
  const void* wp;      // packed image [nblk][chunks][9][32*NF][16] (fp32 or bf16)
  const float* bias;   // [nblk*32*NF] or null
  int chunks_total;    // (c0_decl + c1_decl)/16 of the packed image
  // linear / sigmoid epilogue
  float* out; int out_c, out_s;
  // lstm epilogue
  const float* c_prev; int cprev_s;
  float* c_out; int cout_s;
  float* h_out; int hout_s;
  float* gates; int gates_s;
  int hidp;
  // gru epilogue (h_out / gates / hidp shared with lstm): precomputed x-part [z|r|n] and previous state
  const float* gx; int gx_s;
  const float* h_prev; int hprev_s;
  // storage types (bf16 kernel only; linear / sigmoid epilogues): non-zero = bf16 elements behind src0 / src1 / out
  int bf0, bf1, out_bf;
  int gates_bf;  // lstm epilogue: the saved gates are stored as bf16 (they are read by the backward pass only)
  int hout_bf;   // lstm epilogue (bf16 kernel): the new hidden state is stored as bf16 (it is only ever read as an MFMA operand)
  // bf16 kernel, linear epilogue: per-tile sum / sum of squares of the stored outputs, [tile][stats_np][2] (or null)
  float* stats; int stats_np;
  // bf16 kernel, linear epilogue, single-image tiles: GROUPED weights - image n uses the packed image n / wgroup (0 = one image
  // for all) - and a per-group bias that depends on the pixel's border class (top / interior / bottom x left / interior / right):
  // the folded BatchNorm in front of this convolution (sf_conv3x3_fwd_folded).  `bias` is null then.
  int wgroup; long long wgroup_bytes;
  const float* bias_tab; int np;  // [groups][9][np] or null
  // bf16 kernel, transposed linear epilogue: the BatchNorm backward's affine map applied to the result before it is stored,
  //   out = A * acc + B * x + K,  coefficients bnb_coef[group][3][bnb_c] (group = image / bnb_group), x = the BatchNorm's input
  // (bf16, the output's pixels and channel lanes, pixel stride bnb_xs) - sf_conv3x3_bwd_data_bn
  const float* bnb_coef; const void* bnb_x; int bnb_xs, bnb_group, bnb_c;
  // bf16 kernel, split-K launches (sf_conv3x3_fwd_splitk; few small images with many input channels): workgroup z handles input channels
  // [z * split_c, (z + 1) * split_c) of src0 and stores its raw fp32 partial sums to out + z * split_out (no bias); 0 = not split
  int split_c; long long split_out;
  // bf16 / f16 kernel, fp32-stored src0 only: src0 is read as FOUR SHIFTED VIEWS of one tensor stacked as channels - virtual channel chunk g
  // (g = chunk0 + ci; shift4 = chunks per view) is chunk g % shift4 of view s = g / shift4, whose pixel (y, x) is the tensor's pixel
  // (y + 2 (s >> 1) - 1, x + 2 (s & 1) - 1), zero outside the image: a 5x5 'same' convolution as ONE 3x3 convolution without the padded,
  // four-times-copied input of sf_pad_shift_stack4_fwd (sf_conv5x5_fwd).  c0 = the virtual channel count of this launch / slice; 0 = off.
  int shift4, chunk0;
  // one-wave-per-SIMD bf16 kernel (conv3x3_bf16_persist4.hip, MODE 3): the 2x2 / stride-2 max-pooling behind this convolution taken in the epilogue
  // (sf_conv3x3_fwd_folded_pool): `out` is NOT written; pool_out = the pooled tensor (bf16, [n][H/2][W/2][pool_s]), image i stored as image
  // perm(i) (pool_L > 0: (l * pool_T + t) * pool_B + b  ->  (t * pool_L + l) * pool_B + b, as sf_maxpool2_fwd); pool_route = sf_maxpool2_route_fwd's
  // routing record (one 16-bit word per pooled pixel and channel octet of the out_c channels, 2 bits per channel: first maximum in row-major order)
  void* pool_out; int pool_s, pool_L, pool_T, pool_B;
  unsigned short* pool_route;
  // SF_F32E kernels (conv3x3_f32e.hip): device word with max |src0| (sfTensor::amax, sf_amax) of a GRADIENT source - the kernel scales the source by the
  // power of two that puts that magnitude at 2^14 before it splits it into fp16 parts, and the accumulators by the inverse behind the K loop; null = as is.
  // chunks_total counts the packed image's VIRTUAL chunks there (3 per real 16-channel chunk).
  const float* amax0;
};

// border class of an output pixel (needs H, W >= 2); pixels outside the image (ragged tiles) get some valid class
__device__ __forceinline__ int border_cls(int py, int px, int H, int W) {
  return (py == 0 ? 0 : (py >= H - 1 ? 2 : 1)) * 3 + (px == 0 ? 0 : (px >= W - 1 ? 2 : 1));
}


// acc[mf][nf][reg]: wave `wave` owns tile rows 4*wave..4*wave+3; M fragment mf = rows 2*mf, 2*mf+1 (16 px each).
template <int NF, int EPI>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[2][NF], const ConvParams& p, int n, int nb, int y0, int x0,
                                              int wave, int r, int kh, float* lds_stats = nullptr) {
  constexpr int NB = 32 * NF;
  // The recurrent epilogues read state / pre-activations per output element.  All reads of one M fragment (16 elements)
  // are issued first, from clamped in-image addresses and without per-element conditions, then the arithmetic, then
  // the stores: written element by element (load under `if (inside)`, use, store) hipcc emits load -> vmcnt(0) ->
  // stores -> next load, and because vmcnt also counts the stores each of the 32 elements paid a full memory round
  // trip (measured: 38 of the 40 us of a ConvGRU step).
  if constexpr (EPI == EPI_LSTM) {
    static_assert(NF == 4, "LSTM epilogue needs the 4 gates in one wave");
    const int hc = nb * 32 + r;
    if (hc < p.hidp) {
      float bi = 0.f, bf = 0.f, bo = 0.f, bg = 0.f;
      if (p.bias) {
        bi = p.bias[nb * NB + r]; bf = p.bias[nb * NB + 32 + r];
        bo = p.bias[nb * NB + 64 + r]; bg = p.bias[nb * NB + 96 + r];
      }
      const size_t pix_safe = (size_t)(n * p.H + y0) * p.W + x0;  // the tile origin is always inside the image
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        size_t pix[16]; bool ok[16]; float cp[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int rr = frag_row(reg, kh);
          const int py = y0 + 4 * wave + 2 * mf + (rr >> 4), px = x0 + (rr & 15);
          ok[reg] = py < p.H && px < p.W;
          pix[reg] = ok[reg] ? (size_t)(n * p.H + py) * p.W + px : pix_safe;
        }
        if (p.c_prev) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) cp[reg] = p.c_prev[pix[reg] * p.cprev_s + hc];
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) cp[reg] = 0.f;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float gi = sf_sigmoid(acc[mf][0][reg] + bi);
          const float gf = sf_sigmoid(acc[mf][1][reg] + bf);
          const float go = sf_sigmoid(acc[mf][2][reg] + bo);
          const float gg = sf_tanh(acc[mf][3][reg] + bg);
          const float cn = gf * cp[reg] + gi * gg;
          if (ok[reg]) {
            p.c_out[pix[reg] * p.cout_s + hc] = cn;
            p.h_out[pix[reg] * p.hout_s + hc] = go * sf_tanh(cn);
            if (p.gates) {
              if (p.gates_bf) {
                __bf16* gp = reinterpret_cast<__bf16*>(p.gates) + pix[reg] * p.gates_s + hc;
                gp[0] = (__bf16)gi; gp[p.hidp] = (__bf16)gf; gp[2 * p.hidp] = (__bf16)go; gp[3 * p.hidp] = (__bf16)gg;
              } else {
                float* gp = p.gates + pix[reg] * p.gates_s + hc;
                gp[0] = gi; gp[p.hidp] = gf; gp[2 * p.hidp] = go; gp[3 * p.hidp] = gg;
              }
            }
          }
        }
      }
    }
  } else if constexpr (EPI == EPI_GRU) {
    static_assert(NF == 3, "GRU epilogue: z, r and the candidate's h-part in one wave");
    const int hc = nb * 32 + r;
    if (hc < p.hidp) {
      const float b2 = p.bias ? p.bias[nb * NB + 64 + r] : 0.f;
      const size_t pix_safe = (size_t)(n * p.H + y0) * p.W + x0;
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        size_t pix[16]; bool ok[16]; float gz[16], gr[16], gn[16], hp[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int rr = frag_row(reg, kh);
          const int py = y0 + 4 * wave + 2 * mf + (rr >> 4), px = x0 + (rr & 15);
          ok[reg] = py < p.H && px < p.W;
          pix[reg] = ok[reg] ? (size_t)(n * p.H + py) * p.W + px : pix_safe;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float* gx = p.gx + pix[reg] * p.gx_s + hc;
          gz[reg] = gx[0]; gr[reg] = gx[p.hidp]; gn[reg] = gx[2 * p.hidp];
        }
        if (p.h_prev) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) hp[reg] = p.h_prev[pix[reg] * p.hprev_s + hc];
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) hp[reg] = 0.f;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float z = sf_sigmoid(acc[mf][0][reg] + gz[reg]);
          const float rg = sf_sigmoid(acc[mf][1][reg] + gr[reg]);
          const float h2 = acc[mf][2][reg] + b2;
          const float cand = sf_tanh(sf_gru_cand_arg(gn[reg], rg, h2));
          if (ok[reg]) {
            p.h_out[pix[reg] * p.hout_s + hc] = sf_gru_blend(z, cand, hp[reg]);
            if (p.gates) {
              if (p.gates_bf) {
                __bf16* gp = reinterpret_cast<__bf16*>(p.gates) + pix[reg] * p.gates_s + hc;
                gp[0] = (__bf16)z; gp[p.hidp] = (__bf16)rg; gp[2 * p.hidp] = (__bf16)cand; gp[3 * p.hidp] = (__bf16)h2;
              } else {
                float* gp = p.gates + pix[reg] * p.gates_s + hc;
                gp[0] = z; gp[p.hidp] = rg; gp[2 * p.hidp] = cand; gp[3 * p.hidp] = h2;
              }
            }
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int co = nb * NB + nf * 32 + r;
      if (co < p.out_c) {  // out_c is a multiple of 16: both lanes of an (even, odd) channel pair take the same side
        const float bv = p.bias ? p.bias[co] : 0.f;
        float s1 = 0.f, s2 = 0.f;  // BatchNorm statistics of THIS lane's channel over its valid pixels (stored, i.e. rounded, values)
        if (p.out_bf) {
          // bf16 output: registers 2k / 2k+1 are horizontally adjacent pixels P / P+1 of channel `co`.  Lane pairs swap one
          // value (DPP quad_perm [1,0,3,2]) so that the even lane owns channels (co, co+1) of P and the odd lane channels
          // (co-1, co) of P+1: one packed 4-byte store per lane and register pair instead of two 2-byte stores.
          __bf16* ob = reinterpret_cast<__bf16*>(p.out);
          const int odd = r & 1;
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float a = acc[mf][nf][2 * k] + bv, b = acc[mf][nf][2 * k + 1] + bv;
              if constexpr (EPI == EPI_SIGMOID) { a = sf_sigmoid(a); b = sf_sigmoid(b); }
              a = (float)(__bf16)a; b = (float)(__bf16)b;  // the stored values (packing them below is exact)
              if (lds_stats) {
                const int ra = frag_row(2 * k, kh), rb = frag_row(2 * k + 1, kh);
                const bool va = y0 + 4 * wave + 2 * mf + (ra >> 4) < p.H && x0 + (ra & 15) < p.W;
                const bool vb = y0 + 4 * wave + 2 * mf + (rb >> 4) < p.H && x0 + (rb & 15) < p.W;
                s1 += (va ? a : 0.f) + (vb ? b : 0.f); s2 += (va ? a * a : 0.f) + (vb ? b * b : 0.f);
              }
              const float send = odd ? a : b;
              const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
              const int rr = frag_row(2 * k + odd, kh);
              const int py = y0 + 4 * wave + 2 * mf + (rr >> 4), px = x0 + (rr & 15);
              if (py < p.H && px < p.W) {
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t pr = odd ? f32x2_t{recv, b} : f32x2_t{a, recv};
                *reinterpret_cast<bf16x2_t*>(ob + ((size_t)(n * p.H + py) * p.W + px) * p.out_s + (co - odd)) = __builtin_convertvector(pr, bf16x2_t);
              }
            }
        } else {
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int rr = frag_row(reg, kh);
              const int py = y0 + 4 * wave + 2 * mf + (rr >> 4), px = x0 + (rr & 15);
              if (py < p.H && px < p.W) {
                float v = acc[mf][nf][reg] + bv;
                if constexpr (EPI == EPI_SIGMOID) v = sf_sigmoid(v);
                s1 += v; s2 += v * v;
                p.out[((size_t)(n * p.H + py) * p.W + px) * p.out_s + co] = v;
              }
            }
        }
        if (lds_stats) {  // block-uniform; lanes r and r+32 hold the same channel
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          if (kh == 0) { atomicAdd(lds_stats + nf * 32 + r, s1); atomicAdd(lds_stats + NB + nf * 32 + r, s2); }
        }
      }
    }
  }
}

// Linear / sigmoid epilogue of the TRANSPOSED product (bf16 kernel without BatchNorm statistics: the MFMA is issued with
// the weight fragment as its A operand, so D[i][j] has i = output channel, j = pixel): lane (r, kh) owns ONE pixel of
// each M fragment and, per 32-channel N fragment, the four channel quads 8g + 4kh .. +3 (g = reg / 4).  Outputs leave as
// 16-byte stores: fp32 a quad at a time; bf16 after `v_permlane32_swap` pairs quad g of the two half-waves into the
// octets 8g..8g+7 (lower lanes) and 8g+8..8g+15 (upper lanes) - 2*NF*2 `dwordx4` stores per lane instead of the
// 2*NF*8 4-byte stores of the channel-per-lane layout (the store tail is issue-bound, not bandwidth-bound).
template <int NF, int EPI>
__device__ __forceinline__ void conv_epilogue_tr(f32x16 (&acc)[2][NF], const ConvParams& p, int n, int nb, int y0, int x0,
                                                 int wave, int r, int kh) {
  static_assert(EPI == EPI_LINEAR || EPI == EPI_SIGMOID || EPI == EPI_LSTM, "transposed epilogue: linear / sigmoid / lstm");
  constexpr int NB = 32 * NF;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  auto pk = [](float a, float b) -> unsigned { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t)); };
  if constexpr (EPI == EPI_LSTM) {
    // Fragments 0..3 are the gates i, f, o, g of hidden channels nb*32 + 8g + 4kh + c of this lane's pixel: previous cell
    // state, new cell / hidden state and the saved gates all move as 16-byte quads (bf16 gates as octets after the
    // half-wave swap) - 20 memory instructions per M fragment where the channel-per-lane layout needs 112.
    static_assert(NF == 4, "LSTM epilogue needs the 4 gates in one wave");
    const int hb = nb * 32 + 4 * kh;
    const size_t pix_safe = (size_t)(n * p.H + y0) * p.W + x0;  // the tile origin is always inside the image
    // The bias is ALREADY in the accumulators (the kernel starts them at the bias, conv3x3_bf16.hip), and the previous cell state of BOTH pixel
    // fragments is requested before anything is stored: the stores of fragment 0 may alias the loads of fragment 1 as far as the compiler
    // knows, so left in the loop the two load latencies (and the bias loads' before them) were serial - 34 of the cell's 97 us were epilogue
    // (tools/ablate_lstm_cell.sh: without sigmoid / tanh 94 us, without the saved gates 91 us, without any epilogue 62 us).
    f32x4 cpa[2][4];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int py = y0 + 4 * wave + 2 * mf + (r >> 4), px = x0 + (r & 15);
      const size_t pix = (py < p.H && px < p.W) ? (size_t)(n * p.H + py) * p.W + px : pix_safe;
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // unconditional loads from clamped addresses (see conv_epilogue)
        const int hq = (hb + 8 * g < p.hidp) ? hb + 8 * g : 0;
#ifdef SF_EXP_LSTM_NOCPREV   // ablation: no previous-cell-state loads
        cpa[mf][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
        cpa[mf][g] = p.c_prev ? *reinterpret_cast<const f32x4*>(p.c_prev + pix * p.cprev_s + hq) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
      }
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int py = y0 + 4 * wave + 2 * mf + (r >> 4), px = x0 + (r & 15);
#ifdef SF_EXP_LSTM_NOSTORE   // ablation: every store of the epilogue predicated off by a value the compiler cannot know
      const bool ok = py < p.H && px < p.W && acc[mf][0][0] == 12345.678f;
#else
      const bool ok = py < p.H && px < p.W;
#endif
      const size_t pix = ok ? (size_t)(n * p.H + py) * p.W + px : pix_safe;
      const f32x4 (&cp)[4] = cpa[mf];
      f32x4 gi[4], gf[4], go[4], gg[4], cn[4], hn[4];
      // Ablation builds of tools/ablate_lstm_cell.sh (never part of the shipped library): -DSF_EXP_LSTM_NOTRANS replaces the five transcendental
      // functions per element by one multiply each - the difference to the shipped kernel is what the gate arithmetic costs with the matrix pipes idle.
#ifdef SF_EXP_LSTM_NOTRANS
#define SF_LSTM_SIG(v) (0.25f * (v))
#define SF_LSTM_TANH(v) (0.5f * (v))
#else
#define SF_LSTM_SIG(v) sf_sigmoid(v)
#define SF_LSTM_TANH(v) sf_tanh(v)
#endif
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          gi[g][c] = SF_LSTM_SIG(acc[mf][0][4 * g + c]);
          gf[g][c] = SF_LSTM_SIG(acc[mf][1][4 * g + c]);
          go[g][c] = SF_LSTM_SIG(acc[mf][2][4 * g + c]);
          gg[g][c] = SF_LSTM_TANH(acc[mf][3][4 * g + c]);
          cn[g][c] = gf[g][c] * cp[g][c] + gi[g][c] * gg[g][c];
          hn[g][c] = go[g][c] * SF_LSTM_TANH(cn[g][c]);
        }
#undef SF_LSTM_SIG
#undef SF_LSTM_TANH
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (ok && hb + 8 * g < p.hidp) {
          *reinterpret_cast<f32x4*>(p.c_out + pix * p.cout_s + hb + 8 * g) = cn[g];
          if (!p.hout_bf) *reinterpret_cast<f32x4*>(p.h_out + pix * p.hout_s + hb + 8 * g) = hn[g];
        }
      if (p.hout_bf) {
        __bf16* hp = reinterpret_cast<__bf16*>(p.h_out) + pix * p.hout_s + nb * 32 + 8 * kh;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          unsigned ax = pk(hn[g][0], hn[g][1]), ay = pk(hn[g][2], hn[g][3]);
          unsigned bx = pk(hn[g + 1][0], hn[g + 1][1]), by = pk(hn[g + 1][2], hn[g + 1][3]);
          auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          if (ok && nb * 32 + 8 * g < p.hidp) *reinterpret_cast<u32x4_t*>(hp + 8 * g) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
        }
      }
      if (p.gates) {
        if (p.gates_bf) {
          __bf16* gp = reinterpret_cast<__bf16*>(p.gates) + pix * p.gates_s + nb * 32 + 8 * kh;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4* v = q == 0 ? gi : q == 1 ? gf : q == 2 ? go : gg;
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
              unsigned ax = pk(v[g][0], v[g][1]), ay = pk(v[g][2], v[g][3]);
              unsigned bx = pk(v[g + 1][0], v[g + 1][1]), by = pk(v[g + 1][2], v[g + 1][3]);
              auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
              auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
              if (ok && nb * 32 + 8 * g < p.hidp) *reinterpret_cast<u32x4_t*>(gp + q * p.hidp + 8 * g) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            }
          }
        } else {
          float* gp = p.gates + pix * p.gates_s + hb;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (ok && hb + 8 * g < p.hidp) {
              *reinterpret_cast<f32x4*>(gp + 8 * g) = gi[g];
              *reinterpret_cast<f32x4*>(gp + p.hidp + 8 * g) = gf[g];
              *reinterpret_cast<f32x4*>(gp + 2 * p.hidp + 8 * g) = go[g];
              *reinterpret_cast<f32x4*>(gp + 3 * p.hidp + 8 * g) = gg[g];
            }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const int py = y0 + 4 * wave + 2 * mf + (r >> 4), px = x0 + (r & 15);
    const bool ok = py < p.H && px < p.W;
    const size_t pix = ok ? (size_t)(n * p.H + py) * p.W + px : 0;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int cb = nb * NB + nf * 32;  // out_c is a multiple of 16: a 16-channel half of the fragment is in or out as a whole
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + cb + 8 * g + 4 * kh);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float t = acc[mf][nf][4 * g + c] + bv[c];
          if constexpr (EPI == EPI_SIGMOID) t = sf_sigmoid(t);
          v[4 * g + c] = t;
        }
      }
      if (p.out_bf) {
        __bf16* ob = reinterpret_cast<__bf16*>(p.out) + pix * p.out_s + cb + 8 * kh;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          unsigned ax = pk(v[4 * g], v[4 * g + 1]), ay = pk(v[4 * g + 2], v[4 * g + 3]);
          unsigned bx = pk(v[4 * g + 4], v[4 * g + 5]), by = pk(v[4 * g + 6], v[4 * g + 7]);
          auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          if (ok && cb + 8 * g < p.out_c) *reinterpret_cast<u32x4_t*>(ob + 8 * g) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
        }
      } else {
        float* of = p.out + pix * p.out_s + cb + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (ok && cb + 8 * g < p.out_c) *reinterpret_cast<f32x4*>(of + 8 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
      }
    }
  }
}

// Transposed linear epilogue with the BatchNorm backward's affine map (sf_conv3x3_bwd_data_bn): out = A * acc + B * x + K, bf16 out.
// lds_coef [3][32 * NF]: this N block's (A, B, K), staged by the kernel BEFORE its K loop (zeros past out_c).  The x quads of BOTH M fragments are
// requested before anything else (one memory latency per workgroup instead of one per fragment - the epilogue runs with the
// matrix pipe idle, 1 workgroup per CU); NF = 5 requests them per M fragment (registers).
template <int NF, int HOIST = (NF <= 4 ? 2 : 1)>
__device__ __forceinline__ void conv_epilogue_tr_bnb(f32x16 (&acc)[2][NF], const ConvParams& p, int n, int nb, int y0, int x0, int wave,
                                                     int r, int kh, const float* lds_coef) {
  constexpr int NB = 32 * NF;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  auto pk = [](float a, float b) -> unsigned { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t)); };
  bool ok[2];
  size_t pix[2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const int py = y0 + 4 * wave + 2 * mf + (r >> 4), px = x0 + (r & 15);
    ok[mf] = py < p.H && px < p.W;
    pix[mf] = ok[mf] ? (size_t)(n * p.H + py) * p.W + px : 0;  // 0 for lanes outside: a valid address
  }
  // x is read the way the result is written: as 16-byte channel OCTETS - lane (r, kh) fetches octets g + kh (g = 0, 2) of its pixel,
  // the two half-waves then trade the quads they hold for each other (`v_permlane32_swap`, the store path in reverse).  Half the load
  // instructions of 8-byte quads per lane, and every request is a full 32-byte sector pair (the epilogue's x read was bound by the
  // number of sector requests, not by latency).  Octets past out_c (never stored) read octet 0 instead of running past the tensor.
  int off[NF][2];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int j = 0; j < 2; ++j) off[nf][j] = nb * NB + nf * 32 + 16 * j < p.out_c ? nb * NB + nf * 32 + 16 * j + 8 * kh : 0;
  u32x4_t xo[HOIST][NF][2];
  auto request = [&](int slot, int mf) __attribute__((always_inline)) {
    const __bf16* xp = reinterpret_cast<const __bf16*>(p.bnb_x) + pix[mf] * p.bnb_xs;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int j = 0; j < 2; ++j) xo[slot][nf][j] = *reinterpret_cast<const u32x4_t*>(xp + off[nf][j]);
  };
  if constexpr (HOIST == 2) { request(0, 0); request(1, 1); } else request(0, 0);
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    if constexpr (HOIST == 1) { if (mf == 1) request(0, 1); }
    const int slot = HOIST == 2 ? mf : 0;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int cb = nb * NB + nf * 32;
      float v[16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // octet (dwords d0..d3) -> this lane's quads 2j and 2j + 1: (d0, d1) of the lower half-wave and (d2, d3) of the upper one stay,
        // the other two pairs cross over
        const u32x4_t o = xo[slot][nf][j];
        const auto s0 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * j + gg;
          const unsigned w0 = s0[gg], w1 = s1[gg];
          const float xv[4] = {__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xffff0000u),
                               __builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
          const float* lc = lds_coef + nf * 32 + 8 * g + 4 * kh;
          const f32x4 A = *reinterpret_cast<const f32x4*>(lc), B = *reinterpret_cast<const f32x4*>(lc + NB), Kc = *reinterpret_cast<const f32x4*>(lc + 2 * NB);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[4 * g + c] = __builtin_fmaf(A[c], acc[mf][nf][4 * g + c], __builtin_fmaf(B[c], xv[c], Kc[c]));
        }
      }
      __bf16* ob = reinterpret_cast<__bf16*>(p.out) + pix[mf] * p.out_s + cb + 8 * kh;
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned ax = pk(v[4 * g], v[4 * g + 1]), ay = pk(v[4 * g + 2], v[4 * g + 3]);
        unsigned bx = pk(v[4 * g + 4], v[4 * g + 5]), by = pk(v[4 * g + 6], v[4 * g + 7]);
        auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        if (ok[mf] && cb + 8 * g < p.out_c) *reinterpret_cast<u32x4_t*>(ob + 8 * g) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
      }
    }
  }
}

inline void set_remap(ConvParams& p, const sfTensor& a, const sfTensor& b) {
  p.idiv0 = a.idiv > 1 ? a.idiv : 1; p.imod0 = a.imod > 0 ? a.imod : 0;
  p.idiv1 = b.idiv > 1 ? b.idiv : 1; p.imod1 = b.imod > 0 ? b.imod : 0;
}

inline int check_src(const sfTensor& t, const char* name) {
  if (t.c % SF_CPAD != 0 || t.c < 0) { sf_set_error("%s: channels %d not a multiple of %d", name, t.c, SF_CPAD); return 1; }
  if (t.ptr && (t.stride % (t.dtype == SF_BF16 ? 8 : 4) != 0 || ((uintptr_t)t.ptr & 15))) { sf_set_error("%s: needs 16-byte aligned pixels (stride %d)", name, t.stride); return 1; }
  if (t.ptr && t.dtype != SF_F32 && t.dtype != SF_BF16) { sf_set_error("%s: unknown storage type %d", name, t.dtype); return 1; }
  return 0;
}

}  // namespace sfconv

// pixel tiles per image of the bf16 kernels (32x16 tiles when H > 16, else 16x16)
int sf_conv_bf16_tiles(int h, int w);
// bf16-MFMA launcher (conv3x3_bf16.hip); epi is one of sfconv::EPI_*
int sf_launch_conv_bf16(const sfconv::ConvParams& p, int nf, int nblk, int epi, hipStream_t st);
// the same kernels compiled for fp16 operands (conv3x3_f16.hip; SF_F16: fp32-stored tensors, linear / sigmoid epilogue, split-K)
int sf_launch_conv_f16(const sfconv::ConvParams& p, int nf, int nblk, int epi, hipStream_t st);
void sf_pack_weights_f16(const float* w, int O, int I, const int* nmap, int Np, const int* kmap, int Kp, int NB, int transpose,
                         void* packed, const float* bias, float* bias_packed, hipStream_t st, const float* kscale = nullptr, int groups = 1);
// the same kernels as the SF_F32E compute mode (conv3x3_f32e.hip): three fp16 products per fp32 product, fp32-stored tensors, every epilogue; the packed
// image has 3 * Kp / 16 chunks (27 * Np * Kp halves)
int sf_launch_conv_f32e(const sfconv::ConvParams& p, int nf, int nblk, int epi, hipStream_t st);
void sf_pack_weights_f32e(const float* w, int O, int I, const int* nmap, int Np, const int* kmap, int Kp, int NB, int transpose,
                          void* packed, const float* bias, float* bias_packed, hipStream_t st, const float* kscale = nullptr, int groups = 1);
// persistent variant for the large single-source bf16-stored launches (conv3x3_bf16_persist.hip); bit-identical results
bool sf_conv_bf16_persist_ok(const sfconv::ConvParams& p, int epi, int nf);
int sf_launch_conv_bf16_persist(const sfconv::ConvParams& p, int nf, int nblk, hipStream_t st);
// one-wave-per-SIMD persistent variant (conv3x3_bf16_persist4.hip): NF = 4, >= 3 K chunks (the next-item set-up assumes them), no per-channel bias; bit-identical results
bool sf_conv_bf16_persist4_ok(const sfconv::ConvParams& p, int nf);
int sf_launch_conv_bf16_persist4(const sfconv::ConvParams& p, int nblk, hipStream_t st);
// kscale (nullable): [groups][Kp] per-input-lane factors -> `groups` packed images back to back (folded BatchNorm scale)
void sf_pack_weights_bf16(const float* w, int O, int I, const int* nmap, int Np, const int* kmap, int Kp, int NB, int transpose,
                          void* packed, const float* bias, float* bias_packed, hipStream_t st, const float* kscale = nullptr, int groups = 1);
