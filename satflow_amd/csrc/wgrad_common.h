// Shared by the fp32 and bf16 weight-gradient kernels: parameter block and split-K plan.
#pragma once
#include "sf_common.h"

namespace sfwgrad {

constexpr int CO_T = 128, CI_T = 32;   // dW slab per workgroup: 128 (co) x 32 (ci) x 9 taps
constexpr int KT_W = 16;               // K tile width (pixels)
constexpr int DMA_CO_T = 128, DMA_CI_T = 64;  // slab of the all-bf16-storage kernel (conv3x3_wgrad_bf16_dma.hip)

struct WgradParams {
  const float* src0; const float* src1; int c0, c1, s0, s1;
  int idiv0, imod0, idiv1, imod1;
  const float* dout; int dc, ds;
  int N, H, W, tiles_x, tiles_y, ntiles, KS;
  float* partial; float* partial_db;
  int NpT, KpT;
  int bf;       // bf16 kernel only: the input sources src0 / src1 are stored as bf16
  int bf_dout;  // bf16 kernel only: dout is stored as bf16 (with fp32 sources: the ConvLSTM's bf16-stored gate gradients)
  int tpg, maxseg;  // all-bf16 kernel, folded BatchNorm: tiles per group (0: ungrouped), partial slots per slice
  // loader-wave kernel, fp32-stored tensors: src0 is read as four shifted views stacked as channels (sfconv::ConvParams::shift4; here in CHANNELS per
  // view, a multiple of the 32-channel ci tile): input channel block cit belongs to view (32 cit) / shift4 - sf_conv5x5_bwd_weight.  0 = off
  int shift4;
  // all-bf16 kernel (regular slabs): dout holds AT MOST ONE non-zero per aligned horizontal pixel pair and channel - the gradient behind a 2x2 / stride-2
  // max-pooling - so any four consecutive pixels of a row hold at most two: the 2:4 structured-sparse MFMA (v_smfmac_f32_32x32x32_bf16) takes dout as its
  // sparse operand, half the matrix instructions for the same products (sf_conv3x3_bwd_weight_folded_sparse24)
  // 2: the POOLED form of the same launch - the sparse operand is built from the pooled gradient g [n'][H/2][W/2][pool_s] (bf16; image n' = the pooling's
  // outer permutation of n: (l T + t) B + b -> (t L + l) B + b, pool_L == 0: identity) and the pooling's routing record [n][H/2][W/2][dc/8] (2 bits per
  // channel: 2 dy + dx of the window element that took the maximum) instead of from dout, which this kernel then does not read: 4.5 KB per K tile, not 16
  int sparse24;
  const void* pool_g; const unsigned short* pool_route; int pool_s, pool_L, pool_T, pool_B;
  // SF_F32E kernel (conv3x3_wgrad_f32e.hip): device word with max |dout| (sfTensor::amax) - dout is scaled by the power of two that puts it at 2^14 before
  // the fp16 split, the accumulators by the inverse; null = as is
  const float* amax_dout;
};


struct Plan { int tiles_x, tiles_y, ntiles, KS, cot, cit; size_t ws_floats; int tpg = 0, maxseg = 1; int wide_pairs = 0, units = 0, edge_mode = 0; int tr = 4; };

// kt_h: K-tile height of the kernel variant (4 rows fp32, 8 rows bf16)
inline Plan make_plan(int Np, int Kp, int n, int h, int w, int kt_h) {
  Plan pl;
  pl.tiles_x = (w + KT_W - 1) / KT_W;
  pl.tiles_y = (h + kt_h - 1) / kt_h;
  pl.ntiles = pl.tiles_x * pl.tiles_y * n;
  pl.cot = (Np + CO_T - 1) / CO_T;
  pl.cit = (Kp + CI_T - 1) / CI_T;
  int want = 512 / (pl.cot * pl.cit);  // ~2 workgroups per CU in total: fewer, longer K slices than 1024 measured faster (less split-K reduce)
  // at least 8 slices while that keeps the launch under ~2048 workgroups; a weight of thousands of slabs (the generator's half-resolution ConvGRU
  // convolutions: 2048 x 4096 channels = 2048 slabs of 147 KB) fills the chip unsplit - 8 partial copies of a 302 MB gradient were 2.4 GB written and re-read
  int floor_ks = 2048 / (pl.cot * pl.cit);
  floor_ks = floor_ks < 1 ? 1 : floor_ks > 8 ? 8 : floor_ks;
  if (want < floor_ks) want = floor_ks;
  if (want > 256) want = 256;
  pl.KS = pl.ntiles < want ? pl.ntiles : want;
  if (pl.KS < 1) pl.KS = 1;
  pl.ws_floats = (size_t)pl.KS * ((size_t)9 * pl.cot * CO_T * pl.cit * CI_T + (size_t)pl.cot * CO_T);
  return pl;
}

}  // namespace sfwgrad

// all-bf16-storage variant (conv3x3_wgrad_bf16_dma.hip): LDS-DMA tiles + transposing LDS reads; its own plan (4x16 K tiles,
// 128 x 64 slabs, one workgroup per CU); the launcher places [zero page | partial | partial_db] in the workspace and sets p's
// plan fields
// groups > 0 (folded BatchNorm): n splits into `groups` runs of whole images; a slice stores one partial slab per group it touches
// single_source: the launch reads ONE input tensor (the folded path always; sf_conv3x3_bwd_weight when src1 is empty): wide slabs allowed
sfwgrad::Plan sf_wgrad_bf16_dma_plan(int Np, int Kp, int n, int h, int w, int groups = 0, int single_source = 0, int tr = 4);
int sf_launch_wgrad_bf16_dma(sfwgrad::WgradParams& p, const sfwgrad::Plan& pl, float* workspace, hipStream_t st);
// bf16-MFMA variant (conv3x3_wgrad_bf16.hip): fills the same partial slabs
int sf_launch_wgrad_bf16(const sfwgrad::WgradParams& p, const sfwgrad::Plan& pl, hipStream_t st);
int sf_launch_wgrad_f16(const sfwgrad::WgradParams& p, const sfwgrad::Plan& pl, hipStream_t st);  // fp16 operands (conv3x3_wgrad_f16.hip), fp32-stored tensors
int sf_launch_wgrad_f32e(const sfwgrad::WgradParams& p, const sfwgrad::Plan& pl, hipStream_t st);  // SF_F32E: three fp16 products per fp32 product (conv3x3_wgrad_f32e.hip)
