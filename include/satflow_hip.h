/*
 * satflow_hip.h -- C ABI of libsatflow_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the spatiotemporal hot path of openclimatefix/satflow
 * (SURVEY.md section 8).  The reference has no FFI: its boundary is a Python
 * nn.Module surface, every "kernel" being a chain of ATen ops.  Each entry
 * point below names the reference code (file:line under /root/reference) whose
 * arithmetic it replaces; the Python host in satflow_amd/ mirrors the module
 * surface and calls these through ctypes.
 *
 * Conventions
 *  - Plain C symbols, no global state except a thread-local error string.
 *  - Every call takes a hipStream_t (as void*) and only enqueues work on it;
 *    no allocation, no synchronisation, safe under hipGraph capture.
 *  - All pointers are DEVICE pointers owned by the caller.
 *  - Activations are NHWC ("pixel-major"): element (n, y, x, c) of a tensor
 *    lives at base[((n*H + y)*W + x)*stride + c].  `stride` (elements per
 *    pixel) may exceed the channel count, so a channel slice of a wider
 *    tensor is a valid tensor.  Channel counts seen by the convolution
 *    kernels are padded to a multiple of SF_CPAD with zero-filled pad lanes.
 *  - Return value: 0 = ok, nonzero = error (see sf_last_error_string()).
 */
#ifndef SATFLOW_HIP_H
#define SATFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped on EVERY signature or workspace-layout change (2: workspace arguments of sf_convgru_seq_*, sticky error word; 3: SF_F16, sf_bmm_f16; 4: sf_flash_attention_*; 5: sf_space_to_depth2, sf_regroup5x5_s2d_*, sf_conv5x5_*, sf_linear_fwd with 16-bit operands; 8: SF_F32E, sfTensor.amax, sf_amax) */
#define SF_ABI_VERSION 8
#define SF_CPAD 16 /* channel padding granule of NHWC activations */

typedef void* sfStream; /* hipStream_t */

/* Arithmetic/storage type of a kernel family. */
enum { SF_F32 = 0, SF_BF16 = 1, SF_F16 = 2, SF_F32E = 3 };
/* SF_F32E (ABI 8): fp32-EQUIVALENT products on the fp16 matrix pipe (gfx950 has no TF32; v_mfma_f32_32x32x16_f16 is 16x the exact-fp32 MFMA rate).
 * Every fp32 operand is split as it is staged, x = hi + 2^-11 lo' with hi = fp16(x), lo' = fp16((x - hi) * 2^11) - 22 mantissa bits - and the K loop runs
 * three times over the operands: first [hi(x) * lo'(w) + lo'(x) * hi(w)], then the accumulators are scaled by 2^-11 (exact), then hi(x) * hi(w); fp32
 * accumulation, fp32-stored tensors, fp32 epilogues - the arithmetic behind the fp32 parity gate (rtol 1e-4 / atol 1e-5, BASELINE.json north_star) at
 * three fp16 products per fp32 product (profiles/r06_f32e_numerics.txt: every reference golden passes the unchanged gates with >= 5x margin; the same
 * split in bf16 does not).  fp16's range is handled per tensor: forward activations and weights are taken as they are (|x| >= 65520 overflows to inf ->
 * NaN results, loudly), a GRADIENT operand carries sfTensor.amax and is scaled by the power of two that puts its largest magnitude at 2^14 (undone, exactly,
 * on the accumulators).  Accepted by sf_conv3x3_pack_weights (the packed image then has 3 * Kp / 16 chunks: sf_conv3x3_packed_elems(Np, 3 * Kp) halves),
 * sf_conv3x3_fwd (linear / sigmoid), sf_convlstm_cell_fwd, sf_convgru_step_fwd and sf_conv3x3_bwd_weight; every other entry refuses it (the host runs the
 * exact-fp32 kernels there). */
/* SF_F16 (round 4): fp16 MFMA operands (v_mfma_f32_32x32x16_f16), fp32 accumulate, fp32-STORED tensors - the `precision: 16` of the reference's
 * configs/trainer/half.yaml:33 (BASELINE configs[4]).  Accepted by sf_conv3x3_pack_weights, sf_conv3x3_fwd (linear / sigmoid epilogue),
 * sf_conv3x3_fwd_splitk(+ _workspace_bytes) and sf_conv3x3_bwd_weight; the attention products have sf_bmm_f16.  Every other entry refuses it. */

/* Epilogues of sf_conv3x3_fwd. */
enum {
  SF_EPI_LINEAR = 0,  /* out = conv + bias                                  */
  SF_EPI_SIGMOID = 1, /* out = sigmoid(conv + bias)                         */
};

int sf_abi_version(void);
const char* sf_last_error_string(void);

/* A strided NHWC channel slice.  Convolution INPUTS may additionally remap the image index:
 * the kernel's image n reads image (n / idiv) % imod of this tensor (idiv <= 1: no division,
 * imod == 0: no modulo).  That is how one tensor is broadcast over the lead-time axis of MetNet
 * (frames shared by all lead times: imod = frames; one-hot lead-time planes: idiv = frames)
 * without materialising the copies ConditionTime makes (satflow/models/layers/ConditionTime.py:22-33). */
typedef struct {
  void* ptr;      /* first channel of pixel (0,0,0)            */
  int32_t c;      /* channels (padded; multiple of SF_CPAD for conv inputs) */
  int32_t stride; /* elements per pixel                        */
  int32_t idiv;   /* image-index divisor (0/1 = none)          */
  int32_t imod;   /* image-index modulus (0 = none)            */
  int32_t dtype;  /* element storage: SF_F32 (default) or SF_BF16.  bf16 storage is accepted by the MetNet encoder
                     kernels (sf_metnet_preprocess_fwd, sf_conv3x3_fwd[_stats] / _bwd_weight with the SF_BF16 kernels,
                     sf_leadtime_pool_*, sf_batchnorm_*, sf_maxpool2_*), for the ConvLSTM's saved gates / gate
                     gradients (`gates` of sf_convlstm_cell_fwd, `gates` and `dz` of sf_convlstm_cell_bwd_gates, `dout`
                     of sf_conv3x3_bwd_weight against fp32 sources; `gates` of sf_convgru_step_fwd, `gates`, `dgx`,
                     `dgh` of sf_convgru_bwd_gates) and for the ConvLSTM's layer inputs / hidden states
                     (`x`, `h_prev`, `h_out` of sf_convlstm_cell_fwd with the SF_BF16 kernel); everything else
                     requires SF_F32 */
  const float* amax; /* (ABI 8; SF_F32E kernels only, nullable) device word written by sf_amax: max |element| of this tensor.  A convolution source
                     (`src0` of sf_conv3x3_fwd: the output gradient of an input-gradient launch) or `dout` of sf_conv3x3_bwd_weight that carries it is
                     multiplied by 2^(14 - floor(log2 amax)) before it is split into fp16 parts, and the accumulators by the inverse afterwards. */
} sfTensor;

/* (ABI 8) amax[0] = max |t| over `pixels` pixels x t.c channels of an fp32-stored tensor (0 for an empty one), for sfTensor.amax.  Two launches on the
 * stream (reset word, reduce with one atomic per workgroup: the maximum is order-independent, so the result is deterministic).  amax_acc (nullable): a
 * second word that takes the maximum of ITS current value and this tensor's - the running maximum over the per-step gate gradients of a recurrent cell,
 * whose weight gradient reads all steps in one launch; reset_acc != 0 zeroes it first. */
int sf_amax(sfTensor t, int64_t pixels, float* amax, float* amax_acc, int32_t reset_acc, sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * Weight repack (derived cache; the saved form stays the reference's OIHW fp32 state_dict).
 *
 * w: [O][I][3][3] fp32 (reference nn.Conv2d layout, e.g. ConvLSTMCell.conv.weight,
 *    satflow/models/layers/ConvLSTM.py:34-40).
 * Packed GEMM-B image: [Np/nb][Kp/16][9 taps][nb][16] with
 *    transpose == 0:  B(n, k, tap) = w[nmap[n]][kmap[k]][tap]          (forward)
 *    transpose == 1:  B(n, k, tap) = w[kmap[k]][nmap[n]][8 - tap]      (backward-data: W^T, flipped)
 * nmap[Np], kmap[Kp]: device int32 tables, -1 marks a zero pad lane; nb = 32*nf.
 * bias (nullable) [O] -> bias_packed[Np] gathered through nmap (transpose == 0 only).
 * ------------------------------------------------------------------------------------------- */
size_t sf_conv3x3_packed_elems(int32_t Np, int32_t Kp);
int sf_conv3x3_pack_weights(const float* w, int32_t O, int32_t I, const int32_t* nmap, int32_t Np,
                            const int32_t* kmap, int32_t Kp, int32_t nf, int32_t transpose,
                            void* packed, const float* bias, float* bias_packed, int32_t dtype,
                            sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3 "same" convolution, implicit GEMM on MFMA, K = 9*(src0.c + src1.c) over [src0 ; src1].
 * Replaces nn.Conv2d(k=3, padding=1) call sites on the path: the Conv3d(1,3,3)+Sigmoid head
 * (satflow/models/conv_lstm.py:164-169,200-201), the DownSampler convolutions of metnet.MetNet
 * (call site satflow/models/pl_metnet.py:46-59) and, with a transpose==1 weight image, the
 * input-gradient of every convolution on the path.
 * out.c output channels are written (pad lanes included); Np >= out.c.  The SF_BF16 kernel stores 16-byte
 * channel groups per pixel: out.ptr and bias_packed 16-byte aligned, out.stride a multiple of 4 (fp32) / 8 (bf16
 * storage), out.c a multiple of 8.
 * ------------------------------------------------------------------------------------------- */
int sf_conv3x3_fwd(sfTensor src0, sfTensor src1, int32_t n, int32_t h, int32_t w,
                   const void* wpacked, const float* bias_packed, int32_t Np, int32_t nf,
                   int32_t epilogue, sfTensor out, int32_t dtype, sfStream stream);
/* The same convolution (SF_BF16 kernels, linear epilogue, one source, nf = 4) for FEW SMALL IMAGES WITH MANY INPUT CHANNELS - the state convolutions
 * of the DGMR generator's ConvGRU (reference layers/Generator.py:91-117 calling the ConvGRU cell frame by frame: 2 images of 16x16 .. 64x64 pixels,
 * up to 2048 -> 1024 channels), the deepest discriminator blocks (layers/Discriminator.py:169-226 at 8x8 / 4x4).  sf_conv3x3_fwd would run them on a few
 * dozen workgroups, each streaming megabytes of weights through its LDS; here the input channels are cut into slices (grid z), every slice writes its
 * fp32 partial sums to `workspace`, a second kernel adds the slices in a fixed order (deterministic) and the bias.
 * sf_conv3x3_fwd_splitk_workspace_bytes returns 0 for shapes that are not worth splitting (Kp = source channel lanes): call sf_conv3x3_fwd then. */
size_t sf_conv3x3_fwd_splitk_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t Np, int32_t nf, int32_t Kp, int32_t dtype);
int sf_conv3x3_fwd_splitk(sfTensor src, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_packed, int32_t Np, int32_t nf,
                          sfTensor out, void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream);
/* The same convolution (SF_BF16 or SF_F32E kernels, linear epilogue) that also emits what the BatchNorm2d behind it in the DownSampler
 * needs: per pixel tile, the sum and the sum of squares of every output channel's STORED values,
 * stats[(image * sf_conv3x3_stats_tiles(h, w) + tile)][Np][2] fp32 - consumed by sf_batchnorm_train_fwd_stats, which then
 * does not have to read the activation a first time. */
int32_t sf_conv3x3_stats_tiles(int32_t h, int32_t w);
int sf_conv3x3_fwd_stats(sfTensor src0, sfTensor src1, int32_t n, int32_t h, int32_t w,
                         const void* wpacked, const float* bias_packed, int32_t Np, int32_t nf,
                         sfTensor out, float* stats, int32_t dtype, sfStream stream);

/* The DownSampler's BatchNorm2d -> Conv2d pairs (metnet.DownSampler: BatchNorm2d(160) -> Conv2d(160, 256) and twice
 * BatchNorm2d(256) -> Conv2d(256, 256); call site satflow/models/pl_metnet.py:46-59) with the normalisation FOLDED into the
 * convolution (training mode, SF_BF16 kernels): conv(a_g * x + b_g) = conv_{W * a_g}(x) + T_g[border class], so the normalised
 * tensor is never written or read.
 *   sf_conv3x3_fold_pack : from w [O][I][3][3], bias (nullable) and the BatchNorm's per-group scale / shift [groups][Kp] (as written
 *       by sf_batchnorm_train_fwd[_stats] with y.ptr == NULL) -> `groups` packed weight images (sf_conv3x3_packed_elems each,
 *       What = w[n][k][tap] * scale[g][k] rounded to bf16) and bias_tab [groups][9][Np] fp32: the bias plus the shift's contribution
 *       What * (shift / scale) through the taps that lie inside the image for an output pixel of border class
 *       3 * (top, interior, bottom) + (left, interior, right) - the same rounded weights as x sees, so that the mean part of x and
 *       the shift cancel exactly as they do in the unfolded form
 *   sf_conv3x3_fwd_folded : image i uses weight image i / (n / groups); stats nullable (as sf_conv3x3_fwd_stats).  h, w >= 2.
 * The matching weight gradient is sf_conv3x3_bwd_weight_folded; the input gradient is the plain one (unscaled W) followed by
 * sf_batchnorm_train_bwd. */
int sf_conv3x3_fold_pack(const float* w, int32_t O, int32_t I, const int32_t* nmap, int32_t Np, const int32_t* kmap, int32_t Kp,
                         int32_t nf, const float* bias, const float* scale, const float* shift, int32_t groups, void* packed,
                         float* bias_tab, int32_t dtype, sfStream stream);
int sf_conv3x3_fwd_folded(sfTensor src, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_tab, int32_t Np,
                          int32_t nf, int32_t groups, sfTensor out, float* stats, int32_t dtype, sfStream stream);
/* (ABI 6) The same folded convolution WITH THE MaxPool2d((2, 2), stride 2) BEHIND IT IN ITS EPILOGUE: the DownSampler's conv4 -> pooling pair
 * (upstream metnet DownSampler, reached from satflow/models/pl_metnet.py:46-59).  The convolution's own output is never written: `pooled`
 * [n][h/2][w/2][c] bf16 receives max over each 2x2 window of the bf16-ROUNDED convolution results (= sf_maxpool2_route_fwd of
 * sf_conv3x3_fwd_folded's output, bit for bit), stored as image perm(i) (perm_l, perm_t as sf_maxpool2_fwd), and `route` [n][h/2][w/2][c/8] uint16
 * (8-byte aligned) the routing record of sf_maxpool2_route_fwd (2 bits per channel: first maximum in row-major window order), so that
 * sf_maxpool2_route_bwd is the backward of the pooling.  No dropout here (sf_dropout2_bf16 on `pooled` applies MetNet's two masks).
 * One-wave-per-SIMD kernel only: even h, w, nf = 4, c a multiple of 64, >= 48 input channels, >= 1024 tiles -
 * sf_conv3x3_fwd_folded_pool_supported says whether a shape is taken (1) or the two separate calls must be used (0). */
int32_t sf_conv3x3_fwd_folded_pool_supported(int32_t n, int32_t h, int32_t w, int32_t Np, int32_t nf, int32_t cin, int32_t cout, int32_t groups);
int sf_conv3x3_fwd_folded_pool(sfTensor src, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_tab, int32_t Np, int32_t nf,
                               int32_t groups, sfTensor pooled, int32_t perm_l, int32_t perm_t, void* route, int32_t dtype, sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * Fused ConvLSTM cell step.  Replaces ConvLSTMCell.forward,
 * satflow/models/layers/ConvLSTM.py:42-57 (cat -> conv -> split i,f,o,g -> sigmoid x3, tanh ->
 * c' = f*c + i*g -> h' = o*tanh(c')), one launch instead of ~11 ATen kernels.
 *   x       : layer input (src0), h_prev : previous hidden state (src1; ptr NULL == zeros,
 *             i.e. init_hidden, layers/ConvLSTM.py:59-64, without materialising them)
 *   c_prev  : previous cell state (ptr NULL == zeros);  h_out, c_out : new states (cell states fp32; with the
 *             SF_BF16 kernel x / h_prev / h_out may each be SF_BF16-stored: a hidden state is only ever read
 *             as a bf16 MFMA operand, so storing the rounded value changes no result).  SF_BF16 kernel:
 *             every state / gate tensor needs 16-byte aligned pixels (pointer and stride)
 *   gates   : nullable; receives post-activation i,f,o,g as [.., 4*hidp] (gate-major) for backward
 *             (SF_F32 or SF_BF16 storage: backward-only data, the outputs do not depend on it)
 *   wpacked : image from sf_conv3x3_pack_weights with the LSTM nmap (32 hidden channels x 4 gates
 *             per N-block, nf == 4), bias_packed likewise.  hidp = padded hidden channels.
 * ------------------------------------------------------------------------------------------- */
int sf_convlstm_cell_fwd(sfTensor x, sfTensor h_prev, sfTensor c_prev, int32_t n, int32_t h,
                         int32_t w, const void* wpacked, const float* bias_packed, int32_t hidp,
                         sfTensor h_out, sfTensor c_out, sfTensor gates, int32_t dtype,
                         sfStream stream);

/* Pointwise part of the cell's backward (autograd of layers/ConvLSTM.py:48-55):
 *   dh = sum of up to three incoming hidden-state gradients (NULL ptr = absent; each SF_F32 or SF_BF16: the input-gradient convolutions'
 *        output is stored as bf16 in "bf16a" mode, as a 16-bit autocast leaves it; summed in fp32)
 *   dc_next (nullable) gradient flowing into c'; gates = saved i,f,o,g; c_prev nullable (zeros)
 *   -> dz [.., 4*hidp] gradient wrt the pre-activation conv output, dc_prev (nullable out).
 *   gates and dz share one storage type (SF_F32 or SF_BF16; dz may overwrite gates in place).
 *   dz.amax (ABI 8, nullable; fp32-stored dz): a device word this launch RAISES to max |dz| (atomic maximum on the bit pattern; the caller sets it to 0
 *   beforehand) - the scale word the SF_F32E input-gradient / weight-gradient kernels want with dz (sfTensor.amax), without sf_amax's extra pass.
 *   c_new (ABI 7: nullable): the step's new cell state; NULL = taken again as f * c_prev + i * g from the saved gates (one read of the state
 *   sequence less; with bf16-stored gates it then carries their rounding like the rest of this pass). */
int sf_convlstm_cell_bwd_gates(sfTensor dh0, sfTensor dh1, sfTensor dh2, sfTensor dc_next,
                               sfTensor gates, sfTensor c_prev, sfTensor c_new, int64_t pixels,
                               int32_t hidp, sfTensor dz, sfTensor dc_prev, int32_t dtype,
                               sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * Weight/bias gradient of a 3x3 same convolution (autograd of the call sites above):
 *   dW[o][i][ky][kx] (+)= sum_{n,y,x} dout[n,y,x,o] * in[n,y+ky-1,x+kx-1,i],   db[o] (+)= sum dout
 * in = [src0 ; src1] (padded channels), dout.c = padded output channels.  nmap/kmap translate the
 * padded indices to rows/columns of the OIHW gradient ( -1 = skip ).  `accumulate` != 0 adds into
 * dw/db.  workspace: sf_conv3x3_bwd_weight_workspace_bytes() bytes of scratch.
 * Storage (SF_BF16 kernels): all fp32; all bf16; or bf16 dout with fp32 sources.  src0 and src1 share one type.
 * ------------------------------------------------------------------------------------------- */
size_t sf_conv3x3_bwd_weight_workspace_bytes(int32_t Np, int32_t Kp, int32_t n, int32_t h, int32_t w);
int sf_conv3x3_bwd_weight(sfTensor src0, sfTensor src1, sfTensor dout, int32_t n, int32_t h,
                          int32_t w, const int32_t* nmap, const int32_t* kmap, int32_t O, int32_t I,
                          float* dw, float* db, int32_t accumulate, void* workspace,
                          size_t workspace_bytes, int32_t dtype, sfStream stream);
/* Weight gradient of a convolution whose input BatchNorm was folded (sf_conv3x3_fwd_folded): `src` is the tensor IN FRONT of the
 * normalisation (bf16-stored), scale / shift [groups][src.c] the per-group affine map; with in = scale_g * src + shift_g inside the
 * image (zero padding outside)
 *   dW = sum_g scale_g[i] * dWraw_g[o][i][tap] + shift_g[i] * V_g[tap][o],   V_g = sum of dout over the group's pixels whose tap
 * neighbour lies inside the image; db as above.  SF_BF16 kernels, bf16-stored src and dout, h, w >= 2, no image remap.
 * bn_sums (nullable; then weight = the convolution's OIHW weights, mean / rstd [groups][src.c] of the BatchNorm): also the two
 * reductions of that BatchNorm's backward, [groups][2][src.c] doubles = (sum dn, sum dn * xhat) with dn = conv^T(dout, W) the
 * gradient entering the BatchNorm - formed from V_g and dWraw_g (sum dn = W . V_g, sum dn * x = W . dWraw_g), i.e. without dn:
 * sf_batchnorm_train_bwd_coef turns them into the coefficients sf_conv3x3_bwd_data_bn applies. */
size_t sf_conv3x3_bwd_weight_folded_workspace_bytes(int32_t Np, int32_t Kp, int32_t n, int32_t h, int32_t w, int32_t groups);
int sf_conv3x3_bwd_weight_folded(sfTensor src, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap,
                                 const int32_t* kmap, int32_t O, int32_t I, const float* scale, const float* shift,
                                 int32_t groups, float* dw, float* db, int32_t accumulate, const float* weight,
                                 const float* mean, const float* rstd, double* bn_sums, void* workspace,
                                 size_t workspace_bytes, int32_t dtype, sfStream stream);
/* (ABI 6) The same weight gradient for a dout that is the gradient BEHIND A 2x2 / stride-2 MAX-POOLING (sf_maxpool2_route_bwd's output: the DownSampler's conv4,
 * upstream metnet DownSampler via satflow/models/pl_metnet.py:46-59): one non-zero per pooling window and channel, hence at most two in any four consecutive
 * pixels of an image row - dout goes in as the sparse operand of the 2:4 structured-sparse matrix instruction (v_smfmac_f32_32x32x32_bf16): half the matrix
 * instructions for exactly the same products.  THE CALLER GUARANTEES THE STRUCTURE (at most one non-zero per aligned horizontal pixel pair and channel; the
 * entry point cannot check 1.2 GB cheaply): a dout that breaks it is multiplied wrongly.  sf_conv3x3_bwd_weight_folded_sparse24_supported (Np = dout lanes,
 * Kp = src lanes): 1 if the shape takes the path (even h, w; regular 128 x 64 slabs), else call sf_conv3x3_bwd_weight_folded.  Same workspace. */
int32_t sf_conv3x3_bwd_weight_folded_sparse24_supported(int32_t Np, int32_t Kp, int32_t n, int32_t h, int32_t w, int32_t groups);
/* (ABI 7) pooled_dout / route / perm_l / perm_t: optionally the INPUTS of the sf_maxpool2_route_bwd call that produced dout - the pooled gradient (bf16,
 * dout's lanes, [n][h/2][w/2] in the pooling's output image order), the routing record and the pooling's outer permutation.  With them (and whole 4 x 16
 * pixel tiles, lanes in whole 128-channel tiles) the matrix kernel builds its sparse operand from 4.5 KB of pooled data per K tile instead of reading 16 KB
 * of dout, which then only the small border-sum helper reads.  pooled_dout.ptr NULL: from dout.  Same results either way. */
int sf_conv3x3_bwd_weight_folded_sparse24(sfTensor src, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap,
                                 const int32_t* kmap, int32_t O, int32_t I, const float* scale, const float* shift,
                                 int32_t groups, float* dw, float* db, int32_t accumulate, const float* weight,
                                 const float* mean, const float* rstd, double* bn_sums, sfTensor pooled_dout, const void* route,
                                 int32_t perm_l, int32_t perm_t, void* workspace,
                                 size_t workspace_bytes, int32_t dtype, sfStream stream);
/* Input gradient of the same convolution THROUGH the folded BatchNorm in one launch: dx = A_g * conv^T(dout, W) + B_g * x + K_g
 * with coef [groups][3][x.c] = (A, B, K) from sf_batchnorm_train_bwd_coef (the affine form of the training-mode BatchNorm
 * backward) applied in the convolution's epilogue; wpacked = the transposed weight image (sf_conv3x3_pack_weights, transpose 1).
 * Replaces sf_conv3x3_fwd on dout + sf_batchnorm_train_bwd: the gradient entering the BatchNorm is never written or read.
 * SF_BF16 kernels; dout, x, dx bf16-stored. */
int sf_conv3x3_bwd_data_bn(sfTensor dout, int32_t n, int32_t h, int32_t w, const void* wpacked, int32_t Np, int32_t nf,
                           sfTensor x, const float* coef, int32_t groups, sfTensor dx, int32_t dtype, sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * Layout conversion at the module boundary (the reference keeps NCHW-style tensors throughout:
 * x[B,T,C,H,W] in, [B,C,T,H,W] out, satflow/models/conv_lstm.py:205-228,198-201).
 * The NCHW-side tensor is addressed as  base[b*stride_b + t*stride_t + c*stride_c + y*w + x]
 * for b < nb, t < nt; the NHWC side holds nb*nt images, image index j = t*nb + b (time-major).
 *   nchw_to_nhwc: gathers c real channels, zero-fills the pad lanes dst.c - c; dst storage SF_F32 or SF_BF16 (rounded to nearest even)
 *   nhwc_to_nchw: scatters the first c channels back.
 * ------------------------------------------------------------------------------------------- */
int sf_nchw_to_nhwc(const float* src, int64_t stride_b, int64_t stride_t, int64_t stride_c, int32_t nb,
                    int32_t nt, int32_t c, int32_t h, int32_t w, sfTensor dst, int32_t dtype,
                    sfStream stream);
int sf_nhwc_to_nchw(sfTensor src, int32_t nb, int32_t nt, int32_t c, int32_t h, int32_t w, float* dst,
                    int64_t stride_b, int64_t stride_t, int64_t stride_c, int32_t dtype,
                    sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * Fused Adam step on flat fp32 buffers (p, g, m, v of n elements).  Arithmetic of torch.optim.Adam
 * (no amsgrad / weight decay) as configured by the reference (satflow/models/conv_lstm.py:48-51,
 * pl_metnet.py:70): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  g is multiplied by grad_scale first
 * (1/world_size after a SUM all-reduce).  step counts from 1.
 * ------------------------------------------------------------------------------------------- */
int sf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int32_t step, float grad_scale, sfStream stream);

/* =============================================================================================
 * MetNet stack.  The arithmetic lives in the un-vendored packages metnet>=0.0.3 / axial_attention
 * (reference requirements.txt:18); the reference's call site is satflow/models/pl_metnet.py:46-59,65
 * (LitMetNet -> metnet.MetNet).  Each entry names the upstream module it replaces.
 * ============================================================================================= */

/* MetNetPreprocessor(sat_channels, crop_size, use_space2depth=True, split_input=True):
 * imgs[B][T][C][H][W] (H = W = 4*crop) -> frames j = t*B + b of [crop][crop][out.c]:
 * lanes [0,4*sat) PixelUnshuffle(2) (lane c*4+dh*2+dw) centre-cropped, [4*sat,8*sat) its 2x2 mean,
 * then the C-sat other channels (2x2 mean of the raw image, centre-cropped); pad lanes zero. */
int sf_metnet_preprocess_fwd(const float* imgs, int32_t B, int32_t T, int32_t C, int32_t sat,
                             int32_t H, int32_t W, int32_t crop, sfTensor out, int32_t dtype,
                             sfStream stream);

/* Gradient of the above wrt the raw images (what autograd produces through the reference's preprocessor when imgs
 * requires grad: saliency maps, adversarial inputs): dimgs[B][T][C][H][W] fp32, written completely. */
int sf_metnet_preprocess_bwd(sfTensor dout, int32_t B, int32_t T, int32_t C, int32_t sat, int32_t H, int32_t W,
                             int32_t crop, float* dimgs, int32_t dtype, sfStream stream);

/* nn.MaxPool2d(2, stride 2) of the DownSampler, forward / backward (argmax recomputed from `in`).
 * perm_l > 0: images are reordered on the pooled side, input image (l*perm_t + t)*B + b <->
 * pooled image (t*perm_l + l)*B + b (lead-time-major encoder order -> time-major ConvGRU order). 
 * din.amax (the three backward entries: _bwd, _dropout_bwd, _route_bwd; nullable, fp32-stored din only): a word the CALLER has zeroed, raised to max |din| - the SF_F32E convolutions' scale
 * word of din (sfTensor::amax) without a separate sf_amax pass. */
int sf_maxpool2_fwd(sfTensor in, int64_t n, int32_t h, int32_t w, sfTensor out, int32_t perm_l,
                    int32_t perm_t, int32_t dtype, sfStream stream);
int sf_maxpool2_bwd(sfTensor in, sfTensor dout, int64_t n, int32_t h, int32_t w, sfTensor din,
                    int32_t perm_l, int32_t perm_t, int32_t dtype, sfStream stream);
/* The encoder's LAST max-pool fused with the dropouts that follow it in MetNet (nn.Dropout(temporal_dropout),
 * pl_metnet.py:58, and the ConvGRU's sequence-consistent input dropout): out = sf_dropout2(maxpool(in)) with the masks
 * of sf_dropout2 for the same (p1, p2, period, seeds), indexed by the flat element index of the contiguous pooled tensor;
 * the backward masks the incoming gradient the same way before routing it.  `period` must be a whole number of pooled images. */
int sf_maxpool2_dropout_fwd(sfTensor in, int64_t n, int32_t h, int32_t w, sfTensor out, int32_t perm_l, int32_t perm_t,
                            float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2, int32_t dtype,
                            sfStream stream);
int sf_maxpool2_dropout_bwd(sfTensor in, sfTensor dout, int64_t n, int32_t h, int32_t w, sfTensor din, int32_t perm_l,
                            int32_t perm_t, float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2,
                            int32_t dtype, sfStream stream);
/* The same pooling (p1 = p2 = 0: without the dropouts) that RECORDS which window element every channel took - route
 * [n][h/2][w/2][c/8] uint16, 2 bits per channel, first maximum in row-major window order as max_pool2d - so that the backward
 * pass neither keeps nor re-reads the input tensor (the DownSampler's last pooling: 1.2 GB at the benchmark size). */
int sf_maxpool2_route_fwd(sfTensor in, int64_t n, int32_t h, int32_t w, sfTensor out, int32_t perm_l, int32_t perm_t, float p1,
                          float p2, int64_t period, uint64_t seed1, uint64_t seed2, void* route, int32_t dtype, sfStream stream);
/* (ABI 7) masked_dout (nullable): also stores the pooled gradient AFTER the dropout masks, laid out like dout - the operand source of
 * sf_conv3x3_bwd_weight_folded_sparse24's pooled form. */
int sf_maxpool2_route_bwd(const void* route, sfTensor dout, int64_t n, int32_t h, int32_t w, sfTensor din, int32_t perm_l,
                          int32_t perm_t, float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2, void* masked_dout,
                          int32_t dtype, sfStream stream);

/* nn.BatchNorm2d of the DownSampler.  Training mode: `groups` independent batches of
 * pix_per_group pixels each (one per lead time: the reference calls the encoder once per lead
 * time, so statistics are per call), biased variance, running stats updated group after group with
 * `momentum` (unbiased variance; momentum < 0 selects torch's cumulative average of momentum=None with
 * -momentum - 1 = batches tracked before this call).  Scratch: sums [groups][2][C] doubles; mean/rstd/scale/shift
 * [groups][C] floats (mean, rstd are what the backward needs).  creal = real channels of gamma/beta.
 * y.ptr == NULL: statistics, scale / shift and the running-stat update only (the apply is folded into the next convolution). */
int sf_batchnorm_train_fwd(sfTensor x, int64_t pix_per_group, int32_t groups, int32_t creal,
                           const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* rstd,
                           float* scale, float* shift, double* sums, sfTensor y, int32_t dtype,
                           sfStream stream);
/* sf_batchnorm_train_fwd with the statistics taken from the producing convolution's per-tile records
 * (sf_conv3x3_fwd_stats): group g owns tiles [g*tiles_per_group, (g+1)*tiles_per_group), stats_np = the convolution's Np. */
int sf_batchnorm_train_fwd_stats(sfTensor x, int64_t pix_per_group, int32_t groups, int32_t creal, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                 float* mean, float* rstd, float* scale, float* shift, double* sums, const float* stats,
                                 int32_t tiles_per_group, int32_t stats_np, sfTensor y, int32_t dtype, sfStream stream);
int sf_batchnorm_eval_fwd(sfTensor x, int64_t pixels, int32_t creal, const float* gamma,
                          const float* beta, float eps, const float* running_mean,
                          const float* running_var, float* scale, float* shift, sfTensor y,
                          int32_t dtype, sfStream stream);
/* Backward of the eval-mode BatchNorm (frozen running statistics: fine-tuning / saliency with model.eval()):
 * dx = gamma*rstd*dy, dgamma = sum dy*(x-rm)*rstd, dbeta = sum dy.  scratch: 5*x.c floats; sums: 2*x.c doubles. */
int sf_batchnorm_eval_bwd(sfTensor x, sfTensor dy, int64_t pixels, int32_t creal, const float* gamma, float eps,
                          const float* running_mean, const float* running_var, double* sums, float* scratch,
                          sfTensor dx, float* dgamma, float* dbeta, int32_t dtype, sfStream stream);
/* dx.amax (both backward entries; nullable, fp32-stored dx only): a word the CALLER has zeroed, raised to max |dx| by the apply pass - the SF_F32E
 * convolutions' scale word of dx (sfTensor::amax) without a separate sf_amax pass over it. */
int sf_batchnorm_train_bwd(sfTensor x, sfTensor dy, int64_t pix_per_group, int32_t groups,
                           int32_t creal, const float* gamma, const float* mean, const float* rstd,
                           double* sums, float* coef /* scratch [groups][3][C] */, sfTensor dx,
                           float* dgamma, float* dbeta, int32_t dtype, sfStream stream);
/* Coefficients (A, B, K) [groups][3][c] of dx = A * dy + B * x + K and dgamma / dbeta from PRECOMPUTED reductions
 * sums [groups][2][c] doubles = (sum dy, sum dy * xhat): the tail of sf_batchnorm_train_bwd without its two passes over dy. */
int sf_batchnorm_train_bwd_coef(const double* sums, int64_t pix_per_group, int32_t groups, int32_t c, int32_t creal,
                                const float* gamma, const float* mean, const float* rstd, float* coef, float* dgamma,
                                float* dbeta, int32_t dtype, sfStream stream);

/* nn.LayerNorm([C', W, W]) behind the cell's convolutions (layer_norm=True,
 * satflow/models/layers/SpatioTemporalLSTMCell_memory_decoupling.py:20-62): per sample over all channels and pixels, affine
 * parameters gamma / beta in the reference's [C'][pixels] layout.  Activations NHWC with gate-major padded lanes: lane g*hidp + j is
 * channel g*hid + j (j < hid; pad lanes zero, excluded from the statistics).  partial: n * 32 * 2 doubles (written by _fwd, read by
 * _bwd: per-sample slice sums, the finished (mean, rstd) in slice 0 - opaque to the caller); bwd_partial: scratch of the same size.  eps inside the square root, biased variance. */
int sf_layernorm_chw_fwd(sfTensor x, int64_t n, int64_t pixels, int32_t gates, int32_t hid, int32_t hidp, const float* gamma,
                         const float* beta, float eps, double* partial, sfTensor y, sfStream stream);
int sf_layernorm_chw_bwd(sfTensor x, sfTensor dy, int64_t n, int64_t pixels, int32_t gates, int32_t hid, int32_t hidp,
                         const float* gamma, float eps, const double* partial, double* bwd_partial, sfTensor dx, float* dgamma,
                         float* dbeta, sfStream stream);
/* ---------------------------------------------------------------------------------------------
 * ST-LSTM cell with memory decoupling (PredRNN v2), SURVEY 8f-4: SpatioTemporalLSTMCell.forward,
 * satflow/models/layers/SpatioTemporalLSTMCell_memory_decoupling.py:110-138 (layer_norm=False).  Its 3x3 convolutions
 * (conv_x, conv_h, conv_m, conv_o) are sf_conv3x3_fwd, its 1x1 conv_last is sf_linear_fwd; these are the two pointwise stages.
 * All tensors NHWC fp32 with gate-major blocks of hidp padded hidden channels.
 *   gates (:114-132): gx [..,7*hidp] = conv_x(x): i f g i' f' g' o;  gh [..,4*hidp] = conv_h(h): i f g o;  gm [..,3*hidp] = conv_m(m):
 *     i' f' g'  ->  c' = sig(f_x+f_h+forget_bias) c + delta_c, delta_c = sig(i_x+i_h) tanh(g_x+g_h); m', delta_m likewise from the primed
 *     gates and gm; pre_o = o_x + o_h; mem = [c' | m'] (the input of conv_o / conv_last); gates (nullable) = i f g i' f' g' for backward
 *   out (:135-136): h' = sig(pre_o + conv_o) * tanh(last); saved (nullable) = [o | tanh(last)] for backward
 * The backward entries take NULL for absent incoming gradients. */
int sf_stlstm_gates_fwd(sfTensor gx, sfTensor gh, sfTensor gm, sfTensor c, sfTensor m, int64_t pixels, int32_t hidp,
                        float forget_bias, sfTensor c_new, sfTensor m_new, sfTensor mem, sfTensor delta_c, sfTensor delta_m,
                        sfTensor pre_o, sfTensor gates, int32_t dtype, sfStream stream);
int sf_stlstm_gates_bwd(sfTensor d_c_new, sfTensor d_m_new, sfTensor d_mem, sfTensor d_delta_c, sfTensor d_delta_m,
                        sfTensor d_pre_o, sfTensor gates, sfTensor c, sfTensor m, int64_t pixels, int32_t hidp, sfTensor dgx,
                        sfTensor dgh, sfTensor dgm, sfTensor dc, sfTensor dm, int32_t dtype, sfStream stream);
int sf_stlstm_out_fwd(sfTensor pre_o, sfTensor conv_o, sfTensor last, int64_t pixels, int32_t hidp, sfTensor h_new,
                      sfTensor saved, int32_t dtype, sfStream stream);
int sf_stlstm_out_bwd(sfTensor dh, sfTensor saved, int64_t pixels, int32_t hidp, sfTensor d_a, sfTensor d_last, int32_t dtype,
                      sfStream stream);

/* Lead-time de-duplication of MetNet's first convolution (ConditionTime planes are constant one-hot images and
 * conv1 is linear): with base = conv1_image(frame) + b computed ONCE per frame,
 *   pooled[(l*frames + f)] = maxpool2( base[f] + P_l ),  P_l[y][x][co] = sum of the in-image taps of w1[co][cimg + l]
 * (9 border classes).  Replaces ConditionTime (satflow/models/layers/ConditionTime.py:22-33) + the one-hot part of
 * conv1 + the first MaxPool2d of the DownSampler for all L lead times.  w1: conv1 weight [O][I][3][3] (fp32, OIHW).
 * bwd: dbase[f] = sum_l unpool(dout[l*frames+f]); dw1[:, cimg:cimg+L] = border-aware sums of the routed gradient
 * (other columns of dw1 are left untouched).  workspace: sf_leadtime_pool_workspace_floats(L, base.c) floats. */
size_t sf_leadtime_pool_workspace_floats(int32_t L, int32_t C);
int sf_leadtime_pool_fwd(sfTensor base, int64_t frames, int32_t h, int32_t w, const float* w1, int32_t O,
                         int32_t I, int32_t cimg, int32_t L, float* workspace, sfTensor out,
                         int32_t dtype, sfStream stream);
/* sf_leadtime_pool_fwd that also emits the statistics the BatchNorm behind it needs (the DownSampler's BatchNorm2d(160) follows
 * its first MaxPool2d): stats[(l * sf_leadtime_pool_stats_tiles() + block)][C][2] fp32 = per workgroup sum / sum of squares of the
 * STORED outputs of lead time l - the layout sf_batchnorm_train_fwd_stats reads with tiles_per_group = sf_leadtime_pool_stats_tiles()
 * and groups = L.  L <= 12. */
int32_t sf_leadtime_pool_stats_tiles(void);
int sf_leadtime_pool_fwd_stats(sfTensor base, int64_t frames, int32_t h, int32_t w, const float* w1, int32_t O, int32_t I,
                               int32_t cimg, int32_t L, float* workspace, sfTensor out, float* stats, int32_t dtype,
                               sfStream stream);
int sf_leadtime_pool_bwd(sfTensor base, sfTensor dout, int64_t frames, int32_t h, int32_t w,
                         const float* w1, int32_t O, int32_t I, int32_t cimg, int32_t L,
                         float* workspace, sfTensor dbase, float* dw1, int32_t dtype, sfStream stream);

/* ConvGRUCell step, recurrent half:  given gx = conv_x(x_t) = [z_x | r_x | n_x] (+ their biases; one
 * sf_conv3x3_fwd over all timesteps at once) and h_prev (NULL ptr = zero state):
 *   [z_h | r_h | h2] = conv3x3(h_prev) (+ bias on h2);  z = sig(z_x+z_h), r = sig(r_x+r_h),
 *   n = tanh(n_x + r*h2),  h' = (1-z)*n + z*h_prev.   gates (nullable) <- [z | r | n | h2] (SF_F32, or SF_BF16
 *   storage with the SF_BF16 kernel: backward-only data, the states do not depend on it).
 * wpacked: GRU nmap (32 hidden channels x 3 maps per N block, nf == 3). */
int sf_convgru_step_fwd(sfTensor gx, sfTensor h_prev, int32_t n, int32_t h, int32_t w,
                        const void* wpacked, const float* bias_packed, int32_t hidp, sfTensor h_out,
                        sfTensor gates, int32_t dtype, sfStream stream);
/* The recurrent half of ALL T steps in one launch, hidden state resident on chip (north star: "hidden-state residency across
 * timesteps"; replaces T calls of sf_convgru_step_fwd, i.e. the loop of upstream ConvGRU.forward behind pl_metnet.py:46-59).
 * One workgroup owns one map for the whole sequence: fp32 state in registers, its bf16 image (the next step's MFMA operand) in
 * LDS, recurrent weights streamed from L2 by LDS-DMA.  Maps of at most 16x16 pixels, hidp <= 64, SF_BF16 kernels only.
 *   gx    : [T][n][h][w][3*hidp] x-part of every step (SF_F32, or SF_BF16: then it is prefetched a K loop ahead)
 *   h0    : initial state [n][h][w][hidp] fp32 (ptr NULL = zeros)
 *   hs    : every state h_t, [T][n][h][w][hidp] fp32 (with fp32 gx bit-identical to the per-step calls)
 *   gates : nullable, [T][n][h][w][4*hidp] = [z | r | n | h2] per step (SF_F32 or SF_BF16) for the backward pass
 *   wpacked / bias_packed: as for sf_convgru_step_fwd (GRU map, nf == 3). */
int sf_convgru_seq_fwd(sfTensor gx, sfTensor h0, int32_t T, int32_t n, int32_t h, int32_t w, const void* wpacked,
                       const float* bias_packed, int32_t hidp, sfTensor hs, sfTensor gates, void* workspace,
                       size_t workspace_bytes, int32_t dtype, sfStream stream);
/* workspace (nullable): with sf_convgru_seq_fwd_workspace_bytes(n, h, hidp) bytes (0 = not applicable) the launch may split every
 * map over TWO workgroups of 8 rows each (maps of more than 8 rows, hidp 48..64; used when 2n <= number of CUs - MetNet's 96 maps
 * then run on 192 of the 256 CUs): after every step a workgroup hands the bf16 image of its boundary row to its partner inside the
 * launch ({epoch, value} granules, one write-through store each).  Same arithmetic, same results.
 * Residency is NOT assumed: a workgroup takes a start-order ticket (ticket k = map k / 2, half k % 2), so the launch completes
 * whenever two workgroups can be resident at a time, whatever else occupies the device (e.g. an RCCL kernel on another stream).
 * Layout (64-bit words): [0] STICKY error word - zeroed by the CALLER when it allocates the workspace, never cleared by the
 * library, OR-ed with 1 if a receiver gave up waiting (bounded spin: it never hangs); that workgroup's states are NaN from the
 * failed step on, so the failure also surfaces in the loss without any host synchronisation.  [1] ticket counter and [2 ..] the
 * granule slots are zeroed by every call (on `stream`).  A workspace serves one launch at a time. */
size_t sf_convgru_seq_fwd_workspace_bytes(int32_t n, int32_t h, int32_t hidp);
/* The backward time loop of the same sequence in ONE launch (one workgroup per image, maps of at most 16x16, hidp 32 or 64,
 * SF_BF16 kernels, bf16-stored gates): for t = T-1 .. 0 the gate backward (as sf_convgru_bwd_gates) and the recurrent
 * input-gradient convolution conv^T(dgh_t, Wh) (as sf_conv3x3_fwd with the transposed image wpacked_t: N = hidp, K = 3*hidp),
 * the carried gradient staying in registers.  g_seq [T][n][h][w][hidp] / g_last [n][h][w][hidp]: gradients wrt all states / the
 * last state (fp32, either may be NULL); gates, hs as written by sf_convgru_seq_fwd; out: dgx = [az|ar|an], dgh = [az|ar|dh2]
 * [T][n][h][w][3*hidp] bf16-stored - bit-identical to the per-step entry points. */
int sf_convgru_seq_bwd(sfTensor g_seq, sfTensor g_last, sfTensor gates, sfTensor hs, int32_t T, int32_t n, int32_t h, int32_t w,
                       const void* wpacked_t, int32_t hidp, sfTensor dgx, sfTensor dgh, void* workspace, size_t workspace_bytes,
                       int32_t dtype, sfStream stream);
/* workspace (nullable, sf_convgru_seq_bwd_workspace_bytes(n, h, hidp) bytes, 0 = not applicable): as for sf_convgru_seq_fwd - two
 * workgroups per map of more than 8 rows (hidp 64), the boundary row of dgh handed over inside the launch; same workspace layout,
 * tickets and failure behaviour (every gradient of the failing workgroup is NaN from the failed step on). */
size_t sf_convgru_seq_bwd_workspace_bytes(int32_t n, int32_t h, int32_t hidp);
/* (ABI 6: the test hook sf_convgru_seq_debug and its two process-wide variables left the product library - tests/native builds
 * convgru_seq.hip a second time with -DSF_TEST_HOOKS for the failure-path test.) */
/* Pointwise backward of the step: dh = dh0+dh1+dh2 -> dgx = [da_z|da_r|da_n], dgh = [da_z|da_r|dh2],
 * dh_direct = dh*z (nullable).  gates, and dgx / dgh (alike), may each be SF_BF16-stored: the two gradients are only ever
 * read as bf16 MFMA operands by sf_conv3x3_fwd / sf_conv3x3_bwd_weight.
 * (ABI 8) dgx.amax / dgh.amax (nullable, fp32-stored gradients): device words this launch RAISES to max |dgx| / max |dgh| (atomic maximum; the caller zeroes
 * them - one word may serve all steps of a sequence) - the SF_F32E kernels' scale words (sfTensor.amax) without sf_amax's extra pass. */
int sf_convgru_bwd_gates(sfTensor dh0, sfTensor dh1, sfTensor dh2, sfTensor gates, sfTensor h_prev,
                         int64_t pixels, int32_t hidp, sfTensor dgx, sfTensor dgh,
                         sfTensor dh_direct, int32_t dtype, sfStream stream);

/* Pointwise linear map over pixels (axial-attention to_q/to_kv/to_out, MetNet 1x1 head):
 *   y[p][n] = sum_k x[p][k]*W[n][k] + bias[n],  W row-major [N][x.c];  lanes N..y.c-1 are zeroed.
 * bwd_weight: dW[n][k] = sum_p dy[p][n]*x[p][k], db[n] = sum_p dy[p][n] (db nullable).
 * sf_linear_fwd with dtype SF_BF16 / SF_F16 (more than 64 rows): the operands are rounded to that type as they are loaded (RNE), fp32 accumulate, fp32-stored
 * tensors - a 1x1 Conv2d / Linear under the reference's 16-bit autocast (the DGMR discriminators' 1x1 convolutions, Discriminator.py:36-60,186-190;
 * configs/trainer/half.yaml:33); SF_F32: exact fp32 products. */
int sf_linear_fwd(sfTensor x, int64_t rows, const float* W, int32_t N, const float* bias, sfTensor y,
                  int32_t dtype, sfStream stream);
size_t sf_linear_bwd_weight_workspace_bytes(int32_t N, int32_t K, int64_t rows);
int sf_linear_bwd_weight(sfTensor dy, int32_t N, sfTensor x, int64_t rows, float* dW, float* db,
                         void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream);

/* AxialAttention core: qkv [img][h][w][6*hidp] = [q0|k0|v0|q1|k1|v1] (axis 0 attends along h, axis 1
 * along w; channel = head*(hid/heads)+j) -> att [..][2*hidp] = [axis0 | axis1], softmax(q k^T e^-1/2) v
 * per head.  bwd: datt [..][2*hidp] -> dqkv [..][6*hidp]. */
int sf_axial_attention_core_fwd(sfTensor qkv, int64_t nimg, int32_t h, int32_t w, int32_t hid,
                                int32_t hidp, int32_t heads, sfTensor att, int32_t dtype,
                                sfStream stream);
int sf_axial_attention_core_bwd(sfTensor qkv, sfTensor datt, int64_t nimg, int32_t h, int32_t w,
                                int32_t hid, int32_t hidp, int32_t heads, sfTensor dqkv,
                                int32_t dtype, sfStream stream);

/* MSE loss, its gradient and the per-forecast-frame losses in one pass.  pred/target: n contiguous floats viewed as
 * [outer][frames][inner] (frame index = (i / inner) % frames).  out[0] = mean (pred-target)^2, out[1+f] = mean over frame f,
 * grad (nullable) = 2 (pred - target) / n.  sums: 1+frames doubles of scratch.  Replaces nn.MSELoss and the per-frame
 * `.item()` loop of the Lightning steps (satflow/models/conv_lstm.py:63-69,79-84; pl_metnet.py:118-124). */
int sf_mse_loss(const float* pred, const float* target, int64_t n, int64_t inner, int32_t frames, float* grad,
                double* sums, float* out, sfStream stream);

/* Fused dropouts of the MetNet encoder output: y = x * m1(i) * m2(i % period) with keep-scaling; m1 = nn.Dropout(
 * temporal_dropout) (pl_metnet.py:58), m2 = the ConvGRU's sequence-consistent input dropout (time-major layout, period =
 * elements of one timestep, a multiple of 4).  Masks are a counter-based hash of (seed, index / 4) giving four 16-bit uniforms per
 * channel quad: the backward calls the same function on the gradient with the same seeds. */
int sf_dropout2(const float* x, int64_t n, float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2,
                float* y, sfStream stream);
/* (ABI 7) Up to any number of two-dimensional fp32 block copies in one launch per SF_MAX_BLOCKS: dst[r * dst_stride + c] = src[r * src_stride + c]
 * for r < rows, c < cols (strides in elements; src NULL: zeros).  `blocks` is a HOST array.  Regroups the reference's separate nn.Parameters into
 * the matrices the fused kernels take - upstream ConvGRUCell's conv_zr over [x ; h] + conv_h1 + conv_h2 -> x-part and h-part (call site
 * satflow/models/pl_metnet.py:46-59), the axial attention's to_q / to_kv / to_out of both axes -> one projection each - and scatters the
 * gradients back, instead of torch.cat / slicing (9 launches per cell and step each way). */
#define SF_MAX_BLOCKS 16
typedef struct sfBlock {
  const float* src;
  float* dst;
  int64_t rows, cols, src_stride, dst_stride;
  int64_t transpose; /* (ABI 8) != 0: the block lands TRANSPOSED, dst[c * dst_stride + r] = src[r * src_stride + c] (dst_stride >= rows) - the W^T operand
                        of an input-gradient GEMM built in the launch that builds W */
} sfBlock;
int sf_copy_blocks(const sfBlock* blocks, int32_t n, sfStream stream);

/* (ABI 6) The same masks on a bf16-stored tensor (x == y allowed; n a multiple of 8): the pooled encoder output of sf_conv3x3_fwd_folded_pool. */
int sf_dropout2_bf16(const void* x, int64_t n, float p1, float p2, int64_t period, uint64_t seed1, uint64_t seed2, void* y, sfStream stream);

/* =============================================================================================
 * CloudGAN around the ConvLSTM generator: the PatchGAN discriminator's convolutions and the GAN / L1 losses.
 * Reference: satflow/models/cloudgan.py:88-92,121-189, satflow/models/gan/discriminators.py:70-223.
 * ============================================================================================= */

/* nn.Conv2d(cin, cout, (kh, kw), stride, padding=pad) on NHWC fp32 activations (channel counts padded to 8), weights in the
 * reference's OIHW layout read directly.  leaky_slope != 1 fuses nn.LeakyReLU(slope) (discriminators.py:166-168).  Exact-fp32 MFMA.
 * Output size: (h + 2*pad - kh) / stride + 1.  Pad lanes of y are written as zeros. */
int sf_conv2d_fwd(sfTensor x, int32_t n, int32_t h, int32_t w, const float* weight, const float* bias, int32_t cin, int32_t cout,
                  int32_t kh, int32_t kw, int32_t stride, int32_t pad, float leaky_slope, sfTensor y, int32_t dtype, sfStream stream);
/* dx of the same convolution (h, w = INPUT size); dy [n][oh][ow][..]. */
int sf_conv2d_bwd_data(sfTensor dy, int32_t n, int32_t h, int32_t w, const float* weight, int32_t cin, int32_t cout, int32_t kh,
                       int32_t kw, int32_t stride, int32_t pad, sfTensor dx, int32_t dtype, sfStream stream);
/* dW [cout][cin][kh][kw] (+)= ..., db [cout] (+)= sum dy (db nullable).  workspace: sf_conv2d_bwd_weight_workspace_bytes(). */
size_t sf_conv2d_bwd_weight_workspace_bytes(int32_t n, int32_t oh, int32_t ow, int32_t cin, int32_t cout, int32_t kh, int32_t kw);
int sf_conv2d_bwd_weight(sfTensor x, sfTensor dy, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw,
                         int32_t stride, int32_t pad, float* dw, float* db, int32_t accumulate, void* workspace, size_t workspace_bytes,
                         int32_t dtype, sfStream stream);
/* y = x > 0 ? x : slope*x over n contiguous floats; sign_ref (nullable) supplies the sign instead of x: the backward pass is
 * sf_leaky_relu(dy, y_forward, ...) (slope > 0 preserves signs). */
int sf_leaky_relu(const float* x, const float* sign_ref, int64_t n, float slope, float* y, sfStream stream);
/* (ABI 7) dx = dy * y * (1 - y) over n contiguous floats (dx == dy allowed): backward of the sigmoid the head convolutions apply in their epilogue
 * (satflow/models/conv_lstm.py:200-203 `torch.nn.Sigmoid()(self.decoder_CNN(...))`), in torch's evaluation order. */
int sf_sigmoid_bwd(const float* dy, const float* y, int64_t n, float* dx, sfStream stream);

/* nn.L1Loss (mean) and nn.BCEWithLogitsLoss against a constant label (GANLoss "vanilla", discriminators.py:70-136) over the first c
 * lanes of `rows` NHWC rows split into `groups` equal contiguous groups (timesteps): out[0] = mean over everything, out[1+g] =
 * mean of group g; grad (nullable, same rows; its pad lanes are zeroed) = d out[0] / d pred.  sums: 1 + groups doubles scratch. */
int sf_l1_loss(sfTensor pred, sfTensor target, int64_t rows, int32_t groups, int32_t c, sfTensor grad, double* sums, float* out,
               sfStream stream);
int sf_bce_logits_loss(sfTensor logits, float label, float label_odd, int64_t rows, int32_t groups, int32_t c, sfTensor grad,
                       double* sums, float* out, sfStream stream); /* label for even groups, label_odd for odd ones */
/* The three objectives of GANLoss (gan/discriminators.py:70-136) in the same form: mode 1 "vanilla" (= sf_bce_logits_loss), 2 "lsgan"
 * (nn.MSELoss against the label), 3 "wgangp" (-mean for a real target, label > 0.5; +mean for a generated one). */
int sf_gan_loss(int32_t mode, sfTensor pred, float label, float label_odd, int64_t rows, int32_t groups, int32_t c, sfTensor grad,
                double* sums, float* out, sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * In-tree DGMR / DVD-GAN style networks: satflow/models/layers/{Normalization,GResBlock,Discriminator,Generator}.py.
 * Their convolutions run on sf_conv3x3_* (3x3; a Conv3d(3,3,3) as one 3x3 convolution over the channel stack of its three
 * temporal taps), sf_conv2d_* (5x5) and sf_linear_* (1x1); the entries below are everything else.  fp32, NHWC, 16-byte pixels.
 * ------------------------------------------------------------------------------------------- */
/* SpectralNorm._update_u_v (Normalization.py:19-31): `power_iterations` rounds of v = l2n(W^T u), u = l2n(W v) on the
 * [height][width] view of w_bar (u [height], v [width] are UPDATED IN PLACE - the module's persistent state), then
 * sigma = u . (W v) and w_out = w_bar / sigma.  workspace: sf_spectral_norm_workspace_floats() floats. */
size_t sf_spectral_norm_workspace_floats(int32_t height, int32_t width);
int sf_spectral_norm_fwd(const float* w_bar, int32_t height, int32_t width, float* u, float* v, int32_t power_iterations,
                         float* w_out, float* sigma, float* workspace, sfStream stream);
/* Its backward (u, v constants as in the reference, which iterates on .data): dw_bar = g / sigma - (<g, w_bar> / sigma^2) u v^T.
 * u, v, sigma as left by the forward call.  workspace: 512 floats. */
int sf_spectral_norm_bwd(const float* g, const float* w_bar, const float* u, const float* v, const float* sigma, int32_t height,
                         int32_t width, float* dw_bar, float* workspace, sfStream stream);
/* y [n_out][oh][ow] = scale * (sum over the 2x2 window, and over tpool consecutive frames) of x + addend (nullable).
 * Frames are time-major: image o = k * nb + b reads images (tpool * k + f) * nb + b.  F.avg_pool2d(x, 2) = (tpool 1, scale 1/4),
 * F.avg_pool3d(x, 2) = (tpool 2, scale 1/8) (GResBlock.py:81-82,89-90; Discriminator.py:214-215,267-268,382-383,438-439);
 * with scale 1 the backward pass of the nearest up-sampling.  `addend` fuses the residual sum (GResBlock.py:94). */
int sf_pool2(sfTensor x, int64_t n_out, int32_t oh, int32_t ow, int32_t tpool, int32_t nb, float scale, sfTensor addend, sfTensor y,
             sfStream stream);
/* y [n_in * texp][2h][2w] = scale * x (nearest neighbour; texp 2 also repeats every frame twice): F.interpolate(scale_factor=2)
 * (GResBlock.py:69-70,86-87) with scale 1; the backward pass of the average poolings with scale 1/4, 1/8. */
int sf_expand2(sfTensor x, int64_t n_in, int32_t h, int32_t w, int32_t texp, int32_t nb, float scale, sfTensor y, sfStream stream);
/* The three temporal taps of nn.Conv3d(k=3, padding=1) (Discriminator.py:345-356,405-410) as channels:
 * y[t][p][dt * C + c] = x[t + dt - 1][p][c] (zero outside the clip), time-major dense x [T][pixels_per_frame][C]; and its adjoint. */
int sf_time_stack3_fwd(sfTensor x, int32_t T, int64_t pixels_per_frame, sfTensor y, sfStream stream);
int sf_time_stack3_bwd(sfTensor gy, int32_t T, int64_t pixels_per_frame, sfTensor gx, sfStream stream);
/* A 5x5 'same' convolution (the generator's ConvGRU, Generator.py:29-68 kernel_sizes 5) as ONE 3x3 convolution on the MFMA kernels:
 * ys [n][h+4][w+4][4C], ys[q][s*C + c] = x[q - 2 + d_s][c] (zero outside the image), d_s = (2 (s>>1) - 1, 2 (s&1) - 1): the input on a
 * domain padded by 2, shifted four ways and stacked as channels; the 5x5 kernel is covered by four 3x3 tiles at offsets {0,2}^2
 * (shared middle row / column zeroed in the second tile) and the result is the interior of conv3x3(ys, tiles).  _bwd: the adjoint.
 * sf_border: pad == 0 crops a border of `border` pixels (x [n][h+2b][w+2b] -> y [n][h][w]), pad != 0 adds a zero border. */
int sf_pad_shift_stack4_fwd(sfTensor x, int64_t n, int32_t h, int32_t w, sfTensor y, sfStream stream);
int sf_pad_shift_stack4_bwd(sfTensor gy, int64_t n, int32_t h, int32_t w, sfTensor gx, sfStream stream);
int sf_border(sfTensor x, int64_t n, int32_t h, int32_t w, int32_t border, int32_t pad, sfTensor y, sfStream stream);
/* The weight side of the 5x5-as-3x3 route (sf_pad_shift_stack4_*; the generator's ConvGRU, Generator.py:29-68): [O][I][5][5] (rows `row_pitch` elements apart: a
 * column slice of a wider Conv2d weight is taken in place) -> [O][4 * lanes][3][3], tile (ty, tx) = taps (2 ty + ky, 2 tx + kx), duplicates of the middle
 * row / column masked, lanes >= I zero-padded; _bwd gathers the gradient of the 5x5 weight (dense [O][I][5][5]) back out of the 3x3 form. */
int sf_regroup5x5_fwd(const float* w5, int64_t row_pitch, int32_t O, int32_t I, int32_t lanes, float* w3, sfStream stream);
int sf_regroup5x5_bwd(const float* g3, int32_t O, int32_t I, int32_t lanes, float* g5, sfStream stream);
/* The same 5x5 convolution WITHOUT the stacked tensor (16-bit compute modes, fp32-stored x): the 3x3 kernels read x as four shifted views (view s = x
 * displaced by (2 (s>>1) - 1, 2 (s&1) - 1), zero outside the image) in their halo loader, with the weights of sf_regroup5x5_fwd packed for Kp = 4 * x.c - no padded
 * domain (64 x 64 pixels fill their tiles; 68 x 68 run at 60 %), no copies, no crop, and none of the half-resolution form's 4x weight bytes.  Few small images
 * with many channels are split over the virtual channel axis when a workspace of sf_conv5x5_fwd_workspace_bytes is handed in (0: never split).  The input
 * gradient is the same call on the output gradient with the flipped, transposed 5x5 weight (a 5x5 convolution again). */
size_t sf_conv5x5_fwd_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t Np, int32_t nf, int32_t xc, int32_t dtype);
int sf_conv5x5_fwd(sfTensor x, int32_t n, int32_t h, int32_t w, const void* wpacked, const float* bias_packed, int32_t Np, int32_t nf, sfTensor out,
                   void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream);
/* Its weight gradient: dW3 [O][4 * x.c][3][3] of that 3x3 convolution (sf_regroup5x5_bwd takes it back to the 5x5 weight), the four views of x read in
 * place by the loader waves; maps / workspace / accumulate as sf_conv3x3_bwd_weight (I = 4 * x lanes); x.c a multiple of 32. */
size_t sf_conv5x5_bwd_weight_workspace_bytes(int32_t Np, int32_t xc, int32_t n, int32_t h, int32_t w);
int sf_conv5x5_bwd_weight(sfTensor x, sfTensor dout, int32_t n, int32_t h, int32_t w, const int32_t* nmap, const int32_t* kmap, int32_t O, int32_t I, float* dw,
                          float* db, int32_t accumulate, void* workspace, size_t workspace_bytes, int32_t dtype, sfStream stream);
/* The same 5x5 'same' convolution on the HALF-RESOLUTION domain - the route the generator's ConvGRU SEQUENCE takes (Generator.py:91-117; kernel_sizes 5 at
 * :42-45): 2x2 pixel blocks of input and output folded into channels, xs[n][Y][X][(2 py + px) * C + c] = x[n][2Y + py][2X + px][c] (sf_space_to_depth2,
 * inverse != 0: the way back; a permutation, so each direction is the other's adjoint), then ONE 3x3 'same' convolution from 4C to 4R lanes with
 * w3[(g * 4 + po) * hp + o][pi * lanes + i][ey + 1][ex + 1] = w5[g * hp + o][i][2 ey + qy - py + 2][2 ex + qx - px + 2] (0 outside the 5x5 kernel / for i >= I):
 * `hp` rows per gate block, gate-major then output phase (the z | r halves of the gate tensor stay halves).  No padded domain, no copies, no crop; the
 * pointwise stages of the cell are layout-blind, so a whole sequence stays in this layout.  _bwd: the gradient of the dense [R][I][5][5] weight. */
int sf_space_to_depth2(sfTensor x, int64_t n, int32_t h, int32_t w, int32_t inverse, sfTensor y, sfStream stream);
int sf_regroup5x5_s2d_fwd(const float* w5, int64_t row_pitch, int32_t R, int32_t hp, int32_t I, int32_t lanes, float* w3, sfStream stream);
int sf_regroup5x5_s2d_bwd(const float* g3, int32_t R, int32_t hp, int32_t I, int32_t lanes, float* g5, sfStream stream);
/* A 4x4 stride-2 Conv2d with padding 1 - the down-sampling layers of the PatchGAN discriminator (satflow/models/gan/discriminators.py:166-186,
 * called by CloudGAN, satflow/models/cloudgan.py) - as ONE 3x3 convolution: x padded by 1 with its 2x2 pixel blocks folded into channels,
 * y[n][Y][X][(2 dy + dx) C + c] = x[n][2Y + dy - 1][2X + dx - 1][c] for Y < h/2 + 1, X < w/2 + 1 (h, w even); the 4x4 kernel becomes the taps
 * (1..2, 1..2) of a 3x3 kernel over 4C channels and the result is the first h/2 x w/2 outputs.  _bwd is the adjoint (a gather). */
int sf_pad_s2d_fwd(sfTensor x, int64_t n, int32_t h, int32_t w, sfTensor y, sfStream stream);
int sf_pad_s2d_bwd(sfTensor gy, int64_t n, int32_t h, int32_t w, sfTensor gx, sfStream stream);
/* nn.MaxPool3d of the in-tree attention layers (satflow/models/layers/Attention.py:50: kernel (2,1,1) stride (pf,1,1); :127: kernel 2
 * stride pf) over dense NHWC tokens x [batch][d0][d1][d2][C] -> y [batch][o0][o1][o2][C], o = (d - k) / s + 1; stride >= window.
 * Backward: the gradient goes to the first maximum of each window in scan order (torch's tie rule), recomputed from x. */
int sf_maxpool3d_fwd(sfTensor x, int64_t batch, int32_t d0, int32_t d1, int32_t d2, int32_t k0, int32_t k1, int32_t k2, int32_t s0,
                     int32_t s1, int32_t s2, sfTensor y, sfStream stream);
int sf_maxpool3d_bwd(sfTensor x, sfTensor gy, int64_t batch, int32_t d0, int32_t d1, int32_t d2, int32_t k0, int32_t k1, int32_t k2,
                     int32_t s0, int32_t s1, int32_t s2, sfTensor gx, sfStream stream);
/* ConditionalNorm (Normalization.py:76-85) behind its statistics, fused with what follows it in GResBlock.forward (:63-70,75-78):
 * y = act(gamma[img][c] * (x - mean[c]) * rstd[c] + beta[img][c]), embed [n][2 * creal] = gamma | beta (the Linear's output),
 * relu: F.relu, up: nearest x2 up-sampling of the result (y [n][2h][2w]).  mean / rstd: sf_batchnorm_train_fwd without affine. */
int sf_film_act_fwd(sfTensor x, int64_t n, int32_t h, int32_t w, const float* mean, const float* rstd, const float* embed,
                    int32_t creal, int32_t relu, int32_t up, sfTensor y, sfStream stream);
/* Backward: dxhat = d(normalised x) (then through sf_batchnorm_train_bwd), dembed [n][2 * creal] = d(gamma | beta).
 * workspace: sf_film_act_bwd_workspace_floats() floats. */
size_t sf_film_act_bwd_workspace_floats(int64_t n, int32_t h, int32_t w, int32_t c, int32_t creal);
int sf_film_act_bwd(sfTensor gy, sfTensor x, int64_t n, int32_t h, int32_t w, const float* mean, const float* rstd,
                    const float* embed, int32_t creal, int32_t relu, int32_t up, sfTensor dxhat, float* dembed, float* workspace,
                    sfStream stream);
/* out[img][c] = sum over the image's pixels of relu(x) (Discriminator.py:286-293,452-459) and its backward. */
int sf_relu_sum_pixels_fwd(sfTensor x, int64_t n, int64_t pixels, float* out, sfStream stream);
int sf_relu_sum_pixels_bwd(const float* g, sfTensor x, int64_t n, int64_t pixels, sfTensor gx, sfStream stream);
/* y = a * o + x over n floats, a = *gamma_dev (a device scalar: SelfAttention.gamma, Discriminator.py:125, Attention.py:108) or
 * `alpha` when gamma_dev is NULL; x nullable.  sf_dot: out[0] = <a, b> (gamma's gradient); workspace 512 floats. */
int sf_axpy(const float* o, const float* x, const float* gamma_dev, float alpha, int64_t n, float* y, sfStream stream);
int sf_dot(const float* a, const float* b, int64_t n, float* out, float* workspace, sfStream stream);
/* out = tanh(x) (Generator.py:126); with y_for_backward: out = x * (1 - y^2). */
int sf_tanh(const float* x, const float* y_for_backward, int64_t n, float* out, sfStream stream);
/* Gate arithmetic of the generator's ConvGRU (the module reference Generator.py:5 imports; restated in oracle/dgmr.py):
 *   gates: zr = [sig(gx_z + gh_z) | sig(gx_r + gh_r)], rh = r * h     (gx, gh: [.., 2*hidp], gh / h nullable = zeros)
 *   out:   cand = tanh(gx_o + gh_o), h_new = h (1 - z) + cand z
 * and their backward passes (dpre: gradient wrt the z | r pre-activations; da: wrt the candidate's pre-activation).  The gradient of z travels
 * between the two backward passes either as [.., hidp] or as the gradient of the whole z | r tensor [.., 2*hidp]: sf_dvdgru_out_bwd zeroes the r half
 * of such a dz, sf_dvdgru_gates_bwd reads the first hidp lanes of a row-strided dz. */
int sf_dvdgru_gates_fwd(sfTensor gx, sfTensor gh, sfTensor h, int64_t pixels, int32_t hidp, sfTensor zr, sfTensor rh, sfStream stream);
int sf_dvdgru_gates_bwd(sfTensor dz, sfTensor drh, sfTensor zr, sfTensor h, int64_t pixels, int32_t hidp, sfTensor dpre, sfTensor dh,
                        sfStream stream);
int sf_dvdgru_out_fwd(sfTensor gx, sfTensor gh, sfTensor zr, sfTensor h, int64_t pixels, int32_t hidp, sfTensor cand, sfTensor h_new,
                      sfStream stream);
int sf_dvdgru_out_bwd(sfTensor dh_new, sfTensor cand, sfTensor zr, sfTensor h, int64_t pixels, int32_t hidp, sfTensor da, sfTensor dz,
                      sfTensor dh, sfStream stream);

/* ---------------------------------------------------------------------------------------------
 * torch.bmm / softmax(dim=-1) of the in-tree attention layers (Discriminator.py:104-126; Attention.py:23-223) on exact-fp32 MFMA:
 * C[b][m][n] = alpha * sum_k A[b][m][k] B[b][k][n] + beta * C[b][m][n], every operand an arbitrary strided view (element strides),
 * because the reference multiplies `.view()` / `.permute()` windows of its tensors.  Softmax over contiguous rows of length L
 * (y may alias x; dx may alias g): dx = y * (g - <g, y>).
 * ------------------------------------------------------------------------------------------- */
int sf_bmm_f32(const float* A, int64_t sAb, int64_t sAm, int64_t sAk, const float* B, int64_t sBb, int64_t sBk, int64_t sBn, float* C,
               int64_t sCb, int64_t sCm, int64_t sCn, int32_t batch, int32_t M, int32_t N, int32_t K, float alpha, float beta,
               sfStream stream);
/* The same product with the operands rounded to bf16 (RNE) as they are staged, fp32 accumulate: what torch.bmm computes under torch.autocast, i.e. the
 * attention products of Discriminator.py:104-126 / Attention.py in the reference's 16-bit training mode (configs precision: 16).  fp32 data in memory. */
int sf_bmm_bf16(const float* A, int64_t sAb, int64_t sAm, int64_t sAk, const float* B, int64_t sBb, int64_t sBk, int64_t sBn, float* C,
                int64_t sCb, int64_t sCm, int64_t sCn, int32_t batch, int32_t M, int32_t N, int32_t K, float alpha, float beta,
                sfStream stream);
/* ... and rounded to fp16 (RNE; v_mfma_f32_32x32x16_f16): the SF_F16 compute mode's attention products (configs/trainer/half.yaml:33 `precision: 16`). */
int sf_bmm_f16(const float* A, int64_t sAb, int64_t sAm, int64_t sAk, const float* B, int64_t sBb, int64_t sBk, int64_t sBn, float* C,
               int64_t sCb, int64_t sCm, int64_t sCn, int32_t batch, int32_t M, int32_t N, int32_t K, float alpha, float beta,
               sfStream stream);
/* softmax(q k^T * scale) v in ONE pass, the score matrix never in memory (round 4): the self-attention of both discriminators
 * (Discriminator.py:104-126: `energy = torch.bmm(q, k^T)`, `softmax(dim=-1)`, `torch.bmm(attention, v)`; Attention.py:173-223) as the reference's 16-bit
 * mode computes it - operands and probabilities rounded to the compute type (SF_BF16 / SF_F16: the only dtypes built), fp32 accumulation and
 * softmax statistics.  q, k [batch][n][ld >= dqk], v, out [batch][n][ld >= dv], fp32, rows 16-byte aligned; n a multiple of 128, dqk 16 or 32
 * (narrower operands: zero-padded lanes), dv 32 / 64 / 128 / 256.  lse [batch][n] (nullable in _fwd) receives max + log(sum) of each query's scaled
 * scores: _bwd recomputes the probabilities from it.  _bwd: dq, dk [..][ld >= dqk], dv [..][ld >= dv] are overwritten (lanes >= dqk untouched),
 * delta [batch][n] is scratch.  Deterministic (no atomics). */
int sf_flash_attention_fwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, int32_t batch, int32_t n, int32_t dqk,
                           int32_t dv, float scale, float* out, int32_t ldo, float* lse, int32_t dtype, sfStream stream);
int sf_flash_attention_bwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* out, int32_t ldo,
                           const float* lse, const float* dout, int32_t lddo, int32_t batch, int32_t n, int32_t dqk, int32_t dv, float scale, float* dq,
                           int32_t lddq, float* dk, int32_t lddk, float* dvg, int32_t lddv, float* delta, int32_t dtype, sfStream stream);
int sf_softmax_rows_fwd(const float* x, int64_t rows, int32_t L, float* y, sfStream stream);
int sf_softmax_rows_bwd(const float* g, const float* y, int64_t rows, int32_t L, float* dx, sfStream stream);

#ifdef __cplusplus
}
#endif
#endif /* SATFLOW_HIP_H */
