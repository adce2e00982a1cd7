/* Host-side sanitizer harness (sanitizers row): calls the argument validation of every
 * sf_* entry point of include/satflow_hip.h with NULL / misaligned / oversize / inconsistent arguments under
 * -fsanitize=address,undefined (host code only; GPU ASan is not available on this pool).  Every call must be REFUSED (non-zero
 * return, non-empty sf_last_error_string()) before anything is dereferenced or launched - no GPU is needed or touched.
 * Build + run: tools/sanitize_host.sh */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/satflow_hip.h"

static int failures = 0, calls = 0;
#define REFUSED(expr)                                                                                         \
  do {                                                                                                        \
    int rc_ = (expr);                                                                                         \
    ++calls;                                                                                                  \
    if (rc_ == 0 || strlen(sf_last_error_string()) == 0) { ++failures; printf("NOT REFUSED: %s (rc=%d)\n", #expr, rc_); } \
  } while (0)

static sfTensor T(void* p, int c, int stride, int dtype) { sfTensor t = {p, c, stride, 0, 0, dtype}; return t; }

int main(void) {
  static _Alignas(64) char mem[4096];
  void* ok = mem;          /* a 64-byte aligned non-null address standing in for a device pointer (never dereferenced) */
  void* mis = mem + 4;     /* misaligned */
  const sfTensor N0 = {0, 0, 0, 0, 0, 0};
  sfTensor a16 = T(ok, 16, 16, SF_F32), m16 = T(mis, 16, 16, SF_F32), odd = T(ok, 12, 12, SF_F32), b16 = T(ok, 16, 16, SF_BF16);
  sfTensor a64 = T(ok, 64, 64, SF_F32), a48 = T(ok, 48, 48, SF_F32), a192 = T(ok, 192, 192, SF_F32);
  void* st = 0;

  printf("abi %d\n", sf_abi_version());
  /* weight repack */
  REFUSED(sf_conv3x3_pack_weights(ok, 16, 16, ok, 32, ok, 16, 9, 0, ok, 0, 0, SF_F32, st));          /* nf out of range */
  REFUSED(sf_conv3x3_pack_weights(ok, 16, 16, ok, 32, ok, 16, 1, 0, ok, 0, 0, 7, st));               /* unknown dtype */
  /* 3x3 convolution */
  REFUSED(sf_conv3x3_fwd(odd, N0, 1, 8, 8, ok, 0, 32, 1, SF_EPI_LINEAR, a16, SF_F32, st));           /* channels not padded */
  REFUSED(sf_conv3x3_fwd(b16, N0, 1, 8, 8, ok, 0, 32, 1, SF_EPI_LINEAR, a16, SF_F32, st));           /* bf16 storage with the fp32 kernel */
  REFUSED(sf_conv3x3_fwd(a16, N0, 1, 8, 8, ok, 0, 32, 1, SF_EPI_LINEAR, m16, SF_BF16, st));          /* misaligned output */
  REFUSED(sf_conv3x3_fwd(a16, N0, 1, 8, 8, ok, 0, 32, 1, 9, a16, SF_F32, st));                       /* unknown epilogue */
  REFUSED(sf_conv3x3_fwd(a16, N0, 1, 8, 8, ok, 0, 32, 1, SF_EPI_LINEAR, a16, 5, st));                /* unknown dtype */
  REFUSED(sf_conv3x3_fwd(b16, N0, 1, 8, 8, ok, 0, 32, 1, SF_EPI_LINEAR, a16, SF_F16, st));           /* bf16 storage with the fp16 kernels */
  REFUSED(sf_conv3x3_fwd(b16, N0, 1, 8, 8, ok, 0, 32, 1, SF_EPI_LINEAR, a16, SF_F32E, st));          /* bf16 storage with the split-fp16 (fp32-equivalent) kernels */
  { sfTensor am = a16; am.amax = (const float*)ok;
    REFUSED(sf_conv3x3_fwd(a16, am, 1, 8, 8, ok, 0, 32, 1, SF_EPI_LINEAR, a16, SF_F32E, st)); }      /* an amax word on src1 */
  REFUSED(sf_amax(a16, 64, 0, 0, 0, st));                                                            /* no destination word */
  REFUSED(sf_amax(a16, 64, ok, (float*)(void*)(mem + 2), 0, st));                                                     /* misaligned accumulator word */
  REFUSED(sf_amax(b16, 64, ok, 0, 0, st));                                                           /* bf16 storage */
  REFUSED(sf_amax(m16, 64, ok, 0, 0, st));                                                           /* misaligned tensor */
  REFUSED(sf_conv3x3_fwd_stats(a16, N0, 1, 8, 8, ok, 0, 32, 1, a16, ok, SF_F32, st));                /* stats need the bf16 kernels */
  REFUSED(sf_conv3x3_fwd_splitk(a16, 1, 8, 8, ok, 0, 128, 4, a16, ok, 1 << 20, SF_F32, st));         /* split-K: SF_BF16 kernels only */
  REFUSED(sf_conv3x3_fwd_splitk(a16, 1, 8, 8, ok, 0, 128, 2, a16, ok, 1 << 20, SF_BF16, st));        /* split-K: nf = 4 only */
  REFUSED(sf_conv3x3_fwd_splitk(a16, 1, 8, 8, ok, 0, 128, 4, a16, ok, 1 << 20, SF_BF16, st));        /* a 16-channel source is not split */
  REFUSED(sf_conv3x3_fwd_splitk(T(ok, 256, 256, SF_F32), 1, 8, 8, ok, 0, 128, 4, a16, 0, 0, SF_BF16, st)); /* no workspace */
  REFUSED(sf_conv3x3_fwd_splitk(T(ok, 256, 256, SF_F32), 1, 8, 8, ok, 0, 128, 4, a16, ok, 16, SF_BF16, st)); /* workspace too small */
  /* folded BatchNorm */
  REFUSED(sf_conv3x3_fold_pack(ok, 16, 16, ok, 32, ok, 16, 1, 0, ok, ok, 2, ok, ok, SF_F32, st));        /* SF_BF16 kernels only */
  REFUSED(sf_conv3x3_fold_pack(ok, 16, 16, ok, 32, ok, 16, 1, 0, ok, 0, 2, ok, ok, SF_BF16, st));        /* no shift */
  REFUSED(sf_conv3x3_fwd_folded(b16, 3, 8, 8, ok, ok, 32, 1, 2, b16, 0, SF_BF16, st));                   /* images do not split into the groups */
  REFUSED(sf_conv3x3_fwd_folded(b16, 2, 1, 8, ok, ok, 32, 1, 2, b16, 0, SF_BF16, st));                   /* one-row image: no border classes */
  REFUSED(sf_conv3x3_fwd_folded(b16, 2, 8, 8, ok, mis, 32, 1, 2, b16, 0, SF_BF16, st));                  /* misaligned table */
  REFUSED(sf_conv3x3_fwd_folded_pool(b16, 2, 8, 8, ok, ok, 128, 4, 2, b16, 0, 0, ok, SF_BF16, st));      /* a shape the pooled-epilogue kernel does not take */
  REFUSED(sf_conv3x3_fwd_folded_pool(b16, 2048, 32, 32, ok, ok, 128, 4, 2, b16, 0, 0, 0, SF_BF16, st));  /* no routing buffer */
  REFUSED(sf_conv3x3_fwd_folded_pool(a16, 2048, 32, 32, ok, ok, 128, 4, 2, b16, 0, 0, ok, SF_BF16, st)); /* fp32-stored source */
  REFUSED(sf_conv3x3_fwd_folded_pool(b16, 2048, 32, 32, ok, ok, 128, 4, 2, b16, 3, 5, ok, SF_BF16, st)); /* images do not split into the permutation */
  /* ConvLSTM cell */
  REFUSED(sf_convlstm_cell_fwd(a16, a64, a64, 1, 8, 8, ok, 0, 48, a64, a64, N0, SF_F32, st));        /* hidp inconsistent with the tensors */
  REFUSED(sf_convlstm_cell_fwd(odd, a64, a64, 1, 8, 8, ok, 0, 64, a64, a64, N0, SF_F32, st));
  REFUSED(sf_convlstm_cell_bwd_gates(N0, N0, N0, N0, a64, N0, a64, 64, 60, a64, N0, SF_F32, st));    /* hidp not padded */
  REFUSED(sf_convlstm_cell_bwd_gates(a64, N0, N0, N0, a64, N0, a64, 64, 64, a64, N0, 3, st));
  /* weight gradient */
  REFUSED(sf_conv3x3_bwd_weight(a16, N0, a16, 1, 8, 8, ok, ok, 16, 16, ok, 0, 0, 0, 0, SF_F32, st)); /* no workspace */
  REFUSED(sf_conv3x3_bwd_weight(odd, N0, a16, 1, 8, 8, ok, ok, 16, 16, ok, 0, 0, ok, 1 << 20, SF_F32, st));
  REFUSED(sf_conv3x3_bwd_weight(b16, N0, a16, 1, 8, 8, ok, ok, 16, 16, ok, 0, 0, ok, 1 << 30, SF_BF16, st)); /* bf16 inputs, fp32 dout */
  /* layout */
  REFUSED(sf_nchw_to_nhwc(ok, 0, 0, 0, 1, 1, 20, 4, 4, a16, SF_F32, st));                            /* more channels than lanes */
  REFUSED(sf_nhwc_to_nchw(a16, 1, 1, 20, 4, 4, ok, 0, 0, 0, SF_F32, st));
  /* optimizer */
  REFUSED(sf_adam_step(mis, ok, ok, ok, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 1, 1.f, st));                /* misaligned flat buffer */
  REFUSED(sf_adam_step(ok, ok, ok, ok, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 0, 1.f, st));                 /* step counts from 1 */
  /* MetNet encoder */
  REFUSED(sf_metnet_preprocess_fwd(ok, 1, 1, 12, 12, 250, 256, 64, a192, SF_F32, st));               /* raw size != 4 * crop */
  REFUSED(sf_metnet_preprocess_fwd(ok, 1, 1, 12, 12, 256, 256, 64, a16, SF_F32, st));                /* too few output lanes */
  REFUSED(sf_metnet_preprocess_bwd(a16, 1, 1, 12, 12, 256, 256, 64, ok, SF_F32, st));
  REFUSED(sf_metnet_preprocess_bwd(a192, 1, 1, 12, 12, 256, 256, 64, mis, SF_F32, st));
  REFUSED(sf_maxpool2_fwd(a16, 1, 7, 8, a16, 0, 0, SF_F32, st));                                     /* odd height */
  REFUSED(sf_maxpool2_fwd(a16, 5, 8, 8, a16, 2, 3, SF_F32, st));                                     /* n not divisible by the permutation */
  REFUSED(sf_maxpool2_bwd(a16, a16, 1, 8, 8, a64, 0, 0, SF_F32, st));                                /* channel mismatch */
  { sfTensor bm = b16; bm.amax = (const float*)ok;
    REFUSED(sf_maxpool2_bwd(b16, b16, 1, 8, 8, bm, 0, 0, SF_F32, st)); }                                /* a scale word on a bf16-stored din */
  REFUSED(sf_maxpool2_dropout_fwd(a16, 1, 8, 8, a16, 0, 0, 1.5f, 0.f, 256, 1, 2, SF_F32, st));       /* p >= 1 */
  REFUSED(sf_maxpool2_dropout_bwd(a16, a16, 1, 8, 8, a16, 0, 0, 0.1f, 0.1f, 100, 1, 2, SF_F32, st)); /* period not whole images */
  REFUSED(sf_maxpool2_route_fwd(a16, 1, 8, 8, a16, 0, 0, 0.f, 0.f, 0, 0, 0, 0, SF_F32, st));         /* no routing buffer */
  REFUSED(sf_maxpool2_route_fwd(a16, 1, 8, 8, a16, 0, 0, 0.f, 0.f, 0, 0, 0, (char*)ok + 1, SF_F32, st)); /* misaligned routing buffer */
  REFUSED(sf_maxpool2_route_bwd(ok, a16, 1, 8, 8, N0, 0, 0, 0.f, 0.f, 0, 0, 0, 0, SF_F32, st));         /* no din */
  REFUSED(sf_batchnorm_train_fwd(odd, 64, 1, 12, ok, ok, 1e-5f, 0.1f, 0, 0, ok, ok, ok, ok, ok, odd, SF_F32, st));
  REFUSED(sf_batchnorm_train_fwd_stats(a16, 64, 1, 16, ok, ok, 1e-5f, 0.1f, 0, 0, ok, ok, ok, ok, ok, 0, 4, 32, a16, SF_F32, st)); /* null stats */
  REFUSED(sf_batchnorm_eval_fwd(a16, 64, 32, ok, ok, 1e-5f, ok, ok, ok, ok, a16, SF_F32, st));       /* creal > lanes */
  REFUSED(sf_batchnorm_eval_bwd(a16, a64, 64, 16, ok, 1e-5f, ok, ok, ok, ok, a16, ok, ok, SF_F32, st));
  REFUSED(sf_batchnorm_train_bwd(a16, a64, 64, 1, 16, ok, ok, ok, ok, ok, a16, ok, ok, SF_F32, st));
  { sfTensor bm = b16; bm.amax = (const float*)ok;
    REFUSED(sf_batchnorm_train_bwd(b16, b16, 64, 1, 16, ok, ok, ok, ok, ok, bm, ok, ok, SF_F32, st)); }   /* a scale word on a bf16-stored dx */
  REFUSED(sf_batchnorm_train_bwd_coef(0, 64, 1, 16, 16, ok, ok, ok, ok, ok, ok, SF_F32, st));           /* no sums */
  REFUSED(sf_batchnorm_train_bwd_coef(ok, 64, 1, 16, 24, ok, ok, ok, ok, ok, ok, SF_F32, st));          /* more real channels than lanes */
  REFUSED(sf_leadtime_pool_fwd(a16, 1, 7, 8, ok, 16, 20, 8, 12, ok, a16, SF_F32, st));               /* odd height */
  REFUSED(sf_leadtime_pool_fwd_stats(a16, 4, 8, 8, ok, 16, 20, 4, 4, ok, a16, 0, SF_F32, st));          /* no stats */
  REFUSED(sf_leadtime_pool_fwd_stats(a16, 4, 8, 8, ok, 16, 40, 4, 13, ok, a16, ok, SF_F32, st));        /* more lead times than the kernel keeps in registers */
  REFUSED(sf_leadtime_pool_bwd(a16, a64, 1, 8, 8, ok, 16, 20, 8, 12, ok, a16, ok, SF_F32, st));
  /* ConvGRU */
  REFUSED(sf_convgru_step_fwd(a48, a16, 1, 8, 8, ok, 0, 20, a16, N0, SF_F32, st));                   /* hidp not padded */
  REFUSED(sf_convgru_step_fwd(a48, a16, 1, 8, 8, ok, 0, 16, N0, N0, SF_F32, st));                    /* no output */
  REFUSED(sf_convgru_seq_fwd(a192, N0, 4, 2, 16, 16, ok, 0, 64, a64, N0, 0, 0, SF_F32, st));         /* persistent kernel: bf16 kernels only */
  REFUSED(sf_convgru_seq_bwd(N0, a64, T(ok, 256, 256, SF_BF16), a64, 4, 2, 32, 16, ok, 64, T(ok, 192, 192, SF_BF16), T(ok, 192, 192, SF_BF16), 0, 0, SF_BF16, st)); /* map taller than 16 */
  REFUSED(sf_convgru_seq_bwd(N0, a64, T(ok, 256, 256, SF_F32), a64, 4, 2, 16, 16, ok, 64, T(ok, 192, 192, SF_BF16), T(ok, 192, 192, SF_BF16), 0, 0, SF_BF16, st));  /* fp32-stored gates */
  REFUSED(sf_convgru_seq_bwd(N0, a48, T(ok, 192, 192, SF_BF16), a48, 4, 2, 16, 16, ok, 48, T(ok, 144, 144, SF_BF16), T(ok, 144, 144, SF_BF16), 0, 0, SF_BF16, st)); /* hidp 48 */
  REFUSED(sf_convgru_seq_fwd(a192, N0, 4, 2, 32, 16, ok, 0, 64, a64, N0, 0, 0, SF_BF16, st));        /* map larger than one workgroup */
  REFUSED(sf_convgru_seq_fwd(a192, N0, 4, 2, 16, 16, ok, 0, 128, a64, N0, 0, 0, SF_BF16, st));       /* hidp > 64 */
  REFUSED(sf_convgru_seq_fwd(a192, N0, 4, 2, 16, 16, ok, 0, 64, m16, N0, 0, 0, SF_BF16, st));        /* misaligned / wrong-width states */
  REFUSED(sf_convgru_bwd_gates(N0, N0, N0, a64, N0, 64, 16, a48, a48, N0, SF_F32, st));              /* dh0 missing */
  /* linear / attention */
  REFUSED(sf_linear_fwd(a16, 64, ok, 300, 0, a16, SF_F32, st));                                      /* N exceeds the output lanes */
  REFUSED(sf_linear_fwd(b16, 64, ok, 8, 0, a16, SF_F32, st));                                        /* bf16 storage unsupported */
  REFUSED(sf_linear_bwd_weight(a16, 8, a16, 64, ok, 0, 0, 0, SF_F32, st));                           /* no workspace */
  REFUSED(sf_axial_attention_core_fwd(a16, 1, 16, 16, 64, 64, 8, a16, SF_F32, st));                  /* qkv narrower than 6 * hidp */
  REFUSED(sf_axial_attention_core_bwd(a16, a16, 1, 16, 16, 60, 64, 8, a16, SF_F32, st));             /* hid not divisible by heads */
  /* losses / dropout */
  REFUSED(sf_mse_loss(ok, ok, 100, 7, 3, 0, ok, ok, st));                                            /* n not divisible into frames */
  REFUSED(sf_dropout2(ok, 64, 1.0f, 0.f, 64, 1, 2, ok, st));                                         /* p >= 1 */
  REFUSED(sf_dropout2(ok, 64, 0.1f, 0.1f, 30, 1, 2, ok, st));                                        /* period not a multiple of 4 */
  { sfBlock bad = {ok, ok, 4, 8, 4, 8};   /* source rows narrower than the block */
    REFUSED(sf_copy_blocks(&bad, 1, st));
    REFUSED(sf_copy_blocks(0, 1, st));     /* no table */
    sfBlock trbad = {ok, ok, 8, 4, 4, 4, 1};   /* transposed landing: the destination rows (stride 4) are narrower than the block's 8 rows */
    REFUSED(sf_copy_blocks(&trbad, 1, st));
    sfBlock nodst = {ok, 0, 4, 8, 8, 8};
    REFUSED(sf_copy_blocks(&nodst, 1, st)); }
  REFUSED(sf_dropout2_bf16(ok, 60, 0.1f, 0.1f, 64, 1, 2, ok, st));                                   /* n not a multiple of 8 */
  REFUSED(sf_dropout2_bf16(ok, 64, 0.1f, 1.0f, 64, 1, 2, ok, st));                                   /* p >= 1 */
  /* CloudGAN side network */
  REFUSED(sf_conv2d_fwd(odd, 1, 8, 8, ok, 0, 12, 16, 4, 4, 2, 1, 0.2f, a16, SF_F32, st));            /* channels not padded to 8 */
  REFUSED(sf_conv2d_fwd(a16, 1, 2, 2, ok, 0, 16, 16, 4, 4, 1, 0, 1.f, a16, SF_F32, st));             /* empty output */
  REFUSED(sf_conv2d_fwd(a16, 1, 8, 8, ok, 0, 16, 16, 4, 4, 2, 1, 1.f, a16, SF_BF16, st));            /* fp32 kernel only */
  REFUSED(sf_conv2d_bwd_data(a16, 1, 8, 8, 0, 16, 16, 4, 4, 2, 1, a16, SF_F32, st));                 /* null weight */
  REFUSED(sf_conv2d_bwd_weight(a16, a16, 1, 8, 8, 16, 16, 4, 4, 2, 1, ok, 0, 0, 0, 0, SF_F32, st));  /* no workspace */
  REFUSED(sf_sigmoid_bwd(ok, mis, 64, ok, st));       /* misaligned */
  REFUSED(sf_sigmoid_bwd(ok, ok, 63, ok, st));        /* n % 4 */
  REFUSED(sf_leaky_relu(mis, 0, 64, 0.2f, ok, st));
  REFUSED(sf_leaky_relu(ok, 0, 63, 0.2f, ok, st));
  REFUSED(sf_conv3x3_bwd_weight_folded(a16, b16, 2, 8, 8, ok, ok, 16, 16, ok, ok, 2, ok, 0, 0, 0, 0, 0, 0, ok, 1 << 30, SF_BF16, st)); /* fp32 source */
  REFUSED(sf_conv3x3_bwd_weight_folded(b16, b16, 2, 8, 8, ok, ok, 16, 16, ok, ok, 2, ok, 0, 0, 0, 0, 0, 0, ok, 16, SF_BF16, st));      /* workspace too small */
  REFUSED(sf_conv3x3_bwd_weight_folded(b16, b16, 2, 8, 8, ok, ok, 16, 16, ok, ok, 2, ok, 0, 0, 0, ok, ok, ok, ok, 1 << 30, SF_BF16, st)); /* bn_sums without the weights */
  REFUSED(sf_conv3x3_bwd_weight_folded(b16, b16, 3, 8, 8, ok, ok, 16, 16, ok, ok, 2, ok, 0, 0, 0, 0, 0, 0, ok, 1 << 30, SF_BF16, st)); /* ragged groups */
  REFUSED(sf_conv3x3_bwd_weight_folded_sparse24(a16, b16, 2, 8, 8, ok, ok, 16, 16, ok, ok, 2, ok, 0, 0, 0, 0, 0, 0, N0, 0, 0, 0, ok, 1 << 30, SF_BF16, st)); /* fp32 source */
  REFUSED(sf_conv3x3_bwd_weight_folded_sparse24(b16, b16, 2, 8, 8, ok, ok, 16, 16, ok, ok, 2, ok, 0, 0, 0, 0, 0, 0, N0, 0, 0, 0, ok, 1 << 30, SF_BF16, st)); /* 16 lanes: no 128 x 64 slabs */
  REFUSED(sf_conv3x3_bwd_weight_folded_sparse24(b16, b16, 2, 7, 7, ok, ok, 16, 16, ok, ok, 2, ok, 0, 0, 0, 0, 0, 0, N0, 0, 0, 0, ok, 1 << 30, SF_BF16, st)); /* odd maps: no 2x2 windows */
  REFUSED(sf_conv3x3_bwd_data_bn(b16, 2, 8, 8, ok, 32, 1, a16, ok, 2, b16, SF_BF16, st));               /* fp32-stored x */
  REFUSED(sf_conv3x3_bwd_data_bn(b16, 3, 8, 8, ok, 32, 1, b16, ok, 2, b16, SF_BF16, st));               /* ragged groups */
  REFUSED(sf_conv3x3_bwd_data_bn(b16, 2, 8, 8, ok, 32, 1, b16, mis, 2, b16, SF_BF16, st));              /* misaligned coefficients */
  REFUSED(sf_l1_loss(a16, N0, 64, 1, 16, N0, ok, ok, st));                                           /* no target */
  REFUSED(sf_l1_loss(a16, a16, 64, 3, 16, N0, ok, ok, st));                                          /* rows not divisible into groups */
  REFUSED(sf_bce_logits_loss(a16, 1.f, 0.f, 64, 1, 32, N0, ok, ok, st));                             /* more lanes than the stride */
  /* ST-LSTM pointwise stages */
  REFUSED(sf_stlstm_gates_fwd(a192, a64, a48, a16, a16, 64, 16, 1.f, a16, a16, a16, a16, a16, a16, N0, SF_F32, st));  /* mem narrower than 2*hidp */
  REFUSED(sf_stlstm_gates_fwd(a192, a64, a48, a16, a16, 64, 16, 1.f, a16, a16, a64, a16, a16, a16, N0, SF_BF16, st)); /* dtype not built */
  REFUSED(sf_stlstm_gates_bwd(N0, N0, N0, N0, N0, N0, N0, a16, a16, 64, 16, a192, a64, a48, a16, a16, SF_F32, st));   /* no saved gates */
  REFUSED(sf_stlstm_out_fwd(a16, a16, a16, 64, 12, a16, N0, SF_F32, st));                                             /* hidp not padded */
  REFUSED(sf_stlstm_out_bwd(a16, a16, 64, 16, a16, a16, SF_F32, st));                                                 /* saved narrower than 2*hidp */
  /* DGMR / attention stages */
  REFUSED(sf_spectral_norm_fwd(ok, 8, 8, ok, ok, 0, ok, ok, ok, st));                                    /* zero power iterations */
  REFUSED(sf_spectral_norm_fwd(ok, 8, 8, 0, ok, 1, ok, ok, ok, st));                                     /* no u */
  REFUSED(sf_spectral_norm_bwd(ok, ok, ok, ok, 0, 8, 8, ok, ok, st));                                    /* no sigma */
  REFUSED(sf_pool2(a16, 4, 4, 4, 3, 1, 0.25f, N0, a16, st));                                             /* temporal window 3 */
  REFUSED(sf_pool2(a16, 5, 4, 4, 2, 2, 0.125f, N0, a16, st));                                            /* images do not split into frames */
  REFUSED(sf_pool2(b16, 4, 4, 4, 1, 1, 0.25f, N0, a16, st));                                             /* bf16 storage */
  REFUSED(sf_expand2(a16, 4, 4, 4, 1, 1, 1.f, m16, st));                                                 /* misaligned output */
  REFUSED(sf_time_stack3_fwd(a16, 4, 64, a16, st));                                                      /* output must have 3x the lanes */
  REFUSED(sf_time_stack3_bwd(a16, 4, 64, a16, st));
  REFUSED(sf_pad_shift_stack4_fwd(a16, 1, 8, 8, a48, st));                                               /* output must have 4x the lanes */
  REFUSED(sf_pad_shift_stack4_bwd(a48, 1, 8, 8, a16, st));
  REFUSED(sf_regroup5x5_fwd(ok, 100, 4, 8, 4, ok, st));                                                /* lanes < I */
  REFUSED(sf_regroup5x5_fwd(ok, 10, 4, 8, 16, ok, st));                                                /* row pitch smaller than a row */
  REFUSED(sf_regroup5x5_bwd(0, 4, 8, 16, ok, st));                                                     /* no gradient */
  REFUSED(sf_conv5x5_fwd(b16, 1, 8, 8, ok, 0, 128, 4, a16, 0, 0, SF_BF16, st));                          /* bf16-stored x: fp32 storage only */
  REFUSED(sf_conv5x5_fwd(a16, 1, 8, 8, ok, 0, 128, 4, a16, 0, 0, SF_F32, st));                           /* 16-bit operand kernels only */
  REFUSED(sf_conv5x5_fwd(a16, 1, 8, 8, ok, 0, 160, 5, a16, 0, 0, SF_BF16, st));                          /* NF = 5 has no shifted-view loader */
  REFUSED(sf_conv5x5_bwd_weight(a16, a16, 1, 8, 8, ok, ok, 16, 64, ok, 0, 0, ok, 1 << 20, SF_BF16, st)); /* lanes not a multiple of 32 */
  REFUSED(sf_conv5x5_bwd_weight(a64, a16, 1, 8, 8, ok, ok, 16, 256, ok, 0, 0, ok, 16, SF_BF16, st));     /* workspace too small */
  REFUSED(sf_space_to_depth2(a16, 1, 8, 8, 0, a48, st));                                               /* y must carry 4C lanes */
  REFUSED(sf_space_to_depth2(a16, 1, 7, 8, 0, a64, st));                                               /* odd height */
  REFUSED(sf_space_to_depth2(a16, 1, 8, 8, 1, a64, st));                                               /* inverse: x is the 4C side */
  REFUSED(sf_regroup5x5_s2d_fwd(ok, 100, 24, 16, 8, 16, ok, st));                                      /* rows not whole gate blocks */
  REFUSED(sf_regroup5x5_s2d_fwd(ok, 100, 32, 16, 8, 4, ok, st));                                       /* lanes < I */
  REFUSED(sf_regroup5x5_s2d_bwd(0, 32, 16, 8, 16, ok, st));                                            /* no gradient */
  REFUSED(sf_pad_s2d_fwd(a16, 1, 8, 8, a48, st));                                                      /* y must carry 4C lanes */
  REFUSED(sf_pad_s2d_fwd(a16, 1, 7, 8, a64, st));                                                      /* odd height */
  REFUSED(sf_pad_s2d_bwd(a48, 1, 8, 8, a16, st));                                                      /* gy must carry 4C lanes */
  REFUSED(sf_pad_s2d_bwd(a64, 1, 8, 9, a16, st));                                                      /* odd width */
  REFUSED(sf_border(a16, 1, 8, 8, 0, 0, a16, st));                                                       /* empty border */
  REFUSED(sf_border(a16, 1, 8, 8, 2, 1, a64, st));                                                       /* channel mismatch */
  REFUSED(sf_film_act_fwd(a16, 2, 8, 8, ok, ok, 0, 12, 1, 0, a16, st));                                  /* no embedding */
  REFUSED(sf_film_act_fwd(a16, 2, 8, 8, ok, ok, ok, 20, 1, 0, a16, st));                                 /* more real channels than lanes */
  REFUSED(sf_film_act_bwd(a16, a16, 2, 8, 8, ok, ok, ok, 12, 1, 0, a16, ok, 0, st));                     /* no workspace */
  REFUSED(sf_relu_sum_pixels_fwd(a16, 2, 64, 0, st));                                                    /* no output */
  REFUSED(sf_relu_sum_pixels_bwd(ok, a16, 2, 64, a64, st));                                              /* channel mismatch */
  REFUSED(sf_axpy(ok, ok, 0, 1.f, 6, ok, st));                                                           /* length not a multiple of 4 */
  REFUSED(sf_axpy(mis, 0, 0, 1.f, 8, ok, st));                                                           /* misaligned */
  REFUSED(sf_dot(ok, ok, 8, ok, 0, st));                                                                 /* no workspace */
  REFUSED(sf_tanh(ok, 0, 10, ok, st));                                                                   /* length not a multiple of 4 */
  REFUSED(sf_dvdgru_gates_fwd(a16, N0, N0, 64, 16, a16, a16, st));                                       /* gx narrower than 2*hidp */
  REFUSED(sf_dvdgru_gates_bwd(N0, N0, a16, N0, 64, 16, a16, N0, st));                                    /* zr narrower than 2*hidp */
  REFUSED(sf_dvdgru_out_fwd(a16, N0, a16, N0, 64, 16, N0, a16, st));                                     /* zr narrower than 2*hidp */
  REFUSED(sf_dvdgru_out_bwd(a16, a16, a16, N0, 64, 16, a16, a16, N0, st));
  REFUSED(sf_bmm_f32(ok, 0, 8, 1, 0, 0, 8, 1, ok, 0, 8, 1, 1, 8, 8, 8, 1.f, 0.f, st));                   /* no B */
  REFUSED(sf_bmm_f32(ok, 0, 8, 1, ok, 0, 8, 1, ok, 0, 8, 1, 1, 8, 8, 0, 1.f, 0.f, st));                  /* empty inner dimension */
  REFUSED(sf_bmm_bf16(ok, 0, 8, 1, 0, 0, 8, 1, ok, 0, 8, 1, 1, 8, 8, 8, 1.f, 0.f, st));                  /* no B */
  REFUSED(sf_bmm_bf16(ok, 0, 8, 1, ok, 0, 8, 1, 0, 0, 8, 1, 1, 8, 8, 8, 1.f, 0.f, st));                  /* no C */
  REFUSED(sf_bmm_bf16(ok, 0, 8, 1, ok, 0, 8, 1, ok, 0, 8, 1, 1, 8, 8, 0, 1.f, 0.f, st));                 /* empty inner dimension */
  REFUSED(sf_bmm_f16(ok, 0, 8, 1, 0, 0, 8, 1, ok, 0, 8, 1, 1, 8, 8, 8, 1.f, 0.f, st));                   /* no B */
  REFUSED(sf_bmm_f16(ok, 0, 8, 1, ok, 0, 8, 1, ok, 0, 8, 1, 1, 8, 8, 0, 1.f, 0.f, st));                  /* empty inner dimension */
  REFUSED(sf_flash_attention_fwd(ok, 32, ok, 32, ok, 256, 1, 128, 32, 256, 1.f, ok, 256, ok, SF_F32, st));     /* fp32: only the 16-bit modes have the fused form */
  REFUSED(sf_flash_attention_fwd(ok, 32, ok, 32, ok, 256, 1, 100, 32, 256, 1.f, ok, 256, ok, SF_BF16, st));    /* n not a multiple of 128 */
  REFUSED(sf_flash_attention_fwd(ok, 32, ok, 32, ok, 256, 1, 128, 24, 256, 1.f, ok, 256, ok, SF_BF16, st));    /* key width */
  REFUSED(sf_flash_attention_fwd(ok, 32, ok, 32, 0, 256, 1, 128, 32, 256, 1.f, ok, 256, ok, SF_F16, st));      /* no v */
  REFUSED(sf_flash_attention_bwd(ok, 32, ok, 32, ok, 256, ok, 256, 0, ok, 256, 1, 128, 32, 256, 1.f, ok, 32, ok, 32, ok, 256, ok, SF_BF16, st));   /* no lse */
  REFUSED(sf_flash_attention_bwd(ok, 32, ok, 32, ok, 256, ok, 256, ok, ok, 256, 1, 128, 32, 96, 1.f, ok, 32, ok, 32, ok, 256, ok, SF_BF16, st));   /* value width */
  REFUSED(sf_flash_attention_bwd(ok, 32, ok, 32, ok, 256, ok, 256, ok, ok, 256, 1, 128, 32, 256, 1.f, ok, 32, ok, 32, ok, 256, ok, SF_F32, st));   /* fp32 */
  REFUSED(sf_softmax_rows_fwd(ok, 4, 0, ok, st));                                                        /* empty rows */
  REFUSED(sf_softmax_rows_bwd(ok, 0, 4, 8, ok, st));                                                     /* no softmax output */
  REFUSED(sf_layernorm_chw_fwd(a64, 2, 64, 4, 12, 16, ok, ok, 1e-5f, 0, a64, st));                       /* no partial-sum buffer */
  REFUSED(sf_layernorm_chw_fwd(a48, 2, 64, 4, 12, 16, ok, ok, 1e-5f, ok, a48, st));                      /* lanes != gates * hidp */
  REFUSED(sf_layernorm_chw_bwd(a64, a64, 2, 64, 4, 20, 16, ok, 1e-5f, ok, ok, a64, ok, ok, st));         /* hid > hidp */
  REFUSED(sf_maxpool3d_fwd(a16, 1, 4, 4, 4, 2, 2, 2, 1, 2, 2, a16, st));                                   /* stride below the window */
  REFUSED(sf_maxpool3d_bwd(a16, a16, 1, 4, 4, 4, 8, 1, 1, 8, 1, 1, a16, st));                              /* window beyond the extent */
  REFUSED(sf_gan_loss(4, a16, 1.f, 0.f, 64, 1, 1, N0, ok, ok, st));                                        /* unknown objective */
  /* size queries never fail, must not overflow */
  printf("packed %zu ws %zu %zu %zu\n", sf_conv3x3_packed_elems(256, 256), sf_conv3x3_bwd_weight_workspace_bytes(256, 256, 2304, 32, 32) + sf_conv3x3_bwd_weight_folded_workspace_bytes(256, 256, 2304, 32, 32, 24),
         sf_linear_bwd_weight_workspace_bytes(384, 64, 24576), sf_conv2d_bwd_weight_workspace_bytes(48, 64, 64, 12, 32, 4, 4));
  printf("%d calls, %d not refused\n", calls, failures);
  return failures ? 1 : 0;
}
