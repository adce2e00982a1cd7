// The bf16 3x3 convolution kernels (conv3x3_bf16.hip) compiled for fp16 MFMA operands (v_mfma_f32_32x32x16_f16): SF_F16 compute mode - fp16 operands,
// fp32 accumulate, fp32 storage; linear / sigmoid epilogues and split-K launches.  What the reference's `precision: 16`
// (satflow/configs/trainer/half.yaml:33; BASELINE configs[4] "fp16") asks of the DGMR-style convolutions.
#define SF_OPERAND_F16 1
#include "conv3x3_bf16.hip"
