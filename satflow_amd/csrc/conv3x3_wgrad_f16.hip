// The register-staged weight-gradient kernel (conv3x3_wgrad_bf16.hip) compiled for fp16 MFMA operands: SF_F16 compute mode (fp32-stored tensors).
#define SF_OPERAND_F16 1
#include "conv3x3_wgrad_bf16.hip"
