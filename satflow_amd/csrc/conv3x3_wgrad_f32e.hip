// The register-staged weight-gradient kernel (conv3x3_wgrad_bf16.hip) compiled as the SF_F32E compute mode: fp32-equivalent products from three fp16
// products per pixel tile (see conv3x3_f32e.hip and the SF_SPLIT3 notes in the included source).  Entry point: sf_launch_wgrad_f32e (wgrad_common.h).
#define SF_OPERAND_F16
#define SF_SPLIT3
#include "conv3x3_wgrad_bf16.hip"
