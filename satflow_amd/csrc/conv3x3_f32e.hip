// The 16-bit 3x3 convolution kernels (conv3x3_bf16.hip) compiled as the SF_F32E compute mode: fp32-EQUIVALENT products on the fp16 matrix pipe.
// gfx950 has no TF32, and its exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, conv3x3_f32.hip) runs at 1/16 of the 16-bit rate.  Here every fp32 operand is
// split while it is staged into LDS - x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16((x - hi) * 2^11): 22 mantissa bits - and the K loop makes three passes
// (hi * lo' and lo' * hi first, one exact 2^-11 rescale of the accumulators, then hi * hi): per-product error ~2^-22, inside the fp32 parity gate
// (rtol 1e-4 / atol 1e-5) on every reference golden with >= 5x margin, at 3 of the 16 (tools/probe_f32e_numerics.py, profiles/r06_f32e_numerics.txt).
// Entry points: sf_launch_conv_f32e / sf_pack_weights_f32e (conv_common.h).  fp32-stored tensors; linear, sigmoid, LSTM and GRU epilogues.
#define SF_OPERAND_F16
#define SF_SPLIT3
#include "conv3x3_bf16.hip"
