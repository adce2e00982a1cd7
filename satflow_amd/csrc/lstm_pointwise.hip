// Pointwise half of the ConvLSTM cell backward: from the incoming dh / dc' and the saved gates
// to dz (gradient wrt the pre-activation conv output, gate-major [.., 4*hidp]) and dc_prev.
// HBM-bound streaming kernel: 16-byte accesses, one pass; no atomics except the optional scale word (dz.amax: at most one atomic maximum per wave).
// Autograd of satflow/models/layers/ConvLSTM.py:48-55.
#include "sf_common.h"

namespace {

struct GateBwdParams {
  const void* dh0; const void* dh1; const void* dh2; int s_dh0, s_dh1, s_dh2;
  int bf_dh0, bf_dh1, bf_dh2;   // per source: stored as bf16 (the input-gradient convolutions' output in "bf16a" mode) or fp32
  const float* dc_next; int s_dcn;
  const void* gates; int s_g;
  const float* c_prev; int s_cp;
  const float* c_new; int s_cn;
  void* dz; int s_dz;
  float* dc_prev; int s_dcp;
  long long pixels; int hidp;
  unsigned* amax;   // (nullable) raised to max |dz| by this launch: the SF_F32E kernels' scale word of dz (sfTensor::amax); reset by the caller
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// TG: storage type of the saved gates and of dz (which overwrites them in place): fp32, or bf16 in "bf16a" mode
template <typename TG>
__global__ __launch_bounds__(256) void lstm_bwd_gates_kernel(const GateBwdParams p) {
  const int q = p.hidp >> 2;
  const long long total = p.pixels * q;
  unsigned zmax = 0u;   // largest |dz| this thread wrote (bit pattern: non-negative floats order like unsigned integers)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long pix = idx / q;
    const int c = (int)(idx - pix * q) * 4;
    // (kernel-uniform branches; the sum is taken in fp32 whatever the storage)
    auto ldh = [&](const void* q, int stride, int bf) -> f32x4 {
      if (bf) return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(q) + pix * stride + c), f32x4);
      return ld4(reinterpret_cast<const float*>(q) + pix * stride + c);
    };
    f32x4 dh = ldh(p.dh0, p.s_dh0, p.bf_dh0);
    if (p.dh1) dh += ldh(p.dh1, p.s_dh1, p.bf_dh1);
    if (p.dh2) dh += ldh(p.dh2, p.s_dh2, p.bf_dh2);
    const TG* g = reinterpret_cast<const TG*>(p.gates) + pix * p.s_g + c;
    const f32x4 gi = ldv4(g), gf = ldv4(g + p.hidp), go = ldv4(g + 2 * p.hidp), gg = ldv4(g + 3 * p.hidp);
    f32x4 cp = {0.f, 0.f, 0.f, 0.f};
    if (p.c_prev) cp = ld4(p.c_prev + pix * p.s_cp + c);
    // c' = f c + i g: read back, or (c_new NULL) taken again from the saved gates - with bf16-stored gates one more value of the step that carries their
    // rounding (2^-9 relative, like every other use of the gates here) for 4 of the pass's 36 bytes per element
    f32x4 cn;
    if (p.c_new) cn = ld4(p.c_new + pix * p.s_cn + c);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) cn[j] = gf[j] * cp[j] + gi[j] * gg[j];
    }
    f32x4 dc = {0.f, 0.f, 0.f, 0.f};
    if (p.dc_next) dc = ld4(p.dc_next + pix * p.s_dcn + c);
    f32x4 zi, zf, zo, zg, dcp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tc = sf_tanh(cn[j]);
      const float d_o = dh[j] * tc;
      const float d_c = dc[j] + dh[j] * go[j] * (1.f - tc * tc);
      zi[j] = d_c * gg[j] * gi[j] * (1.f - gi[j]);
      zf[j] = d_c * cp[j] * gf[j] * (1.f - gf[j]);
      zo[j] = d_o * go[j] * (1.f - go[j]);
      zg[j] = d_c * gi[j] * (1.f - gg[j] * gg[j]);
      dcp[j] = d_c * gf[j];
    }
    if (p.amax) {   // kernel-uniform
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m = fmaxf(fmaxf(fabsf(zi[j]), fabsf(zf[j])), fmaxf(fabsf(zo[j]), fabsf(zg[j])));
        const unsigned b = __builtin_bit_cast(unsigned, m);
        zmax = b > zmax ? b : zmax;
      }
    }
    TG* z = reinterpret_cast<TG*>(p.dz) + pix * p.s_dz + c;
    stv4(z, zi); stv4(z + p.hidp, zf); stv4(z + 2 * p.hidp, zo); stv4(z + 3 * p.hidp, zg);
    if (p.dc_prev) *reinterpret_cast<f32x4*>(p.dc_prev + pix * p.s_dcp + c) = dcp;
  }
  if (p.amax) {   // one atomic per wave at most, and only while the word is still below this wave's maximum (the word only grows)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)zmax, off);
      zmax = o > zmax ? o : zmax;
    }
    if ((threadIdx.x & 63) == 0 && zmax > __atomic_load_n(p.amax, __ATOMIC_RELAXED)) atomicMax(p.amax, zmax);
  }
}

bool aligned4(const sfTensor& t) { return t.ptr == nullptr || ((((uintptr_t)t.ptr) & 15) == 0 && t.stride % 4 == 0 && t.dtype == SF_F32); }  // fp32 storage only
bool aligned4g0(const sfTensor& t);
bool aligned4g(const sfTensor& t) {  // gates / dz: fp32 or bf16 storage
  return ((((uintptr_t)t.ptr) & (t.dtype == SF_BF16 ? 7 : 15)) == 0 && t.stride % 4 == 0 && (t.dtype == SF_F32 || t.dtype == SF_BF16));
}

bool aligned4g0(const sfTensor& t) { return t.ptr == nullptr || aligned4g(t); }  // nullable dh source, fp32 or bf16

}  // namespace

extern "C" int sf_convlstm_cell_bwd_gates(sfTensor dh0, sfTensor dh1, sfTensor dh2, sfTensor dc_next, sfTensor gates,
                                          sfTensor c_prev, sfTensor c_new, int64_t pixels, int32_t hidp, sfTensor dz,
                                          sfTensor dc_prev, int32_t dtype, sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_convlstm_cell_bwd_gates: dtype %d not built", dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && hidp > 0, "bwd_gates: hidp=%d", hidp);
  SF_REQUIRE(dh0.ptr && gates.ptr && dz.ptr, "bwd_gates: dh0, gates, dz must be non-null");
  SF_REQUIRE(aligned4g0(dh0) && aligned4g0(dh1) && aligned4g0(dh2) && aligned4(dc_next) && aligned4g(gates) && aligned4(c_prev) &&
                 aligned4(c_new) && aligned4g(dz) && aligned4(dc_prev) && gates.dtype == dz.dtype,
             "bwd_gates: tensors must be 16-byte aligned with stride %% 4 == 0, fp32 (gates / dz: fp32 or bf16, both alike; dh sources: fp32 or bf16 each)");
  GateBwdParams p{};
  p.dh0 = dh0.ptr; p.dh1 = dh1.ptr; p.dh2 = dh2.ptr;
  p.bf_dh0 = dh0.dtype == SF_BF16; p.bf_dh1 = dh1.ptr && dh1.dtype == SF_BF16; p.bf_dh2 = dh2.ptr && dh2.dtype == SF_BF16;
  p.s_dh0 = dh0.stride; p.s_dh1 = dh1.stride; p.s_dh2 = dh2.stride;
  p.dc_next = (const float*)dc_next.ptr; p.s_dcn = dc_next.stride;
  p.gates = gates.ptr; p.s_g = gates.stride;
  p.c_prev = (const float*)c_prev.ptr; p.s_cp = c_prev.stride;
  p.c_new = (const float*)c_new.ptr; p.s_cn = c_new.stride;
  p.dz = dz.ptr; p.s_dz = dz.stride;
  p.dc_prev = (float*)dc_prev.ptr; p.s_dcp = dc_prev.stride;
  p.pixels = pixels; p.hidp = hidp;
  SF_REQUIRE(!dz.amax || (dz.dtype == SF_F32 && ((uintptr_t)dz.amax & 3) == 0), "bwd_gates: dz.amax goes with an fp32-stored dz (4-byte aligned word)");
  p.amax = (unsigned*)dz.amax;
  const long long total = pixels * (hidp / 4);
  if (total == 0) return 0;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (gates.dtype == SF_BF16) hipLaunchKernelGGL(lstm_bwd_gates_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(lstm_bwd_gates_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("lstm_bwd_gates");
  return 0;
}
