// Pointwise half of the ConvGRU cell backward (upstream metnet ConvGRUCell):
//   h' = (1-z)*n + z*h,  n = tanh(gx_n + r*h2),  z = sigmoid(gx_z + hz), r = sigmoid(gx_r + hr)
// From dh' and the saved (z, r, n, h2) produce the gradient wrt the pre-activations:
//   dgx = [da_z | da_r | da_n]   (x-part conv outputs; also the h-part's z/r pre-activations)
//   dgh = [da_z | da_r | dh2]    (h-part conv outputs)
//   dh_direct = dh' * z          (the blend's direct path to the previous state)
// HBM-bound streaming kernel, 16-byte accesses.
#include "sf_common.h"

namespace {

struct GruBwdParams {
  const float* dh0; const float* dh1; const float* dh2; int s0, s1, s2;
  const void* gates; int s_g;
  const float* h_prev; int s_hp;
  void* dgx; int s_dgx;
  void* dgh; int s_dgh;
  float* dh_direct; int s_dd;
  long long pixels; int hidp;
  unsigned* amax_gx; unsigned* amax_gh;   // (nullable; SF_F32E: dgx.amax / dgh.amax) raised to max |dgx| / max |dgh| of this launch; reset by the caller
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// TG: storage of the saved gates, TD: storage of dgx / dgh (fp32, or bf16 in "bf16a" mode: the gates are backward-only data
// and the two gradient tensors are only ever read as bf16 MFMA operands by the convolutions behind them)
template <typename TG, typename TD>
__global__ __launch_bounds__(256) void gru_bwd_gates_kernel(const GruBwdParams p) {
  const int q = p.hidp >> 2;
  const long long total = p.pixels * q;
  float mx = 0.f, mh = 0.f;   // largest |dgx| / |dgh| element this thread wrote
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long pix = idx / q;
    const int c = (int)(idx - pix * q) * 4;
    f32x4 dh = ld4(p.dh0 + pix * p.s0 + c);
    if (p.dh1) dh += ld4(p.dh1 + pix * p.s1 + c);
    if (p.dh2) dh += ld4(p.dh2 + pix * p.s2 + c);
    const TG* g = reinterpret_cast<const TG*>(p.gates) + pix * p.s_g + c;
    const f32x4 z = ldv4(g), r = ldv4(g + p.hidp), n = ldv4(g + 2 * p.hidp), h2 = ldv4(g + 3 * p.hidp);
    f32x4 hp = {0.f, 0.f, 0.f, 0.f};
    if (p.h_prev) hp = ld4(p.h_prev + pix * p.s_hp + c);
    f32x4 az, ar, an, d2, dd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const sfGruBwd o = sf_gru_bwd(dh[j], z[j], r[j], n[j], h2[j], hp[j]);
      az[j] = o.az; ar[j] = o.ar; an[j] = o.an; d2[j] = o.d2; dd[j] = o.dd;
    }
    if (p.amax_gx || p.amax_gh) {   // kernel-uniform
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float zr = fmaxf(fabsf(az[j]), fabsf(ar[j]));
        mx = fmaxf(mx, fmaxf(zr, fabsf(an[j])));
        mh = fmaxf(mh, fmaxf(zr, fabsf(d2[j])));
      }
    }
    TD* a = reinterpret_cast<TD*>(p.dgx) + pix * p.s_dgx + c;
    stv4(a, az); stv4(a + p.hidp, ar); stv4(a + 2 * p.hidp, an);
    TD* b = reinterpret_cast<TD*>(p.dgh) + pix * p.s_dgh + c;
    stv4(b, az); stv4(b + p.hidp, ar); stv4(b + 2 * p.hidp, d2);
    if (p.dh_direct) st4(p.dh_direct + pix * p.s_dd + c, dd);
  }
  if (p.amax_gx || p.amax_gh) {   // at most one atomic maximum per wave and word, skipped once the word covers it (lstm_pointwise.hip)
    unsigned bx = __builtin_bit_cast(unsigned, mx), bh = __builtin_bit_cast(unsigned, mh);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned ox = (unsigned)__shfl_xor((int)bx, off), oh = (unsigned)__shfl_xor((int)bh, off);
      bx = ox > bx ? ox : bx; bh = oh > bh ? oh : bh;
    }
    if ((threadIdx.x & 63) == 0) {
      if (p.amax_gx && bx > __atomic_load_n(p.amax_gx, __ATOMIC_RELAXED)) atomicMax(p.amax_gx, bx);
      if (p.amax_gh && bh > __atomic_load_n(p.amax_gh, __ATOMIC_RELAXED)) atomicMax(p.amax_gh, bh);
    }
  }
}

bool ok4(const sfTensor& t) { return t.ptr == nullptr || ((((uintptr_t)t.ptr) & 15) == 0 && t.stride % 4 == 0 && t.dtype == SF_F32); }  // fp32 storage only
bool ok4s(const sfTensor& t) {  // fp32 or bf16 storage, channel quads on 8- / 16-byte boundaries
  return t.ptr == nullptr || ((t.dtype == SF_F32 || t.dtype == SF_BF16) && (((uintptr_t)t.ptr) & (t.dtype == SF_BF16 ? 7 : 15)) == 0 && t.stride % 4 == 0);
}

}  // namespace

extern "C" int sf_convgru_bwd_gates(sfTensor dh0, sfTensor dh1, sfTensor dh2, sfTensor gates, sfTensor h_prev, int64_t pixels,
                                    int32_t hidp, sfTensor dgx, sfTensor dgh, sfTensor dh_direct, int32_t dtype,
                                    sfStream stream) {
  SF_REQUIRE(dtype == SF_F32, "sf_convgru_bwd_gates: dtype %d not built", dtype);
  SF_REQUIRE(hidp % SF_CPAD == 0 && hidp > 0, "gru bwd_gates: hidp=%d", hidp);
  SF_REQUIRE(dh0.ptr && gates.ptr && dgx.ptr && dgh.ptr, "gru bwd_gates: dh0, gates, dgx, dgh must be non-null");
  SF_REQUIRE(ok4(dh0) && ok4(dh1) && ok4(dh2) && ok4s(gates) && ok4(h_prev) && ok4s(dgx) && ok4s(dgh) && ok4(dh_direct) && dgx.dtype == dgh.dtype,
             "gru bwd_gates: tensors must be 16-byte aligned with stride %% 4 == 0 (fp32; gates and dgx / dgh may be bf16-stored, dgx and dgh alike)");
  GruBwdParams p{};
  p.dh0 = (const float*)dh0.ptr; p.dh1 = (const float*)dh1.ptr; p.dh2 = (const float*)dh2.ptr;
  p.s0 = dh0.stride; p.s1 = dh1.stride; p.s2 = dh2.stride;
  p.gates = gates.ptr; p.s_g = gates.stride;
  p.h_prev = (const float*)h_prev.ptr; p.s_hp = h_prev.stride;
  p.dgx = dgx.ptr; p.s_dgx = dgx.stride;
  p.dgh = dgh.ptr; p.s_dgh = dgh.stride;
  p.dh_direct = (float*)dh_direct.ptr; p.s_dd = dh_direct.stride;
  p.pixels = pixels; p.hidp = hidp;
  SF_REQUIRE((!dgx.amax && !dgh.amax) || (dgx.dtype == SF_F32 && (((uintptr_t)dgx.amax | (uintptr_t)dgh.amax) & 3) == 0),
             "gru bwd_gates: dgx.amax / dgh.amax go with fp32-stored gradients (4-byte aligned words)");
  p.amax_gx = (unsigned*)dgx.amax; p.amax_gh = (unsigned*)dgh.amax;
  const long long total = pixels * (hidp / 4);
  if (total == 0) return 0;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  const bool gb = gates.dtype == SF_BF16, db = dgx.dtype == SF_BF16;
  if (gb && db) hipLaunchKernelGGL((gru_bwd_gates_kernel<__bf16, __bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else if (gb) hipLaunchKernelGGL((gru_bwd_gates_kernel<__bf16, float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else if (db) hipLaunchKernelGGL((gru_bwd_gates_kernel<float, __bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gru_bwd_gates_kernel<float, float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  SF_CHECK_LAUNCH("gru_bwd_gates");
  return 0;
}
