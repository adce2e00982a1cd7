// Shared device/host helpers for libsatflow_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/satflow_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing -------------------------------------------------------------------------
void sf_set_error(const char* fmt, ...);

#define SF_REQUIRE(cond, ...)  \
  do {                         \
    if (!(cond)) {             \
      sf_set_error(__VA_ARGS__); \
      return 1;                \
    }                          \
  } while (0)

#define SF_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      sf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return 2;                                                            \
    }                                                                      \
  } while (0)

// Byte fill of device memory on a stream, implemented as a kernel (error.hip): use this, NOT hipMemsetAsync, wherever the call may be
// captured into a hipGraph - memset nodes replayed a wrong pattern from the second graph launch on (ROCm 7.2 / gfx950).
hipError_t sf_fill_async(void* ptr, int value, size_t bytes, hipStream_t st);

// ---- MFMA C/D fragment geometry (32x32 tile, dtype independent on gfx950) --------------------
// acc[reg] of lane l holds C[row][col] with col = l & 31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
__device__ __forceinline__ int frag_row(int reg, int lane_hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * lane_hi; }

// Gate nonlinearities on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): absolute error ~1e-7, two orders
// below the parity gate, for 4 instructions instead of the ~25 (sigmoid) / ~40 (tanh) of the libm forms - the fused
// cells evaluate 5 of them per output element in their epilogue.
__device__ __forceinline__ float sf_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
}
__device__ __forceinline__ float sf_tanh(float v) {
  const float t = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(v));  // e^(-2|v|) in (0, 1]
  return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), v);
}

// ConvGRU blend and candidate pre-activation with EXPLICIT fused multiply-adds: the per-step kernel (conv_common.h) and the
// persistent sequence kernel (convgru_seq.hip) must round identically - left to hipcc's contraction the two epilogues differed by
// one fp32 ulp here and there, which flips a bf16 rounding of the next step's MFMA operand now and then (3.7e-4 on the states).
//   h' = (1 - z) * n + z * h  ==  n + z * (h - n)
__device__ __forceinline__ float sf_gru_blend(float z, float cand, float hp) { return __builtin_fmaf(z, hp - cand, cand); }
__device__ __forceinline__ float sf_gru_cand_arg(float gn, float rg, float h2) { return __builtin_fmaf(rg, h2, gn); }
// Gate backward of one element, shared by gru_bwd_gates_kernel and the persistent backward kernel (convgru_seq.hip) with a fixed
// operation order (no contraction): the two must produce the same bf16 MFMA operands for the recurrent input-gradient convolution.
//   in: dh' and the saved z, r, n (candidate), h2, previous state hp
//   out: az, ar (pre-activation gradients of z, r), an (of the candidate's argument), d2 = d(h2), dd = dh' * z (direct path to hp)
struct sfGruBwd { float az, ar, an, d2, dd; };
__device__ __forceinline__ sfGruBwd sf_gru_bwd(float dh, float z, float r, float n, float h2, float hp) {
#pragma clang fp contract(off)
  sfGruBwd o;
  const float dn = dh * (1.f - z);
  const float dz = dh * (hp - n);
  o.an = dn * (1.f - n * n);
  o.ar = o.an * h2 * r * (1.f - r);
  o.d2 = o.an * r;
  o.az = dz * z * (1.f - z);
  o.dd = dh * z;
  return o;
}

// ---- storage-typed 4-channel access (fp32 or bf16 activations) ---------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ldv4(const T* p);
template <> __device__ __forceinline__ f32x4 ldv4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ldv4<__bf16>(const __bf16* p) {
  return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
}
// the value a storage type keeps of v (what stv4 writes, widened again)
template <typename T> __device__ __forceinline__ f32x4 rnd4(f32x4 v);
template <> __device__ __forceinline__ f32x4 rnd4<float>(f32x4 v) { return v; }
template <> __device__ __forceinline__ f32x4 rnd4<__bf16>(f32x4 v) { return __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4); }
template <typename T> __device__ __forceinline__ void stv4(T* p, f32x4 v);
template <> __device__ __forceinline__ void stv4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void stv4<__bf16>(__bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }

// 8-channel forms (one 16-byte access for bf16 storage)
typedef float f32x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ f32x8_t ldv8(const T* p);
template <> __device__ __forceinline__ f32x8_t ldv8<float>(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  return f32x8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
template <> __device__ __forceinline__ f32x8_t ldv8<__bf16>(const __bf16* p) {
  return __builtin_convertvector(*reinterpret_cast<const bf16x8_t*>(p), f32x8_t);
}
template <typename T> __device__ __forceinline__ void stv8(T* p, f32x8_t v);
template <> __device__ __forceinline__ void stv8<float>(float* p, f32x8_t v) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void stv8<__bf16>(__bf16* p, f32x8_t v) { *reinterpret_cast<bf16x8_t*>(p) = __builtin_convertvector(v, bf16x8_t); }

#define SF_F32_ONLY(t, name) SF_REQUIRE((t).ptr == nullptr || (t).dtype == SF_F32, "%s: bf16 storage is not supported by this entry point", name)

// ---- counter-based dropout masks ---------------------------------------------------------------------------------
// Elements are masked in groups of 4 consecutive indices (a channel quad): two 32-bit hashes give four 16-bit uniforms.
// keep-scale of element 4*g + j under probability p:  u16 < thr(p) ? 0 : 1/(1-p),  thr = round(p * 65536).
struct sfDrop { float p1, p2; unsigned thr1, thr2; float k1, k2; unsigned s1lo, s1hi, s2lo, s2hi; };
__host__ inline sfDrop sf_make_drop(float p1, float p2, unsigned long long seed1, unsigned long long seed2) {
  sfDrop d;
  d.p1 = p1; d.p2 = p2;
  d.thr1 = (unsigned)(p1 * 65536.f + 0.5f); d.thr2 = (unsigned)(p2 * 65536.f + 0.5f);
  d.k1 = p1 > 0.f ? 1.f / (1.f - p1) : 1.f; d.k2 = p2 > 0.f ? 1.f / (1.f - p2) : 1.f;
  d.s1lo = (unsigned)seed1; d.s1hi = (unsigned)(seed1 >> 32); d.s2lo = (unsigned)seed2; d.s2hi = (unsigned)(seed2 >> 32);
  return d;
}
__device__ __forceinline__ unsigned sf_hash32(unsigned slo, unsigned shi, unsigned long long ctr) {
  unsigned h = ((unsigned)ctr * 0x9E3779B1u) ^ slo;
  h += ((unsigned)(ctr >> 32) ^ shi) * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;  // murmur3 finaliser
  return h;
}
// scales of the 4 elements of group g (mask 1) times those of group g2 (mask 2, the period-reduced index)
__device__ __forceinline__ f32x4 sf_drop_scales(const sfDrop& d, unsigned long long g, unsigned long long g2) {
  f32x4 s = {1.f, 1.f, 1.f, 1.f};
  if (d.p1 > 0.f) {
    const unsigned a = sf_hash32(d.s1lo, d.s1hi, 2 * g), b = sf_hash32(d.s1lo, d.s1hi, 2 * g + 1);
    s[0] = (a & 0xffffu) < d.thr1 ? 0.f : d.k1; s[1] = (a >> 16) < d.thr1 ? 0.f : d.k1;
    s[2] = (b & 0xffffu) < d.thr1 ? 0.f : d.k1; s[3] = (b >> 16) < d.thr1 ? 0.f : d.k1;
  }
  if (d.p2 > 0.f) {
    const unsigned a = sf_hash32(d.s2lo, d.s2hi, 2 * g2), b = sf_hash32(d.s2lo, d.s2hi, 2 * g2 + 1);
    s[0] *= (a & 0xffffu) < d.thr2 ? 0.f : d.k2; s[1] *= (a >> 16) < d.thr2 ? 0.f : d.k2;
    s[2] *= (b & 0xffffu) < d.thr2 ? 0.f : d.k2; s[3] *= (b >> 16) < d.thr2 ? 0.f : d.k2;
  }
  return s;
}
