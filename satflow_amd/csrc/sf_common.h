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

// ---- MFMA C/D fragment geometry (32x32 tile, dtype independent on gfx950) --------------------
// acc[reg] of lane l holds C[row][col] with col = l & 31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
__device__ __forceinline__ int frag_row(int reg, int lane_hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * lane_hi; }

__device__ __forceinline__ float sf_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- storage-typed 4-channel access (fp32 or bf16 activations) ---------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ldv4(const T* p);
template <> __device__ __forceinline__ f32x4 ldv4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ldv4<__bf16>(const __bf16* p) {
  return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
}
template <typename T> __device__ __forceinline__ void stv4(T* p, f32x4 v);
template <> __device__ __forceinline__ void stv4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void stv4<__bf16>(__bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }

#define SF_F32_ONLY(t, name) SF_REQUIRE((t).ptr == nullptr || (t).dtype == SF_F32, "%s: bf16 storage is not supported by this entry point", name)
