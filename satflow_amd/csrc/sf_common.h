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
