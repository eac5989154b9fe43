// Shared helpers of libsopro_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sopro_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void sopro_set_error(const char* fmt, ...);

#define SOPRO_CHECK_ARG(cond, msg)                                  \
  do {                                                              \
    if (!(cond)) {                                                  \
      sopro_set_error("%s: bad argument: %s", __func__, msg);       \
      return -2;                                                    \
    }                                                               \
  } while (0)

#define SOPRO_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      sopro_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_));       \
      return -1;                                                                          \
    }                                                                                     \
  } while (0)

#define SOPRO_LAUNCH_CHECK()                                                              \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess) {                                                               \
      sopro_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e_));   \
      return -1;                                                                          \
    }                                                                                     \
    return 0;                                                                             \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
// ELU(alpha=1).  exp(v)-1 with the hardware exponential: absolute error <= ~2e-7 on v <= 0 (fp32 round-off of the
// surrounding contractions is larger); expm1f's software expansion was the dominant VALU cost of the SEANet tail.
__device__ __forceinline__ float eluf_(float v) { return v > 0.0f ? v : __expf(v) - 1.0f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
