// Shared helpers of libsopro_hip (gfx950 only).
#pragma once
// Developer A/B switches whose measurement is on record (profiles/rNN_experiments.md) are COMPILE-TIME (round 5 housekeeping): a
// product build reads none of them.  `make DEV=1` builds libsopro_hip_dev.so (-DSOPRO_DEV_SWITCHES), which reads them from the
// environment again (load it with SOPRO_HIP_LIB=.../libsopro_hip_dev.so for a re-measurement).
#ifdef SOPRO_DEV_SWITCHES
#define SOPRO_DEV_ENV(name) getenv(name)
#else
#define SOPRO_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sopro_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void sopro_set_error(const char* fmt, ...);
// Floor of the dynamic LDS request of the long-running contraction kernels (sopro_set_lds_floor): above half of a CU's
// 160 KB it caps them at ONE workgroup per CU, which leaves registers, wave slots and LDS on every CU for the short kernels
// of a concurrently generating AR frame (co-scheduling without CU masks).
extern int g_sopro_lds_floor;

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

// hipFuncAttributeMaxDynamicSharedMemorySize must be set per device: `done` is one flag byte per device of the calling site (a
// process may drive several GPUs; two threads racing on the first call both set the attribute, which is harmless).
#define SOPRO_SET_MAX_LDS_ONCE(kern, bytes)                                                                              \
  do {                                                                                                                   \
    static unsigned char done_[64] = {};                                                                                 \
    int dev_ = 0;                                                                                                        \
    SOPRO_HIP(hipGetDevice(&dev_));                                                                                      \
    if (!done_[dev_ & 63]) {                                                                                             \
      SOPRO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      done_[dev_ & 63] = 1;                                                                                              \
    }                                                                                                                    \
  } while (0)

__device__ __forceinline__ float gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }
// GELU (erf form) for the contraction epilogues of the throughput phases (round 6): erf by Abramowitz-Stegun 7.1.26 - one reciprocal, a
// degree-5 polynomial, one exp2 - 22 issue slots against the library erff's 41 (a branchy two-range expansion: both ranges run in a
// divergent wave).  |gelu_fast - gelu| <= 4.7e-7 absolute, 2.9e-7 |v| relative (float64 reference; torch's own fp32 GELU is 1.1e-6 off it
// at |v| ~ 8): the class of one fp32 rounding of the result, and below what the operand split keeps (2^-17 bf16x3, 2^-22 f16x3).  The
// 128 x 128 x 384 feed-forward tile of the refinement spent more vector issue on the epilogue's erff (64 elements per thread) than its
// twelve K-steps spend on the matrix cores.  The exact-fp32 paths (AR frame, conditioning, six-pass) keep erff.
__device__ __forceinline__ float gelu_fast(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(x * x * -1.44269504088896340736f);
  const float y = fmaf(-p, e, 1.0f);  // erf(|v| / sqrt 2)
  const float h = 0.5f * v;
  return fmaf(h, copysignf(y, v), h);
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
// Round 6: the GLU gate of the f16 three-pass / one-pass contraction epilogues (the refinement's GLU projection, 24 launches per pass): the
// hardware exponential and reciprocal (1 ulp each; <= ~3 ulp relative against the precise form's <= 1) - 5 issue slots per element
// instead of ~27 (software expf + IEEE division), whose dependency chains were 8.3 k of the GLU launch's 44.5 k cycles per tile
// (profiles/r06_tile_life_*.txt).  Large |v| end where the precise form does: exp2 -> inf / 0, rcp -> 0 / 1.  The six-pass kernel (the
// refinement's fallback, the conditioning) and the exact-fp32 kernels (the AR frame) keep sigmoidf_.
__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v)); }
// ELU(alpha=1).  exp(v)-1 with the hardware exponential: absolute error <= ~2e-7 on v <= 0 (fp32 round-off of the
// surrounding contractions is larger); expm1f's software expansion was the dominant VALU cost of the SEANet tail.
// Round 6: the selection is ONE v_med3_f32 instead of a compare + select: w = exp(v) - 1 >= v everywhere (convexity), so for v > 0 the
// order is 0 < v < w and for v < 0 it is v < w <= 0 - the median of (v, w, 0) is the ELU either way (w = inf for large v: still v).
// In fp32 the computed w can fall below a small positive v by the exponential's round-off (v < 6e-8: w = 0): the result is then w
// instead of v, an absolute error <= ~2e-7 - the same class as the negative side's, far inside the waveform contract (1e-4 of peak).
// 7 issue slots per element instead of 8 (v_mul, v_exp_f32 at quarter rate = 4, v_add, v_med3): the fused SEANet kernels are bound by
// their vector issue and ELU is 44 % of it (profiles/r04_experiments.md section 5).
__device__ __forceinline__ float eluf_(float v) { return __builtin_amdgcn_fmed3f(v, __expf(v) - 1.0f, 0.0f); }

// fp32 -> two bf16 halves, x = hi + lo with hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even: |lo| <= 2^-9 |x|
// with either sign, so products that drop lo*lo are 2^-18-relative and zero-mean (a truncated hi: 2^-14, one-signed).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_bf16(float x, float y, unsigned& hi, unsigned& lo) {
  const f32x2_t v = {x, y};
  const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
  hi = *reinterpret_cast<const unsigned*>(&h);
  const f32x2_t r = {x - __uint_as_float(hi << 16), y - __uint_as_float(hi & 0xffff0000u)};  // exact
  const bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
  lo = *reinterpret_cast<const unsigned*>(&l);
}

// The throughput phases' output streams - every contraction's C / activated copy, the 128-channel residual block's output - and
// the fused last level's input carry the non-temporal hint: they then displace less of what the generation partition next door
// re-reads every frame (its weights and folded text operands, ~240 MB).  Measured in the pipeline (profiles/r04_experiments.md
// section 7, same box, alternating): never 23.45 / 23.66 k audio-s/s, only tensors >= 64 MB 23.65 / 23.74 k, ALWAYS 23.87 / 23.88 k
// (AR phases 33.3-33.9 against 34.5-34.9 ms per step, Mimi decode 13.5 against 13.8-13.9).  Policy SOPRO_NT_BULK: 1 = always (the
// product), 2 = by size (`big`), 0 = never (developer A/B: tools/micro/build_nt_bulk.sh).
#ifndef SOPRO_NT_BULK
#define SOPRO_NT_BULK 1
#endif
__device__ __forceinline__ bool bulk_nt(bool big) { return SOPRO_NT_BULK == 1 || (SOPRO_NT_BULK == 2 && big); }
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bulk_store4(float* p, const float4& v, bool big) {
  if (bulk_nt(big)) __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ void bulk_store1(float* p, float v, bool big) {
  if (bulk_nt(big)) __builtin_nontemporal_store(v, p);
  else *p = v;
}
__device__ __forceinline__ void bulk_store_u2(void* p, const uint2& v, bool big) {
  if (bulk_nt(big)) __builtin_nontemporal_store((u32x2_t){v.x, v.y}, reinterpret_cast<u32x2_t*>(p));
  else *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void bulk_store_u1(void* p, unsigned v, bool big) {
  if (bulk_nt(big)) __builtin_nontemporal_store(v, reinterpret_cast<unsigned*>(p));
  else *reinterpret_cast<unsigned*>(p) = v;
}
__device__ __forceinline__ uint4 bulk_load16(const void* p, bool big) {
  if (bulk_nt(big)) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const uint4*>(p);
}
constexpr int64_t SOPRO_BIG_BYTES = (int64_t)64 << 20;

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2.  This maps
// the workgroup index so that XCD x walks ONE contiguous range of tile indices: the tiles an L2 serves at the same time
// share their A rows / W columns instead of every L2 fetching every operand.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int per = n >> 3, rem = n & 7;
  const int x = b & 7, i = b >> 3;
  return x * per + min(x, rem) + i;
}

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

// library-side launch timing (prof.hip): a scope around one heavy launch of a stage sequence; a no-op unless sopro_prof_enable(1)
struct sopro_prof_scope {
  void* rec;
  hipStream_t s;
  sopro_prof_scope(const char* family, double flops, hipStream_t s);
  ~sopro_prof_scope();
  sopro_prof_scope(const sopro_prof_scope&) = delete;
  sopro_prof_scope& operator=(const sopro_prof_scope&) = delete;
};
