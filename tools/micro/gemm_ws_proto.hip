// Developer prototype (not part of libsopro_hip): the three-pass split-bf16 contraction with WAVE SPECIALISATION - four
// producer waves stage operands (A: fp32 rows -> bf16 hi/lo pieces, W: packed fragments) into a ring of LDS stages while four
// consumer waves run the MFMAs; the two halves meet through LDS counters (release / acquire at workgroup scope), never through
// a workgroup barrier, so a SIMD always has one wave converting / loading and one wave on the matrix core.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_ws_proto gemm_ws_proto.hip && ./gemm_ws_proto [M N K]
// 128x128 tile, consumers 2x2 (64x64 each), NSTAGE LDS stages of one 32-wide K-step.  Spins are bounded: a protocol error
// shows up as a wrong result and a message, not as a hang.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                    \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

#ifndef NSTAGE
#define NSTAGE 4
#endif
#ifndef PD
#define PD 3   // K-steps the producers' global loads run ahead (register sets = PD + 1)
#endif
constexpr int BM = 128, BN = 128, BK = 32, NTH = 512;
constexpr int AROW = 2 * 64 + 16;                      // bytes per LDS row of the split A tile
constexpr int A_STAGE = BM * AROW;                     // 18432
constexpr int B_STAGE = (BN / 32) * 2 * 2 * 64 * 16;   // 4 column tiles x 2 k-subs x 2 pieces x 64 lanes x 16 B = 16384
constexpr int STAGE = A_STAGE + B_STAGE;               // 34816
constexpr int LDS_BYTES = NSTAGE * STAGE + 256;        // + the counters
constexpr int SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ void split_pair2(float x, float y, unsigned (&pc)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x2_t v = {x, y};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    pc[p] = __builtin_bit_cast(unsigned, h);
    x -= __uint_as_float(pc[p] << 16);
    y -= __uint_as_float(pc[p] & 0xffff0000u);
  }
}

__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int per = n >> 3, rem = n & 7;
  const int x = b & 7, i = b >> 3;
  return x * per + min(x, rem) + i;
}

// wait until *ctr >= want (workgroup-scope acquire); bounded
__device__ __forceinline__ bool wait_ge(int* ctr, int want) {
  for (int it = 0; it < SPIN_LIMIT; ++it) {
    if (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

__global__ __launch_bounds__(NTH) void gemm_ws_kernel(const float* __restrict__ A, int64_t lda, const u32x4* __restrict__ Wp,
                                                     const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M,
                                                     int N, int K, int* __restrict__ err) {
  extern __shared__ u32x4 smem4[];
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem4);
  int* full = reinterpret_cast<int*>(smem + NSTAGE * STAGE);        // full[s]: producer waves that have filled stage s (monotonic)
  int* empty = full + NSTAGE;                                         // empty[s]: consumer waves that have drained it (monotonic)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = N / BN;
  const int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int mt = bid / ntn, nt = bid % ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int KT = K / BK, ksubs = KT * 2;
  if (tid < 2 * NSTAGE) full[tid] = 0;
  __syncthreads();  // the only workgroup barrier: counters start at zero

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers (256 threads)
    const int pt = tid - 256;
    const int lrow = pt >> 3, lc4 = pt & 7;  // A staging: rows lrow + 32 i, float4 column lc4
    const float* ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ap[i] = A + (int64_t)min(m0 + lrow + i * 32, M - 1) * lda + lc4 * 4;
    // W staging: u32x4 index idx = pt + i * 256 of the 1024 of a K-step: lane = idx & 63, piece = (idx >> 6) & 1, s = (idx >> 7) & 1, j = idx >> 8
    const u32x4* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = pt + i * 256;
      const int l = idx & 63, p = (idx >> 6) & 1, s = (idx >> 7) & 1, j = idx >> 8;
      bp[i] = Wp + (((int64_t)((n0 >> 5) + j) * ksubs + s) * 2 + p) * 64 + l;
    }
    constexpr int NSET = PD + 1;
    f32x4 ra[NSET][4];
    u32x4 rb[NSET][4];
    // The loads are opaque to the compiler (inline asm) and waited for with a COUNTED vmcnt: left to itself hipcc drains every
    // outstanding load (vmcnt(0)) at the first use after the spin loop, which collapses the prefetch distance.
    auto gload = [&](int kt, int set) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* pa = ap[i] + kt * BK;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[set][i]) : "v"(pa) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4* pb = bp[i] + (int64_t)kt * 256;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[set][i]) : "v"(pb) : "memory");
      }
    };
    auto gwait = [&](int set) {  // the 8 loads of `set` are the oldest outstanding ones: PD * 8 younger loads may stay in flight
      asm volatile("s_waitcnt vmcnt(%8)"
                   : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(ra[set][2]), "+v"(ra[set][3]), "+v"(rb[set][0]), "+v"(rb[set][1]),
                     "+v"(rb[set][2]), "+v"(rb[set][3])
                   : "n"(PD * 8)
                   : "memory");
    };
#pragma unroll
    for (int d = 0; d < PD; ++d) gload(min(d, KT - 1), d);
    // the loop is unrolled over the register sets so that every set index is a compile-time constant
    for (int kt0 = 0; kt0 < KT; kt0 += NSET)
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const int kt = kt0 + u;
      if (kt >= KT) break;
      constexpr int dummy = 0;
      (void)dummy;
      const int set = u;
#ifndef ABL_NOLOAD
      gload(min(kt + PD, KT - 1), (u + PD) % NSET);
#endif
      const int st = kt % NSTAGE, round = kt / NSTAGE;
      // the stage must have been drained `round` times by all four consumer waves
      if (round > 0 && !wait_ge(empty + st, 4 * round)) {
        if (lane == 0) atomicExch(err, 1);
        return;
      }
#ifndef ABL_NOLOAD
      gwait(set);
#endif
      unsigned char* a = smem + st * STAGE + lrow * AROW + lc4 * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned c0[2], c1[2];
#ifdef ABL_NOSPLIT
        c0[0] = __float_as_uint(ra[set][i][0]); c0[1] = __float_as_uint(ra[set][i][1]);
        c1[0] = __float_as_uint(ra[set][i][2]); c1[1] = __float_as_uint(ra[set][i][3]);
#else
        split_pair2(ra[set][i][0], ra[set][i][1], c0);
        split_pair2(ra[set][i][2], ra[set][i][3], c1);
#endif
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x2*>(a + i * 32 * AROW + p * 64) = (u32x2){c0[p], c1[p]};
      }
      u32x4* b = reinterpret_cast<u32x4*>(smem + st * STAGE + A_STAGE);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[pt + i * 256] = rb[set][i];
      // this wave's stores are done before its arrival is visible (release), one arrival per wave
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      if (lane == 0) __hip_atomic_fetch_add(full + st, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  // -------------------------------------------------------------------- consumers (4 waves, 64x64 each)
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, fg = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragments of K-step kt+1 are requested from LDS before the MFMAs of K-step kt are issued (two register sets)
#ifdef CONSUMER_PREFETCH
  constexpr int CSETS = 2;
#else
  constexpr int CSETS = 1;
#endif
  u32x4 af[CSETS][2][2][2], bf[CSETS][2][2][2];  // [set][s][i or j][piece]
  bool ok = true;
  auto fetch = [&](int kt, int set) {
    const int st = kt % NSTAGE, round = kt / NSTAGE;
    if (!wait_ge(full + st, 4 * (round + 1))) {
      if (lane == 0) atomicExch(err, 2);
      ok = false;
      return;
    }
    const unsigned char* a = smem + st * STAGE + (wm * 64 + frow) * AROW + fg * 16;
    const u32x4* b = reinterpret_cast<const u32x4*>(smem + st * STAGE + A_STAGE) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          af[set][s][i][p] = *reinterpret_cast<const u32x4*>(a + i * 32 * AROW + p * 64 + s * 32);
          bf[set][s][i][p] = b[(((wn * 2 + i) * 2 + s) * 2 + p) * 64];
        }
  };
  auto release = [&](int kt) {  // every fragment of the stage is in registers: hand the stage back
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane == 0) __hip_atomic_fetch_add(empty + (kt % NSTAGE), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto mma = [&](int set) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af[set][s][i][PA[q]]), as_frag(bf[set][s][j][PB[q]]), acc[i][j], 0, 0, 0);
  };
#ifdef CONSUMER_PREFETCH   // fragments of K-step kt+1 requested before the MFMAs of kt (two register sets): measured, no gain
  fetch(0, 0);
  release(0);
  for (int kt = 0; kt < KT && ok; kt += 2) {
    if (kt + 1 < KT) fetch(kt + 1, 1);
    mma(0);
    if (kt + 1 >= KT || !ok) break;
    release(kt + 1);
    if (kt + 2 < KT) fetch(kt + 2, 0);
    mma(1);
    if (kt + 2 < KT && ok) release(kt + 2);
  }
#else
  for (int kt = 0; kt < KT && ok; ++kt) {
    fetch(kt, 0);
    if (!ok) break;
    release(kt);
#ifdef ABL_NOMMA
    acc[0][0][0] += __uint_as_float((af[0][0][0][0][0] ^ bf[0][1][1][1][1]) & 0x3fffffffu);
#else
    mma(0);
#endif
  }
#endif
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + frow;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mb = m0 + wm * 64 + i * 32 + 4 * fg;
      float* cp = C + (int64_t)mb * ldc + n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2);
        if (mb + dm < M) cp[(int64_t)dm * ldc] = acc[i][j][r] + bv;
      }
    }
  }
}

static uint16_t bf16_rn(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_to_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 2048;
  if (N % BN || K % BK) {
    printf("N %% 128 == 0 and K %% 32 == 0 expected\n");
    return 1;
  }
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
  uint32_t st = 12345u;
  auto rnd = [&]() {
    st = st * 1664525u + 1013904223u;
    return ((st >> 8) & 0xffff) / 32768.0f - 1.0f;
  };
  for (auto& v : hA) v = rnd();
  const float ws = 1.0f / sqrtf((float)K);
  for (auto& v : hW) v = rnd() * ws;
  for (auto& v : hb) v = rnd();
  const int ksubs = K / 16;
  std::vector<uint16_t> hWp((size_t)(N / 32) * ksubs * 2 * 64 * 8);
  for (int t = 0; t < N / 32; ++t)
    for (int sub = 0; sub < ksubs; ++sub)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = t * 32 + (lane & 31), k0 = sub * 16 + (lane >> 5) * 8;
        for (int e = 0; e < 8; ++e) {
          const float x = hW[(size_t)n * K + k0 + e];
          const uint16_t hi = bf16_rn(x), lo = bf16_rn(x - bf16_to_f(hi));
          hWp[((((size_t)t * ksubs + sub) * 2 + 0) * 64 + lane) * 8 + e] = hi;
          hWp[((((size_t)t * ksubs + sub) * 2 + 1) * 64 + lane) * 8 + e] = lo;
        }
      }
  float *dA, *dC, *db;
  u32x4* dW;
  int* derr;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&db, N * 4));
  CK(hipMalloc(&dW, hWp.size() * 2));
  CK(hipMalloc(&derr, 4));
  CK(hipMemset(derr, 0, 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int tiles = ((M + BM - 1) / BM) * (N / BN);
  auto launch = [&]() {
    hipLaunchKernelGGL(gemm_ws_kernel, dim3(tiles), dim3(NTH), LDS_BYTES, 0, dA, (int64_t)K, dW, db, dC, (int64_t)N, M, N, K, derr);
  };
  launch();
  CK(hipDeviceSynchronize());
  int herr = 0;
  CK(hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost));
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int sidx = 0; sidx < 2000; ++sidx) {
    st = st * 1664525u + 1013904223u;
    const int m = (int)((st >> 8) % (uint32_t)M);
    st = st * 1664525u + 1013904223u;
    const int n = (int)((st >> 8) % (uint32_t)N);
    double ref = hb[n], mag = fabs(hb[n]);
    for (int k = 0; k < K; ++k) {
      const double p = (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
      ref += p;
      mag += fabs(p);
    }
    const double e = fabs((double)hC[(size_t)m * N + n] - ref) / mag;
    if (e > worst) worst = e;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms / 5 * 1e3;
  printf("gemm_ws proto (stages %d) M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s fp32-equivalent  worst rel err (of |A||W|) %.2e  protocol error %d\n",
         NSTAGE, M, N, K, us, 2.0 * M * N * K / us / 1e6, worst, herr);
  return 0;
}
