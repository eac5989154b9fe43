// Developer prototype (not part of libsopro_hip): the three-pass split-bf16 contraction with WAVE SPECIALISATION - four
// producer waves stage operands (A: fp32 rows -> bf16 hi/lo pieces, W: packed fragments) into a ring of LDS stages while four
// consumer waves run the MFMAs; the two halves meet through LDS counters (release / acquire at workgroup scope), never through
// a workgroup barrier, so a SIMD always has one wave converting / loading and one wave on the matrix core.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_ws_proto gemm_ws_proto.hip && ./gemm_ws_proto [M N K]
// 256x128 tile, FOUR consumer waves 2x2 of 128x64 each (12 KB of fragment reads per 24 MFMAs instead of 8 KB per 12) + four
// producer waves, NSTAGE LDS stages of one 32-wide K-step;
// A rows are requested PD steps ahead (HBM), W fragments one step ahead (L2).  Spins are bounded: a protocol error
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
#define NSTAGE 3
#endif
#ifndef PD
#define PD 3   // K-steps the producers' global loads run ahead (register sets = PD + 1)
#endif
constexpr int BM = 256, BN = 128, BK = 32, NTH = 512;
constexpr int NCONS = 4, NPROD = 4;  // waves per role
constexpr int AROW = 2 * 64 + 16;                      // bytes per LDS row of the split A tile
constexpr int A_STAGE = BM * AROW;                     // 18432
constexpr int B_STAGE = (BN / 32) * 2 * 2 * 64 * 16;   // 4 column tiles x 2 k-subs x 2 pieces x 64 lanes x 16 B = 16384
constexpr int STAGE = A_STAGE + B_STAGE;               // 53248
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

  if (wave >= NCONS) {
    // ------------------------------------------------------------------ producers (256 threads)
    const int pt = tid - NCONS * 64;
    const int lrow = pt >> 3, lc4 = pt & 7;  // A staging: rows lrow + 32 i, float4 column lc4
    // one base pointer per operand (M is a multiple of 128 here): row group i of A is i * 32 rows further, column tile i of W
    // is i * ksubs * 128 fragments further (idx = pt + 256 i keeps lane, piece and k-sub, and moves the column tile by i)
    const float* abase = A + (int64_t)(m0 + lrow) * lda + lc4 * 4;
    const int64_t astep = 32 * lda;
    const int bl = pt & 63, bpz = (pt >> 6) & 1, bs = (pt >> 7) & 1;
    const u32x4* bbase = Wp + (((int64_t)(n0 >> 5) * ksubs + bs) * 2 + bpz) * 64 + bl;
    const int64_t bstep = (int64_t)ksubs * 128;
    constexpr int NSET = PD + 1;
    f32x4 ra[NSET][8];
    u32x4 rb[2][4];
    // The loads are opaque to the compiler (inline asm) and waited for with COUNTED vmcnt values: left to itself hipcc drains
    // every outstanding load at the first use after the spin loop.  Issue order per step: A(kt + PD) [4], W(kt + 1) [8].
    auto gloadA = [&](int kt, int set) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float* pa = abase + i * astep + kt * BK;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[set][i]) : "v"(pa) : "memory");
      }
    };
    auto gloadB = [&](int kt, int set) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4* pb = bbase + i * bstep + (int64_t)kt * 256;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[set][i]) : "v"(pb) : "memory");
      }
    };
    static_assert((PD + 1) % 2 == 0, "the W register set index u & 1 needs an even number of A sets");
    // issue order (a W group BEFORE the A group of the same step, see gwait): A0 .. A(PD-2), W0, A(PD-1), then per step W(kt+1), A(kt+PD)
#pragma unroll
    for (int d = 0; d < PD - 1; ++d) gloadA(min(d, KT - 1), d);
    gloadB(0, 0);
    gloadA(min(PD - 1, KT - 1), PD - 1);
    // the loop is unrolled over the register sets so that every set index is a compile-time constant
    for (int kt0 = 0; kt0 < KT; kt0 += NSET)
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const int kt = kt0 + u;
      if (kt >= KT) break;
      const int set = u, bset = u & 1;
      gloadB(min(kt + 1, KT - 1), bset ^ 1);
      gloadA(min(kt + PD, KT - 1), (u + PD) % NSET);
      const int st = kt % NSTAGE, round = kt / NSTAGE;
      // the stage must have been drained `round` times by all consumer waves
      if (round > 0 && !wait_ge(empty + st, NCONS * round)) {
        if (lane == 0) atomicExch(err, 1);
        return;
      }
      // W(kt) is followed in issue order by A(kt+PD-1), W(kt+1), A(kt+PD): 20 loads may stay in flight (loads return in order, so
      // A(kt) .. A(kt+PD-2), all older than W(kt), are complete as well)
      asm volatile("s_waitcnt vmcnt(20)"
                   : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(ra[set][2]), "+v"(ra[set][3]), "+v"(ra[set][4]), "+v"(ra[set][5]),
                     "+v"(ra[set][6]), "+v"(ra[set][7]), "+v"(rb[bset][0]), "+v"(rb[bset][1]), "+v"(rb[bset][2]), "+v"(rb[bset][3])
                   :
                   : "memory");
      unsigned char* a = smem + st * STAGE + lrow * AROW + lc4 * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
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
      for (int i = 0; i < 4; ++i) b[pt + i * 256] = rb[bset][i];
      // this wave's stores are done before its arrival is visible (release), one arrival per wave
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      if (lane == 0) __hip_atomic_fetch_add(full + st, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  // -------------------------------------------------------------------- consumers (4 waves, 128x64 each)
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, fg = lane >> 5;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // one 16-wide k-sub of fragments at a time (168 registers per wave with 12 waves per CU): the stage is handed back once the
  // second half has been read; the other consumer wave of the SIMD covers the LDS latency
  bool ok = true;
  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
  for (int kt = 0; kt < KT; ++kt) {
    const int st = kt % NSTAGE, round = kt / NSTAGE;
    if (!wait_ge(full + st, NPROD * (round + 1))) {
      if (lane == 0) atomicExch(err, 2);
      ok = false;
      break;
    }
    const unsigned char* a = smem + st * STAGE + (wm * 128 + frow) * AROW + fg * 16;
    const u32x4* b = reinterpret_cast<const u32x4*>(smem + st * STAGE + A_STAGE) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 af[4][2], bf[2][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[i][p] = *reinterpret_cast<const u32x4*>(a + i * 32 * AROW + p * 64 + s * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[j][p] = b[(((wn * 2 + j) * 2 + s) * 2 + p) * 64];
      if (s == 1) {  // every fragment of the stage is in registers: hand the stage back
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        if (lane == 0) __hip_atomic_fetch_add(empty + st, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#ifndef ABL_NOMMA
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af[i][PA[q]]), as_frag(bf[j][PB[q]]), acc[i][j], 0, 0, 0);
#else
      acc[0][0][s] += __uint_as_float((af[0][0][0] ^ bf[1][1][1]) & 0x3fffffffu);
#endif
    }
  }
  (void)ok;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + frow;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mb = m0 + wm * 128 + i * 32 + 4 * fg;
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
  printf("gemm_ws3 proto 256x128, 4 consumers of 128x64 + 4 producers (stages %d) M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s fp32-equivalent  worst rel err (of |A||W|) %.2e  protocol error %d\n",
         NSTAGE, M, N, K, us, 2.0 * M * N * K / us / 1e6, worst, herr);
  return 0;
}
