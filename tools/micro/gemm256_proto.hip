// Developer prototype (not part of libsopro_hip): the three-pass split-bf16 contraction on a 256x256 tile with 8 waves and
// BOTH operands staged through LDS (W fragments shared by the waves of a column, no duplicate W loads): how fast is that
// main loop next to the product kernel's 128x128 / W-in-registers loop?  Epilogue = plain stores (bias only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm256_proto gemm256_proto.hip && ./gemm256_proto [M N K]
// Operand bytes per flop are a third of the product kernel's (64 KB per K-step for 4x the flops of a 128x128 tile).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (whole-struct uint4 copies keep arrays in scratch memory)

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                    \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

#ifndef PROTO_BM
#define PROTO_BM 256
#endif
constexpr int BM = PROTO_BM, BN = 256, BK = 32, NTH = 512;
constexpr int TM = BM / 64;    // 32-row groups per wave (2 wave rows)
constexpr int AF4 = BM / 64;   // float4 per staging thread
constexpr int AROW = 2 * 64 + 16;            // bytes per LDS row of the split A tile: [32 hi | 32 lo] bf16 + pad
constexpr int A_STAGE = BM * AROW;           // 36864
constexpr int B_STAGE = (BN / 32) * 2 * 2 * 64 * 16;  // 8 column tiles x 2 k-subs x 2 pieces x 64 lanes x 16 B = 32768
constexpr int LDS_BYTES = 2 * A_STAGE + 2 * B_STAGE;  // 139264

__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ void split_pair2(float x, float y, unsigned (&pc)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x2_t v = {x, y};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    pc[p] = *reinterpret_cast<const unsigned*>(&h);
    x -= __uint_as_float(pc[p] << 16);
    y -= __uint_as_float(pc[p] & 0xffff0000u);
  }
}

__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int per = n >> 3, rem = n & 7;
  const int x = b & 7, i = b >> 3;
  return x * per + min(x, rem) + i;
}

// Wp: [n/32][k/16][piece][lane][8 bf16] (the product library's sopro_pack_w_bf16 order, pieces = 2)
__global__ __launch_bounds__(NTH) void gemm256_kernel(const float* __restrict__ A, int64_t lda, const uint4* __restrict__ Wp,
                                                      const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M,
                                                      int N, int K) {
  extern __shared__ uint4 smem4[];
  unsigned char* smem = reinterpret_cast<unsigned char*>(smem4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;  // wave tile: 128 rows x 64 columns
  const int ntn = N / BN;
  const int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int mt = bid / ntn, nt = bid % ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lrow = tid >> 3, lc4 = tid & 7;
  const int KT = K / BK, ksubs = KT * 2;

  const float* ap[AF4];
#pragma unroll
  for (int i = 0; i < AF4; ++i) ap[i] = A + (int64_t)min(m0 + lrow + i * 64, M - 1) * lda + lc4 * 4;
  // W staging: uint4 index idx = tid + i * 512 of the 2048 of a K-step: lane = idx & 63, piece = (idx >> 6) & 1, s = (idx >> 7) & 1,
  // column tile j = idx >> 8
  const u32x4* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * NTH;
    const int l = idx & 63, p = (idx >> 6) & 1, s = (idx >> 7) & 1, j = idx >> 8;
    bp[i] = reinterpret_cast<const u32x4*>(Wp) + (((int64_t)((n0 >> 5) + j) * ksubs + s) * 2 + p) * 64 + l;
  }

  float4 ra[AF4];
  u32x4 rbs[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < AF4; ++i) ra[i] = *reinterpret_cast<const float4*>(ap[i] + kt * BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) rbs[i] = bp[i][(int64_t)kt * 2 * 2 * 64];
  };
  auto lstore = [&](int buf) {
    unsigned char* a = smem + buf * A_STAGE + lrow * AROW + lc4 * 8;
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      unsigned c0[2], c1[2];
      split_pair2(ra[i].x, ra[i].y, c0);
      split_pair2(ra[i].z, ra[i].w, c1);
#pragma unroll
      for (int p = 0; p < 2; ++p) *reinterpret_cast<uint2*>(a + i * 64 * AROW + p * 64) = make_uint2(c0[p], c1[p]);
    }
    u32x4* b = reinterpret_cast<u32x4*>(smem + 2 * A_STAGE + buf * B_STAGE);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[tid + i * NTH] = rbs[i];
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fg = lane >> 5;
  auto compute = [&](int buf) {
    const unsigned char* a = smem + buf * A_STAGE + (wm * (BM / 2) + frow) * AROW + fg * 16;
    const u32x4* b = reinterpret_cast<const u32x4*>(smem + 2 * A_STAGE + buf * B_STAGE) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 bf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[j][p] = b[(((wn * 2 + j) * 2 + s) * 2 + p) * 64];
      // one 32-row group of A fragments at a time (register budget: 128 accumulator registers per lane);
      // (A piece, W piece): (lo, hi), (hi, lo), (hi, hi) - smallest terms first
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        u32x4 af[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const u32x4*>(a + i * 32 * AROW + p * 64 + s * 32);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af[PA[q]]), as_frag(bf[j][PB[q]]), acc[i][j], 0, 0, 0);
      }
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < KT; kt += 2) {
    gload(min(kt + 1, KT - 1));
    compute(0);
    lstore(1);
    __syncthreads();
    if (kt + 1 >= KT) break;
    gload(min(kt + 2, KT - 1));
    compute(1);
    lstore(0);
    __syncthreads();
  }

  // plain epilogue: lane holds column (lane & 31) and 16 rows of each 32x32 accumulator
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + frow;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * fg;
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
    printf("N %% 256 == 0 and K %% 32 == 0 expected\n");
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
  uint4* dW;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&db, N * 4));
  CK(hipMalloc(&dW, hWp.size() * 2));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int tiles = ((M + BM - 1) / BM) * (N / BN);
  auto launch = [&]() {
    hipLaunchKernelGGL(gemm256_kernel, dim3(tiles), dim3(NTH), LDS_BYTES, 0, dA, (int64_t)K, dW, db, dC, (int64_t)N, M, N, K);
  };
  launch();
  CK(hipDeviceSynchronize());
  // sampled check against fp64
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
  printf("gemm256 proto M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s fp32-equivalent  worst rel err (of |A||W|) %.2e  tiles %d\n", M, N, K, us,
         2.0 * M * N * K / us / 1e6, worst, tiles);
  return 0;
}
