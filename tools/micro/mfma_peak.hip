// Developer probe (round 4, VERDICT r3 item 4, first step): what does the bare matrix-core loop of the split contractions sustain -
// v_mfma_f32_32x32x16_bf16 on register operands, no memory traffic at all - on ZERO operands and on RANDOM operands (clock /
// power under load)?  The product kernel's three-pass loop can never exceed this.  Whole chip, 256 threads per workgroup, 8 and
// 16 waves per CU, 8 independent accumulator chains per wave (the 128x128-tile kernel has 4 per wave and pass).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ ops, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = ops[(tid * 6 + i) & 0xffff];
#pragma unroll
  for (int i = 0; i < 2; ++i) b[i] = ops[(tid * 6 + 4 + i) & 0xffff];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a[i & 3]), *reinterpret_cast<const bf16x8*>(&b[i >> 2]), acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[tid] = s;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint4* ops; float* out;
  const size_t nops = 1 << 16;
  CK(hipMalloc(&ops, nops * 16)); CK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
  uint4* h = (uint4*)malloc(nops * 16);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int mode = 0; mode < 3; ++mode) {
    srand(11);
    for (size_t i = 0; i < nops; ++i) {
      unsigned w[4];
      for (int j = 0; j < 4; ++j) {
        // two bf16 per word: zeros / random values in [-1, 1) with random mantissas / small values (the "lo" pieces of a split operand)
        unsigned lo16 = 0, hi16 = 0;
        if (mode >= 1) {
          auto mk = [&](void) { float v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (mode == 2 ? 1.0f / 256 : 1.0f); unsigned u; memcpy(&u, &v, 4); return u >> 16; };
          lo16 = mk(); hi16 = mk();
        }
        w[j] = lo16 | (hi16 << 16);
      }
      h[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    CK(hipMemcpy(ops, h, nops * 16, hipMemcpyHostToDevice));
    for (int wg_per_cu = 2; wg_per_cu <= 4; wg_per_cu += 2) {
      const int grid = cus * wg_per_cu;
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, ops, out, 200);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, ops, out, iters);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = (double)grid * 4 /* waves */ * iters * 8 * (2.0 * 32 * 32 * 16);
      printf("%-28s %2d waves per CU: %7.1f TFLOP/s of bf16 MFMA work (%.2f ms) = %.0f TFLOP/s fp32-equivalent at three passes\n",
             mode == 0 ? "zero operands" : (mode == 1 ? "random operands in [-1, 1)" : "random operands, |x| < 2^-8"), wg_per_cu * 4, flop / ms / 1e9, ms, flop / ms / 1e9 / 3);
    }
  }
  return 0;
}
