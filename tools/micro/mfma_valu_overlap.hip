// Developer probe: does vector work issued behind a matrix-core instruction run under it?  One workgroup of eight waves per CU (two
// per SIMD, as in seanet_uptail.hip); every wave loops over [1 x v_mfma_f32_32x32x16_bf16 (a dependent chain on one accumulator),
// N x v_fma_f32 (independent chains on other registers)] and reports WALL time per iteration (host events) next to clock64 ticks
// (a tick lasts 0.46 .. 1.15 ns depending on the load: compare wall times):
//   overlapped:   ~ max(MFMA, vector) per SIMD, the waves of a SIMD sharing both pipes;   serialised: their sum.
// ACC = 0: accumulator in VGPRs, 1: in AGPRs.   build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int ACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(long long* out, int iters) {
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = 1.0f + i + threadIdx.x;
  const float c = 0.999f;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
    for (int j = 0; j < N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j & 15]) : "v"(c));
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i] + x[i];
  if (s == 12345.678f) out[1] = 1;  // keep the results alive
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int N, int ACC, int WAVES>
void run(long long* d) {
  const int iters = 4000;
  probe<N, ACC, WAVES><<<256, WAVES * 64>>>(d, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<N, ACC, WAVES><<<256, WAVES * 64>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("[%.1f ns per iteration by the host's events -> one clock64 tick = %.2f ns] ", ms * 1e6 / iters, ms * 1e6 / (double)h);
  printf("waves/SIMD %d  acc in %s  N = %2d vector ops per MFMA: %7.1f clocks per iteration (serial %d, overlapped %d per wave; x waves/SIMD sharing)\n", WAVES / 4,
         ACC ? "AGPRs" : "VGPRs", N, (double)h / iters, 32 + 4 * N, 32 > 4 * N ? 32 : 4 * N);
}

int main() {
  long long* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  run<0, 0, 4>(d); run<4, 0, 4>(d); run<8, 0, 4>(d); run<16, 0, 4>(d);
  run<0, 1, 4>(d); run<8, 1, 4>(d); run<16, 1, 4>(d);
  run<0, 0, 8>(d); run<4, 0, 8>(d); run<8, 0, 8>(d); run<12, 0, 8>(d); run<16, 0, 8>(d);
  run<0, 1, 8>(d); run<4, 1, 8>(d); run<8, 1, 8>(d); run<12, 1, 8>(d); run<16, 1, 8>(d);
  run<8, 0, 16>(d); run<8, 1, 16>(d);
  return 0;
}
