// Developer probe: what the LDS pipe of one CU delivers to ds_read_b128 / ds_read_b64 / ds_read_b32 (GB/s by the host's events; bytes per clock64 tick beside it - a tick is NOT a fixed time, see r04_experiments.md), for 4 /
// 8 / 16 waves per CU, with the row strides the fused SEANet kernels use (272 B and 528 B rows: a 16-lane group covers all banks) and
// a plain contiguous pattern.   build: hipcc --offload-arch=gfx950 -O3 -w lds_bw.hip -o lds_bw
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int BYTES, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(long long* out, int iters, int row_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 16 - 64; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // row_stride == 0: lane-contiguous BYTES-sized pieces; otherwise the MFMA A-fragment pattern: row = lane & 31, 16-byte column = lane >> 5
  const unsigned base = row_stride ? (unsigned)((lane & 31) * row_stride + (lane >> 5) * 16 + wave * 32 * row_stride % 8192) : (unsigned)(lane * BYTES + wave * 4096);
  unsigned acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned addr = base + u * (row_stride ? 32 : 64 * BYTES) % 16384;
      if (BYTES == 16) {
        uint4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(6)");
        acc += v.x;
      } else if (BYTES == 8) {
        uint2 v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(6)");
        acc += v.x;
      } else {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(6)");
        acc += v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const long long t1 = clock64();
  if (acc == 0x12345678u) out[1] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int BYTES, int WAVES>
void run(long long* d, int row_stride) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<BYTES, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  probe<BYTES, WAVES><<<256, WAVES * 64, 160 * 1024>>>(d, iters, row_stride);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<BYTES, WAVES><<<256, WAVES * 64, 160 * 1024>>>(d, iters, row_stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("[tick = %.2f ns; %.0f GB/s per CU] ", ms * 1e6 / (double)h, (double)iters * 8 * WAVES * 64 * BYTES / (ms * 1e6));
  const double bytes = (double)iters * 8 * WAVES * 64 * BYTES;
  printf("ds_read_b%-3d %2d waves per CU, %s: %7.1f bytes per clock per CU (%.1f clocks per wave instruction)\n", BYTES * 8, WAVES,
         row_stride ? (row_stride == 272 ? "rows of 272 B" : "rows of 528 B") : "contiguous   ", bytes / (double)h, (double)h / (iters * 8));
}

int main() {
  long long* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  run<16, 4>(d, 0); run<16, 8>(d, 0); run<16, 16>(d, 0);
  run<16, 8>(d, 272); run<16, 16>(d, 272); run<16, 8>(d, 528); run<16, 16>(d, 528);
  run<8, 8>(d, 0); run<8, 16>(d, 0); run<4, 8>(d, 0); run<4, 16>(d, 0);
  return 0;
}
