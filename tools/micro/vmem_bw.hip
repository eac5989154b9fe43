// Developer probe: what one CU's vector memory path delivers to global_load_dwordx4 when the data sits in its L1 (a 8 KB window per
// CU), in L2 (a 1 MB window per CU, re-read), or comes from the memory side (a 64 MB window per CU walked once) - GB/s per CU by
// the host's events, 8 and 16 waves per CU, all CUs at once.  The split contractions take their W fragments straight from L2
// through this path (and their A rows from memory): how much room is there?
// build: hipcc --offload-arch=gfx950 -O3 -w vmem_bw.hip -o vmem_bw
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(const uint4* __restrict__ src, long long window_u4, int iters, unsigned* out, int shared) {
  // shared: all workgroups of an XCD (blockIdx % 8) walk the SAME window - operands every workgroup needs (weights): L2 hits
  const uint4* base = src + (long long)(shared ? blockIdx.x % 8 : blockIdx.x) * window_u4;
  unsigned acc = 0;
  const int tid = threadIdx.x;
  long long off = tid;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint4 v = base[off];
      acc += (v.x ^ v.y) + (v.z ^ v.w);
      off += WAVES * 64;
      if (off >= window_u4) off -= window_u4;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int WAVES>
void run(const uint4* d, unsigned* o, long long window_bytes, int iters, const char* what, int shared = 0) {
  const long long w4 = window_bytes / 16;
  probe<WAVES><<<256, WAVES * 64>>>(d, w4, iters / 4, o, shared);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<WAVES><<<256, WAVES * 64>>>(d, w4, iters, o, shared);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 8 * WAVES * 64 * 16;
  printf("%-28s %2d waves per CU: %7.1f GB/s per CU, %6.2f TB/s on the chip\n", what, WAVES, bytes / (ms * 1e6), bytes * 256 / (ms * 1e9));
}

int main() {
  const long long total = 256LL * (64LL << 20);
  uint4* d;
  unsigned* o;
  if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&o, 64);
  hipMemset(d, 1, total);
  run<8>(d, o, 8 << 10, 20000, "L1 window (8 KB per CU)");
  run<16>(d, o, 8 << 10, 20000, "L1 window (8 KB per CU)");
  run<8>(d, o, 2 << 20, 4000, "L2: 2 MB shared per XCD", 1);
  run<16>(d, o, 2 << 20, 4000, "L2: 2 MB shared per XCD", 1);
  run<8>(d, o, 1 << 20, 4000, "cache side: 1 MB per CU");
  run<16>(d, o, 1 << 20, 4000, "cache side: 1 MB per CU");
  run<8>(d, o, 64 << 20, 1024, "memory (64 MB per CU, once)");
  run<16>(d, o, 64 << 20, 512, "memory (64 MB per CU, once)");
  return 0;
}
