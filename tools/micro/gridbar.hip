// Micro-benchmark: cost of a device-scope grid barrier among G co-resident workgroups (bounded spin: never hangs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22)) { *fail = 1; ok = false; break; }
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

__global__ void k(unsigned* ctr, unsigned* fail, float* data, int nbar, int work) {
  float acc = 0.f;
  for (int i = 0; i < nbar; ++i) {
    // a little dependent traffic between barriers, like a stage would have
    for (int w = 0; w < work; ++w) acc += data[(blockIdx.x * 256 + threadIdx.x + w * 65536 + i * 4096) & 0xFFFFF];
    if (threadIdx.x == 0) data[(blockIdx.x + i * 64) & 0xFFFFF] = acc;
    if (!grid_barrier(ctr, (unsigned)(i + 1) * gridDim.x, fail)) return;
  }
  if (acc == 123.456f) data[0] = acc;
}

int main(int argc, char** argv) {
  unsigned *ctr, *fail; float* data;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&data, 4 << 20));
  CK(hipMemset(data, 0, 4 << 20));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int total = p.multiProcessorCount;
  for (int cus : {32, 64, 128, 256}) {
    uint32_t mask[16] = {0};
    for (int c = 0; c < cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (total + 31) / 32, mask));
    for (int work : {0, 4}) {
      for (int G : {cus / 2, cus}) {
        const int nbar = 2000;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipMemsetAsync(ctr, 0, 4, s)); CK(hipMemsetAsync(fail, 0, 4, s));
          CK(hipEventRecord(e0, s));
          hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, s, ctr, fail, data, nbar, work);
          CK(hipEventRecord(e1, s));
          CK(hipStreamSynchronize(s));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("cus=%3d G=%3d work=%d: %.3f us per barrier%s\n", cus, G, work, ms * 1e3 / nbar, f ? "  (SPIN LIMIT HIT)" : "");
      }
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
