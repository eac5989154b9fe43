// Which XCD does CU-mask bit i belong to?  One workgroup on a stream masked to a single bit reports its XCC_ID.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 0xf);
}
int main() {
  int* d; CK(hipMalloc(&d, 64 * sizeof(int)));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int total = p.multiProcessorCount;
  for (int bit : {0, 1, 2, 3, 7, 8, 9, 15, 16, 31, 32, 33, 64, 65, 128, 255}) {
    if (bit >= total) continue;
    uint32_t mask[16] = {0};
    mask[bit >> 5] = 1u << (bit & 31);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (total + 31) / 32, mask));
    CK(hipMemsetAsync(d, 0xff, 64 * sizeof(int), s));
    hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, s, d);
    int h[8]; CK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    printf("mask bit %3d -> XCC_ID of 8 workgroups:", bit);
    for (int i = 0; i < 8; ++i) printf(" %d", h[i]);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}
