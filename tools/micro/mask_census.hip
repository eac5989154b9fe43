// Which CUs does a CU-masked stream really use?  4096 short workgroups per mask record (XCC_ID, SE, SH/SA, CU) from the hardware-id
// registers; the census prints distinct CUs and workgroups per XCD.  Masks: the pipeline's partitions (range and whole-XCD forms).
// build: hipcc --offload-arch=gfx950 -O2 -o mask_census mask_census.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const long long t0 = clock64();
  while (clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(8);  // ~10 us: long enough for every enabled CU to get work
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
}
static void census(const char* name, const std::vector<int>& bits, int total, unsigned* d) {
  uint32_t mask[16] = {0};
  for (int b : bits) mask[b >> 5] |= 1u << (b & 31);
  hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (total + 31) / 32, mask));
  const int n = 4096;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, s, d);
  CK(hipEventRecord(e1, s));
  std::vector<unsigned> h(2 * n);
  CK(hipMemcpyAsync(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::set<unsigned> cus[8]; int wgs[8] = {0};
  for (int i = 0; i < n; ++i) {
    const unsigned x = h[2 * i] & 7, hw = h[2 * i + 1];
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    cus[x].insert((hw >> 8) & 0xff);
    wgs[x]++;
  }
  int tot = 0;
  printf("%-34s %3zu bits, %7.3f ms | distinct CUs per XCD:", name, bits.size(), ms);
  for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); tot += (int)cus[x].size(); }
  printf(" = %3d | workgroups per XCD:", tot);
  for (int x = 0; x < 8; ++x) printf(" %4d", wgs[x]);
  printf(" | XCD of workgroups 0-23:");
  for (int i = 0; i < 24; ++i) printf(" %u", h[2 * i] & 7);
  printf("\n");
  CK(hipStreamDestroy(s));
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int total = p.multiProcessorCount;
  unsigned* d; CK(hipMalloc(&d, 2 * 4096 * sizeof(unsigned)));
  auto sel = [&](auto pred) { std::vector<int> v; for (int i = 0; i < total; ++i) if (pred(i)) v.push_back(i); return v; };
  census("all", sel([](int) { return true; }), total, d);
  census("range [0, 64)", sel([](int i) { return i < 64; }), total, d);
  census("range [64, 256)", sel([](int i) { return i >= 64; }), total, d);
  census("range [0, 96)", sel([](int i) { return i < 96; }), total, d);
  census("range [96, 256)", sel([](int i) { return i >= 96; }), total, d);
  census("whole XCDs 0-1", sel([](int i) { return i % 8 < 2; }), total, d);
  census("whole XCDs 2-7", sel([](int i) { return i % 8 >= 2; }), total, d);
  census("whole XCDs 0-3", sel([](int i) { return i % 8 < 4; }), total, d);
  census("whole XCDs 4-7", sel([](int i) { return i % 8 >= 4; }), total, d);
  census("16 slots of XCDs 0-3", sel([](int i) { return i % 8 < 4 && i / 8 < 16; }), total, d);
  census("the rest of that", sel([](int i) { return !(i % 8 < 4 && i / 8 < 16); }), total, d);
  census("one bit (5)", sel([](int i) { return i == 5; }), total, d);
  return 0;
}
