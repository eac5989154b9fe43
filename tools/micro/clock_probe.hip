// Developer probe (not part of libsopro_hip): what shader clock does the chip run at WHILE a workload runs?  One wave reads the
// shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around a sleep loop of `real_ticks` RTC ticks:
// effective MHz = 100 * dshader / dreal.  Launched on a stream of its own beside the workload (tools/r06/saturation_probe.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o clock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_probe_kernel(uint64_t* out, uint64_t real_ticks) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint64_t r0 = wall_clock64(), c0 = clock64();
  uint64_t r1;
  do {
    __builtin_amdgcn_s_sleep(64);
    r1 = wall_clock64();
  } while (r1 - r0 < real_ticks);
  const uint64_t c1 = clock64();
  if (threadIdx.x == 0) {
    out[0] = r1 - r0;
    out[1] = c1 - c0;
    out[2] = xcc & 15;
    out[3] = r0;
  }
}

extern "C" int clock_probe_alloc(void** host, int n_slots) { return (int)hipHostMalloc(host, (size_t)n_slots * 32, hipHostMallocDefault); }
extern "C" int clock_probe_stream(void** st) { return (int)hipStreamCreateWithFlags(reinterpret_cast<hipStream_t*>(st), hipStreamNonBlocking); }
extern "C" int clock_probe_launch(void* slot, uint64_t real_ticks, void* st) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(st), reinterpret_cast<uint64_t*>(slot), real_ticks);
  return (int)hipGetLastError();
}
extern "C" int clock_probe_sync(void* st) { return (int)hipStreamSynchronize(reinterpret_cast<hipStream_t>(st)); }
