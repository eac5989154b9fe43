// Error reporting, introspection and hipGraph helpers of the C ABI.
#include <stdarg.h>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void sopro_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int g_sopro_lds_floor = 0;

extern "C" {

const char* sopro_last_error(void) { return g_err; }

int sopro_abi_version(void) { return SOPRO_ABI_VERSION; }
int sopro_build_flags(void) {
#ifdef SOPRO_DEV_SWITCHES
  return 1;
#else
  return 0;
#endif
}

int sopro_set_lds_floor(int bytes) {
  SOPRO_CHECK_ARG(bytes >= 0 && bytes <= 96 * 1024, "LDS floor must be within 0..96 KiB");
  g_sopro_lds_floor = bytes;
  return 0;
}

int sopro_set_host_wait(int blocking) {
  // hipSetDeviceFlags on the CURRENT device: how host threads of this process wait in hipStreamSynchronize / hipEventSynchronize /
  // synchronous copies from now on.  0 = spin (the runtime's default: lowest wake-up latency, one busy core per waiting thread);
  // 1 = block on the completion signal's interrupt.
  SOPRO_HIP(hipSetDeviceFlags(blocking ? hipDeviceScheduleBlockingSync : hipDeviceScheduleSpin));
  return 0;
}

int sopro_device_info(int device, int* out4) {
  SOPRO_CHECK_ARG(out4 != nullptr, "out4 is NULL");
  hipDeviceProp_t p;
  SOPRO_HIP(hipGetDeviceProperties(&p, device));
  out4[0] = p.multiProcessorCount;
  out4[1] = (int)p.sharedMemPerBlock;
  out4[2] = p.clockRate;
  int arch = 0;
  const char* g = strstr(p.gcnArchName, "gfx");
  if (g) arch = atoi(g + 3);
  out4[3] = arch;
  return 0;
}

int sopro_capture_begin(void* stream) {
  SOPRO_CHECK_ARG(stream != nullptr, "capture needs a non-default stream");
  SOPRO_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}

int sopro_capture_end(void* stream, void** graph_exec_out) {
  SOPRO_CHECK_ARG(stream != nullptr && graph_exec_out != nullptr, "NULL argument");
  hipGraph_t graph = nullptr;
  SOPRO_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    sopro_set_error("sopro_capture_end: hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return -1;
  }
  *graph_exec_out = (void*)exec;
  return 0;
}

int sopro_graph_launch(void* graph_exec, void* stream) {
  SOPRO_CHECK_ARG(graph_exec != nullptr, "graph is NULL");
  SOPRO_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return 0;
}

int sopro_graph_launch_n(void* graph_exec, void* stream, int32_t n) {
  SOPRO_CHECK_ARG(graph_exec != nullptr && n >= 0, "graph is NULL or n < 0");
  for (int32_t i = 0; i < n; ++i) SOPRO_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return 0;
}

/* A HIP stream whose kernels may only use CUs [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask): lets a
 * latency-bound launch sequence keep a slice of the chip while a throughput-bound phase runs on the rest. */
int sopro_stream_create_cu_range(int first_cu, int n_cus, void** stream_out) {
  SOPRO_CHECK_ARG(stream_out != nullptr && first_cu >= 0 && n_cus > 0, "bad CU range");
  int dev = 0;
  SOPRO_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  SOPRO_HIP(hipGetDeviceProperties(&p, dev));
  const int total = p.multiProcessorCount;
  SOPRO_CHECK_ARG(first_cu + n_cus <= total, "CU range exceeds the device");
  uint32_t mask[16];
  memset(mask, 0, sizeof(mask));
  for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
  hipStream_t s = nullptr;
  SOPRO_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask));
  *stream_out = (void*)s;
  return 0;
}

/* The general form: `mask` holds one bit per CU (bit i of word i/32), `words` 32-bit words. */
int sopro_stream_create_cu_mask(const uint32_t* mask, int words, void** stream_out) {
  SOPRO_CHECK_ARG(mask != nullptr && stream_out != nullptr && words > 0 && words <= 16, "bad CU mask");
  uint32_t m[16];
  memset(m, 0, sizeof(m));
  bool any = false;
  for (int i = 0; i < words; ++i) { m[i] = mask[i]; any = any || mask[i] != 0; }
  SOPRO_CHECK_ARG(any, "empty CU mask");
  hipStream_t s = nullptr;
  SOPRO_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, m));
  *stream_out = (void*)s;
  return 0;
}

int sopro_stream_destroy(void* stream) {
  if (stream) SOPRO_HIP(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

int sopro_host_alloc(int64_t bytes, void** out) {
  SOPRO_CHECK_ARG(bytes > 0 && out != nullptr, "bytes must be positive, out non-NULL");
  SOPRO_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  memset(*out, 0, (size_t)bytes);
  return 0;
}

int sopro_host_free(void* p) {
  if (p) SOPRO_HIP(hipHostFree(p));
  return 0;
}

int sopro_copy_to_host_async(void* dst_host, const void* src_dev, int64_t bytes, void* stream) {
  SOPRO_CHECK_ARG(dst_host && src_dev && bytes > 0, "NULL pointer or empty copy");
  SOPRO_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}

int sopro_graph_destroy(void* graph_exec) {
  if (graph_exec) SOPRO_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}

}  // extern "C"
