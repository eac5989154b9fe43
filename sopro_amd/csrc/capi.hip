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

extern "C" {

const char* sopro_last_error(void) { return g_err; }

int sopro_abi_version(void) { return SOPRO_ABI_VERSION; }

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
  hipGraphDestroy(graph);
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

int sopro_graph_destroy(void* graph_exec) {
  if (graph_exec) SOPRO_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}

}  // extern "C"
