// Runtime glue of libflowtrack_hip.so: status/error reporting, device query, HIP-graph capture of
// a launch sequence, hipEvent timing on the caller's stream.  No global device state is kept; the
// last-error text is thread-local.
#include <stdio.h>
#include <string.h>

#include "ft_common.h"

namespace ft {

static thread_local char g_last_error[256] = "";

int record_hip_error(hipError_t e, const char* what) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
  return FT_ERR_HIP;
}

}  // namespace ft

using namespace ft;

extern "C" int ft_version(void) { return 100; }  // 0.1.0

extern "C" const char* ft_status_string(int status) {
  switch (status) {
    case FT_OK: return "ok";
    case FT_ERR_INVALID_ARG: return "invalid argument";
    case FT_ERR_UNSUPPORTED: return "unsupported configuration";
    case FT_ERR_HIP: return "HIP runtime error";
    case FT_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

extern "C" const char* ft_last_hip_error(void) { return g_last_error; }

extern "C" int ft_device_info(int device, char* name, int name_len, int* cu_count, uint64_t* hbm_bytes) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FT_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return FT_ERR_INVALID_ARG;
  hipDeviceProp_t prop;
  FT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (name && name_len > 0) {
    snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
  return FT_OK;
}

extern "C" int ft_graph_begin_capture(ft_stream_t stream) {
  if (!stream) return FT_ERR_INVALID_ARG;  // the legacy default stream cannot be captured
  FT_HIP_CHECK(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return FT_OK;
}

extern "C" int ft_graph_end_capture(ft_stream_t stream, void** graph_exec_out) {
  if (!stream || !graph_exec_out) return FT_ERR_INVALID_ARG;
  hipGraph_t graph = nullptr;
  FT_HIP_CHECK(hipStreamEndCapture(as_stream(stream), &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return record_hip_error(e, "hipGraphInstantiate");
  *graph_exec_out = exec;
  return FT_OK;
}

extern "C" int ft_graph_launch(void* graph_exec, ft_stream_t stream) {
  if (!graph_exec) return FT_ERR_INVALID_ARG;
  FT_HIP_CHECK(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), as_stream(stream)));
  return FT_OK;
}

extern "C" int ft_graph_destroy(void* graph_exec) {
  if (!graph_exec) return FT_OK;
  FT_HIP_CHECK(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
  return FT_OK;
}

extern "C" int ft_event_create(void** event_out) {
  if (!event_out) return FT_ERR_INVALID_ARG;
  hipEvent_t ev;
  FT_HIP_CHECK(hipEventCreate(&ev));
  *event_out = ev;
  return FT_OK;
}

extern "C" int ft_event_record(void* event, ft_stream_t stream) {
  if (!event) return FT_ERR_INVALID_ARG;
  FT_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(event), as_stream(stream)));
  return FT_OK;
}

extern "C" int ft_stream_wait_event(ft_stream_t stream, void* event) {
  if (!event) return FT_ERR_INVALID_ARG;
  FT_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), static_cast<hipEvent_t>(event), 0));
  return FT_OK;
}

extern "C" int ft_event_synchronize(void* event) {
  if (!event) return FT_ERR_INVALID_ARG;
  FT_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return FT_OK;
}

extern "C" int ft_event_elapsed_ms(void* start, void* stop, float* ms_out) {
  if (!start || !stop || !ms_out) return FT_ERR_INVALID_ARG;
  FT_HIP_CHECK(hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return FT_OK;
}

extern "C" int ft_event_destroy(void* event) {
  if (!event) return FT_OK;
  FT_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return FT_OK;
}

extern "C" int ft_memcpy_async(void* dst, const void* src, size_t bytes, ft_stream_t stream) {
  if (!dst || !src) return FT_ERR_INVALID_ARG;
  if (bytes == 0) return FT_OK;
  FT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, as_stream(stream)));
  return FT_OK;
}

extern "C" int ft_stream_synchronize(ft_stream_t stream) {
  FT_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  return FT_OK;
}
