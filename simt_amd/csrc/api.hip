// Error plumbing of the C ABI (see include/simt_hip.h).
#include "common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void simt_set_error(const char* file, int line, const char* msg) {
  const char* base = strrchr(file, '/');
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", base ? base + 1 : file, line, msg);
}

extern "C" const char* simt_last_error(void) { return g_err; }
extern "C" int simt_abi_version(void) { return SIMT_ABI_VERSION; }

// ---- device-scope events (round 5).  The launch lists order their two HIP streams with events (weight gradients behind the dgrad chain,
// BatchNorm passes beside the frozen model's convs).  A default hipEvent performs a SYSTEM-scope release when it is recorded -- write-back
// and invalidation of the caches so that the host could read the data -- and the stream's next kernel waits for it: ~6.5 us of idle queue
// behind every record, ~12 us for a cross-stream wait (rocprofv3 kernel trace, profiles/r05_conv_attribution.txt section 6).  Both streams of a plan live on
// one device: a device-scope release is all the ordering they need.  scope 0 (default): hipEventReleaseToDevice, the documented device-scope
// release; scope 2: hipEventDisableSystemFence, which drops the release from the marker altogether (the edge then rests on the producing kernel's
// own end-of-kernel agent-scope release -- undocumented runtime behaviour, explicit opt-in); scope 1: the default system-scope event.  (The
// runtime accepts ONE of the two flags: both together are rejected.)
extern "C" int simt_event_create(void** ev, int scope) {
  SIMT_CHECK(ev && scope >= 0 && scope <= 2);
  hipEvent_t e;
  const unsigned fl = hipEventDisableTiming | (scope == 0 ? hipEventReleaseToDevice : scope == 2 ? hipEventDisableSystemFence : 0u);
  hipError_t err = hipEventCreateWithFlags(&e, fl);
  if (err != hipSuccess) { (void)hipGetLastError(); simt_set_error(__FILE__, __LINE__, "hipEventCreateWithFlags"); return SIMT_ERR_LAUNCH; }
  *ev = (void*)e;
  return SIMT_OK;
}
extern "C" int simt_event_destroy(void* ev) { return (ev && hipEventDestroy((hipEvent_t)ev) == hipSuccess) ? SIMT_OK : SIMT_ERR_INVALID; }
extern "C" int simt_event_record(void* ev, simt_stream_t stream) {
  SIMT_CHECK(ev);
  if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) { simt_set_error(__FILE__, __LINE__, "hipEventRecord"); return SIMT_ERR_LAUNCH; }
  return SIMT_OK;
}
extern "C" int simt_stream_wait_event(simt_stream_t stream, void* ev) {
  SIMT_CHECK(ev);
  if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0) != hipSuccess) { simt_set_error(__FILE__, __LINE__, "hipStreamWaitEvent"); return SIMT_ERR_LAUNCH; }
  return SIMT_OK;
}
