// -DSIMT_ABLATION builds only: the shader clock as a function of time.  One wave on its own stream samples s_memtime (shader clock counter) and
// s_memrealtime (constant 100 MHz) every `period` real-time ticks while other streams run the step; consecutive samples give the clock the chip
// ran at in that interval (profiles/tools/clock_timeline.py, profiles/r05_power_clock.txt).  simt_debug_mark: a one-thread launch that writes
// s_memrealtime -- markers of the phases of a step on the stream they are enqueued on.
#include "../common.h"

__global__ void clock_sampler_kernel(unsigned long long* out, int n, int period) {
  if (threadIdx.x != 0) return;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    const unsigned long long target = rt0 + (unsigned long long)(i + 1) * (unsigned long long)period;
    while (__builtin_amdgcn_s_memrealtime() < target) __builtin_amdgcn_s_sleep(8);
    const unsigned long long c = __builtin_amdgcn_s_memtime();
    const unsigned long long r = __builtin_amdgcn_s_memrealtime();
    out[2 * i] = c;
    out[2 * i + 1] = r;
  }
}
extern "C" int simt_debug_clock_sampler(unsigned long long* out, int n, int period_ticks, simt_stream_t stream) {
  SIMT_CHECK(out && n > 0 && period_ticks > 0);
  hipLaunchKernelGGL(clock_sampler_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, n, period_ticks);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

__global__ void mark_kernel(unsigned long long* out) { *out = __builtin_amdgcn_s_memrealtime(); }
extern "C" int simt_debug_mark(unsigned long long* out, simt_stream_t stream) {
  SIMT_CHECK(out);
  hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
