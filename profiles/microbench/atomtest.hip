// cost of per-channel int64 atomic accumulation at the end of a conv-like kernel: NWG workgroups x 512 threads, each workgroup adds C*2 values
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(512) void k(long long* acc, float* slots, int C, int mode, int spin) {
  // some work first so that all workgroups arrive at about the same time
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  const int t = threadIdx.x;
  if (mode == 0) {            // per-tile fp32 slots (what the product kernels write)
    if (t < C) { slots[((long)blockIdx.x * 2 + 0) * C + t] = v; slots[((long)blockIdx.x * 2 + 1) * C + t] = v * v; }
  } else {                    // + int64 atomics
    if (t < C) { slots[((long)blockIdx.x * 2 + 0) * C + t] = v; slots[((long)blockIdx.x * 2 + 1) * C + t] = v * v; }
    for (int c = t; c < 2 * C; c += 512) atomicAdd((unsigned long long*)&acc[c], (unsigned long long)(long long)(v * 65536.f));
  }
}
int main() {
  long long* acc; float* slots;
  hipMalloc(&acc, 2 * 2048 * 8); hipMalloc(&slots, 2048 * 2 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int C : {256, 1024, 2048})
    for (int nwg : {255, 295, 1180})
      for (int mode : {0, 1}) {
        std::vector<float> ts;
        for (int r = 0; r < 12; ++r) {
          hipMemsetAsync(acc, 0, 2 * 2048 * 8);
          hipEventRecord(e0);
          for (int q = 0; q < 10; ++q) hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 0, 0, acc, slots, C, mode, 20000);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) ts.push_back(ms * 100.f);
        }
        std::sort(ts.begin(), ts.end());
        printf("C %4d  workgroups %4d  %s: %.2f us per launch\n", C, nwg, mode ? "slots + int64 atomics" : "slots only           ", ts[ts.size() / 2]);
      }
  return 0;
}
