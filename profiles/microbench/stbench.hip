// Store-stream micro-benchmark: how fast can 256 persistent workgroups write a [M][1024] bf16 output in the access pattern of the
// rows kernel (each workgroup: 32-row x 512-byte slabs of one 256-column band), by number of storing waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
// PAT 0: rows-kernel pattern (band of 256 columns); PAT 1: each workgroup writes whole 2048-byte rows (contiguous)
template <int NW, int PAT>
__global__ __launch_bounds__(NW * 64) void st_kernel(char* y, int M, int ntm, int ntn) {
  const int tid = threadIdx.x;
  const int G = gridDim.x, nwg = ntm * ntn;
  const int my_n = (nwg - (int)blockIdx.x + G - 1) / G;
  const int tile0 = xcd_remap(blockIdx.x, nwg), tstep = G >> 3;
  const uint4 v = make_uint4(tid, tid, tid, tid);
  for (int i = 0; i < my_n; ++i) {
    const int t = tile0 + i * tstep;
    if (PAT == 0) {
      const int mt = t / ntn, nt = t % ntn;
      // 128 rows x 512 B = 4096 pieces of 16 B
      for (int p = tid; p < 4096; p += NW * 64) {
        const int r = p >> 5, c = p & 31;
        const int m = mt * 128 + r;
        if (m < M) *(uint4*)(y + (size_t)m * 2048 + nt * 512 + c * 16) = v;
      }
    } else {
      // tile t covers 32 whole rows: 4096 pieces
      for (int p = tid; p < 4096; p += NW * 64) {
        const size_t off = (size_t)t * 65536 + (size_t)p * 16;
        if (off < (size_t)M * 2048) *(uint4*)(y + off) = v;
      }
    }
  }
}
template <typename F> float time_it(F&& launch, int nsets) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> r;
  for (int rnd = 0; rnd < 5; ++rnd) {
    for (int s = 0; s < nsets; ++s) launch(s);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 4; ++rep) for (int s = 0; s < nsets; ++s) launch(s);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); r.push_back(ms * 1e3f / (4 * nsets));
  }
  std::sort(r.begin(), r.end());
  return r[r.size() / 2];
}
int main() {
  const int M = 37636, ntm = (M + 127) / 128, ntn = 4, NS = 6;
  std::vector<char*> ys(NS);
  for (auto& p : ys) { hipMalloc(&p, (size_t)(M + 256) * 2048); hipMemset(p, 0, (size_t)(M + 256) * 2048); }
  const double bytes = (double)M * 2048;
#define RUN(NW, PAT, G) { float us = time_it([&](int s) { st_kernel<NW, PAT><<<G, NW * 64>>>(ys[s], M, ntm, ntn); }, NS); \
    printf("waves %2d pattern %d grid %4d: %6.1f us  %.2f TB/s\n", NW, PAT, G, us, bytes / us / 1e6); }
  RUN(4, 0, 256); RUN(8, 0, 256); RUN(12, 0, 256); RUN(16, 0, 256);
  RUN(4, 1, 256); RUN(8, 1, 256); RUN(16, 1, 256);
  RUN(4, 0, 512); RUN(4, 0, 1024); RUN(8, 0, 1180);
  printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
}
