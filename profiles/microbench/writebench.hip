// HBM write / read / copy rates of plain streaming kernels (16 B per lane), the roofs the HBM-bound kernels are held against.
// build: hipcc --offload-arch=gfx950 -O3 writebench.hip -o writebench ; run: ./writebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0 write, 1 write nontemporal, 2 read (sum), 3 copy, 4 copy nt store
__global__ __launch_bounds__(256) void k(uint4* __restrict__ dst, const uint4* __restrict__ src, long n, unsigned* sink) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  uint4 acc = make_uint4(threadIdx.x, 1, 2, 3);
  for (; i < n; i += stride) {
    if (MODE == 0) dst[i] = acc;
    else if (MODE == 1) __builtin_nontemporal_store(*(const u4*)&acc, (u4*)(dst + i));
    else if (MODE == 2) { uint4 v = src[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    else if (MODE == 3) dst[i] = src[i];
    else { const uint4 v = src[i]; __builtin_nontemporal_store(*(const u4*)&v, (u4*)(dst + i)); }
  }
  if (MODE == 2 && acc.x == 0x12345678u && acc.y == 77u) *sink = acc.z;
}

int main() {
  const long mb[] = {19, 77, 308, 1024};
  const int grids[] = {1024, 2048, 8192};
  unsigned* sink; CK(hipMalloc(&sink, 4));
  for (long m : mb) {
    const long bytes = m << 20, n = bytes / 16;
    const int NB = 6;                                  // rotate buffers: data comes from / goes to HBM, not a warm cache
    std::vector<uint4*> a(NB), b(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&a[i], bytes)); CK(hipMalloc(&b[i], bytes)); CK(hipMemset(a[i], 1, bytes)); CK(hipMemset(b[i], 2, bytes)); }
    for (int g : grids) {
      float t[5];
      for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 5;
        for (int warm = 0; warm < 2; ++warm) {
          CK(hipEventRecord(e0));
          for (int r = 0; r < reps; ++r)
            for (int i = 0; i < NB; ++i) {
              if (mode == 0) k<0><<<g, 256>>>(a[i], b[i], n, sink);
              if (mode == 1) k<1><<<g, 256>>>(a[i], b[i], n, sink);
              if (mode == 2) k<2><<<g, 256>>>(a[i], b[i], n, sink);
              if (mode == 3) k<3><<<g, 256>>>(a[i], b[i], n, sink);
              if (mode == 4) k<4><<<g, 256>>>(a[i], b[i], n, sink);
            }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&t[mode], e0, e1));
        t[mode] = t[mode] * 1e3f / (reps * NB);        // us per launch
      }
      printf("%5ld MB grid %5d: write %7.1f us (%5.2f TB/s)  write-nt %7.1f (%5.2f)  read %7.1f (%5.2f)  copy %7.1f (%5.2f of r+w)  copy-nt %7.1f (%5.2f)\n", m, g,
             t[0], bytes / t[0] / 1e6, t[1], bytes / t[1] / 1e6, t[2], bytes / t[2] / 1e6, t[3], 2.0 * bytes / t[3] / 1e6, t[4], 2.0 * bytes / t[4] / 1e6);
    }
    for (int i = 0; i < NB; ++i) { hipFree(a[i]); hipFree(b[i]); }
  }
  return 0;
}
