// Micro-benchmark (round 4): does the PITCH of the 128-byte pieces bound the activation stream of the long-K 1x1 conv (1024 -> 256, M = 37 636)?
// conv_igemm2_kernel reads its pixel operand as [148 rows] x 128 B per 64-deep K stage; in NHWC with C = 1024 the rows are 2 KB apart, and the
// kernel moves 77 MB at 2.6 TB/s with two stages in flight per CU (loads-only build: 28 us).  Same piece count, same ring depth, three layouts:
//   0: NHWC, pitch 2048 B          piece (row r, stage k) at (m0 + r) * 2048 + k * 128
//   1: channel-blocked [C/256][M][256]: pitch 512 B   plane k / 4: plane * M * 512 + (m0 + r) * 512 + (k % 4) * 128
//   2: K-major [C/64][M][64]: pitch 128 B (a stage is one contiguous 19-KB run)
// plus the ring depth (stages in flight) as a second axis.  build: hipcc --offload-arch=gfx950 -O3 -o stridebench stridebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vmcnt() {
#define C(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C(3); C(6); C(9); C(12);
#undef C
}
constexpr int NT = 512, ROWS = 160, NK = 16;       // 160-row slots (148 live), 16 K stages of 128 B
struct Args { const char* x; int M, rows; unsigned* sink; };

template <int LAYOUT, int INFLIGHT, int TILES = 1, bool NOBAR = false>
__global__ __launch_bounds__(NT) void stream_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSLOT = INFLIGHT + 1, SB = ROWS * 128;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int m0 = blockIdx.x * a.rows * TILES;
  unsigned off[3];
  bool live[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = i * 64 + (tid >> 3);
    const int m = m0 + row;
    live[i] = row < a.rows && m < a.M && (i < 2 || wave < 4);
    const unsigned pitch = LAYOUT == 0 ? 2048u : LAYOUT == 1 ? 512u : 128u;
    off[i] = (unsigned)(live[i] ? m : 0) * pitch + (tid & 7) * 16;
  }
  auto issue = [&](int kk, int slot) {
    const int k = kk % NK;
    const size_t tile_off = (size_t)(kk / NK) * a.rows * (LAYOUT == 0 ? 2048u : LAYOUT == 1 ? 512u : 128u);
    unsigned so;
    if (LAYOUT == 0) so = (unsigned)k * 128u;
    else if (LAYOUT == 1) so = (unsigned)(k >> 2) * (unsigned)a.M * 512u + (unsigned)(k & 3) * 128u;
    else so = (unsigned)k * (unsigned)a.M * 128u;
    char* sb = smem + slot * SB;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(a.x + tile_off + (size_t)(off[i] + so)), LPTR(sb + (i * NT + wave * 64) * 16), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < INFLIGHT; ++s) issue(s, s);
  constexpr int NKT = NK * TILES;
  for (int k = 0; k < NKT; ++k) {
    if (k + INFLIGHT <= NKT) wait_vmcnt<3 * (INFLIGHT - 1)>(); else wait_vmcnt<0>();
    if (!NOBAR) __builtin_amdgcn_s_barrier();
    if (k + INFLIGHT < NKT) issue(k + INFLIGHT, (k + INFLIGHT) % NSLOT);
  }
  if (tid == 0 && a.sink) a.sink[blockIdx.x] = *(unsigned*)smem;
}

// how the operand is prepared before every timed launch: 0 = a 512-MB memset of another buffer (evicts, but leaves the caches DIRTY: the timed
// reads then pay for the write-backs), 1 = a 600-MB READ of another buffer (evicts, clean), 2 = the operand itself freshly WRITTEN by a
// streaming kernel (what the conv sees in the step: its input was just produced by the BatchNorm pass)
static int g_prep = 0;
__global__ void read_kernel(const uint4* p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void write_kernel(uint4* p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v, v, v);
}
template <int LAYOUT, int INFLIGHT, int TILES = 1, bool NOBAR = false> static float run(Args a, int grid, char* flush, size_t fbytes) {
  const size_t lds = (size_t)(INFLIGHT + 1) * ROWS * 128;
  hipFuncSetAttribute((const void*)stream_kernel<LAYOUT, INFLIGHT, TILES, NOBAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<float> ts;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 7; ++rep) {
    if (g_prep == 0) hipMemsetAsync(flush, rep, fbytes, 0);          // push the operand out of L2 / Infinity Cache (512 MB > 256 MiB)
    else {
      hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, 0, (const uint4*)flush, fbytes / 16, a.sink);
      if (g_prep == 2) hipLaunchKernelGGL(write_kernel, dim3(4096), dim3(256), 0, 0, (uint4*)a.x, (size_t)a.M * 2048 * TILES / 16, (unsigned)rep);
    }
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((stream_kernel<LAYOUT, INFLIGHT, TILES, NOBAR>), dim3(grid), dim3(NT), lds, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const int M = 37636, rows = 148, grid = (M + rows - 1) / rows;
  const size_t bytes = (size_t)M * 2048, fbytes = 600u << 20;
  char *x, *flush; unsigned* sink;
  hipMalloc(&x, 4 * bytes + 4096); hipMemset(x, 1, 4 * bytes + 4096);
  hipMalloc(&flush, fbytes); hipMalloc(&sink, 4096 * 4);
  Args a{x, M, rows, sink};
  const char* names[3] = {"NHWC pitch 2048", "blocked [C/256][M][256] pitch 512", "K-major [C/64][M][64] pitch 128"};
  printf("%d workgroups x %d rows x 2048 B = %.1f MB, 16 stages of 128 B per row; us (TB/s)\n", grid, rows, bytes / 1e6);
  printf("%-40s %18s %18s %18s\n", "layout", "2 stages in flight", "3 in flight", "4 in flight");
  for (g_prep = 0; g_prep < 3; ++g_prep) {
  printf("-- operand prepared by: %s\n", g_prep == 0 ? "512-MB memset of another buffer (dirty caches)" : g_prep == 1 ? "600-MB read of another buffer (clean, cold)" : "its own producer (freshly written, 77 MB)");
  float r[3][3];
  r[0][0] = run<0, 2>(a, grid, flush, fbytes); r[0][1] = run<0, 3>(a, grid, flush, fbytes); r[0][2] = run<0, 4>(a, grid, flush, fbytes);
  r[1][0] = run<1, 2>(a, grid, flush, fbytes); r[1][1] = run<1, 3>(a, grid, flush, fbytes); r[1][2] = run<1, 4>(a, grid, flush, fbytes);
  r[2][0] = run<2, 2>(a, grid, flush, fbytes); r[2][1] = run<2, 3>(a, grid, flush, fbytes); r[2][2] = run<2, 4>(a, grid, flush, fbytes);
  for (int l = 0; l < 3; ++l)
    printf("%-40s %8.1f (%5.2f)    %8.1f (%5.2f)    %8.1f (%5.2f)\n", names[l], r[l][0], bytes / r[l][0] / 1e6, r[l][1], bytes / r[l][1] / 1e6, r[l][2],
           bytes / r[l][2] / 1e6);
  // steady state: every workgroup streams FOUR consecutive 148-row tiles (308 MB in all; M rows per plane stay as above, so layouts 1 / 2
  // read planes of the first M rows four times over -- layout 0 only), and the same without the per-stage workgroup barrier
  {
    const float t4 = run<0, 2, 4>(a, grid, flush, fbytes), t4b = run<0, 3, 4>(a, grid, flush, fbytes), t4n = run<0, 2, 4, true>(a, grid, flush, fbytes);
    const float t1n = run<0, 2, 1, true>(a, grid, flush, fbytes);
    printf("NHWC, 4 tiles per workgroup (308 MB): 2 in flight %.1f us (%.2f TB/s), 3 in flight %.1f (%.2f), 2 in flight without the stage barrier %.1f (%.2f)\n",
           t4, 4 * bytes / t4 / 1e6, t4b, 4 * bytes / t4b / 1e6, t4n, 4 * bytes / t4n / 1e6);
    printf("NHWC, 1 tile, without the stage barrier: %.1f us (%.2f TB/s)\n", t1n, bytes / t1n / 1e6);
  }
  }
  return 0;
}
