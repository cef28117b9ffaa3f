// MEASUREMENT ONLY (profiles/tools/dp_emulate.py): a stand-in for the CU footprint of a collective's persistent kernels on a
// 1-GPU box.  `n` workgroups of `threads` threads with `lds` bytes of LDS stay resident for `us` microseconds (s_memrealtime: constant
// 100 MHz), sleeping -- like an RCCL channel waiting on its peers they hold their CU slots and do no memory traffic.  Never part of the product
// library: built into its own profiles/tools/liboccupier.so.
#include <hip/hip_runtime.h>
__global__ void occupier_kernel(unsigned long long ticks, int touch) {
  extern __shared__ char smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (touch) smem[threadIdx.x] = 1;          // (keeps the dynamic LDS allocation alive)
}
extern "C" int occ_launch(int n, int threads, int lds, double us, void* stream) {
  if (n <= 0) return 0;
  static int attr_lds = 0;
  if (lds > attr_lds) { (void)hipFuncSetAttribute((const void*)occupier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_lds = lds; }
  hipLaunchKernelGGL(occupier_kernel, dim3(n), dim3(threads), lds, (hipStream_t)stream, (unsigned long long)(us * 100.0), 0);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
