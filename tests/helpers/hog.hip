// A kernel that does nothing but HOLD compute units for a while: `blocks` workgroups, each keeping `lds_bytes` of LDS
// (so that a kernel that needs most of a CU's LDS cannot share the CU) for `micros` microseconds of wall-clock time.
// Test infrastructure (tests/test_gpu_scan.py: the one-pass time-parallel kernel beside foreign work); built by
// __graft_entry__.build() into tests/helpers/libhog.so.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void k_hog(long long ticks, int *sink) {
  extern __shared__ char lds[];
  lds[threadIdx.x] = (char)threadIdx.x;
  const long long t0 = wall_clock64();                  // 100 MHz, independent of the shader clock
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (lds[(threadIdx.x + 1) & 63] == 77 && sink) *sink = 1;
}

extern "C" int hog_launch(int blocks, int lds_bytes, long long micros, void *stream) {
  if (hipFuncSetAttribute((const void *)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_hog, dim3((unsigned)blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, micros * 100, (int *)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
