// Write-only streams with the access pattern of k_pipe's storer wave (configs[3]: y [16384 rows][65536] float64, one
// workgroup per 64 rows), to learn what bounds its 8.6 GB of stores: k_pipe with the arithmetic switched off still
// needs 1.93 ms (4.45 TB/s) where its LDS hand-over alone takes 1.21 ms (profiles/NOTES_r04.md).
//   CH   bytes one workgroup writes to one row before it moves to the next row set (k_pipe: 128 = one 16-sample tile)
//   W    waves per workgroup that store (k_pipe: 1)
//   NT   non-temporal policy
// Every variant writes the same 8 GiB once; 256 workgroups (one per CU) x 64 rows each, like the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void st16(d2 *p, d2 v, bool nt) {
  if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// A workgroup owns rows [64 g, 64 g + 64).  Per step it writes CH bytes to each of its 64 rows: a store instruction
// covers 1 KiB = (1024 / CH) rows x CH bytes when CH <= 1024, else one row's 1 KiB piece.
template <int CH, int W, bool NT>
__global__ __launch_bounds__(64 * W) void k_wpat(d2 *__restrict__ y, size_t row_d2, int sleep) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t row0 = (size_t)blockIdx.x * 64;
  const d2 v = {1.0, 2.0};
  constexpr int LPR = CH >= 1024 ? 64 : CH / 16;          // lanes per row in one instruction
  constexpr int RPI = 64 / LPR;                            // rows per instruction
  constexpr int IPS = 64 * CH / 1024;                      // instructions per step (64 rows x CH bytes)
  const size_t steps = row_d2 * 16 / CH;
  for (size_t t = wave; t < steps; t += W) {
#pragma unroll 8
    for (int i = 0; i < IPS; ++i) {
      size_t row, off;
      if (CH >= 1024) { row = i / (CH / 1024); off = (size_t)(i % (CH / 1024)) * 64 + lane; }
      else { row = (size_t)i * RPI + lane / LPR; off = lane % LPR; }
      st16(y + (row0 + row) * row_d2 + t * (CH / 16) + off, v, NT);
    }
    if (sleep) __builtin_amdgcn_s_sleep(8);
  }
}

int main() {
  const size_t rows = 16384, row_d2 = 65536 / 2, bytes = rows * row_d2 * 16;
  d2 *y;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  CK(hipMalloc(&y, bytes));
  CK(hipMemset(y, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(label, ...) { __VA_ARGS__; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int it = 0; it < 5; ++it) { __VA_ARGS__; } \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-58s %7.3f ms  %6.2f TB/s\n", label, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12); }
#define V(CH, W, NT) RUN("chunk " #CH " B/row, " #W " storing wave(s)" #NT, hipLaunchKernelGGL((k_wpat<CH, W, NT>), dim3(256), dim3(64 * W), 0, 0, y, row_d2, 0))
  V(128, 1, true) V(128, 1, false) V(128, 2, true) V(128, 4, true)
  V(256, 1, true) V(256, 2, true) V(256, 4, true)
  V(512, 1, true) V(512, 2, true) V(512, 4, true)
  V(1024, 1, true) V(1024, 4, true) V(4096, 1, true) V(4096, 4, true) V(4096, 4, false)
  CK(hipMemsetAsync(y, 0, bytes)); 
  RUN("hipMemset of the same 8 GiB", CK(hipMemsetAsync(y, 0, bytes)));
  return 0;
}
