// ubench_clock.hip -- what clock does a NEARLY IDLE chip run a lone workgroup at?
// configs[3] in the reference's own shape (one signal through 256 bands) is 4 k_pipe workgroups on a 256-CU
// chip, and a stage wave advances 61 ns per step there against 34 ns when 256 workgroups are resident
// (profiles/r02_kernel_dispatches.csv).  This measures the engine clock as shader cycles (s_memtime) per
// constant-rate tick (s_memrealtime, 100 MHz) and the time per dependent FP64 operation, for grids of
// 1, 4, 16, 256 and 1024 workgroups, each running the same dependent mul/add chain for ~20 ms.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_clock.hip -o tools/ubench_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(384) void k(double *out, long long *ticks, int iters, int lds_traffic) {
  __shared__ double buf[6 * 64 * 2];
  const int lane = threadIdx.x;
  double m1 = 0.1 + lane * 1e-6, m2 = 0.2, na1 = 1.2, na2 = -0.5;
  asm volatile("" : "+v"(na1), "+v"(na2));
  buf[lane] = 1e-3;
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();       // s_memtime: shader clock
  const long long r0 = wall_clock64();                     // s_memrealtime: 100 MHz
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double acc = (1e-3 + na1 * m1) + na2 * m2;      // mul -> add -> add: the recurrence's chain
      m2 = m1;
      m1 = acc * 0.5;
    }
    if (lds_traffic) {                                      // one hand-over per 16 steps, like a k_pipe stage
      buf[lane] = m1;
      __syncthreads();
      m2 += buf[(lane + 64) % 384] * 1e-9;
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long r1 = wall_clock64();
  out[blockIdx.x * 384 + lane] = m1 + m2;
  if (lane == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  double *out; long long *ticks;
  CK(hipMalloc(&out, 1024 * 384 * sizeof(double)));
  CK(hipMalloc(&ticks, 2 * 1024 * sizeof(long long)));
  const int iters = 40000;
  for (int lds = 0; lds < 2; ++lds)
    for (int blocks : {1, 4, 16, 256, 1024}) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(384), 0, 0, out, ticks, 1000, lds);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(384), 0, 0, out, ticks, iters, lds);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long h[2];
      CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
      const double steps = 16.0 * iters;
      printf("workgroups %5d (6 waves each) lds_handover %d: %.2f ms, shader clock %.0f MHz (cycles / 100 MHz ticks), "
             "%.1f cycles = %.2f ns per step (3 dependent f64 ops + 1 mul)\n",
             blocks, lds, ms, 100.0 * (double)h[0] / (double)h[1], (double)h[0] / steps, ms * 1e6 / steps);
    }
  return 0;
}
