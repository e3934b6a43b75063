// ubench_f64.hip -- gfx950 micro-benchmarks that size the DF-I kernel design:
//   (1) cycles per v_mul_f64/v_add_f64 pair for a DEPENDENT chain and for 4 independent
//       chains, with 64 / 32 / 16 / 8 active lanes (does the SIMD skip idle quarter-waves?)
//   (2) the same with 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_f64.hip -o /tmp/ubench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int CHAINS>
__global__ void k_chain(double *out, long long *cycles, int active, int iters, double a, double b) {
  const int lane = threadIdx.x & 63;
  double v[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) v[c] = 1.0 + lane * 1e-3 + c;
  long long t0 = 0, t1 = 0;
  if (lane < active) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) v[c] = v[c] * a + b;   // mul then add, unfused
      }
    }
    t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  }
}

template <int CHAINS>
void run(int active, int waves_per_block, int iters) {
  double *out; long long *cyc;
  const int blocks = 256;
  CK(hipMalloc(&out, blocks * waves_per_block * 64 * sizeof(double)));
  CK(hipMalloc(&cyc, blocks * sizeof(long long)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_chain<CHAINS><<<blocks, waves_per_block * 64>>>(out, cyc, active, 10, 0.999, 0.001);
  CK(hipEventRecord(e0));
  k_chain<CHAINS><<<blocks, waves_per_block * 64>>>(out, cyc, active, iters, 0.999, 0.001);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double ops = (double)iters * 8 * CHAINS;  // mul+add pairs per lane
  printf("chains=%d active=%2d waves/block=%d : %.2f ms, %.2f clk(100MHz-ticks?)/pair, wall %.2f ns/pair/wave\n",
         CHAINS, active, waves_per_block, ms, (double)h[0] / ops, ms * 1e6 / ops);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  const int iters = 200000;
  for (int wpb : {1, 4, 8, 16}) {
    for (int active : {64, 32, 16, 8}) {
      run<1>(active, wpb, iters);
      run<4>(active, wpb, iters);
    }
  }
  return 0;
}
