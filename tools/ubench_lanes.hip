// ubench_lanes.hip -- can the four sections of a gammatone cascade live in the four 16-lane groups of
// ONE wave (16 channels per wave, section q in lanes 16q..16q+15, skewed D steps in time), handing
// samples from group to group with ds_bpermute_b32 every step?  Measures cycles per step of a lone
// wave per SIMD for the step body: 7 f64 ops (b0 x + b1 x1, + (-a1) y1, + (-a2) y2) + 2 ds_bpermute
// + the group-0 input select, for skews D = 1, 2, 4, 8 (the received value is used D steps later).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_lanes.hip -o tools/ubench_lanes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double pull_prev_group(double v, int src_byte) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(src_byte, (int)b);
  const int hi = __builtin_amdgcn_ds_bpermute(src_byte, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int D, bool XLANE>
__global__ __launch_bounds__(64) void k_lanes(double *out, long long *cyc, int iters, double b0, double b1, double na1, double na2) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x;
  const bool first = lane < 16;
  const int src = ((lane + 48) & 63) * 4;          // the lane 16 below (group q pulls from group q - 1)
  double m1 = 0.1 * lane, m2 = 0.05, xp = 0.0, x = 1.0 + lane * 1e-3;
  double fifo[D];
#pragma unroll
  for (int d = 0; d < D; ++d) fifo[d] = 0.0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      x = x * 0.999 + 1e-3;                          // stands for the LDS read of the next input sample
      const double in = first ? x : fifo[d];         // group 0 filters x, the others what they received D steps ago
      const double p = b0 * in + b1 * xp;
      const double y = (p + na1 * m1) + na2 * m2;
      m2 = m1; m1 = y; xp = in;
      fifo[d] = XLANE ? pull_prev_group(y, src) : y * 1.0000001;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = m1 + fifo[0];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int D, bool XLANE>
void run() {
  double *out; long long *cyc;
  const int blocks = 1024, iters = 20000;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipMalloc(&cyc, blocks * sizeof(long long)));
  CK(hipFuncSetAttribute((const void *)k_lanes<D, XLANE>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_lanes<D, XLANE><<<blocks, 64, 40 * 1024>>>(out, cyc, 100, 0.3, 0.2, 0.9, -0.5);
  CK(hipEventRecord(e0));
  k_lanes<D, XLANE><<<blocks, 64, 40 * 1024>>>(out, cyc, iters, 0.3, 0.2, 0.9, -0.5);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double steps = (double)iters * D;
  printf("skew D=%d %s: %.3f ms for %d steps -> %.1f ns/step = %.1f cycles at 2.4 GHz; 1024 waves x 16 channels -> %.0f Gsamples/s\n",
         D, XLANE ? "ds_bpermute hand-over" : "no cross-lane op (floor)", ms, (int)steps, ms * 1e6 / steps, ms * 1e6 / steps * 2.4,
         1024.0 * 16 * steps / (ms * 1e-3) / 1e9);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  run<4, false>();
  run<1, true>(); run<2, true>(); run<4, true>(); run<8, true>();
  run<4, false>();
  return 0;
}
