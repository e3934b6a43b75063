// ubench_biquad.hip -- how fast can one wave run the bit-exact DF-I biquad step from registers?
// Variants:  0 = source order  t1,t2,add,t3,add,t4,add   (what hipcc emits for k_wave)
//            1 = FF for 8 steps first, then the 4-op recurrence per step
//            2 = recurrence interleaved with the next chunk's FF ops by hand (sched_barrier pinned)
//            3 = variant 0 with SGPR (wave-uniform) coefficients
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_biquad.hip -o tools/ubench_biquad
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int V>
__global__ void k(double *out, const double *coef, int iters, double sb0, double sb2, double sa1, double sa2) {
  const int lane = threadIdx.x;
  double b0 = coef[lane], b2 = coef[64 + lane], na1 = coef[128 + lane], na2 = coef[192 + lane];
  if (V == 3) { b0 = sb0; b2 = sb2; na1 = sa1; na2 = sa2; }
  double x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = 0.01 * (u + 1) + lane * 1e-4;
  double d1 = 0, d2 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < iters; ++i) {
    if (V == 0 || V == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double d0 = x[u];
        double acc = b0 * d0;
        acc = acc + b2 * d2;
        acc = acc + na1 * m1;
        acc = acc + na2 * m2;
        m2 = m1; m1 = acc; d2 = d1; d1 = d0;
        x[u] = acc;   // feed back so nothing is loop invariant
      }
    } else if (V == 1) {
      double p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double d0 = x[u];
        p[u] = b0 * d0 + b2 * d2;
        d2 = d1; d1 = d0;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double t3 = na1 * m1;
        const double t4 = na2 * m2;
        const double s1 = p[u] + t3;
        const double y = s1 + t4;
        m2 = m1; m1 = y;
        x[u] = y;
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // V == 2: p[] for this chunk was computed during the previous chunk's recurrence
      static_assert(V <= 3, "");
      double p[8], pn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { p[u] = b0 * x[u] + b2 * d2; d2 = d1; d1 = x[u]; }
      for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double t3 = na1 * m1;
          __builtin_amdgcn_sched_barrier(0);
          const double f1 = b0 * x[u];             // filler: next chunk's FF (uses stale x: timing only)
          __builtin_amdgcn_sched_barrier(0);
          const double s1 = p[u] + t3;
          __builtin_amdgcn_sched_barrier(0);
          const double f2 = b2 * d2;
          __builtin_amdgcn_sched_barrier(0);
          const double t4 = na2 * m2;
          __builtin_amdgcn_sched_barrier(0);
          pn[u] = f2 + f1;
          __builtin_amdgcn_sched_barrier(0);
          const double y = s1 + t4;
          __builtin_amdgcn_sched_barrier(0);
          m2 = m1; m1 = y; d2 = d1; d1 = y;
          x[u] = y;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = pn[u];
      }
      i += 3;
    }
  }
  out[blockIdx.x * 64 + lane] = m1 + m2 + d1 + d2 + x[3];
}

template <int V>
void run(int lanes_note) {
  const int blocks = 256, iters = 100000;
  double *out, *coef;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipMalloc(&coef, 256 * sizeof(double)));
  double h[256];
  for (int i = 0; i < 64; ++i) { h[i] = 0.01; h[64 + i] = -0.01; h[128 + i] = 1.2 + i * 1e-3; h[192 + i] = -0.5; }
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<V><<<blocks, 64>>>(out, coef, 100, 0.01, -0.01, 1.2, -0.5);
  CK(hipEventRecord(e0));
  k<V><<<blocks, 64>>>(out, coef, iters, 0.01, -0.01, 1.2, -0.5);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double steps = (double)iters * 8;
  printf("variant %d: %.3f ms, %.2f ns/step = %.1f cycles/step @2.4GHz\n", V, ms, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
}

int main() {
  run<0>(0); run<1>(0); run<2>(0); run<3>(0);
  run<0>(0); run<1>(0); run<2>(0); run<3>(0);
  return 0;
}
