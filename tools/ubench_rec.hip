// ubench_rec.hip -- the recurrence wave's arithmetic alone: y = (p + na1*m1) + na2*m2, four orderings.
//  V0: source order, compiler schedule   V1: both products pinned before the adds
//  V2: a2 product computed one step ahead (from the current m1) and pinned before the a1 product
//  V3: like V2 but pinned after the a1 product
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_rec.hip -o tools/ubench_rec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define PIN(v) asm volatile("" : "+v"(v))

template <int V>
__global__ __launch_bounds__(64) void k(double *out, const double *coef, int iters) {
  const int lane = threadIdx.x;
  double na1 = coef[128 + (lane & 15)], na2 = coef[192 + (lane & 15)];
  double pv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) pv[u] = 1e-3 * (u + 1) + lane * 1e-5;
  double m1 = 0.1, m2 = 0.2;
  asm volatile("" : "+v"(na1), "+v"(na2));
  double t4n = na2 * m2;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      double acc = pv[u];
      if (V == 0) {
        acc = acc + na1 * m1;
        acc = acc + na2 * m2;
      } else if (V == 1) {
        double t3 = na1 * m1, t4 = na2 * m2;
        PIN(t3); PIN(t4);
        acc = (acc + t3) + t4;
      } else if (V == 2) {
        double t4 = t4n;
        t4n = na2 * m1; PIN(t4n);          // next step's a2 product: its input is already known
        double t3 = na1 * m1; PIN(t3);
        acc = (acc + t3) + t4;
      } else {
        double t4 = t4n;
        double t3 = na1 * m1; PIN(t3);
        t4n = na2 * m1; PIN(t4n);
        acc = (acc + t3) + t4;
      }
      m2 = m1; m1 = acc;
      pv[u] = acc * 1e-3;   // keep inputs changing (one extra independent mul per step, same in all variants)
    }
  }
  out[blockIdx.x * 64 + lane] = m1 + m2 + pv[3];
}

template <int V>
void run() {
  const int blocks = 256, iters = 200000;
  double *out, *coef;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipMalloc(&coef, 256 * sizeof(double)));
  double h[256];
  for (int i = 0; i < 64; ++i) { h[i] = 0.01; h[64 + i] = -0.01; h[128 + i] = 1.2 + i * 1e-3; h[192 + i] = -0.5; }
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<V><<<blocks, 64>>>(out, coef, 100);
  CK(hipEventRecord(e0));
  k<V><<<blocks, 64>>>(out, coef, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double steps = (double)iters * 8;
  printf("V%d: %.2f ns/step = %.1f cycles/step @2.4GHz\n", V, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
}
int main() { for (int r = 0; r < 2; ++r) { run<0>(); run<1>(); run<2>(); run<3>(); } return 0; }
