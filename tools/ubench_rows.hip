// ubench_rows.hip -- timing prototypes for the recurrence wave's data movement (gfx950):
//   V0: registers only, compiler order (baseline, = ubench_biquad V0)
//   V5: V0 + one ds_read_b64 and one ds_write_b64 per step, ACTIVE lanes only (16 or 64)
//   V7: "rows as time": 64-lane ds_read_b64 per 4 steps, FF on all lanes, permlane swaps to feed row 0,
//       4 recurrence steps, permlane assembly, 64-lane ds_write_b64 per 4 steps
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_rows.hip -o tools/ubench_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double swap16_src(double vdst, double vsrc, double *vdst_out) {
  unsigned dl = __double2loint(vdst), dh = __double2hiint(vdst), sl = __double2loint(vsrc), sh = __double2hiint(vsrc);
  auto r0 = __builtin_amdgcn_permlane16_swap(dl, sl, false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(dh, sh, false, false);
  *vdst_out = __hiloint2double(r1[0], r0[0]);
  return __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ double swap32_src(double vdst, double vsrc, double *vdst_out) {
  unsigned dl = __double2loint(vdst), dh = __double2hiint(vdst), sl = __double2loint(vsrc), sh = __double2hiint(vsrc);
  auto r0 = __builtin_amdgcn_permlane32_swap(dl, sl, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(dh, sh, false, false);
  *vdst_out = __hiloint2double(r1[0], r0[0]);
  return __hiloint2double(r1[1], r0[1]);
}

template <int V, int ACTIVE>
__global__ __launch_bounds__(64) void k(double *out, const double *coef, int iters) {
  __shared__ double lds[64 * 64 + 64];
  const int lane = threadIdx.x;
  const double b0 = coef[lane & 15], b2 = coef[64 + (lane & 15)], na1 = coef[128 + (lane & 15)], na2 = coef[192 + (lane & 15)];
  for (int i = lane; i < 64 * 64; i += 64) lds[i] = 0.001 * (i % 97);
  __syncthreads();
  double d1 = 0, d2 = 0, m1 = 0, m2 = 0;
  if (V == 0 || V == 5) {
    if (lane < ACTIVE) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {           // 8 chunks of 8 steps = one 64-row tile
          double xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) xv[u] = (V == 5) ? lds[(c * 8 + u) * 16 + (lane & 15) + (lane >> 4) * 1024] : 0.01 * u + m2;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const double d0 = xv[u];
            double acc = b0 * d0;
            acc = acc + b2 * d2;
            acc = acc + na1 * m1;
            acc = acc + na2 * m2;
            m2 = m1; m1 = acc; d2 = d1; d1 = d0;
            if (V == 5) lds[(c * 8 + u) * 16 + (lane & 15) + (lane >> 4) * 1024] = acc;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    // V7
    double xprev = 0.0;
    const bool lo_half = lane < 32;
    for (int i = 0; i < iters; ++i) {
      double X = lds[lane];
#pragma unroll
      for (int g = 0; g < 16; ++g) {             // 16 groups of 4 rows = one 64-row tile
        const double Xn = lds[((g + 1) & 15) * 64 + lane];   // prefetch next group
        double a_out, junk;
        const double b_out = swap32_src(xprev, X, &a_out);     // a_out=[prev.lo;cur.lo]  b_out=[prev.hi;cur.hi]
        const double xd2 = lo_half ? b_out : a_out;            // [prev rows 2,3 ; cur rows 0,1]
        double P = b0 * X + b2 * xd2;
        // distribute p1..p3 to row 0
        double Pm;                                             // P with odd rows replaced (rows 0,2 intact)
        const double Q1 = swap16_src(P, P, &Pm);               // Q1 even rows = P odd rows: row0=p1,row2=p3
        const double Q2 = swap32_src(Pm, Pm, &junk);           // Q2[0..31] = Pm[32..63]: row0 = p2
        const double Q3 = swap32_src(Q1, Q1, &junk);           // row0 = p3
        double y0, y1, y2, y3;
        { const double t3 = na1 * m1, t4 = na2 * m2; y0 = (P + t3) + t4; m2 = m1; m1 = y0; }
        { const double t3 = na1 * m1, t4 = na2 * m2; y1 = (Q1 + t3) + t4; m2 = m1; m1 = y1; }
        { const double t3 = na1 * m1, t4 = na2 * m2; y2 = (Q2 + t3) + t4; m2 = m1; m1 = y2; }
        { const double t3 = na1 * m1, t4 = na2 * m2; y3 = (Q3 + t3) + t4; m2 = m1; m1 = y3; }
        // assemble Y = [y0,y1,y2,y3]
        double A, C2, Y;
        swap16_src(y0, y1, &A);                                // A = [y0, y1, ., .]
        swap16_src(y2, y3, &C2);                               // C2 = [y2, y3, ., .]
        swap32_src(A, C2, &Y);                                 // Y[32..63] = C2[0..31]
        lds[g * 64 + lane] = Y;
        xprev = X;
        X = Xn;
      }
    }
    d1 = xprev;
  }
  out[blockIdx.x * 64 + lane] = m1 + m2 + d1 + d2;
}

template <int V, int ACTIVE>
void run() {
  const int blocks = 256, iters = 20000;
  double *out, *coef;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipMalloc(&coef, 256 * sizeof(double)));
  double h[256];
  for (int i = 0; i < 64; ++i) { h[i] = 0.01; h[64 + i] = -0.01; h[128 + i] = 1.2 + i * 1e-3; h[192 + i] = -0.5; }
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<V, ACTIVE><<<blocks, 64>>>(out, coef, 100);
  CK(hipEventRecord(e0));
  k<V, ACTIVE><<<blocks, 64>>>(out, coef, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double steps = (double)iters * 64;
  printf("V%d active=%2d: %.3f ms, %.2f ns/step = %.1f cycles/step @2.4GHz\n", V, ACTIVE, ms, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 16>(); run<0, 64>(); run<5, 16>(); run<5, 64>(); run<7, 64>();
  }
  return 0;
}
