// ubench_loop.hip -- the k_wave tile loop in isolation (no DMA, no global stores): where do the
// cycles go?  Flags: RD = ds_read x from LDS (else registers), WR = ds_write y, GHOST = ghost lanes
// write to dummy slots (else all lanes write the same slot), PINNED = hand-pinned interleave.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_loop.hip -o tools/ubench_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define PIN(v) asm volatile("" : "+v"(v))

template <bool RD, int WR, bool GHOST, bool PINNED, bool FF = true>
__global__ __launch_bounds__(64) void k(double *out, const double *coef, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 16, T = 64, NCH = 8, SLOT = 8192 + 128;
  const int lane = threadIdx.x, cl = lane & 15;
  double b0 = coef[cl], b2 = coef[64 + cl], na1 = coef[128 + cl], na2 = coef[192 + cl];
  for (int i = lane; i < 4 * SLOT / 8; i += 64) reinterpret_cast<double *>(smem)[i] = 0.001 * (i % 97);
  __syncthreads();
  double d1 = 0, d2 = 0, m1 = 0, m2 = 0;
  asm volatile("" : "+v"(b0), "+v"(b2), "+v"(na1), "+v"(na2));
  for (int t = 0; t < tiles; ++t) {
    char *tile = smem + (t & 3) * SLOT;
    const char *rd = tile + cl * 8;
    char *wr = (WR >= 2) ? tile + cl * 8 + (lane / G) * 128 : ((lane < G || !GHOST) ? tile : smem + (3 + lane / G) * SLOT) + cl * 8;
#define EOFF(u) ((u) * G * 8 + (((u) * G) >> 7) * 16)
    double xr[3][8], pp[2][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xr[0][u] = RD ? *reinterpret_cast<const double *>(rd + EOFF(u)) : m2 + u;
#pragma unroll
    for (int u = 0; u < 8; ++u) xr[1][u] = RD ? *reinterpret_cast<const double *>(rd + EOFF(8 + u)) : m1 + u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double x0 = xr[0][u], x2 = u >= 2 ? xr[0][u - 2] : (u == 1 ? d1 : d2);
      pp[0][u] = b0 * x0 + b2 * x2;
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int xc = k % 3, xn = (k + 1) % 3, xl = (k + 2) % 3;
      if (k + 2 < NCH) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          xr[xl][u] = RD ? *reinterpret_cast<const double *>(rd + EOFF((k + 2) * 8 + u)) : m2 + u;
      }
      __builtin_amdgcn_sched_barrier(0);
      double yv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ff = k + 1 < NCH;
        const double x0 = xr[xn][u], x2 = u >= 2 ? xr[xn][u - 2] : xr[xc][6 + u];
        double fa = 0, fb = 0, acc = pp[k & 1][u];
        if (PINNED) {
          double t3 = na1 * m1; PIN(t3);
          if (ff) { fa = b0 * x0; PIN(fa); }
          acc = acc + t3; PIN(acc);
          if (ff) { fb = b2 * x2; PIN(fb); }
          double t4 = na2 * m2; PIN(t4);
          if (ff) { double pn = fa + fb; PIN(pn); pp[(k + 1) & 1][u] = pn; }
          acc = acc + t4; PIN(acc);
        } else {
          if (ff) pp[(k + 1) & 1][u] = FF ? b0 * x0 + b2 * x2 : x0;
          acc = (acc + na1 * m1) + na2 * m2;
        }
        yv[u] = acc;
        if (WR == 1) *reinterpret_cast<double *>(wr + EOFF(k * 8 + u)) = acc;
        if (WR == 3 && (u & 3) == 3) {
          // row-select with DPP row_mask moves: rows r of yw take y[u-3+r]
          unsigned lo = __double2loint(yv[u - 3]), hi = __double2hiint(yv[u - 3]);
          lo = __builtin_amdgcn_update_dpp(lo, __double2loint(yv[u - 2]), 0xE4, 0x2, 0xF, false);
          hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(yv[u - 2]), 0xE4, 0x2, 0xF, false);
          lo = __builtin_amdgcn_update_dpp(lo, __double2loint(yv[u - 1]), 0xE4, 0x4, 0xF, false);
          hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(yv[u - 1]), 0xE4, 0x4, 0xF, false);
          lo = __builtin_amdgcn_update_dpp(lo, __double2loint(yv[u]), 0xE4, 0x8, 0xF, false);
          hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(yv[u]), 0xE4, 0x8, 0xF, false);
          *reinterpret_cast<double *>(wr + EOFF(k * 8 + u - 3)) = __hiloint2double(hi, lo);
        }
        if (WR == 2 && (u & 3) == 3) {
          double yw = yv[u - 3];
#pragma unroll
          for (int r = 1; r < 4; ++r) yw = (lane / G == r) ? yv[u - 3 + r] : yw;
          *reinterpret_cast<double *>(wr + EOFF(k * 8 + u - 3)) = yw;
        }
        m2 = m1; m1 = acc;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    d1 = xr[(NCH - 1) % 3][7]; d2 = xr[(NCH - 1) % 3][6];
  }
  out[blockIdx.x * 64 + lane] = m1 + m2 + d1 + d2;
}

template <bool RD, int WR, bool GHOST, bool PINNED, bool FF = true>
void run() {
  const int blocks = 256, tiles = 4000;
  double *out, *coef;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipMalloc(&coef, 256 * sizeof(double)));
  double h[256];
  for (int i = 0; i < 64; ++i) { h[i] = 0.01; h[64 + i] = -0.01; h[128 + i] = 1.2 + i * 1e-3; h[192 + i] = -0.5; }
  CK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  auto fn = k<RD, WR, GHOST, PINNED, FF>;
  CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  fn<<<blocks, 64, 96 * 1024>>>(out, coef, 10);
  CK(hipEventRecord(e0));
  fn<<<blocks, 64, 96 * 1024>>>(out, coef, tiles);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double steps = (double)tiles * 64;
  printf("FF=%d RD=%d WR=%d GHOST=%d PINNED=%d: %.2f ns/step = %.1f cycles/step @2.4GHz\n", (int)FF, RD, WR, GHOST, PINNED,
         ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
}

int main() {
  run<true, 3, true, false, false>();
  run<true, 3, true, false, true>();
  run<true, 2, true, false, false>();
  run<true, 0, true, false, false>();
  run<false, 0, true, false, false>();
  run<false, 0, false, false>();
  run<true, 0, false, false>();
  run<true, 1, true, false>();
  run<true, 2, true, false>();
  run<false, 2, true, false>();
  run<true, 2, true, true>();
  return 0;
}
