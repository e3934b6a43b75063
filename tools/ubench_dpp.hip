// ubench_dpp.hip -- what does k_quad's step cost a lone wave, and which part of it?  One wave per SIMD (1024 blocks),
// the step of alz_quad.inc in the same pinned order: 7 FP64 operations + the hand-over, in several forms:
//   0  no hand-over (the input is a register)              1  two plain v_mov_b32 instead of the DPP moves
//   2  v_mov_b32_dpp row_shr:4 bank_mask:0xe of the value finished in the PREVIOUS step (k_quad as first built)
//   3  the same of the value finished TWO steps ago         4  row_shr:4 without a write mask (bank_mask:0xf, bound_ctrl)
//   5  quad_perm instead of row_shr                        6  the DPP moves at the END of the step (after the last add)
//   7  FMA form (4 operations) with the moves of form 2     8  FMA form without hand-over
//   9  form 3 but the moved value goes through a v_mov_b64 copy first (does a non-DP reader of a DP result wait?)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_dpp.hip -o tools/ubench_dpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int CTRL, int BANK, bool BC>
__device__ __forceinline__ double pull(double old, double src) {
  const long long o = __double_as_longlong(old), v = __double_as_longlong(src);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)v, CTRL, 0xf, BANK, BC);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(v >> 32), CTRL, 0xf, BANK, BC);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double plain_mov(double src) {
  long long v = __double_as_longlong(src);
  int lo = (int)v, hi = (int)(v >> 32), a, b;
  asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(lo));
  asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "v"(hi));
  return __longlong_as_double(((long long)b << 32) | (unsigned)a);
}

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int iters, double b0, double b1, double na1, double na2) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x;
  asm volatile("" : "+v"(b0), "+v"(b1), "+v"(na1), "+v"(na2));
  double y1 = 0.1 * lane, y2 = 0.05, inK = 0.3, xraw = 1.0 + lane * 1e-3, inD = 0.7;   // inD: a dead register the move may overwrite (k_quad: the loaded x)
  double pN = b0 * inK + b1 * 0.2, n2N = na2 * y2, y3 = 0.01;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      double y, inN;
      if constexpr (MODE == 7 || MODE == 8) {
        const double a = __builtin_fma(na1, y1, pN); PIN();
        if (MODE == 7) inN = pull<0x114, 0xe, false>(inD, y1); else inN = xraw;
        PIN();
        double q = b0 * inN; PIN();
        q = __builtin_fma(b1, inK, q); PIN();
        y = __builtin_fma(na2, y2, a); PIN();
        pN = q;
      } else {
        const double m = na1 * y1; PIN();
        if (MODE == 0) inN = xraw;
        if (MODE == 1) inN = plain_mov(y1);
        if (MODE == 2) inN = pull<0x114, 0xe, false>(inD, y1);
        if (MODE == 3) inN = pull<0x114, 0xe, false>(inD, y2);
        if (MODE == 4) inN = pull<0x114, 0xf, true>(inD, y1);
        if (MODE == 5) inN = pull<0x90, 0xf, true>(inD, y1);      // quad_perm:[0,0,1,2]
        if (MODE == 9) { double c = y2; asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "v"(y2)); inN = pull<0x114, 0xe, false>(inD, c); }
        if (MODE == 6) inN = xraw;
        if (MODE == 10) inN = pull<0x114, 0xe, false>(inD, y3);
        if (MODE == 11) { long long v = __double_as_longlong(y2), o = __double_as_longlong(inD); int lo = (int)v, hi = (int)(v >> 32), ol = (int)o, oh = (int)(o >> 32);
          asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(ol), "+v"(lo)); asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(oh), "+v"(hi));
          inN = __longlong_as_double(((long long)oh << 32) | (unsigned)ol); }
        PIN();
        const double a = pN + m; PIN();
        double acc = b0 * inN; PIN();
        const double tq = b1 * inK; PIN();
        y = a + n2N; PIN();
        acc = acc + tq; PIN();
        n2N = na2 * y1; PIN();
        pN = acc;
        if (MODE == 6) { inN = pull<0x114, 0xe, false>(inD, y1); PIN(); pN = pN + inN * 1e-30; }
      }
      y3 = y2; y2 = y1; y1 = y; inD = xraw; xraw = inK; inK = inN;    // (three names rotate: no copies)
      PIN();
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = y1 + pN + n2N;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *what) {
  double *out; long long *cyc;
  const int blocks = 1024, iters = 4000;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipMalloc(&cyc, blocks * sizeof(long long)));
  CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 64, 36 * 1024>>>(out, cyc, 100, 0.3, 0.2, 0.9, -0.5);
  CK(hipEventRecord(e0));
  k<MODE><<<blocks, 64, 36 * 1024>>>(out, cyc, iters, 0.3, 0.2, 0.9, -0.5);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c0; CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
  const double steps = (double)iters * 16;
  printf("mode %d  %-62s %.2f ns/step  (%5.1f ticks of the 100 MHz counter x 21 = cycles at 2.1 GHz: %5.1f)  -> %4.0f Gsamples/s for 1024 waves x 16 channels\n",
         MODE, what, ms * 1e6 / steps, (double)c0 / steps, ms * 1e6 / steps * 2.1, 1024.0 * 16 * steps / (ms * 1e-3) / 1e9);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  run<0>("7 f64, no hand-over");
  run<1>("7 f64 + 2 plain v_mov_b32 of y(t-1)");
  run<2>("7 f64 + 2 dpp row_shr:4 bank_mask:0xe of y(t-1)");
  run<3>("7 f64 + 2 dpp row_shr:4 bank_mask:0xe of y(t-2)");
  run<4>("7 f64 + 2 dpp row_shr:4 unmasked, bound_ctrl, of y(t-1)");
  run<5>("7 f64 + 2 dpp quad_perm of y(t-1)");
  run<6>("7 f64 + 2 dpp at the end of the step (+2 f64)");
  run<10>("7 f64 + 2 dpp row_shr:4 bank_mask:0xe of y(t-3)");
  run<11>("7 f64 + 2 v_permlane16_swap of y(t-2)");
  run<9>("7 f64 + v_mov_b64 copy + 2 dpp of the copy of y(t-2)");
  run<7>("FMA: 4 f64 + 2 dpp of y(t-1)");
  run<8>("FMA: 4 f64, no hand-over");
  run<0>("7 f64, no hand-over (again)");
  return 0;
}
