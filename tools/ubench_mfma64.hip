// ubench_mfma64.hip -- does v_mfma_f64_16x16x4_f64 add FP64 throughput ON TOP of v_fma_f64 on gfx950?
//
// cfg3 (256-tap FIR) is FP64-issue-bound; the FMA form of k_fir_ring reaches ~42 TFLOP/s at a clock
// the chip lets sag to ~1.8 GHz.  The matrix pipe is separate from the VALU, so a Toeplitz-form FIR
// on v_mfma_f64_16x16x4_f64 running next to the VALU kernel could in principle add to it.  Before
// building that kernel, measure the ceiling: whole-chip FP64 rate of (a) VALU FMA only, (b) MFMA
// only, (c) both at once (half of the waves of every SIMD on each pipe).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma64.hip -o tools/ubench_mfma64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: every wave VALU; 1: every wave MFMA; 2: even waves VALU, odd waves MFMA
__global__ __launch_bounds__(256) void k_mix(double *out, int iters, int mode, double a, double b) {
  const int wave = threadIdx.x >> 6;
  const bool mfma = mode == 1 || (mode == 2 && (wave & 1));
  double r = 0;
  if (!mfma) {
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 1.0 + threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = __builtin_fma(v[c], a, b);     // 16 independent chains
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) r += v[c];
  } else {
    d4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (d4){0, 0, 0, 0};
    const double av = a + threadIdx.x * 1e-6, bv = b + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) r += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  double *out;
  const int blocks = 256 * 2, threads = 256;       // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  CK(hipMalloc(&out, (size_t)blocks * threads * sizeof(double)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 40000;
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      k_mix<<<blocks, threads>>>(out, 100, mode, 0.999, 0.001);
      CK(hipEventRecord(e0));
      k_mix<<<blocks, threads>>>(out, iters, mode, 0.999, 0.001);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double waves = (double)blocks * 4;
      // flops per wave per iteration: VALU 64 fma x 64 lanes x 2; MFMA 4 x (16 x 16 x 4 x 2)
      const double valu_w = mode == 0 ? waves : mode == 2 ? waves / 2 : 0, mfma_w = waves - valu_w;
      const double flops = (double)iters * (valu_w * 64 * 64 * 2 + mfma_w * 4 * 2048);
      printf("mode %d (%s): %.2f ms, %.1f TFLOP/s (VALU part %.1f, MFMA part %.1f)\n", mode,
             mode == 0 ? "VALU v_fma_f64 only" : mode == 1 ? "MFMA f64 16x16x4 only" : "half the waves on each pipe",
             ms, flops / ms / 1e9, iters * valu_w * 64 * 64 * 2 / ms / 1e9, iters * mfma_w * 4 * 2048 / ms / 1e9);
    }
  return 0;
}
