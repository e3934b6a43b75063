// ubench_ifetch.hip (round 6) -- does the LENGTH of a lone wave's instruction stream set its issue rate?
// One wave per SIMD (1024 waves) or two; a loop whose body is BODY independent FP64 instructions (8 rotating chains, so no operand
// is younger than 8 instructions) in three encodings: v_mul_f64 (VOP3, 8 bytes), v_fma_f64 (VOP3, 8 bytes), v_fmac_f64_e32 (VOP2, 4 bytes).
// Prints cycles per instruction per wave.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ifetch.hip -o /tmp/ubench_ifetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define STR2(x) #x
#define STR(x) STR2(x)
// the body as one asm block: REPT x 8 instructions on eight registers (the unroller caps a C++ loop at 1024 copies)
#define BODY_ASM(INS, REPT)                                                                                          \
  asm volatile(".rept " STR(REPT) "\n\t" INS " %0, %0, %8 \n\t" INS " %1, %1, %8 \n\t" INS " %2, %2, %8 \n\t" INS " %3, %3, %8 \n\t"          \
               INS " %4, %4, %8 \n\t" INS " %5, %5, %8 \n\t" INS " %6, %6, %8 \n\t" INS " %7, %7, %8 \n\t.endr"                         \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a))
#define BODY_FMA(REPT)                                                                                               \
  asm volatile(".rept " STR(REPT) "\n\t v_fma_f64 %0, %0, %8, %8 \n\t v_fma_f64 %1, %1, %8, %8 \n\t v_fma_f64 %2, %2, %8, %8 \n\t v_fma_f64 %3, %3, %8, %8 \n\t" \
               " v_fma_f64 %4, %4, %8, %8 \n\t v_fma_f64 %5, %5, %8, %8 \n\t v_fma_f64 %6, %6, %8, %8 \n\t v_fma_f64 %7, %7, %8, %8 \n\t.endr"                 \
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a))
template <int KIND, int BODY>
__global__ __launch_bounds__(64) void k_body(double *out, long long *cycles, int iters, double a) {
  double v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = 1.0 + threadIdx.x * 1e-3 + c;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#define ONE(R)                                                                     \
    if constexpr (BODY == 8 * R) {                                                 \
      if constexpr (KIND == 0) BODY_ASM("v_mul_f64", R);                           \
      else if constexpr (KIND == 1) BODY_FMA(R);                                   \
      else BODY_ASM("v_fmac_f64_e32", R);                                          \
    }
    ONE(4) ONE(64) ONE(512) ONE(1024)
#undef ONE
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += v[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int BODY>
void run(int waves, const char *name) {
  double *out; long long *cyc;
  CK(hipMalloc(&out, waves * 64 * sizeof(double)));
  CK(hipMalloc(&cyc, waves * sizeof(long long)));
  const int iters = (1 << 22) / BODY;
  k_body<KIND, BODY><<<waves, 64>>>(out, cyc, 4, 1.0000001);
  CK(hipDeviceSynchronize());
  k_body<KIND, BODY><<<waves, 64>>>(out, cyc, iters, 1.0000001);
  CK(hipDeviceSynchronize());
  long long h[8]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  printf("%-22s body %5d instructions (%6d bytes), %4d waves: %.2f cycles per instruction per wave\n", name, BODY,
         BODY * (KIND == 2 ? 4 : 8), waves, (double)h[3] / ((double)iters * BODY));
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  for (int waves : {1024, 2048}) {
    run<0, 32>(waves, "v_mul_f64 (8 B)");   run<0, 512>(waves, "v_mul_f64 (8 B)");   run<0, 4096>(waves, "v_mul_f64 (8 B)");   run<0, 8192>(waves, "v_mul_f64 (8 B)");
    run<1, 32>(waves, "v_fma_f64 (8 B)");   run<1, 512>(waves, "v_fma_f64 (8 B)");   run<1, 4096>(waves, "v_fma_f64 (8 B)"); run<1, 8192>(waves, "v_fma_f64 (8 B)");
    run<2, 32>(waves, "v_fmac_f64_e32 (4 B)"); run<2, 512>(waves, "v_fmac_f64_e32 (4 B)"); run<2, 4096>(waves, "v_fmac_f64_e32 (4 B)"); run<2, 8192>(waves, "v_fmac_f64_e32 (4 B)");
  }
  return 0;
}
