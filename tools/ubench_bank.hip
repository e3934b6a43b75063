// ubench_bank.hip (round 6) -- VGPR bank conflicts of FP64 operands.  A 64-bit operand v[n:n+1] occupies two of the four VGPR banks
// (n % 4, n % 4 + 1); v_mul_f64 / v_add_f64 with both sources in the same bank pair cost a lone wave ~5.1 cycles instead of ~4.1.
// The recurrence chunk of k_duo as hipcc allocated it (transcribed from the shipped ISA) against a bank-aware allocation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59"
template <int KIND>
__global__ __launch_bounds__(64) void k(long long *cyc, int iters) {
  asm volatile("v_mov_b32 v0, 0\n v_mov_b32 v1, 0x3ff00000\n v_mov_b64 v[2:3], v[0:1]\n v_mov_b64 v[4:5], v[0:1]\n v_mov_b64 v[6:7], v[0:1]\n v_mov_b64 v[8:9], v[0:1]\n"
               "v_mov_b64 v[10:11], v[0:1]\n v_mov_b64 v[12:13], v[0:1]\n v_mov_b64 v[14:15], v[0:1]\n v_mov_b64 v[16:17], v[0:1]\n v_mov_b64 v[18:19], v[0:1]\n"
               "v_mov_b64 v[20:21], v[0:1]\n v_mov_b64 v[22:23], v[0:1]\n v_mov_b64 v[24:25], v[0:1]\n v_mov_b64 v[26:27], v[0:1]\n v_mov_b64 v[28:29], v[0:1]\n"
               "v_mov_b64 v[30:31], v[0:1]\n v_mov_b64 v[32:33], v[0:1]\n v_mov_b64 v[34:35], v[0:1]\n v_mov_b64 v[36:37], v[0:1]\n v_mov_b64 v[38:39], v[0:1]\n" ::: CLOB);
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {
      // two steps as hipcc allocated them (k_duo<false,5,3,...,NT>, shipped library): na1 = v[0:1], na2 = v[2:3]
      asm volatile(".rept 32\n"
                   "v_mul_f64 v[22:23], v[0:1], v[20:21]\n v_mul_f64 v[18:19], v[2:3], v[18:19]\n v_add_f64 v[22:23], v[22:23], v[24:25]\n v_add_f64 v[18:19], v[18:19], v[22:23]\n"
                   "v_mul_f64 v[22:23], v[0:1], v[18:19]\n v_mul_f64 v[20:21], v[2:3], v[20:21]\n v_add_f64 v[22:23], v[22:23], v[26:27]\n v_add_f64 v[20:21], v[20:21], v[22:23]\n"
                   ".endr" ::: CLOB);
    } else if constexpr (KIND == 1) {
      // bank-aware: y (v[6:7] / v[10:11]: H), na1 = v[0:1], na2 = v[4:5] (L); even step: p in L (v[12:13]), t1 in H (v[14:15]), t2 in L;
      // odd step: p in H (v[18:19]), t1 in L (v[16:17]), t2 in H
      asm volatile(".rept 32\n"
                   "v_mul_f64 v[14:15], v[0:1], v[6:7]\n v_mul_f64 v[22:23], v[4:5], v[6:7]\n v_add_f64 v[14:15], v[14:15], v[12:13]\n v_add_f64 v[10:11], v[20:21], v[14:15]\n"
                   "v_mul_f64 v[16:17], v[0:1], v[10:11]\n v_mul_f64 v[20:21], v[4:5], v[10:11]\n v_add_f64 v[16:17], v[16:17], v[18:19]\n v_add_f64 v[6:7], v[22:23], v[16:17]\n"
                   ".endr" ::: CLOB);
    } else if constexpr (KIND == 2) {
      // every source pair in the SAME bank pair (worst case)
      asm volatile(".rept 32\n"
                   "v_mul_f64 v[8:9], v[0:1], v[4:5]\n v_mul_f64 v[12:13], v[0:1], v[4:5]\n v_add_f64 v[8:9], v[8:9], v[16:17]\n v_add_f64 v[4:5], v[12:13], v[8:9]\n"
                   "v_mul_f64 v[8:9], v[0:1], v[4:5]\n v_mul_f64 v[12:13], v[0:1], v[4:5]\n v_add_f64 v[8:9], v[8:9], v[16:17]\n v_add_f64 v[4:5], v[12:13], v[8:9]\n"
                   ".endr" ::: CLOB);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(const char *name) {
  long long *cyc; const int waves = 1024, iters = 4000; CK(hipMalloc(&cyc, waves * 8));
  k<KIND><<<waves, 64>>>(cyc, 4); CK(hipDeviceSynchronize()); k<KIND><<<waves, 64>>>(cyc, iters); CK(hipDeviceSynchronize());
  long long h[8]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  printf("%-80s %.2f cycles per instruction, %.1f per recurrence step\n", name, (double)h[3] / (iters * 256.0), (double)h[3] / (iters * 64.0));
}
int main() {
  run<0>("recurrence steps with hipcc's registers (shipped k_duo)");
  run<1>("the same steps, sources of every instruction in different bank pairs");
  run<2>("the same steps, sources of every instruction in the SAME bank pair");
  return 0;
}
