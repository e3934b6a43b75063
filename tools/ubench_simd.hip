// ubench_simd.hip -- on which SIMD of its CU does wave w of a workgroup land?  (HW_ID: SIMD_ID = bits 5:4, CU_ID = bits 11:8)
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_simd.hip -o tools/ubench_simd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k(int *out, int waves) {
  extern __shared__ char smem[];
  const int w = threadIdx.x >> 6;
  const unsigned id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * waves + w] = (int)id;
  // stay a while so that the whole grid is resident together
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 2000000) {}
}
int main() {
  const int blocks = 256;
  for (int waves : {4, 5, 6, 8}) {
    int *out;
    CK(hipMalloc(&out, blocks * waves * 4));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    k<<<blocks, waves * 64, 100 * 1024>>>(out, waves);
    CK(hipDeviceSynchronize());
    int *h = (int *)malloc(blocks * waves * 4);
    CK(hipMemcpy(h, out, blocks * waves * 4, hipMemcpyDeviceToHost));
    printf("%d waves per workgroup: SIMD of wave 0..%d in the first 6 workgroups:", waves, waves - 1);
    for (int b = 0; b < 6; ++b) { printf("  ["); for (int w = 0; w < waves; ++w) printf("%d", (h[b * waves + w] >> 4) & 3); printf("]"); }
    int hist[16] = {0};
    for (int b = 0; b < blocks; ++b) { int m = 0; for (int w = 0; w < 4; ++w) m |= 1 << ((h[b * waves + w] >> 4) & 3); hist[m]++; }
    printf("   workgroups whose waves 0-3 cover all four SIMDs: %d of %d\n", hist[15], blocks);
    CK(hipFree(out)); free(h);
  }
  return 0;
}
