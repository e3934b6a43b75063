// ubench_ldsq.hip -- what the CU's LDS pipe charges for the operations k_quad issues: one wave per SIMD (1024 blocks of
// 64 threads, 36 KiB of LDS each: four per CU), every wave issuing the same LDS operation back to back (8 independent
// operations in flight).  Reports cycles of the CU's LDS pipe per operation = 4 waves x ops / elapsed CU cycles (the
// pipe is shared by the CU's four SIMDs) and the per-wave issue interval.
//   A read b128, 64 lanes, conflict-free (lane * 16)          B read b128, 16 lanes active (k_quad's section-0 lanes), row stride 128 XOR-swizzled
//   C read b128, 64 lanes, the four lanes of a channel read the same 16 bytes (k_quad as built: 4 x redundant)
//   D write b128, 64 lanes, lane * 16                          E write b128, 16 lanes active, row stride 144
//   F write b64, 64 lanes (48 of them to a private dummy slot) G read b64, 64 lanes, 16 x 8 contiguous bytes, each read by four lanes
//   H read b64, 16 lanes active, contiguous
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsq.hip -o tools/ubench_ldsq
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int s = (lane >> 2) & 3, cl = ((lane >> 4) << 2) | (lane & 3);
  for (int i = lane; i < 36 * 1024 / 8; i += 64) reinterpret_cast<double *>(smem)[i] = i;
  __syncthreads();
  int addr = 0;
  bool act = true;
  if (MODE == 0 || MODE == 3) addr = lane * 16;
  if (MODE == 1) { addr = cl * 128 + ((3 ^ (cl & 7)) << 4); act = s == 0; }
  if (MODE == 2) addr = cl * 128 + ((3 ^ (cl & 7)) << 4);
  if (MODE == 4) { addr = cl * 144; act = s == 3; }
  if (MODE == 5) addr = s == 3 ? cl * 8 : 4096 + lane * 8;
  if (MODE == 6) addr = cl * 8;
  if (MODE == 7) { addr = cl * 8; act = s == 0; }
  dbl2 acc = {0.0, 0.0};
  dbl2 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = dbl2{1.0 * lane, 2.0 * u};
  if (act) {
    for (int i = 0; i < iters; ++i) {
      if (MODE <= 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const dbl2 *>(smem + addr + u * 2048);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      } else if (MODE == 3 || MODE == 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<dbl2 *>(smem + addr + u * 2304) = v[u] + acc;
        asm volatile("" ::: "memory");
      } else if (MODE == 5) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<double *>(smem + addr + u * 128) = v[u].x + acc.x;
        asm volatile("" ::: "memory");
      } else {
        double w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const double *>(smem + addr + u * 128);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc.x += w[u];
      }
    }
  }
  out[blockIdx.x * 64 + lane] = acc.x + acc.y + reinterpret_cast<double *>(smem)[lane];
}

template <int MODE>
void run(const char *what) {
  double *out;
  const int blocks = 1024, iters = 20000;
  CK(hipMalloc(&out, blocks * 64 * sizeof(double)));
  CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 64, 36 * 1024>>>(out, 100);
  CK(hipEventRecord(e0));
  k<MODE><<<blocks, 64, 36 * 1024>>>(out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double ops = (double)iters * 8;
  const double ns = ms * 1e6 / ops;
  printf("%-90s %6.2f ns per operation and wave = %5.1f cycles at 2.4 GHz; LDS pipe per operation (4 waves per CU): %5.1f cycles\n",
         what, ns, ns * 2.4, ns * 2.4 / 4);
  CK(hipFree(out));
}

int main() {
  run<0>("A read  b128, 64 lanes, lane * 16");
  run<1>("B read  b128, 16 lanes active, rows of 128 B swizzled");
  run<2>("C read  b128, 64 lanes, four lanes per address (k_quad input, first DMA version)");
  run<3>("D write b128, 64 lanes, lane * 16");
  run<4>("E write b128, 16 lanes active, rows of 144 B (k_quad output, channel-major)");
  run<5>("F write b64,  64 lanes, 16 to a row + 48 private (k_quad output, time-major)");
  run<6>("G read  b64,  64 lanes, 16 x 8 contiguous bytes read by four lanes each");
  run<7>("H read  b64,  16 lanes active, contiguous");
  return 0;
}
