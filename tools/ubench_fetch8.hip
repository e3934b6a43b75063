// What does FETCH_SIZE count for k_fir_ring's loads?  ("calibrate FETCH_SIZE for the 8 B/lane pattern on
// a known byte count".)  The guide records that on gfx950 the counter reports HALF the bytes of a wide (16 B/lane) streaming
// read; k_fir_ring reads 8 bytes per lane -- one 512-byte row piece per wave and buffer load, rows 64 KiB apart.  Here every
// byte of a [rows, channels] float64 block is read EXACTLY ONCE in that shape (wave (x, y) reads rows 48 y .. 48 y + 47 of
// channels 64 x .. 64 x + 63, no window overlap), and once more with 16-byte-per-lane global loads over the same buffer, so
// that rocprofv3 --pmc FETCH_SIZE on this binary gives the counter's factor for both shapes on a known byte count.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch8.hip -o tools/variants/ubench_fetch8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// the FIR kernel's load: raw buffer load, 8 bytes per lane, wave-uniform row offset
__global__ __launch_bounds__(64) void k_rows8(const double *x, double *sink, long rows, long channels) {
  const long c = (long)blockIdx.x * 64 + threadIdx.x;
  const long r0 = (long)blockIdx.y * 48;
  double acc = 0.0;
  const long pitch = channels * 8;
#pragma unroll 4
  for (int r = 0; r < 48; ++r) {
    if (r0 + r >= rows) break;
    const char *base = (const char *)x + (r0 + r) * pitch;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)pitch, 0x00020000);
    acc += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (unsigned)c * 8u, 0, 0));
  }
  if (acc == 1.2345e300) sink[c] = acc;   // (never true for the data below: the loads stay, nothing is written)
}

typedef double d2 __attribute__((ext_vector_type(2)));
// a plain streaming read: 16 bytes per lane, one piece per thread
__global__ __launch_bounds__(256) void k_flat16(const d2 *x, double *sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const d2 v = x[i]; if (v.x + v.y == 1.2345e300) sink[0] = v.x; }
}
// ... and 8 bytes per lane
__global__ __launch_bounds__(256) void k_flat8(const double *x, double *sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const double v = x[i]; if (v == 1.2345e300) sink[0] = v; }
}

int main(int argc, char **argv) {
  const long channels = argc > 1 ? atol(argv[1]) : 8192, rows = argc > 2 ? atol(argv[2]) : (1L << 18);
  const size_t n = (size_t)rows * channels;
  double *x, *sink;
  CK(hipMalloc(&x, n * 8));
  CK(hipMalloc(&sink, channels * 8));
  CK(hipMemset(x, 0, n * 8));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_rows8, dim3((unsigned)(channels / 64), (unsigned)((rows + 47) / 48)), dim3(64), 0, 0, x, sink, rows, channels);
    hipLaunchKernelGGL(k_flat16, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, 0, (const d2 *)x, sink, n / 2);
    hipLaunchKernelGGL(k_flat8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, sink, n);
  }
  CK(hipDeviceSynchronize());
  printf("bytes read by every launch: %zu\n", n * 8);
  return 0;
}
