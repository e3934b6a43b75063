// What a plain read+write stream reaches on this box, to put k_duo's data path (5.0 - 5.1 TB/s of reads + writes) in
// scale: 4 GiB in, 4 GiB out (configs[1]'s block), 16 bytes per lane and access, U accesses in flight per lane,
// persistent grids of B blocks per CU, default / non-temporal policy; and the ROW-TILED access pattern k_duo has
// (every workgroup moves 128-byte pieces of 32 KiB rows, 8 rows per access) with the same knobs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const d2 *__restrict__ x, d2 *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * 256); else y[i + u * 256] = v[u]; }
  }
}

// k_duo's pattern: block g of G = C / 16 moves channels 16 g .. 16 g + 15 (128 bytes) of every row; a wave's access is
// 8 rows x 128 bytes; TILES tiles of 64 rows in flight per wave (one wave per block, as the kernel's AUX wave)
template <int TILES, bool NT>
__global__ __launch_bounds__(64) void k_rows(const d2 *__restrict__ x, d2 *__restrict__ y, size_t rows, size_t row_d2) {
  const int lane = threadIdx.x, r = lane >> 3, c = lane & 7;
  const size_t col = (size_t)blockIdx.x * 8 + c;
  for (size_t t = 0; t < rows; t += 64 * TILES) {
    d2 v[TILES][8];
#pragma unroll
    for (int k = 0; k < TILES; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t row = t + k * 64 + j * 8 + r;
        const d2 *p = x + (row < rows ? row : rows - 1) * row_d2 + col;
        v[k][j] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
    for (int k = 0; k < TILES; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t row = t + k * 64 + j * 8 + r;
        d2 *p = y + row * row_d2 + col;
        if (row < rows) { if (NT) __builtin_nontemporal_store(v[k][j], p); else *p = v[k][j]; }
      }
  }
}

int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;
  d2 *x, *y;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes));
  CK(hipMemset(x, 1, bytes)); CK(hipMemset(y, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char *what, float ms) { printf("%-64s %7.3f ms  %6.2f TB/s (read + write)\n", what, ms, 2.0 * bytes / (ms * 1e-3) / 1e12); };
#define RUN(label, ...) { __VA_ARGS__; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int it = 0; it < 5; ++it) { __VA_ARGS__; } \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(label, ms / 5); }
  RUN("hipMemcpyDtoD", CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)))
  char label[128];
#define COPY(U, NT, B) { snprintf(label, sizeof label, "flat copy, %d x 16 B in flight per lane, %d blocks per CU%s", U, B, NT ? ", nt" : ""); \
    auto fn = k_copy<U, NT>; RUN(label, fn<<<dim3(256 * B), dim3(256)>>>(x, y, n)) }
  COPY(1, false, 8) COPY(4, false, 4) COPY(4, false, 8) COPY(8, false, 2) COPY(8, false, 4) COPY(8, false, 8)
  COPY(4, true, 8) COPY(8, true, 4) COPY(8, true, 8)
  const size_t row_d2 = 4096 * 8 / 16, rows = bytes / (4096 * 8);
#define ROWS(T, NT) { snprintf(label, sizeof label, "k_duo's pattern (256 one-wave blocks, 128 B x 8 rows), %d tiles in flight%s", T, NT ? ", nt" : ""); \
    auto fn = k_rows<T, NT>; RUN(label, fn<<<dim3(256), dim3(64)>>>(x, y, rows, row_d2)) }
  ROWS(1, false) ROWS(2, false) ROWS(3, false) ROWS(4, false) ROWS(3, true) ROWS(4, true)
  return 0;
}
