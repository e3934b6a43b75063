// Where is the box's HBM ceiling?  (the guide records 6.29 TB/s for a float4 copy; round 3's best flat copy
// here moved 5.33 TB/s of reads + writes at 4 GiB + 4 GiB, and the paced one-pole bank has since done 5.7.)
// Sweep: buffer size (64 MiB ... 4 GiB per side: the Infinity Cache holds 256 MiB), grid shape (one 16-byte piece per thread
// / U pieces per thread, one-shot / persistent grid-stride), cache policy (default / nt), and the two directions alone
// (read-only with a sum that is never stored, write-only fill).  TB/s counts bytes read + bytes written.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_copy2.hip -o tools/ubench_copy2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// one-shot: thread t of block b moves pieces (b * U + u) * 256 + t
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_shot(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n) {
  const size_t i0 = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (i0 + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(x + i0 + u * 256) : x[i0 + u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) if (i0 + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], y + i0 + u * 256); else y[i0 + u * 256] = v[u]; }
}
// persistent grid-stride
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_pers(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * 256); else y[i + u * 256] = v[u]; }
  }
}
// persistent, every block owns ONE contiguous span (block b: [b * span, (b + 1) * span)) instead of interleaving with the others
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_span(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n) {
  const size_t span = n / gridDim.x, lo = (size_t)blockIdx.x * span;
  for (size_t i = lo + threadIdx.x; i < lo + span; i += 256 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * 256); else y[i + u * 256] = v[u]; }
  }
}
// persistent grid-stride with the blocks held in step: a block starts iteration k only when every block has finished
// iteration k - D (one global counter; all blocks resident: 8 per CU).  Is the one-shot grid's advantage the ORDER in
// which the chip walks the buffers (a compact advancing window), which free-running persistent blocks lose by drifting?
template <int U, bool NT, int D>
__global__ __launch_bounds__(256) void k_lock(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n, unsigned *counter) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  unsigned k = 0;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride, ++k) {
    if (k >= (unsigned)D) {
      if (threadIdx.x == 0) {
        const unsigned need = (k - D + 1) * gridDim.x;
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(2);
      }
      __syncthreads();
    }
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * 256); else y[i + u * 256] = v[u]; }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x == 123.456f) y[threadIdx.x] = acc;     // (never true: the loads stay)
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_fill(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v, y + i + u * 256); else y[i + u * 256] = v; }
  }
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const size_t cap = (size_t)4 << 30;
  f4 *x, *y;
  CK(hipMalloc(&x, cap)); CK(hipMalloc(&y, cap));
  unsigned *counter;
  CK(hipMalloc(&counter, 64));
  CK(hipMemset(x, 1, cap)); CK(hipMemset(y, 0, cap));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t only = argc > 1 ? (size_t)atol(argv[1]) : 0;      // (MiB per side; default: the whole sweep)
  for (size_t mib : {64, 256, 1024, 4096}) {
    if (only && mib != only) continue;
    const size_t bytes = mib << 20, n = bytes / 16;
    const int reps = mib <= 256 ? 40 : 8;
    printf("---- %zu MiB in, %zu MiB out ----\n", mib, mib);
#define RUN(label, moved, ...) { __VA_ARGS__; __VA_ARGS__; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int it = 0; it < reps; ++it) { __VA_ARGS__; } \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps; \
    printf("%-72s %8.4f ms  %6.2f TB/s\n", label, ms, (double)(moved) / (ms * 1e-3) / 1e12); }
    char label[160];
    RUN("hipMemcpyDtoD", 2 * bytes, CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)))
#define SHOT(U, NT) { snprintf(label, sizeof label, "one-shot grid, %d x 16 B per thread%s", U, NT ? ", nt" : ""); \
    auto fn = k_shot<U, NT>; RUN(label, 2 * bytes, fn<<<dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256)>>>(x, y, n)) }
    SHOT(1, false) SHOT(2, false) SHOT(4, false) SHOT(8, false) SHOT(1, true) SHOT(4, true)
#define PERS(K, U, NT, B) { snprintf(label, sizeof label, "%s, %d x 16 B in flight, %d blocks per CU%s", #K, U, B, NT ? ", nt" : ""); \
    auto fn = K<U, NT>; RUN(label, 2 * bytes, fn<<<dim3(256 * B), dim3(256)>>>(x, y, n)) }
    PERS(k_pers, 4, false, 8) PERS(k_pers, 8, false, 8) PERS(k_pers, 8, true, 8) PERS(k_pers, 16, true, 4) PERS(k_pers, 8, true, 16)
    PERS(k_span, 8, false, 8) PERS(k_span, 8, true, 8) PERS(k_span, 8, true, 4)
    PERS(k_pers, 1, false, 8) PERS(k_pers, 1, true, 8) PERS(k_pers, 2, true, 8)
#define LOCK(U, NT, D) { snprintf(label, sizeof label, "k_lock (blocks in step, slack %d iterations), %d x 16 B in flight, 8 blocks per CU%s", D, U, NT ? ", nt" : ""); \
    auto fn = k_lock<U, NT, D>; RUN(label, 2 * bytes, CK(hipMemsetAsync(counter, 0, 4, 0)); fn<<<dim3(256 * 8), dim3(256)>>>(x, y, n, counter)) }
    LOCK(1, true, 1) LOCK(1, true, 4) LOCK(4, true, 1) LOCK(4, true, 4) LOCK(8, true, 2) LOCK(4, false, 2)
#define ONE(K, U, NT, B, what) { snprintf(label, sizeof label, "%s only, %d x 16 B in flight, %d blocks per CU%s", what, U, B, NT ? ", nt" : ""); \
    auto fn = K<U, NT>; RUN(label, bytes, fn<<<dim3(256 * B), dim3(256)>>>(x, y, n)) }
    ONE(k_read, 8, false, 8, "read") ONE(k_read, 8, true, 8, "read") ONE(k_read, 16, false, 4, "read")
    ONE(k_fill, 8, false, 8, "write") ONE(k_fill, 8, true, 8, "write") ONE(k_fill, 4, true, 16, "write")
  }
  return 0;
}
