// ubench_stride.hip -- what does HBM give the LPC kernel's access pattern?  65 536 frames of 3840 bytes,
// a wave owns 64 consecutive frames and walks them in chunks of PIECE bytes per frame (the kernel: 128),
// fetched with global_load_lds_dwordx4 into a DEPTH-slot LDS ring; nothing is computed.  Reports the read
// rate for PIECE = 128 / 256 / 512 / 1280 and ring depths 2 and 3, next to a plain contiguous sweep.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_stride.hip -o tools/ubench_stride
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// PIECE bytes per frame per chunk; a chunk = 64 frames x PIECE bytes = PIECE/16 transfers of 1 KiB
template <int PIECE, int DEPTH, bool CONTIG>
__global__ __launch_bounds__(64) void k_sweep(const char *sig, int frame_bytes, double *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = PIECE / 16;                      // transfers per chunk
  constexpr int PPF = PIECE / 16;                     // 16-byte pieces per frame per chunk
  constexpr int FPT = 64 / PPF > 0 ? 64 / PPF : 1;    // frames covered by one transfer
  const int lane = threadIdx.x;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int64_t f0 = (int64_t)blockIdx.x * 64;
  const int nchunks = frame_bytes / PIECE;
  auto queue = [&](int c) {
    const unsigned slot = lds0 + (unsigned)(c % DEPTH) * (64u * PIECE);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const char *src;
      if (CONTIG) src = sig + f0 * frame_bytes + ((int64_t)c * NT + j) * 1024 + lane * 16;   // the wave's region, in order
      else {
        const int fr = PPF <= 64 ? j * FPT + lane / PPF : j / (PPF / 64);
        const int pc = PPF <= 64 ? lane % PPF : (j % (PPF / 64)) * 64 + lane;
        src = sig + (f0 + fr) * frame_bytes + (int64_t)c * PIECE + pc * 16;
      }
      dma16(src, slot + j * 1024);
    }
  };
  for (int c = 0; c < DEPTH - 1 && c < nchunks; ++c) queue(c);
  double acc = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (c + DEPTH - 1 < nchunks) queue(c + DEPTH - 1);
    // wait for chunk c: the transfers of the chunks queued after it may stay in flight
    const int after = nchunks - 1 - c < DEPTH - 1 ? nchunks - 1 - c : DEPTH - 1;
    if (after * NT >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (after * NT >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (after * NT >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (after * NT >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += *reinterpret_cast<const double *>(smem + (c % DEPTH) * (64 * PIECE) + lane * 8);
  }
  if (acc == 12345.678) sink[0] = acc;
}

template <int PIECE, int DEPTH, bool CONTIG>
void run(const char *sig, double *sink, int frames, int frame_bytes) {
  const size_t lds = (size_t)DEPTH * 64 * PIECE;
  CK(hipFuncSetAttribute((const void *)k_sweep<PIECE, DEPTH, CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_sweep<PIECE, DEPTH, CONTIG><<<frames / 64, 64, lds>>>(sig, frame_bytes, sink);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) k_sweep<PIECE, DEPTH, CONTIG><<<frames / 64, 64, lds>>>(sig, frame_bytes, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("%s piece %4d B, ring depth %d (%3zu KiB LDS): %.1f us  %.2f TB/s\n", CONTIG ? "contiguous per wave," : "strided frames,     ",
         PIECE, DEPTH, lds / 1024, ms * 1e3, (double)frames * frame_bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const int frames = 65536, frame_bytes = 3840;
  char *sig; double *sink;
  CK(hipMalloc(&sig, (size_t)frames * frame_bytes + 4096));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(sig, 1, (size_t)frames * frame_bytes + 4096));
  run<128, 3, false>(sig, sink, frames, frame_bytes);
  run<128, 4, false>(sig, sink, frames, frame_bytes);
  run<256, 2, false>(sig, sink, frames, frame_bytes);
  run<256, 3, false>(sig, sink, frames, frame_bytes);
  run<768, 2, false>(sig, sink, frames, frame_bytes);
  run<128, 3, true>(sig, sink, frames, frame_bytes);
  run<256, 3, true>(sig, sink, frames, frame_bytes);
  return 0;
}
