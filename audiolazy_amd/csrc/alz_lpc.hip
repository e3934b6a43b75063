// alz_lpc.hip -- batched lpc.kautocor: autocorrelation + Levinson-Durbin per frame.
//
// Replaces, for a batch of frames,
//   acorr(blk, order)            reference audiolazy/lazy_analysis.py:277-312
//   levinson_durbin(acdata, ord) reference audiolazy/lazy_lpc.py:52-136
//   lpc.kautocor(blk, order)     reference audiolazy/lazy_lpc.py:229-272
//
// Mapping: a frame owns a SLOT of 32 (order <= 31) or 64 lanes of a wavefront;
// lane i of the slot owns lag i / coefficient i.  The frame is staged once in
// LDS (coalesced HBM read), then every lag lane walks it left to right --
// the same summation order as the reference's sum(...), so acorr is bit-exact
// (file built with -ffp-contract=off).  Levinson-Durbin then runs across the
// lanes of the slot with cross-lane shuffles: the standard O(order^2)
// recursion, mathematically the reference's update
//   A -= inner(A, z**-m) / inner(B, B) * B      (lazy_lpc.py:128-131)
// with inner(B, B) carried as the running prediction error.
#include <stdlib.h>

#include <type_traits>

#include "alz_lev.h"

namespace alz {

template <int SLOT>
__device__ __forceinline__ double slot_sum(double v) {
#pragma unroll
  for (int off = SLOT / 2; off > 0; off >>= 1) v = v + __shfl_xor(v, off, SLOT);
  return v;
}

template <int SLOT>
__global__ __launch_bounds__(64) void k_lpc(const double *__restrict__ sig, int64_t n_frames,
                                            int frame_len, int64_t hop, int order,
                                            double *__restrict__ coefs, double *__restrict__ err,
                                            int *__restrict__ status, double *__restrict__ r_out,
                                            int from_r) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *smem = reinterpret_cast<double *>(smem_raw);
  constexpr int FPW = 64 / SLOT;  // frames per wave
  const int lane = threadIdx.x;
  const int slot = lane / SLOT, i = lane % SLOT;
  const int64_t f0 = (int64_t)blockIdx.x * FPW;

  // stage the wave's frames in LDS, coalesced (from_r: "frames" are ready-made lag lists
  // r[0..order] and the autocorrelation pass is skipped -- levinson_durbin alone)
  for (int s = 0; s < FPW; ++s) {
    const int64_t f = f0 + s;
    if (f >= n_frames) break;
    const double *src = sig + f * hop;
    for (int n = lane; n < frame_len; n += 64) smem[s * frame_len + n] = src[n];
  }
  __syncthreads();

  const int64_t f = f0 + slot;
  const bool live = f < n_frames;
  const double *fr = smem + slot * frame_len;

  // acorr: lane i sums lag i, left to right (lazy_analysis.py:311-312)
  double r = 0.0;
  if (from_r) {
    if (live && i <= order && i < frame_len) r = fr[i];   // short lag lists are zero-extended (lazy_lpc.py:117-118)
  } else if (live && i <= order) {
    const int cnt = frame_len - i;
    for (int n = 0; n < cnt; ++n) r = r + fr[n] * fr[n + i];
  }
  if (r_out) {
    if (live && i <= order) r_out[f * (order + 1) + i] = r;
    if (!coefs) return;
  }

  // Levinson-Durbin across the slot's lanes
  double a = (i == 0) ? 1.0 : 0.0;
  double E = __shfl(r, 0, SLOT);
  int st = ALZ_OK;
  for (int m = 1; m <= order; ++m) {
    const int src = (m - i) & (SLOT - 1);
    const double rr = __shfl(r, src, SLOT);   // r[m - i]
    const double ar = __shfl(a, src, SLOT);   // a[m - i]
    const double num = slot_sum<SLOT>((i < m) ? a * rr : 0.0);
    if (E == 0.0) st = ALZ_E_PARCOR;          // inner(B, B) == 0, lazy_lpc.py:132-133
    const double k = (st == ALZ_OK) ? -(num / E) : 0.0;
    if (i >= 1 && i < m) a = a + k * ar;
    if (i == m) a = k;
    E = E * (1.0 - k * k);
  }
  if (live) {
    if (i <= order) coefs[f * (order + 1) + i] = a;
    if (i == 0) {
      err[f] = E;
      status[f] = st;
    }
  }
}

// ---------------------------------------------------------------------------
// Second generation, two kernels.
//
// k_acorr_dense: lanes are (frame, lag) pairs packed densely -- pair index = block*64 + lane,
// frame = index / P, lag = index % P with P = max_lag + 1 -- so all 64 lanes work (the slot
// kernel above keeps 17 of 32 busy at order 16).  The few frames a wave touches are staged in
// LDS; every lane walks its lag left to right: bit-exact acorr (lazy_analysis.py:311-312).
//
// k_levinson_lane: one lane per frame, the whole O(order^2) recursion in registers (the loops
// are fully unrolled up to kLevMax so the coefficient array is statically indexed); no
// cross-lane traffic at all.  Same update as the slot kernel, floating-point parity (<= 1e-9).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_acorr_dense(const double *__restrict__ sig, int64_t n_frames,
                                                     int frame_len, int64_t hop, int P,
                                                     double *__restrict__ r_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *smem = reinterpret_cast<double *>(smem_raw);
  const int lane = threadIdx.x;
  const int64_t idx0 = (int64_t)blockIdx.x * 64;
  const int64_t total = n_frames * P;
  const int64_t f_lo = idx0 / P;
  int64_t f_hi = (idx0 + 63) / P;
  if (f_hi > n_frames - 1) f_hi = n_frames - 1;
  const int nfr = (int)(f_hi - f_lo + 1);
  // frames sit at a stride = 16 (mod 32) doubles: the two frames a 32-lane half touches then
  // occupy opposite halves of the 64 LDS banks
  const int fstride = ((frame_len + 15) / 32) * 32 + 16;
  for (int s = 0; s < nfr; ++s) {
    const double *src = sig + (f_lo + s) * hop;
    for (int n = lane; n < frame_len; n += 64) smem[s * fstride + n] = src[n];
  }
  __syncthreads();
  const int64_t idx = idx0 + lane;
  if (idx >= total) return;
  const int64_t f = idx / P;
  const int i = (int)(idx - f * P);
  const double *fr = smem + (int)(f - f_lo) * fstride;
  const int cnt = frame_len - i;
  double acc = 0.0;
  int n = 0;
  for (; n + 4 <= cnt; n += 4) {
    const double x0 = fr[n], x1 = fr[n + 1], x2 = fr[n + 2], x3 = fr[n + 3];
    const double y0 = fr[n + i], y1 = fr[n + i + 1], y2 = fr[n + i + 2], y3 = fr[n + i + 3];
    acc = acc + x0 * y0;
    acc = acc + x1 * y1;
    acc = acc + x2 * y2;
    acc = acc + x3 * y3;
  }
  for (; n < cnt; ++n) acc = acc + fr[n] * fr[n + i];
  r_out[idx] = acc;
}

// k_acorr_global: the same (frame, lag) pairs straight from global memory -- the catch-all for
// frames too long to stage (acorr's default lag list is the whole block, lazy_analysis.py:309-310).
__global__ __launch_bounds__(64) void k_acorr_global(const double *__restrict__ sig, int64_t n_frames,
                                                      int frame_len, int64_t hop, int P,
                                                      double *__restrict__ r_out) {
  const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (idx >= n_frames * P) return;
  const int64_t f = idx / P;
  const int i = (int)(idx - f * P);
  const double *fr = sig + f * hop;
  double acc = 0.0;
  for (int n = 0; n < frame_len - i; ++n) acc = acc + fr[n] * fr[n + i];
  r_out[idx] = acc;
}

// k_acorr_lane<P>: one lane per frame, P = max_lag + 1 accumulators per lane.  Per sample the
// lane needs ONE new value (its window fr[n .. n+P-1] slides in registers, rotated by unrolling
// P steps) for P multiply-adds: no LDS, no cross-lane traffic, f64-issue-bound.  Every lag still
// sums left to right, so the result is bit-identical to the reference (lazy_analysis.py:311-312).
template <int P>
__global__ __launch_bounds__(64) void k_acorr_lane(const double *__restrict__ sig, int64_t n_frames,
                                                    int frame_len, int64_t hop, double *__restrict__ r_out) {
  int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = f < n_frames;
  if (!live) f = n_frames - 1;
  const double *fr = sig + f * hop;
  const int L = frame_len;
  double acc[P], w[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    acc[i] = 0.0;
    w[i] = (i < L) ? fr[i] : 0.0;
  }
  int n = 0;
  // full steps: every lag of steps n .. n+P-1 is inside the frame and so is the refill fr[n+u+P]
  for (; n + 2 * P - 1 < L; n += P) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const double x = w[u];                               // fr[n+u]; w[(u+i) % P] = fr[n+u+i]
#pragma unroll
      for (int i = 0; i < P; ++i) acc[i] = acc[i] + x * w[(u + i) % P];
      w[u] = fr[n + u + P];
    }
  }
  // tail: lags run off the end of the frame one by one
  for (; n < L; ++n) {
    const double x = fr[n];
#pragma unroll
    for (int i = 0; i < P; ++i)
      if (n + i < L) acc[i] = acc[i] + x * fr[n + i];
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < P; ++i) r_out[f * P + i] = acc[i];
  }
}

// k_acorr_stage<P>: the same lane-per-frame sums, with the samples staged through LDS so that
// HBM/L2 see full 128-byte rows: a wave owns 64 frames and walks them in chunks of 16 samples;
// each chunk (64 frames x 128 B) arrives as eight 1 KiB global_load_lds DMA transfers into a
// 3-slot ring, XOR-swizzled on the global side so the lanes (one per frame) read their 16-byte
// pieces from distinct banks.  Products are formed "by later index" -- a new sample x[m] meets
// x[m-i] for every lag i -- which adds each lag's terms in the same ascending order as the
// reference (bit-exact) and needs only the previous chunk(s) as history, all statically indexed.
// Needs frame starts on 16-byte boundaries (even hop) and P <= 33.
__device__ __forceinline__ void lpc_dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// the same with the non-temporal policy (NT instantiations: batches far larger than the Infinity Cache, read exactly once)
__device__ __forceinline__ void lpc_dma16_nt(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// LEV = 1: Levinson-Durbin runs right away on the lane's P lags (still in registers) and the
// kernel writes coefficients / error / status instead of the lags: lpc.kautocor in one launch.
// LEV = 2 (ALZ_LPC_DENSE): the same with the reference's own dense Levinson-Durbin (alz_lev.h): coefficients
// and error bit-identical to lpc.kautocor on every frame, still one launch.
// FMA = true (opt-in, ALZ_LPC_FUSED): every term is one v_fma_f64 instead of a separately rounded
// multiply and add -- half the instructions of a kernel that is bound by FP64 issue (moving the same
// 252 MB without computing takes 34 us, tools/ubench_stride.hip), same ascending order per lag, NOT the
// reference's doubles (differences ~1e-16 relative per lag; the contract is 1e-6).
// ALZ_LPC_FULL 1: chunks that lie wholly inside the frame skip the per-step length check, so their 16 steps
// are one basic block (61.2 against 65.5 us per call, profiles/r02_lpc_split.log).  Measured and not kept:
// all 17 products of a step before the 17 additions (the compiler pairs each multiply with its add through
// one temporary) -- 67.1 us: back-to-back dependent FP64 pairs cost nothing here.
#ifndef ALZ_LPC_FULL
#define ALZ_LPC_FULL 1
#endif
// RING: chunk slots in LDS (RING - 1 chunks of DMA ahead of the one being summed).  Two: 16 KiB instead of 24 put ten
// one-wave workgroups on a CU instead of six (+7 .. 23 % at 2^20 frames, +3 % at configs[4]'s 65 536; deeper rings are
// slower still: NOTES_r03.md 10).
// NT (round 5): the tile DMA with the non-temporal policy, for batches of >= 512 MiB -- a read-only stream moves 6.9 TB/s
// that way against 6.1 (profiles/r05_copy_ceiling.log): 2^20 frames +2.3 % (+1.4 % with the dense Levinson-Durbin).  A
// COMPILE-TIME choice: as a wave-uniform branch around the two asm statements it cost configs[4]'s own 65 536-frame launch
// 2 % (profiles/r05_lpc_nt_ab.log; k_duo's lesson).  The three-slot instantiations it replaces in the shipped library are
// reachable in -DALZ_TUNING builds only (two slots have been the measured choice since round 3).
template <int P, int LEV, bool FMA = false, int kLpcRing = 3, bool NT = false>
__global__ __launch_bounds__(64) void k_acorr_stage(const double *__restrict__ sig, int64_t n_frames,
                                                     int frame_len, int64_t hop, double *__restrict__ r_out,
                                                     double *__restrict__ coefs, double *__restrict__ err,
                                                     int *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = (P - 1 + 15) / 16;       // chunks of history needed (1 for P <= 17, 2 for P <= 33)
  const int lane = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * 64;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int L = frame_len;
  const int nchunks = (L + 15) / 16;
  // DMA source of this lane for transfer j of chunk c: frame f0 + 8j + lane/8, piece (lane%8)^(frame%8)
  // (frames past the end are clamped to the last one: loaded, never stored)
  auto src_of = [&](int j, int c) -> const double * {
    int64_t fr = f0 + 8 * j + lane / 8;
    if (fr > n_frames - 1) fr = n_frames - 1;
    const int piece = (lane % 8) ^ (int)((8 * j + lane / 8) & 7);
    int64_t s0 = (int64_t)16 * c + 2 * piece;
    if (s0 > L - 2) s0 = (L >= 2) ? ((L - 2) & ~1) : 0;   // keep the (aligned) 16-byte read inside the frame
    return sig + fr * hop + s0;
  };
  auto queue = [&](int c) {
    const unsigned slot = lds0 + (unsigned)(c % kLpcRing) * 8192u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (NT) lpc_dma16_nt(src_of(j, c), slot + j * 1024);
      else lpc_dma16(src_of(j, c), slot + j * 1024);
    }
  };
  const int64_t f = f0 + lane;
  const bool live = f < n_frames;
  double acc[P];
#pragma unroll
  for (int i = 0; i < P; ++i) acc[i] = 0.0;
  // The chunk being summed and the H chunks before it live in H + 1 register sets that ROTATE by name (the chunk loop is unrolled
  // H + 1 times): round 5 copied the finished chunk into a reversed history array every chunk -- 16 v_mov_b64 per chunk, 3 % of the
  // VALU instructions of a kernel that is bound by VALU issue (round 6).
  double buf[H + 1][16];
#pragma unroll
  for (int h = 0; h <= H; ++h)
#pragma unroll
    for (int k = 0; k < 16; ++k) buf[h][k] = 0.0;

  for (int c0 = 0; c0 < kLpcRing - 1 && c0 < nchunks; ++c0) queue(c0);
  // chunk c: samples into `cur`; `p1` / `p2` hold chunks c - 1 / c - 2 (x[16 c + d], d < 0, is p1[16 + d] or p2[32 + d])
  auto do_chunk = [&](int c, double (&cur)[16], const double (&p1)[16], const double (&p2)[16]) {
    if (c + kLpcRing - 1 < nchunks) queue(c + kLpcRing - 1);
    // transfers issued after chunk c's: chunks c+1 .. c+kLpcRing-1 (those that exist)
    const int after = (nchunks - 1 - c < kLpcRing - 1) ? (nchunks - 1 - c) : kLpcRing - 1;
    static_assert(kLpcRing >= 2 && (kLpcRing - 1) * 8 <= 56, "vmcnt is a 6-bit count");
    switch (after) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    }
    const char *slot = smem + (c % kLpcRing) * 8192 + lane * 128;
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      const d2 v = *reinterpret_cast<const d2 *>(slot + ((pc ^ (lane & 7)) * 16));
      cur[2 * pc] = v.x;
      cur[2 * pc + 1] = v.y;
    }
    const int valid = L - 16 * c;             // samples of this chunk inside the frame (>= 1)
    // x[m - i] is inside this chunk (u >= i) or in one of the chunks before it; only the first H chunks have lags that reach
    // before the start of the frame (checked form)
    auto chunk = [&](auto checked, auto full) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (decltype(full)::value || u < valid) {
          const double xm = cur[u];
#pragma unroll
          for (int i = 0; i < P; ++i) {
            const int d = u - i;              // (folded: u and i are unrolled loop indices)
            const double xe = d >= 0 ? cur[d >= 0 ? d : 0] : d >= -16 ? p1[d >= -16 && d < 0 ? 16 + d : 0] : p2[d < -16 && d >= -32 ? 32 + d : 0];
            if constexpr (decltype(checked)::value) {
              if (u - i >= 0 || 16 * c + u - i >= 0) acc[i] = FMA ? __builtin_fma(xe, xm, acc[i]) : acc[i] + xe * xm;
            } else {
              acc[i] = FMA ? __builtin_fma(xe, xm, acc[i]) : acc[i] + xe * xm;
            }
          }
        }
      }
    };
#if ALZ_LPC_FULL
    // a whole chunk inside the frame (every chunk but the last one of a ragged frame length): no per-step
    // length check, so the 16 steps are one basic block
    if (c < H) chunk(std::true_type{}, std::false_type{});
    else if (valid >= 16) chunk(std::false_type{}, std::true_type{});
    else chunk(std::false_type{}, std::false_type{});
#else
    if (c < H) chunk(std::true_type{}, std::false_type{});
    else chunk(std::false_type{}, std::false_type{});
#endif
  };
  static_assert(H == 1 || H == 2, "one or two chunks of history");
  for (int c = 0; c < nchunks; c += H + 1) {
    if constexpr (H == 1) {
      do_chunk(c, buf[0], buf[1], buf[1]);
      if (c + 1 < nchunks) do_chunk(c + 1, buf[1], buf[0], buf[0]);
    } else {
      do_chunk(c, buf[0], buf[2], buf[1]);
      if (c + 1 < nchunks) do_chunk(c + 1, buf[1], buf[0], buf[2]);
      if (c + 2 < nchunks) do_chunk(c + 2, buf[2], buf[1], buf[0]);
    }
  }
  if constexpr (LEV == 0) {
    if (live) {
#pragma unroll
      for (int i = 0; i < P; ++i) r_out[f * P + i] = acc[i];
    }
  } else if constexpr (LEV == 2) {
    double A[P], e;
    int st;
    levinson_dense_regs<P>(acc, A, e, st);
    if (live) {
#pragma unroll
      for (int i = 0; i < P; ++i) coefs[f * P + i] = A[i];
      err[f] = e;
      status[f] = st;
    }
  } else {
    constexpr int order = P - 1;
    double a[P];
#pragma unroll
    for (int i = 0; i < P; ++i) a[i] = 0.0;
    a[0] = 1.0;
    double E = acc[0];
    int st = ALZ_OK;
#pragma unroll
    for (int m = 1; m <= order; ++m) {
      double num = acc[m];
#pragma unroll
      for (int i = 1; i < m; ++i) num = num + a[i] * acc[m - i];
      if (E == 0.0) st = ALZ_E_PARCOR;               // inner(B, B) == 0, lazy_lpc.py:132-133
      const double k = (st == ALZ_OK) ? -(num / E) : 0.0;
#pragma unroll
      for (int i = 1; 2 * i <= m; ++i) {
        const double ai = a[i], aj = a[m - i];
        a[i] = ai + k * aj;
        if (2 * i != m) a[m - i] = aj + k * ai;
      }
      a[m] = k;
      E = E * (1.0 - k * k);
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < P; ++i) coefs[f * P + i] = a[i];
      err[f] = E;
      status[f] = st;
    }
  }
}

// lag_matrix (lazy_analysis.py:315-342), the covariance-method statistics of lpc.covar / lpc.kcovar
// (lazy_lpc.py:285, :310): cell (j, i) of a frame is sum(blk[n - i] * blk[n - j] for n = max_lag .. L - 1), added
// left to right from 0 like Python's sum.  One lane per cell; consecutive lanes take consecutive i, so a step's
// blk[n - i] reads of a row are one contiguous run and blk[n - j] is a broadcast.  A caller-side helper (P^2
// independent sums of L - P + 1 terms), not one of the streaming kernels.
__global__ __launch_bounds__(256) void k_lag_matrix(const double *__restrict__ sig, int64_t n_frames, int frame_len,
                                                     int64_t hop, int P, double *__restrict__ out) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t cells = (int64_t)P * P;
  if (cell >= n_frames * cells) return;
  const int64_t f = cell / cells;
  const int r = (int)(cell % cells);
  const int j = r / P, i = r % P;
  const double *blk = sig + f * hop;
  double acc = 0.0;
  for (int n = P - 1; n < frame_len; ++n) acc = acc + blk[n - i] * blk[n - j];
  out[cell] = acc;
}

// The same cells, a lane per (frame, row j): the P sums of a row live in registers and every step multiplies the
// row's own sample blk[n - j] into a window of the last P samples that is shared by all rows of the frame -- 2 P
// FP64 operations per two (cached) loads instead of two loads per multiply-add.  The window rotates through its
// registers (logical blk[n - k] sits in slot (k - phase) mod P), so a step moves nothing.  Each sum still adds its
// terms in ascending n from 0.0: the reference's doubles.
template <int P>
__global__ __launch_bounds__(256) void k_lag_rows(const double *__restrict__ sig, int64_t n_frames, int frame_len,
                                                   int64_t hop, double *__restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_frames * P) return;
  const int64_t f = row / P;
  const int j = (int)(row - f * P);
  const double *blk = sig + f * hop;
  double acc[P], win[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    acc[i] = 0.0;
    win[i] = blk[P - 1 - i];                               // phase 0 at n = P - 1: slot k holds blk[n - k]
  }
  for (int n0 = P - 1; n0 < frame_len; n0 += P) {
#pragma unroll
    for (int ph = 0; ph < P; ++ph) {
      const int n = n0 + ph;
      if (n < frame_len) {                                 // (uniform: every lane of the launch has the same length)
        const double bj = blk[n - j];
#pragma unroll
        for (int i = 0; i < P; ++i) acc[i] = acc[i] + win[(i + P - ph) % P] * bj;
        if (n + 1 < frame_len) win[(2 * P - 1 - ph) % P] = blk[n + 1];   // the oldest sample's slot takes the next one
      }
    }
  }
#pragma unroll
  for (int i = 0; i < P; ++i) out[row * P + i] = acc[i];
}

typedef void (*lag_rows_fn)(const double *, int64_t, int, int64_t, double *);
static lag_rows_fn pick_lag_rows(int P) {
  switch (P) {
    case 3: return k_lag_rows<3>;
    case 5: return k_lag_rows<5>;
    case 9: return k_lag_rows<9>;
    case 11: return k_lag_rows<11>;
    case 13: return k_lag_rows<13>;
    case 17: return k_lag_rows<17>;
    case 21: return k_lag_rows<21>;
    case 25: return k_lag_rows<25>;
    case 33: return k_lag_rows<33>;
    default: return nullptr;
  }
}

typedef void (*acorr_lane_fn)(const double *, int64_t, int, int64_t, double *);
typedef void (*acorr_stage_fn)(const double *, int64_t, int, int64_t, double *, double *, double *, int *);
template <int LEV, bool FMA = false, int RING = 3, bool NT = false>
static acorr_stage_fn pick_acorr_stage(int P) {
  switch (P) {
    case 9: return k_acorr_stage<9, LEV, FMA, RING, NT>;
    case 11: return k_acorr_stage<11, LEV, FMA, RING, NT>;
    case 13: return k_acorr_stage<13, LEV, FMA, RING, NT>;
    case 17: return k_acorr_stage<17, LEV, FMA, RING, NT>;
    case 21: return k_acorr_stage<21, LEV, FMA, RING, NT>;
    case 25: return k_acorr_stage<25, LEV, FMA, RING, NT>;
    case 33: return k_acorr_stage<33, LEV, FMA, RING, NT>;
    default: return nullptr;
  }
}

// the bit-identical one-launch form for the orders the dense Levinson-Durbin is unrolled for (alz_lev.hip has
// the same list); other orders run the two-launch form
template <int RING = 3, bool NT = false>
static acorr_stage_fn pick_acorr_stage_dense(int P) {
  switch (P) {
    case 9: return k_acorr_stage<9, 2, false, RING, NT>;
    case 11: return k_acorr_stage<11, 2, false, RING, NT>;
    case 13: return k_acorr_stage<13, 2, false, RING, NT>;
    case 17: return k_acorr_stage<17, 2, false, RING, NT>;
    default: return nullptr;
  }
}
// three chunk slots: tuning builds only (see k_acorr_stage)
#ifdef ALZ_TUNING
#define ALZ_LPC_RING3(expr) (expr)
#else
#define ALZ_LPC_RING3(expr) ((alz::acorr_stage_fn) nullptr)
#endif
// a batch of >= 512 MiB of signal is read exactly once and does not fit the Infinity Cache: non-temporal tiles
static bool stage_nt(int64_t n_frames, int frame_len) { return n_frames * (int64_t)frame_len >= (INT64_C(1) << 26); }

// two chunk slots (one chunk of DMA ahead): same-box pairs 1.04 / 1.06 against 1.01 / 1.02 Gframes/s at 65 536 frames,
// 1.01 against 0.94 at 2^20 (profiles/r03_lpc_ring.log); the three-slot instantiations stay selectable in tuning builds
static bool stage_ring2(int64_t) {
  static const int force = ALZ_TUNE("ALZ_LPC_RING2", 1);
  return force != 0;
}

static bool stage_ok(const double *sig, int64_t n_frames, int frame_len, int64_t hop) {
  return n_frames >= 16384 && frame_len >= 32 && (hop % 2) == 0 && ((uintptr_t)sig & 15) == 0;
}

static acorr_lane_fn pick_acorr_lane(int P) {
  switch (P) {
    case 9: return k_acorr_lane<9>;
    case 11: return k_acorr_lane<11>;
    case 13: return k_acorr_lane<13>;
    case 17: return k_acorr_lane<17>;
    case 21: return k_acorr_lane<21>;
    case 25: return k_acorr_lane<25>;
    case 33: return k_acorr_lane<33>;
    default: return nullptr;
  }
}

static constexpr int kLevMax = 32;

__global__ __launch_bounds__(64) void k_levinson_lane(const double *__restrict__ r_in, int64_t n_frames,
                                                       int n_lags, int order, double *__restrict__ coefs,
                                                       double *__restrict__ err, int *__restrict__ status) {
  const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (f >= n_frames) return;
  double r[kLevMax + 1], a[kLevMax + 1];
#pragma unroll
  for (int i = 0; i <= kLevMax; ++i) {
    r[i] = (i <= order && i < n_lags) ? r_in[f * n_lags + i] : 0.0;   // zero-extended lags (lazy_lpc.py:117-118)
    a[i] = 0.0;
  }
  a[0] = 1.0;
  double E = r[0];
  int st = ALZ_OK;
#pragma unroll
  for (int m = 1; m <= kLevMax; ++m) {
    if (m <= order) {   // (no break: the loop must unroll fully to keep a[] and r[] in registers)
    double num = r[m];
#pragma unroll
    for (int i = 1; i < m; ++i) num = num + a[i] * r[m - i];
    if (E == 0.0) st = ALZ_E_PARCOR;                 // inner(B, B) == 0, lazy_lpc.py:132-133
    const double k = (st == ALZ_OK) ? -(num / E) : 0.0;
#pragma unroll
    for (int i = 1; 2 * i <= m; ++i) {
      const double ai = a[i], aj = a[m - i];
      a[i] = ai + k * aj;
      if (2 * i != m) a[m - i] = aj + k * ai;
    }
    a[m] = k;
    E = E * (1.0 - k * k);
    }
  }
#pragma unroll
  for (int i = 0; i <= kLevMax; ++i)
    if (i <= order) coefs[f * (order + 1) + i] = a[i];
  err[f] = E;
  status[f] = st;
}

// dense acorr launch; returns false (and launches nothing) when the staged frames do not fit LDS
static bool launch_acorr_dense(const double *sig, int64_t n_frames, int frame_len, int64_t hop,
                               int max_lag, double *r_out, hipStream_t st, int *rc) {
  *rc = ALZ_OK;
  const int P = max_lag + 1;
  // lane-per-frame form for the usual orders when there are enough frames to fill the chip
  const bool ring2 = stage_ring2(n_frames);
  const bool nt = stage_nt(n_frames, frame_len);
  if (acorr_stage_fn st_fn = !stage_ok(sig, n_frames, frame_len, hop) ? nullptr
                             : ring2 ? (nt ? pick_acorr_stage<0, false, 2, true>(P) : pick_acorr_stage<0, false, 2>(P))
                                     : ALZ_LPC_RING3(pick_acorr_stage<0>(P))) {
    hipLaunchKernelGGL(st_fn, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), (ring2 ? 2 : 3) * 8192, st, sig, n_frames,
                       frame_len, hop, r_out, (double *)nullptr, (double *)nullptr, (int *)nullptr);
    if (hipGetLastError() != hipSuccess) *rc = fail(ALZ_E_HIP, "k_acorr_stage launch failed");
    note_kernel("k_acorr_stage<" + std::to_string(P) + ",lags," + (ring2 ? "2" : "3") + " slots>", true);
    return true;
  }
  if (acorr_lane_fn lane_fn = (n_frames >= 16384 && frame_len >= 2 * P) ? pick_acorr_lane(P) : nullptr) {
    hipLaunchKernelGGL(lane_fn, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0, st, sig, n_frames,
                       frame_len, hop, r_out);
    if (hipGetLastError() != hipSuccess) *rc = fail(ALZ_E_HIP, "k_acorr_lane launch failed");
    note_kernel("k_acorr_lane<" + std::to_string(P) + ">", true);
    return true;
  }
  const int nfr_max = 64 / P + 2;     // frames one wave's 64 (frame, lag) pairs can touch
  const size_t lds = (size_t)nfr_max * (((frame_len + 15) / 32) * 32 + 16) * sizeof(double);
  const int64_t total = n_frames * P;
  if (lds > 64 * 1024) {
    hipLaunchKernelGGL(k_acorr_global, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, sig, n_frames,
                       frame_len, hop, P, r_out);
    if (hipGetLastError() != hipSuccess) *rc = fail(ALZ_E_HIP, "k_acorr_global launch failed");
    note_kernel("k_acorr_global", true);
    return true;
  }
  hipLaunchKernelGGL(k_acorr_dense, dim3((unsigned)((total + 63) / 64)), dim3(64), lds, st, sig, n_frames,
                     frame_len, hop, P, r_out);
  if (hipGetLastError() != hipSuccess) *rc = fail(ALZ_E_HIP, "k_acorr_dense launch failed");
  note_kernel("k_acorr_dense", true);
  return true;
}

static int launch_lpc(const double *sig, int64_t n_frames, int frame_len, int64_t hop, int order,
                      double *coefs, double *err, int *status, double *r_out, int from_r,
                      hipStream_t st) {
  if (n_frames < 0 || frame_len < 1 || hop < 0 || order < 0)
    return fail(ALZ_E_ARG, "lpc: bad frame geometry");
  if (n_frames == 0) return ALZ_OK;
  if (order > 63) return fail(ALZ_E_UNSUPPORTED, "lpc: order > 63 runs the dense Levinson-Durbin only");
  const int slot = order <= 31 ? 32 : 64;
  const int fpw = 64 / slot;
  const size_t lds = (size_t)fpw * frame_len * sizeof(double);
  if (lds > 160 * 1024) return fail(ALZ_E_UNSUPPORTED, "lpc: frame does not fit the 160 KiB LDS");
  const dim3 grid((unsigned)((n_frames + fpw - 1) / fpw)), block(64);
  if (slot == 32) {
    if (lds > 64 * 1024)
      ALZ_HIP_CHECK(hipFuncSetAttribute((const void *)k_lpc<32>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lpc<32>, grid, block, lds, st, sig, n_frames, frame_len, hop, order, coefs,
                       err, status, r_out, from_r);
  } else {
    if (lds > 64 * 1024)
      ALZ_HIP_CHECK(hipFuncSetAttribute((const void *)k_lpc<64>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lpc<64>, grid, block, lds, st, sig, n_frames, frame_len, hop, order, coefs,
                       err, status, r_out, from_r);
  }
  ALZ_HIP_CHECK(hipGetLastError());
  note_kernel(std::string("k_lpc<") + (slot == 32 ? "32" : "64") + ">", true);
  return ALZ_OK;
}

}  // namespace alz

extern "C" {

int alz_lpc_kautocor_dev_ex(const double *sig_dev, int64_t n_frames, int frame_len, int64_t hop,
                            int order, double *coefs_dev, double *err_dev, int *status_dev, int flags, int device,
                            void *stream) {
  if (!sig_dev || !coefs_dev || !err_dev || !status_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  int rc = ALZ_OK;
  bool done = false;
  alz::note_kernel("");
  const bool dense = (flags & ALZ_LPC_DENSE) != 0 || order > 63;     // (orders past the register kernels: dense form only)
  if (n_frames > 0 && alz::stage_ok(sig_dev, n_frames, frame_len, hop)) {
    // one launch: autocorrelation and Levinson-Durbin in the same lane
    const bool fused = (flags & ALZ_LPC_FUSED) != 0;
    const bool ring2 = alz::stage_ring2(n_frames);
    const int P = order + 1;
    const bool nt = alz::stage_nt(n_frames, frame_len);
    alz::acorr_stage_fn fn =
        dense ? (fused ? nullptr
                       : ring2 ? (nt ? alz::pick_acorr_stage_dense<2, true>(P) : alz::pick_acorr_stage_dense<2>(P))
                               : ALZ_LPC_RING3(alz::pick_acorr_stage_dense<3>(P)))
        : fused ? (ring2 ? (nt ? alz::pick_acorr_stage<1, true, 2, true>(P) : alz::pick_acorr_stage<1, true, 2>(P))
                         : ALZ_LPC_RING3((alz::pick_acorr_stage<1, true, 3>(P))))
                : (ring2 ? (nt ? alz::pick_acorr_stage<1, false, 2, true>(P) : alz::pick_acorr_stage<1, false, 2>(P))
                         : ALZ_LPC_RING3((alz::pick_acorr_stage<1, false, 3>(P))));
    if (fn) {
      hipLaunchKernelGGL(fn, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), (ring2 ? 2 : 3) * 8192, (hipStream_t)stream,
                         sig_dev, n_frames, frame_len, hop, (double *)nullptr, coefs_dev, err_dev, status_dev);
      if (hipGetLastError() != hipSuccess) rc = alz::fail(ALZ_E_HIP, "k_acorr_stage launch failed");
      alz::note_kernel("k_acorr_stage<" + std::to_string(P) + (dense ? ",dense Levinson-Durbin" : fused ? ",lev,fma" : ",lev") +
                       (ring2 ? ",2 slots" : ",3 slots") + (nt ? ",nt>" : ">"));
      done = true;
    }
  }
  if (!done && dense && n_frames > 0) {
    // bit-exact end to end in two launches: the lags from the exact acorr kernels, then the reference's dense
    // Levinson-Durbin (small batches, odd frame geometry, orders without an unrolled kernel)
    double *r_tmp = nullptr;
    const size_t bytes = (size_t)n_frames * (order + 1) * sizeof(double);
    if (hipMallocAsync((void **)&r_tmp, bytes, (hipStream_t)stream) != hipSuccess)
      rc = alz::fail(ALZ_E_NOMEM, "lpc: scratch allocation failed");
    if (rc == ALZ_OK) {
      (void)alz::launch_acorr_dense(sig_dev, n_frames, frame_len, hop, order, r_tmp, (hipStream_t)stream, &rc);
      if (rc == ALZ_OK)
        rc = alz::launch_levinson_dense(r_tmp, n_frames, order + 1, order, coefs_dev, err_dev, status_dev,
                                        (hipStream_t)stream);
      alz::note_kernel("k_levinson_dense", true);
      (void)hipFreeAsync(r_tmp, (hipStream_t)stream);
    }
    done = true;
  }
  if (!done && order <= alz::kLevMax && n_frames > 0) {
    // two passes through a stream-ordered scratch array of lags
    double *r_tmp = nullptr;
    const size_t bytes = (size_t)n_frames * (order + 1) * sizeof(double);
    if (hipMallocAsync((void **)&r_tmp, bytes, (hipStream_t)stream) == hipSuccess) {
      if (alz::launch_acorr_dense(sig_dev, n_frames, frame_len, hop, order, r_tmp, (hipStream_t)stream, &rc)) {
        if (rc == ALZ_OK) {
          hipLaunchKernelGGL(alz::k_levinson_lane, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0,
                             (hipStream_t)stream, r_tmp, n_frames, order + 1, order, coefs_dev, err_dev,
                             status_dev);
          if (hipGetLastError() != hipSuccess) rc = alz::fail(ALZ_E_HIP, "k_levinson_lane launch failed");
          alz::note_kernel("k_levinson_lane", true);
        }
        done = true;
      }
      (void)hipFreeAsync(r_tmp, (hipStream_t)stream);
    } else {
      (void)hipGetLastError();
    }
  }
  if (!done)
    rc = alz::launch_lpc(sig_dev, n_frames, frame_len, hop, order, coefs_dev, err_dev, status_dev,
                         nullptr, 0, (hipStream_t)stream);
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

int alz_lpc_kautocor_dev(const double *sig_dev, int64_t n_frames, int frame_len, int64_t hop,
                         int order, double *coefs_dev, double *err_dev, int *status_dev, int device,
                         void *stream) {
  return alz_lpc_kautocor_dev_ex(sig_dev, n_frames, frame_len, hop, order, coefs_dev, err_dev, status_dev, 0, device,
                                 stream);
}

int alz_levinson_dev_ex(const double *r_dev, int64_t n_frames, int n_lags, int order,
                        double *coefs_dev, double *err_dev, int *status_dev, int flags, int device, void *stream) {
  if (!r_dev || !coefs_dev || !err_dev || !status_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (n_lags < 1) return alz::fail(ALZ_E_ARG, "levinson: need at least lag 0");
  if (n_frames < 0 || order < 0) return alz::fail(ALZ_E_ARG, "levinson: bad geometry");
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  int rc = ALZ_OK;
  alz::note_kernel("");
  if ((flags & ALZ_LPC_DENSE) != 0 || order > 63) {
    rc = alz::launch_levinson_dense(r_dev, n_frames, n_lags, order, coefs_dev, err_dev, status_dev, (hipStream_t)stream);
    alz::note_kernel("k_levinson_dense", true);
  } else if (order <= alz::kLevMax) {
    if (n_frames > 0) {
      hipLaunchKernelGGL(alz::k_levinson_lane, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0,
                         (hipStream_t)stream, r_dev, n_frames, n_lags, order, coefs_dev, err_dev, status_dev);
      if (hipGetLastError() != hipSuccess) rc = alz::fail(ALZ_E_HIP, "k_levinson_lane launch failed");
      alz::note_kernel("k_levinson_lane", true);
    }
  } else {
    rc = alz::launch_lpc(r_dev, n_frames, n_lags, n_lags, order, coefs_dev, err_dev, status_dev, nullptr, 1,
                         (hipStream_t)stream);
  }
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

int alz_levinson_dev(const double *r_dev, int64_t n_frames, int n_lags, int order,
                     double *coefs_dev, double *err_dev, int *status_dev, int device, void *stream) {
  return alz_levinson_dev_ex(r_dev, n_frames, n_lags, order, coefs_dev, err_dev, status_dev, 0, device, stream);
}

int alz_acorr_dev(const double *sig_dev, int64_t n_frames, int frame_len, int64_t hop, int max_lag,
                  double *r_dev, int device, void *stream) {
  if (!sig_dev || !r_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  int rc = ALZ_OK;
  alz::note_kernel("");
  if (n_frames <= 0 || !alz::launch_acorr_dense(sig_dev, n_frames, frame_len, hop, max_lag, r_dev,
                                                (hipStream_t)stream, &rc))
    rc = alz::launch_lpc(sig_dev, n_frames, frame_len, hop, max_lag, nullptr, nullptr, nullptr, r_dev, 0,
                         (hipStream_t)stream);
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

int alz_lag_matrix_dev(const double *sig_dev, int64_t n_frames, int frame_len, int64_t hop, int max_lag,
                       double *phi_dev, int device, void *stream) {
  if (!sig_dev || !phi_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (max_lag < 0 || frame_len < 1 || hop < 0 || n_frames < 0) return alz::fail(ALZ_E_ARG, "bad lag_matrix shape");
  if (max_lag >= frame_len) return alz::fail(ALZ_E_ARG, "Block length should be higher than order");
  const int64_t cells = (int64_t)(max_lag + 1) * (max_lag + 1);
  if (n_frames > 0 && cells > (INT64_C(0x7fffffff) * 256) / n_frames) return alz::fail(ALZ_E_ARG, "lag_matrix batch too large");
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  int rc = ALZ_OK;
  alz::note_kernel("");
  if (n_frames > 0) {
    const int P = max_lag + 1;
    if (alz::lag_rows_fn rows = alz::pick_lag_rows(P)) {    // the curated orders: a lane per row of the matrix
      const int64_t blocks = (n_frames * P + 255) / 256;
      rows<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(sig_dev, n_frames, frame_len, hop, phi_dev);
      alz::note_kernel("k_lag_matrix(rows)", true);
    } else {
      const int64_t blocks = (n_frames * cells + 255) / 256;
      alz::k_lag_matrix<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(sig_dev, n_frames, frame_len, hop, P,
                                                                                      phi_dev);
      alz::note_kernel("k_lag_matrix", true);
    }
    if (hipGetLastError() != hipSuccess) rc = alz::fail(ALZ_E_HIP, "k_lag_matrix launch failed");
  }
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

}  // extern "C"
