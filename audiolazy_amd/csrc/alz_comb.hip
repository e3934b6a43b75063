// alz_comb.hip -- sparse long-delay sections: comb filters and their relatives.
//
// Replaces LinearFilter.__call__ (reference audiolazy/lazy_filters.py:141-264) for filters such
// as comb.fb / comb.tau  y[n] = x[n] + alpha*y[n-D]  (lazy_filters.py:1090-1147), comb.ff
// (:1150-1173), their linearize()d fractional-delay forms (two adjacent taps, :339-373) and
// karplus_strong (lazy_synth.py:624-657): a handful of non-zero taps at large delays.  The
// reference's generated loop shifts all D memory variables on every sample (:254-255).
//
// What the structure gives (round 6): y[n] depends on y[n - D] and nothing nearer, so the D_min samples of one
// channel that follow any point in time are INDEPENDENT of each other -- a Karplus-Strong string is not one serial
// chain but D_min of them.  Both kernels advance a channel in STEPS of T <= D_min samples, all T computed side by
// side, the delay line a ring in LDS:
//
//   k_comb_tm  time-major blocks [N, C] (the reference's vector-valued samples): a workgroup owns 16 channels (128-byte
//              row pieces); thread = (row of the step, channel pair), 16-byte loads and stores, the next step's input
//              rows requested before the current step is computed; the y ring [D_max + T rows][16 channels] is shared
//              by the workgroup, one barrier per step.  Bound: HBM, 16 B per channel-sample.
//   k_comb_cm  channel-major blocks [C, N] and single strings: a WAVE owns a channel, lanes over the delay (lane l
//              takes samples n0 + l, n0 + 64 + l, ...).  Input arrives by 1 KiB global -> LDS transfers two chunks
//              ahead of the steps, output leaves from the y ring in 1 KiB stores once a chunk is complete, so the
//              memory traffic does not care how short a step is; no barriers (the rings are the wave's own, and the
//              LDS executes a wave's operations in order).  One string: a step costs the LDS round trip of the
//              delayed samples plus the DF-I sum -- T samples per ~250 cycles where the lane-per-channel kernel
//              (k_sparse, round 1) paid a memory round trip per 8 samples.  Many channels: HBM-bound.
//
// Arithmetic is the same bit-exact DF-I sum in both (ascending numerator delays, then ascending denominator
// delays, separately rounded mul / add, absent taps absent: lazy_filters.py:197-224).  In place when the
// numerator is the single tap b0 (comb.fb / comb.tau / karplus_strong: no input history to keep).  Shapes outside
// (more than three taps on a side, feedback delays under 16, rings larger than the LDS, ragged channel counts in
// time-major blocks) stay on k_sparse -- the block itself as the delay line, lane = channel -- or k_generic.
#include "alz_common.h"

namespace alz {

static constexpr int kMaxTaps = 8;      // per side
static constexpr int kMinDelay = 16;    // smallest feedback delay these kernels accept
static constexpr int kBatch = 8;        // k_sparse: rows per batch (< kMinDelay)

constexpr int kCombPaceGBps = 5900;          // k_comb_tm's common step clock (launch_sparse)

struct SArgs {
  const double *x;
  double *y;
  int64_t n, sxn, sxc, syn, syc;
  int64_t channels, n_inputs, n_sets;
  int64_t c_first, c_end;
  int mode, map_input;
  int nb, na;
  int nfb, nff;                 // number of present taps
  int kb[kMaxTaps], ka[kMaxTaps];
  const double *b, *a;
  const double *xh, *yh;        // histories: xh[k*C + c] = x[-1-k]
  int T;                        // k_comb_*: samples per step (<= the shortest feedback delay)
  int ring, xring;              // k_comb_tm: rows of the y ring; k_comb_cm: samples of the y / x ring (powers of two)
  int step_pace;                // k_comb_tm: the common step clock (alz_common.h pace_wait; 0: free-running)
};

__global__ __launch_bounds__(64) void k_sparse(SArgs p) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  int64_t in, set;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
  double bc[kMaxTaps], nac[kMaxTaps];
#pragma unroll
  for (int j = 0; j < kMaxTaps; ++j) {
    bc[j] = (j < p.nff) ? p.b[(int64_t)p.kb[j] * p.n_sets + set] : 0.0;
    nac[j] = (j < p.nfb) ? -p.a[(int64_t)p.ka[j] * p.n_sets + set] : 0.0;
  }
  const double *xc = p.x + in;
  double *yc = p.y + c;
  const double *xhc = p.xh + c;
  const double *yhc = p.yh + c;

  auto xval = [&](int64_t t) -> double { return t >= 0 ? xc[t * p.sxn] : xhc[(-t - 1) * p.channels]; };
  auto yval = [&](int64_t t) -> double { return t >= 0 ? yc[t * p.syn] : yhc[(-t - 1) * p.channels]; };

  for (int64_t n0 = 0; n0 < p.n; n0 += kBatch) {
    const int rows = (p.n - n0 < kBatch) ? (int)(p.n - n0) : kBatch;
    double xv[kMaxTaps][kBatch], yv[kMaxTaps][kBatch];
    // every load of the batch first (no feedback delay is shorter than the batch)
#pragma unroll
    for (int j = 0; j < kMaxTaps; ++j) {
      if (j < p.nff) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) xv[j][u] = (u < rows) ? xval(n0 + u - p.kb[j]) : 0.0;
      }
      if (j < p.nfb) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) yv[j][u] = (u < rows) ? yval(n0 + u - p.ka[j]) : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      if (u < rows) {
        double acc = -0.0;   // additive identity: the first present term initialises the sum
#pragma unroll
        for (int j = 0; j < kMaxTaps; ++j)
          if (j < p.nff) acc = acc + bc[j] * xv[j][u];
#pragma unroll
        for (int j = 0; j < kMaxTaps; ++j)
          if (j < p.nfb) acc = acc + nac[j] * yv[j][u];
        yc[(n0 + u) * p.syn] = acc;
      }
    }
  }
}

typedef double dbl2 __attribute__((ext_vector_type(2)));
// NT (a template parameter of both step kernels): non-temporal input loads and output stores for blocks of 256 MiB and more --
// measured +9 % time-major, +6 % channel-major, +10 % at 16 384 channels on the 4096-channel x 2^18 bank (round 6, call 4:
// profiles/r06_comb_shapes.log); smaller blocks, whose result the next call finds in the Infinity Cache, keep the default.

// ---------------------------------------------------------------------------------------------------------------
// k_comb_tm: time-major.  blockDim = 8 P threads (P rows per pass, a multiple of 8: whole waves), T = P U rows per step.
// ---------------------------------------------------------------------------------------------------------------
template <int NFF, int NFB, int U, bool NT>
__global__ __launch_bounds__(512) void k_comb_tm(SArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = (int)threadIdx.x, cp = tid & 7, r = tid >> 3;
  const int P = (int)blockDim.x >> 3;
  const int T = P * U, R = p.ring;
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 16 + 2 * cp;      // this thread's channels c, c + 1
  int64_t in, set0, set1;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;                              // (n_inputs % 16 == 0: the pair shares a set)
    set0 = set1 = c / p.n_inputs;
  } else {
    in = c;
    set0 = (p.n_sets == 1) ? 0 : c;
    set1 = (p.n_sets == 1) ? 0 : c + 1;
  }
  dbl2 bc[NFF > 0 ? NFF : 1], nac[NFB > 0 ? NFB : 1];
  int kb[NFF > 0 ? NFF : 1], ka[NFB > 0 ? NFB : 1];
#pragma unroll
  for (int j = 0; j < NFF; ++j) {
    kb[j] = p.kb[j];
    bc[j] = dbl2{p.b[(int64_t)kb[j] * p.n_sets + set0], p.b[(int64_t)kb[j] * p.n_sets + set1]};
  }
#pragma unroll
  for (int j = 0; j < NFB; ++j) {
    ka[j] = p.ka[j];
    nac[j] = dbl2{-p.a[(int64_t)ka[j] * p.n_sets + set0], -p.a[(int64_t)ka[j] * p.n_sets + set1]};
  }
  dbl2 *ring = reinterpret_cast<dbl2 *>(smem);                          // ring[row * 8 + cp]; row of time t: t mod R
  if constexpr (NFB > 0) {
    // the delay line before the block: time -1-k sits R - 1 - k rows into the ring
    const double *yhb = p.yh + c;
    const int64_t Cy = p.channels;
    for (int k = r; k < ka[NFB - 1]; k += P) ring[(R - 1 - k) * 8 + cp] = *reinterpret_cast<const dbl2 *>(yhb + (int64_t)k * Cy);
  }
  // every load unconditional (a row past the block reads the block's last row and is never stored): no branches and no
  // waits between the requests of a step
  const int64_t N = p.n, sxn = p.sxn, syn = p.syn, C = p.channels;
  const double *xb = p.x + in, *xhb = p.xh + c;
  double *yb = p.y + c;
  auto xrow = [&](int64_t t) -> dbl2 {
    t = t < N ? t : N - 1;
    const double *src = t >= 0 ? xb + t * sxn : xhb + (-t - 1) * C;
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const dbl2 *>(src));
    else return *reinterpret_cast<const dbl2 *>(src);
  };
  dbl2 xv[U][NFF > 0 ? NFF : 1];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t n = r + P * u;
#pragma unroll
    for (int j = 0; j < NFF; ++j) xv[u][j] = xrow(n - kb[j]);
  }
  if constexpr (NFB > 0) __syncthreads();
  int q0 = 0;                                                           // n0 mod R
  const long long pace0 = p.step_pace > 0 ? (long long)wall_clock64() : 0;

  long long pace_shift = 0;
  long long step = 0;
  for (int64_t n0 = 0; n0 < N; n0 += T, ++step) {
    if (p.step_pace > 0) pace_wait(pace0, step, p.step_pace, pace_shift);
    // the next step's input rows first: they travel while this step is computed
    dbl2 xn[U][NFF > 0 ? NFF : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t n = n0 + T + r + P * u;
#pragma unroll
      for (int j = 0; j < NFF; ++j) xn[u][j] = xrow(n - kb[j]);
    }
    dbl2 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      dbl2 v = dbl2{-0.0, -0.0};                                        // additive identity: the first present term initialises the sum
#pragma unroll
      for (int j = 0; j < NFF; ++j) v = v + bc[j] * xv[u][j];
#pragma unroll
      for (int j = 0; j < NFB; ++j) {
        int pos = q0 + r + P * u - ka[j];                               // in (-R, R): T <= ka[j] <= R - T
        pos += pos < 0 ? R : 0;
        v = v + nac[j] * ring[pos * 8 + cp];
      }
      acc[u] = v;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t n = n0 + r + P * u;
      if (n < N) {
        if constexpr (NT) __builtin_nontemporal_store(acc[u], reinterpret_cast<dbl2 *>(yb + n * syn));
        else *reinterpret_cast<dbl2 *>(yb + n * syn) = acc[u];
        if constexpr (NFB > 0) {
          int w = q0 + r + P * u;
          w -= w >= R ? R : 0;
          ring[w * 8 + cp] = acc[u];
        }
      }
    }
    if constexpr (NFB > 0) __syncthreads();                             // this step's rows are the next steps' delay line
    q0 += T;
    q0 -= q0 >= R ? R : 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NFF; ++j) xv[u][j] = xn[u][j];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_comb_cm: channel-major / single strings.  A wave per channel, lanes over the delay.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int kCombChunk = 256;   // samples per chunk of the wave's global traffic (two 1 KiB transfers)

template <bool NT>
__device__ __forceinline__ void comb_dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <bool NT>
__device__ __forceinline__ void comb_store16(double *gdst, dbl2 v) {
  if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
}

template <int NFF, int NFB, int U, bool NT>
__global__ __launch_bounds__(256) void k_comb_cm(SArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = kCombChunk;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)threadIdx.x & 63;
  const int wpb = (int)blockDim.x >> 6;
  const int64_t c = p.c_first + (int64_t)blockIdx.x * wpb + wave;
  if (c >= p.c_end) return;                                             // (no barriers in this kernel)
  int64_t in, set;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
  double bc[NFF > 0 ? NFF : 1], nac[NFB > 0 ? NFB : 1];
  int kb[NFF > 0 ? NFF : 1], ka[NFB > 0 ? NFB : 1];
#pragma unroll
  for (int j = 0; j < NFF; ++j) { kb[j] = p.kb[j]; bc[j] = p.b[(int64_t)kb[j] * p.n_sets + set]; }
#pragma unroll
  for (int j = 0; j < NFB; ++j) { ka[j] = p.ka[j]; nac[j] = -p.a[(int64_t)ka[j] * p.n_sets + set]; }
  // (in registers NOW: a value the compiler still has in flight when the loop starts gets its s_waitcnt vmcnt(0) INSIDE the loop,
  // where it drains the transfers queued ahead -- k_string's first build spent 1.2 us per step that way)
#pragma unroll
  for (int j = 0; j < NFF; ++j) asm volatile("" : "+v"(bc[j]));
#pragma unroll
  for (int j = 0; j < NFB; ++j) asm volatile("" : "+v"(nac[j]));
  const int RX = p.xring, RY = p.ring, MX = RX - 1, MY = RY - 1;        // powers of two
  double *xr = reinterpret_cast<double *>(smem) + (size_t)wave * (RX + RY);   // x[t] at xr[t & MX]
  double *yr = xr + RX;                                                 // y[t] at yr[t & MY]
  const unsigned xr_lds = (unsigned)(uintptr_t)xr;
  const double *xc = p.x + in * p.sxc;
  double *yc = p.y + c * p.syc;
  // the delay lines before the block
  if constexpr (NFF > 0)
    for (int k = lane; k < kb[NFF - 1]; k += 64) xr[(-1 - k) & MX] = p.xh[(int64_t)k * p.channels + c];
  if constexpr (NFB > 0)
    for (int k = lane; k < ka[NFB - 1]; k += 64) yr[(-1 - k) & MY] = p.yh[(int64_t)k * p.channels + c];
  const int64_t n_chunks = (p.n + CH - 1) / CH;
  // A few channels of a TIME-major block (stereo rows [N, 2]: sxn = channels) come through the same kernel with strided
  // plain loads / stores: no 16-byte pieces, the transfers are not queued ahead -- still lanes over the delay line instead
  // of one lane per channel.
  const int64_t sxn = p.sxn, syn = p.syn;
  const bool dma_in = sxn == 1, wide_out = syn == 1;
  // chunk k of the input into the x ring: whole chunks by global -> LDS transfers (nothing waits here), the ragged last
  // one by plain loads
  auto fetch = [&](int64_t k) {
    if (k >= n_chunks) return;
    const int64_t t0 = k * CH;
    if (dma_in && t0 + CH <= p.n) {
#pragma unroll
      for (int i = 0; i < CH / 128; ++i)
        comb_dma16<NT>(xc + t0 + 128 * i + 2 * lane, xr_lds + (unsigned)(((int)(t0 + 128 * i) & MX) * 8));
    } else {
      const int64_t te = t0 + CH < p.n ? t0 + CH : p.n;
      for (int64_t t = t0 + lane; t < te; t += 64) xr[(int)t & MX] = xc[t * sxn];
    }
  };
  fetch(0);
  fetch(1);
  const int T = p.T;
  for (int64_t k = 0; k < n_chunks; ++k) {
    // chunk k has landed when at most the transfers and stores issued after it are outstanding: F(k) S(k-2) F(k+1) S(k-1)
    // -- two operations each; near the end of the block (ragged chunk: compiler-issued loads) simply everything
    // (the first two iterations have fewer behind them: everything)
    if (dma_in && wide_out && k >= 2 && k + 3 < n_chunks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fetch(k + 2);
    const int64_t t_end = (k + 1) * CH < p.n ? (k + 1) * CH : p.n;
    for (int64_t n0 = k * CH; n0 < t_end; n0 += T) {
      const int Tn = t_end - n0 < T ? (int)(t_end - n0) : T;
      const int tb = (int)n0 + lane;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (lane + 64 * u < Tn) {
          const int t = tb + 64 * u;
          double v = -0.0;                                              // additive identity: the first present term initialises the sum
#pragma unroll
          for (int j = 0; j < NFF; ++j) v = v + bc[j] * xr[(t - kb[j]) & MX];
#pragma unroll
          for (int j = 0; j < NFB; ++j) v = v + nac[j] * yr[(t - ka[j]) & MY];
          yr[t & MY] = v;
        }
      }
    }
    // the finished chunk leaves from the ring
    const int64_t t0 = k * CH;
    if (wide_out && t0 + CH <= p.n) {
#pragma unroll
      for (int i = 0; i < CH / 128; ++i) {
        const dbl2 v = *reinterpret_cast<const dbl2 *>(&yr[((int)t0 + 128 * i + 2 * lane) & MY]);
        comb_store16<NT>(yc + t0 + 128 * i + 2 * lane, v);
      }
    } else {
      const int64_t te = t0 + CH < p.n ? t0 + CH : p.n;
      for (int64_t t = t0 + lane; t < te; t += 64) yc[t * syn] = yr[(int)t & MY];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_string: one Karplus-Strong string (or a few) -- a step is exactly ONE PERIOD of the delay line, held in registers.
//
// y[n] = b0 x[n] + (-a_D) y[n - D] (+ (-a_{D+1}) y[n - D - 1]): with a step of D samples, lane l's slot u is sample
// k D + 64 u + l of step k, and y[n - D] is the SAME lane's slot of the step before -- a register.  The second tap of a
// linearize()d delay (lazy_filters.py:339-373) is that register's left neighbour: one wavefront shift (v_mov_b32_dpp
// wave_shr:1, two per double), lane 0 of a slot taking lane 63 of the slot before it and, for the period's first sample,
// the period's LAST sample of two steps ago.  The LDS round trip that k_comb_cm has between a step and the next is gone
// from the chain: a step costs its own ~15 instructions per slot.  x still arrives by 1 KiB global -> LDS transfers, three
// groups of 1024 samples ahead of the steps, and y leaves from an LDS ring in 1 KiB stores (both off the chain; the first
// builds did this bookkeeping per 256-sample chunk inside the step loop and were SLOWER than k_comb_cm: ~1100 cycles per step).  D <= 512 (eight slots), numerator b0
// alone, channel-major rows / single strings.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double shr1(double v, double lane0) {
  // every lane takes its left neighbour's v; lane 0 takes lane0 (wave-uniform)
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_of(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void comb_wait_vm(int n) {    // at most n vector-memory operations outstanding (n even; rounded down)
#define ALZ_VMC(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * k) : "memory"); break;
  switch (n >> 1) {
    ALZ_VMC(0) ALZ_VMC(1) ALZ_VMC(2) ALZ_VMC(3) ALZ_VMC(4) ALZ_VMC(5) ALZ_VMC(6) ALZ_VMC(7) ALZ_VMC(8) ALZ_VMC(9) ALZ_VMC(10) ALZ_VMC(11)
    ALZ_VMC(12) ALZ_VMC(13) ALZ_VMC(14) ALZ_VMC(15) ALZ_VMC(16) ALZ_VMC(17) ALZ_VMC(18) ALZ_VMC(19) ALZ_VMC(20) ALZ_VMC(21) ALZ_VMC(22) ALZ_VMC(23)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
#undef ALZ_VMC
}
// k_string: its transfers and stores go in groups of 1024 samples (eight 1 KiB operations), three groups of input in flight --
// a step of one period takes ~0.1 us, a transfer ~2 us to land -- so the bookkeeping is per 1024 samples, not per step
constexpr int kStringChunk = 1024, kStringXRing = 4 * kStringChunk, kStringYRing = 2 * kStringChunk;

template <int NFB, int U, bool NT>
__global__ __launch_bounds__(256) void k_string(SArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SC = kStringChunk;                                       // samples per group of transfers / stores (eight 1 KiB operations)
  constexpr int MX = kStringXRing - 1, MY = kStringYRing - 1;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)threadIdx.x & 63;
  const int wpb = (int)blockDim.x >> 6;
  const int64_t c = p.c_first + (int64_t)blockIdx.x * wpb + wave;
  if (c >= p.c_end) return;                                             // (no barriers in this kernel)
  int64_t in, set;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
  const int D = p.ka[0];
  double b0 = p.b[set];
  double na1 = -p.a[(int64_t)D * p.n_sets + set];
  double na2 = 0.0;
  if constexpr (NFB == 2) na2 = -p.a[(int64_t)(D + 1) * p.n_sets + set];
  double *xr = reinterpret_cast<double *>(smem) + (size_t)wave * (kStringXRing + kStringYRing);   // x[t] at xr[t & MX]
  double *yr = xr + kStringXRing;                                                                 // y[t] at yr[t & MY]
  const unsigned xr_lds = (unsigned)(uintptr_t)xr;
  const double *xc = p.x + in * p.sxc;
  double *yc = p.y + c * p.syc;
  const int N = (int)p.n;                                               // (the launcher keeps blocks below 2^31 samples)
  const int last_lane = (D - 1) & 63;
  // the period before the block: slot u of lane l is y[-D + 64 u + l] = yh[D - 1 - 64 u - l]
  double prev[U], old_last = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = 64 * u + lane;
    prev[u] = idx < D ? p.yh[(int64_t)(D - 1 - idx) * p.channels + c] : 0.0;
  }
  if constexpr (NFB == 2) old_last = p.yh[(int64_t)D * p.channels + c];   // y[-D-1]
  // (everything loaded so far is in registers NOW: a value still in flight when the loop starts gets its s_waitcnt vmcnt(0)
  // INSIDE the loop, where it drains the transfers queued ahead)
  asm volatile("" : "+v"(b0), "+v"(na1), "+v"(na2), "+v"(old_last));
#pragma unroll
  for (int u = 0; u < U; ++u) asm volatile("" : "+v"(prev[u]));
  const int n_sc = (N + SC - 1) / SC, full_sc = N / SC;                   // groups; those that are whole
  // Vector-memory operations of this wave, in issue order: fetch groups F(k) and store groups S(k) of eight operations each
  // (whole groups; the ragged last group goes through compiler-issued loads / stores and is simply waited for in full).
  // F(0..2) up front; after the steps of group k: S(k), then F(k + 3).  F(k) has landed when at most the groups issued
  // after it are outstanding: F(k+1), F(k+2) and S(k-1) .. -- up to four groups, 32 operations.
  // (a few channels of a time-major block -- stereo rows [N, 2] -- take the strided plain path for every group: see k_comb_cm)
  const int64_t sxn = p.sxn, syn = p.syn;
  const bool dma_in = sxn == 1, wide_out = syn == 1, queued = dma_in && wide_out;
  auto fetch = [&](int k) {
    if (k >= n_sc) return;
    const int t0 = k * SC;
    if (dma_in && k < full_sc) {
#pragma unroll
      for (int i = 0; i < SC / 128; ++i) comb_dma16<NT>(xc + t0 + 128 * i + 2 * lane, xr_lds + (unsigned)(((t0 + 128 * i) & MX) * 8));
    } else {
      const int te = t0 + SC < N ? t0 + SC : N;
      for (int t = t0 + lane; t < te; t += 64) xr[t & MX] = xc[(int64_t)t * sxn];
    }
  };
  auto store_group = [&](int k) {
    const int t0 = k * SC;
    if (wide_out && k < full_sc) {
#pragma unroll
      for (int i = 0; i < SC / 128; ++i) {
        const dbl2 v = *reinterpret_cast<const dbl2 *>(&yr[(t0 + 128 * i + 2 * lane) & MY]);
        comb_store16<NT>(yc + t0 + 128 * i + 2 * lane, v);
      }
    } else {
      const int te = t0 + SC < N ? t0 + SC : N;
      for (int t = t0 + lane; t < te; t += 64) yc[(int64_t)t * syn] = yr[t & MY];
    }
  };
  fetch(0);
  fetch(1);
  fetch(2);
  int landed_end = 0, landed = 0, stored = 0;                             // samples whose input has landed; groups landed / stored
  for (int n0 = 0; n0 < N; n0 += D) {
    const int rem = N - n0 < D ? N - n0 : D;
    while (n0 + rem > landed_end) {                                       // (once per group)
      // groups issued after F(landed): the fetches up to F(stored + 2) and the stores S(landed - 1 .. stored - 1) issued since
      // (issue order F0 F1 F2 S0 F3 S1 F4 ...: F(j), j >= 3, follows S(j - 3))
      const int fetches_after = (stored + 2 < n_sc - 1 ? stored + 2 : n_sc - 1) - landed;
      const int stores_after = landed <= 2 ? stored : stored - (landed - 2);
      const int after = fetches_after + stores_after;
      if (!queued || landed >= full_sc - 3 || after < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the ragged tail, strided rows: everything)
      else comb_wait_vm(8 * after);
      ++landed;
      landed_end = landed * SC < N ? landed * SC : N;
    }
    double carry = old_last, nxt[U], xv[U], left[U];
    if constexpr (NFB == 2) old_last = lane_of(prev[U - 1], last_lane);   // the period's last sample, before this step replaces it
    const int tb = n0 + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = xr[(tb + 64 * u) & MX];           // (a slot past the period reads something: never used)
    if constexpr (NFB == 2) {
#pragma unroll
      for (int u = 0; u < U; ++u) {                                       // (every lane active: the shift reads its neighbour's register)
        left[u] = shr1(prev[u], carry);
        carry = lane_of(prev[u], 63);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double v = b0 * xv[u] + na1 * prev[u];
      if constexpr (NFB == 2) v = v + na2 * left[u];
      const bool on = 64 * u + lane < rem;
      if (on) yr[(tb + 64 * u) & MY] = v;
      nxt[u] = on ? v : prev[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) prev[u] = nxt[u];
    // a finished group leaves; the x ring then has room for one more
    while ((stored + 1) * SC <= n0 + rem) {
      store_group(stored);
      ++stored;
      fetch(stored + 2);
    }
  }
  if (stored < n_sc) store_group(stored);                                 // the ragged last group
}

// histories after the block, for both delay lines (written to the spare halves of the slabs)
__global__ void k_sparse_state(SArgs p, double *xh_new, double *yh_new, int do_x, int do_y) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? c % p.n_inputs : c;
  if (do_x)
    for (int k = (int)blockIdx.y; k < p.nb - 1; k += (int)gridDim.y) {
      const int64_t t = p.n - 1 - k;
      xh_new[(int64_t)k * p.channels + c] = (t >= 0) ? p.x[t * p.sxn + in * p.sxc] : p.xh[(-t - 1) * p.channels + c];
    }
  if (do_y)
    for (int k = (int)blockIdx.y; k < p.na - 1; k += (int)gridDim.y) {
      const int64_t t = p.n - 1 - k;
      yh_new[(int64_t)k * p.channels + c] = (t >= 0) ? p.y[t * p.syn + c * p.syc] : p.yh[(-t - 1) * p.channels + c];
    }
}

__global__ void k_copy_rows(double *dst, const double *src, int64_t count, int64_t stride, int rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  for (int k = (int)blockIdx.y; k < rows; k += (int)gridDim.y) dst[(int64_t)k * stride + i] = src[(int64_t)k * stride + i];
}

typedef void (*comb_fn)(SArgs);
template <int NFF, int NFB, bool NT>
static comb_fn pick_tm_u(int u) {
  switch (u) {
    case 1: return (comb_fn)k_comb_tm<NFF, NFB, 1, NT>;
    case 2: return (comb_fn)k_comb_tm<NFF, NFB, 2, NT>;
    case 4: return (comb_fn)k_comb_tm<NFF, NFB, 4, NT>;
    default: return (comb_fn)k_comb_tm<NFF, NFB, 6, NT>;
  }
}
template <int NFF, int NFB, bool NT>
static comb_fn pick_cm_u(int u) {
  switch (u) {
    case 1: return (comb_fn)k_comb_cm<NFF, NFB, 1, NT>;
    case 2: return (comb_fn)k_comb_cm<NFF, NFB, 2, NT>;
    default: return (comb_fn)k_comb_cm<NFF, NFB, 4, NT>;
  }
}
static comb_fn pick_comb(bool cm, int nff, int nfb, int u, bool nt = false) {
#define ALZ_COMB(F, B) if (nff == F && nfb == B) return cm ? (nt ? pick_cm_u<F, B, true>(u) : pick_cm_u<F, B, false>(u)) : (nt ? pick_tm_u<F, B, true>(u) : pick_tm_u<F, B, false>(u));
  // comb.fb / comb.tau, their linearize()d forms and karplus_strong; comb.ff and its linearize()d form; a numerator pair or triple in front of either
  ALZ_COMB(1, 1) ALZ_COMB(1, 2) ALZ_COMB(2, 0) ALZ_COMB(3, 0) ALZ_COMB(2, 1) ALZ_COMB(2, 2) ALZ_COMB(3, 1) ALZ_COMB(3, 2)
#undef ALZ_COMB
  return nullptr;
}

template <int NFB, bool NT>
static comb_fn pick_string_u(int u) {
  switch (u) {
    case 1: return (comb_fn)k_string<NFB, 1, NT>;
    case 2: return (comb_fn)k_string<NFB, 2, NT>;
    case 3: return (comb_fn)k_string<NFB, 3, NT>;
    case 4: return (comb_fn)k_string<NFB, 4, NT>;
    case 5: return (comb_fn)k_string<NFB, 5, NT>;
    case 6: return (comb_fn)k_string<NFB, 6, NT>;
    case 7: return (comb_fn)k_string<NFB, 7, NT>;
    default: return (comb_fn)k_string<NFB, 8, NT>;
  }
}
static comb_fn pick_string(int nfb, int u, bool nt) {
  return nfb == 1 ? (nt ? pick_string_u<1, true>(u) : pick_string_u<1, false>(u)) : (nt ? pick_string_u<2, true>(u) : pick_string_u<2, false>(u));
}

struct CombPlan {
  bool string = false;              // k_string: a period of the delay line in registers
  bool ok = false, cm = false;
  int u = 1, threads = 0, T = 0, ring = 0, xring = 0;
  size_t lds = 0;
  unsigned grid = 0;
};

// Which of the two step kernels takes this section and block, and how (nothing is launched).
static CombPlan plan_comb(const SectionDev &sec, const BlockIO &io) {
  CombPlan pl;
  if (sec.n_ff < 1 || sec.n_ff > 3 || sec.n_fb < 0 || sec.n_fb > 2 || !sec.uniform || sec.any_div) return pl;
  if (!pick_comb(true, sec.n_ff, sec.n_fb, 1)) return pl;              // (not one of the instantiated tap counts)
  if (ALZ_TUNE("ALZ_COMB_OFF", 0)) return pl;                           // (tuning builds: round 1's k_sparse, for A/B timing)
  const int dmin = sec.n_fb ? sec.tap_a[0] : (1 << 20);
  const int kamax = sec.n_fb ? sec.tap_a[sec.n_fb - 1] : 0, kbmax = sec.tap_b[sec.n_ff - 1];
  if (dmin < kMinDelay) return pl;
  if (io.x == io.y && !(sec.n_ff == 1 && sec.tap_b[0] == 0)) return pl;      // (in place: no input history may be read back)
  if (io.mode == ALZ_BANK_OUTER && io.x == io.y) return pl;
  if (((uintptr_t)io.x | (uintptr_t)io.y) & 15) return pl;
  const bool tm = io.sxc == 1 && io.syc == 1;
  // a few channels of a time-major block (stereo rows [N, 2], ragged counts up to 64): the wave-per-channel kernels with strided
  // plain loads and stores -- lanes over the delay line still beat a lane per channel by far
  const bool narrow_tm = tm && !(io.sxn == 1 && io.syn == 1) && io.c_count <= 64 && (io.c_count < 16 || io.c_count % 16 != 0) &&
                         io.mode != ALZ_BANK_OUTER;
  const bool cm = (io.sxn == 1 && io.syn == 1) || narrow_tm;
  if (cm) {
    // a wave per channel; 16-byte pieces of a channel's row
    if (!narrow_tm && io.channels > 1 && ((io.sxc | io.syc) & 1)) return pl;
    // a few strings whose period fits a wave's registers: steps of one period, no LDS round trip between them
    if (io.c_count <= 256 && dmin <= 512 && sec.n_ff == 1 && sec.tap_b[0] == 0 && sec.n_fb >= 1 &&
        (sec.n_fb == 1 || sec.tap_a[1] == sec.tap_a[0] + 1) && !ALZ_TUNE("ALZ_STRING_OFF", 0)) {
      const int wpb = io.c_count < 2 ? 1 : 2;                          // (48 KiB of rings per wave)
      pl.ok = true; pl.cm = true; pl.string = true; pl.u = (dmin + 63) / 64; pl.threads = 64 * wpb; pl.T = dmin; pl.ring = kStringYRing; pl.xring = kStringXRing;
      pl.lds = (size_t)(kStringXRing + kStringYRing) * 8 * wpb;
      pl.grid = (unsigned)((io.c_count + wpb - 1) / wpb);
      return pl;
    }
    int u = dmin >= 256 ? 4 : dmin > 64 ? 2 : 1;
    if (u == 2 && dmin > 128) u = 4;
    int T = dmin < 64 * u ? dmin : 64 * u;
    if (T > kCombChunk) T = kCombChunk;
    int ry = 1, rx = 1;
    while (ry < kamax + kCombChunk) ry <<= 1;
    while (rx < kbmax + 3 * kCombChunk) rx <<= 1;
    const size_t per_wave = (size_t)(rx + ry) * 8;
    if (per_wave > 64 * 1024) return pl;
    int wpb = (int)((64 * 1024) / per_wave);
    wpb = wpb > 4 ? 4 : wpb;
    if (io.c_count < wpb) wpb = (int)io.c_count;
    pl.ok = true; pl.cm = true; pl.u = u; pl.threads = 64 * wpb; pl.T = T; pl.ring = ry; pl.xring = rx;
    pl.lds = per_wave * wpb;
    pl.grid = (unsigned)((io.c_count + wpb - 1) / wpb);
    return pl;
  }
  if (!tm) return pl;
  if (io.c_count % 16 || io.c_first % 2 || io.channels % 2 || ((io.sxn | io.syn) & 1)) return pl;
  if (io.mode == ALZ_BANK_OUTER && io.map_input && io.n_inputs % 16) return pl;
  // rows per step T = P U <= dmin, P (rows per pass) a multiple of 8 up to 64, U in {1, 2, 4, 6}
  int u = (dmin + 63) / 64;
  u = u >= 6 ? 6 : u >= 4 ? 4 : u >= 2 ? 2 : 1;
  int P = dmin / u / 8 * 8;
  P = P > 64 ? 64 : P;
  if (P < 8) return pl;
  int T = P * u;
  int R = kamax ? ((kamax + T + 7) & ~7) : 0;
  while ((size_t)R * 128 > 150 * 1024 && u > 1) {     // a long delay line: shorter steps
    u = u == 6 ? 4 : u == 4 ? 2 : 1;
    T = P * u;
    R = (kamax + T + 7) & ~7;
  }
  if ((size_t)R * 128 > 150 * 1024) return pl;
  pl.ok = true; pl.cm = false; pl.u = u; pl.threads = 8 * P; pl.T = T; pl.ring = R; pl.xring = 0;
  pl.lds = (size_t)R * 128;
  pl.grid = (unsigned)(io.c_count / 16);
  return pl;
}

bool comb_takes_in_place(const SectionDev &sec, const BlockIO &io) { return io.x == io.y && plan_comb(sec, io).ok; }

// Sparse section with every feedback delay >= kMinDelay, a0 == 1, uniform zero pattern.  The host-side tap lists
// come from the coefficient scan done at create time.
int launch_sparse(const SectionDev &sec, const BlockIO &io, hipStream_t stream, bool *taken,
                  const char **kernel_name) {
  *taken = false;
  if (sec.n_ff < 0 || sec.n_ff > kMaxTaps || sec.n_fb > kMaxTaps || sec.n_fb < 0) return ALZ_OK;
  if (sec.n_ff + sec.n_fb == 0 || !sec.uniform || sec.any_div) return ALZ_OK;
  const CombPlan pl = plan_comb(sec, io);
  const bool sparse_ok = sec.n_fb >= 1 && sec.tap_a[0] >= kMinDelay && io.sxc == 1 && io.syc == 1 && io.x != io.y;
  if (!pl.ok && !sparse_ok) return ALZ_OK;
  SArgs p;
  p.x = io.x; p.y = io.y; p.n = io.n; p.sxn = io.sxn; p.sxc = io.sxc; p.syn = io.syn; p.syc = io.syc;
  p.channels = io.channels; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.c_first = io.c_first; p.c_end = io.c_first + io.c_count;
  p.mode = io.mode; p.map_input = io.map_input;
  p.nb = sec.nb; p.na = sec.na; p.nff = sec.n_ff; p.nfb = sec.n_fb;
  for (int j = 0; j < kMaxTaps; ++j) { p.kb[j] = sec.tap_b[j]; p.ka[j] = sec.tap_a[j]; }
  p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.yh = sec.yh;
  p.T = pl.T; p.ring = pl.ring; p.xring = pl.xring;
  // k_comb_tm's workgroups on one step clock (alz_common.h pace_wait; grid x 16 channels x T rows x 16 B per step) while they are all
  // resident (one per CU).  4096 channels x 2^18, D = 441 (profiles/r06_pace_others.log, r06_pace2.log): free-running 284 - 300
  // Gsamples/s, 5600 GB/s 301 - 313, 5900 306 - 311, 6200 314 - 317, 6500 325 / 296 (the knee).
  const int cus = device_cus() > 0 ? device_cus() : 256;
  // (several full rounds of workgroups on the clock, as k_tvpc has them: 8192 channels 313 - 316 -> 289 / 314, 12288 channels 293 - 310 -> 325,
  // 7680 channels 299 - 302 -> 305 - 313, profiles/r06_pace_rounds.log: not a clear gain here, so launches above the CU count run free)
  p.step_pace = (pl.ok && !pl.cm && !pl.string && (int)pl.grid <= cus) ? tile_pace16((long long)pl.grid * 256ll * pl.T, ALZ_TUNE("ALZ_COMB_PACE_GBPS", kCombPaceGBps)) : 0;
  const unsigned gx = (unsigned)((io.c_count + 63) / 64);
  const int64_t nx = (int64_t)(sec.nb - 1) * io.channels, ny = (int64_t)(sec.na - 1) * io.channels;
  double *xh_new = sec.xh + nx, *yh_new = sec.yh + ny;
  const bool nt = (uint64_t)io.n * (uint64_t)io.c_count * 8u >= (256ull << 20);
  comb_fn fn = !pl.ok ? nullptr : pl.string ? pick_string(sec.n_fb, pl.u, nt) : pick_comb(pl.cm, sec.n_ff, sec.n_fb, pl.u, nt);
  if (fn) {
    if (pl.lds > 0) {
      const int rc = ensure_dynamic_lds((const void *)fn, (int)pl.lds);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(fn, dim3(pl.grid), dim3((unsigned)pl.threads), pl.lds, stream, p);
    *kernel_name = pl.string ? "k_string" : pl.cm ? "k_comb_cm" : "k_comb_tm";
  } else {
    if (!sparse_ok) return ALZ_OK;
    hipLaunchKernelGGL(k_sparse, dim3(gx), dim3(64), 0, stream, p);
    *kernel_name = "k_sparse";
  }
  if (sec.nb > 1 || sec.na > 1)
    hipLaunchKernelGGL(k_sparse_state, dim3(gx, 16), dim3(64), 0, stream, p, xh_new, yh_new, sec.nb > 1 ? 1 : 0, sec.na > 1 ? 1 : 0);
  const unsigned gc = (unsigned)((io.c_count + 255) / 256);
  if (sec.nb > 1)
    hipLaunchKernelGGL(k_copy_rows, dim3(gc, 16), dim3(256), 0, stream, sec.xh + io.c_first,
                       xh_new + io.c_first, io.c_count, io.channels, sec.nb - 1);
  if (sec.na > 1)
    hipLaunchKernelGGL(k_copy_rows, dim3(gc, 16), dim3(256), 0, stream, sec.yh + io.c_first,
                       yh_new + io.c_first, io.c_count, io.channels, sec.na - 1);
  ALZ_HIP_CHECK(hipGetLastError());
  *taken = true;
  return ALZ_OK;
}

}  // namespace alz
