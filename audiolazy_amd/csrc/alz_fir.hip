// alz_fir.hip -- long tapped-delay-line FIR, time-parallel, bit-exact.
//
// Replaces LinearFilter.__call__ (reference audiolazy/lazy_filters.py:141-264) for the
// feedback-free case  y[n] = (b0*x[n] + b1*x[n-1] + ... + b_{nb-1}*x[n-nb+1]) / a0  with
// many taps (BASELINE config 3: 256 taps x 8192 channels).  No recurrence, so time is
// parallel too: a wave owns 64 adjacent channels (lane = channel, 512-byte coalesced rows of
// the time-major block) and a run of TB output rows; it keeps R = 16 output accumulators per
// lane in VGPRs and walks the taps in ascending order in blocks of K = 8, so every output sees
// exactly the reference's left-to-right sum  ((b0*d0 + b1*d1) + b2*d2) + ...  with separately
// rounded multiply and add (file built with -ffp-contract=off).  acc starts at -0.0, which is
// the additive identity for every double, and zero taps are skipped (they are absent from the
// reference's expression, lazy_filters.py:209).
//
// Taps shared by the whole bank (n_sets == 1) are wave-uniform and live in SGPRs (scalar loads);
// per-channel taps are one coalesced vector load per tap.  The x rows a tap block needs
// (R + K - 1 of them) come from L1/L2: neighbouring time blocks of the same channel group re-read
// each other's rows, HBM sees each row about once.
//
// Bound: FP64 issue, not HBM: 2*nb - 1 f64 ops per output sample (511 for 256 taps) at one op
// per ~4.5 cycles per SIMD.
#include "alz_common.h"

namespace alz {

#ifndef ALZ_FIR_R
#define ALZ_FIR_R 32
#endif
#ifndef ALZ_FIR_K
#define ALZ_FIR_K 16
#endif
static constexpr int kFirR = ALZ_FIR_R;   // outputs per lane held in registers
static constexpr int kFirK = ALZ_FIR_K;   // taps per block
static constexpr int kFirTB = 8 * kFirR; // output rows per wave
#ifndef ALZ_FIR_SK
#define ALZ_FIR_SK 8
#endif
static constexpr int kFirSK = ALZ_FIR_SK; // taps per block in k_fir_s (two row buffers: 8 fit, 16 spill)

struct FArgs {
  const double *x;
  double *y;
  int64_t n, sxn, syn;     // time stride (1 for channel-major blocks)
  int64_t sxc, syc;        // channel stride (1 for time-major blocks)
  int64_t channels, n_inputs, n_sets;
  int64_t c_first, c_end;
  int mode, map_input;
  int nb;
  const double *b, *a;     // b[k * n_sets + set], a[set] (a0 only)
  const double *xh;        // xh[k * channels + c] = x[-1-k]
  int div;                 // some a0 != 1
  double zero;             // what an all-zero tap set yields (lazy_filters.py:227-231)
  // k_fir_ring: block y = q * run_group + j computes the runs of kRingR output rows
  //   q * run_span + run_first + j * run_first_mul + s * run_stride,  s = 0 .. run_count - 1
  // (those that start inside the block).  Interleaved mapping: run_group > gridDim.y (q = 0), run y + s * gridDim.y.
  // Chains: run_group = W, run (q + 1) * W * M - 1 - j - s * W -- W waves walk W * M consecutive runs TOWARDS THE PAST.
  int64_t run_group, run_span, run_first, run_first_mul, run_stride, run_count;
  // Pacing of the chains (launch_fir): a run posts (chain_epoch << 32 | wall clock at its start) in chain_flags[run *
  // gridDim.x + blockIdx.x]; run r starts once run r + 1 of its channel group has been going for 1 / W of this wave's
  // own start-to-start time (less the hand-over's latency).  A hint only: nothing is read through the flags, a stale
  // or missing one costs L2 hits, not results, and a wait gives up after chain_bound percent of the lead (shipped: 1.5
  // leads -- with another kernel's blocks on the chip the chains lose their gain and no more, profiles/NOTES_r05.md 9).
  // Times are ticks of wall_clock64(), the 100 MHz constant clock (s_memrealtime).
  unsigned long long *chain_flags;
  unsigned long long *chain_stats;   // -DALZ_TUNING builds: wait ticks, waits, waits given up, run ticks, runs
  unsigned chain_epoch;
  int chain_bound;         // a wait gives up after this many percent of the lead (0: after 0.4 ms)
  int chain_w;             // waves per chain
  int chain_lead0;         // the lead in 10 ns ticks before a wave has timed a run of its own
  int chain_share, chain_handover;   // lead = start-to-start time * chain_share / (100 W) - chain_handover ticks
};

template <bool SHARED>
__global__ __launch_bounds__(64) void k_fir(FArgs p) {
  const int lane = threadIdx.x;
  int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + lane;
  const bool live = c < p.c_end;
  if (!live) c = p.c_end - 1;  // clamp: keep the wave's loads in bounds, mask its stores
  int64_t in, set;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
  const double a0 = p.a[set];
  const int64_t tb0 = (int64_t)blockIdx.y * kFirTB;
  // shared taps are staged once in LDS, zero-padded to a multiple of the tap block: every later
  // tap read is one broadcast ds_read with no bounds test (a zero tap is skipped anyway)
  extern __shared__ __attribute__((aligned(16))) double tap_lds[];
  if constexpr (SHARED) {
    const int padded = ((p.nb + kFirK - 1) / kFirK) * kFirK;
    for (int k = lane; k < padded; k += 64) tap_lds[k] = (k < p.nb) ? p.b[k] : 0.0;
    __syncthreads();
  }
  bool all_zero = false;
  if constexpr (!SHARED) {   // a channel whose taps are all zero yields `zero`
    int nz = 0;
    for (int k = 0; k < p.nb; ++k) nz += p.b[(int64_t)k * p.n_sets + set] != 0.0;
    all_zero = nz == 0;
  }

  for (int sub = 0; sub < kFirTB; sub += kFirR) {
    const int64_t t0 = tb0 + sub;
    if (t0 >= p.n) break;
    double acc[kFirR];
#pragma unroll
    for (int r = 0; r < kFirR; ++r) acc[r] = -0.0;

    // x window: xw[j] = row t0 - kb - (K-1) + j, j = 0 .. R+K-2.  Going to the next tap block
    // moves the window K rows into the past: R-1 rows are kept (register moves), K are loaded.
    double xw[kFirR + kFirK - 1];
    auto load_row = [&](int64_t t) -> double {
      if (t > p.n - 1) t = p.n - 1;                         // past the block: value is never used
      int64_t hk = -t - 1;                                  // before the stream: history row
      if (hk > p.nb - 2) hk = p.nb - 2;                     // beyond the delay line: never used
      const double *src = (t >= 0) ? p.x + t * p.sxn + in : p.xh + (hk < 0 ? 0 : hk) * p.channels + c;
      return *src;
    };
#pragma unroll
    for (int j = kFirK; j < kFirR + kFirK - 1; ++j) xw[j] = load_row(t0 - (kFirK - 1) + j);
    for (int kb = 0; kb < p.nb; kb += kFirK) {
      if (kb > 0) {
#pragma unroll
        for (int j = kFirR + kFirK - 2; j >= kFirK; --j) xw[j] = xw[j - kFirK];
      }
      const int64_t tb = t0 - kb - (kFirK - 1);
      if (tb >= 0) {                                        // whole row group inside the block
        const double *r0 = p.x + tb * p.sxn + in;
#pragma unroll
        for (int j = 0; j < kFirK; ++j) xw[j] = r0[j * p.sxn];
      } else {                                              // reaches into the history rows
#pragma unroll
        for (int j = 0; j < kFirK; ++j) xw[j] = load_row(tb + j);
      }
      if constexpr (SHARED) {
        double bk[kFirK];                                   // wave-uniform taps: broadcast LDS reads
#pragma unroll
        for (int kk = 0; kk < kFirK; ++kk) bk[kk] = tap_lds[kb + kk];
#pragma unroll
        for (int kk = 0; kk < kFirK; ++kk) {
          if (bk[kk] == 0.0) continue;                      // absent from the reference's sum
#pragma unroll
          for (int r = 0; r < kFirR; ++r) acc[r] = acc[r] + bk[kk] * xw[r - kk + (kFirK - 1)];
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < kFirK; ++kk) {
          const int k = kb + kk;
          if (k >= p.nb) break;
          const double bk = p.b[(int64_t)k * p.n_sets + set];
          const bool nz = bk != 0.0;
#pragma unroll
          for (int r = 0; r < kFirR; ++r) {
            const double s = acc[r] + bk * xw[r - kk + (kFirK - 1)];
            acc[r] = nz ? s : acc[r];
          }
        }
      }
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < kFirR; ++r) {
        const int64_t t = t0 + r;
        if (t < p.n) p.y[t * p.syn + c] = all_zero ? p.zero : (p.div ? acc[r] / a0 : acc[r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// (k_fir_s, round 1, kept only as this note:) k_fir for taps shared by the whole bank, restructured around what bounded k_fir<true>
// (62 % of the separately-rounded f64 rate): the K new rows of a tap block were loaded at the top
// of the block and needed by its first tap, so each of the two waves of a SIMD sat out a whole
// L2 / HBM latency per block.  Here the rows of block kb + K are requested before the sums of
// block kb start (a second K-row register buffer), the taps are wave-uniform SGPR pairs
// (readfirstlane of the broadcast LDS read: the multiply takes them as its scalar operand and
// 2K VGPRs are free for the buffer), and the row addresses are one uniform base per block plus a
// per-lane 32-bit offset.  Same sums, same order.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_uniform(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)bits);
  const int hi = __builtin_amdgcn_readfirstlane((int)(bits >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// ---------------------------------------------------------------------------
// k_fir_ring: k_fir_s without its non-arithmetic vector instructions (PMC: 18 % of the VALU slots
// of k_fir_s were window moves, per-lane row pointers, tap broadcasts and zero tests, on ALUs that
// are 93 % busy).  The x window is a register ring of NG = R/K + 2 groups of K rows: a tap block
// uses NG - 1 of them and the last one receives the rows of the next block, so going to the next
// block renames groups instead of moving registers -- the block loop is unrolled NG times with the
// ring phase a template parameter.  Rows come through buffer loads (uniform descriptor per group,
// per-lane 32-bit channel offset, uniform row offset: no vector address arithmetic), taps through
// scalar loads from the constant address space (SGPR operands of the multiplies), and the
// "tap is +-0, hence absent from the sum" test is scalar too.
// ---------------------------------------------------------------------------
typedef const double __attribute__((address_space(4))) *const_taps_t;
// register tile of the ring kernel: 48 output rows per lane (R = 32: 53.3 / 77.9 Gsamples/s bit-exact /
// FMA on cfg3, R = 48: 55.2 / 82.5, R = 64 spills; profiles/r02_fir_tile_sweep.log)
#ifndef ALZ_FIR_RING_R
#define ALZ_FIR_RING_R 48
#endif
#ifndef ALZ_FIR_MG
#define ALZ_FIR_MG 4      // products formed per group in the bit-exact instantiation (4 or 2)
#endif
// taps per block (= rows per ring group) and the number of groups fetched AHEAD of the block that uses them:
// a block is K * R multiply-adds long, so the rows of block kb + PF * K have PF blocks of arithmetic to arrive
// in.  K = 8 / PF = 1 (rounds 1 - 2) left one 384-instruction block (~0.7 us per wave) -- less than a loaded
// HBM round trip; K = 4 / PF = 3 keeps the same 64 window rows in registers and fetches 576 instructions ahead.
#ifndef ALZ_FIR_RING_K
#define ALZ_FIR_RING_K 4
#endif
#ifndef ALZ_FIR_I32
#define ALZ_FIR_I32 1     // the ring groups' row clamps in 32-bit scalar arithmetic (0: round 5's 64-bit form, for A/B builds)
#endif
#ifndef ALZ_FIR_RING_PF
#define ALZ_FIR_RING_PF 3
#endif
static constexpr int kRingR = ALZ_FIR_RING_R;
static constexpr int kRingTB = 8 * kRingR;   // output rows per wave (blocked mapping)
static constexpr int kRingK = ALZ_FIR_RING_K;
static constexpr int kRingPF = ALZ_FIR_RING_PF;
static constexpr int kRingG = kRingR / kRingK + 1 + kRingPF;   // window groups + groups in flight
static_assert(kRingR % kRingK == 0 && kRingK <= 8 && kRingPF >= 1, "ring groups");

__device__ __forceinline__ bool tap_absent(double t) {
  long long sh;
  asm("s_lshl_b64 %0, %1, 1" : "=s"(sh) : "s"(__double_as_longlong(t)) : "scc");
  return sh == 0;
}

// Round 6, second half: the four taps of a block as ONE s_load_dwordx8 requested a whole block ahead, and the in-run row groups
// without clamps.  hipcc had sunk `ring_load_taps(kb + K)` to the head of the block that uses it -- s_load + s_waitcnt lgkmcnt(0)
// back to back, a scalar-cache round trip per 192 multiply-adds with the wave standing still (two waves per SIMD: the other one alone
// fills ~80 % of it) -- and every group load of an interior run recomputed a 64-bit row product and three clamps that can never
// bind there (~30 scalar instructions per block, in order, in the same wave that feeds the VALU).  0: the round-6 first-half form.
#ifndef ALZ_FIR_PFTAPS
#define ALZ_FIR_PFTAPS 1
#endif
typedef double fir_dbl4 __attribute__((ext_vector_type(4)));
#define ALZ_FIR_SLOAD(d, ptr) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(d) : "s"(ptr) : "memory")
#define ALZ_FIR_SWAIT(d) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d) : : "memory")
#define ALZ_FIR_SWAIT2(d0, d1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1) : : "memory")

struct RingCtx {
  const FArgs *p;
  int64_t in, c, row_bytes;
  unsigned lane_off, chan_off;   // byte offsets of this lane's input column / output channel
  const_taps_t taps;
};

// rows tb .. tb + K - 1 (all <= n - 1) of this wave's channels into one ring group.  The buffer
// loads are issued whatever tb is (from row 0 when the group reaches before the block) and the rare
// history case overwrites them afterwards: with the loads behind a branch the compiler's waitcnt
// pass has to assume at the join that none were issued and waits for ALL outstanding loads before
// the first use of the PREVIOUS group -- which is exactly the latency this prefetch is there to hide.
template <bool EDGE>
__device__ __forceinline__ void ring_load_group(const RingCtx &q, int64_t tb, double (&dst)[kRingK]) {
  if constexpr (!EDGE) {
#if ALZ_FIR_I32
    // Row arithmetic in 32 bits (round 6; launch_fir keeps blocks below 2^31 rows): gfx9 has no 64-bit scalar compare, so the
    // clamps below were v_cmp_*_i64 + v_mov_b64 -- most of the 11 % of this kernel's VALU instructions that are not multiply-adds
    // (profiles/r02_pmc_fir_fma.txt) -- on the unit the multiply-adds need.  Same box, interleaved runs: 56.5 - 57.1 -> 58.4 - 58.7
    // Gsamples/s bit-exact, 90.3 - 90.6 -> 95.0 - 95.3 in the FMA mode (profiles/r06_fir_i32_ab.log).
    const int n32 = (int)q.p->n, rb32 = (int)q.row_bytes;
    int tc = (int)tb;
    tc = tc < 0 ? 0 : tc;
    tc = tc > n32 - 1 ? n32 - 1 : tc;                        // a group wholly past the block (window of the last run): never used
    const char *base = (const char *)q.p->x + (int64_t)tc * q.row_bytes;   // wave-uniform
    const int rows_left = n32 - tc, max_rows = 0x7fffffff / rb32;
    const int valid = (rows_left < max_rows ? rows_left : max_rows) * rb32;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, valid, 0x00020000);
    const int last = (n32 - 1 - tc) < (kRingK - 1) ? (n32 - 1 - tc) : (kRingK - 1);
#pragma unroll
    for (int j = 0; j < kRingK; ++j)
      dst[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, q.lane_off, (j < last ? j : last) * rb32, 0));
#else
    int64_t tc = tb < 0 ? 0 : tb;
    if (tc > q.p->n - 1) tc = q.p->n - 1;                    // a group wholly past the block (window of the last run): never used
    const char *base = (const char *)q.p->x + tc * q.row_bytes;            // wave-uniform
    int64_t valid = (q.p->n - tc) * q.row_bytes;
    if (valid > 0x7fffffff) valid = 0x7fffffff;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)valid, 0x00020000);
    // rows past the block are never used; they are clamped to its last row with scalar arithmetic (the
    // descriptor's range check does not cover the scalar offset)
    const int last = (int)((q.p->n - 1 - tc) < (kRingK - 1) ? (q.p->n - 1 - tc) : (kRingK - 1));
#pragma unroll
    for (int j = 0; j < kRingK; ++j)
      dst[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, q.lane_off, (int)((j < last ? j : last) * q.row_bytes), 0));
#endif
  } else {
    // first row tiles of a block: a row is either inside the block (t >= 0) or a row of the delay line
    // (p.xh[(-t - 1) * channels + c]).  Which one is wave-uniform, so it is a scalar choice of descriptor and
    // row offset for the same buffer load -- no per-lane 64-bit pointers, no branch for the wait-count pass to
    // lose track of (launch_fir keeps both byte ranges below 2^31).
    const FArgs &p = *q.p;
    int64_t xbytes = p.n * q.row_bytes;
    if (xbytes > 0x7fffffff) xbytes = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.xh, 0, (int)((int64_t)(p.nb - 1) * p.channels * 8), 0x00020000);
#pragma unroll
    for (int j = 0; j < kRingK; ++j) {
      int64_t t = tb + j;
      if (t > p.n - 1) t = p.n - 1;                           // past the block: never used
      int64_t hk = -t - 1;                                    // before the stream: history row
      if (hk > p.nb - 2) hk = p.nb - 2;                       // beyond the delay line: never used
      const bool in_block = t >= 0;
      const int soff = in_block ? (int)(t * q.row_bytes) : (int)(hk * p.channels * 8);
      const unsigned voff = in_block ? q.lane_off : q.chan_off;
      dst[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(in_block ? rx : rh, voff, soff, 0));
    }
  }
}

#if ALZ_FIR_PFTAPS && ALZ_FIR_RING_K != 4
#undef ALZ_FIR_PFTAPS
#define ALZ_FIR_PFTAPS 0      // (tile variants with another tap-block length: the round-6 first-half form)
#endif
#if ALZ_FIR_PFTAPS
static_assert(kRingK == 4, "the prefetched tap block is one s_load_dwordx8");
// one tap block (taps kb .. kb + K - 1) at ring phase PH: window row j lives in group
// (j / K - PH) mod NG; group (-PF - PH) mod NG is free and takes the first rows of block kb + PF * K.
// `tapv` was requested while the block before this one ran (ring_run: before the loop) and is waited for here;
// `tapv_next` is requested here.  `gb` (interior runs): the address of the first row of the group this block requests --
// it moves K rows towards the past per block and never leaves the input block (rows >= 0: `reach` in k_fir_ring;
// rows <= t0 - PF K < n), so the descriptor is the bare address and the four row offsets are loop invariants.
template <int PH, bool FMA, bool EDGE>
__device__ __forceinline__ void ring_step(const RingCtx &q, int64_t t0, int kb, double (&xr)[kRingG][kRingK],
                                          double (&acc)[kRingR], fir_dbl4 &tapv, fir_dbl4 &tapv_next, const char *&gb) {
  constexpr int R = kRingR, K = kRingK, NG = kRingG;
  constexpr int PF = kRingPF;
  if constexpr (EDGE) {
    ring_load_group<EDGE>(q, t0 - (kb + PF * K) - (K - 1), xr[(4 * NG - PF - PH) % NG]);   // unused after the last blocks
  } else {
    const int rb32 = (int)q.row_bytes;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gb, 0, 0x7fffffff, 0x00020000);
    double (&dst)[K] = xr[(4 * NG - PF - PH) % NG];
#pragma unroll
    for (int j = 0; j < K; ++j)
      dst[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, q.lane_off, j * rb32, 0));
    gb -= (int64_t)K * q.row_bytes;
  }
  ALZ_FIR_SWAIT(tapv);                                            // this block's taps (requested a block ago)
  ALZ_FIR_SLOAD(tapv_next, q.taps + kb + K);                      // the next block's (all 0.0 past the end: padded array + the test below)
  double tap[K];
  {
    const int nb = q.p->nb;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) tap[kk] = (kb + kk < nb) ? tapv[kk] : 0.0;
  }
#define ALZ_RING_X(j) xr[(((j) / K) + NG - PH) % NG][(j) % K]
  auto one_tap = [&](auto KK) {
    constexpr int kk = decltype(KK)::value;
#pragma unroll
    for (int r = 0; r < R; r += 4) {
      if constexpr (FMA) {
        // opt-in throughput mode (alz_bank_set_fused): one v_fma_f64 per tap and output, same ascending
        // tap order, one rounding per term instead of two -- not the reference's doubles
        acc[r] = __builtin_fma(tap[kk], ALZ_RING_X(r - kk + (K - 1)), acc[r]);
        acc[r + 1] = __builtin_fma(tap[kk], ALZ_RING_X(r + 1 - kk + (K - 1)), acc[r + 1]);
        acc[r + 2] = __builtin_fma(tap[kk], ALZ_RING_X(r + 2 - kk + (K - 1)), acc[r + 2]);
        acc[r + 3] = __builtin_fma(tap[kk], ALZ_RING_X(r + 3 - kk + (K - 1)), acc[r + 3]);
      } else {
#if ALZ_FIR_MG == 2
        {
          const double m0 = tap[kk] * ALZ_RING_X(r - kk + (K - 1));
          const double m1 = tap[kk] * ALZ_RING_X(r + 1 - kk + (K - 1));
          acc[r] = acc[r] + m0;
          acc[r + 1] = acc[r + 1] + m1;
        }
        {
          const double m2 = tap[kk] * ALZ_RING_X(r + 2 - kk + (K - 1));
          const double m3 = tap[kk] * ALZ_RING_X(r + 3 - kk + (K - 1));
          acc[r + 2] = acc[r + 2] + m2;
          acc[r + 3] = acc[r + 3] + m3;
        }
#else
        const double m0 = tap[kk] * ALZ_RING_X(r - kk + (K - 1));
        const double m1 = tap[kk] * ALZ_RING_X(r + 1 - kk + (K - 1));
        const double m2 = tap[kk] * ALZ_RING_X(r + 2 - kk + (K - 1));
        const double m3 = tap[kk] * ALZ_RING_X(r + 3 - kk + (K - 1));
        acc[r] = acc[r] + m0;
        acc[r + 1] = acc[r + 1] + m1;
        acc[r + 2] = acc[r + 2] + m2;
        acc[r + 3] = acc[r + 3] + m3;
#endif
      }
    }
  };
  if (!tap_absent(tap[0])) one_tap(std::integral_constant<int, 0>{});
  if (!tap_absent(tap[1])) one_tap(std::integral_constant<int, 1>{});
  if (!tap_absent(tap[2])) one_tap(std::integral_constant<int, 2>{});
  if (!tap_absent(tap[3])) one_tap(std::integral_constant<int, 3>{});
#undef ALZ_RING_X
}

// NG is even: the two tap buffers swap roles with the parity of the phase
static_assert(kRingG % 2 == 0, "the tap buffers alternate with the phase");
template <int PH, bool FMA, bool EDGE>
__device__ __forceinline__ void ring_steps(const RingCtx &q, int64_t t0, int &kb, double (&xr)[kRingG][kRingK],
                                           double (&acc)[kRingR], fir_dbl4 &tap_a, fir_dbl4 &tap_b, const char *&gb,
                                           bool &done) {
  if constexpr (PH < kRingG) {
    if (!done) {
      if constexpr (PH % 2 == 0) ring_step<PH, FMA, EDGE>(q, t0, kb, xr, acc, tap_a, tap_b, gb);
      else ring_step<PH, FMA, EDGE>(q, t0, kb, xr, acc, tap_b, tap_a, gb);
      kb += kRingK;
      done = kb >= q.p->nb;
    }
    ring_steps<PH + 1, FMA, EDGE>(q, t0, kb, xr, acc, tap_a, tap_b, gb, done);
  }
}

#ifndef ALZ_FIR_WAVES
#define ALZ_FIR_WAVES 2
#endif
// One run of R output rows per lane.  EDGE = false: every row the run touches (t0 - nb - K .. t0 + R - 1) lies
// inside the block, so the window comes through buffer loads only -- no per-lane row pointers, no history
// selects in the tap loop (that code kept ~16 more VGPRs live across the loop: the bit-exact instantiation
// spilled 36 bytes per lane at R = 48).  EDGE = true: the first row tiles of a block, which reach into the
// delay line (p.xh), and nothing else.
template <bool FMA, bool EDGE>
__device__ __forceinline__ void ring_run(const FArgs &p, const RingCtx &q, int64_t t0, bool live, double a0) {
  constexpr int R = kRingR, K = kRingK, NG = kRingG;
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = -0.0;
  double xr[NG][K];
  fir_dbl4 tap_a, tap_b;
  ALZ_FIR_SLOAD(tap_a, q.taps);
  // logical group g holds rows t0 - (K - 1) + g K ..: g = 0 .. R / K are the window of block 0, g = -1 .. -(PF - 1)
  // the first rows of blocks 1 .. PF - 1 (already in flight); logical g lives in physical (g + NG) % NG at phase 0
#pragma unroll
  for (int g = R / K; g >= -(kRingPF - 1); --g)
    ring_load_group<EDGE>(q, t0 - (K - 1) + (int64_t)g * K, xr[(g + NG) % NG]);
#pragma unroll
  for (int j = 0; j < K; ++j) xr[(NG - kRingPF) % NG][j] = 0.0;   // (the slot the first step loads into)
  int kb = 0;
  bool done = false;
  const char *gb = (const char *)p.x + (t0 - (int64_t)kRingPF * K - (K - 1)) * q.row_bytes;   // (interior runs only)
  asm volatile("" : "=s"(tap_b));                              // (no request yet: whatever it holds is never used)
  while (!done) ring_steps<0, FMA, EDGE>(q, t0, kb, xr, acc, tap_a, tap_b, gb, done);
  ALZ_FIR_SWAIT2(tap_a, tap_b);                                // the request behind the last block lands before its registers go
  if (live) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t t = t0 + r;
      if (t < p.n) p.y[t * p.syn + q.c] = p.div ? acc[r] / a0 : acc[r];
    }
  }
}
#else
// one tap block (taps kb .. kb + K - 1) at ring phase PH: window row j lives in group
// (j / K - PH) mod NG; group (-PF - PH) mod NG is free and takes the first rows of block kb + PF * K
__device__ __forceinline__ void ring_load_taps(const RingCtx &q, int kb, double (&tap)[kRingK]) {
  const int nb = q.p->nb;
#pragma unroll
  for (int kk = 0; kk < kRingK; ++kk) {
    const double t = q.taps[kb + kk];                       // the taps array is padded (alz_api.hip)
    tap[kk] = (kb + kk < nb) ? t : 0.0;
  }
}

template <int PH, bool FMA, bool EDGE>
__device__ __forceinline__ void ring_step(const RingCtx &q, int64_t t0, int kb, double (&xr)[kRingG][kRingK],
                                          double (&acc)[kRingR], const double (&tap)[kRingK],
                                          double (&tap_next)[kRingK]) {
  constexpr int R = kRingR, K = kRingK, NG = kRingG;
  constexpr int PF = kRingPF;
  ring_load_group<EDGE>(q, t0 - (kb + PF * K) - (K - 1), xr[(4 * NG - PF - PH) % NG]);   // unused after the last blocks
  ring_load_taps(q, kb + K, tap_next);                            // likewise (all 0.0 past the end)
#define ALZ_RING_X(j) xr[(((j) / K) + NG - PH) % NG][(j) % K]
  auto one_tap = [&](auto KK) {
    constexpr int kk = decltype(KK)::value;
#pragma unroll
    for (int r = 0; r < R; r += 4) {
      if constexpr (FMA) {
        // opt-in throughput mode (alz_bank_set_fused): one v_fma_f64 per tap and output, same ascending
        // tap order, one rounding per term instead of two -- not the reference's doubles
        acc[r] = __builtin_fma(tap[kk], ALZ_RING_X(r - kk + (K - 1)), acc[r]);
        acc[r + 1] = __builtin_fma(tap[kk], ALZ_RING_X(r + 1 - kk + (K - 1)), acc[r + 1]);
        acc[r + 2] = __builtin_fma(tap[kk], ALZ_RING_X(r + 2 - kk + (K - 1)), acc[r + 2]);
        acc[r + 3] = __builtin_fma(tap[kk], ALZ_RING_X(r + 3 - kk + (K - 1)), acc[r + 3]);
      } else {
#if ALZ_FIR_MG == 2
        // products formed two at a time: 48 accumulators + 64 window rows leave room for two product
        // temporaries, four spilled 36 bytes per lane (rocprofv3 Scratch_Size, profiles/r02_kernel_dispatches.csv)
        {
          const double m0 = tap[kk] * ALZ_RING_X(r - kk + (K - 1));
          const double m1 = tap[kk] * ALZ_RING_X(r + 1 - kk + (K - 1));
          acc[r] = acc[r] + m0;
          acc[r + 1] = acc[r + 1] + m1;
        }
        {
          const double m2 = tap[kk] * ALZ_RING_X(r + 2 - kk + (K - 1));
          const double m3 = tap[kk] * ALZ_RING_X(r + 3 - kk + (K - 1));
          acc[r + 2] = acc[r + 2] + m2;
          acc[r + 3] = acc[r + 3] + m3;
        }
#else
        const double m0 = tap[kk] * ALZ_RING_X(r - kk + (K - 1));
        const double m1 = tap[kk] * ALZ_RING_X(r + 1 - kk + (K - 1));
        const double m2 = tap[kk] * ALZ_RING_X(r + 2 - kk + (K - 1));
        const double m3 = tap[kk] * ALZ_RING_X(r + 3 - kk + (K - 1));
        acc[r] = acc[r] + m0;
        acc[r + 1] = acc[r + 1] + m1;
        acc[r + 2] = acc[r + 2] + m2;
        acc[r + 3] = acc[r + 3] + m3;
#endif
      }
    }
  };
  // (one test per tap BLOCK with a second, test-free copy of the K * R multiply-adds was tried in round 6: the two copies cost the
  // register allocator 900 bytes of scratch per lane -- the per-tap tests stay)
  if (!tap_absent(tap[0])) one_tap(std::integral_constant<int, 0>{});
  if constexpr (K > 1) { if (!tap_absent(tap[1])) one_tap(std::integral_constant<int, 1>{}); }
  if constexpr (K > 2) { if (!tap_absent(tap[2])) one_tap(std::integral_constant<int, 2>{}); }
  if constexpr (K > 3) { if (!tap_absent(tap[3])) one_tap(std::integral_constant<int, 3>{}); }
  if constexpr (K > 4) { if (!tap_absent(tap[4])) one_tap(std::integral_constant<int, 4>{}); }
  if constexpr (K > 5) { if (!tap_absent(tap[5])) one_tap(std::integral_constant<int, 5>{}); }
  if constexpr (K > 6) { if (!tap_absent(tap[6])) one_tap(std::integral_constant<int, 6>{}); }
  if constexpr (K > 7) { if (!tap_absent(tap[7])) one_tap(std::integral_constant<int, 7>{}); }
#undef ALZ_RING_X
}

// NG is even: the two tap buffers swap roles with the parity of the phase
template <int PH, bool FMA, bool EDGE>
__device__ __forceinline__ void ring_steps(const RingCtx &q, int64_t t0, int &kb, double (&xr)[kRingG][kRingK],
                                           double (&acc)[kRingR], double (&tap_a)[kRingK], double (&tap_b)[kRingK],
                                           bool &done) {
  if constexpr (PH < kRingG) {
    if (!done) {
      if constexpr (PH % 2 == 0) ring_step<PH, FMA, EDGE>(q, t0, kb, xr, acc, tap_a, tap_b);
      else ring_step<PH, FMA, EDGE>(q, t0, kb, xr, acc, tap_b, tap_a);
      kb += kRingK;
      done = kb >= q.p->nb;
    }
    ring_steps<PH + 1, FMA, EDGE>(q, t0, kb, xr, acc, tap_a, tap_b, done);
  }
}

#ifndef ALZ_FIR_WAVES
#define ALZ_FIR_WAVES 2
#endif
// One run of R output rows per lane.  EDGE = false: every row the run touches (t0 - nb - K .. t0 + R - 1) lies
// inside the block, so the window comes through buffer loads only -- no per-lane row pointers, no history
// selects in the tap loop (that code kept ~16 more VGPRs live across the loop: the bit-exact instantiation
// spilled 36 bytes per lane at R = 48).  EDGE = true: the first row tiles of a block, which reach into the
// delay line (p.xh), and nothing else.
template <bool FMA, bool EDGE>
__device__ __forceinline__ void ring_run(const FArgs &p, const RingCtx &q, int64_t t0, bool live, double a0) {
  constexpr int R = kRingR, K = kRingK, NG = kRingG;
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = -0.0;
  double xr[NG][K];
  // logical group g holds rows t0 - (K - 1) + g K ..: g = 0 .. R / K are the window of block 0, g = -1 .. -(PF - 1)
  // the first rows of blocks 1 .. PF - 1 (already in flight); logical g lives in physical (g + NG) % NG at phase 0
#pragma unroll
  for (int g = R / K; g >= -(kRingPF - 1); --g)
    ring_load_group<EDGE>(q, t0 - (K - 1) + (int64_t)g * K, xr[(g + NG) % NG]);
#pragma unroll
  for (int j = 0; j < K; ++j) xr[(NG - kRingPF) % NG][j] = 0.0;   // (the slot the first step loads into)
  int kb = 0;
  bool done = false;
  double tap_a[K], tap_b[K];
  ring_load_taps(q, 0, tap_a);
  while (!done) {
    ring_steps<0, FMA, EDGE>(q, t0, kb, xr, acc, tap_a, tap_b, done);
    if constexpr (NG % 2 != 0) {                             // odd ring: the roles end up swapped
#pragma unroll
      for (int kk = 0; kk < K; ++kk) { const double t = tap_a[kk]; tap_a[kk] = tap_b[kk]; tap_b[kk] = t; }
    }
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t t = t0 + r;
      if (t < p.n) p.y[t * p.syn + q.c] = p.div ? acc[r] / a0 : acc[r];
    }
  }
}

#endif

template <bool FMA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ALZ_FIR_WAVES, ALZ_FIR_WAVES)))
void k_fir_ring(FArgs p) {
  constexpr int R = kRingR, K = kRingK;
  const int lane = threadIdx.x;
  RingCtx q;
  q.p = &p;
  q.c = p.c_first + (int64_t)blockIdx.x * 64 + lane;
  const bool live = q.c < p.c_end;
  if (!live) q.c = p.c_end - 1;
  q.in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? q.c % p.n_inputs : q.c;
  q.lane_off = (unsigned)q.in * 8u;                         // launch_fir keeps channels * 8 < 2^31
  q.chan_off = (unsigned)q.c * 8u;
  q.row_bytes = p.sxn * 8;
  q.taps = (const_taps_t)(uintptr_t)p.b;
  const double a0 = p.a[0];
  // the lowest row a run starting at t0 reads is t0 - (nb_padded - K + PF K) - (K - 1) (the prefetch issued by
  // the last block): runs that start past that are interior
  const int64_t reach = (int64_t)((p.nb + K - 1) / K) * K + (kRingPF + 1) * K;
  const int64_t yq = (int64_t)blockIdx.y / p.run_group, yj = (int64_t)blockIdx.y - yq * p.run_group;
  int64_t run = yq * p.run_span + p.run_first + yj * p.run_first_mul;
  unsigned lead = (unsigned)p.chain_lead0;
  uint64_t started = 0;
  for (int64_t s = 0; s < p.run_count; ++s, run += p.run_stride) {
    const int64_t t0 = run * R;
    if (t0 >= p.n) {
      if (p.run_stride > 0) break;
      continue;                                              // chains walk towards the past: their first runs may not exist
    }
    if (p.chain_flags != nullptr) {
      const int64_t dep = run + 1;
      if (dep % p.run_span != 0 && dep * R < p.n) {          // the run above, unless it is another chain's or past the block
        const unsigned long long *f = p.chain_flags + dep * gridDim.x + blockIdx.x;
        const uint64_t wait_from = wall_clock64();
        const unsigned patience = p.chain_bound != 0 ? lead * (unsigned)p.chain_bound / 100u : 40000u;
        for (;;) {
          const unsigned long long v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
          const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
          const uint64_t now = wall_clock64();
          if (hi == p.chain_epoch && (int)((unsigned)now - lo) >= (int)lead) break;
          if (now - wait_from >= patience) {
#ifdef ALZ_TUNING
            if (p.chain_stats != nullptr && threadIdx.x == 0) atomicAdd(p.chain_stats + 2, 1ull);
#endif
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
#ifdef ALZ_TUNING
        if (p.chain_stats != nullptr && threadIdx.x == 0) {
          atomicAdd(p.chain_stats + 0, (unsigned long long)(wall_clock64() - wait_from));
          atomicAdd(p.chain_stats + 1, 1ull);
        }
#endif
      }
      const uint64_t now = wall_clock64();
      if (threadIdx.x == 0)
        __hip_atomic_store(p.chain_flags + run * gridDim.x + blockIdx.x,
                           ((unsigned long long)p.chain_epoch << 32) | (unsigned)now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (started != 0) {
        // start to start: the run and whatever this wave waited -- what one turn of the chain takes
        const uint64_t turn = (now - started) * (unsigned)p.chain_share / (100u * (unsigned)p.chain_w);
        lead = turn > (unsigned)p.chain_handover ? (unsigned)turn - (unsigned)p.chain_handover : 0u;
#ifdef ALZ_TUNING
        if (p.chain_stats != nullptr && threadIdx.x == 0) { atomicAdd(p.chain_stats + 3, (unsigned long long)(now - started)); atomicAdd(p.chain_stats + 4, 1ull); }
#endif
      }
      started = now;
    }
    if (t0 >= reach) ring_run<FMA, false>(p, q, t0, live, a0);
    else ring_run<FMA, true>(p, q, t0, live, a0);
  }
}

// ---------------------------------------------------------------------------
// k_fir_cm: the same FIR on channel-major blocks ([C, N], one Stream per row).  A wave owns ONE
// channel and a run of 64 x 32 = 2048 outputs; lane l computes the 32 consecutive outputs
// t0 + 32 l .. t0 + 32 l + 31.  The channel's input window (2048 + nb - 1 samples) is staged in LDS
// by coalesced loads, padded by one double per 32 so that the lanes' 33-double stride is
// conflict-free; the taps of the channel (shared or per channel -- a wave has one channel, so they
// are wave-uniform either way) are staged too.  Per output the sum is the same ascending
// left-to-right sum as k_fir / the reference; results go back through LDS so the stores are
// coalesced rows again.
// ---------------------------------------------------------------------------
static constexpr int kCmOut = 64 * kFirR;          // outputs per wave

__device__ __forceinline__ int cm_pad(int j) { return j + (j >> 5); }

__global__ __launch_bounds__(64) void k_fir_cm(FArgs p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int64_t c = p.c_first + blockIdx.x;
  int64_t in, set;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
  const int64_t t0 = (int64_t)blockIdx.y * kCmOut;
  const int hist = p.nb - 1;
  const int padded_taps = ((p.nb + kFirK - 1) / kFirK) * kFirK;
  double *taps = lds;                               // [padded_taps]
  double *win = lds + padded_taps;                  // window: index j <-> time t0 - hist + j, padded
  double *outb = win + cm_pad(kCmOut + hist) + 1;   // [kCmOut] padded, for the coalesced write-back
  int nz = 0;
  for (int k = lane; k < padded_taps; k += 64) {
    const double bk = (k < p.nb) ? p.b[(int64_t)k * p.n_sets + set] : 0.0;
    taps[k] = bk;
    nz += bk != 0.0;
  }
  const bool all_zero = __ballot(nz != 0) == 0;     // wave-uniform: the channel has no taps at all
  const double *xc = p.x + in * p.sxc;
  for (int j = lane; j < kCmOut + hist; j += 64) {
    int64_t t = t0 - hist + j;
    double v = 0.0;
    if (t >= p.n) t = p.n - 1;                      // past the block: never used
    if (t >= 0) v = xc[t * p.sxn];
    else v = p.xh[(-t - 1) * p.channels + c];       // before the stream: history row (-t-1 <= hist-1)
    win[cm_pad(j)] = v;
  }
  __syncthreads();
  const double a0 = p.a[set];

  double acc[kFirR];
#pragma unroll
  for (int r = 0; r < kFirR; ++r) acc[r] = -0.0;
  // window registers: xw[j] = x[t0 + 32 lane + j - kb - (K-1)], j = 0 .. R+K-2
  double xw[kFirR + kFirK - 1];
  const int base = hist + kFirR * lane - (kFirK - 1);       // window index of xw[0] at kb = 0
#pragma unroll
  for (int j = kFirK; j < kFirR + kFirK - 1; ++j) xw[j] = win[cm_pad(base + j)];
  for (int kb = 0; kb < p.nb; kb += kFirK) {
    if (kb > 0) {
#pragma unroll
      for (int j = kFirR + kFirK - 2; j >= kFirK; --j) xw[j] = xw[j - kFirK];
    }
#pragma unroll
    for (int j = 0; j < kFirK; ++j) {
      const int idx = base - kb + j;                        // may run before the window for k >= nb
      xw[j] = win[cm_pad(idx < 0 ? 0 : idx)];
    }
    double bk[kFirK];
#pragma unroll
    for (int kk = 0; kk < kFirK; ++kk) bk[kk] = taps[kb + kk];
#pragma unroll
    for (int kk = 0; kk < kFirK; ++kk) {
      if (bk[kk] == 0.0) continue;                          // absent from the reference's sum
#pragma unroll
      for (int r = 0; r < kFirR; ++r) acc[r] = acc[r] + bk[kk] * xw[r - kk + (kFirK - 1)];
    }
  }
#pragma unroll
  for (int r = 0; r < kFirR; ++r) {
    const double y = all_zero ? p.zero : (p.div ? acc[r] / a0 : acc[r]);
    outb[cm_pad(kFirR * lane + r)] = y;
  }
  __syncthreads();
  double *yc = p.y + c * p.syc;
  for (int j = lane; j < kCmOut; j += 64) {
    const int64_t t = t0 + j;
    if (t < p.n) yc[t * p.syn] = outb[cm_pad(j)];
  }
}

// new input history after the block: xh_new[k] = x[n-1-k], or the old history when the
// block was shorter than the delay line
__global__ void k_fir_state(FArgs p, double *xh_new) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? c % p.n_inputs : c;
  for (int k = (int)blockIdx.y; k < p.nb - 1; k += (int)gridDim.y) {
    const int64_t t = p.n - 1 - k;
    xh_new[(int64_t)k * p.channels + c] = (t >= 0) ? p.x[t * p.sxn + in * p.sxc] : p.xh[(-t - 1) * p.channels + c];
  }
}

__global__ void k_copy_doubles(double *dst, const double *src, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = src[i];
}

static int cm_pad_host(int j) { return j + (j >> 5); }

// Feedback-free section with more taps than the register kernels take, time-major block,
// x and y distinct.  Returns ALZ_OK with *taken = false when the shape is not this kernel's.
int launch_fir(const SectionDev &sec, const BlockIO &io, hipStream_t stream, bool *taken,
               const char **kernel_name) {
  *taken = false;
  if (sec.na != 1 || sec.nb < 2) return ALZ_OK;
  // one stream (what the filter call protocol sends) is contiguous in time whichever layout it is
  // called: it goes to the time-parallel kernel, where 64 lanes share the channel instead of one
  const bool one = io.channels == 1 && io.n_inputs == 1 && io.sxn == 1 && io.syn == 1;
  const bool tm = !one && io.sxc == 1 && io.syc == 1;
  const bool cm = one || (!tm && io.sxn == 1 && io.syn == 1);
  if (!tm && !cm) return ALZ_OK;
  if (io.x == io.y) return ALZ_OK;
  if ((sec.present_b) == 0) return ALZ_OK;                 // all-zero filter: k_generic yields `zero`
  FArgs p;
  p.x = io.x; p.y = io.y; p.n = io.n; p.sxn = io.sxn; p.syn = io.syn;
  p.sxc = io.sxc; p.syc = io.syc;
  p.channels = io.channels; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.c_first = io.c_first; p.c_end = io.c_first + io.c_count;
  p.mode = io.mode; p.map_input = io.map_input;
  p.nb = sec.nb; p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.div = sec.any_div ? 1 : 0;
  p.zero = io.zero;
  p.run_group = (int64_t)1 << 40; p.run_span = 1; p.run_first = 0; p.run_first_mul = 0; p.run_stride = 1; p.run_count = 0;
  p.chain_flags = nullptr; p.chain_stats = nullptr; p.chain_epoch = 0; p.chain_bound = 0; p.chain_w = 1; p.chain_lead0 = 0;
  p.chain_share = 0; p.chain_handover = 0;
  const unsigned gx = (unsigned)((io.c_count + 63) / 64);
  const unsigned gy = (unsigned)((io.n + kFirTB - 1) / kFirTB);
  if (gy > 65535u) return ALZ_OK;  // block longer than the grid's y range: caller falls back
  const char *shared_name = "k_fir<shared>";
  const size_t tap_bytes = (size_t)((sec.nb + kFirK - 1) / kFirK) * kFirK * sizeof(double);   // kFirSK divides kFirK
  static_assert(kFirK % kFirSK == 0, "tap padding");
  if (sec.shared_sets && tap_bytes > 48 * 1024) return ALZ_OK;   // absurdly long: let k_generic have it
  if (cm) {
    const int hist = sec.nb - 1;
    const size_t lds = tap_bytes + (size_t)(cm_pad_host(kCmOut + hist) + 1 + cm_pad_host(kCmOut) + 1) * sizeof(double);
    if (lds > 60 * 1024) return ALZ_OK;
    const unsigned gyc = (unsigned)((io.n + kCmOut - 1) / kCmOut);
    if (gyc > 65535u || io.c_count > 0x7fffffff) return ALZ_OK;
    hipLaunchKernelGGL(k_fir_cm, dim3((unsigned)io.c_count, gyc), dim3(64), lds, stream, p);
  } else if (sec.shared_sets && io.n < ((int64_t)1 << 31) - 4096 && io.n_inputs * 8 < ((int64_t)1 << 31) && io.channels * 8 < ((int64_t)1 << 31) &&
             io.sxn * 8 * kRingK < ((int64_t)1 << 31) &&
             (int64_t)(sec.nb + (kRingPF + 3) * kRingK + 2 * kRingR) * io.sxn * 8 < ((int64_t)1 << 31) &&   // edge runs: 32-bit row offsets
             (int64_t)(sec.nb - 1) * io.channels * 8 < ((int64_t)1 << 31)) {
    // Run-to-wave mapping.  A run (kRingR output rows of 64 channels) reads a window of kRingR + nb - 1 input rows, so
    // neighbouring runs share most of their input -- and whether they find it in L2 is a matter of WHEN they read it.
    // FETCH_SIZE counts half the bytes of these loads (tools/ubench_fetch8.hip on a known byte count, as for wide loads);
    // per launch of configs[2], doubled, in units of the input block (17.2 GB; read once = 1.0):
    //   interleaved (rounds 3 - 4: run y + s GY, the resident waves of a channel group on ADJACENT runs)   5.8
    //     -- of the 6.6 the waves load: they start together, so at every moment they read rows kRingR apart; what one wave
    //     reads now the next one reads R taps later, by which time the XCD's 256 waves have pulled 6 MB through 4 MB of L2
    //   chains, free-running (W waves walk W M consecutive runs towards the past)                            3.9
    //   chains of 8, paced, waits bounded (what runs below)                                                  1.3 (FMA: 2.1)
    //   chains of 16, a run waits as long as it takes                                                        1.3, 2 % slower
    // and the kernel is 4 % (two roundings per term) / 6 % (FMA) faster for it: 53.4 / 81.3 Gsamples/s against 51.3 / 77.0.
    // Chains: tap k of run r + 1 and tap k - R of run r read the same row, so with run r + 1 AHEAD of run r by R taps'
    // time the ~(R + nb) / R runs that need a row read it together.  Towards the past because the window reaches that
    // way: a wave finishes run r and goes on to r - W, which is W R rows further down, where run r - W + 1 (its neighbour
    // in the chain) has just been.  The W waves of a chain are the waves of a channel group that are resident together
    // (they share blockIdx.x, hence the XCD and its L2); spread evenly over one start-to-start time their leads are
    // 1 / W of it each -- R taps' time when W = nb / R, a few taps off otherwise, which the L2 absorbs.  Pacing is by
    // start stamps in global memory (FArgs): hints, nothing is read through them.
    const int64_t runs_total = (io.n + kRingR - 1) / kRingR;
    int map_sel = ALZ_TUNE("ALZ_FIR_MAP", -1);
    const int cus = device_cus() > 0 ? device_cus() : 256;               // (per device: round-5 advisor)
    const int64_t per_group = (int64_t)ALZ_FIR_WAVES * 4 * cus / gx;   // waves of one channel group the chip holds at a time
    // Chain width (profiles/NOTES_r05.md 9: widths 2 .. 16 on banks of 2048 .. 32768 channels, 96 .. 512 taps): 8 waves
    // when the chip holds two chains or more per group, 4 when it holds eight waves of it, and chains of 4 WITHOUT pacing
    // below that (one chain per group: its waves are all there is to wait for, and waiting did not pay).  2 never paid.
    int64_t W = per_group >= 16 ? 8 : 4;
    bool paced = per_group >= 8;
    const int fill_c = ALZ_TUNE("ALZ_FIR_CFILL", 16);
    const int64_t gy_c = ((int64_t)fill_c * 4 * cus + gx - 1) / gx;    // blocks per channel group: fill_c per SIMD
    int64_t Qc = gy_c / W > 0 ? gy_c / W : 1;
    int64_t Mc = (runs_total + Qc * W - 1) / (Qc * W);
    if (map_sel < 0) {
      // chains when the XCD follows blockIdx.x (gx a multiple of 8: a chain's waves share an L2), the chip holds a whole
      // number of waves of every group and whole chains of them (forced on 1536 ... 20 480 channels that do not divide the
      // chip, paced chains lost 2 - 7 %: a chain's waves are then not resident together), the taps reach at least two runs back, and every wave has 16 runs or more to walk (the
      // leads a chain starts with are idle time: at 8 runs per wave the interleaved mapping was 1 % ahead); that mapping otherwise
      const bool chains = gx % 8 == 0 && (int64_t)ALZ_FIR_WAVES * 4 * cus % gx == 0 && per_group >= 4 && per_group % W == 0 &&
                          sec.nb >= 2 * kRingR && Mc >= 16;
      map_sel = chains ? 4 : 1;
    }
    unsigned gyr;
    if (map_sel == 4) {
      W = ALZ_TUNE("ALZ_FIR_W", (int)W);
      paced = ALZ_TUNE("ALZ_FIR_PACED", paced ? 1 : 0) != 0;
      Qc = gy_c / W > 0 ? gy_c / W : 1;
      Mc = (runs_total + Qc * W - 1) / (Qc * W);
      Qc = (runs_total + Mc * W - 1) / (Mc * W);
      if (Qc * W > 65535) return ALZ_OK;
      gyr = (unsigned)(Qc * W);
      p.run_group = W; p.run_span = W * Mc; p.run_first = W * Mc - 1; p.run_first_mul = -1; p.run_stride = -W; p.run_count = Mc;
      FirChains *fc = sec.chains;
      const size_t need = (size_t)(Qc * W * Mc + 1) * gx + 8;
      if (paced && fc != nullptr && need > fc->len) {
        // stream-ordered (round-5 advisor): the old slab is released behind whatever launch still polls it, the new one
        // exists before this call's kernel -- no device-wide synchronisation inside a process call
        if (fc->flags) (void)hipFreeAsync(fc->flags, stream);
        fc->flags = nullptr; fc->len = 0;
        if (hipMallocAsync((void **)&fc->flags, need * sizeof(unsigned long long), stream) == hipSuccess &&
            hipMemsetAsync(fc->flags, 0, need * sizeof(unsigned long long), stream) == hipSuccess)
          fc->len = need;
        else (void)hipGetLastError();                         // no slab: the chains run free
      }
      if (paced && fc != nullptr && fc->len >= need) {
        p.chain_flags = fc->flags;
        p.chain_epoch = ++fc->epoch;
        p.chain_w = (int)W;
        // a wait gives up after 1.5 leads: 50 .. 200 % time alike with chains of 8 (200 % cost 7 % with chains of 4), the
        // longer ones fetch less; waiting "until it comes" costs 2 - 15 % (the SIMD's other wave alone does not fill it)
        p.chain_bound = ALZ_TUNE("ALZ_FIR_BOUND", 150);
        // lead = start-to-start time * share / (100 W) - hand-over: the hand-over (stamp stored, polled, seen: ~2.5 us)
        // is part of every lead, and a chain whose W leads add up to more than a turn makes its waves wait for it
        p.chain_share = ALZ_TUNE("ALZ_FIR_SHARE", 85);
        p.chain_handover = ALZ_TUNE("ALZ_FIR_HANDOVER", 250);
        // before a wave has timed itself: ~0.98 (two roundings per term) / 0.65 (FMA) ticks per tap and output row
        const int64_t turn0 = (int64_t)sec.nb * kRingR * (io.fused ? 65 : 98) / 100 * p.chain_share / (100 * W);
        p.chain_lead0 = (int)(turn0 > p.chain_handover ? turn0 - p.chain_handover : 0);
#ifdef ALZ_TUNING
        if (getenv("ALZ_FIR_WAITSTAT")) {
          p.chain_stats = fc->flags + fc->len - 8;
          (void)hipMemsetAsync(p.chain_stats, 0, 8 * sizeof(unsigned long long), stream);
        }
#endif
      }
    } else if (map_sel == 0) {
      gyr = (unsigned)((io.n + kRingTB - 1) / kRingTB);
      p.run_group = 1; p.run_span = kRingTB / kRingR; p.run_stride = 1; p.run_count = kRingTB / kRingR;
    } else {
      // interleaved: a grid 64 times what fills the chip (a grid of exactly that size left 7 - 10 % on the table: waves
      // that finish early leave their SIMD half empty, profiles/NOTES_r03.md 5), block y takes runs y, y + GY, y + 2 GY, ...
      static const int fill = ALZ_TUNE("ALZ_FIR_FILL", 64);
      int64_t gy_fill = ((int64_t)fill * 1024 + gx - 1) / gx;  // waves that fill 1024 SIMDs `fill` times
      if (gy_fill < 1) gy_fill = 1;
      if (gy_fill > 65535) gy_fill = 65535;                      // (the grid's y range)
      gyr = (unsigned)(runs_total < gy_fill ? runs_total : gy_fill);
      p.run_first_mul = 1; p.run_stride = gyr; p.run_count = (runs_total + gyr - 1) / gyr;
    }
    if (io.fused) hipLaunchKernelGGL(k_fir_ring<true>, dim3(gx, gyr), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(k_fir_ring<false>, dim3(gx, gyr), dim3(64), 0, stream, p);
    shared_name = map_sel == 4 ? (io.fused ? "k_fir_ring<fma,chains>" : "k_fir_ring<chains>") : (io.fused ? "k_fir_ring<fma>" : "k_fir_ring");
#ifdef ALZ_TUNING
    if (p.chain_stats != nullptr) {
      unsigned long long st[8];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(st, p.chain_stats, sizeof(st), hipMemcpyDeviceToHost);
      fprintf(stderr, "[fir chains] waits %llu (given up %llu), %.1f us each; runs %llu, %.1f us each; waiting %.2f %% of the run time\n",
              st[1], st[2], st[1] ? st[0] * 0.01 / st[1] : 0.0, st[4], st[4] ? st[3] * 0.01 / st[4] : 0.0,
              st[3] ? 100.0 * st[0] / st[3] : 0.0);
    }
#endif
  }
  else if (sec.shared_sets)
    hipLaunchKernelGGL(k_fir<true>, dim3(gx, gy), dim3(64), tap_bytes, stream, p);
  else
    hipLaunchKernelGGL(k_fir<false>, dim3(gx, gy), dim3(64), 0, stream, p);
  // histories: into the spare half of the state slab, then over the live half
  const int64_t nx = (int64_t)(sec.nb - 1) * io.channels;
  double *xh_new = sec.xh + nx;
  hipLaunchKernelGGL(k_fir_state, dim3(gx, 8), dim3(64), 0, stream, p, xh_new);
  // only the channels of this launch were refreshed in xh_new: copy just those back
  if (io.c_first == 0 && io.c_count == io.channels) {
    hipLaunchKernelGGL(k_copy_doubles, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream,
                       sec.xh, xh_new, nx);
  } else {
    for (int k = 0; k < sec.nb - 1; ++k)
      hipLaunchKernelGGL(k_copy_doubles, dim3((unsigned)((io.c_count + 255) / 256)), dim3(256), 0,
                         stream, sec.xh + (int64_t)k * io.channels + io.c_first,
                         xh_new + (int64_t)k * io.channels + io.c_first, io.c_count);
  }
  ALZ_HIP_CHECK(hipGetLastError());
  *taken = true;
  *kernel_name = cm ? "k_fir_cm" : sec.shared_sets ? shared_name : "k_fir<per-channel>";
  return ALZ_OK;
}

}  // namespace alz
