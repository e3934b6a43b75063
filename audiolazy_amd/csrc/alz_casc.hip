// alz_casc.hip -- fused cascades: up to four sections per pass, intermediate samples never
// leave registers (BASELINE config 4, the gammatone bank).
//
// Replaces CascadeFilter.__call__ (reference audiolazy/lazy_filters.py:988-990: four nested
// generators) for the gammatone cascades of lazy_auditory.py:158-218.  Every section is the same
// bit-exact DF-I step as everywhere else (separately rounded mul/add, reference term order,
// absent taps absent); stage s+1 consumes the output stream of stage s, so the fused result is
// identical to running the sections one after the other.
//
// Shape of the work: a filterbank is an OUTER bank -- channel = band * n_inputs + stream -- with
// hundreds of thousands of channels, so all 64 lanes of a wave are real channels (no ghosts) and
// the kernel is bound by f64 issue (28 ops per output sample for gammatone.slaney) and by the
// 8 B/sample of output it has to write; the input is tiny by comparison (every input sample is
// shared by all bands and comes from L2 / Infinity Cache).  Data movement is k_wave's: 8 KiB
// tiles (64 channels x 16 samples) arrive by global_load_lds DMA into a 4-slot LDS ring, results
// go back through the slot as 1 KiB stores.  One wave per workgroup, no barriers.
#include <stdlib.h>

#include "alz_common.h"
#include <type_traits>

namespace alz {

// 1: the channel-major broadcast form stores two tiles at a time (256-byte pieces per row instead of 128).  Measured
// against tile-by-tile stores on the same box, three alternating pairs of runs (profiles/r05_pairstore_ab.log): 396.0 /
// 393.5 / 394.0 against 396.1 / 394.1 / 395.2 Gsamples/s -- nothing; the distance to the time-major form (446 - 458) is
// not the piece size (the 64 rows of a wave's store lie 8 MiB apart in [bands, 2^20] doubles: one HBM channel).  Off.
#ifndef ALZ_CASC_PAIRSTORE
#define ALZ_CASC_PAIRSTORE 0
#endif
static constexpr int kCRing = 4;
static constexpr int kCChunks = 8;
static constexpr int kCSlot = 8192 + kCChunks * 16;

struct CArgs {
  const double *x;
  double *y;
  int64_t ldx, ldy;
  int64_t n_tiles;
  int64_t channels, n_inputs, n_sets;
  int64_t c_first;
  int mode;
  int map_input;   // OUTER bank: x is indexed by input (channel % n_inputs); 0: x already has one row / column per channel
  int nsec;
  int nb[4], na[4];
  const double *b[4], *a[4];
  double *xh[4], *yh[4];
  int dbg;  // ALZ_WAVE_DEBUG ablation bits (wrong output!): 1 no DMA, 2 no section arithmetic, 4 no stores
  // time-parallel mode (alz_scan.hip, channel-major blocks only): kchunks > 0 cuts every channel's block into
  // kchunks chunks of ldx == ldy == chunk-length samples and runs them as "virtual channels"
  // vc = real_channel * kchunks + chunk.  A 64-channel group is then 64 consecutive chunks of ONE real channel
  // (kchunks % 64 == 0): its rows start at real * ld?_outer + chunk * ld? and are ld? apart; coefficient set
  // and input index follow the real channel; state lives in per-virtual-channel arrays (channels = all vc).
  int64_t kchunks, ldx_outer, ldy_outer;
  int nostore;   // the zero-state pass: run for the end state only
  int stagger;   // channel-major blocks: workgroup g starts g * stagger ticks (10 ns) late (alz_common.h: stagger_start)
  // chunk-major virtual channels (k_casc only): vc = chunk * creal + real_channel, so a 64-channel group is 64 adjacent
  // real channels of ONE chunk (creal % 64 == 0).  Time-major blocks: whole 512-byte row pieces in x and y, chunk_len
  // rows further down per chunk.  Channel-major blocks with ONE input stream (BC): 64 output rows, chunk_len samples in.
  // ldx / ldy stay the block's own in both.
  int chunk_tm;
  int64_t creal, chunk_len;
};

// where a group of 64 (virtual) channels starting at c0 finds its rows, set and input
struct CGroup {
  int64_t xbase, ybase;   // element offsets of the group's first row in x / y
  int64_t creal0;         // real channel of the group's first lane
};
__device__ __forceinline__ CGroup c_group(const CArgs &p, int64_t c0) {
  CGroup g;
  const bool outer = p.mode == ALZ_BANK_OUTER;
  if (p.kchunks > 0 && p.chunk_tm) {
    // chunk-major virtual channels on a channel-major block: the group's 64 lanes are 64 adjacent real channels (rows
    // ldy apart) of chunk jc, chunk_len samples into every row; the ONE input row is read by wave-uniform loads (BC)
    const int64_t jc = c0 / p.creal, real = c0 - jc * p.creal;
    const int64_t in = (outer && p.map_input) ? real % p.n_inputs : real;
    g.xbase = in * p.ldx_outer + jc * p.chunk_len;
    g.ybase = real * p.ldy_outer + jc * p.chunk_len;
    g.creal0 = real;
  } else if (p.kchunks > 0) {
    const int64_t real = c0 / p.kchunks, j0 = c0 - real * p.kchunks;
    const int64_t in = (outer && p.map_input) ? real % p.n_inputs : real;
    g.xbase = in * p.ldx_outer + j0 * p.ldx;
    g.ybase = real * p.ldy_outer + j0 * p.ldy;
    g.creal0 = real;
  } else {
    const int64_t in0 = (outer && p.map_input) ? c0 % p.n_inputs : c0;
    g.xbase = in0 * p.ldx;      // (channel-major; the time-major callers use in0 itself)
    g.ybase = c0 * p.ldy;
    g.creal0 = c0;
  }
  return g;
}
// coefficient set of (virtual) channel c
__device__ __forceinline__ int64_t c_set(const CArgs &p, int64_t c) {
  const int64_t real = p.kchunks > 0 ? (p.chunk_tm ? c % p.creal : c / p.kchunks) : c;
  return p.mode == ALZ_BANK_OUTER ? real / p.n_inputs : ((p.n_sets == 1) ? 0 : real);
}

__device__ __forceinline__ void c_dma16(const void *gsrc, unsigned lds_dst) {
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
typedef double cdbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void c_store16(double *gdst, cdbl2 v) {
  // nt: the output is never read back by this launch, and a filterbank's input tiles -- re-read by
  // every band from L2 -- should not be pushed out by it (+6 % on cfg4)
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
}
__device__ __forceinline__ void c_wait_vm(int n) {   // n: a multiple of 4 (half-width tiles) or 8
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
}

// One section over a chunk of W samples held in registers: v[] in, v[] out (in place).
// NB taps (pattern PB, bit k <=> b_k present), PA bit k-1 <=> a_k present; dx[] = the NB-1
// previous inputs (dx[0] most recent), m1/m2 the previous outputs.
// FMA (opt-in, alz_bank_set_fused): every term after the first is one v_fma_f64 -- 4 instead of 7
// instructions per step of a gammatone.slaney section, same term order, NOT the reference's doubles.
template <int W, int NB, unsigned PB, unsigned PA, bool FMA = false>
__device__ __forceinline__ void section_chunk(double (&v)[W], const double (&bc)[8], double na1,
                                              double na2, double (&dx)[7], double &m1, double &m2) {
  double p[W];
  // feed-forward sums: independent of this section's recurrence, free to overlap with it
#pragma unroll
  for (int u = 0; u < W; ++u) {
    double acc = 0.0;
    bool first = true;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if ((PB >> k) & 1u) {
        const double xv = (u - k >= 0) ? v[u - k < 0 ? 0 : u - k] : dx[k - u - 1 < 0 ? 0 : (k - u - 1 > 6 ? 6 : k - u - 1)];
        if (FMA && !first) {
          acc = __builtin_fma(bc[k], xv, acc);
        } else {
          const double t = bc[k] * xv;
          acc = first ? t : acc + t;
        }
        first = false;
      }
    }
    p[u] = acc;
  }
  // new input history (before v is overwritten)
  double ndx[7];
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) ndx[k] = (W - 1 - k >= 0) ? v[W - 1 - k < 0 ? 0 : W - 1 - k] : dx[k - W < 0 ? 0 : k - W];
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) dx[k] = ndx[k];
#pragma unroll
  for (int u = 0; u < W; ++u) {
    double acc = p[u];
    if constexpr (PB != 0u) {
      if constexpr (PA & 1u) acc = FMA ? __builtin_fma(na1, m1, acc) : acc + na1 * m1;
      if constexpr (PA & 2u) acc = FMA ? __builtin_fma(na2, m2, acc) : acc + na2 * m2;
    } else {
      bool first = true;
      if constexpr (PA & 1u) { acc = na1 * m1; first = false; }
      if constexpr (PA & 2u) { const double t = na2 * m2; acc = first ? t : acc + t; }
    }
    v[u] = acc;
    m2 = m1;
    m1 = acc;
  }
}

// The same section over a 16-sample tile, handing every finished pair of outputs (2j, 2j + 1) to
// `emit` as soon as it exists, and pinning that order for the instruction scheduler: group j =
// {feed-forward sums of pair j + 1, recurrence of pair j, emit(j)}.  k_pipe's stage waves use it to
// spread the LDS writes of the hand-over through the arithmetic of the interval instead of queueing
// them all (four waves x 8 KiB at ~64 B/clk) behind the last recurrence step.  Same operations in the
// same order as section_chunk: identical doubles.
// `pre(j)` runs at the head of group j (k_pipe: the LDS read of piece j of the NEXT tile).
#ifndef ALZ_PIPE_ILV
#define ALZ_PIPE_ILV 1
#endif
#ifndef ALZ_PIPE_PHASE
#define ALZ_PIPE_PHASE 1
#endif
// where in a group stage s issues its hand-over's LDS operations: hex digit 3 - s of ALZ_PIPE_PHMAP (0x0123: stage s at slot s)
#ifndef ALZ_PIPE_PHMAP
#define ALZ_PIPE_PHMAP 0x0123
#endif
constexpr int c_pipe_slot(int stage) { return ALZ_PIPE_PHASE ? ((ALZ_PIPE_PHMAP >> (4 * (3 - stage))) & 0xf) : 0; }
constexpr int c_popcount(unsigned v) { int n = 0; for (int k = 0; k < 8; ++k) n += (v >> k) & 1u; return n; }
constexpr int c_kth_tap(unsigned pb, int n) {      // delay of the n-th present tap
  for (int k = 0; k < 8; ++k)
    if ((pb >> k) & 1u) { if (n == 0) return k; --n; }
  return 0;
}
// The feed-forward sum of ONE sample as a sequence of single operations (ff_ops of them), so that they can be dealt
// out one at a time between the operations of the recurrence:  op 0: acc = b_k0 x;  then per further tap
// t = b_k x; acc = acc + t  (FMA: acc = fma(b_k, x, acc)) -- the same operations in the same order as section_chunk.
template <unsigned PB, bool FMA>
struct FfSeq {
  static constexpr int terms = c_popcount(PB);
  static constexpr int ops = terms == 0 ? 0 : (FMA ? terms : 2 * terms - 1);
};
template <unsigned PB, bool FMA, int I>
__device__ __forceinline__ void ff_op(int u, const double (&v)[16], const double (&bc)[8], const double (&dx)[7],
                                      double &acc, double &tmp) {
  constexpr int term = FMA ? I : (I + 1) / 2;
  constexpr int k = c_kth_tap(PB, term);
  constexpr bool is_mul = FMA ? true : (I == 0 || (I & 1));
  if constexpr (is_mul) {
    const double xv = (u - k >= 0) ? v[u - k < 0 ? 0 : u - k] : dx[k - u - 1 < 0 ? 0 : (k - u - 1 > 6 ? 6 : k - u - 1)];
    if constexpr (I == 0) acc = bc[k] * xv;
    else if constexpr (FMA) acc = __builtin_fma(bc[k], xv, acc);
    else tmp = bc[k] * xv;
  } else {
    acc = acc + tmp;
  }
}

// The same section over a 16-sample tile, handing every finished pair of outputs (2j, 2j + 1) to
// `emit` as soon as it exists.  Group j = {recurrence of pair j with the feed-forward operations of pair j + 1
// dealt out BETWEEN its dependent operations, emit(j)}: a lone in-order wave issues an FP64 operation every ~4.5
// cycles but can use a result only ~7.5 - 9 cycles after its producer, so a recurrence whose three dependent
// operations per sample follow each other directly waits ~4 cycles at each of them while the six feed-forward
// operations of the pair, all issued before it (what the compiler's scheduler does with the plain loop: round 3,
// ~40 cycles per sample), fill nothing.  One independent operation after every dependent one and the pair costs its
// issue time, 14 x 4.5 cycles.  The order is pinned with sched_barrier between the single operations.
// k_pipe's stage waves use `emit` / `pre` to spread the LDS writes of the hand-over (and the reads of the NEXT tile,
// `pre(j)` at the head of group j) through the arithmetic of the interval.  Same operations in the same order per
// sample as section_chunk: identical doubles.
//
// PH (0..3, k_pipe passes the stage index): WHERE in a group the hand-over's two LDS operations are issued -- the
// write of the pair finished in the previous group and the read of piece j of the next tile.  The four stage waves of
// a workgroup leave every barrier together and run the same instruction mix, so with the LDS operations at the same
// place in every wave all four requests (4 x (13 + 4) cycles of the CU's one LDS pipe) arrive at once, the waves
// stand in its queue, and then the pipe idles while all four compute: the interval was the SUM of arithmetic and
// hand-over (profiles/NOTES_r02.md 12).  Issued a quarter of a group apart, the requests of the four waves interleave.
#if ALZ_PIPE_ILV
// The core: `hook(j)` is called once in group j, at the place PH selects; o[0 .. 2j - 1] are final by then (the caller
// owns o and decides what is written or read when).
template <int NB, unsigned PB, unsigned PA, bool FMA, int PH, typename Hook>
__device__ __forceinline__ void section_tile_hook(const double (&v)[16], const double (&bc)[8], double na1, double na2,
                                                  double (&dx)[7], double &m1, double &m2, double (&o)[16], Hook hook) {
  double p[16];
  using Seq = FfSeq<PB, FMA>;
  constexpr int NF = Seq::ops;                       // feed-forward operations per sample
  double tmpa = 0.0, tmpb = 0.0;
#define ALZ_PIN() __builtin_amdgcn_sched_barrier(0)
  // feed-forward operation number F of the PAIR (u, u + 1): sample u's and sample u + 1's operations alternate
  auto ffp = [&](auto FI, int u) {
    constexpr int F = decltype(FI)::value;
    if constexpr (F < 2 * NF) {
      if constexpr ((F & 1) == 0) ff_op<PB, FMA, F / 2>(u, v, bc, dx, p[u], tmpa);
      else ff_op<PB, FMA, F / 2>(u + 1, v, bc, dx, p[u + 1], tmpb);
      ALZ_PIN();
    }
  };
#define ALZ_FF(F, u) ffp(std::integral_constant<int, (F)>{}, (u))
  // pair 0's feed-forward sums, before the first group
  {
    ALZ_FF(0, 0); ALZ_FF(1, 0); ALZ_FF(2, 0); ALZ_FF(3, 0); ALZ_FF(4, 0); ALZ_FF(5, 0); ALZ_FF(6, 0); ALZ_FF(7, 0);
    ALZ_FF(8, 0); ALZ_FF(9, 0); ALZ_FF(10, 0); ALZ_FF(11, 0); ALZ_FF(12, 0); ALZ_FF(13, 0); ALZ_FF(14, 0); ALZ_FF(15, 0);
    ALZ_FF(16, 0); ALZ_FF(17, 0); ALZ_FF(18, 0); ALZ_FF(19, 0); ALZ_FF(20, 0); ALZ_FF(21, 0); ALZ_FF(22, 0); ALZ_FF(23, 0);
    ALZ_FF(24, 0); ALZ_FF(25, 0);
  }
  static_assert(2 * NF <= 26, "at most seven taps per section");
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // the hand-over's LDS operations of this group, at the place PH selects
    auto lds_ops = [&](int slot) {
      if (slot == c_pipe_slot(PH)) {
        hook(j);
        ALZ_PIN();
      }
    };
    lds_ops(0);
    const int un = 2 * j + 2;                        // the pair whose feed-forward sums ride in this group
    const bool more = j + 1 < 8;
    if constexpr (PB != 0u && PA == 3u && !FMA) {
      // sample 2j:  t1 = na1 m1 | F | a = p + t1 | t2 = na2 m2 | F | o = a + t2      (| = pinned order)
      double t1 = na1 * m1; ALZ_PIN();
      if (more) ALZ_FF(0, un);
      double a = p[2 * j] + t1; ALZ_PIN();
      double t2 = na2 * m2; ALZ_PIN();
      lds_ops(1);
      if (more) ALZ_FF(1, un);
      const double o0 = a + t2; ALZ_PIN();
      if (more) ALZ_FF(2, un);
      lds_ops(2);
      // sample 2j + 1
      t1 = na1 * o0; ALZ_PIN();
      if (more) ALZ_FF(3, un);
      a = p[2 * j + 1] + t1; ALZ_PIN();
      t2 = na2 * m1; ALZ_PIN();
      lds_ops(3);
      if (more) ALZ_FF(4, un);
      const double o1 = a + t2; ALZ_PIN();
      if (more) ALZ_FF(5, un);
      o[2 * j] = o0; o[2 * j + 1] = o1;
      m2 = o0; m1 = o1;
      if (more) {                                    // (sections with more than two taps: the rest of the pair's sums)
        ALZ_FF(6, un); ALZ_FF(7, un); ALZ_FF(8, un); ALZ_FF(9, un); ALZ_FF(10, un); ALZ_FF(11, un); ALZ_FF(12, un);
        ALZ_FF(13, un); ALZ_FF(14, un); ALZ_FF(15, un); ALZ_FF(16, un); ALZ_FF(17, un); ALZ_FF(18, un); ALZ_FF(19, un);
        ALZ_FF(20, un); ALZ_FF(21, un); ALZ_FF(22, un); ALZ_FF(23, un); ALZ_FF(24, un); ALZ_FF(25, un);
      }
    } else if constexpr (PB != 0u && PA == 3u && FMA) {
      // FMA mode: two dependent operations per sample, one of the next pair's sums after each
      double a = __builtin_fma(na1, m1, p[2 * j]); ALZ_PIN();
      lds_ops(1);
      if (more) ALZ_FF(0, un);
      const double o0 = __builtin_fma(na2, m2, a); ALZ_PIN();
      if (more) ALZ_FF(1, un);
      lds_ops(2);
      a = __builtin_fma(na1, o0, p[2 * j + 1]); ALZ_PIN();
      if (more) ALZ_FF(2, un);
      lds_ops(3);
      const double o1 = __builtin_fma(na2, m1, a); ALZ_PIN();
      if (more) ALZ_FF(3, un);
      o[2 * j] = o0; o[2 * j + 1] = o1;
      m2 = o0; m1 = o1;
      if (more) {
        ALZ_FF(4, un); ALZ_FF(5, un); ALZ_FF(6, un); ALZ_FF(7, un); ALZ_FF(8, un); ALZ_FF(9, un); ALZ_FF(10, un);
        ALZ_FF(11, un); ALZ_FF(12, un); ALZ_FF(13, un);
      }
    } else {
      // other shapes (one-pole sections, sections without a numerator): the recurrence as written, the next pair's
      // sums dealt between its two samples
#pragma unroll
      for (int u = 2 * j; u < 2 * j + 2; ++u) {
        double acc = p[u];
        if constexpr (PB != 0u) {
          if constexpr (PA & 1u) acc = FMA ? __builtin_fma(na1, m1, acc) : acc + na1 * m1;
          if constexpr (PA & 2u) acc = FMA ? __builtin_fma(na2, m2, acc) : acc + na2 * m2;
        } else {
          bool first = true;
          if constexpr (PA & 1u) { acc = na1 * m1; first = false; }
          if constexpr (PA & 2u) { const double t = na2 * m2; acc = first ? t : acc + t; }
        }
        o[u] = acc;
        m2 = m1;
        m1 = acc;
        ALZ_PIN();
        if (more && u == 2 * j) {
          lds_ops(1);
          ALZ_FF(0, un); ALZ_FF(1, un); ALZ_FF(2, un); ALZ_FF(3, un); ALZ_FF(4, un); ALZ_FF(5, un); ALZ_FF(6, un);
          lds_ops(2);
          ALZ_FF(7, un); ALZ_FF(8, un); ALZ_FF(9, un); ALZ_FF(10, un); ALZ_FF(11, un); ALZ_FF(12, un);
        } else if (more) {
          lds_ops(3);
          ALZ_FF(13, un); ALZ_FF(14, un); ALZ_FF(15, un); ALZ_FF(16, un); ALZ_FF(17, un); ALZ_FF(18, un); ALZ_FF(19, un);
          ALZ_FF(20, un); ALZ_FF(21, un); ALZ_FF(22, un); ALZ_FF(23, un); ALZ_FF(24, un); ALZ_FF(25, un);
        } else if (u == 2 * j) {
          lds_ops(1);
          lds_ops(2);
        } else {
          lds_ops(3);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef ALZ_FF
#undef ALZ_PIN
  double ndx[7];
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) ndx[k] = (15 - k >= 0) ? v[15 - k < 0 ? 0 : 15 - k] : dx[k - 16 < 0 ? 0 : k - 16];
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) dx[k] = ndx[k];
}
#endif

template <int NB, unsigned PB, unsigned PA, bool FMA, int PH = 0, typename Emit, typename Pre>
__device__ __forceinline__ void section_tile_emit(const double (&v)[16], const double (&bc)[8], double na1, double na2,
                                                  double (&dx)[7], double &m1, double &m2, Emit emit, Pre pre) {
#if ALZ_PIPE_ILV
  // (the FMA mode keeps the compiler's own order inside a group: four operations per sample with two dependent ones
  // leave it nothing to gain from the pinned order -- measured 511 - 516 pinned against 533 Gsamples/s, NOTES_r04.md 3)
  if constexpr (!FMA) {
    // group j: the write of the pair finished in group j - 1, the read of piece j of the next tile; the last pair after the loop
    double o[16];
    section_tile_hook<NB, PB, PA, FMA, PH>(v, bc, na1, na2, dx, m1, m2, o, [&](int j) {
      if (j > 0) emit(j - 1, o[2 * j - 2], o[2 * j - 1]);
      pre(j);
    });
    emit(7, o[14], o[15]);
    __builtin_amdgcn_sched_barrier(0);
    return;
  }
#endif
  double p[16], o[16];
  auto ff = [&](int u) {
    double acc = 0.0;
    bool first = true;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if ((PB >> k) & 1u) {
        const double xv = (u - k >= 0) ? v[u - k < 0 ? 0 : u - k] : dx[k - u - 1 < 0 ? 0 : (k - u - 1 > 6 ? 6 : k - u - 1)];
        if (FMA && !first) {
          acc = __builtin_fma(bc[k], xv, acc);
        } else {
          const double t = bc[k] * xv;
          acc = first ? t : acc + t;
        }
        first = false;
      }
    }
    p[u] = acc;
  };
  ff(0);
  ff(1);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pre(j);
    if (j + 1 < 8) {
      ff(2 * j + 2);
      ff(2 * j + 3);
    }
#pragma unroll
    for (int u = 2 * j; u < 2 * j + 2; ++u) {
      double acc = p[u];
      if constexpr (PB != 0u) {
        if constexpr (PA & 1u) acc = FMA ? __builtin_fma(na1, m1, acc) : acc + na1 * m1;
        if constexpr (PA & 2u) acc = FMA ? __builtin_fma(na2, m2, acc) : acc + na2 * m2;
      } else {
        bool first = true;
        if constexpr (PA & 1u) { acc = na1 * m1; first = false; }
        if constexpr (PA & 2u) { const double t = na2 * m2; acc = first ? t : acc + t; }
      }
      o[u] = acc;
      m2 = m1;
      m1 = acc;
    }
    emit(j, o[2 * j], o[2 * j + 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
  double ndx[7];
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) ndx[k] = (15 - k >= 0) ? v[15 - k < 0 ? 0 : 15 - k] : dx[k - 16 < 0 ? 0 : k - 16];
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) dx[k] = ndx[k];
}

// Two consecutive sections over a chunk in ONE loop: step u of section B only needs step u of
// section A, so step u+1 of A and step u of B are independent chains and the in-order wave can
// fill one recurrence's result latency with the other's ops (twice the ILP of running the two
// sections back to back).  Same operations in the same order per section: identical doubles.
template <int W, int NBA, unsigned PBA, unsigned PAA, int NBB, unsigned PBB, unsigned PAB, bool FMA = false>
__device__ __forceinline__ void section_pair_chunk(double (&v)[W], const double (&bA)[8], double na1A,
                                                   double na2A, double (&dxA)[7], double &m1A, double &m2A,
                                                   const double (&bB)[8], double na1B, double na2B,
                                                   double (&dxB)[7], double &m1B, double &m2B) {
  double pA[W];
#pragma unroll
  for (int u = 0; u < W; ++u) {
    double acc = 0.0;
    bool first = true;
#pragma unroll
    for (int k = 0; k < NBA; ++k) {
      if ((PBA >> k) & 1u) {
        const double xv = (u - k >= 0) ? v[u - k < 0 ? 0 : u - k] : dxA[k - u - 1 < 0 ? 0 : (k - u - 1 > 6 ? 6 : k - u - 1)];
        if (FMA && !first) {
          acc = __builtin_fma(bA[k], xv, acc);
        } else {
          const double t = bA[k] * xv;
          acc = first ? t : acc + t;
        }
        first = false;
      }
    }
    pA[u] = acc;
  }
  double ndx[7];
#pragma unroll
  for (int k = 0; k < NBA - 1; ++k) ndx[k] = (W - 1 - k >= 0) ? v[W - 1 - k < 0 ? 0 : W - 1 - k] : dxA[k - W < 0 ? 0 : k - W];
#pragma unroll
  for (int k = 0; k < NBA - 1; ++k) dxA[k] = ndx[k];
  double av[W];
#pragma unroll
  for (int u = 0; u < W; ++u) {
    // section A, step u
    double a = pA[u];
    if constexpr (PBA != 0u) {
      if constexpr (PAA & 1u) a = FMA ? __builtin_fma(na1A, m1A, a) : a + na1A * m1A;
      if constexpr (PAA & 2u) a = FMA ? __builtin_fma(na2A, m2A, a) : a + na2A * m2A;
    } else {
      bool first = true;
      if constexpr (PAA & 1u) { a = na1A * m1A; first = false; }
      if constexpr (PAA & 2u) { const double t = na2A * m2A; a = first ? t : a + t; }
    }
    av[u] = a;
    m2A = m1A;
    m1A = a;
    // section B, step u (its inputs are av[u], av[u-1], ... or its history)
    double b = 0.0;
    bool firstb = true;
#pragma unroll
    for (int k = 0; k < NBB; ++k) {
      if ((PBB >> k) & 1u) {
        const double xv = (u - k >= 0) ? av[u - k < 0 ? 0 : u - k] : dxB[k - u - 1 < 0 ? 0 : (k - u - 1 > 6 ? 6 : k - u - 1)];
        if (FMA && !firstb) {
          b = __builtin_fma(bB[k], xv, b);
        } else {
          const double t = bB[k] * xv;
          b = firstb ? t : b + t;
        }
        firstb = false;
      }
    }
    if constexpr (PBB != 0u) {
      if constexpr (PAB & 1u) b = FMA ? __builtin_fma(na1B, m1B, b) : b + na1B * m1B;
      if constexpr (PAB & 2u) b = FMA ? __builtin_fma(na2B, m2B, b) : b + na2B * m2B;
    } else {
      bool first = true;
      if constexpr (PAB & 1u) { b = na1B * m1B; first = false; }
      if constexpr (PAB & 2u) { const double t = na2B * m2B; b = first ? t : b + t; }
    }
    v[u] = b;
    m2B = m1B;
    m1B = b;
  }
#pragma unroll
  for (int k = 0; k < NBB - 1; ++k) ndx[k] = (W - 1 - k >= 0) ? av[W - 1 - k < 0 ? 0 : W - 1 - k] : dxB[k - W < 0 ? 0 : k - W];
#pragma unroll
  for (int k = 0; k < NBB - 1; ++k) dxB[k] = ndx[k];
}

constexpr int nb_of(unsigned pb) {
  int n = 1;
  for (int k = 0; k < 8; ++k)
    if ((pb >> k) & 1u) n = k + 1;
  return n;
}

// CM: channel-major blocks ([C, N]); else time-major.  Section s has pattern (PBs, PAs);
// PB == 0 && PA == 0 marks "no such section".
// BC (time-major only): an OUTER bank on ONE input stream -- the reference's own filterbank shape as vector-valued
// samples, [N] in and [N, bands] out -- whose 64 lanes all read the same input sample: no input tiles at all, the 16
// samples of a tile come by wave-uniform loads one tile ahead (the LDS ring then only stages the output).
template <bool CM, bool BC, unsigned PB0, unsigned PA0, unsigned PB1, unsigned PA1, unsigned PB2, unsigned PA2,
          unsigned PB3, unsigned PA3>
__global__ __launch_bounds__(64) void k_casc(CArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 64, T = 16;
  constexpr int NS = (PB3 | PA3) ? 4 : (PB2 | PA2) ? 3 : (PB1 | PA1) ? 2 : 1;
  constexpr unsigned PBS[4] = {PB0, PB1, PB2, PB3};
  constexpr unsigned PAS[4] = {PA0, PA1, PA2, PA3};
  const int lane = threadIdx.x;
  const int64_t c0 = p.c_first + (int64_t)blockIdx.x * G;
  const int64_t c = c0 + lane;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  // OUTER: channel = set * n_inputs + input; the 64 channels of a wave share one set
  const bool outer = p.mode == ALZ_BANK_OUTER;
  const int64_t set = c_set(p, c);
  const CGroup grp = c_group(p, c0);
  stagger_start((unsigned)p.stagger, blockIdx.x & 63u);

  int64_t x_off, y_off, x_chunk, y_chunk, x_tile, y_tile;
  const double *xb = p.x;                                // BC: the group's input stream at its chunk's first row
  if (!CM) {
    const int row = lane / 32, cp = lane % 32;           // 2 rows of 64 channels per 1 KiB chunk
    int64_t r0 = c0, trow0 = 0;                          // the group's first real channel, its chunk's first row
    if (p.kchunks > 0) {
      const int64_t jc = c0 / p.creal;
      r0 = c0 - jc * p.creal;
      trow0 = jc * p.chunk_len;
    }
    const int64_t in0 = (outer && p.map_input) ? r0 % p.n_inputs : r0;
    x_off = (trow0 + row) * p.ldx + in0 + 2 * cp;
    y_off = (trow0 + row) * p.ldy + r0 + 2 * cp;
    xb = p.x + trow0 * p.ldx + in0;
    x_chunk = 2 * p.ldx; y_chunk = 2 * p.ldy;
    x_tile = (int64_t)T * p.ldx; y_tile = (int64_t)T * p.ldy;
  } else {
    // 8 channels x 16 samples per chunk.  The 16-byte pieces of a channel row are XOR-swizzled
    // with the channel index on the GLOBAL side (the DMA lands linearly in LDS): lane (ch, k)
    // moves piece k ^ ch, so the 64 lanes that later read "their" channel hit distinct banks.
    const int ch = lane / 8, sp = (lane % 8) ^ (ch & 7);
    x_off = grp.xbase + ch * p.ldx + 2 * sp;
    y_off = grp.ybase + ch * p.ldy + 2 * sp;
    x_chunk = 8 * p.ldx; y_chunk = 8 * p.ldy;
    x_tile = T; y_tile = T;
    xb = p.x + grp.xbase;
  }
  const int64_t xstep = CM ? 1 : p.ldx;                  // BC: elements between consecutive samples of the input stream

  // coefficients and state of every section, in registers
  double bc[4][8], na1[4], na2[4], dx[4][7], m1[4], m2[4];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      bc[s][k] = ((PBS[s] >> k) & 1u) ? p.b[s][(int64_t)k * p.n_sets + set] : 0.0;
    na1[s] = (PAS[s] & 1u) ? -p.a[s][1 * p.n_sets + set] : 0.0;
    na2[s] = (PAS[s] & 2u) ? -p.a[s][2 * p.n_sets + set] : 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) dx[s][k] = (k < p.nb[s] - 1) ? p.xh[s][(int64_t)k * p.channels + c] : 0.0;
    m1[s] = (p.na[s] > 1) ? p.yh[s][0 * p.channels + c] : 0.0;
    m2[s] = (p.na[s] > 2) ? p.yh[s][1 * p.channels + c] : 0.0;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    asm volatile("" : "+v"(na1[s]), "+v"(na2[s]), "+v"(m1[s]), "+v"(m2[s]));
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(bc[s][k]));
#pragma unroll
    for (int k = 0; k < 7; ++k) asm volatile("" : "+v"(dx[s][k]));
  }

  const double *xg = p.x + x_off;
  double *yg = p.y + y_off;
  const int64_t nt = p.n_tiles;
  double xcur[16], xnext[16];
  if constexpr (BC) {
#pragma unroll
    for (int u = 0; u < 16; ++u) xcur[u] = xb[(int64_t)u * xstep];
  } else {
    for (int t = 0; t < kCRing - 1 && t < nt; ++t) {
#pragma unroll
      for (int j = 0; j < kCChunks; ++j) c_dma16(xg + t * x_tile + j * x_chunk, lds0 + t * kCSlot + j * 1040);
    }
  }
  // element (sample u, lane) of a slot.  TIME: (u*64 + lane)*8 + (u/2)*16.
  // CHAN: chunk lane/8 (1040 B each), row lane%8 (128 B), piece (u/2) ^ (lane%8), half u&1.
  const int lane_off = CM ? (lane / 8) * 1040 + (lane % 8) * 128 : lane * 8;
  int swz[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) swz[k] = CM ? ((k ^ lane) & 7) * 16 : 0;
#define ALZ_COFF(u) (CM ? swz[((u) >> 1) & 7] + ((u) & 1) * 8 : (u) * G * 8 + (((u) * G) >> 7) * 16)

  // (BC: no input tiles -- the ring only stages the output, two slots in turn; the smaller footprint lets a CU hold two
  // of these one-wave workgroups per SIMD when the launch has them)
  constexpr int kOutRing = BC ? 2 : kCRing;
  for (int64_t i = 0; i < nt; ++i) {
    const int slot = (int)(i % kOutRing);
    const int64_t tn = i + kCRing - 1;
    if constexpr (BC) {
      const int64_t t1 = i + 1 < nt ? i + 1 : i;           // (the last tile requests itself again)
#pragma unroll
      for (int u = 0; u < 16; ++u) xnext[u] = xb[(t1 * T + u) * xstep];
    } else {
      if (tn < nt) {
        const int sn = (int)(tn % kCRing);
#pragma unroll
        for (int j = 0; j < kCChunks; ++j) c_dma16(xg + tn * x_tile + j * x_chunk, lds0 + sn * kCSlot + j * 1040);
      }
      const int64_t loads_after = (nt - 1 - i < kCRing - 1) ? (nt - 1 - i) : (kCRing - 1);
      const int64_t stores_after = p.nostore ? 0 : (i < kCRing - 1) ? i : (kCRing - 1);
      c_wait_vm((int)(loads_after + stores_after) * kCChunks);
    }
    char *tile = smem + slot * kCSlot + lane_off;
#pragma unroll
    for (int h = 0; h < T / 8; ++h) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (BC) v[u] = xcur[h * 8 + u];
        else v[u] = *reinterpret_cast<const double *>(tile + ALZ_COFF(h * 8 + u));
      }
      section_chunk<8, nb_of(PB0), PB0, PA0>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0]);
      if constexpr (NS > 1) section_chunk<8, nb_of(PB1), PB1, PA1>(v, bc[1], na1[1], na2[1], dx[1], m1[1], m2[1]);
      if constexpr (NS > 2) section_chunk<8, nb_of(PB2), PB2, PA2>(v, bc[2], na1[2], na2[2], dx[2], m1[2], m2[2]);
      if constexpr (NS > 3) section_chunk<8, nb_of(PB3), PB3, PA3>(v, bc[3], na1[3], na2[3], dx[3], m1[3], m2[3]);
      if (!p.nostore) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<double *>(tile + ALZ_COFF(h * 8 + u)) = v[u];
      }
    }
    if (!p.nostore && CM && BC && ALZ_CASC_PAIRSTORE && ((i & 1) || i + 1 < nt)) {
      // Channel-major broadcast form: a channel's row takes 128 bytes per tile -- stored tile by tile the output leaves as
      // 128-byte pieces of 64 different rows, where the time-major form writes 512-byte pieces (458 against 393
      // Gsamples/s).  The two output slots hold an even tile and the odd one behind it: stored together, every channel
      // gets 256 contiguous bytes -- lane l of store J: channel 4 J + l / 16, piece l % 16 of its 32 samples (pieces 0 - 7
      // in the even tile's slot, 8 - 15 in the odd one's; the slots keep the compute phase's XOR swizzle).
      if (i & 1) {
        const int c4 = lane >> 4, p16 = lane & 15, hsel = p16 >> 3, piece = p16 & 7;
        cdbl2 w[2 * kCChunks];
#pragma unroll
        for (int J = 0; J < 2 * kCChunks; ++J) {
          const int ch64 = 4 * J + c4;
          w[J] = *reinterpret_cast<const cdbl2 *>(smem + hsel * kCSlot + (ch64 >> 3) * 1040 + (ch64 & 7) * 128 + ((piece ^ (ch64 & 7)) * 16));
        }
        double *yp = p.y + grp.ybase + (int64_t)c4 * p.ldy + (i - 1) * T + 2 * p16;
#pragma unroll
        for (int J = 0; J < 2 * kCChunks; ++J) c_store16(yp + (int64_t)(4 * J) * p.ldy, w[J]);
      }
    } else if (!p.nostore) {
      double *yt = yg + i * y_tile;
      const char *ts = smem + slot * kCSlot;
      cdbl2 w[kCChunks];
#pragma unroll
      for (int j = 0; j < kCChunks; ++j) w[j] = *reinterpret_cast<const cdbl2 *>(ts + j * 1040 + lane * 16);
#pragma unroll
      for (int j = 0; j < kCChunks; ++j) c_store16(yt + j * y_chunk, w[j]);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the slot is about to be refilled by DMA)
    }
    if constexpr (BC) {
#pragma unroll
      for (int u = 0; u < 16; ++u) xcur[u] = xnext[u];
    }
  }
#undef ALZ_COFF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k < p.nb[s] - 1 && !p.nostore) p.xh[s][(int64_t)k * p.channels + c] = dx[s][k];   // (zero-state pass: the prepared input history stays)
    if (p.na[s] > 1) p.yh[s][0 * p.channels + c] = m1[s];
    if (p.na[s] > 2) p.yh[s][1 * p.channels + c] = m2[s];
  }
}

// ---------------------------------------------------------------------------
// k_pipe: the same four-section cascade as a wave pipeline.  A workgroup is five waves on the
// four SIMDs of a CU: wave s (s = 0..3) is cascade section s, wave 4 (AUX) queues the tile DMA and
// stores finished tiles.  At barrier interval t section s works on tile t - s; tiles are handed
// from section to section through 2-slot LDS rings in a lane-private layout
// [8 pieces][64 lanes][16 B] (ds_read/write_b128, conflict-free).  Each wave then issues ~7 f64
// ops per sample instead of 28, so with 16384 channels (256 workgroups) the step is ~4x shorter.
// Section order, term order and rounding are unchanged: stage s+1 consumes exactly the doubles
// stage s produced.
// ---------------------------------------------------------------------------
static constexpr int kPXRing = 7;   // x ring: six 8 KiB tiles in flight per workgroup (a tile is only 16 steps long)
#ifndef ALZ_PIPE_OVERLAP
#define ALZ_PIPE_OVERLAP 1
#endif
// ALZ_PIPE_OVERLAP 1: a stage wave reads the tile it will work on in the NEXT interval while it does
// the arithmetic of the current one (two register sets that swap roles every interval), so the LDS
// read latency of the hand-over is no longer in series with the recurrence; a stage then lags its
// predecessor by two barrier intervals instead of one.  2: the results of the PREVIOUS tile are
// written out at the start of the interval as well (three register sets in rotation, lag 3), so
// neither direction of the hand-over waits in series with the arithmetic -- measured SLOWER (364 vs
// 394 Gsamples/s on cfg4: all the LDS traffic of the six waves then lands at the start of the
// interval), kept for A/B only.

// ALZ_PIPE_DIRECT (time-major blocks, G = 64, OVL = 1 only; bit 1: input, bit 2: output): the first
// stage wave reads its tile straight from global memory into registers (a row of 64 channels is 512
// contiguous bytes; three tiles in flight in four rotating register sets) and the last stage wave
// stores its results straight from registers, instead of both going through LDS and the two helper
// waves: 32 of the 80 KiB an interval moves through the CU's LDS pipe disappear.
// Measured on cfg4 (profiles/r02_pipe_direct.log, Gsamples/s, bit-exact / FMA mode): LDS path 393 - 407 /
// 437 - 453; direct input 424 - 442 / 499 - 522 in both layouts (channel-major: bit 4, a lane reads the 128
// contiguous bytes of its own row with eight 16-byte loads); direct output 387 (slower: the last stage's 16
// stores sit in its critical path), both 429.  Direct output in channel-major blocks (16-byte pieces of a lane's
// own row) ran at 82 - 85: partial-line writes; two workgroups per CU without the x ring and a per-stage skew
// after the barrier gained nothing either (profiles/NOTES_r02.md 12; none of the three is kept in the source).
// Shipped: 5 = direct input in both layouts.
#ifndef ALZ_PIPE_DIRECT
#define ALZ_PIPE_DIRECT 5
#endif
// ALZ_PIPE_EARLYW (one section per stage wave, OVL = 1).  1: a stage writes each 16-byte piece of its
// output tile as soon as the two samples exist (section_tile_emit) instead of all eight after the
// last step.  2: in the steady state the eight LDS reads of the stage's NEXT input tile are spread over
// the same eight groups as well (stage_pf), so the CU's LDS pipe sees an even stream of one read and
// one write per wave and group instead of two bursts per interval.
// Measured on cfg4, channel-major, on top of the direct input (profiles/r02_pipe_direct.log): 0: 424 - 438,
// 1: 436 - 451, 2: 474 Gsamples/s; time-major 425 - 442 throughout.
#ifndef ALZ_PIPE_EARLYW
#define ALZ_PIPE_EARLYW 2
#endif

// SPW = sections per stage wave (1: four stage waves, 2: two stage waves); NW = 4 / SPW.
// G = channels per workgroup.  64: every lane of a stage wave is a channel, one workgroup fills a CU's
// LDS.  32: half-width workgroups with 4 KiB tiles, so that TWO of them share a CU -- lanes 32..63 of
// a stage wave are ghosts that mirror lanes 0..31 (a partially masked wave issues f64 ops ~36 % slower,
// so EXEC stays full; ghosts read the same LDS words by broadcast and do not write).  A bank that is
// only 256 workgroups wide at G = 64 (cfg4: 256 bands x 64 streams) then has two workgroups per CU
// whose barrier intervals drift apart: one's section arithmetic runs while the other hands tiles over.
// timing ablations (-DALZ_ABLATE builds only; wrong results): bit 8 = no barriers, bit 16 = no LDS drain before them
#if defined(ALZ_ABLATE) && defined(ALZ_PIPE_TIMING)
// per-wave cycle accounting (variant builds only; a clock read also waits for the wave's outstanding LDS operations, so
// "work" includes the drain): work = barrier exit -> barrier entry, wait = inside the barrier; the wave that waits
// least is the one the interval waits for
#define PIPE_CLOCK_DECL long long pc_work = 0, pc_wait = 0, pc_last = __builtin_readcyclecounter(); long long pc_n = 0; \
    const long long pc_c0 = pc_last, pc_w0 = wall_clock64();
#define PIPE_BARRIER() do { const long long c0_ = __builtin_readcyclecounter(); if (!ALZ_DBG(p, 8)) __builtin_amdgcn_s_barrier(); \
    const long long c1_ = __builtin_readcyclecounter(); pc_work += c0_ - pc_last; pc_wait += c1_ - c0_; pc_last = c1_; ++pc_n; } while (0)
#define PIPE_CLOCK_REPORT() do { if ((blockIdx.x == 5 || blockIdx.x == 200) && lane == 0) printf("k_pipe block %d wave %d: %.1f cycles of work + %.1f in the barrier per interval (%lld intervals); shader clock %.0f MHz over the launch (cycle counter against the 100 MHz wall clock)\n", \
    (int)blockIdx.x, wave, (double)pc_work / (double)pc_n, (double)pc_wait / (double)pc_n, pc_n, \
    (double)(__builtin_readcyclecounter() - pc_c0) / ((double)(wall_clock64() - pc_w0) / 100.0)); } while (0)
#else
#define PIPE_CLOCK_DECL
#define PIPE_CLOCK_REPORT() do {} while (0)
#define PIPE_BARRIER() do { if (!ALZ_DBG(p, 8)) __builtin_amdgcn_s_barrier(); } while (0)
#endif
// (the builtin, not inline asm: the compiler's own wait-count pass then knows that the LDS reads of this
// interval have landed and does not guard next interval's arithmetic with waits of its own)
#ifndef ALZ_PIPE_DRAINB
#define ALZ_PIPE_DRAINB 1
#endif
#if ALZ_PIPE_DRAINB
#define PIPE_DRAIN() do { if (!ALZ_DBG(p, 16)) { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); } } while (0)
#else
#define PIPE_DRAIN() do { if (!ALZ_DBG(p, 16)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#endif
template <bool CM, int SPW, int G, unsigned PB0, unsigned PA0, unsigned PB1, unsigned PA1, unsigned PB2,
          unsigned PA2, unsigned PB3, unsigned PA3, bool FMA = false>
__global__ __launch_bounds__(64 * (4 / SPW + 2), G == 32 ? 4 : 1) void k_pipe(CArgs p) {   // (second figure: waves per SIMD)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PIPE_CLOCK_DECL
  constexpr int T = 16, NW = 4 / SPW;
  constexpr int NCHK = G / 8;                    // 1 KiB DMA / store chunks per tile
  constexpr int kSlot = G * 128 + NCHK * 16;     // tile + 16 bytes of pad per chunk
  constexpr int kPiece = G * 16;                 // bytes per piece row of the lane-private hand-off layout
  // half-width workgroups do not overlap the fetch of the next tile with the arithmetic of the
  // current one inside a wave (the second register set would push them past 128 VGPRs = 4 waves per
  // SIMD, and with two workgroups of six waves on a CU a SIMD may have to host four): the other
  // workgroup's waves are what runs meanwhile
  constexpr int OVL = G == 32 ? 0 : ALZ_PIPE_OVERLAP;
  constexpr int LAG = OVL + 1;
  // OVL == 3: de-phased stages, two barriers per interval.  Even stages do their section arithmetic
  // in the first half of an interval and their LDS hand-over (write the finished tile, fetch the next
  // one) in the second half; odd stages the other way round.  At any time two stage waves compute
  // while the other two (and the helpers) own the LDS pipe, instead of all six waves computing and
  // then all six queueing 80 KiB of LDS traffic behind one barrier.  Stage w then works on tile
  // t - kDeLag[w]; the storer writes out tile t - 5 in the second half.
  constexpr bool DEPHASE = OVL == 3 && SPW == 1;
  constexpr int kDeLag[4] = {0, 1, 3, 4};
  constexpr bool DIRECT_IN = (ALZ_PIPE_DIRECT & (CM ? 4 : 1)) && G == 64 && OVL == 1;
  constexpr bool DIRECT_OUT = (ALZ_PIPE_DIRECT & 2) && !CM && G == 64 && OVL == 1;
  constexpr unsigned PBS[4] = {PB0, PB1, PB2, PB3};
  constexpr unsigned PAS[4] = {PA0, PA1, PA2, PA3};
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cl = lane & (G - 1);                 // the channel this lane computes (ghosts: lane - 32)
  const bool real = lane < G;
  const int64_t c0 = p.c_first + (int64_t)blockIdx.x * G;
  const int64_t c = c0 + cl;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const bool outer = p.mode == ALZ_BANK_OUTER;
  const int64_t in0 = (outer && p.map_input) ? c0 % p.n_inputs : c0;
  const int64_t set = c_set(p, c);
  const CGroup grp = c_group(p, c0);
  const int64_t nt = p.n_tiles;
  stagger_start((unsigned)p.stagger, blockIdx.x & 63u);
  // stage w reads tile t - LAG w (and, overlapped, computes tile t - LAG w - 1) in interval t;
  // the storer writes out tile t - store_lag; every wave passes the same n_iv barriers
  constexpr int store_lag = DEPHASE ? 5 : LAG * NW;
  const int64_t n_iv = (nt + store_lag + 1 + 11) / 12 * 12;  // a multiple of the 2-, 3- and 4-interval unrolls
  char *xring = smem;
  char *qring = smem + kPXRing * kSlot;                  // NW-1 hand-off rings, 2 slots each
  char *yring = qring + (NW - 1) * 2 * kSlot;
  const int lane_off = CM ? (cl / 8) * 1040 + (cl % 8) * 128 : cl * 8;
  int swz[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) swz[k] = CM ? ((k ^ cl) & 7) * 16 : 0;
#define ALZ_COFF(u) (CM ? swz[((u) >> 1) & 7] + ((u) & 1) * 8 : (u) * G * 8 + (((u) * G) >> 7) * 16)

  if (wave >= NW) {
    // ---------------- helpers: wave NW queues the tile DMA, wave NW+1 stores finished tiles ----------------
    // (two waves, because loads and stores of one wave share one in-order vmcnt counter of 63)
    int64_t x_off, y_off, x_chunk, y_chunk, x_tile, y_tile;
    if (!CM) {
      constexpr int PPR = G / 2;                            // 16-byte pieces per row
      const int row = lane / PPR, cp = lane % PPR;
      x_off = (int64_t)row * p.ldx + in0 + 2 * cp;
      y_off = (int64_t)row * p.ldy + c0 + 2 * cp;
      x_chunk = (64 / PPR) * p.ldx; y_chunk = (64 / PPR) * p.ldy;
      x_tile = (int64_t)T * p.ldx; y_tile = (int64_t)T * p.ldy;
    } else {
      const int ch = lane / 8, sp = (lane % 8) ^ (ch & 7);
      x_off = grp.xbase + ch * p.ldx + 2 * sp;
      y_off = grp.ybase + ch * p.ldy + 2 * sp;
      x_chunk = 8 * p.ldx; y_chunk = 8 * p.ldy;
      x_tile = T; y_tile = T;
    }
    const double *xg = p.x + x_off;
    double *yg = p.y + y_off;
    constexpr int D = kPXRing - 1;                          // tiles queued ahead
    if (wave == NW) {
      auto queue_tile = [&](int64_t t) {
        const int s = (int)(t % kPXRing);
#pragma unroll
        for (int j = 0; j < NCHK; ++j) c_dma16(xg + t * x_tile + j * x_chunk, lds0 + s * kSlot + j * 1040);
      };
      const bool on = !ALZ_DBG(p, 1) && !DIRECT_IN;
      for (int t = 0; t < D && t < nt && on; ++t) queue_tile(t);
      {
        const int64_t after = ((nt < D ? nt : D) - 1);
        c_wait_vm(on ? (int)(after > 5 ? 5 : after) * NCHK : 0);   // tile 0 has landed
      }
      PIPE_BARRIER();
      for (int64_t t = 0; t < n_iv; ++t) {
        if (t + D < nt && on) queue_tile(t + D);
        if (t + 1 < nt) {
          const int64_t last = (t + D < nt - 1) ? t + D : nt - 1;
          c_wait_vm(on ? (int)(last - (t + 1)) * NCHK : 0);  // tile t+1 has landed
        }
        PIPE_BARRIER();
        if constexpr (DEPHASE) PIPE_BARRIER();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      PIPE_BARRIER();
      for (int64_t t = 0; t < n_iv; ++t) {
        if constexpr (DEPHASE) PIPE_BARRIER();   // (first half: stage 3 writes the tile read below)
        if (t >= store_lag && t - store_lag < nt && !ALZ_DBG(p, 4) && !DIRECT_OUT && !p.nostore) {
          const int64_t tt = t - store_lag;
          const char *ys = yring + (int)(tt % 2) * kSlot;
          double *yt = yg + tt * y_tile;
          cdbl2 w[NCHK];
#pragma unroll
          for (int j = 0; j < NCHK; ++j) w[j] = *reinterpret_cast<const cdbl2 *>(ys + j * 1040 + lane * 16);
#pragma unroll
          for (int j = 0; j < NCHK; ++j) c_store16(yt + j * y_chunk, w[j]);
        }
        PIPE_DRAIN();
        PIPE_BARRIER();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    // ------------------ stage `wave`: sections wave*SPW .. wave*SPW + SPW - 1 ------------------
    double bc[SPW][8], na1[SPW], na2[SPW], dx[SPW][7], m1[SPW], m2[SPW];
    int nbv[SPW], nav[SPW];
    const double *bsrc[SPW], *asrc[SPW];
    double *xhs[SPW], *yhs[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
      unsigned pbv = 0, pav = 0;
      nbv[j] = 1; nav[j] = 1;
      bsrc[j] = p.b[0]; asrc[j] = p.a[0]; xhs[j] = p.xh[0]; yhs[j] = p.yh[0];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (wave * SPW + j == s) {
          pbv = PBS[s]; pav = PAS[s]; nbv[j] = p.nb[s]; nav[j] = p.na[s];
          bsrc[j] = p.b[s]; asrc[j] = p.a[s]; xhs[j] = p.xh[s]; yhs[j] = p.yh[s];
        }
#pragma unroll
      for (int k = 0; k < 8; ++k) bc[j][k] = ((pbv >> k) & 1u) ? bsrc[j][(int64_t)k * p.n_sets + set] : 0.0;
      na1[j] = (pav & 1u) ? -asrc[j][1 * p.n_sets + set] : 0.0;
      na2[j] = (pav & 2u) ? -asrc[j][2 * p.n_sets + set] : 0.0;
#pragma unroll
      for (int k = 0; k < 7; ++k) dx[j][k] = (k < nbv[j] - 1) ? xhs[j][(int64_t)k * p.channels + c] : 0.0;
      m1[j] = (nav[j] > 1) ? yhs[j][0 * p.channels + c] : 0.0;
      m2[j] = (nav[j] > 2) ? yhs[j][1 * p.channels + c] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
      asm volatile("" : "+v"(na1[j]), "+v"(na2[j]), "+v"(m1[j]), "+v"(m2[j]));
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(bc[j][k]));
#pragma unroll
      for (int k = 0; k < 7; ++k) asm volatile("" : "+v"(dx[j][k]));
    }

    auto read_tile = [&](int64_t tile, double (&v)[16]) {
      // input: stage 0 reads the DMA layout, the others the lane-private hand-off layout
      if (wave == 0) {
        const char *src = xring + (int)(tile % kPXRing) * kSlot + lane_off;
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const double *>(src + ALZ_COFF(u));
      } else {
        const char *src = qring + ((wave - 1) * 2 + (int)(tile % 2)) * kSlot + cl * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const cdbl2 w = *reinterpret_cast<const cdbl2 *>(src + j * kPiece);
          v[2 * j] = w.x;
          v[2 * j + 1] = w.y;
        }
      }
    };
    auto do_sections = [&](double (&v)[16]) {
      if (!ALZ_DBG(p, 2)) {
        if constexpr (SPW == 2) {
          if (wave == 0)
            section_pair_chunk<16, nb_of(PB0), PB0, PA0, nb_of(PB1), PB1, PA1, FMA>(
                v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0], bc[1], na1[1], na2[1], dx[1], m1[1], m2[1]);
          else
            section_pair_chunk<16, nb_of(PB2), PB2, PA2, nb_of(PB3), PB3, PA3, FMA>(
                v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0], bc[1], na1[1], na2[1], dx[1], m1[1], m2[1]);
        } else {
#pragma unroll
          for (int j = 0; j < SPW; ++j) {
            const int s = wave * SPW + j;
            if (s == 0) section_chunk<16, nb_of(PB0), PB0, PA0, FMA>(v, bc[j], na1[j], na2[j], dx[j], m1[j], m2[j]);
            else if (s == 1) section_chunk<16, nb_of(PB1), PB1, PA1, FMA>(v, bc[j], na1[j], na2[j], dx[j], m1[j], m2[j]);
            else if (s == 2) section_chunk<16, nb_of(PB2), PB2, PA2, FMA>(v, bc[j], na1[j], na2[j], dx[j], m1[j], m2[j]);
            else section_chunk<16, nb_of(PB3), PB3, PA3, FMA>(v, bc[j], na1[j], na2[j], dx[j], m1[j], m2[j]);
          }
        }
      }
    };
    auto write_tile = [&](int64_t tile, const double (&v)[16]) {
      if (G < 64 && !real) return;                          // ghost lanes hold the same doubles: one copy is written
      if (DIRECT_OUT && wave == NW - 1) {
        double *dst = p.y + (tile * T) * p.ldy + c0 + lane;   // 16 rows of 512 contiguous bytes
#pragma unroll
        for (int u = 0; u < 16; ++u) __builtin_nontemporal_store(v[u], dst + u * p.ldy);
      } else if (wave == NW - 1) {
        char *dst = yring + (int)(tile % 2) * kSlot + lane_off;
#pragma unroll
        for (int u = 0; u < 16; ++u) *reinterpret_cast<double *>(dst + ALZ_COFF(u)) = v[u];
      } else {
        char *dst = qring + (wave * 2 + (int)(tile % 2)) * kSlot + cl * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          cdbl2 w;
          w.x = v[2 * j];
          w.y = v[2 * j + 1];
          *reinterpret_cast<cdbl2 *>(dst + j * kPiece) = w;
        }
      }
    };
    auto work_tile = [&](int64_t tile, double (&v)[16]) {
      if constexpr (ALZ_PIPE_EARLYW && SPW == 1 && G == 64 && OVL == 1) {
        if (ALZ_DBG(p, 2)) return;
        if (wave == NW - 1) {
          if (DIRECT_OUT) {
            double *dst = p.y + (tile * T) * p.ldy + c0 + lane;
            section_tile_emit<nb_of(PB3), PB3, PA3, FMA, 3>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0],
                [&](int j, double a, double b) {
                  if (ALZ_DBG(p, 32)) return;
                  __builtin_nontemporal_store(a, dst + (2 * j) * p.ldy);
                  __builtin_nontemporal_store(b, dst + (2 * j + 1) * p.ldy);
                }, [](int) {});
          } else {
            char *dst = yring + (int)(tile % 2) * kSlot + lane_off;
            section_tile_emit<nb_of(PB3), PB3, PA3, FMA, 3>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0],
                [&](int j, double a, double b) {
                  if (ALZ_DBG(p, 32)) return;
                  if constexpr (CM) {
                    cdbl2 w;
                    w.x = a;
                    w.y = b;
                    *reinterpret_cast<cdbl2 *>(dst + swz[j]) = w;
                  } else {
                    *reinterpret_cast<double *>(dst + ALZ_COFF(2 * j)) = a;
                    *reinterpret_cast<double *>(dst + ALZ_COFF(2 * j + 1)) = b;
                  }
                }, [](int) {});
          }
        } else {
          char *dst = qring + (wave * 2 + (int)(tile % 2)) * kSlot + cl * 16;
          auto emit = [&](int j, double a, double b) {
            if (ALZ_DBG(p, 32)) return;
            cdbl2 w;
            w.x = a;
            w.y = b;
            *reinterpret_cast<cdbl2 *>(dst + j * kPiece) = w;
          };
          if (wave == 0) section_tile_emit<nb_of(PB0), PB0, PA0, FMA, 0>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0], emit, [](int) {});
          else if (wave == 1) section_tile_emit<nb_of(PB1), PB1, PA1, FMA, 1>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0], emit, [](int) {});
          else section_tile_emit<nb_of(PB2), PB2, PA2, FMA, 2>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0], emit, [](int) {});
        }
      } else {
        do_sections(v);
        write_tile(tile, v);
      }
    };
    // ALZ_PIPE_EARLYW == 2, stages 1..3 in their steady state: work on `tile` (in v) while the eight LDS
    // reads of the stage's next input tile ride in the same groups as the eight writes of this one, so
    // that the CU's LDS pipe sees an even stream instead of bursts.  The stage index is a compile-time
    // constant here: straight-line code per stage, no flow merges (which made the compiler guard the
    // arithmetic with LDS waits of its own).
    auto stage_pf = [&](auto SI, int64_t tile, double (&v)[16], int64_t pf_tile, double (&nxt)[16]) {
      constexpr int S = decltype(SI)::value;
      constexpr unsigned pbS = S == 1 ? PB1 : S == 2 ? PB2 : PB3;
      constexpr unsigned paS = S == 1 ? PA1 : S == 2 ? PA2 : PA3;
      const char *src = qring + ((S - 1) * 2 + (int)(pf_tile & 1)) * kSlot + cl * 16;
      auto pre = [&](int j) {
        if (ALZ_DBG(p, 64)) return;                       // (timing ablation: the tile's values are whatever is there)
        const cdbl2 w = *reinterpret_cast<const cdbl2 *>(src + j * kPiece);
        nxt[2 * j] = w.x;
        nxt[2 * j + 1] = w.y;
      };
      if constexpr (S == NW - 1 && DIRECT_OUT) {
        double *dst = p.y + (tile * T) * p.ldy + c0 + lane;
        section_tile_emit<nb_of(pbS), pbS, paS, FMA, S>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0],
            [&](int j, double a, double b) {
              if (ALZ_DBG(p, 32)) return;
              __builtin_nontemporal_store(a, dst + (2 * j) * p.ldy);
              __builtin_nontemporal_store(b, dst + (2 * j + 1) * p.ldy);
            }, pre);
      } else if constexpr (S == NW - 1) {
        char *dst = yring + (int)(tile & 1) * kSlot + lane_off;
        section_tile_emit<nb_of(pbS), pbS, paS, FMA, S>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0],
            [&](int j, double a, double b) {
              if (ALZ_DBG(p, 32)) return;
              if constexpr (CM) {
                cdbl2 w;
                w.x = a;
                w.y = b;
                *reinterpret_cast<cdbl2 *>(dst + swz[j]) = w;
              } else {
                *reinterpret_cast<double *>(dst + ALZ_COFF(2 * j)) = a;
                *reinterpret_cast<double *>(dst + ALZ_COFF(2 * j + 1)) = b;
              }
            }, pre);
      } else {
        char *dst = qring + (S * 2 + (int)(tile & 1)) * kSlot + cl * 16;
        section_tile_emit<nb_of(pbS), pbS, paS, FMA, S>(v, bc[0], na1[0], na2[0], dx[0], m1[0], m2[0],
            [&](int j, double a, double b) {
              if (ALZ_DBG(p, 32)) return;
              cdbl2 w;
              w.x = a;
              w.y = b;
              *reinterpret_cast<cdbl2 *>(dst + j * kPiece) = w;
            }, pre);
      }
    };
    PIPE_BARRIER();
    if constexpr (DEPHASE) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = 0.0;
      const bool even = (wave & 1) == 0;
      const int lag = kDeLag[wave & 3];
      if (wave == 0 && nt > 0) {                              // tile 0 is in the x ring (the loader saw to it)
        read_tile(0, v);
        PIPE_DRAIN();
      }
      for (int64_t t = 0; t < n_iv; ++t) {
        const int64_t a = t - lag;                            // the tile this stage computes in interval t
        // ---- first half ----
        if (even) {
          if (a >= 0 && a < nt) do_sections(v);
        } else {
          if (a - 1 >= 0 && a - 1 < nt) write_tile(a - 1, v);  // finished in the second half of t - 1
          if (a >= 0 && a < nt) read_tile(a, v);
        }
        PIPE_DRAIN();
        PIPE_BARRIER();
        // ---- second half ----
        if (even) {
          if (a >= 0 && a < nt) write_tile(a, v);
          if (a + 1 >= 0 && a + 1 < nt) read_tile(a + 1, v);
        } else {
          if (a >= 0 && a < nt) do_sections(v);
        }
        PIPE_DRAIN();
        PIPE_BARRIER();
      }
    } else if constexpr (OVL == 2) {
      double va[16], vb[16], vc[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) va[u] = vb[u] = vc[u] = 0.0;
      // interval t: write out tile a - 2 (done in the previous interval), fetch tile a, work on tile a - 1
      auto interval = [&](int64_t t, double (&out)[16], double (&cur)[16], double (&nxt)[16]) {
        const int64_t a = t - LAG * wave;
        if (a >= 2 && a - 2 < nt) write_tile(a - 2, out);
        if (a >= 0 && a < nt) read_tile(a, nxt);
        asm volatile("" ::: "memory");                        // LDS traffic is issued before the arithmetic
        if (a >= 1 && a - 1 < nt) do_sections(cur);
        PIPE_DRAIN();
        PIPE_BARRIER();
      };
      for (int64_t t = 0; t < n_iv; t += 3) {
        interval(t, va, vb, vc);
        interval(t + 1, vb, vc, va);
        interval(t + 2, vc, va, vb);
      }
    } else if (DIRECT_IN && wave == 0) {
      // tile k lives in register set k % 4; interval t fetches tile t + 2 and works on tile t - 1
      // (the same schedule as the LDS path below), so three tiles of loads are in flight
      const double *xl = CM ? p.x + grp.xbase + lane * p.ldx : p.x + in0 + lane;
      double s0[16], s1[16], s2[16], s3[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) s2[u] = s3[u] = 0.0;
      auto fetch = [&](int64_t tile, double (&v)[16]) {
        const int64_t tt = tile < nt ? tile : nt - 1;       // past the end: a valid tile again, never used
        if constexpr (CM) {                                   // the lane's own row: 128 contiguous bytes per tile
          const cdbl2 *src = reinterpret_cast<const cdbl2 *>(xl + tt * T);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const cdbl2 w = src[j];
            v[2 * j] = w.x;
            v[2 * j + 1] = w.y;
          }
        } else {
          const double *src = xl + tt * T * p.ldx;
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = src[u * p.ldx];
        }
      };
      fetch(0, s0);
      fetch(1, s1);
      auto interval = [&](int64_t t, double (&cur)[16], double (&far)[16]) {
        fetch(t + 2, far);
        asm volatile("" ::: "memory");                        // the loads are issued before the arithmetic
        if (t >= 1 && t - 1 < nt) work_tile(t - 1, cur);
        PIPE_DRAIN();
        PIPE_BARRIER();
      };
      for (int64_t t = 0; t < n_iv; t += 4) {
        interval(t, s3, s2);
        interval(t + 1, s0, s3);
        interval(t + 2, s1, s0);
        interval(t + 3, s2, s1);
      }
    } else if constexpr (OVL == 1) {
      double va[16], vb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) va[u] = vb[u] = 0.0;
      auto interval = [&](int64_t t, double (&cur)[16], double (&nxt)[16]) {
        const int64_t ahead = t - LAG * wave;               // tile to fetch; the one before it is in `cur`
        if (ahead >= 0 && ahead < nt) read_tile(ahead, nxt);
        asm volatile("" ::: "memory");                        // the reads are issued before the arithmetic
        if (ahead >= 1 && ahead - 1 < nt) work_tile(ahead - 1, cur);
        PIPE_DRAIN();
        PIPE_BARRIER();
      };
      int64_t t = 0;
      if (ALZ_PIPE_EARLYW == 2 && SPW == 1 && G == 64 && wave > 0 && !ALZ_DBG(p, 2)) {
        // steady intervals of stage `wave` (fetch tile t - lagw, work on the one before): t - lagw in [1, nt - 1];
        // whole (even, odd) pairs of them run through stage_pf, the rest through the general form
        const int64_t lagw = LAG * wave;
        const int64_t t0 = lagw + 2, t1 = lagw + nt - 1;
        const int64_t npairs = t1 >= t0 + 1 ? (t1 - t0 + 1) / 2 : 0;
        for (; t < t0; t += 2) {
          interval(t, va, vb);
          interval(t + 1, vb, va);
        }
        const int64_t tend = t0 + 2 * npairs;
        auto steady = [&](auto SI) {
          for (int64_t tt = t0; tt < tend; tt += 2) {
            stage_pf(SI, tt - lagw - 1, va, tt - lagw, vb);
            PIPE_DRAIN();
            PIPE_BARRIER();
            stage_pf(SI, tt - lagw, vb, tt + 1 - lagw, va);
            PIPE_DRAIN();
            PIPE_BARRIER();
          }
        };
        if (wave == 1) steady(std::integral_constant<int, 1>{});
        else if (wave == 2) steady(std::integral_constant<int, 2>{});
        else steady(std::integral_constant<int, 3>{});
        t = tend;
      }
      for (; t < n_iv; t += 2) {
        interval(t, va, vb);
        interval(t + 1, vb, va);
      }
    } else {
      for (int64_t t = 0; t < n_iv; ++t) {
        const int64_t tile = t - wave;
        if (tile >= 0 && tile < nt) {
          double v[16];
          read_tile(tile, v);
          work_tile(tile, v);
        }
        PIPE_DRAIN();
        PIPE_BARRIER();
      }
    }
    if (real) {
#pragma unroll
      for (int j = 0; j < SPW; ++j) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
          if (k < nbv[j] - 1 && !p.nostore) xhs[j][(int64_t)k * p.channels + c] = dx[j][k];
        if (nav[j] > 1) yhs[j][0 * p.channels + c] = m1[j];
        if (nav[j] > 2) yhs[j][1 * p.channels + c] = m2[j];
      }
    }
  }
  PIPE_CLOCK_REPORT();
#undef ALZ_COFF
}

#undef PIPE_BARRIER
#undef PIPE_DRAIN

typedef void (*casc_fn)(CArgs);

template <bool CM, bool BC = false>
static casc_fn pick_casc(const unsigned *pb, const unsigned *pa, int ns) {
#define ALZ_CASC(B0, A0, B1, A1, B2, A2, B3, A3, NS_)                                          \
  if (ns == NS_ && pb[0] == B0 && pa[0] == A0 && (NS_ < 2 || (pb[1] == B1 && pa[1] == A1)) && \
      (NS_ < 3 || (pb[2] == B2 && pa[2] == A2)) && (NS_ < 4 || (pb[3] == B3 && pa[3] == A3)))  \
    return (casc_fn)k_casc<CM, BC, B0, A0, B1, A1, B2, A2, B3, A3>;
  ALZ_CASC(3, 3, 3, 3, 3, 3, 3, 3, 4)        // gammatone.slaney
  ALZ_CASC(5, 3, 1, 3, 5, 3, 1, 3, 4)        // gammatone.klapuri
  ALZ_CASC(0xFE, 3, 1, 3, 1, 3, 1, 3, 4)     // gammatone.sampled (8-tap numerator, b0 == 0)
  ALZ_CASC(1, 3, 1, 3, 1, 3, 0, 0, 3)        // gammatone.sampled's sections 1 - 3 (time-parallel mode: section 0 runs by itself)
  ALZ_CASC(1, 1, 1, 1, 0, 0, 0, 0, 2)        // lowpass.pole twice as a cascade
  ALZ_CASC(7, 3, 7, 3, 0, 0, 0, 0, 2)        // two general biquads
  ALZ_CASC(7, 3, 7, 3, 7, 3, 7, 3, 4)        // four general biquads
#undef ALZ_CASC
  return nullptr;
}

template <bool CM, int SPW, int G = 64, bool FMA = false>
static casc_fn pick_pipe(const unsigned *pb, const unsigned *pa) {
#define ALZ_PIPE(B0, A0, B1, A1, B2, A2, B3, A3)                                                 \
  if (pb[0] == B0 && pa[0] == A0 && pb[1] == B1 && pa[1] == A1 && pb[2] == B2 && pa[2] == A2 &&  \
      pb[3] == B3 && pa[3] == A3)                                                                \
    return (casc_fn)k_pipe<CM, SPW, G, B0, A0, B1, A1, B2, A2, B3, A3, FMA>;
  ALZ_PIPE(3, 3, 3, 3, 3, 3, 3, 3)        // gammatone.slaney
  ALZ_PIPE(5, 3, 1, 3, 5, 3, 1, 3)        // gammatone.klapuri
  ALZ_PIPE(0xFE, 3, 1, 3, 1, 3, 1, 3)     // gammatone.sampled
#undef ALZ_PIPE
  return nullptr;
}

// Whole cascade in one pass when its section patterns are one of the fused combinations.
// Handles the full 16-sample tiles of the full 64-channel groups; reports what it covered.
// With `ch` (time-parallel mode, alz_scan.hip; time-major blocks: ch->time_major, k_casc only): every channel's block is cut into
// ch->n_chunks chunks of ch->chunk_len samples which run as n_chunks x channels virtual channels, each from /
// into its own state slot of ch->vxh[s] / ch->vyh[s]; the launch then covers the whole bank or nothing.
static int launch_cascade_impl(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream,
                               const CascChunks *ch, int64_t *done_samples, int64_t *done_channels,
                               const char **kernel_name) {
  *done_samples = 0;
  *done_channels = 0;
  if (nsec < 2 || nsec > 4) return ALZ_OK;
  unsigned pb[4] = {0, 0, 0, 0}, pa[4] = {0, 0, 0, 0};
  for (int s = 0; s < nsec; ++s) {
    if (secs[s].nb > 8 || secs[s].na > 3 || !secs[s].uniform || secs[s].any_div) return ALZ_OK;
    pb[s] = secs[s].present_b;
    pa[s] = secs[s].present_a;
    if ((pb[s] | pa[s]) == 0) return ALZ_OK;
    // the kernel keeps exactly as much input history as the highest present tap needs
    if (secs[s].nb != nb_of(pb[s])) return ALZ_OK;
    if (secs[s].na != (pa[s] & 2u ? 3 : pa[s] & 1u ? 2 : 1)) return ALZ_OK;
  }
  const bool cm = io.sxn == 1 && io.syn == 1 && !(io.sxc == 1 && io.syc == 1);
  const bool tm = io.sxc == 1 && io.syc == 1;
  if (!cm && !tm) return ALZ_OK;
  const int64_t ldx = cm ? io.sxc : io.sxn, ldy = cm ? io.syc : io.syn;
  // OUTER banks that read their input by input index: a workgroup's channels must be adjacent inputs of one band
  const bool by_input = io.mode == ALZ_BANK_OUTER && io.map_input;
  // (time-parallel mode on time-major blocks: ONE input stream is read by wave-uniform loads, any pitch)
  bool bcast = ch && ch->chunk_major && by_input && io.n_inputs == 1;
  if (((uintptr_t)io.x | (uintptr_t)io.y) & 15) return ALZ_OK;
  if (((bcast ? 0 : ldx) | ldy) & 1) return ALZ_OK;
  const int g = 64;
  int64_t tiles = io.n / 16, groups = io.channels / g;
  if (ch) {
    if (ch->chunk_len % 16 != 0 || (ch->chunk_len & 1)) return ALZ_OK;
    if (ch->n_chunks * ch->chunk_len != io.n) return ALZ_OK;
    if (ch->chunk_major) {
      // virtual channels: the 64 adjacent real channels of a group, in one chunk (k_casc only); channel-major blocks
      // only with the one-stream broadcast input
      if (io.channels % g != 0) return ALZ_OK;
      if (cm && !bcast) return ALZ_OK;
      if (!bcast && by_input && (io.n_inputs % g) != 0) return ALZ_OK;
    } else {
      // virtual channels: 64 consecutive chunks of one real channel per group
      if (!cm || ch->n_chunks % g != 0) return ALZ_OK;
    }
    tiles = ch->chunk_len / 16;
    groups = io.channels * ch->n_chunks / g;
  } else if (by_input && (io.n_inputs % g) != 0) {
    return ALZ_OK;
  }
  if (groups == 0 || tiles == 0) return ALZ_OK;
  const bool chunk_tm = ch && ch->chunk_major;
  // Four sections: the wave pipeline (one section per stage wave) while there are fewer 64-channel groups than
  // SIMDs; from 1024 groups up every SIMD has a whole single-wave cascade of its own and the hand-over only
  // costs (256 bands x 256 streams: k_casc 528 - 538 against k_pipe 436 - 472 Gsamples/s, profiles/NOTES_r02.md 14).
  // ALZ_TUNE: a run-time override exists in -DALZ_TUNING builds only (tools/variants).
  const int pipe_sel = chunk_tm ? 0 : ALZ_TUNE("ALZ_PIPE", groups >= 1024 ? 0 : 1);
  const bool fma = io.fused != 0;
  casc_fn pipe = nullptr;
  if (nsec == 4 && pipe_sel != 0)
    pipe = fma ? (cm ? pick_pipe<true, 1, 64, true>(pb, pa) : pick_pipe<false, 1, 64, true>(pb, pa))
               : (cm ? pick_pipe<true, 1>(pb, pa) : pick_pipe<false, 1>(pb, pa));
  const int pipe_waves = 6;   // four stage waves + loader + storer
  casc_fn fn = pipe ? pipe
               : cm ? (bcast ? pick_casc<true, true>(pb, pa, nsec) : pick_casc<true>(pb, pa, nsec))
                    : (bcast ? pick_casc<false, true>(pb, pa, nsec) : pick_casc<false>(pb, pa, nsec));
  if (!fn) return ALZ_OK;
  if (ch && ch->probe) {                  // (would the launch below take the block?)
    *done_samples = io.n;
    *done_channels = io.channels;
    *kernel_name = pipe ? (fma ? "k_pipe<fma>" : "k_pipe") : bcast ? "k_casc<bc>" : "k_casc";
    return ALZ_OK;
  }
  CArgs p;
  p.x = io.x; p.y = io.y; p.ldx = ldx; p.ldy = ldy; p.n_tiles = tiles;
  p.channels = io.channels; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.c_first = 0; p.mode = io.mode; p.map_input = io.map_input; p.nsec = nsec;
  static const int dbg_env = ALZ_DBG_ENV();
  p.dbg = dbg_env;
  p.kchunks = 0; p.ldx_outer = 0; p.ldy_outer = 0; p.nostore = 0;
  p.chunk_tm = 0; p.creal = io.channels; p.chunk_len = 0;
  p.stagger = ALZ_TUNE("ALZ_CASC_STAGGER", 0);   // (-DALZ_TUNING builds only: stagger_start, alz_common.h)
  for (int s = 0; s < 4; ++s) {
    const SectionDev &d = secs[s < nsec ? s : 0];
    p.nb[s] = d.nb; p.na[s] = d.na; p.b[s] = d.b; p.a[s] = d.a; p.xh[s] = d.xh; p.yh[s] = d.yh;
  }
  if (ch) {
    p.kchunks = ch->n_chunks; p.ldx_outer = ldx; p.ldy_outer = ldy;
    if (chunk_tm) { p.chunk_tm = 1; p.chunk_len = ch->chunk_len; }
    else p.ldx = p.ldy = ch->chunk_len;
    p.n_tiles = tiles;
    p.channels = io.channels * ch->n_chunks;
    p.nostore = ch->nostore ? 1 : 0;
    for (int s = 0; s < nsec; ++s) { p.xh[s] = ch->vxh[s]; p.yh[s] = ch->vyh[s]; }
  }
  const size_t pipe_slot = (size_t)g * 128 + (size_t)(g / 8) * 16;
  const size_t lds = pipe ? (size_t)(kPXRing + (pipe_waves - 3) * 2 + 2) * pipe_slot : (size_t)(bcast ? 2 : kCRing) * kCSlot;
  if (pipe) {
    const int rc = ensure_dynamic_lds((const void *)fn, (int)lds);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)groups), dim3(pipe ? 64 * pipe_waves : 64), lds, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *done_samples = ch ? io.n : tiles * 16;
  *done_channels = ch ? io.channels : groups * g;
  *kernel_name = pipe ? (fma ? "k_pipe<fma>" : "k_pipe") : bcast ? "k_casc<bc>" : "k_casc";
  return ALZ_OK;
}

int launch_cascade(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream,
                   int64_t *done_samples, int64_t *done_channels, const char **kernel_name) {
  return launch_cascade_impl(secs, nsec, io, stream, nullptr, done_samples, done_channels, kernel_name);
}

int launch_cascade_chunks(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream,
                          const CascChunks &ch, bool *taken, const char **kernel_name) {
  int64_t dn = 0, dc = 0;
  const int rc = launch_cascade_impl(secs, nsec, io, stream, &ch, &dn, &dc, kernel_name);
  *taken = dc > 0;
  return rc;
}

}  // namespace alz
