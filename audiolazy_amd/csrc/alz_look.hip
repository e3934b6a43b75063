// alz_look.hip -- the time-parallel mode of a narrow biquad-class bank in ONE pass over the block.
//
// alz_scan.hip runs a narrow bank (512 channels: one GPU's share of configs[1] sharded over 8) as chunks x channels
// lanes in three launches: zero-state pass (reads the block), scan of the chunk states, replay (reads it again,
// writes the result) -- 24 bytes of HBM traffic per sample for 16 algorithmic.  Here a chunk is small enough to stay
// in LDS between the two uses (512 samples x 16 channels = 64 KiB), so the block is read ONCE:
//
//   * a workgroup owns 16 channels and every W-th chunk of them (W workgroups per channel group, all resident, one per
//     CU); its four waves (one per SIMD) each loop over the workgroup's 64-sample tiles at their own pace and meet
//     only through progress counters in LDS:
//       LOAD    queues the tile DMA (global_load_lds) into a ring of 17 tile slots and turns a landed tile into
//               feed-forward sums p[n] IN PLACE;
//       REPLAY  runs only the serial part y[n] = (p[n] - a1 y[n-1]) - a2 y[n-2] from the chunk's TRUE start state,
//               k_duo's way (16 channels x 4 skewed copies, one LDS write per four rows), overwriting p with y; its
//               register ring of p rows runs on across tile boundaries (p is prepared many tiles ahead);
//       HELP    stores finished tiles, and forms every chunk's zero-state end state z_j WITHOUT a second recurrence:
//               the end state of a tile is a dot product of its p rows with the impulse response of 1/A(z) (32 FMAs
//               per lane and tile, weights computed once per launch), carried from tile to tile through A^64 and
//               published in global memory with the chunk's last tile (64-bit agent-scope atomic stores, every word
//               twice: as it is and XORed with a key no other launch has -- a reader takes a word only when the pair
//               matches, so whatever the array held before, from an earlier launch or from nobody, is simply "not
//               yet published": no flags, no fences, and since round 6 no pre-filling either);
//       CHAIN   hands REPLAY the true start state of every chunk, which needs no other workgroup's replay: it is
//               chained in zero-state space, S_j = M ( ... M (M S_{j-W} + z_{j-W}) + z_{j-W+1} ... ) + z_{j-1}
//               (M = A^512 per channel, the matrix alz_scan.hip caches), from the workgroup's own previous start state
//               and the z of the W chunks in between, fetched (eight 1 KiB global -> LDS transfers, repeated until
//               every pair matches the key) as soon as the neighbours publish them.  It also CHECKS the result of all
//               this where it can be checked (round 6): every start state is published like the z, and two chunks
//               later the replay's TRUE state at the end of chunk j is compared with the start state the neighbour
//               used for chunk j + 1 -- a chunk that started from a wrong state cannot leave the kernel unreported.
//     Waits only ever point to smaller chunk indices and earlier tiles, and every workgroup of the launch is resident:
//     no deadlock; bounded spins guard against the impossible.
//
// What was measured on the way (profiles/NOTES_r03.md 7, profiles/r03_look_ablations.log): the zero-state pass as a
// second recurrence -- in a third wave, or in half the replay wave's lanes -- cost the replay wave 15 .. 40 % (its
// dependent chain of 3 FP64 operations per step is the budget: 28.3 cycles per step, 64 steps per tile); as a dot
// product it is free.  One barrier per tile made every wave wait for the slowest of each interval (memory stalls
// included): counters instead; the chain of the start states in a wave of its own (it took 3000 - 4000 cycles once
// per chunk out of HELP's or LOAD's tile budget).  260 Gsamples/s at 512 channels x 2^20 (267 at 2048) with 16.5 B/sample of HBM
// traffic, against 237 with 24 B/sample for the three-launch form: ALZ_TP_AUTO takes it from 256 channels up.  The
// replay wave alone runs at 276; the chain itself would allow ~305 at the clock the chip holds under this load.
//
// Every output sample is still the reference's DF-I statement (lazy_filters.py:197-257) in the kernels' own order;
// only the chunk-start states carry a different rounding -- the same numerics as the three-launch mode (1e-10 ..
// 1e-9 on the configs[1] resonators).  a0 == 1, channels % 16 == 0, blocks of whole 512-sample chunks; everything else
// stays on the three-launch mode.
//
// Round 5: both layouts, the |x| input map, in place.  CM (channel-major blocks [C, N]): a tile arrives as eight 1 KiB
// transfers of two channel rows each (k_duo's channel-major DMA layout) and LOAD's feed-forward pass -- which reads the
// whole tile into registers before it writes anything -- leaves p in rows of one CHANNEL (528-byte pitch: the replay's
// four skewed copies then read and write eight bytes apart, conflict-free per half-wave), from which HELP stores 16-byte
// pieces of a channel's row.  PRE = 1: |x| on LOAD's reads (the bank's input history is kept mapped, as k_duo keeps it).
// In place (x == y): a tile is stored long after it was read, and by the workgroup that read it; only the two rows in
// front of a chunk belong to another workgroup, so a small launch saves those for every chunk first (k_look_hsave).
#include "alz_common.h"
#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace alz {

namespace {

constexpr int kChunks = 8;                   // 1 KiB DMA transfers per tile
constexpr int kNT = 8;                       // tiles per chunk (L = 512)
// (the #ifndef values can be overridden in tools/variants builds: profiles/NOTES_r03.md 7 has what was tried)
#ifndef ALZ_LOOK_SLOTS
#define ALZ_LOOK_SLOTS 17
#endif
#ifndef ALZ_LOOK_LEAD
#define ALZ_LOOK_LEAD 3      // tiles of DMA in flight ahead of the one being prepared
#endif
#ifndef ALZ_LOOK_BEHIND
#define ALZ_LOOK_BEHIND 13   // HELP stores tile i - 13 in the iteration in which it sums tile i
#endif
#ifndef ALZ_LOOK_CHUNKWAIT
#define ALZ_LOOK_CHUNKWAIT 1
#endif
#ifndef ALZ_LOOK_VAR
#define ALZ_LOOK_VAR 0       // experiments: 1 no LDS writes, 2 no LDS reads in the replay's tile (WRONG output)
#endif
constexpr int kSlots = ALZ_LOOK_SLOTS;       // tile slots in LDS
constexpr int kSlot = 8192 + kChunks * 16;   // a tile in the DMA layout (16 bytes of pad per 1 KiB chunk)
constexpr int kHist = 256;                   // the two rows before a tile: [2][16] doubles
constexpr int kPub = 64;                     // 64-bit words per published chunk state: [value 1, value 1 ^ key, value 2, value 2 ^ key][16 channels]
constexpr int kMaxW = 16;                    // workgroups per channel group (the predecessors' states live in registers)
#ifdef ALZ_ABLATE
#define ALZ_LOOK_CAP(p) (((p).dbg & 1023) ? 1 : kSpinCap)      // (an ablated run publishes nothing: do not wait for it)
#else
#define ALZ_LOOK_CAP(p) kSpinCap
#endif
constexpr int kSpinCap = 1 << 18;            // ~0.1 s: three orders of magnitude beyond any legitimate wait

struct LArgs {
  const double *x;
  double *y;
  int64_t ldx, ldy;
  int64_t n_chunks;            // K: chunks of 512 samples per channel
  int64_t channels, n_inputs, n_sets;
  int workers;                 // W workgroups per 16-channel group
  int nb, na;
  const double *b, *a;
  double *xh, *yh;             // the bank's state [taps-1][channels]
  const double *power;         // M = A^512 per channel: [4][channels] (M11 M12 M21 M22)
  unsigned long long *z;       // [groups][n_chunks][kPub] published zero-state end states
  unsigned long long *s;       // [groups][n_chunks][kPub] published chunk start states (for the check at the chunk boundaries)
  unsigned long long key;      // this launch's key: a published word counts when word ^ its twin == key
  const double *hsave;         // in place: the two input rows in front of every chunk, [groups][n_chunks][32] in the
                               // history transfer's own order (time-major [2][16], channel-major [16][2]); else nullptr
  int *err;                    // [kLookErrWords] of pinned host memory: word W_* set when that wait ran into its cap
  int dbg;                     // -DALZ_ABLATE builds only (timing experiments, WRONG output): 2 no replay arithmetic,
                               // 4 no feed-forward pass, 8 no stores, 16 no tile DMA, 32 no chunk-state chain,
                               // 64 no zero-state sums; with -DALZ_LOOK_TIMING 256 / 512 print the replay / the other waves' cycle counts;
                               // 1024 the workgroups of chunk 0 start ~40 us late, 2048 the input history is written as soon as the last
                               // tile lands (round 5's order) -- together they show the race of NOTES_r06.md 1 on every launch
};

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
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

// the same with agent-scope coherence (sc1: the data was written by another XCD's atomic stores, L2s are per XCD)
__device__ __forceinline__ void dma16_coherent(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off sc1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

typedef double dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(double *gdst, dbl2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
}

}  // namespace

// One 64-sample tile of the replay wave: 16 channels x 4 skewed copies (copy q runs q steps behind, so one
// ds_write_b64 of the wave stores four finished rows -- k_duo's scheme), over p in the tile slot, overwriting it with y.
//   cur     the lane's column in the tile slot, minus its skew (the lane's step u is row u - q = cur + u * 128)
//   nxt     the same for the NEXT tile of the workgroup's sequence: p is prepared many tiles ahead here, so the
//           register ring of 8-row groups (pr, four deep so that a tile's eight groups keep their places) simply runs
//           on across the tile boundary and no tile starts by waiting for its first rows
//   CS      instantiation for a tile that opens a chunk: copy q switches to the chunk's start state (s1, s2) at its
//           step q (its steps before that finish the previous tile of the sequence)
//   k1..k3  p of the previous tile's last three rows (the lagging copies' first steps; the replay has overwritten
//           them by now), replaced by this tile's for the next call
struct LookState { double m1, m2, t2, k1, k2, k3; };
template <unsigned PA, bool CS, int kStep>
__device__ __forceinline__ void look_tile(char *cur, const char *nxt, int q, double s1, double s2, double na1,
                                          double na2, LookState &st, double (&pr)[4][8]) {
  constexpr int T = 64, NCH = T / 8;
  double m1 = st.m1, m2 = st.m2, t2 = st.t2;
  double n1 = 0.0, n2 = 0.0, n3 = 0.0;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
#if (ALZ_LOOK_VAR & 2)
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(pr[(k + 2) % 4][u]));
#else
    if (k + 2 < NCH) {
#pragma unroll
      for (int u = 0; u < 8; ++u) pr[(k + 2) % 4][u] = *reinterpret_cast<const double *>(cur + ((k + 2) * 8 + u) * kStep);
    } else {
      const int kk = k + 2 - NCH;                          // the next tile's group 0 / 1
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = kk * 8 + u;                          // (rows before the tile: any address of it, see k1..k3)
        pr[(k + 2) % 4][u] = *reinterpret_cast<const double *>(nxt + ((kk == 0 && u < 3 && u < q) ? q : r) * kStep);
      }
    }
#endif
    if (k == NCH - 2) {                                    // before the last group overwrites them
      n1 = *reinterpret_cast<const double *>(cur + (q + T - 1) * kStep);
      n2 = *reinterpret_cast<const double *>(cur + (q + T - 2) * kStep);
      n3 = *reinterpret_cast<const double *>(cur + (q + T - 3) * kStep);
    }
    __builtin_amdgcn_sched_barrier(0);
#if ALZ_LOOK_CHUNKWAIT
    {
      // ONE wait per group for its four paired reads, through the builtin so that the compiler's wait-count pass sees
      // it and does not put a wait of its own into the chain (k_duo's finding).  LDS operations issued after group k's
      // reads that may stay outstanding: one paired write per group, four paired reads per group, and the eight reads
      // that open the next tile (its first rows and this tile's last three); a count too small only waits longer.
      switch (k == 0 ? 4 : k == 1 ? 9 : k < NCH - 2 ? 10 : 13) {   // (k is an unrolled loop index: folded at compile time)
        case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
        case 9: __builtin_amdgcn_s_waitcnt(0xC97F); break;
        case 10: __builtin_amdgcn_s_waitcnt(0xCA7F); break;
        default: __builtin_amdgcn_s_waitcnt(0xCD7F); break;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      double acc = pr[k % 4][u];
      if (k == 0 && u < 3) {
        const int d = q - u;                               // > 0: row -d of the tile
        acc = d == 1 ? st.k1 : d == 2 ? st.k2 : d == 3 ? st.k3 : acc;
      }
      if constexpr (CS) {
        if (k == 0 && u <= 3) {
          const bool now = u == q;
          m1 = now ? s1 : m1;
          m2 = now ? s2 : m2;
          t2 = now ? na2 * s2 : t2;
        }
      }
      double t2n = 0.0;
      if constexpr (PA == 3u) {
        const double t1 = na1 * m1;
        t2n = na2 * m1;
        acc = (acc + t1) + t2;
      } else {
        if constexpr (PA & 1u) acc = acc + na1 * m1;
        if constexpr (PA & 2u) acc = acc + na2 * m2;
      }
      m2 = m1;
      m1 = acc;
      t2 = t2n;
#if !(ALZ_LOOK_VAR & 1)
      if ((u & 3) == 3) *reinterpret_cast<double *>(cur + (k * 8 + u) * kStep) = acc;   // rows u - 3 .. u, one per copy
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  st = LookState{m1, m2, t2, n1, n2, n3};
}

// The sum of a value over the wave's four 16-lane rows, in every lane: gfx950's row / half swaps (two VALU operations per
// 32-bit half and level) instead of four LDS round trips through ds_bpermute.  v_permlane16_swap exchanges the odd rows
// of its first operand with the even rows of its second, v_permlane32_swap the upper half with the lower half: with
// both operands the same value, the two results hold the two partners of every lane.
__device__ __forceinline__ double sum_rows(double x) {
  unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  auto a0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto a1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double y = __hiloint2double((int)a1[0], (int)a0[0]) + __hiloint2double((int)a1[1], (int)a0[1]);
  lo = (unsigned)__double2loint(y); hi = (unsigned)__double2hiint(y);
  auto b0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b1[0], (int)b0[0]) + __hiloint2double((int)b1[1], (int)b0[1]);
}

enum { F_PREPARED = 0, F_STORED, F_REPLAYED, F_CHUNK, F_ZDONE, F_INIT, F_COUNT = 8 };
// Which bounded wait ran out (word of LArgs::err; launch_look's caller reads them all): wave / what it waited for.
enum { W_ANY = 0, W_LOAD_STORED, W_HELP_PREPARED, W_HELP_REPLAYED, W_HELP_INIT, W_CHAIN_Z, W_CHAIN_REPLAYED, W_CHAIN_ZDONE,
       W_CHAIN_PREPARED, W_REPLAY_CHUNK, W_LOAD_FINAL, W_CHAIN_CHECK_WAIT, W_CHECK_FAILED, W_COUNT };
static_assert(W_COUNT <= kLookErrWords, "one word of pinned host memory per wait site");

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#if defined(ALZ_ABLATE) && defined(ALZ_LOOK_TIMING)     // (its own switch: a clock read also waits for the wave's LDS operations)
#define ALZ_LOOK_CLOCK(n) long long cyc[n] = {}, c_prev = __builtin_readcyclecounter();
#define ALZ_LOOK_MARK(k) { const long long c_now = __builtin_readcyclecounter(); cyc[k] += c_now - c_prev; c_prev = c_now; }
#else
#define ALZ_LOOK_CLOCK(n)
#define ALZ_LOOK_MARK(k)
#endif

constexpr int kChanPitch = 64 * 8 + 16;       // channel-major p / y rows in a tile slot
constexpr int kSlotCM = 16 * kChanPitch;      // 8448 bytes (the DMA layout's 8 x 1040 fits inside)
template <int PRE>
__device__ __forceinline__ double look_pre(double v) {
  if constexpr (PRE == 1) return __builtin_fabs(v);
  return v;
}

template <unsigned PB, unsigned PA, bool CM, int PRE>
__global__ __launch_bounds__(256) void k_look(LArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 16, T = 64, NT = kNT;
  constexpr int kStep = CM ? 8 : G * 8;        // bytes from a channel's sample to its next in the p / y image of a tile
  constexpr int kSlot = CM ? kSlotCM : alz::kSlot;
  // The four waves each run their own loop over this workgroup's tile sequence and meet only through progress
  // counters (no barrier after the first: a barrier per tile made every wave wait for the slowest of each interval,
  // memory stalls included).  In its iteration i
  //   LOAD    queues the DMA of tile i + kDmaLead (its slot must have been stored) and prepares tile i + 1;
  //   HELP    forms tile i's part of its chunk's zero-state end state (published with the chunk's last tile) and stores
  //           tile i - kStoreBehind (the replay must be done with it);
  //   REPLAY  works on tile i, waiting only at a chunk's first tile for the chunk's start state;
  //   CHAIN   (one iteration per chunk) fetches the states the chunk's start state is chained from, as soon as they
  //           are published, chains it, and hands it to REPLAY once the chunk's tiles are prepared and HELP has read
  //           them (the replay overwrites p with y).
  constexpr int kDmaLead = ALZ_LOOK_LEAD, kStoreBehind = ALZ_LOOK_BEHIND;
  static_assert(kSlots >= kStoreBehind + kDmaLead + 1, "a slot is stored before it is refilled");
  static_assert(kStoreBehind >= NT + 3, "HELP must be able to finish a chunk's sums before the replay needs its state");
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cl = lane & 15, q = lane >> 4;
  const int W = p.workers;
  const int64_t group = (int64_t)blockIdx.x / W;
  const int w = (int)((int64_t)blockIdx.x - group * W);
  const int64_t c0 = group * G, c = c0 + cl;
  const int64_t K = p.n_chunks;
  const int my_chunks = (int)((K - w + W - 1) / W);           // chunks w, w + W, ...
  const int TOT = my_chunks * NT;
  const int n_iv = TOT + kStoreBehind + 1;
  const int64_t set = p.n_inputs ? c / p.n_inputs : ((p.n_sets == 1) ? 0 : c);
  char *hist = smem + kSlots * kSlot;                         // [kSlots][2][16] doubles: rows -2, -1 of every tile
  char *zlds = hist + kSlots * kHist;                         // [kMaxW][kPub] words: the requested chunk end states
  char *sbuf = zlds + kMaxW * kPub * 8;                       // [2][2][16] doubles: chunk start states, by chunk parity
  char *ebuf = sbuf + 512;                                    // [3][2][16] doubles: the replay's true state at the end of a chunk, by chunk % 3
  char *vlds = ebuf + 768;                                    // [kPub] words: a neighbour's published start state (the boundary check)
  int *flags = reinterpret_cast<int *>(vlds + kPub * 8);      // [F_COUNT] progress counters
  int cap = 1 << 22;                                          // (~0.3 s of polling)
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lane_off = CM ? cl * kChanPitch : cl * 8;        // the lane's channel in the p / y image of a tile
  // the same channel in the tile as the DMA leaves it (channel-major: transfer cl / 2, its first or second 512 bytes)
  const int xlane_off = CM ? (cl >> 1) * (1024 + 16) + (cl & 1) * 512 : cl * 8;
#define ALZ_EOFF(u) (CM ? (u) * 8 : (u) * G * 8 + (((u) * G) >> 7) * 16)
  // global row of tile t's first sample: chunk (w + (t / NT) W), tile t % NT of it
  auto tile_row = [&](int t) -> int64_t { return ((int64_t)w + (int64_t)(t / NT) * W) * (NT * T) + (t % NT) * T; };
  unsigned long long *zg = p.z + group * K * kPub;            // this group's [K][kPub]
  unsigned long long *sg = p.s + group * K * kPub;
  const unsigned long long key = p.key;
  auto publish_pair = [&](unsigned long long *dst, double v1, double v2) {   // (the 16 lanes of one copy; dst = the chunk's kPub words)
    const unsigned long long b1 = (unsigned long long)__double_as_longlong(v1), b2 = (unsigned long long)__double_as_longlong(v2);
    __hip_atomic_store(dst + cl, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 16 + cl, b1 ^ key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 32 + cl, b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 48 + cl, b2 ^ key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  double na1 = 0, na2 = 0;
  if (PA & 1u) na1 = -p.a[1 * p.n_sets + set];
  if (PA & 2u) na2 = -p.a[2 * p.n_sets + set];
  if (threadIdx.x < F_COUNT) flags[threadIdx.x] = 0;
  __syncthreads();
#if defined(ALZ_ABLATE)
  if (ALZ_DBG(p, 1024) && w == 0) {                          // (a late start of chunk 0's workgroups, as a slow XCD would give)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 4000) __builtin_amdgcn_s_sleep(8);
  }
#endif

  if (wave == 1) {
    // ------------------------------ LOAD ------------------------------
    const int row = lane / 8, cp = lane % 8;
    // time-major: transfer j = rows 8 j .. 8 j + 7 of the tile, 128 bytes (16 channels) each;
    // channel-major: transfer j = channels 2 j, 2 j + 1, the tile's 512 bytes of each
    const double *xg = CM ? p.x + (c0 + (lane >> 5)) * p.ldx + 2 * (lane & 31) : p.x + (int64_t)row * p.ldx + c0 + 2 * cp;
    const int64_t x_chunk = CM ? 2 * p.ldx : 8 * p.ldx;
    double b0 = 0, b1 = 0, b2 = 0;
    if (PB & 1u) b0 = p.b[0 * p.n_sets + set];
    if (PB & 2u) b1 = p.b[1 * p.n_sets + set];
    if (PB & 4u) b2 = p.b[2 * p.n_sets + set];
    const double h1 = (p.nb > 1) ? p.xh[0 * p.channels + c] : 0.0;   // x[-1], x[-2] of the stream (chunk 0 only)
    const double h2 = (p.nb > 2) ? p.xh[1 * p.channels + c] : 0.0;
    double bb0 = b0, bb1 = b1, bb2 = b2, hh1 = h1, hh2 = h2;
    asm volatile("" : "+v"(bb0), "+v"(bb1), "+v"(bb2), "+v"(hh1), "+v"(hh2));
    // nine transfers per tile: the tile, and (16 lanes only) the two rows before it
    auto queue_tile = [&](int t) {
      const int s = t % kSlots;
      const int64_t r0 = tile_row(t);
      const double *src = CM ? xg + r0 : xg + r0 * p.ldx;
#pragma unroll
      for (int j = 0; j < kChunks; ++j) dma16(src + j * x_chunk, lds0 + s * kSlot + j * (1024 + 16));
      // rows r0 - 2, r0 - 1 (clamped to row 0 for the very first tile, which uses the bank's history instead):
      // time-major [2][16] -- lane l: row l / 8, channels 2 (l % 8), + 1; channel-major [16][2] -- lane l: channel l.
      // In place, the rows in front of a CHUNK are another workgroup's to overwrite: those come from the saved copy.
      const int64_t rh = r0 >= 2 ? r0 - 2 : 0;
      const double *hsrc = CM ? p.x + (c0 + (lane & 15)) * p.ldx + rh : p.x + (rh + (lane >> 3 & 1)) * p.ldx + c0 + 2 * (lane & 7);
      if (p.hsave && (t % NT) == 0 && r0 > 0) hsrc = p.hsave + (group * K + r0 / (NT * T)) * 32 + 2 * (lane & 15);
      if (lane < 16) dma16(hsrc, lds0 + (unsigned)(hist - smem) + s * kHist);
    };
    auto prepare_tile = [&](int t) {
      char *ps = smem + (t % kSlots) * kSlot + lane_off;            // where the lane's p goes
      const char *xs = smem + (t % kSlots) * kSlot + xlane_off;     // where its x lies (time-major: the same column)
      const char *hs = hist + (t % kSlots) * kHist + (CM ? cl * 16 : cl * 8);
      // (time-major: 16 bytes of pad after every eighth row of the landed tile; sample 4 j + q - 1 / - 2 lies before the
      // pad of sample 4 j + q's block when j is even and q is small)
      const int adj1 = (!CM && q == 0) ? 16 : 0, adj2 = (!CM && q < 2) ? 16 : 0;
      constexpr int kXStep = CM ? 8 : G * 8;
      const char *x_d0 = xs + q * kXStep;
      const char *x_d1[2] = {xs + (q - 1) * kXStep - adj1, xs + (q - 1) * kXStep};   // [j odd]
      const char *x_d2[2] = {xs + (q - 2) * kXStep - adj2, xs + (q - 2) * kXStep};
      double x0[16], x1[16], x2[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if constexpr (PB & 1u) x0[j] = look_pre<PRE>(*reinterpret_cast<const double *>(x_d0 + ALZ_EOFF(4 * j)));
        if constexpr (PB & 2u) {
          if (j > 0) x1[j] = look_pre<PRE>(*reinterpret_cast<const double *>(x_d1[j & 1] + ALZ_EOFF(4 * j)));
        }
        if constexpr (PB & 4u) {
          if (j > 0) x2[j] = look_pre<PRE>(*reinterpret_cast<const double *>(x_d2[j & 1] + ALZ_EOFF(4 * j)));
        }
      }
      if constexpr ((PB & 6u) != 0) {
        // x[-2], x[-1] relative to this tile: the landed history rows, or the bank's history at the start of the stream
        // (which the bank keeps MAPPED, like k_duo)
        const bool stream_start = tile_row(t) == 0;
        const double q2 = look_pre<PRE>(*reinterpret_cast<const double *>(hs));
        const double q1 = look_pre<PRE>(*reinterpret_cast<const double *>(hs + (CM ? 8 : 128)));
        const double pm1 = stream_start ? hh1 : q1, pm2 = stream_start ? hh2 : q2;
        const double s0 = look_pre<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(0)));
        const double s1 = look_pre<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(1)));
        const double s2 = look_pre<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(2)));
        if constexpr (PB & 2u) x1[0] = q == 0 ? pm1 : q == 1 ? s0 : q == 2 ? s1 : s2;
        if constexpr (PB & 4u) x2[0] = q == 0 ? pm2 : q == 1 ? pm1 : q == 2 ? s0 : s1;
      }
      double acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double v = 0.0;
        bool first = true;
        if constexpr (PB & 1u) { v = bb0 * x0[j]; first = false; }
        if constexpr (PB & 2u) { const double t2 = bb1 * x1[j]; v = first ? t2 : v + t2; first = false; }
        if constexpr (PB & 4u) { const double t2 = bb2 * x2[j]; v = first ? t2 : v + t2; first = false; }
        acc[j] = v;
      }
      // in place: every read of the tile is done (one wave, program order: the whole tile is in registers), p goes to
      // unpadded rows of the same slot (channel-major: to rows of one channel each, a different image than the DMA's)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) *reinterpret_cast<double *>(ps + (4 * j + q) * kStep) = acc[j];
    };
    // The input history the bank keeps for the next block: the last two x rows of the block (owner of the last chunk).
    // They are picked up when the last tile lands and WRITTEN ONLY AFTER THE LOOP: the workgroup of chunk 0 reads the
    // same words (hh1, hh2) when it starts, and nothing orders the start of one workgroup against the progress of
    // another's LOAD wave -- with one or two chunks per workgroup this wave reaches its last tile without waiting for
    // anybody (round 6: the race behind the one wrong block of round 5, profiles/NOTES_r06.md 1).  After the loop every
    // tile of this workgroup has been stored, hence replayed, hence chained from z_0, which chunk 0's workgroup
    // published after it had prepared -- i.e. after it had read its history.
    const bool owns_last = ((K - 1) % W) == w;
    double keep_x1 = 0.0, keep_x2 = 0.0;
#if defined(ALZ_ABLATE)
    const bool early_state = ALZ_DBG(p, 2048);               // (the round-5 order, for the demonstration of the race)
#else
    constexpr bool early_state = false;
#endif
    for (int t = 0; t < kDmaLead && t < TOT; ++t) queue_tile(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prepare_tile(0);
    publish(flags + F_PREPARED, 1, lane);
    ALZ_LOOK_CLOCK(4)                                          // slot wait / queue / transfer wait / prepare
    for (int i = 0; i < n_iv; ++i) {
      // the slot of tile i + kDmaLead must have been stored (this also paces the idle iterations of the tail)
      {
        int need = i + kDmaLead - kSlots + 1;
        need = need > TOT ? TOT : need;
        if (need > 0) await(flags + F_STORED, need, cap, p.err, W_LOAD_STORED);
      }
      ALZ_LOOK_MARK(0)
      if (i + kDmaLead < TOT && !ALZ_DBG(p, 16)) queue_tile(i + kDmaLead);
      ALZ_LOOK_MARK(1)
      // issued after tile i + 1's transfers: those of kDmaLead - 1 more tiles (the last tiles of the sequence simply
      // wait for everything)
      static_assert((kDmaLead - 1) * (kChunks + 1) <= 63, "vmcnt is a 6-bit count");
      if (i + kDmaLead < TOT) wait_vm<(kDmaLead - 1) * (kChunks + 1)>();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (i + 1 < TOT) {
        if (owns_last && i + 1 == TOT - 1 && q == 3) {        // (before the tile is overwritten with p)
          const char *xs = smem + ((i + 1) % kSlots) * kSlot + xlane_off;
          keep_x1 = look_pre<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 1)));
          keep_x2 = look_pre<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 2)));
          if (early_state) {
            if (p.nb > 1) p.xh[0 * p.channels + c] = keep_x1;
            if (p.nb > 2) p.xh[1 * p.channels + c] = keep_x2;
          }
        }
        ALZ_LOOK_MARK(2)
        if (!ALZ_DBG(p, 4)) prepare_tile(i + 1);
        publish(flags + F_PREPARED, i + 2, lane);
      }
      ALZ_LOOK_MARK(3)
    }
#if defined(ALZ_ABLATE) && defined(ALZ_LOOK_TIMING)
    if (blockIdx.x == 5 && lane == 0 && (p.dbg & 512))
      printf("k_look LOAD wave, cycles per iteration: slot wait %.1f queue %.1f transfer wait %.1f prepare %.1f\n",
             (double)cyc[0] / n_iv, (double)cyc[1] / n_iv, (double)cyc[2] / n_iv, (double)cyc[3] / n_iv);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    await(flags + F_STORED, TOT, cap, p.err, W_LOAD_FINAL);  // (the loop's last iteration has waited for this already)
    if (owns_last && q == 3 && !early_state) {
      if (p.nb > 1) p.xh[0 * p.channels + c] = keep_x1;
      if (p.nb > 2) p.xh[1 * p.channels + c] = keep_x2;
    }
  } else if (wave == 2) {
    // ------------------------------ HELP ------------------------------
    const int row = lane / 8, cp = lane % 8;
    // store j of a tile -- time-major: rows 8 j .. 8 j + 7 (128 bytes each); channel-major: 512 bytes of channels 2 j, 2 j + 1
    double *yg = CM ? p.y + (c0 + (lane >> 5)) * p.ldy + 2 * (lane & 31) : p.y + (int64_t)row * p.ldy + c0 + 2 * cp;
    const int64_t y_chunk = CM ? 2 * p.ldy : 8 * p.ldy;
    // The zero-state end state of a tile is a dot product, not a recurrence: with h the impulse response of 1/A(z),
    // (y[63], y[62]) = sum_r (h[63-r], h[62-r]) p[r].  This lane takes rows 4 j + q: its 2 x 16 weights, and
    // A^64 = [[h64, -a2 h63], [h63, -a2 h62]] to carry the sum from tile to tile, come from 65 steps of h's own
    // recurrence, once per launch.
    double g1[16], g2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { g1[j] = 0.0; g2[j] = 0.0; }
    double M11, M12, M21, M22;
    {
      double hk = 1.0, hp = 0.0, h62 = 0.0, h63 = 0.0, h64 = 0.0;
#pragma unroll
      for (int k = 0; k <= 64; ++k) {
        if (k <= 63) { const int r = 63 - k; g1[r >> 2] = (q == (r & 3)) ? hk : g1[r >> 2]; }
        if (k <= 62) { const int r = 62 - k; g2[r >> 2] = (q == (r & 3)) ? hk : g2[r >> 2]; }
        if (k == 62) h62 = hk;
        if (k == 63) h63 = hk;
        if (k == 64) h64 = hk;
        const double nx = __builtin_fma(na1, hk, na2 * hp);
        hp = hk;
        hk = nx;
      }
      M11 = h64; M12 = na2 * h63; M21 = h63; M22 = na2 * h62;
    }
    double Z1 = 0.0, Z2 = 0.0;
    ALZ_LOOK_CLOCK(7)                                          // - / prepared wait / store / zero-state / - / replayed wait / -
    for (int i = 0; i < n_iv; ++i) {
      // tile i's rows for the zero-state sums, read first: they arrive while the finished tile is being stored
      const bool zs_on = i < TOT && !ALZ_DBG(p, 64);
      const int ts = i - kStoreBehind;
      const bool st_on = ts >= 0 && ts < TOT;
      if (i < TOT) await(flags + F_PREPARED, i + 1, cap, p.err, W_HELP_PREPARED);
      ALZ_LOOK_MARK(1)
      if (st_on) await(flags + F_REPLAYED, ts + 1, cap, p.err, W_HELP_REPLAYED);
      ALZ_LOOK_MARK(5)
      double zp[16];
      if (zs_on) {
        const char *zsrc = smem + (i % kSlots) * kSlot + lane_off + q * kStep;
#pragma unroll
        for (int j = 0; j < 16; ++j) zp[j] = *reinterpret_cast<const double *>(zsrc + j * 4 * kStep);
      }
      if (i < TOT) publish(flags + F_ZDONE, i + 1, lane);    // (behind the reads in the LDS's order: the replay may overwrite)
      // the tile the replay has finished
      if (st_on) {
        const char *ys = smem + (ts % kSlots) * kSlot;
        double *yt = CM ? yg + tile_row(ts) : yg + tile_row(ts) * p.ldy;
        dbl2 v[kChunks];
#pragma unroll
        for (int j = 0; j < kChunks; ++j)
          v[j] = *reinterpret_cast<const dbl2 *>(CM ? ys + (2 * j + (lane >> 5)) * kChanPitch + (lane & 31) * 16 : ys + j * 1024 + lane * 16);
        publish(flags + F_STORED, ts + 1, lane);             // (behind the reads in the LDS's order: the slot is free)
        if (!ALZ_DBG(p, 8)) {
#pragma unroll
          for (int j = 0; j < kChunks; ++j) store16(yt + j * y_chunk, v[j]);
        }
      }
      ALZ_LOOK_MARK(2)
      // tile i's part of its chunk's zero-state end state
      if (zs_on) {
        double u1 = 0.0, u2 = 0.0, v1 = 0.0, v2 = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          u1 = __builtin_fma(g1[j], zp[j], u1);
          u2 = __builtin_fma(g2[j], zp[j], u2);
          v1 = __builtin_fma(g1[j + 1], zp[j + 1], v1);
          v2 = __builtin_fma(g2[j + 1], zp[j + 1], v2);
        }
        // the four row classes of a channel sit 16 lanes apart; then carry the chunk's sum over this tile
        const double z1 = sum_rows(u1 + v1), z2 = sum_rows(u2 + v2);
        const int tt = i % NT;
        const double c1 = __builtin_fma(M11, Z1, __builtin_fma(M12, Z2, z1));
        const double c2 = __builtin_fma(M21, Z1, __builtin_fma(M22, Z2, z2));
        Z1 = tt == 0 ? z1 : c1;
        Z2 = tt == 0 ? z2 : c2;
        // (the first state goes out only once this workgroup's CHAIN wave has read the bank's state: the owner of the last
        // chunk overwrites that state when everything before it -- every workgroup's first chunk included -- is done)
        if (i == NT - 1) await(flags + F_INIT, 1, cap, p.err, W_HELP_INIT);
        if (tt == NT - 1 && q == 0) {
          const int64_t j = (int64_t)w + (int64_t)(i / NT) * W;
          publish_pair(zg + j * kPub, Z1, Z2);
        }
      }
      ALZ_LOOK_MARK(3)
    }
#if defined(ALZ_ABLATE) && defined(ALZ_LOOK_TIMING)
    if (blockIdx.x == 5 && lane == 0 && (p.dbg & 512))
      printf("k_look HELP wave, cycles per iteration: prepared wait %.1f replayed wait %.1f store %.1f zero-state %.1f\n",
             (double)cyc[1] / n_iv, (double)cyc[5] / n_iv, (double)cyc[2] / n_iv, (double)cyc[3] / n_iv);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (wave == 3) {
    // ------------------------------ CHAIN ------------------------------
    const double m11 = p.power[0 * p.channels + c], m12 = p.power[1 * p.channels + c];
    const double m21 = p.power[2 * p.channels + c], m22 = p.power[3 * p.channels + c];
    // S: the true state at the start of this workgroup's chunks, advanced in zero-state space --
    // S_c = M ( ... M (M S_{c-W} + z_{c-W}) + z_{c-W+1} ... ) + z_{c-1} -- from the states every workgroup publishes;
    // the first chunk starts the chain from the bank's state
    double S1 = (p.na > 1) ? p.yh[0 * p.channels + c] : 0.0;
    double S2 = (p.na > 2) ? p.yh[1 * p.channels + c] : 0.0;
    asm volatile("" : "+v"(S1), "+v"(S2));
    publish(flags + F_INIT, 1, lane);                          // (the state words are in registers)
    // Round 6: what the chain produced is CHECKED at every chunk boundary.  The replay of chunk j ends in the true state
    // (the DF-I statement sample by sample from chunk j's start state); the workgroup of chunk j + 1 started from the state
    // its own chain gave it, M S + z over up to W chunks -- the same number within the mode's error (1e-10 .. 1e-7 of the
    // signal's scale) unless some z, some start state or some replay was wrong.  A mismatch sets W_CHECK_FAILED: the
    // process call then runs the block again in three launches (ALZ_LOOK_CHECK_CALL) or the next call raises.
    double smax = 0.0;
    auto check_boundary = [&](int vs) {
      const int64_t cv = (int64_t)w + (int64_t)vs * W;         // the chunk whose end is checked
      if (cv + 1 >= K || ALZ_DBG(p, 32 | 2)) return;           // (the block's last chunk ends in the bank's state: no successor)
      await(flags + F_REPLAYED, NT * (vs + 1), cap, p.err, W_CHAIN_CHECK_WAIT);
      unsigned long long v1 = 0, v2 = 0;
      int tries = 0;
      while (true) {
        if (lane < 32) dma16_coherent(sg + (cv + 1) * kPub + 2 * lane, lds0 + (unsigned)(vlds - smem));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long *vl = reinterpret_cast<const unsigned long long *>(vlds) + cl;
        v1 = vl[0];
        v2 = vl[32];
        const bool missing = (v1 ^ vl[16]) != key || (v2 ^ vl[48]) != key;
        if (__builtin_amdgcn_ballot_w64(missing) == 0) break;
        if (++tries > ALZ_LOOK_CAP(p)) { p.err[W_CHAIN_CHECK_WAIT] = 1; return; }
        __builtin_amdgcn_s_sleep(16);
      }
      const double n1 = __longlong_as_double((long long)v1), n2 = __longlong_as_double((long long)v2);
      const double e1 = *reinterpret_cast<const double *>(ebuf + (vs % 3) * 256 + cl * 8);
      const double e2 = *reinterpret_cast<const double *>(ebuf + (vs % 3) * 256 + 128 + cl * 8);
      const double d = __builtin_fabs(e1 - n1) + __builtin_fabs(e2 - n2);
      const double sc = __builtin_fabs(e1) + __builtin_fabs(e2) + __builtin_fabs(n1) + __builtin_fabs(n2);
      smax = sc > smax ? sc : smax;                            // (the scale of this channel's states so far: a boundary in near-silence is not held to its own size)
      if (d > 1e-5 * smax) p.err[W_CHECK_FAILED] = 1;          // (a NaN -- a NaN in the input -- is not a mismatch)
    };
    for (int seq = 0; seq < my_chunks; ++seq) {
      if ((seq > 0 || w > 0) && !ALZ_DBG(p, 32)) {
        const int64_t cj = (int64_t)w + (int64_t)seq * W;
        const int64_t req_first = seq > 0 ? cj - W : 0;      // z_first .. z_{cj-1}
        const int req_cnt = (int)(cj - req_first);
        // all of them at once, as eight 1 KiB global -> LDS transfers, again until every requested pair matches the key
        // (this wave has nothing else to do; the last of them appears when the neighbour's HELP wave has summed its chunk)
        unsigned long long a1[kMaxW], a2[kMaxW];
        int tries = 0;
        while (true) {
#pragma unroll
          for (int o = 0; o < kMaxW / 2; ++o) {              // lane l of transfer o: chunk 2 o + l / 32, 16-byte piece l % 32
            int64_t ch = req_first + 2 * o + (lane >> 5);
            if (ch > K - 1) ch = K - 1;                      // (beyond the request: any valid address, never read)
            dma16_coherent(zg + ch * kPub + 2 * (lane & 31), lds0 + (unsigned)(zlds - smem) + o * 1024);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          bool missing = false;
#pragma unroll
          for (int e = 0; e < kMaxW; ++e) {
            const unsigned long long *zl = reinterpret_cast<const unsigned long long *>(zlds) + e * kPub + cl;
            a1[e] = zl[0];
            a2[e] = zl[32];
            missing |= e < req_cnt && ((a1[e] ^ zl[16]) != key || (a2[e] ^ zl[48]) != key);
          }
          if (__builtin_amdgcn_ballot_w64(missing) == 0) break;
          if (++tries > ALZ_LOOK_CAP(p)) {                   // (cannot happen: the states come from earlier chunks)
            p.err[W_CHAIN_Z] = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(16);
        }
#pragma unroll
        for (int e = 0; e < kMaxW; ++e) {
          if (e < req_cnt) {
            const double z1 = __longlong_as_double((long long)a1[e]), z2 = __longlong_as_double((long long)a2[e]);
            const double n1 = __builtin_fma(m11, S1, __builtin_fma(m12, S2, z1));
            const double n2 = __builtin_fma(m21, S1, __builtin_fma(m22, S2, z2));
            S1 = n1; S2 = n2;
          }
        }
      }
#if defined(ALZ_ABLATE)
      if (ALZ_DBG(p, 4096) && group == 1 && seq == 1 && w == 2) S1 += 0.125;   // (a wrong start state, for the demonstration of the boundary check)
#endif
      // the start state goes out like the z (chunk 0 starts from the bank's state: nothing to compare it with)
      if (q == 0 && (seq > 0 || w > 0)) publish_pair(sg + ((int64_t)w + (int64_t)seq * W) * kPub, S1, S2);
      // the buffer of this parity held the state of chunk number seq - 2: the replay has read it
      if (seq >= 2) await(flags + F_REPLAYED, NT * (seq - 2) + 1, cap, p.err, W_CHAIN_REPLAYED);
      if (q == 0) {
        *reinterpret_cast<double *>(sbuf + (seq & 1) * 256 + cl * 8) = S1;
        *reinterpret_cast<double *>(sbuf + (seq & 1) * 256 + 128 + cl * 8) = S2;
      }
      // the replay overwrites the chunk's tiles and reads two rows groups into the next one: HELP must have read the
      // former, LOAD prepared the latter
      {
        const int zneed = NT * seq + NT, pneed = NT * seq + NT + 2;
        await(flags + F_ZDONE, zneed > TOT ? TOT : zneed, cap, p.err, W_CHAIN_ZDONE);
        await(flags + F_PREPARED, pneed > TOT ? TOT : pneed, cap, p.err, W_CHAIN_PREPARED);
      }
      publish(flags + F_CHUNK, seq + 1, lane);
      // the replay is past chunk seq - 2 by now (the tiles this round waited for are prepared at most two chunks ahead of
      // it): that chunk's true end state against the start state its successor was given
      if (seq >= 2) check_boundary(seq - 2);
    }
    for (int vs = my_chunks >= 2 ? my_chunks - 2 : 0; vs < my_chunks; ++vs) check_boundary(vs);
  } else {
    // ------------------------------ REPLAY ------------------------------
    asm volatile("" : "+v"(na1), "+v"(na2));
    LookState st = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double pr[4][8];
    ALZ_LOOK_CLOCK(3)                                          // tile / tail / chunk wait
    await(flags + F_CHUNK, 1, cap, p.err, W_REPLAY_CHUNK);
    {
      const char *cur = smem + lane_off - q * kStep;         // tile 0: groups 0 and 1 of the register ring
#pragma unroll
      for (int u = 0; u < 8; ++u) pr[0][u] = *reinterpret_cast<const double *>(cur + ((u < 3 && u < q) ? q : u) * kStep);
#pragma unroll
      for (int u = 0; u < 8; ++u) pr[1][u] = *reinterpret_cast<const double *>(cur + (8 + u) * kStep);
    }
    int sr = 0, tic = 0, seq = 0;                            // slot, tile of the chunk, chunk number: counted by hand
    ALZ_LOOK_MARK(2)
    for (int t = 0; t < TOT; ++t) {
      char *cur = smem + sr * kSlot + lane_off - q * kStep;
      const int sn = (sr + 1 == kSlots) ? 0 : sr + 1;
      const char *nxt = (t + 1 < TOT) ? smem + sn * kSlot + lane_off - q * kStep : cur;
      if (tic == 0) {
        if (t > 0) await(flags + F_CHUNK, seq + 1, cap, p.err, W_REPLAY_CHUNK);
        ALZ_LOOK_MARK(2)
        if (!ALZ_DBG(p, 2)) {
          const double s1 = *reinterpret_cast<const double *>(sbuf + (seq & 1) * 256 + cl * 8);
          const double s2 = *reinterpret_cast<const double *>(sbuf + (seq & 1) * 256 + 128 + cl * 8);
          look_tile<PA, true, kStep>(cur, nxt, q, s1, s2, na1, na2, st, pr);
        }
      } else if (!ALZ_DBG(p, 2)) {
        look_tile<PA, false, kStep>(cur, nxt, q, 0.0, 0.0, na1, na2, st, pr);
      }
      ALZ_LOOK_MARK(0)
      if (tic == NT - 1 && q == 0) {                         // copy 0's state after a chunk's last tile: the true end state (CHAIN's check)
        *reinterpret_cast<double *>(ebuf + (seq % 3) * 256 + cl * 8) = st.m1;
        *reinterpret_cast<double *>(ebuf + (seq % 3) * 256 + 128 + cl * 8) = st.m2;
      }
      publish(flags + F_REPLAYED, t + 1, lane);              // (behind the tile's writes in the LDS's order)
      // copy 0 has just finished the tile: after the last one its (m1, m2) is the bank's state after the block
      if (t == TOT - 1 && q == 0 && ((K - 1) % W) == w) {
        if (p.na > 1) p.yh[0 * p.channels + c] = st.m1;
        if (p.na > 2) p.yh[1 * p.channels + c] = st.m2;
      }
      sr = sn;
      tic = (tic + 1 == NT) ? 0 : tic + 1;
      seq += (tic == 0);
      ALZ_LOOK_MARK(1)
    }
#if defined(ALZ_ABLATE) && defined(ALZ_LOOK_TIMING)
    if (blockIdx.x == 5 && lane == 0 && (p.dbg & 256))
      printf("k_look replay wave, cycles per tile: tile %.1f tail %.1f chunk wait %.1f (%d tiles)\n",
             (double)cyc[0] / TOT, (double)cyc[1] / TOT, (double)cyc[2] / TOT, TOT);
#endif
  }
#undef ALZ_EOFF
}
#undef ALZ_LOOK_CLOCK
#undef ALZ_LOOK_MARK

const char *look_wait_name(int site) {
  static const char *const names[W_COUNT] = {"?", "LOAD/stored", "HELP/prepared", "HELP/replayed", "HELP/state-read", "CHAIN/published-states",
                                             "CHAIN/replayed", "CHAIN/summed", "CHAIN/prepared", "REPLAY/start-state", "LOAD/all-stored", "CHAIN/neighbour-start-state",
                                             "CHUNK-BOUNDARY-CHECK-FAILED"};
  return site >= 0 && site < W_COUNT ? names[site] : "?";
}

typedef void (*look_fn)(LArgs);
template <bool CM, int PRE>
static look_fn pick_look_in(unsigned pb, unsigned pa) {
#define ALZ_PAT(PB_, PA_) if (pb == PB_ && pa == PA_) return (look_fn)k_look<PB_, PA_, CM, PRE>;
  ALZ_PAT(1, 1) ALZ_PAT(3, 1) ALZ_PAT(1, 3) ALZ_PAT(3, 3) ALZ_PAT(5, 3) ALZ_PAT(7, 3) ALZ_PAT(1, 2)
#undef ALZ_PAT
  return nullptr;
}
static look_fn pick_look(unsigned pb, unsigned pa, bool cm, int pre) {
  if (pre == 0) return cm ? pick_look_in<true, 0>(pb, pa) : pick_look_in<false, 0>(pb, pa);
  if (pre == ALZ_MAP_ABS) return cm ? pick_look_in<true, 1>(pb, pa) : pick_look_in<false, 1>(pb, pa);
  return nullptr;
}

// in place: the two input rows in front of every chunk, saved before any workgroup overwrites them (LArgs::hsave)
__global__ __launch_bounds__(256) void k_look_hsave(const double *x, int64_t ldx, int cm, int64_t groups, int64_t K, double *hsave) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups * K * 32) return;
  const int64_t gk = i >> 5, g = gk / K, j = gk - g * K;
  const int e = (int)(i & 31);
  if (j == 0) return;                                   // (chunk 0 starts from the bank's own history)
  const int r = cm ? (e & 1) : (e >> 4), ch = cm ? (e >> 1) : (e & 15);
  const int64_t t = j * kLookChunk - 2 + r, c = g * 16 + ch;
  hsave[i] = cm ? x[c * ldx + t] : x[t * ldx + c];
}

// The shape test of the one-pass form, without side effects: what launch_look itself checks before it touches anything
// (the caller uses it to decide whether an input map may ride on the kernel's reads).
bool look_takes(const SectionDev &sec, const BlockIO &io, int cus) {
  if (!(sec.nb <= 3 && sec.na <= 3 && sec.na >= 2 && sec.uniform) || sec.any_div || io.fused) return false;
  const bool cm = io.sxn == 1 && io.syn == 1, tm = io.sxc == 1 && io.syc == 1;
  if (!(cm || tm) || io.map_input) return false;
  const int64_t C = io.channels, L = kNT * 64;
  if (C % 16 || io.c_first != 0 || io.c_count != C) return false;
  if (io.x == io.y && (io.sxn != io.syn || io.sxc != io.syc)) return false;
  const int64_t ldx = (cm && !tm) ? io.sxc : io.sxn, ldy = (cm && !tm) ? io.syc : io.syn;
  if ((((uintptr_t)io.x | (uintptr_t)io.y) & 15) || ((ldx | ldy) & 1)) return false;
  const int64_t K = io.n / L, groups = C / 16;
  int64_t W = groups > 0 ? cus / groups : 0;
  if (W > K) W = K;
  if (W > kMaxW) W = kMaxW;
  if (W < 2 || K < 4) return false;
  return pick_look(sec.present_b, sec.present_a, cm && !tm, io.pre_op) != nullptr;
}

uint64_t look_scratch_bytes(int64_t groups, int64_t chunks) { return (uint64_t)groups * chunks * (2 * kPub + 32) * sizeof(double); }

// One-pass time-parallel run of a biquad-class section over whole 512-sample chunks of a block (either layout, also in
// place, optionally with |x| on the input reads).  `power` = the section's M = A^512 per channel ([4][channels],
// alz_scan.hip); `zbuf` (look_scratch_bytes(groups, chunks): the published end states, the published start states, then
// the saved history rows of an in-place run; never cleared) and `err` are scratch of the handle.  *done_samples: the whole chunks covered (0: not this kernel's shape).
int launch_look(const SectionDev &sec, const BlockIO &io, hipStream_t stream, const double *power, double *zbuf,
                uint64_t zbuf_bytes, int *err, int64_t *done_samples, const char **kernel_name) {
  *done_samples = 0;
  // every workgroup of the launch must be resident (one per CU: 137 KiB of LDS each)
  int dev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&dev));
  // device attributes and the kernel's occupancy are constants of (device, kernel): asked once (round-4 advisor)
  struct DevInfo { int cus = 0, coop = -1; };
  static std::mutex mu;
  static std::map<int, DevInfo> devs;
  static std::map<std::pair<int, const void *>, int> per_cu_of;
  DevInfo di;
  {
    std::lock_guard<std::mutex> lock(mu);
    DevInfo &d = devs[dev];
    if (d.coop < 0) {
      ALZ_HIP_CHECK(hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev));
      ALZ_HIP_CHECK(hipDeviceGetAttribute(&d.coop, hipDeviceAttributeCooperativeLaunch, dev));
    }
    di = d;
  }
  const int cus = di.cus;
  if (!look_takes(sec, io, cus)) return ALZ_OK;
  const bool cm = io.sxn == 1 && io.syn == 1 && !(io.sxc == 1 && io.syc == 1);
  const int64_t C = io.channels, L = kNT * 64;
  const int64_t K = io.n / L, groups = C / 16;
  int W = (int)(cus / groups);
  if (W > K) W = (int)K;
  if (W > kMaxW) W = kMaxW;
  const bool inplace = io.x == io.y;
  const uint64_t zwords = (uint64_t)groups * K * kPub;      // published end states, then as many for the start states, then the
  if (look_scratch_bytes(groups, K) > zbuf_bytes) return ALZ_OK;   // saved history rows of an in-place run
  look_fn fn = pick_look(sec.present_b, sec.present_a, cm, io.pre_op);
  if (!fn) return ALZ_OK;
  LArgs p;
  p.x = io.x; p.y = io.y; p.ldx = cm ? io.sxc : io.sxn; p.ldy = cm ? io.syc : io.syn; p.n_chunks = K; p.channels = C;
  p.n_inputs = io.mode == ALZ_BANK_OUTER ? io.n_inputs : 0; p.n_sets = io.n_sets; p.workers = W;
  p.nb = sec.nb; p.na = sec.na; p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.yh = sec.yh;
  p.power = power; p.z = (unsigned long long *)zbuf; p.err = err;
  p.s = (unsigned long long *)zbuf + zwords;
  p.hsave = inplace ? zbuf + 2 * zwords : nullptr;
  {
    // a key no other launch of this process has had (the scratch may hold any earlier launch's words, of this bank or,
    // after a free and a malloc, of another): splitmix64 of a process-wide counter, never zero
    static std::atomic<unsigned long long> counter{0x9E3779B97F4A7C15ull};
    unsigned long long z = counter.fetch_add(0x9E3779B97F4A7C15ull, std::memory_order_relaxed);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    p.key = z ? z : 0x5851F42D4C957F2Dull;
  }
  static const int dbg_env = ALZ_DBG_ENV();
  p.dbg = dbg_env;
  const size_t slot = cm ? (size_t)kSlotCM : (size_t)kSlot;
  const size_t lds = (size_t)kSlots * slot + (size_t)kSlots * kHist + (size_t)kMaxW * kPub * 8 + 512 + 768 + kPub * 8 + 64;
  const int rc = ensure_dynamic_lds((const void *)fn, (int)lds);
  if (rc) return rc;
  // Every workgroup of the launch waits on others: they must all be resident at once.  The runtime is asked to
  // guarantee exactly that (a cooperative launch: it refuses a grid that the device cannot hold -- occupancy of this
  // kernel with its 137 KiB of LDS times the CUs visible to this process, CU masks included); where it refuses, or
  // the device has no cooperative launches, the caller runs the three-launch form instead.  What no launch API can
  // promise is that OTHER work leaves the CUs free in time: a workgroup that starts late only delays its neighbours,
  // and one that starts later than the spin cap allows makes the kernel give up -- which every entry point of the
  // handle reports (alz_api.hip take_look_error), never a silently bad block.
  if (!di.coop) return ALZ_OK;
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_cu_of.find({dev, (const void *)fn});
    if (it == per_cu_of.end()) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, 256, lds) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 0;
      }
      per_cu_of[{dev, (const void *)fn}] = per_cu;
    } else {
      per_cu = it->second;
    }
  }
  if ((int64_t)per_cu * cus < groups * W) return ALZ_OK;     // (not co-resident: the three-launch form takes the block)
  if (inplace)
    hipLaunchKernelGGL(k_look_hsave, dim3((unsigned)((groups * K * 32 + 255) / 256)), dim3(256), 0, stream, io.x, p.ldx, cm ? 1 : 0, groups, K,
                       zbuf + 2 * zwords);
  void *args[] = {(void *)&p};
  if (ALZ_TUNE("ALZ_LOOK_COOP", 1) == 0) {       // (tuning builds: the plain launch of round 3, for A/B timing)
    hipLaunchKernelGGL(fn, dim3((unsigned)(groups * W)), dim3(256), lds, stream, p);
    ALZ_HIP_CHECK(hipGetLastError());
    *done_samples = K * L;
    *kernel_name = "k_look(plain launch)";
    return ALZ_OK;
  }
  const hipError_t le = hipLaunchCooperativeKernel((const void *)fn, dim3((unsigned)(groups * W)), dim3(256), args, (unsigned)lds, stream);
  if (le == hipErrorCooperativeLaunchTooLarge || le == hipErrorNotSupported || le == hipErrorLaunchOutOfResources) {
    (void)hipGetLastError();                                     // not co-resident here: the three-launch form takes the block
    return ALZ_OK;
  }
  if (le != hipSuccess) return fail(ALZ_E_HIP, std::string("k_look cooperative launch: ") + hipGetErrorString(le));
  *done_samples = K * L;
  *kernel_name = "k_look";
  return ALZ_OK;
}

}  // namespace alz
