// alz_wave.hip -- the streaming DF-I kernel: independent wavefront workers, LDS ring
// fed by direct global->LDS DMA, wide coalesced stores.
//
// Same arithmetic as k_small in alz_iir.hip (reference audiolazy/lazy_filters.py
// :197-257, bit-exact, -ffp-contract=off); what changes is how samples move:
//
//   * a wavefront owns G adjacent channels (lanes 0..G-1 run the recurrences, filter
//     state in VGPRs) and walks time in tiles of 8 KiB = 512 16-byte pieces;
//   * tiles arrive with global_load_lds_dwordx4 (no VGPR round trip, 64 x 16 B per
//     instruction, 8 instructions per tile) into a ring of R = 4 LDS slots, three
//     tiles (24 KiB) in flight per wave under counted s_waitcnt vmcnt(N);
//   * the recurrence reads x from the slot with ds_read_b64, writes y back in place,
//     and the finished tile leaves as 8 ds_read_b128 + global_store_dwordx4;
//   * no workgroup barrier anywhere: one wave per workgroup, waves never talk.
//
// Why G matters: the recurrence is a serial dependence per channel (about 3 dependent
// f64 ops, ~7.5 cycles each, and 4.7 cycles of issue per f64 op -- measured with
// tools/ubench_f64.hip), so a lane advances one sample per ~30 cycles no matter how
// many lanes are active.  Throughput is channels x step rate as long as every wave has
// its own SIMD; small banks therefore use G = 16 so that 4096 channels become 256
// waves, one per CU, each pulling only ~13 B/clk through its CU's memory path.
//
// Layouts (element (n, c) of the block):
//   TIME_MAJOR  tile = [T rows][G ch], piece q -> row q/(G/2), channel pair q%(G/2)
//   CHAN_MAJOR  tile = [G ch][T samples], piece q -> channel q/(T/2), sample pair q%(T/2);
//               each 1 KiB DMA chunk is padded by 16 B in LDS to spread banks.
// T = 512/G*... = 8192 / (8*G) samples per tile.
#include <stdlib.h>

#include "alz_common.h"

namespace alz {

static constexpr int kRing = 4;           // LDS slots per wave
static constexpr int kChunks = 8;         // 1 KiB DMA instructions per tile
static constexpr int kSlotBytes = 8192 + kChunks * 16;

#ifndef ALZ_PACE_ALL
#define ALZ_PACE_ALL 0   // (variant builds: the paced feed-forward pass for every tap pattern)
#endif
struct WArgs {
  const double *x;
  double *y;
  int64_t ldx, ldy;       // leading dimensions in elements
  int64_t n_tiles;        // full tiles to process
  int64_t channels;       // state array stride
  int64_t c_first;        // first channel handled by this launch
  int64_t n_inputs;       // OUTER banks: channel = set * n_inputs + input (0: diagonal bank)
  int map_input;          // OUTER, first section: x is indexed by input, not by channel
  int64_t n_sets;
  int nb, na;
  const double *b, *a;
  double *xh, *yh;
  int dbg;  // ALZ_WAVE_DEBUG ablation bits: 1 no DMA, 2 no recurrence, 4 no stores, 8 no tile barriers (wrong output!)
  // time-parallel mode (alz_scan.hip): blockIdx.y = chunk j of the time axis.  Chunk j reads / writes
  // at element offset j * chunk_x / j * chunk_y and keeps its state in the slot of "virtual channel"
  // j * chunk_c + c of the state arrays (whose stride `channels` is then n_chunks * chunk_c).
  // All three are 0 in an ordinary launch (gridDim.y == 1).
  int64_t chunk_x, chunk_y, chunk_c;
  int aux_pace;   // k_duo, one-pole banks that fill the chip: pauses (x 64 cycles) between the quarters of AUX's feed-forward pass
  int stagger;    // k_duo, channel-major: workgroup g starts g * stagger ticks (10 ns) late (0: all together) -- see launch_wave
  int tile_pace;  // k_duo: the helper wave requests tile i + 3 no earlier than tile_pace / 16 ticks (10 ns) x i after its start (0: free-running)
  unsigned *convoy;      // (experiment, tuning builds) the counter ring of alz_common.h convoy_sync, nullptr: none
  int convoy_cfg;
  unsigned convoy_groups;
};

// one 1 KiB DMA chunk: every lane supplies its own 16-byte global source, the data lands
// at lds_dst + lane*16.  M0 is saved/restored inside the same statement (hipcc reserves it).
template <bool NT = false>
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
  } else {
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
}

// one 1 KiB store, 16 B per lane.  Inline asm on purpose: a store hipcc knows about makes it
// insert its own s_waitcnt vmcnt(7) in front of the next tile's stores, which (with the DMA
// loads it cannot see in the queue) drains the whole prefetch ring every tile.  The trailing
// s_nop covers the "VMEM store of more than 8 bytes, then overwrite of its data VGPRs" hazard.
// NT: non-temporal policy for blocks that are read once and not read back by the same call -- with the STORE wave
// at its side configs[1] gains 3 - 5 % from both hints together (profiles/NOTES_r03.md 11); cascades and small
// blocks, whose next section or next pass finds the data in the Infinity Cache, keep the default.  A COMPILE-TIME
// choice: as a wave-uniform run-time branch around the two statements it cost the two-wave instantiations 25 %
// (the tile loop was no longer one basic block).
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <bool NT = false>
__device__ __forceinline__ void store16(double *gdst, dbl2 v) {
  if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
}

// wait until at most `n` vector-memory operations of this wave are outstanding.
// n is a multiple of 8; anything above 56 waits for 56 (vmcnt is a 6-bit counter). s_waitcnt needs a literal.
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
  }
}

// PRE: an elementwise stage fused into the section's input (alz_bank_set_input_map): 0 none, 1 abs --
// ``lowpass(cutoff)(abs(sig))`` of envelope.abs (lazy_analysis.py:468-493).  abs is exact and, on the
// operand of a v_mul_f64, a source modifier: the fused form costs no instruction.
template <int PRE>
__device__ __forceinline__ double pre_in(double v) {
  if constexpr (PRE == 1) return __builtin_fabs(v);
  return v;
}

template <unsigned PB, unsigned PA>
__device__ __forceinline__ double wave_step(double d0, double d1, double d2, double m1, double m2,
                                            double b0, double b1, double b2, double na1,
                                            double na2) {
  double acc = 0.0;
  bool first = true;
  if constexpr (PB & 1u) { acc = b0 * d0; first = false; }
  if constexpr (PB & 2u) { const double t = b1 * d1; acc = first ? t : acc + t; first = false; }
  if constexpr (PB & 4u) { const double t = b2 * d2; acc = first ? t : acc + t; first = false; }
  if constexpr (PA & 1u) { const double t = na1 * m1; acc = first ? t : acc + t; first = false; }
  if constexpr (PA & 2u) { const double t = na2 * m2; acc = first ? t : acc + t; first = false; }
  return acc;
}

// ALZ_WAVE_NT (variant builds, A/B only): the single-wave kernel's tile DMA and stores with the non-temporal policy
#ifndef ALZ_WAVE_NT
#define ALZ_WAVE_NT 0
#endif
// G: channels per wave (16, 32 or 64).  CM: channel-major layout.
// NOSTORE: pass 1 of the time-parallel mode -- the recurrence runs for its end state only; no output
// tile and no input history is written.
template <int G, bool CM, unsigned PB, unsigned PA, bool NOSTORE = false, int PRE = 0>
__global__ __launch_bounds__(64) void k_wave(WArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int T = 8192 / (8 * G);          // samples per channel per tile
  const int lane = threadIdx.x;
  const int64_t c0 = p.c_first + (int64_t)blockIdx.x * G;
  // OUTER bank (n_inputs % G == 0): the G channels of a wave are G adjacent inputs of one set
  const int64_t in0 = (p.n_inputs && p.map_input) ? c0 % p.n_inputs : c0;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;  // low 32 bits of the flat address == LDS offset

  // per-lane source/destination offsets (in elements) of the lane's piece in chunk 0 of tile 0,
  // and the element step from one chunk to the next / one tile to the next
  int64_t x_off, y_off, x_chunk, y_chunk, x_tile, y_tile;
  if (!CM) {
    // chunk j holds rows j*R8 .. j*R8+R8-1 with R8 = 128/G rows, each row G ch = G/2 pieces
    constexpr int PPR = G / 2;               // pieces per row
    const int row = lane / PPR, cp = lane % PPR;
    x_off = (int64_t)row * p.ldx + in0 + 2 * cp;
    y_off = (int64_t)row * p.ldy + c0 + 2 * cp;
    x_chunk = (int64_t)(64 / PPR) * p.ldx;
    y_chunk = (int64_t)(64 / PPR) * p.ldy;
    x_tile = (int64_t)T * p.ldx;
    y_tile = (int64_t)T * p.ldy;
  } else {
    // chunk j holds channels j*C8 .. with C8 = 128/T channels, each channel T samples = T/2 pieces
    constexpr int PPC = T / 2;               // pieces per channel
    const int ch = lane / PPC, sp = lane % PPC;
    x_off = (in0 + ch) * p.ldx + 2 * sp;
    y_off = (c0 + ch) * p.ldy + 2 * sp;
    x_chunk = (int64_t)(64 / PPC) * p.ldx;
    y_chunk = (int64_t)(64 / PPC) * p.ldy;
    x_tile = T;
    y_tile = T;
  }

  // Every lane runs the recurrence: with G < 64 the lanes G..63 are "ghosts" that mirror the
  // channel of lane & (G-1).  A partially masked wave issues f64 ops ~36 % slower on gfx950
  // (tools/ubench_rows.hip: 56 vs 41 cycles/step), so nothing here is exec-masked; ghosts read
  // the same LDS words as their real lane (broadcast) and compute the same y.
  const int cl = lane & (G - 1);
  const bool real = lane < G;
  const int64_t c = c0 + cl;
  const int64_t sc = (int64_t)blockIdx.y * p.chunk_c + c;     // state slot (== c unless time-parallel)
  const int64_t set = p.n_inputs ? c / p.n_inputs : ((p.n_sets == 1) ? 0 : c);
  double b0 = 0, b1 = 0, b2 = 0, na1 = 0, na2 = 0;
  if (PB & 1u) b0 = p.b[0 * p.n_sets + set];
  if (PB & 2u) b1 = p.b[1 * p.n_sets + set];
  if (PB & 4u) b2 = p.b[2 * p.n_sets + set];
  if (PA & 1u) na1 = -p.a[1 * p.n_sets + set];
  if (PA & 2u) na2 = -p.a[2 * p.n_sets + set];
  double d1 = (p.nb > 1) ? p.xh[0 * p.channels + sc] : 0.0;
  double d2 = (p.nb > 2) ? p.xh[1 * p.channels + sc] : 0.0;
  double m1 = (p.na > 1) ? p.yh[0 * p.channels + sc] : 0.0;
  double m2 = (p.na > 2) ? p.yh[1 * p.channels + sc] : 0.0;

  // Consume the coefficient/state loads here, before any DMA is queued: hipcc then waits for
  // them now, its own vmcnt scoreboard is empty for the rest of the kernel, and it places no
  // s_waitcnt vmcnt(0) of its own inside the tile loop (where it would drain the DMA ring).
  asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(na1), "+v"(na2));
  asm volatile("" : "+v"(d1), "+v"(d2), "+v"(m1), "+v"(m2));

  const double *xg = p.x + x_off + (int64_t)blockIdx.y * p.chunk_x;
  double *yg = p.y + y_off + (int64_t)blockIdx.y * p.chunk_y;
  const int64_t nt = p.n_tiles;

  // prologue: tiles 0 .. kRing-2 into slots 0 .. kRing-2
  for (int t = 0; t < kRing - 1 && t < nt && !ALZ_DBG(p, 1); ++t) {
#pragma unroll
    for (int j = 0; j < kChunks; ++j)
      dma16<ALZ_WAVE_NT != 0>(xg + t * x_tile + j * x_chunk, lds0 + t * kSlotBytes + j * 1040);
  }

  const long long pace0 = p.tile_pace > 0 ? (long long)wall_clock64() : 0;

  long long pace_shift = 0;
  for (int64_t i = 0; i < nt; ++i) {
    const int slot = (int)(i % kRing);
    if (p.tile_pace > 0) pace_wait(pace0, i, p.tile_pace, pace_shift);   // (tuning builds: the common tile clock, alz_common.h)
    // refill the slot freed by tile i-1 with tile i+kRing-1
    const int64_t tn = i + kRing - 1;
    if (tn < nt && !ALZ_DBG(p, 1)) {
      const int sn = (int)(tn % kRing);
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        dma16<ALZ_WAVE_NT != 0>(xg + tn * x_tile + j * x_chunk, lds0 + sn * kSlotBytes + j * 1040);
    }
    // operations issued after tile i's DMA: the DMA of the following tiles plus the
    // stores of the preceding ones (completion is in issue order)
    {
      const int64_t loads_after = (nt - 1 - i < kRing - 1) ? (nt - 1 - i) : (kRing - 1);
      const int64_t stores_after = NOSTORE ? 0 : ((i < kRing - 1) ? i : (kRing - 1));
      wait_vm((int)(loads_after + stores_after) * kChunks);
    }

    char *tile = smem + slot * kSlotBytes;
    if (!ALZ_DBG(p, 2)) {
      // element e(u) = u*G + cl (TIME) or cl*T + u (CHAN); byte = e*8 + (e/128)*16 (chunk pad)
      // (T and G divide 128, so the pad term splits into a per-lane part and a per-u part)
      const int lane_off = CM ? cl * T * 8 + ((cl * T) >> 7) * 16 : cl * 8;
      const char *rd = tile + lane_off;
      // write address: lane group r = lane / G owns row r of each group of 64/G rows
      char *wr = tile + lane_off + (CM ? (lane / G) * 8 : (lane / G) * G * 8);
#define ALZ_EOFF(u) (CM ? (u) * 8 : (u) * G * 8 + (((u) * G) >> 7) * 16)
      // Three-stage software pipeline over chunks of 8 samples, pinned with sched_barrier:
      //   LDS reads of chunk k+2  |  feed-forward sums p[] of chunk k+1  |  recurrence of chunk k
      // The feed-forward ops (independent of the recurrence) are slotted between the dependent
      // ops t3 -> s1 -> y of the recurrence, whose ~7.5-cycle result latency would otherwise
      // stall the in-order wave (29 instead of 41 cycles/step, tools/ubench_biquad.hip).
      constexpr int NCH = T / 8;
      double xr[3][8], pp[2][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xr[0][u] = pre_in<PRE>(*reinterpret_cast<const double *>(rd + ALZ_EOFF(u)));
      if (NCH > 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) xr[1][u] = pre_in<PRE>(*reinterpret_cast<const double *>(rd + ALZ_EOFF(8 + u)));
      }
      // feed-forward of chunk 0 (not overlapped: once per tile)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double x0 = xr[0][u];
        const double x1 = u >= 1 ? xr[0][u - 1] : d1;
        const double x2 = u >= 2 ? xr[0][u - 2] : (u == 1 ? d1 : d2);
        double acc = 0.0;
        bool first = true;
        if constexpr (PB & 1u) { acc = b0 * x0; first = false; }
        if constexpr (PB & 2u) { const double t = b1 * x1; acc = first ? t : acc + t; first = false; }
        if constexpr (PB & 4u) { const double t = b2 * x2; acc = first ? t : acc + t; first = false; }
        pp[0][u] = acc;
      }
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int xc = k % 3, xn = (k + 1) % 3, xl = (k + 2) % 3;
        if (k + 2 < NCH) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            xr[xl][u] = pre_in<PRE>(*reinterpret_cast<const double *>(rd + ALZ_EOFF((k + 2) * 8 + u)));
        }
        __builtin_amdgcn_sched_barrier(0);
        double yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          // feed-forward of step u of chunk k+1: independent of the recurrence below, so hipcc
          // slots these ops into the stalls of the dependent chain t3 -> s1 -> y (hand-pinning
          // the order was measured slower than its own schedule: tools/ubench_loop.hip)
          if (k + 1 < NCH) {
            const double x0 = xr[xn][u];
            const double x1 = u >= 1 ? xr[xn][u - 1] : xr[xc][7];
            const double x2 = u >= 2 ? xr[xn][u - 2] : xr[xc][6 + u];
            double pn = 0.0;
            bool first = true;
            if constexpr (PB & 1u) { pn = b0 * x0; first = false; }
            if constexpr (PB & 2u) { const double t = b1 * x1; pn = first ? t : pn + t; first = false; }
            if constexpr (PB & 4u) { const double t = b2 * x2; pn = first ? t : pn + t; first = false; }
            pp[(k + 1) & 1][u] = pn;
          }
          double acc = pp[k & 1][u];
          if constexpr (PA & 1u) acc = acc + na1 * m1;
          if constexpr (PA & 2u) acc = acc + na2 * m2;
          yv[u] = acc;
          m2 = m1; m1 = acc;
          // Leaving through LDS: a 64-lane ds_write_b64 costs the wave ~18 cycles, so one per step
          // would be a third of the loop.  The ghost lanes hold the SAME y as their real lane at
          // every step, so lane group r = lane / G keeps the y of step (u0 + r) and a single
          // ds_write_b64 stores 64 / G consecutive rows of the tile.
          constexpr int RPW = 64 / G;  // rows per write
          if (!NOSTORE && (u + 1) % RPW == 0) {
            double yw = yv[u - (RPW - 1)];
#pragma unroll
            for (int r = 1; r < RPW; ++r) yw = (lane / G == r) ? yv[u - (RPW - 1) + r] : yw;
            *reinterpret_cast<double *>(wr + ALZ_EOFF(k * 8 + u - (RPW - 1))) = yw;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      d1 = xr[(NCH - 1) % 3][7];
      d2 = xr[(NCH - 1) % 3][6];
#undef ALZ_EOFF
    }
    // the finished tile leaves as eight 1 KiB stores (all 64 lanes, 16 B each): all eight LDS
    // reads are issued back to back (one exposed LDS latency per tile instead of eight)
    double *yt = yg + i * y_tile;
    if (!NOSTORE && !ALZ_DBG(p, 4)) {
      dbl2 v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        v[j] = *reinterpret_cast<const dbl2 *>(tile + j * 1040 + lane * 16);
#pragma unroll
      for (int j = 0; j < kChunks; ++j) store16<ALZ_WAVE_NT != 0>(yt + j * y_chunk, v[j]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the asm stores are invisible to hipcc

  if (real) {
    if (!NOSTORE) {
      if (p.nb > 1) p.xh[0 * p.channels + sc] = d1;
      if (p.nb > 2) p.xh[1 * p.channels + sc] = d2;
    }
    if (p.na > 1) p.yh[0 * p.channels + sc] = m1;
    if (p.na > 2) p.yh[1 * p.channels + sc] = m2;
  }
}

// ---------------------------------------------------------------------------
// k_duo: the same streaming scheme with the work split over the waves of a workgroup.
//
//   AUX wave  queues the tile DMA, waits for it, computes the feed-forward sums
//             p[n] = b0*x[n] (+ b1*x[n-1]) (+ b2*x[n-2]) for a whole tile with all 64 lanes
//             (time-parallel: lane group q owns samples 4j + q);
//   REC wave  runs only the serial part  y[n] = (p[n] + (-a1)*y[n-1]) + (-a2)*y[n-2]
//             (4 f64 ops and one LDS read per step, one row-select LDS write per 4 steps);
//   STORE wave (round 3; the AUX wave's job before) reads the finished y tile from LDS and stores it: the LDS round
//             trip and the eight store issues came out of the AUX wave's interval, which every tile's barrier waits
//             for -- +3 % at 4096 channels, +5 % at 512 .. 2048 and channel-major, +4 % for the one-pole envelope bank
//             (profiles/NOTES_r03.md 11).  Not in the FMA mode, which runs at the data path's limit: a separate
//             storing wave interleaves its writes with the reads instead of batching them, 9 % slower there.
//
// p and the recurrence term order are exactly the reference's left-to-right sum (numerator
// terms first, lazy_filters.py:198-224), so the split changes nothing numerically.  The waves
// meet at one s_barrier per tile: while REC works on tile i, STORE stores tile i-1, AUX queues tile
// i+3 and prepares p for tile i+1.  LDS: a 4-slot x ring (DMA target) + a 3-slot p/y ring.
// G = 16 channels per workgroup (the REC wave's other 48 lanes are ghosts as in k_wave).
// ---------------------------------------------------------------------------
#ifndef ALZ_DUO_XRING
#define ALZ_DUO_XRING 4
#endif
static constexpr int kXRing = ALZ_DUO_XRING, kPRing = 3, kYRing = 2;
#ifndef ALZ_DUO_CHUNKWAIT
#define ALZ_DUO_CHUNKWAIT 1
#endif

#ifndef ALZ_DUO_STORER
#define ALZ_DUO_STORER 1     // a third wave stores the finished tiles (not in the FMA mode; profiles/NOTES_r03.md 11)
#endif
// variant builds (FMA mode): 1 = the storing wave, non-temporal tiles and (with ALZ_PACE_ALL) the paced pass for the fused
// instantiations too; ALZ_DUO_FMA_ORDER 1 = fma(na1, y1, fma(na2, y2, p)): one dependent operation per step instead of two
#ifndef ALZ_DUO_FMA3
#define ALZ_DUO_FMA3 0
#endif
#ifndef ALZ_DUO_FMA_ORDER
#define ALZ_DUO_FMA_ORDER 0
#endif
#ifndef ALZ_DUO_AUXROWS
#define ALZ_DUO_AUXROWS 1    // time-major: the helper wave's lane groups own 16 consecutive rows each (0: rows 4 j + q)
#endif
// Which FMA instantiations have the storing wave.  Round 6 (profiles/r06_duo_fma_storer.log, 4096 channels x 2^20, one box,
// interleaved): CHANNEL-major with non-temporal tiles 349 - 350 Gsamples/s (0.70) against 323 for the two-wave FMA kernel; TIME-major
// free-running 298 - 307 against the default kernel's 323 - 326 -- but 358 (0.72) once every workgroup's helper wave requests its
// tiles on ONE clock (WArgs::tile_pace, profiles/r06_duo_tilepace.log): a time-major block streams whole rows only while the
// workgroups touch the same rows together.  So the non-temporal FMA instantiations of both layouts have the third wave.
constexpr bool duo_fma_storer(bool cm, bool nt) { return ALZ_DUO_FMA3 || nt; }
#ifndef ALZ_DUO_PACE_GBPS
#define ALZ_DUO_PACE_GBPS 5600
#endif
static constexpr int kDuoPaceGBps = ALZ_DUO_PACE_GBPS;   // the common tile clock of the time-major FMA kernel, see launch_wave
static constexpr int kDuoPaceGBpsOnePole = 5750;         // ... of the one-pole banks' bit-exact kernel (followed up to 6100 on four boxes, not on a fifth)
static constexpr int kDuoPaceGBpsInPlace = 5000;         // two-pole banks bit-exact IN PLACE: 296 free-running, 4900 306, 5100 314, 5300 299 (the recurrence
                                                         // wave's floor of that lease, 13.6 ms; past it the free-running rate again: r06_pace_inplace3.log)
static constexpr int kDuoPaceGBpsShared = 5000;          // 257 - 416 groups: some CUs hold two workgroups
static constexpr int kDuoPaceGBpsTwo = 5600;             // 417 - 512 groups: (nearly) all do
#ifndef ALZ_DUO_SLOT
#define ALZ_DUO_SLOT (8192 + kChunks * 16)
#endif
#ifndef ALZ_DUO_CMPAD
#define ALZ_DUO_CMPAD 16
#endif
static constexpr int kDuoSlot = ALZ_DUO_SLOT;   // ring slot stride (tile + pads); its residue mod 256 matters, see DESIGN.md

// Skew.  The four lane groups of the REC wave are exact copies of the same 16 recurrences; group
// q runs q steps behind group 0 (it reads p[s - q] at step s).  At every step with s % 4 == 3
// the wave then holds y[s-3..s] -- one row per lane group -- and a single ds_write_b64 of the
// value each lane has just computed stores four rows: no per-step 18-cycle LDS write and no
// row-select moves either.  y goes to its own small ring (p must stay readable for the lagging
// groups across the tile boundary).  Group 0 is never behind, so it owns the final state.
// FMA = true is the opt-in fused mode (alz_bank_set_fused): v_fma_f64 instead of separately rounded
// mul + add -- half the recurrence chain, NOT bit-identical to the reference (differences at the
// 1e-13 level, far inside the 1e-6 contract).
// DIV = true divides the finished sum by a0 (``(...) / gain``, lazy_filters.py:236-240): the banks
// whose a0 is not 1 -- the correctly rounded division is a ~12-instruction dependent sequence, so
// it has its own instantiation.
// per-wave cycle accounting (variant builds: -DALZ_DUO_TIMING; a clock read waits for the wave's outstanding LDS operations, so "work"
// includes that): work = barrier exit -> barrier entry, wait = inside the barrier, per tile
#ifdef ALZ_DUO_TIMING
#define DUO_CLOCK_DECL long long dc_work = 0, dc_wait = 0, dc_last = __builtin_readcyclecounter(); long long dc_n = 0;
#define DUO_BARRIER() do { const long long c0_ = __builtin_readcyclecounter(); __builtin_amdgcn_s_barrier(); \
    const long long c1_ = __builtin_readcyclecounter(); dc_work += c0_ - dc_last; dc_wait += c1_ - c0_; dc_last = c1_; ++dc_n; } while (0)
#define DUO_CLOCK_REPORT(who) do { if (blockIdx.x == 5 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && dc_n > 1000) printf("k_duo %s%s%s wave %s: %.1f cycles of work + %.1f in the barrier per tile (%lld tiles)\n", \
    CM ? "chan" : "time", FMA ? " fma" : "", STORER ? " 3 waves" : " 2 waves", who, (double)dc_work / (double)dc_n, (double)dc_wait / (double)dc_n, dc_n); } while (0)
#else
#define DUO_CLOCK_DECL
#define DUO_BARRIER() __builtin_amdgcn_s_barrier()
#define DUO_CLOCK_REPORT(who) do {} while (0)
#endif
template <bool CM, unsigned PB, unsigned PA, bool FMA, bool DIV = false, bool NOSTORE = false, int PRE = 0, bool NT = false>
__global__ __launch_bounds__((ALZ_DUO_STORER && (!FMA || duo_fma_storer(CM, NT)) && !NOSTORE) ? 192 : 128) void k_duo(WArgs p) {
  constexpr bool STORER = ALZ_DUO_STORER && (!FMA || duo_fma_storer(CM, NT)) && !NOSTORE;    // a third wave stores the finished tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 16, T = 64;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cl = lane & 15, q = lane >> 4;
  const int64_t c0 = p.c_first + (int64_t)blockIdx.x * G;
  const int64_t c = c0 + cl;
  const int64_t sc = (int64_t)blockIdx.y * p.chunk_c + c;     // state slot (== c unless time-parallel)
  const int64_t in0 = (p.n_inputs && p.map_input) ? c0 % p.n_inputs : c0;   // OUTER bank: inputs of this group
  const int64_t set = p.n_inputs ? c / p.n_inputs : ((p.n_sets == 1) ? 0 : c);
  const int64_t nt = p.n_tiles;
  DUO_CLOCK_DECL
  if constexpr (CM) stagger_start((unsigned)p.stagger, blockIdx.x);
  char *xring = smem;
  // p and y rings are written and read only by this kernel's own lanes, so their layout is free:
  // channel-major keeps 16 bytes after EVERY channel (a half-wave -- 16 channels x 2 lane groups --
  // then touches 32 distinct bank pairs; a pad per two channels, as the DMA layout of the x ring
  // has it, made every access 2-way conflicted)
  constexpr int kChanPitch = 64 * 8 + 16;                      // bytes per channel row in the p / y rings (CM)
  constexpr int kPYSlot = CM ? 16 * kChanPitch : kDuoSlot;
  char *pring = smem + kXRing * kDuoSlot;
  char *yring = pring + kPRing * kPYSlot;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  // x ring (DMA target): byte offset of element (u, cl) = lane_off + ALZ_EOFF(u), 16-byte pad
  // after every 1 KiB chunk.  p and y rings: TIME rows are unpadded (row u at u*128), CHAN rows
  // keep the pad between channel pairs (REC's lanes read one channel each).
  // channel-major: a DMA chunk is two channels of 512 B, which share their banks; a 32-byte pad
  // per chunk spreads the eight chunks and the four skewed lane groups over all 32 bank pairs
  // (16 bytes left them 4-way conflicted).  Time-major rows keep the 16-byte pad.
  constexpr int kPad = CM ? ALZ_DUO_CMPAD : 16, kChunkLds = 1024 + kPad;
  const int lane_off = CM ? cl * T * 8 + ((cl * T) >> 7) * kPad : cl * 8;
#define ALZ_EOFF(u) (CM ? (u) * 8 : (u) * G * 8 + (((u) * G) >> 7) * 16)
  constexpr int kStep = CM ? 8 : G * 8;               // bytes from sample u to u+1 in the p/y rings
  constexpr int kOutChunk = CM ? kChunkLds : 1024;         // bytes per 1 KiB store chunk in the y ring
  const int lane_off_p = CM ? cl * kChanPitch : cl * 8;

  if (wave >= 1) {
    // ------------------------------ AUX (and, with ALZ_DUO_STORER, the storing wave) ------------------------------
    int64_t x_off, y_off, x_chunk, y_chunk, x_tile, y_tile;
    if (!CM) {
      const int row = lane / 8, cp = lane % 8;
      x_off = (int64_t)row * p.ldx + in0 + 2 * cp;
      y_off = (int64_t)row * p.ldy + c0 + 2 * cp;
      x_chunk = 8 * p.ldx; y_chunk = 8 * p.ldy;
      x_tile = (int64_t)T * p.ldx; y_tile = (int64_t)T * p.ldy;
    } else {
      const int ch = lane / 32, sp = lane % 32;
      x_off = (in0 + ch) * p.ldx + 2 * sp;
      y_off = (c0 + ch) * p.ldy + 2 * sp;
      x_chunk = 2 * p.ldx; y_chunk = 2 * p.ldy;
      x_tile = T; y_tile = T;
    }
    double b0 = 0, b1 = 0, b2 = 0;
    if (PB & 1u) b0 = p.b[0 * p.n_sets + set];
    if (PB & 2u) b1 = p.b[1 * p.n_sets + set];
    if (PB & 4u) b2 = p.b[2 * p.n_sets + set];
    double d1 = (p.nb > 1) ? p.xh[0 * p.channels + sc] : 0.0;   // x[-1], x[-2] of the stream
    double d2 = (p.nb > 2) ? p.xh[1 * p.channels + sc] : 0.0;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(d1), "+v"(d2));
    const double *xg = p.x + x_off + (int64_t)blockIdx.y * p.chunk_x;
    double *yg = p.y + y_off + (int64_t)blockIdx.y * p.chunk_y;

    auto queue_tile = [&](int64_t t) {
      const int s = (int)(t % kXRing);
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        dma16<NT>(xg + t * x_tile + j * x_chunk, lds0 + s * kDuoSlot + j * kChunkLds);
    };
    // feed-forward of tile t: lane (q, cl) owns samples 4j + q (j = 0..15) of channel cl, so the
    // four lane groups of one ds_write_b64 fill four consecutive rows of the p ring (512
    // contiguous bytes; with 16q + j the groups were 2 KiB apart: a 4-way bank conflict)
    auto feed_forward = [&](int64_t t) {
      const char *xs = xring + (int)(t % kXRing) * kDuoSlot + lane_off;
      const char *xp = xring + (int)((t + kXRing - 1) % kXRing) * kDuoSlot + lane_off;  // tile t-1
      char *ps = pring + (int)(t % kPRing) * kPYSlot + lane_off_p;
      auto xat = [&](int u) -> double {       // x[u] of this tile; u = -1, -2 reach into tile t-1
        return pre_in<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(u)));
      };
      // sample 4j + q - d sits q - d steps after sample 4j; in TIME layout the 16-byte pad after
      // every 8 rows is crossed only for even j with q - d < 0 (adj), never otherwise
      const int adj1 = (!CM && q == 0) ? 16 : 0, adj2 = (!CM && q < 2) ? 16 : 0;
      const char *x_d0 = xs + q * kStep;
      const char *x_d1[2] = {xs + (q - 1) * kStep - adj1, xs + (q - 1) * kStep};   // [j odd]
      const char *x_d2[2] = {xs + (q - 2) * kStep - adj2, xs + (q - 2) * kStep};
      double x0[16], x1[16], x2[16];
      auto read_rows = [&](int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; ++j) {
          if constexpr (PB & 1u) x0[j] = pre_in<PRE>(*reinterpret_cast<const double *>(x_d0 + ALZ_EOFF(4 * j)));
          if constexpr (PB & 2u) {
            if (j > 0) x1[j] = pre_in<PRE>(*reinterpret_cast<const double *>(x_d1[j & 1] + ALZ_EOFF(4 * j)));
          }
          if constexpr (PB & 4u) {
            if (j > 0) x2[j] = pre_in<PRE>(*reinterpret_cast<const double *>(x_d2[j & 1] + ALZ_EOFF(4 * j)));
          }
        }
      };
      auto sum_rows = [&](int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; ++j) {
          double acc = 0.0;
          bool first = true;
          if constexpr (PB & 1u) { acc = b0 * x0[j]; first = false; }
          if constexpr (PB & 2u) {
            if (FMA && !first) acc = __builtin_fma(b1, x1[j], acc);
            else { const double v = b1 * x1[j]; acc = first ? v : acc + v; }
            first = false;
          }
          if constexpr (PB & 4u) {
            if (FMA && !first) acc = __builtin_fma(b2, x2[j], acc);
            else { const double v = b2 * x2[j]; acc = first ? v : acc + v; }
            first = false;
          }
          *reinterpret_cast<double *>(ps + (4 * j + q) * kStep) = acc;
        }
      };
      auto edge_rows = [&]() {
        // j == 0: samples q - 1 and q - 2 may lie before the tile
        if constexpr ((PB & 6u) != 0) {
          double pm1, pm2;                        // x[-1], x[-2] relative to this tile
          if (t > 0) {
            pm1 = pre_in<PRE>(*reinterpret_cast<const double *>(xp + ALZ_EOFF(T - 1)));
            pm2 = pre_in<PRE>(*reinterpret_cast<const double *>(xp + ALZ_EOFF(T - 2)));
          } else {
            pm1 = d1;
            pm2 = d2;
          }
          const double c0 = xat(0), c1 = xat(1);
          if constexpr (PB & 2u) x1[0] = q == 0 ? pm1 : q == 1 ? c0 : q == 2 ? c1 : xat(2);
          if constexpr (PB & 4u) x2[0] = q == 0 ? pm2 : q == 1 ? pm1 : q == 2 ? c0 : c1;
        }
      };
      // A one-pole bank's recurrence wave steps in ~13 cycles instead of 28 and needs its p rows that much sooner; this
      // wave's 16 reads + 16 writes in one burst then stand in the LDS queue in front of them.  With every CU busy
      // (the launcher decides: aux_pace > 0) the tile goes out in four quarters with pauses between them -- this wave
      // has most of the interval to spare: 4096 channels x 2^20 282 - 293 -> 309 - 340 Gsamples/s over three boxes
      // (channel-major 267 - 313 -> 319 - 356; profiles/NOTES_r04.md 4).
      bool paced = false;
      if constexpr (((PB == 1u && PA == 1u) || ALZ_PACE_ALL) && (!FMA || ALZ_DUO_FMA3) && !NOSTORE) paced = p.aux_pace > 0;
      if constexpr (!CM && (PB & 6u) != 0 && ALZ_DUO_AUXROWS && FMA && !STORER) {
        // Time-major two-wave FMA instantiations, numerators with more than b0 (round 6): lane group q owns the 16 CONSECUTIVE rows
        // 16 q .. 16 q + 15, so the delayed samples of a row are the registers that held the rows before it -- 18 LDS reads per
        // tile instead of 48 (the interleaved ownership 4 j + q reads every tap's row again).  Same products and sums per
        // (row, channel): the p ring receives identical doubles.  Measured (profiles/r06_duo_rows.log, same box, interleaved):
        // 2048 channels x 2^20 in the FMA mode 171.3 -> 180.9 Gsamples/s (+5.6 %); with the storing wave -- the default kernel
        // (324.8 -> 323.7) and the three-wave FMA variant (294.6 -> 287.2) -- it loses (the four lane groups then write p rows
        // 2 KiB apart), so those keep rows 4 j + q.
        const char *xq = xs + q * (16 * kStep + 32);           // row 16 q of the tile (a 16-byte pad after every 8 rows)
        double xr[16], xm1, xm2;
#pragma unroll
        for (int j = 0; j < 16; ++j) xr[j] = pre_in<PRE>(*reinterpret_cast<const double *>(xq + ALZ_EOFF(j)));
        {
          double pm1, pm2;                                     // the two rows before the tile
          if (t > 0) {
            pm1 = pre_in<PRE>(*reinterpret_cast<const double *>(xp + ALZ_EOFF(T - 1)));
            pm2 = pre_in<PRE>(*reinterpret_cast<const double *>(xp + ALZ_EOFF(T - 2)));
          } else {
            pm1 = d1;
            pm2 = d2;
          }
          // rows 16 q - 1 and 16 q - 2 of this tile for q > 0 (the pad before row 16 q lies between)
          const double t1 = pre_in<PRE>(*reinterpret_cast<const double *>(xq - 16 - kStep + (q == 0 ? 16 + kStep : 0)));
          const double t2 = pre_in<PRE>(*reinterpret_cast<const double *>(xq - 16 - 2 * kStep + (q == 0 ? 16 + 2 * kStep : 0)));
          xm1 = q == 0 ? pm1 : t1;
          xm2 = q == 0 ? pm2 : t2;
        }
        char *pq = ps + q * 16 * kStep;                        // rows 16 q + j of the p tile (unpadded rows)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const double a1 = j >= 1 ? xr[j >= 1 ? j - 1 : 0] : xm1;
          const double a2 = j >= 2 ? xr[j >= 2 ? j - 2 : 0] : (j == 1 ? xm1 : xm2);
          double acc = 0.0;
          bool first = true;
          if constexpr (PB & 1u) { acc = b0 * xr[j]; first = false; }
          if constexpr (PB & 2u) {
            if (FMA && !first) acc = __builtin_fma(b1, a1, acc);
            else { const double v = b1 * a1; acc = first ? v : acc + v; }
            first = false;
          }
          if constexpr (PB & 4u) {
            if (FMA && !first) acc = __builtin_fma(b2, a2, acc);
            else { const double v = b2 * a2; acc = first ? v : acc + v; }
            first = false;
          }
          *reinterpret_cast<double *>(pq + j * kStep) = acc;
        }
      } else
      if (paced) {
        const int units = p.aux_pace & 15;
        if (p.aux_pace & 16) {                   // eighths
          read_rows(0, 2);
          edge_rows();
          sum_rows(0, 2);
#pragma unroll
          for (int g = 1; g < 8; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            for (int r = 0; r < units; ++r) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_sched_barrier(0);
            read_rows(2 * g, 2 * g + 2);
            sum_rows(2 * g, 2 * g + 2);
          }
        } else {
          read_rows(0, 4);
          edge_rows();
          sum_rows(0, 4);
#pragma unroll
          for (int g = 1; g < 4; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            for (int r = 0; r < units; ++r) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_sched_barrier(0);
            read_rows(4 * g, 4 * g + 4);
            sum_rows(4 * g, 4 * g + 4);
          }
        }
      } else {
        read_rows(0, 16);
        edge_rows();
        sum_rows(0, 16);
      }
    };
    auto store_tile = [&](int64_t t) {
      const char *ys = yring + (int)(t % kYRing) * kPYSlot;
      double *yt = yg + t * y_tile;
      dbl2 v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        v[j] = *reinterpret_cast<const dbl2 *>(ys + (CM ? (2 * j + lane / 32) * kChanPitch + (lane % 32) * 16
                                                          : j * kOutChunk + lane * 16));
#pragma unroll
      for (int j = 0; j < kChunks; ++j) store16<NT>(yt + j * y_chunk, v[j]);
    };

    if (STORER && wave == 2) {
      // the storing wave: tile i - 1 while REC works on tile i
      __builtin_amdgcn_s_barrier();
      [[maybe_unused]] bool convoy_off_s = false;
      for (int64_t i = 0; i < nt; ++i) {
        if (!NOSTORE && i >= 1 && !ALZ_DBG(p, 4)) store_tile(i - 1);
#ifdef ALZ_TUNING   // (experiment: the convoy's checkpoint in THIS wave, which has the time -- cfg bit 16; the barrier below holds the workgroup)
        if (p.convoy && (p.convoy_cfg & 0x10000)) convoy_sync(p.convoy, p.convoy_cfg, p.convoy_groups, i, convoy_off_s);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!ALZ_DBG(p, 8)) DUO_BARRIER();
      }
      if (!NOSTORE) store_tile(nt - 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DUO_CLOCK_REPORT("STORE");
      return;
    }
    for (int t = 0; t < kXRing - 1 && t < nt && !ALZ_DBG(p, 1); ++t) queue_tile(t);
    wait_vm((int)((nt < kXRing - 1 ? nt : kXRing - 1) - 1) * kChunks);   // tile 0 has landed
    feed_forward(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long pace0 = p.tile_pace > 0 ? (long long)wall_clock64() : 0;

    long long pace_shift = 0;
    [[maybe_unused]] bool convoy_off = false;
    for (int64_t i = 0; i < nt; ++i) {
      if (!STORER && !NOSTORE && i >= 1 && !ALZ_DBG(p, 4)) store_tile(i - 1);
      if (p.tile_pace > 0) pace_wait(pace0, i, p.tile_pace, pace_shift);   // all workgroups keep to one clock (alz_common.h)
#ifdef ALZ_TUNING   // (experiment, profiles/r06_convoy*.log: not in the shipped kernels)
      if (p.convoy && !(STORER && (p.convoy_cfg & 0x10000))) convoy_sync(p.convoy, p.convoy_cfg, p.convoy_groups, i, convoy_off);
#endif
      if (i + kXRing - 1 < nt && !ALZ_DBG(p, 1)) queue_tile(i + kXRing - 1);
      if (i + 1 < nt) {
        // operations issued after tile i+1's DMA: the DMA of tiles i+2 .. i+kXRing-1 and the stores
        // of the kXRing-2 tiles finished since (a count above the 6-bit vmcnt range is clamped in
        // wait_vm: waiting for a few more of the oldest operations is always safe)
        const int64_t last = (i + kXRing - 1 < nt - 1) ? i + kXRing - 1 : nt - 1;
        const int64_t dma_after = last - (i + 1);
        const int64_t stores_after = (NOSTORE || STORER) ? 0 : (i < kXRing - 2 ? i : kXRing - 2);
        wait_vm(ALZ_DBG(p, 5) ? 0 : (int)(dma_after + stores_after) * kChunks);
        if (!ALZ_DBG(p, 2)) feed_forward(i + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!ALZ_DBG(p, 8)) DUO_BARRIER();
    }
    DUO_CLOCK_REPORT("AUX");
    if (!STORER && !NOSTORE) store_tile(nt - 1);
    // input history for the next block: the last two x samples (held by the q == 3 lanes)
    if (!NOSTORE && q == 3) {
      const char *xs = xring + (int)((nt - 1) % kXRing) * kDuoSlot + lane_off;
      if (p.nb > 1) p.xh[0 * p.channels + sc] = pre_in<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 1)));
      if (p.nb > 2) p.xh[1 * p.channels + sc] = pre_in<PRE>(*reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 2)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ------------------------------ REC ------------------------------
    double na1 = 0, na2 = 0;
    if (PA & 1u) na1 = -p.a[1 * p.n_sets + set];
    if (PA & 2u) na2 = -p.a[2 * p.n_sets + set];
    double m1 = (p.na > 1) ? p.yh[0 * p.channels + sc] : 0.0;
    double m2 = (p.na > 2) ? p.yh[1 * p.channels + sc] : 0.0;
    double a0 = 1.0;
    if constexpr (DIV) a0 = p.a[0 * p.n_sets + set];
    asm volatile("" : "+v"(na1), "+v"(na2), "+v"(m1), "+v"(m2), "+v"(a0));
    double t2 = na2 * m2;                                    // (-a2) * y[n-2] for the next step
    __builtin_amdgcn_s_barrier();                            // p of tile 0 is ready
    constexpr int NCH = T / 8;
    int ps_cur = 0, ps_prv = kPRing - 1, ys_cur = 0;         // ring slots of tile i, rotated by hand
    for (int64_t i = 0; i < nt; ++i) {
      // this lane reads sample (u - q) of the tile; u - q < 0 lives in the previous tile's slot
      const char *cur = pring + ps_cur * kPYSlot + lane_off_p - q * kStep;
      const char *prv = pring + ps_prv * kPYSlot + lane_off_p + (T - q) * kStep;
      char *wr = yring + ys_cur * kPYSlot + lane_off_p - q * kStep;
      ps_prv = ps_cur;
      ps_cur = (ps_cur + 1 == kPRing) ? 0 : ps_cur + 1;
      ys_cur = (ys_cur + 1 == kYRing) ? 0 : ys_cur + 1;
      double pr[3][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const char *src = (u < 3 && u < q) ? prv : cur;       // (u < q is per-lane; u >= 3 never)
        pr[0][u] = *reinterpret_cast<const double *>(src + u * kStep);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) pr[1][u] = *reinterpret_cast<const double *>(cur + (8 + u) * kStep);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k + 2 < NCH) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            pr[(k + 2) % 3][u] = *reinterpret_cast<const double *>(cur + ((k + 2) * 8 + u) * kStep);
        }
        __builtin_amdgcn_sched_barrier(0);
#if ALZ_DUO_CHUNKWAIT
        {
          // ONE wait per chunk for the four paired reads of chunk k, through the builtin so that the
          // compiler's wait-count pass sees it and does not emit a wait of its own before every second
          // step (an s_waitcnt costs a lone wave an issue slot even when it does not stall).  Newer LDS
          // operations that may stay outstanding: the reads of chunks k+1 and k+2 (four ds_read2 each)
          // and the y write that closed chunk k-1.
          const int newer = (k + 1 < NCH ? 4 : 0) + (k + 2 < NCH ? 4 : 0) + ((!NOSTORE && k > 0) ? 1 : 0);
          switch (newer) {                                   // (k is an unrolled loop index: folded at compile time)
            case 0: __builtin_amdgcn_s_waitcnt(0xC07F); break;
            case 1: __builtin_amdgcn_s_waitcnt(0xC17F); break;
            case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
            case 5: __builtin_amdgcn_s_waitcnt(0xC57F); break;
            case 8: __builtin_amdgcn_s_waitcnt(0xC87F); break;
            default: __builtin_amdgcn_s_waitcnt(0xC97F); break;
          }
          __builtin_amdgcn_sched_barrier(0);                  // (the wait stays ahead of the chunk's arithmetic)
        }
#endif
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          double acc = pr[k % 3][u];
          double t2n = 0.0;
          if constexpr (FMA) {
            if constexpr (ALZ_DUO_FMA_ORDER && PA == 3u) {
              acc = __builtin_fma(na1, m1, __builtin_fma(na2, m2, acc));
            } else {
              if constexpr (PA & 1u) acc = __builtin_fma(na1, m1, acc);
              if constexpr (PA & 2u) acc = __builtin_fma(na2, m2, acc);
            }
          } else if constexpr (PA == 3u) {
            // the product with y[n-2] was formed one step ago (t2), so only mul -> add -> add
            // sits on the serial chain and the other product fills the first latency slot
            const double t1 = na1 * m1;
            t2n = na2 * m1;
            acc = (acc + t1) + t2;
          } else {
            if constexpr (PA & 1u) acc = acc + na1 * m1;
            if constexpr (PA & 2u) acc = acc + na2 * m2;
          }
          if constexpr (DIV) acc = acc / a0;                  // (the carried product used the old y[n-1])
          if (k == 0 && u < 3 && i == 0) {
            // start of the stream: group q has nothing to do before step q; hold its state
            const bool on = u >= q;
            m2 = on ? m1 : m2;
            m1 = on ? acc : m1;
            t2 = on ? t2n : t2;
          } else {
            m2 = m1;
            m1 = acc;
            t2 = t2n;
          }
          if (!NOSTORE && (u & 3) == 3) *reinterpret_cast<double *>(wr + (k * 8 + u) * kStep) = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!ALZ_DBG(p, 8)) DUO_BARRIER();                       // y of tile i done, p of tile i+1 ready
    }
    DUO_CLOCK_REPORT("REC");
    if (lane < G) {
      if (p.na > 1) p.yh[0 * p.channels + sc] = m1;
      if (p.na > 2) p.yh[1 * p.channels + sc] = m2;
    }
  }
#undef ALZ_EOFF
}

// ---------------------------------------------------------------------------
// dispatch: curated tap patterns (everything else stays on k_small)
// ---------------------------------------------------------------------------
typedef void (*wave_fn)(WArgs);

template <int G, bool CM, bool NOSTORE = false, int PRE = 0>
static wave_fn pick_pattern(unsigned pb, unsigned pa) {
#define ALZ_PAT(PB_, PA_) \
  if (pb == PB_ && pa == PA_) return (wave_fn)k_wave<G, CM, PB_, PA_, NOSTORE, PRE>;
  ALZ_PAT(1, 1)  // b0           / a1        lowpass.pole, highpass.pole
  ALZ_PAT(3, 1)  // b0 b1        / a1        lowpass.z, highpass.z
  ALZ_PAT(1, 3)  // b0           / a1 a2     resonator.poles_exp, lowpass.pole**2, gammatone poles
  ALZ_PAT(3, 3)  // b0 b1        / a1 a2     gammatone.slaney sections
  ALZ_PAT(5, 3)  // b0    b2     / a1 a2     resonator.z_exp
  ALZ_PAT(7, 3)  // b0 b1 b2     / a1 a2     general biquad
  ALZ_PAT(1, 2)  // b0           /    a2
  if constexpr (!NOSTORE) {
    ALZ_PAT(7, 0)  // 3-tap FIR
    ALZ_PAT(3, 0)  // 2-tap FIR
  }
#undef ALZ_PAT
  return nullptr;
}

template <bool CM, bool FMA, bool DIV = false, bool NOSTORE = false, int PRE = 0, bool NT = false>
static wave_fn pick_duo_pattern(unsigned pb, unsigned pa) {
#define ALZ_PAT(PB_, PA_) \
  if (pb == PB_ && pa == PA_) return (wave_fn)k_duo<CM, PB_, PA_, FMA, DIV, NOSTORE, PRE, NT>;
  ALZ_PAT(1, 1) ALZ_PAT(3, 1) ALZ_PAT(1, 3) ALZ_PAT(3, 3) ALZ_PAT(5, 3) ALZ_PAT(7, 3) ALZ_PAT(1, 2)
#undef ALZ_PAT
  return nullptr;
}

template <int PRE>
static wave_fn pick_wave(int g, bool cm, unsigned pb, unsigned pa) {
  if (g == 16) return cm ? pick_pattern<16, true, false, PRE>(pb, pa) : pick_pattern<16, false, false, PRE>(pb, pa);
  if (g == 32) return cm ? pick_pattern<32, true, false, PRE>(pb, pa) : pick_pattern<32, false, false, PRE>(pb, pa);
  return cm ? pick_pattern<64, true, false, PRE>(pb, pa) : pick_pattern<64, false, false, PRE>(pb, pa);
}

// Runs the streaming kernel over the part of the block it can take (full tiles of full
// channel groups) and reports that part; the caller finishes the rest with k_small.
//   *done_tiles_samples: samples per channel consumed (multiple of the tile length)
//   *done_channels:      channels covered (multiple of G), starting at channel 0
// With `ch` (time-parallel mode, alz_scan.hip) the launch covers ch->n_chunks chunks of ch->chunk_len
// samples at once, every chunk from / into its own state slot of ch->vxh / ch->vyh; it then takes
// either the whole bank (all channels, whole chunks) or nothing.
static int launch_wave_impl(const SectionDev &sec, const BlockIO &io, hipStream_t stream, const WaveChunks *ch,
                            int64_t *done_samples, int64_t *done_channels, const char **kernel_name) {
  *done_samples = 0;
  *done_channels = 0;
  if (!(sec.nb <= 3 && sec.na <= 3 && sec.uniform)) return ALZ_OK;
  const bool cm = io.sxn == 1 && io.syn == 1 && !(io.sxc == 1 && io.syc == 1);
  const bool tm = io.sxc == 1 && io.syc == 1;
  if (!cm && !tm) return ALZ_OK;
  const int64_t ldx = cm ? io.sxc : io.sxn, ldy = cm ? io.syc : io.syn;
  // 16-byte pieces: base pointers 16-byte aligned and even leading dimensions
  if (((uintptr_t)io.x | (uintptr_t)io.y) & 15) return ALZ_OK;
  if ((ldx | ldy) & 1) return ALZ_OK;
  // lanes that run a recurrence: the channels, times the chunks in time-parallel mode
  const int64_t lanes = ch ? ch->n_chunks * io.channels : io.channels;
  // group width: keep at least ~256 waves in flight for small banks
  // measured on MI355X (profiles/r02_bank_width_sweep.log): the two-wave kernel (G = 16) wins below
  // ~12k channels, the single-wave kernel with 64 real lanes from 16k up
  int g = 64;
  if (lanes < 64 * 256) g = 32;
  if (lanes < 48 * 256) g = 16;
  // ... except time-major streaming blocks of whole ROUNDS of 512 two-wave workgroups (8192, 16 384, 24 576, 32 768 channels on 256 CUs):
  // on the common tile clock (tile_pace below; each round starts where the one before, kept in step, ends) k_duo does 341 - 350
  // Gsamples/s bit-exact against k_wave<16>'s 300 - 305 at 8192 channels and k_wave<64>'s 330 / 306 / 319 at 16 384 / 24 576 / 32 768
  // (profiles/r06_pace5.log, r06_duo_rounds.log; free-running it is the slower one)
  const int64_t round_lanes = 32ll * (device_cus() > 0 ? device_cus() : 256);
  // (a block of that size between the sections of a cascade -- temporal tiles: ONE round only -- 8192 channels 302 -> 349.5 in both
  // modes; 16 384 and 32 768 channels in rounds lose with temporal tiles, 289 - 292 / 263 - 265 against k_wave<64>'s 323 / 305:
  // profiles/r06_pace_inplace2.log.  In place the tiles are non-temporal, nt_in_place below, and the rounds win: 348 at 16 384)
  const bool big_block = !ch && (uint64_t)io.n * (uint64_t)io.channels * 8u >= (256ull << 20);
  const int64_t rounds_max = (io.stream_once || (big_block && io.x == io.y && ALZ_TUNE("ALZ_NT_INPLACE", 1) != 0))   // (in place: with non-temporal tiles, below)
                                 ? ALZ_TUNE("ALZ_DUO_ROUNDS_MAX", 4) : (big_block ? 1 : 0);
  const bool clocked_rounds = !cm && !ch && !sec.any_div && !io.pre_op && lanes % round_lanes == 0 &&
                              lanes / round_lanes <= rounds_max && ALZ_TUNE("ALZ_DUO_PACED", 1) != 0;
  if (clocked_rounds) g = 16;
  static const int g_env = ALZ_TUNE("ALZ_G", 0);   // tuning override
  if (g_env == 16 || g_env == 32 || g_env == 64) g = g_env;
  if (sec.any_div) g = 16;      // a0 != 1 somewhere: only the two-wave kernel has the dividing form
  const bool outer = io.mode == ALZ_BANK_OUTER;
  if (outer && io.map_input) {
    // the first section of an OUTER bank reads x by input index: a wave's channels must be adjacent inputs
    // of ONE coefficient set.  Later cascade sections (and a bank whose input was expanded to one column
    // per channel, alz_api.hip) read by the channel's own index, like a diagonal bank.
    while (g > 16 && io.n_inputs % g) g /= 2;
    if (io.n_inputs % g) return ALZ_OK;
  }
  if (ch) {
    if (sec.any_div) return ALZ_OK;
    if (g == 32) g = 16;                       // the store-less kernels exist for G = 64 and G = 16
    if (g == 64 && io.channels % 64) g = 16;
    if (io.channels % g) return ALZ_OK;        // a group must not straddle two chunks
  }
  const int64_t groups = io.channels / g;
  const int t = 8192 / (8 * g);
  const int64_t tiles = (ch ? ch->chunk_len : io.n) / t;
  if (groups == 0 || tiles == 0) return ALZ_OK;
  if (ch && tiles * t != ch->chunk_len) return ALZ_OK;
  // The opt-in FMA mode ALLOWS contraction; it does not have to be used where it loses.  A time-major bank that fills the
  // chip with one workgroup per CU (256 - 320 groups of 16 channels) is bound by the helper wave's pass over the tile, not
  // by the recurrence: the default kernel (storing wave, non-temporal tiles, paced pass) does 306 - 326 Gsamples/s there
  // against the two-wave FMA kernel's 295 - 304 (profiles/NOTES_r04.md 5, NOTES_r05.md 10).  A block of streaming size has
  // the three-wave FMA kernel on a common tile clock instead (tile_pace below: 358); smaller or in-place blocks of that
  // width keep the default kernel, whose doubles are within every contract of the mode.  Everywhere else the FMA kernels
  // are 2 - 12 % ahead.
  const bool chip_wide_tm = g == 16 && !cm && !ch && groups >= 256 && groups <= 320;
  const bool paced_tm = chip_wide_tm && io.fused && io.stream_once && !io.pre_op && !sec.any_div && ALZ_TUNE("ALZ_DUO_NT", 1) != 0 && ALZ_TUNE("ALZ_DUO_PACED_FMA", 1) != 0;
  const bool fused = io.fused && (ALZ_DUO_FMA3 || !chip_wide_tm || paced_tm);   // (variant builds with the storing wave in every FMA kernel: no exception)
  // small banks: the two-wave kernel (recurrence wave + helper wave per 16 channels)
  static const int duo_env = ALZ_TUNE("ALZ_DUO", 1);
  const bool nostore = ch && ch->nostore;
  const bool pre_abs = io.pre_op == ALZ_MAP_ABS;
  if (io.pre_op && (!pre_abs || ch || fused || sec.any_div)) return ALZ_OK;   // the caller maps the input first
  // the single-wave kernel overtakes the two-wave one once a CU holds more than two workgroups' worth
  // of channels (profiles/r02_bank_width_sweep.log: 8192 channels 297 vs 288, 12288 283 vs 253)
  static const int single_from = ALZ_TUNE("ALZ_DUO_MAX_LANES", 8192);
  // ... unless the launch is whole rounds of workgroups on the tile clock (clocked_rounds above)
  const bool duo_clocked_wide = g == 16 && clocked_rounds;
  const bool prefer_single = g == 16 && lanes >= single_from && !ch && !duo_clocked_wide;
  // non-temporal tile traffic (its own instantiations): large blocks that this call reads once and does not read back
  // ... and a block of that size processed IN PLACE (alz_api.hip does not call it `stream_once`): measured in place with and without
  // (profiles/r06_nt_inplace.log) 16 384 channels k_wave<64> 315 - 324 -> k_duo in rounds 348, 6144 channels 275 -> 292, channel-major
  // 4096 channels 302 -> 310 bit-exact and 305 -> 334 in the FMA mode, one-pole banks and 8192 channels unchanged -- except the
  // time-major two-pole bank that fills the chip once, which would lose its in-place clock (312 -> 295 - 306 in both modes) and
  // keeps the tiles temporal
  const bool nt_in_place = !io.stream_once && big_block && io.x == io.y && !(chip_wide_tm && !(sec.present_b == 1u && sec.present_a == 1u)) &&
                           ALZ_TUNE("ALZ_NT_INPLACE", 1) != 0;
  const bool nt_tiles = (io.stream_once || nt_in_place) && !ch && (!fused || ALZ_DUO_FMA3 || cm || paced_tm) && ALZ_TUNE("ALZ_DUO_NT", 1) != 0;
  wave_fn duo = nullptr;
  bool duo_fma = false;
  if (g == 16 && sec.any_div) {
    duo = cm ? pick_duo_pattern<true, false, true>(sec.present_b, sec.present_a)
             : pick_duo_pattern<false, false, true>(sec.present_b, sec.present_a);
    if (!duo) return ALZ_OK;
  } else if (g == 16 && nostore) {
    duo = cm ? pick_duo_pattern<true, false, false, true>(sec.present_b, sec.present_a)
             : pick_duo_pattern<false, false, false, true>(sec.present_b, sec.present_a);
    if (!duo) return ALZ_OK;
  } else if (g == 16 && pre_abs && duo_env && !prefer_single) {
    duo = nt_tiles ? (cm ? pick_duo_pattern<true, false, false, false, 1, true>(sec.present_b, sec.present_a)
                         : pick_duo_pattern<false, false, false, false, 1, true>(sec.present_b, sec.present_a))
                   : (cm ? pick_duo_pattern<true, false, false, false, 1>(sec.present_b, sec.present_a)
                         : pick_duo_pattern<false, false, false, false, 1>(sec.present_b, sec.present_a));
  } else if (g == 16 && ((duo_env && !prefer_single) || ch)) {
    if (fused && nt_tiles && cm)
      duo_fma = true, duo = pick_duo_pattern<true, true, false, false, 0, true>(sec.present_b, sec.present_a);
    else if (fused && nt_tiles)   // (shipped build: only the chip-wide streaming blocks, paced_tm above)
      duo_fma = true, duo = pick_duo_pattern<false, true, false, false, 0, true>(sec.present_b, sec.present_a);
    else if (fused)
      duo_fma = true, duo = cm ? pick_duo_pattern<true, true>(sec.present_b, sec.present_a)
                               : pick_duo_pattern<false, true>(sec.present_b, sec.present_a);
    else
      duo = nt_tiles ? (cm ? pick_duo_pattern<true, false, false, false, 0, true>(sec.present_b, sec.present_a)
                           : pick_duo_pattern<false, false, false, false, 0, true>(sec.present_b, sec.present_a))
                     : (cm ? pick_duo_pattern<true, false>(sec.present_b, sec.present_a)
                           : pick_duo_pattern<false, false>(sec.present_b, sec.present_a));
  }
  wave_fn fn = duo;
  if (!fn && nostore)
    fn = g != 64 ? nullptr
                 : cm ? pick_pattern<64, true, true>(sec.present_b, sec.present_a)
                      : pick_pattern<64, false, true>(sec.present_b, sec.present_a);
  else if (!fn)
    fn = pre_abs ? pick_wave<1>(g, cm, sec.present_b, sec.present_a) : pick_wave<0>(g, cm, sec.present_b, sec.present_a);
  if (!fn) return ALZ_OK;

  WArgs p;
  p.x = io.x; p.y = io.y; p.ldx = ldx; p.ldy = ldy;
  p.n_tiles = tiles; p.channels = io.channels; p.c_first = 0;
  p.n_inputs = outer ? io.n_inputs : 0;
  p.map_input = io.map_input;
  p.n_sets = io.n_sets;
  p.nb = sec.nb; p.na = sec.na; p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.yh = sec.yh;
  p.chunk_x = p.chunk_y = p.chunk_c = 0;
  if (ch) {
    p.channels = lanes;                         // stride of the per-chunk state arrays
    p.chunk_c = io.channels;
    p.chunk_x = ch->chunk_len * (cm ? 1 : ldx);
    p.chunk_y = ch->chunk_len * (cm ? 1 : ldy);
    p.xh = ch->vxh; p.yh = ch->vyh;
  }
  static const int dbg_env = ALZ_DBG_ENV();
  p.dbg = dbg_env;
  // The common tile clock (alz_common.h pace_wait), time-major streaming blocks of a launch that fills the chip with one or two
  // resident workgroups per CU: the helper wave requests tile i + 3 no earlier than i x pace after its own start, pace = the
  // time in which the whole launch's tile row (groups x 16 KiB in + out) passes at the rate below.  Free-running, the workgroups
  // drift apart and the rows they touch spread over DRAM pages; a workgroup that is late does not wait, so a slower box or a
  // shared GPU degrades to the free-running rate, not below it.  Measured (GB/s of the clock -> Gsamples/s):
  //  * 256 groups, FMA kernel with the storing wave, 2^20 samples (profiles/r06_duo_tilepace*.log, two boxes): free 303 - 308,
  //    5500 343, 5750 357 - 358, 5850 363, 5900 360 / 328 (the knee), 6000 324; 2^18 samples 351 against 285; 2^14 (256 tiles)
  //    308 - 315 against 318 - 329, 2^16 292 - 295 against 298 - 301, 2^17 338 - 339 against 329 - 335 (the start-up ramp): blocks
  //    under 2048 tiles run free.  Four more boxes (r06_pace_check.log, r06_pace6_feedback_ratchet.log, r06_pace7_forgiving_clock.log):
  //    5750 356 - 359 on two of them, 292 / 348 on one, 303 once in three first runs on another; 6000 never followed.  Overshooting
  //    costs 10 - 18 %, a per-cent of margin 0.9 %: the shipped rate is 5600 (349), 4 - 5 % under the usual knee.
  //  * 256 groups, one-pole banks bit-exact (envelope: |x| -> lowpass; profiles/r06_pace3.log, r06_pace4.log): the paced pass of
  //    round 4 (aux_pace) 305 - 315; the clock INSTEAD of it 5500 343, 5700 356, 5900 366 - 368, 6100 378 - 379; 2^16 samples 328 -
  //    335 against 293 - 296, 2^17 345 against 308, 2^18 353 against 309.
  //  * 256 groups, two-pole banks bit-exact: 327 with or without (the recurrence wave's issue rate bounds them): no clock -- except in
  //    place (no non-temporal tiles: 14.5 ms, above that floor), where a clock just under the floor brings 296 -> 306 - 314.
  //  * 257 - 512 groups (some or all CUs hold two workgroups), 2^19 samples, bit-exact (r06_pace4.log): 4608 channels 215 - 232 ->
  //    224 - 236; 5120 211 - 229 -> 244 - 249 at 4800, 237 - 243 at 5200; 5632 224 - 245 -> 266 - 274 (4800 - 5200), 256 - 263 (5600);
  //    6144 240 - 255 -> 289 - 294 (4800 - 5200); 7168 252 - 269 -> 300 (4800), 310 - 318 (5200), 312 - 320 (5600); 7680 270 - 272 ->
  //    300, 324, 327 - 334.  FMA mode the same: 5120 206 - 219 -> 247 - 251, 7168 270 - 272 -> 309 - 315.  One-pole, 5120: 217 -> 249 - 254.
  //  * whole rounds of 512 groups (16 384, 24 576, 32 768 channels; the clock sized for ONE round, each round starting where the one
  //    before -- kept in step -- ends; r06_duo_rounds.log): 346 - 348 / 347 / 341 bit-exact against k_wave<64>'s 330 / 306 / 319.
  p.tile_pace = 0;
  {
    const int cus = device_cus() > 0 ? device_cus() : 256;
    const bool one_pole = sec.present_b == 1u && sec.present_a == 1u;
    int64_t groups_paced = groups;
    const bool rounds = groups > 2 * cus && groups % (2 * cus) == 0;      // (several full rounds of two workgroups per CU)
    // Blocks of the streaming size that are processed in place or lie between the sections of a cascade (not `stream_once`: the kernels
    // without non-temporal tiles) gain the same way (profiles/r06_pace_inplace.log, in place): one-pole banks 303 -> 359, 6144 channels
    // 233 - 249 -> 269 - 274, 7168 channels in the FMA mode 265 -> 293 - 304.
    const bool big = io.stream_once || (ALZ_TUNE("ALZ_DUO_PACE_ANY", 1) != 0 && (uint64_t)io.n * (uint64_t)io.channels * 8u >= (256ull << 20));
    if (duo && !cm && !ch && big && !sec.any_div && groups >= cus && (groups <= 2 * cus || rounds) && ALZ_TUNE("ALZ_DUO_PACED", 1) != 0) {
      int gbps = 0, min_tiles = 2048;
      if (rounds) groups_paced = 2 * cus, gbps = kDuoPaceGBpsTwo;
      else if (groups > cus) gbps = groups <= cus + 5 * cus / 8 ? kDuoPaceGBpsShared : kDuoPaceGBpsTwo;
      else if (duo_fma) gbps = nt_tiles ? kDuoPaceGBps : 0;
      else if (one_pole) gbps = kDuoPaceGBpsOnePole, min_tiles = 1024;
      else if (!io.stream_once) gbps = kDuoPaceGBpsInPlace;   // two-pole, in place: 14.5 ms free-running, not at the recurrence wave's floor
      gbps = ALZ_TUNE("ALZ_DUO_PACE_GBPS", gbps);
      if (tiles >= ALZ_TUNE("ALZ_DUO_PACE_MIN_TILES", min_tiles)) p.tile_pace = tile_pace16(groups_paced * 16384ll, gbps);
    }
  }
  p.convoy = nullptr; p.convoy_cfg = 0; p.convoy_groups = (unsigned)groups;
  if (p.tile_pace > 0 && ALZ_TUNE("ALZ_CONVOY", 0) > 0) {       // (experiment: the convoy in place of the clock, same launches)
    p.convoy = convoy_ring(stream);
    p.convoy_cfg = ALZ_TUNE("ALZ_CONVOY", 0);
    if (p.convoy && ALZ_TUNE("ALZ_CONVOY_CLOCK", 0) == 0) p.tile_pace = 0;     // (ALZ_CONVOY_CLOCK=1: the clock AND the convoy)
  }
  // One-pole banks that fill the chip once, long blocks NOT on the clock (in place, or under the streaming size): AUX's
  // feed-forward pass in quarters with 2 x 64-cycle pauses.  Measured on three boxes (profiles/r04_duo_patterns.log): 4096
  // channels x 2^20 time-major +2 / +13 / +10 ... 16 %, channel-major +14 / +20 %, 2^18 +1 ... 13 %, 5120 channels +6 %; three
  // pauses are better on one box and worse on the others; blocks of 2^16 and banks of 6144 - 7680 channels lose 1 - 3 %, half a
  // chip of workgroups a quarter: excluded.
  p.aux_pace = (duo && !ch && p.tile_pace == 0 && !p.convoy && groups >= 256 && groups <= 320 && tiles >= 2048 && (!fused || ALZ_DUO_FMA3) && ((sec.present_b == 1u && sec.present_a == 1u) || ALZ_PACE_ALL))
                   ? ALZ_TUNE("ALZ_DUO_AUXPACE", 2) : 0;
  // (-DALZ_TUNING builds: a staggered start of the workgroups of a channel-major launch, stagger_start in alz_common.h -- measured on
  // identical buffers, round 6: no effect; what decides between 12.3 and 16 ms there is where the blocks lie physically)
  p.stagger = (duo && cm && !ch) ? ALZ_TUNE("ALZ_DUO_STAGGER", 0) : 0;
  if (duo && !ch && ALZ_TUNE("ALZ_DUO_TILEPACE", -1) >= 0) p.tile_pace = ALZ_TUNE("ALZ_DUO_TILEPACE", -1);
  if (!ch && !cm && ALZ_TUNE("ALZ_WAVE_PACE_GBPS", 0) > 0) p.tile_pace = tile_pace16(groups * 16384ll, ALZ_TUNE("ALZ_WAVE_PACE_GBPS", 0));   // (any kernel of this file)
  // one wave per workgroup; when the whole launch fits one wave per CU, ask for enough LDS
  // that no two workgroups share a CU (each wave then owns a SIMD and a CU's memory path)
  size_t lds = duo ? (size_t)kXRing * kDuoSlot + (size_t)(kPRing + kYRing) * (cm ? 16 * (64 * 8 + 16) : kDuoSlot)
                   : (size_t)kRing * kSlotBytes;
  const int64_t blocks = groups * (ch ? ch->n_chunks : 1);
  if (blocks <= 256) lds = 96 * 1024;
  {
    const int rc = ensure_dynamic_lds((const void *)fn, 96 * 1024);
    if (rc) return rc;
  }
  if (ch && ch->n_chunks > 65535) return ALZ_OK;
  hipLaunchKernelGGL(fn, dim3((unsigned)groups, (unsigned)(ch ? ch->n_chunks : 1)), dim3(duo ? ((ALZ_DUO_STORER && (!duo_fma || duo_fma_storer(cm, nt_tiles)) && !nostore) ? 192 : 128) : 64), lds,
                     stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *done_samples = tiles * t;
  *done_channels = groups * g;
  *kernel_name = duo ? (sec.any_div ? "k_duo<16,div>" : fused && !nostore ? "k_duo<16,fma>" : "k_duo<16>") : g == 16 ? "k_wave<16>" : g == 32 ? "k_wave<32>" : "k_wave<64>";
  return ALZ_OK;
}

int launch_wave(const SectionDev &sec, const BlockIO &io, hipStream_t stream,
                int64_t *done_samples, int64_t *done_channels, const char **kernel_name) {
  return launch_wave_impl(sec, io, stream, nullptr, done_samples, done_channels, kernel_name);
}

int launch_wave_chunks(const SectionDev &sec, const BlockIO &io, hipStream_t stream, const WaveChunks &ch,
                       bool *taken, const char **kernel_name) {
  int64_t dn = 0, dc = 0;
  const int rc = launch_wave_impl(sec, io, stream, &ch, &dn, &dc, kernel_name);
  *taken = rc == ALZ_OK && dc == io.channels && dn == ch.chunk_len;
  return rc;
}

}  // namespace alz
