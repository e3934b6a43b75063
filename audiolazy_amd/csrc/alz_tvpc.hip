// alz_tvpc.hip -- time-varying biquad-class banks whose coefficient series are PER CHANNEL, on the two-wave
// streaming scheme of k_duo / k_tvduo plus a third wave for the coefficient traffic.
//
// The statement is the reference's generated loop with ``next(b_k) * d_k`` / ``-next(a_k) * m_k`` terms whose
// items are rows of C values -- ``repeat(ndarray)``-style series, one coefficient per channel and sample
// (audiolazy/lazy_filters.py:197-224; examples/formants.py:70-72 steers one resonator per formant):
//
//     y[n, c] = ((b0[n, c] x[n, c] + b1[n, c] x[n-1, c] + b2[n, c] x[n-2, c]) + (-a1[n, c]) y[n-1, c]) + (-a2[n, c]) y[n-2, c]
//
// same order, separately rounded: bit-identical to k_tvp (lane per channel, ~180 cycles per step: 49 - 55
// Gsamples/s at 4096 channels = 25 % of the 40 B/sample this shape moves with three series).  Here a workgroup
// owns 16 channels and three waves split the work of a 64-sample tile:
//
//   SER wave  queues the tile DMA (global_load_lds, 1 KiB per instruction) of every series tap: 8 KiB per tap
//             and tile, into rings three tiles deep -- the 63-entry vmcnt counter of ONE wave cannot hold the x
//             tiles, the stores and up to 40 coefficient transfers per tile;
//   AUX wave  queues the x tile DMA, forms the feed-forward sums p[n] with the per-row, per-channel b_k[n]
//             (time-parallel, all 64 lanes) and stores finished y tiles;
//   REC wave  runs y[n] = (p[n] + na1[n] y[n-1]) + na2[n] y[n-2], reading the step's a_k[n, c] straight from the
//             landed series tiles (unpadded rows, like the p ring) -- ghost lanes, skewed lane groups, one
//             ds_write_b64 per four rows as in k_duo.  The last three rows of a tile's a-series are kept in
//             registers for the lagging lane groups, so a series tile is only needed while its own tile runs.
//
// Time-major blocks and series ([N, C], the reference's vector-valued rows), a0 == 1, 16-channel groups, full
// 64-sample tiles, at most three series taps of which at most two in the denominator (149 KiB of LDS);
// everything else stays on k_tvp, which also continues with the ragged tail from the same state arrays.
#include "alz_common.h"

namespace alz {

namespace {

constexpr int kChunks = 8;
constexpr int kXRing = 4, kPRing = 3, kYRing = 2, kSRing = 3;
constexpr int kSlot = 8192 + kChunks * 16;   // x / p / y / b-series slot: a tile + 16 bytes of pad per 1 KiB chunk
constexpr int kASlot = 8192;                 // a-series slot: unpadded rows of 128 bytes (the recurrence wave's layout)

constexpr int kTvpcPaceGBps = 5600;          // the common tile clock, 5 % under the knee (launch_tvpc)

struct PCArgs {
  const double *x;
  double *y;
  int64_t ldx, ldy, n_tiles, channels;
  int nb, na;
  int kind[5];              // b0 b1 b2 a1 a2: 0 absent, 1 constant, 2 per-channel series
  double value[5];
  const double *series[5];  // element (n, c) at series[k][n * lds_[k] + c]
  int64_t lds_[5];
  int bslot[3], aslot[2];   // ring index of a series tap among the b / a series taps
  double *xh, *yh;
  int tile_pace;            // the common tile clock (alz_common.h pace_wait; 0: free-running)
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

typedef double dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(double *gdst, dbl2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
}

__device__ __forceinline__ void wait_vm(int n) {
#define ALZ_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    ALZ_W(0) ALZ_W(8) ALZ_W(16) ALZ_W(24) ALZ_W(32) ALZ_W(40) ALZ_W(48)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;   // (waiting for a few more is always safe)
  }
#undef ALZ_W
}

}  // namespace

// PB / PA: taps present (b0 b1 b2 / a1 a2); SB / SA: which of them are per-channel series -- compile-time, because a
// run-time "is this tap a series" test in the recurrence wave's step became a branch per LDS read (first version:
// 108 cycles per step, 86 Gsamples/s; profiles/NOTES_r03.md); NEG: the a-series already hold -a_k[n]
// (ALZ_TV_NEGATED: the host negated them in Python arithmetic)
constexpr int pc_bits(unsigned m) { return (int)((m & 1u) + ((m >> 1) & 1u) + ((m >> 2) & 1u)); }
template <unsigned PB, unsigned PA, unsigned SB, unsigned SA, bool NEG>
__global__ __launch_bounds__(192) void k_tvpc(PCArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 16, T = 64;
  constexpr int NSB = pc_bits(SB), NSA = pc_bits(SA);
  static_assert((SB & ~PB) == 0 && (SA & ~PA) == 0 && NSB + NSA >= 1 && NSB + NSA <= 3, "series taps are present taps");
  // ring of a series tap = its rank among the series taps of its side
  constexpr int kBSlot[3] = {0, pc_bits(SB & 1u), pc_bits(SB & 3u)};
  constexpr int kASlot2[2] = {0, pc_bits(SA & 1u)};
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cl = lane & 15, q = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * G;
  const int64_t c = c0 + cl;
  const int64_t nt = p.n_tiles;
  char *xring = smem;
  char *pring = xring + kXRing * kSlot;
  char *yring = pring + kPRing * kSlot;
  char *bring = yring + kYRing * kSlot;              // NSB rings of kSRing slots (DMA layout of the x ring)
  char *aring = bring + NSB * kSRing * kSlot;        // NSA rings of kSRing slots (unpadded rows)
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lane_off = cl * 8;
#define ALZ_EOFF(u) ((u) * G * 8 + (((u) * G) >> 7) * 16)
  constexpr int kStep = G * 8;
  // DMA source of this lane inside a 1 KiB transfer (8 rows of 16 channels): row lane / 8, channel pair lane % 8
  const int drow = lane / 8, dcp = lane % 8;

  if (wave == 2) {
    // ------------------------------ SER: the coefficient tiles ------------------------------
    const double *src[5];
    int64_t tile_step[5], chunk_step[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      src[k] = p.series[k] + (int64_t)drow * p.lds_[k] + c0 + 2 * dcp;
      tile_step[k] = (int64_t)T * p.lds_[k];
      chunk_step[k] = 8 * p.lds_[k];
    }
    // b tile t is read by AUX in interval t - 1, a tile t by REC in interval t, so with rings three deep the slot of
    // b tile t is free from interval t on (-> b tile t + 3) and that of a tile t from interval t + 1 on (-> a tile
    // t + 3): either way a transfer has TWO intervals to land.  At the end of interval i everything queued before
    // this interval must be in (b tile i + 2, a tile i + 1).
    static_assert((NSB + NSA) * kChunks <= 48, "vmcnt range");
    auto queue_b = [&](int64_t t) {
      const unsigned s = (unsigned)(t % kSRing);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if ((SB >> k) & 1u) {
          const unsigned dst = lds0 + (unsigned)(bring - smem) + ((unsigned)kBSlot[k] * kSRing + s) * kSlot;
#pragma unroll
          for (int j = 0; j < kChunks; ++j) dma16(src[k] + t * tile_step[k] + j * chunk_step[k], dst + j * (1024 + 16));
        }
    };
    auto queue_a = [&](int64_t t) {
      const unsigned s = (unsigned)(t % kSRing);
#pragma unroll
      for (int k = 3; k < 5; ++k)
        if ((SA >> (k - 3)) & 1u) {
          const unsigned dst = lds0 + (unsigned)(aring - smem) + ((unsigned)kASlot2[k - 3] * kSRing + s) * kASlot;
#pragma unroll
          for (int j = 0; j < kChunks; ++j) dma16(src[k] + t * tile_step[k] + j * chunk_step[k], dst + j * 1024);
        }
    };
    for (int t = 0; t < 3 && t < nt; ++t) queue_b(t);
    for (int t = 0; t < 2 && t < nt; ++t) queue_a(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // (AUX prepares tile 0)
    __builtin_amdgcn_s_barrier();                            // start of interval 0
    for (int64_t i = 0; i < nt; ++i) {
      int now = 0;
      if (i + 3 < nt) { queue_b(i + 3); now += NSB * kChunks; }
      if (i + 2 < nt) { queue_a(i + 2); now += NSA * kChunks; }
      wait_vm(now);
      __builtin_amdgcn_s_barrier();
    }
  } else if (wave == 1) {
    // ------------------------------ AUX ------------------------------
    const int row = lane / 8, cp = lane % 8;
    const int64_t x_off = (int64_t)row * p.ldx + c0 + 2 * cp, y_off = (int64_t)row * p.ldy + c0 + 2 * cp;
    const int64_t x_chunk = 8 * p.ldx, y_chunk = 8 * p.ldy;
    const int64_t x_tile = (int64_t)T * p.ldx, y_tile = (int64_t)T * p.ldy;
    double d1 = (p.nb > 1) ? p.xh[0 * p.channels + c] : 0.0;   // x[-1], x[-2] of the stream
    double d2 = (p.nb > 2) ? p.xh[1 * p.channels + c] : 0.0;
    asm volatile("" : "+v"(d1), "+v"(d2));
    const double *xg = p.x + x_off;
    double *yg = p.y + y_off;
    auto queue_tile = [&](int64_t t) {
      const int s = (int)(t % kXRing);
#pragma unroll
      for (int j = 0; j < kChunks; ++j) dma16(xg + t * x_tile + j * x_chunk, lds0 + s * kSlot + j * (1024 + 16));
    };
    constexpr bool ser0 = (SB & 1u) != 0, ser1 = (SB & 2u) != 0, ser2 = (SB & 4u) != 0;
    // the 16 coefficients of tap k for this lane's rows 4j + q: from the landed series tile (same addressing as x)
    // or the constant
    auto fill16 = [&](auto is_series, auto kk, int64_t t, double (&out)[16]) {
      constexpr int k = decltype(kk)::value;
      if constexpr (decltype(is_series)::value) {
        const char *bs = bring + (kBSlot[k] * kSRing + (int)(t % kSRing)) * kSlot + lane_off + q * kStep;
#pragma unroll
        for (int j = 0; j < 16; ++j) out[j] = *reinterpret_cast<const double *>(bs + ALZ_EOFF(4 * j));
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) out[j] = p.value[k];
      }
    };
    auto prepare_tile = [&](int64_t t) {
      const char *xs = xring + (int)(t % kXRing) * kSlot + lane_off;
      const char *xp = xring + (int)((t + kXRing - 1) % kXRing) * kSlot + lane_off;  // tile t-1
      char *ps = pring + (int)(t % kPRing) * kSlot + lane_off;
      const int adj1 = (q == 0) ? 16 : 0, adj2 = (q < 2) ? 16 : 0;
      const char *x_d0 = xs + q * kStep;
      const char *x_d1[2] = {xs + (q - 1) * kStep - adj1, xs + (q - 1) * kStep};   // [j odd]
      const char *x_d2[2] = {xs + (q - 2) * kStep - adj2, xs + (q - 2) * kStep};
      double x0[16], x1[16], x2[16], cb0[16], cb1[16], cb2[16];
      if constexpr (PB & 1u) fill16(std::integral_constant<bool, ser0>{}, std::integral_constant<int, 0>{}, t, cb0);
      if constexpr (PB & 2u) fill16(std::integral_constant<bool, ser1>{}, std::integral_constant<int, 1>{}, t, cb1);
      if constexpr (PB & 4u) fill16(std::integral_constant<bool, ser2>{}, std::integral_constant<int, 2>{}, t, cb2);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if constexpr (PB & 1u) x0[j] = *reinterpret_cast<const double *>(x_d0 + ALZ_EOFF(4 * j));
        if constexpr (PB & 2u) {
          if (j > 0) x1[j] = *reinterpret_cast<const double *>(x_d1[j & 1] + ALZ_EOFF(4 * j));
        }
        if constexpr (PB & 4u) {
          if (j > 0) x2[j] = *reinterpret_cast<const double *>(x_d2[j & 1] + ALZ_EOFF(4 * j));
        }
      }
      if constexpr ((PB & 6u) != 0) {
        double pm1, pm2;                        // x[-1], x[-2] relative to this tile
        if (t > 0) {
          pm1 = *reinterpret_cast<const double *>(xp + ALZ_EOFF(T - 1));
          pm2 = *reinterpret_cast<const double *>(xp + ALZ_EOFF(T - 2));
        } else {
          pm1 = d1;
          pm2 = d2;
        }
        const double s0 = *reinterpret_cast<const double *>(xs + ALZ_EOFF(0));
        const double s1 = *reinterpret_cast<const double *>(xs + ALZ_EOFF(1));
        const double s2 = *reinterpret_cast<const double *>(xs + ALZ_EOFF(2));
        if constexpr (PB & 2u) x1[0] = q == 0 ? pm1 : q == 1 ? s0 : q == 2 ? s1 : s2;
        if constexpr (PB & 4u) x2[0] = q == 0 ? pm2 : q == 1 ? pm1 : q == 2 ? s0 : s1;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double acc = 0.0;
        bool first = true;
        if constexpr (PB & 1u) { acc = cb0[j] * x0[j]; first = false; }
        if constexpr (PB & 2u) { const double v = cb1[j] * x1[j]; acc = first ? v : acc + v; first = false; }
        if constexpr (PB & 4u) { const double v = cb2[j] * x2[j]; acc = first ? v : acc + v; first = false; }
        *reinterpret_cast<double *>(ps + (4 * j + q) * kStep) = acc;       // (p / y rings: unpadded rows)
      }
    };
    auto store_tile = [&](int64_t t) {
      const char *ys = yring + (int)(t % kYRing) * kSlot;
      double *yt = yg + t * y_tile;
      dbl2 v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j) v[j] = *reinterpret_cast<const dbl2 *>(ys + j * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < kChunks; ++j) store16(yt + j * y_chunk, v[j]);
    };

    for (int t = 0; t < kXRing - 1 && t < nt; ++t) queue_tile(t);
    wait_vm((int)((nt < kXRing - 1 ? nt : kXRing - 1) - 1) * kChunks);   // x tile 0 has landed
    __builtin_amdgcn_s_barrier();                            // ... and so has series tile 0 (SER)
    prepare_tile(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // start of interval 0: p of tile 0 ready, series tile 1 in
    const long long pace0 = p.tile_pace > 0 ? (long long)wall_clock64() : 0;

    long long pace_shift = 0;
    for (int64_t i = 0; i < nt; ++i) {
      if (i >= 1) store_tile(i - 1);
      if (p.tile_pace > 0) pace_wait(pace0, i, p.tile_pace, pace_shift);
      if (i + kXRing - 1 < nt) queue_tile(i + kXRing - 1);
      if (i + 1 < nt) {
        const int64_t last = (i + kXRing - 1 < nt - 1) ? i + kXRing - 1 : nt - 1;
        const int64_t loads_after = last - (i + 1);
        const int64_t stores_after = i < kXRing - 2 ? i : kXRing - 2;
        wait_vm((int)(loads_after + stores_after) * kChunks);
        prepare_tile(i + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    store_tile(nt - 1);
    if (q == 3) {      // input history for the next block: the last two x samples
      const char *xs = xring + (int)((nt - 1) % kXRing) * kSlot + lane_off;
      if (p.nb > 1) p.xh[0 * p.channels + c] = *reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 1));
      if (p.nb > 2) p.xh[1 * p.channels + c] = *reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 2));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ------------------------------ REC ------------------------------
    double m1 = (p.na > 1) ? p.yh[0 * p.channels + c] : 0.0;
    double m2 = (p.na > 2) ? p.yh[1 * p.channels + c] : 0.0;
    constexpr bool sa1 = (SA & 1u) != 0, sa2 = (SA & 2u) != 0;
    double k1 = (PA & 1u) ? -p.value[3] : 0.0, k2 = (PA & 2u) ? -p.value[4] : 0.0;    // constant taps: -a_k
    asm volatile("" : "+v"(m1), "+v"(m2), "+v"(k1), "+v"(k2));
    // a_k of the previous tile's last three rows (rows T - 3, T - 2, T - 1), for the lagging groups
    double k1a = 0.0, k1b = 0.0, k1c = 0.0, k2a = 0.0, k2b = 0.0, k2c = 0.0;
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                            // p of tile 0 and series tile 0 are ready
    constexpr int NCH = T / 8;
    int ps_cur = 0, ps_prv = kPRing - 1, ys_cur = 0, as_cur = 0;
    for (int64_t i = 0; i < nt; ++i) {
      // this lane works on row (u - q) of the tile; u - q < 0 lives in the previous tile (p: its ring slot; a_k: keep)
      const char *cur = pring + ps_cur * kSlot + lane_off - q * kStep;
      const char *prv = pring + ps_prv * kSlot + lane_off + (T - q) * kStep;
      const char *a1c = aring + (kASlot2[0] * kSRing + as_cur) * kASlot + lane_off - q * kStep;
      const char *a2c = aring + (kASlot2[1] * kSRing + as_cur) * kASlot + lane_off - q * kStep;
      char *wr = yring + ys_cur * kSlot + lane_off - q * kStep;
      ps_prv = ps_cur;
      ps_cur = (ps_cur + 1 == kPRing) ? 0 : ps_cur + 1;
      ys_cur = (ys_cur + 1 == kYRing) ? 0 : ys_cur + 1;
      as_cur = (as_cur + 1 == kSRing) ? 0 : as_cur + 1;
      constexpr int R1 = sa1 ? 3 : 1, R2 = sa2 ? 3 : 1;   // (a constant tap has no coefficient buffers)
      double pr[3][8], c1[R1][8], c2[R2][8];
      // kept row T - 3 + j for a per-lane j (registers cannot be indexed by a lane value: selects)
      auto kept = [](double a, double b, double c, int j) { return j <= 0 ? a : j == 1 ? b : c; };
      auto load_chunk = [&](int k, double (&pp)[8], double (&cc1)[8], double (&cc2)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k == 0 && u < 3) {
            const bool before = u < q;                           // (per-lane; u >= 3 never)
            pp[u] = *reinterpret_cast<const double *>((before ? prv : cur) + u * kStep);
            if constexpr (sa1) {
              { const double v = *reinterpret_cast<const double *>(a1c + (before ? q : 8 * k + u) * kStep); cc1[u] = before ? kept(k1a, k1b, k1c, u + 3 - q) : v; }
            }
            if constexpr (sa2) {
              { const double v = *reinterpret_cast<const double *>(a2c + (before ? q : 8 * k + u) * kStep); cc2[u] = before ? kept(k2a, k2b, k2c, u + 3 - q) : v; }
            }
          } else {
            pp[u] = *reinterpret_cast<const double *>(cur + (8 * k + u) * kStep);
            if constexpr (sa1) cc1[u] = *reinterpret_cast<const double *>(a1c + (8 * k + u) * kStep);
            if constexpr (sa2) cc2[u] = *reinterpret_cast<const double *>(a2c + (8 * k + u) * kStep);
          }
        }
      };
      load_chunk(0, pr[0], c1[0], c2[0]);
      load_chunk(1, pr[1], c1[1 % R1], c2[1 % R2]);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k + 2 < NCH) load_chunk(k + 2, pr[(k + 2) % 3], c1[(k + 2) % R1], c2[(k + 2) % R2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          double acc = pr[k % 3][u];
          if constexpr (PA & 1u) {
            const double co = sa1 ? (NEG ? c1[k % R1][u] : -c1[k % R1][u]) : k1;
            acc = acc + co * m1;
          }
          if constexpr (PA & 2u) {
            const double co = sa2 ? (NEG ? c2[k % R2][u] : -c2[k % R2][u]) : k2;
            acc = acc + co * m2;
          }
          if (k == 0 && u < 3 && i == 0) {
            // start of the stream: group q has nothing to do before step q; hold its state
            const bool on = u >= q;
            m2 = on ? m1 : m2;
            m1 = on ? acc : m1;
          } else {
            m2 = m1;
            m1 = acc;
          }
          if ((u & 3) == 3) *reinterpret_cast<double *>(wr + (8 * k + u) * kStep) = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // a_k of this tile's last three rows, for the lagging groups' first steps of the next tile:
      // keep[j] = a_k[row T - 3 + j]
      if constexpr (NSA > 0) {
        if constexpr (sa1) {
          k1a = *reinterpret_cast<const double *>(a1c + (q + T - 3) * kStep);
          k1b = *reinterpret_cast<const double *>(a1c + (q + T - 2) * kStep);
          k1c = *reinterpret_cast<const double *>(a1c + (q + T - 1) * kStep);
        }
        if constexpr (sa2) {
          k2a = *reinterpret_cast<const double *>(a2c + (q + T - 3) * kStep);
          k2b = *reinterpret_cast<const double *>(a2c + (q + T - 2) * kStep);
          k2c = *reinterpret_cast<const double *>(a2c + (q + T - 1) * kStep);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // y of tile i done, p of tile i+1 ready, series tile i+2 in
    }
    if (lane < G) {
      if (p.na > 1) p.yh[0 * p.channels + c] = m1;
      if (p.na > 2) p.yh[1 * p.channels + c] = m2;
    }
  }
#undef ALZ_EOFF
}

typedef void (*tvpc_fn)(PCArgs);

// (present taps, series taps) combinations with an instantiation: every split of one to three series taps over the
// biquad-class patterns (two at most on the two densest ones); anything else stays on the lane-per-channel kernel
static tvpc_fn pick_tvpc(unsigned pb, unsigned pa, unsigned sb, unsigned sa, bool neg) {
#define ALZ_PC(PB_, PA_, SB_, SA_)                                                                   \
  if (pb == PB_ && pa == PA_ && sb == SB_ && sa == SA_)                                              \
    return (SA_ != 0 && neg) ? (tvpc_fn)k_tvpc<PB_, PA_, SB_, SA_, (SA_ != 0)> : (tvpc_fn)k_tvpc<PB_, PA_, SB_, SA_, false>;
  ALZ_PC(1, 1, 0, 1) ALZ_PC(1, 1, 1, 0) ALZ_PC(1, 1, 1, 1) ALZ_PC(3, 1, 0, 1)
  ALZ_PC(3, 1, 1, 0) ALZ_PC(3, 1, 1, 1) ALZ_PC(3, 1, 2, 0) ALZ_PC(3, 1, 2, 1)
  ALZ_PC(3, 1, 3, 0) ALZ_PC(3, 1, 3, 1) ALZ_PC(1, 3, 0, 1) ALZ_PC(1, 3, 0, 2)
  ALZ_PC(1, 3, 0, 3) ALZ_PC(1, 3, 1, 0) ALZ_PC(1, 3, 1, 1) ALZ_PC(1, 3, 1, 2)
  ALZ_PC(1, 3, 1, 3) ALZ_PC(3, 3, 0, 1) ALZ_PC(3, 3, 0, 2) ALZ_PC(3, 3, 0, 3)
  ALZ_PC(3, 3, 1, 0) ALZ_PC(3, 3, 1, 1) ALZ_PC(3, 3, 1, 2) ALZ_PC(3, 3, 1, 3)
  ALZ_PC(3, 3, 2, 0) ALZ_PC(3, 3, 2, 1) ALZ_PC(3, 3, 2, 2) ALZ_PC(3, 3, 3, 0)
  ALZ_PC(5, 3, 0, 1) ALZ_PC(5, 3, 0, 2) ALZ_PC(5, 3, 0, 3) ALZ_PC(5, 3, 1, 0)
  ALZ_PC(5, 3, 1, 1) ALZ_PC(5, 3, 1, 2) ALZ_PC(5, 3, 1, 3) ALZ_PC(5, 3, 4, 0)
  ALZ_PC(5, 3, 4, 1) ALZ_PC(5, 3, 4, 2) ALZ_PC(5, 3, 4, 3) ALZ_PC(5, 3, 5, 0)
  ALZ_PC(5, 3, 5, 1) ALZ_PC(5, 3, 5, 2) ALZ_PC(7, 3, 0, 1) ALZ_PC(7, 3, 0, 2)
  ALZ_PC(7, 3, 0, 3) ALZ_PC(7, 3, 1, 0) ALZ_PC(7, 3, 1, 1) ALZ_PC(7, 3, 1, 2)
  ALZ_PC(7, 3, 1, 3) ALZ_PC(7, 3, 2, 0) ALZ_PC(7, 3, 2, 1) ALZ_PC(7, 3, 2, 2)
  ALZ_PC(7, 3, 3, 0) ALZ_PC(7, 3, 4, 0) ALZ_PC(7, 3, 4, 1) ALZ_PC(7, 3, 4, 2)
  ALZ_PC(7, 3, 5, 0) ALZ_PC(7, 3, 6, 0)
#undef ALZ_PC
  return nullptr;
}

// The part of a time-varying block with per-channel series the three-wave kernel can take: *done_samples full
// 64-row tiles of all channels (0: not this kernel's shape).  taps: b0 b1 b2 a1 a2 as (kind, value, series, row stride
// of the series, negated); kind 2 = per-channel series [n][channels-wide rows].
int launch_tvpc(const double *x, double *y, int64_t n, int64_t ldx, int64_t ldy, int64_t channels, int nb, int na,
                const int *kind, const double *value, const double *const *series, const int64_t *series_ld,
                const int *negated, double *xh, double *yh, hipStream_t stream, int64_t *done_samples) {
  *done_samples = 0;
  if (nb > 3 || na > 3 || channels % 16 || channels / 16 > 512 || n < 64) return ALZ_OK;
  if ((((uintptr_t)x | (uintptr_t)y) & 15) || ((ldx | ldy) & 1)) return ALZ_OK;
  unsigned pb = 0, pa = 0;
  for (int k = 0; k < 3; ++k) pb |= (unsigned)(kind[k] != 0) << k;
  for (int k = 3; k < 5; ++k) pa |= (unsigned)(kind[k] != 0) << (k - 3);
  PCArgs p;
  int nsb = 0, nsa = 0, negs = 0;
  for (int k = 0; k < 5; ++k) {
    p.kind[k] = kind[k]; p.value[k] = value[k]; p.series[k] = series[k] ? series[k] : x; p.lds_[k] = series_ld[k];
    if (kind[k] == 2) {
      if (((uintptr_t)series[k] & 15) || (series_ld[k] & 1)) return ALZ_OK;      // 16-byte pieces
      if (k < 3) p.bslot[k] = nsb++;
      else { p.aslot[k - 3] = nsa++; negs += negated[k] ? 1 : 0; }
    } else {
      if (k < 3) p.bslot[k] = 0;
      else p.aslot[k - 3] = 0;
    }
  }
  if (nsb + nsa == 0 || nsb + nsa > 3) return ALZ_OK;
  if (negs != 0 && negs != nsa) return ALZ_OK;               // (one sign convention per call)
  unsigned sb = 0, sa = 0;
  for (int k = 0; k < 3; ++k) sb |= (unsigned)(kind[k] == 2) << k;
  for (int k = 3; k < 5; ++k) sa |= (unsigned)(kind[k] == 2) << (k - 3);
  tvpc_fn fn = pick_tvpc(pb, pa, sb, sa, negs != 0);
  if (!fn) return ALZ_OK;
  p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy; p.n_tiles = n / 64; p.channels = channels;
  p.nb = nb; p.na = na; p.xh = xh; p.yh = yh;
  // The workgroups on one tile clock (alz_common.h pace_wait): a tile step of the launch moves 16 x 64 x 8 B x (x, y and the series)
  // per group.  4096 channels x 2^18, three series, two runs (profiles/r06_pace_others.log, r06_pace2.log): free-running 112 - 115
  // Gsamples/s, 5200 GB/s 130, 5500 137, 5800 138 - 144 (0.72 of 8 TB/s), 6000 118 / 136 (the knee), 6200 and more 110 - 111; every
  // block length gains (2^14 +21 %, 2^16 +25 %, 2^19 +26 %); 2048 channels +- 0.  The clock only means something while all groups
  // are resident (one workgroup per CU: ~147 KiB of LDS): 320 and 512 groups on a clock sized for all of them lost 20 - 37 %.  Sized
  // for ONE round of workgroups instead (paced_groups_one_per_cu: each round starts where the one before, kept in step, ends and
  // paces itself from its own start) launches of full rounds gain as well: 8192 channels 114 - 119 -> 139.5 - 139.8, 7680 channels
  // (a last round of 224) 113 - 115 -> 130 - 131, 12288 channels 112 - 115 -> 115 - 118; launches with a short last round (5120, 6144
  // channels) run free (profiles/r06_pace_rounds.log).
  const int cus = device_cus() > 0 ? device_cus() : 256;
  p.tile_pace = tile_pace16(paced_groups_one_per_cu(channels / 16, cus) * 8192ll * (2 + nsb + nsa), ALZ_TUNE("ALZ_TVPC_PACE_GBPS", kTvpcPaceGBps));
  const size_t lds = (size_t)(kXRing + kPRing + kYRing) * kSlot + (size_t)nsb * kSRing * kSlot + (size_t)nsa * kSRing * kASlot;
  const int rc = ensure_dynamic_lds((const void *)fn, (int)lds);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)(channels / 16)), dim3(192), lds, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *done_samples = p.n_tiles * 64;
  return ALZ_OK;
}

}  // namespace alz
