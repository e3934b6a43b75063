// alz_tvduo.hip -- time-varying biquad-class filters whose coefficient series are SHARED by the bank,
// on the two-wave streaming scheme of k_duo (alz_wave.hip).
//
// Same statement as k_tvp / the reference's generated loop with ``next(b_k) * d_k`` and
// ``-next(a_k) * m_k`` terms (audiolazy/lazy_filters.py:197-224), same order, separately rounded:
//
//     y[n] = ((b0[n] x[n] + b1[n] x[n-1] + b2[n] x[n-2]) + (-a1[n]) y[n-1]) + (-a2[n]) y[n-2]
//
// k_tvp runs it lane-per-channel: ~40 instructions per step, and throughput is channels x step rate,
// so a bank of a few thousand channels (64 waves of 1024 SIMDs busy) sits at 55 Gsamples/s.  A
// control stream that steers a whole bank -- resonator.z_exp(Stream(freqs), bw) over thousands of
// channels -- has ONE value per tap and step for every channel, which is what makes k_duo's split
// possible here too:
//
//   AUX wave  queues the x tile DMA and, with it, the tile's 64 coefficient values of every series
//             tap (global_load_lds, two taps per 1 KiB transfer); when a tile has landed it forms the
//             feed-forward sums p[n] with the per-row b_k[n] (time-parallel, all 64 lanes), writes
//             the pairs (-a1[n], -a2[n]) of the tile to a small LDS ring;
//   STORE wave stores finished y tiles (round 3: out of the AUX wave's interval, as in k_duo);
//   REC wave  runs y[n] = (p[n] + na1[n] y[n-1]) + na2[n] y[n-2]: one 16-byte LDS read for the step's
//             coefficient pair on top of k_duo's recurrence (ghost lanes, skewed lane groups, one
//             ds_write_b64 per four rows).
//
// Constant taps are allowed next to series taps (they are written into the same rings).  Both
// layouts, a0 == 1, 16-channel groups, full 64-sample tiles; everything else stays on k_tvp (the ragged
// tail of a block continues there from the same state arrays).
#include "alz_common.h"

namespace alz {

namespace {

constexpr int kChunks = 8;
constexpr int kXRing = 4, kPRing = 3, kYRing = 2;
constexpr int kSlot = 8192 + kChunks * 16;   // x / p / y ring slot: a tile + 16 bytes of pad per 1 KiB chunk
constexpr int kRawSlot = 3 * 1024;           // raw series values of one tile: up to five taps x 512 B
constexpr int kPairSlot = 1024;              // (na1, na2) of the 64 rows of a tile

struct TDArgs {
  const double *x;
  double *y;
  int64_t ldx, ldy, n_tiles, channels;
  int nb, na;
  int kind[5];              // b0 b1 b2 a1 a2: 0 absent, 1 constant, 2 shared series
  double value[5];
  const double *series[5];  // value of sample n at series[k][n]
  int negated[5];           // a-series that already holds -a_k[n]
  int slot_of[5];           // position of a series tap among the series taps (its 512-byte piece of the raw slot)
  int n_dma;                // 1 KiB coefficient transfers per tile = ceil(series taps / 2)
  const double *dma_src[3][2];
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

// wait until at most n vector-memory operations of this wave are outstanding (n <= 63; a literal per case)
__device__ __forceinline__ void wait_vm(int n) {
#define ALZ_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    ALZ_W(0) ALZ_W(1) ALZ_W(2) ALZ_W(3) ALZ_W(4) ALZ_W(5) ALZ_W(6) ALZ_W(7) ALZ_W(8) ALZ_W(9)
    ALZ_W(10) ALZ_W(11) ALZ_W(12) ALZ_W(13) ALZ_W(14) ALZ_W(15) ALZ_W(16) ALZ_W(17) ALZ_W(18) ALZ_W(19)
    ALZ_W(20) ALZ_W(21) ALZ_W(22) ALZ_W(23) ALZ_W(24) ALZ_W(25) ALZ_W(26) ALZ_W(27) ALZ_W(28) ALZ_W(29)
    ALZ_W(30) ALZ_W(31) ALZ_W(32) ALZ_W(33) ALZ_W(34) ALZ_W(35) ALZ_W(36) ALZ_W(37) ALZ_W(38) ALZ_W(39)
    ALZ_W(40) ALZ_W(41) ALZ_W(42) ALZ_W(43) ALZ_W(44) ALZ_W(45) ALZ_W(46) ALZ_W(47) ALZ_W(48)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;   // (waiting for a few more is always safe)
  }
#undef ALZ_W
}

template <int N>
__device__ __forceinline__ void wait_vm_literal() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

}  // namespace

// ND: 1 KiB coefficient transfers per tile (compile-time so that the steady-state s_waitcnt is a literal
// and the queueing loop is straight-line code: a run-time count cost ~100 scalar instructions per tile)
// CM: channel-major blocks ([C, N], one Stream per row), k_duo's layout: a 1 KiB DMA chunk is two channels
// of 512 B with a 16-byte pad, the p / y rings keep 16 bytes after every channel.
template <unsigned PB, unsigned PA, int ND, bool CM>
__global__ __launch_bounds__(192) void k_tvduo(TDArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 16, T = 64;
  constexpr int kChanPitch = 64 * 8 + 16;                  // bytes per channel row of the p / y rings (CM)
  constexpr int kPYSlot = CM ? 16 * kChanPitch : kSlot;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cl = lane & 15, q = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * G;
  const int64_t c = c0 + cl;
  const int64_t nt = p.n_tiles;
  char *xring = smem;
  char *pring = smem + kXRing * kSlot;
  char *yring = pring + kPRing * kPYSlot;
  char *rawring = yring + kYRing * kPYSlot;        // kXRing slots: lands with the x tile of the same index
  char *pairring = rawring + kXRing * kRawSlot;    // kPRing slots: written with the p tile of the same index
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lane_off = CM ? cl * T * 8 + ((cl * T) >> 7) * 16 : cl * 8;    // x ring (DMA layout)
  const int lane_off_p = CM ? cl * kChanPitch : cl * 8;                     // p / y rings
#define ALZ_EOFF(u) (CM ? (u) * 8 : (u) * G * 8 + (((u) * G) >> 7) * 16)
  constexpr int kStep = CM ? 8 : G * 8;

  if (wave >= 1) {
    // ------------------------------ AUX (wave 1) and the storing wave (wave 2, as in k_duo) ------------------------------
    int64_t x_off, y_off, x_chunk, y_chunk, x_tile, y_tile;
    if (!CM) {
      const int row = lane / 8, cp = lane % 8;
      x_off = (int64_t)row * p.ldx + c0 + 2 * cp;
      y_off = (int64_t)row * p.ldy + c0 + 2 * cp;
      x_chunk = 8 * p.ldx; y_chunk = 8 * p.ldy;
      x_tile = (int64_t)T * p.ldx; y_tile = (int64_t)T * p.ldy;
    } else {
      const int ch = lane / 32, sp = lane % 32;
      x_off = (c0 + ch) * p.ldx + 2 * sp;
      y_off = (c0 + ch) * p.ldy + 2 * sp;
      x_chunk = 2 * p.ldx; y_chunk = 2 * p.ldy;
      x_tile = T; y_tile = T;
    }
    double d1 = (p.nb > 1) ? p.xh[0 * p.channels + c] : 0.0;   // x[-1], x[-2] of the stream
    double d2 = (p.nb > 2) ? p.xh[1 * p.channels + c] : 0.0;
    asm volatile("" : "+v"(d1), "+v"(d2));
    const double *xg = p.x + x_off;
    double *yg = p.y + y_off;
    constexpr int per_tile = kChunks + ND;          // vector-memory loads queued per tile
    // this lane's source inside a coefficient transfer: lanes 0..31 the even tap, 32..63 the odd one
    const int half = lane >> 5, piece = lane & 31;

    const double *coef_src[ND > 0 ? ND : 1];
#pragma unroll
    for (int d = 0; d < ND; ++d) coef_src[d] = (half ? p.dma_src[d][1] : p.dma_src[d][0]) + 2 * piece;
    auto queue_tile = [&](int64_t t) {
      const int s = (int)(t % kXRing);
#pragma unroll
      for (int j = 0; j < kChunks; ++j) dma16(xg + t * x_tile + j * x_chunk, lds0 + s * kSlot + j * (1024 + 16));
#pragma unroll
      for (int d = 0; d < ND; ++d)
        dma16(coef_src[d] + t * T, lds0 + (unsigned)(rawring - smem) + s * kRawSlot + d * 1024);
    };
    // values of tap k (0..4) in this tile: a constant, or 64 doubles of the raw slot.  The pointer is
    // typed as LDS and the constant / series decision is one wave-uniform branch per tap and tile: a
    // select between an LDS load and a kernel argument made hipcc fall back to flat loads, whose
    // waits drain the DMA ring (first version: 843 instructions per tile, 109 Gsamples/s)
    typedef const double __attribute__((address_space(3))) *lds_cd;
    const bool ser0 = p.kind[0] == 2, ser1 = p.kind[1] == 2, ser2 = p.kind[2] == 2, ser3 = p.kind[3] == 2, ser4 = p.kind[4] == 2;
    const unsigned raw0 = (unsigned)(rawring - smem) + lds0;
    auto fill16 = [&](bool is_series, int k, unsigned raw_slot, double (&out)[16]) {
      if (is_series) {
        lds_cd r = (lds_cd)(uintptr_t)(raw_slot + (unsigned)p.slot_of[k] * 512u + (unsigned)q * 8u);
#pragma unroll
        for (int j = 0; j < 16; ++j) out[j] = r[4 * j];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) out[j] = p.value[k];
      }
    };
    auto row_value = [&](bool is_series, int k, unsigned raw_slot) -> double {
      if (is_series) {
        lds_cd r = (lds_cd)(uintptr_t)(raw_slot + (unsigned)p.slot_of[k] * 512u);
        return r[lane];
      }
      return p.value[k];
    };
    auto prepare_tile = [&](int64_t t) {
      const char *xs = xring + (int)(t % kXRing) * kSlot + lane_off;
      const char *xp = xring + (int)((t + kXRing - 1) % kXRing) * kSlot + lane_off;  // tile t-1
      const unsigned raw = raw0 + (unsigned)(t % kXRing) * kRawSlot;
      char *ps = pring + (int)(t % kPRing) * kPYSlot + lane_off_p;
      // (-a1[n], -a2[n]) of row n = lane, for the recurrence wave
      {
        dbl2 pr;
        pr.x = 0.0; pr.y = 0.0;
        if constexpr (PA & 1u) { const double v = row_value(ser3, 3, raw); pr.x = p.negated[3] ? v : -v; }
        if constexpr (PA & 2u) { const double v = row_value(ser4, 4, raw); pr.y = p.negated[4] ? v : -v; }
        *reinterpret_cast<dbl2 *>(pairring + (int)(t % kPRing) * kPairSlot + lane * 16) = pr;
      }
      // feed-forward: lane (q, cl) owns rows 4j + q (j = 0..15) of channel cl
      const int adj1 = (!CM && q == 0) ? 16 : 0, adj2 = (!CM && q < 2) ? 16 : 0;
      const char *x_d0 = xs + q * kStep;
      const char *x_d1[2] = {xs + (q - 1) * kStep - adj1, xs + (q - 1) * kStep};   // [j odd]
      const char *x_d2[2] = {xs + (q - 2) * kStep - adj2, xs + (q - 2) * kStep};
      double x0[16], x1[16], x2[16], cb0[16], cb1[16], cb2[16];
      if constexpr (PB & 1u) fill16(ser0, 0, raw, cb0);
      if constexpr (PB & 2u) fill16(ser1, 1, raw, cb1);
      if constexpr (PB & 4u) fill16(ser2, 2, raw, cb2);
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
        *reinterpret_cast<double *>(ps + (4 * j + q) * kStep) = acc;
      }
    };
    auto store_tile = [&](int64_t t) {
      const char *ys = yring + (int)(t % kYRing) * kPYSlot;
      double *yt = yg + t * y_tile;
      dbl2 v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        v[j] = *reinterpret_cast<const dbl2 *>(ys + (CM ? (2 * j + lane / 32) * kChanPitch + (lane % 32) * 16
                                                          : j * 1024 + lane * 16));
#pragma unroll
      for (int j = 0; j < kChunks; ++j) store16(yt + j * y_chunk, v[j]);
    };

    if (wave == 2) {
      // the storing wave: tile i - 1 while REC works on tile i (in the AUX wave the LDS round trip and the store issues
      // lengthened the interval every barrier waits for)
      __builtin_amdgcn_s_barrier();
      for (int64_t i = 0; i < nt; ++i) {
        if (i >= 1) store_tile(i - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      store_tile(nt - 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    for (int t = 0; t < kXRing - 1 && t < nt; ++t) queue_tile(t);
    wait_vm((int)((nt < kXRing - 1 ? nt : kXRing - 1) - 1) * per_tile);   // tile 0 has landed
    prepare_tile(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long pace0 = p.tile_pace > 0 ? (long long)wall_clock64() : 0;

    long long pace_shift = 0;
    for (int64_t i = 0; i < nt; ++i) {
      if (p.tile_pace > 0) pace_wait(pace0, i, p.tile_pace, pace_shift);
      if (i + kXRing - 1 < nt) queue_tile(i + kXRing - 1);
      if (i + 1 < nt) {
        // operations issued after tile i+1's loads: the loads of tiles i+2 .. i+kXRing-1
        if (i + kXRing - 1 < nt) {                              // steady state: two tiles of loads
          constexpr int kSteady = (kXRing - 2) * per_tile;
          static_assert(kSteady <= 48, "vmcnt range");
          wait_vm_literal<kSteady>();
        } else {
          const int64_t last = nt - 1;
          const int64_t loads_after = last - (i + 1);
          wait_vm((int)(loads_after * per_tile));
        }
        prepare_tile(i + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
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
    asm volatile("" : "+v"(m1), "+v"(m2));
    __builtin_amdgcn_s_barrier();                            // p and the coefficient pairs of tile 0 are ready
    constexpr int NCH = T / 8;
    int ps_cur = 0, ps_prv = kPRing - 1, ys_cur = 0;
    for (int64_t i = 0; i < nt; ++i) {
      // this lane works on row (u - q) of the tile; u - q < 0 lives in the previous tile's slots
      const char *cur = pring + ps_cur * kPYSlot + lane_off_p - q * kStep;
      const char *prv = pring + ps_prv * kPYSlot + lane_off_p + (T - q) * kStep;
      const char *ccur = pairring + ps_cur * kPairSlot - q * 16;
      const char *cprv = pairring + ps_prv * kPairSlot + (T - q) * 16;
      char *wr = yring + ys_cur * kPYSlot + lane_off_p - q * kStep;
      ps_prv = ps_cur;
      ps_cur = (ps_cur + 1 == kPRing) ? 0 : ps_cur + 1;
      ys_cur = (ys_cur + 1 == kYRing) ? 0 : ys_cur + 1;
      double pr[3][8];
      dbl2 cq[3][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool before = u < 3 && u < q;                   // (u < q is per-lane; u >= 3 never)
        pr[0][u] = *reinterpret_cast<const double *>((before ? prv : cur) + u * kStep);
        cq[0][u] = *reinterpret_cast<const dbl2 *>((before ? cprv : ccur) + u * 16);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        pr[1][u] = *reinterpret_cast<const double *>(cur + (8 + u) * kStep);
        cq[1][u] = *reinterpret_cast<const dbl2 *>(ccur + (8 + u) * 16);
      }
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k + 2 < NCH) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pr[(k + 2) % 3][u] = *reinterpret_cast<const double *>(cur + ((k + 2) * 8 + u) * kStep);
            cq[(k + 2) % 3][u] = *reinterpret_cast<const dbl2 *>(ccur + ((k + 2) * 8 + u) * 16);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const dbl2 co = cq[k % 3][u];
          double acc = pr[k % 3][u];
          if constexpr (PA & 1u) acc = acc + co.x * m1;
          if constexpr (PA & 2u) acc = acc + co.y * m2;
          if (k == 0 && u < 3 && i == 0) {
            // start of the stream: group q has nothing to do before step q; hold its state
            const bool on = u >= q;
            m2 = on ? m1 : m2;
            m1 = on ? acc : m1;
          } else {
            m2 = m1;
            m1 = acc;
          }
          if ((u & 3) == 3) *reinterpret_cast<double *>(wr + (k * 8 + u) * kStep) = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // y of tile i done, p / pairs of tile i+1 ready
    }
    if (lane < G) {
      if (p.na > 1) p.yh[0 * p.channels + c] = m1;
      if (p.na > 2) p.yh[1 * p.channels + c] = m2;
    }
  }
#undef ALZ_EOFF
}

typedef void (*tvduo_fn)(TDArgs);

template <int ND, bool CM>
static tvduo_fn pick_tvduo_nd(unsigned pb, unsigned pa) {
#define ALZ_PAT(PB_, PA_) if (pb == PB_ && pa == PA_) return (tvduo_fn)k_tvduo<PB_, PA_, ND, CM>;
  ALZ_PAT(1, 1) ALZ_PAT(3, 1) ALZ_PAT(1, 3) ALZ_PAT(3, 3) ALZ_PAT(5, 3) ALZ_PAT(7, 3) ALZ_PAT(1, 2)
#undef ALZ_PAT
  return nullptr;
}

static tvduo_fn pick_tvduo(unsigned pb, unsigned pa, int n_dma, bool cm) {
  switch (n_dma) {
    case 1: return cm ? pick_tvduo_nd<1, true>(pb, pa) : pick_tvduo_nd<1, false>(pb, pa);
    case 2: return cm ? pick_tvduo_nd<2, true>(pb, pa) : pick_tvduo_nd<2, false>(pb, pa);
    case 3: return cm ? pick_tvduo_nd<3, true>(pb, pa) : pick_tvduo_nd<3, false>(pb, pa);
    default: return nullptr;          // no series tap at all: a plain LTI bank, not this entry point's case
  }
}

// The part of a time-varying block the two-wave kernel can take: *done_samples full 64-row tiles of
// all channels (0: not this kernel's shape).  taps: b0 b1 b2 a1 a2 as (kind, value, series, negated);
// series must be shared by the channels and contiguous in time.
int launch_tvduo(const double *x, double *y, int64_t n, int64_t ldx, int64_t ldy, int cm, int64_t channels, int nb, int na,
                 const int *kind, const double *value, const double *const *series, const int *negated,
                 double *xh, double *yh, hipStream_t stream, int64_t *done_samples) {
  *done_samples = 0;
  if (nb > 3 || na > 3 || channels % 16 || channels / 16 > 512 || n < 64) return ALZ_OK;
  if ((((uintptr_t)x | (uintptr_t)y) & 15) || ((ldx | ldy) & 1)) return ALZ_OK;
  unsigned pb = 0, pa = 0;
  for (int k = 0; k < 3; ++k) pb |= (unsigned)(kind[k] != 0) << k;
  for (int k = 3; k < 5; ++k) pa |= (unsigned)(kind[k] != 0) << (k - 3);
  int n_series = 0;
  for (int k = 0; k < 5; ++k) n_series += kind[k] == 2;
  tvduo_fn fn = pick_tvduo(pb, pa, (n_series + 1) / 2, cm != 0);
  if (!fn) return ALZ_OK;
  TDArgs p;
  p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy; p.n_tiles = n / 64; p.channels = channels;
  p.nb = nb; p.na = na; p.xh = xh; p.yh = yh;
  int ns = 0;
  const double *list[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < 5; ++k) {
    p.kind[k] = kind[k]; p.value[k] = value[k]; p.series[k] = series[k]; p.negated[k] = negated[k];
    p.slot_of[k] = 0;
    if (kind[k] == 2) {
      if ((uintptr_t)series[k] & 15) return ALZ_OK;          // 16-byte pieces
      p.slot_of[k] = ns;
      list[ns++] = series[k];
    }
  }
  p.n_dma = (ns + 1) / 2;
  for (int d = 0; d < 3; ++d) {
    p.dma_src[d][0] = list[2 * d < ns ? 2 * d : 0];
    p.dma_src[d][1] = list[2 * d + 1 < ns ? 2 * d + 1 : (2 * d < ns ? 2 * d : 0)];
  }
  // (tuning builds: the workgroups of a time-major launch on one tile clock)
  p.tile_pace = cm ? 0 : tile_pace16((channels / 16) * 16384ll, ALZ_TUNE("ALZ_TVDUO_PACE_GBPS", 0));
  const size_t py_slot = cm ? (size_t)16 * (64 * 8 + 16) : (size_t)kSlot;
  const size_t lds = (size_t)kXRing * kSlot + (size_t)(kPRing + kYRing) * py_slot + (size_t)kXRing * kRawSlot +
                     (size_t)kPRing * kPairSlot;
  const int rc = ensure_dynamic_lds((const void *)fn, 96 * 1024);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)(channels / 16)), dim3(192), channels / 16 <= 256 ? (size_t)96 * 1024 : lds, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *done_samples = p.n_tiles * 64;
  return ALZ_OK;
}

}  // namespace alz
