// alz_look.hip -- the time-parallel mode of a narrow biquad-class bank in ONE pass over the block.
//
// alz_scan.hip runs a narrow bank (512 channels: one GPU's share of configs[1] sharded over 8) as chunks x channels
// lanes in three launches: zero-state pass (reads the block), scan of the chunk states, replay (reads it again,
// writes the result) -- 24 bytes of HBM traffic per sample for 16 algorithmic.  Here a chunk is small enough to stay
// in LDS between the two uses (512 samples x 16 channels = 64 KiB), so the block is read ONCE:
//
//   * a workgroup owns 16 channels and every W-th chunk of them (W workgroups per channel group, all resident);
//     three waves pipeline over 64-sample tiles with one barrier per tile, as k_duo does:
//       AUX  queues the tile DMA (global_load_lds) into a ring of 16 tile slots, turns a landed tile into
//            feed-forward sums p[n] IN PLACE, and stores finished tiles;
//       ZS   runs the recurrence over p from a ZERO state, one chunk + 2 tiles ahead of REC, only for the chunk's
//            end state z_j, which it publishes in global memory (64-bit agent-scope atomic stores into an array
//            pre-filled with a NaN pattern no computation produces: no flags, no fences);
//       REC  runs the recurrence from the TRUE state and overwrites p with y.
//   * the true state of chunk j needs no other workgroup's replay: REC keeps the exact end state of its own previous
//     chunk j - W and applies  S <- M S + z  for the W - 1 chunks in between (M = A^L per channel, the matrix
//     alz_scan.hip caches; z_{j-W+1} .. z_{j-1} from the other workgroups' ZS waves, which run a chunk ahead).
//     Waits only ever point to smaller chunk indices, and every workgroup of the launch is resident (<= one per
//     CU): no deadlock; a bounded spin guards against the impossible.
//
// Every output sample is still the reference's DF-I statement (lazy_filters.py:197-257) in the kernels' own order;
// only the chunk-start states carry a different rounding -- the same numerics as the three-launch mode (1e-10 ..
// 1e-9 on the configs[1] resonators).  Time-major blocks, a0 == 1, channels % 16 == 0, blocks of whole 512-sample
// chunks; everything else stays on the three-launch mode.
#include "alz_common.h"

namespace alz {

namespace {

constexpr int kChunks = 8;                   // 1 KiB DMA transfers per tile
constexpr int kNT = 8;                       // tiles per chunk (L = 512)
constexpr int kSlots = 2 * kNT;              // tile slots in LDS
constexpr int kSlot = 8192 + kChunks * 16;   // a tile in the DMA layout (16 bytes of pad per 1 KiB chunk)
constexpr int kHist = 256;                   // the two rows before a tile: [2][16] doubles
constexpr unsigned long long kSentinel = ~0ull;   // the "not yet published" pattern (hipMemset 0xFF)
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
  unsigned long long *z;       // [groups][n_chunks][2][16] published zero-state end states
  int *err;                    // set when a spin ran into its cap
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

}  // namespace

template <unsigned PB, unsigned PA>
__global__ __launch_bounds__(192) void k_look(LArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 16, T = 64, NT = kNT;
  constexpr int kStep = G * 8;
  // pipeline offsets, in tiles of this workgroup's own tile sequence (see the header): at interval i
  //   AUX stores tile i - kRecLag - 1, queues the DMA of tile i + kDmaLead, prepares tile i + 1,
  //   ZS works on tile i, REC on tile i - kRecLag
  constexpr int kRecLag = NT + 2, kDmaLead = 4;
  static_assert(kSlots >= kRecLag + 1 + kDmaLead + 1, "a slot is stored before it is refilled");
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cl = lane & 15, q = lane >> 4;
  const int W = p.workers;
  const int64_t group = (int64_t)blockIdx.x / W;
  const int w = (int)((int64_t)blockIdx.x - group * W);
  const int64_t c0 = group * G, c = c0 + cl;
  const int64_t K = p.n_chunks;
  const int64_t my_chunks = (K - w + W - 1) / W;             // chunks w, w + W, ...
  const int64_t TOT = my_chunks * NT;
  const int64_t set = p.n_inputs ? c / p.n_inputs : ((p.n_sets == 1) ? 0 : c);
  char *hist = smem + kSlots * kSlot;                         // [kSlots][2][16] doubles: rows -2, -1 of every tile
  char *exch = hist + kSlots * kHist;                         // [2][16] doubles: REC's end state, group 0 -> all groups
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lane_off = cl * 8;
#define ALZ_EOFF(u) ((u) * G * 8 + (((u) * G) >> 7) * 16)
  // global row of tile t's first sample: chunk (w + (t / NT) W), tile t % NT of it
  auto tile_row = [&](int64_t t) -> int64_t { return ((int64_t)w + (t / NT) * W) * (NT * T) + (t % NT) * T; };
  unsigned long long *zg = p.z + group * K * 32;              // this group's [K][2][16]

  if (wave == 1) {
    // ------------------------------ AUX ------------------------------
    const int row = lane / 8, cp = lane % 8;
    const double *xg = p.x + (int64_t)row * p.ldx + c0 + 2 * cp;
    double *yg = p.y + (int64_t)row * p.ldy + c0 + 2 * cp;
    const int64_t x_chunk = 8 * p.ldx, y_chunk = 8 * p.ldy;
    double b0 = 0, b1 = 0, b2 = 0;
    if (PB & 1u) b0 = p.b[0 * p.n_sets + set];
    if (PB & 2u) b1 = p.b[1 * p.n_sets + set];
    if (PB & 4u) b2 = p.b[2 * p.n_sets + set];
    const double h1 = (p.nb > 1) ? p.xh[0 * p.channels + c] : 0.0;   // x[-1], x[-2] of the stream (chunk 0 only)
    const double h2 = (p.nb > 2) ? p.xh[1 * p.channels + c] : 0.0;
    double bb0 = b0, bb1 = b1, bb2 = b2, hh1 = h1, hh2 = h2;
    asm volatile("" : "+v"(bb0), "+v"(bb1), "+v"(bb2), "+v"(hh1), "+v"(hh2));
    // nine transfers per tile: the tile, and (16 lanes only) the two rows before it
    auto queue_tile = [&](int64_t t) {
      const int s = (int)(t % kSlots);
      const int64_t r0 = tile_row(t);
      const double *src = xg + r0 * p.ldx;
#pragma unroll
      for (int j = 0; j < kChunks; ++j) dma16(src + j * x_chunk, lds0 + s * kSlot + j * (1024 + 16));
      // rows r0 - 2, r0 - 1 (clamped to row 0 for the very first tile, which uses the bank's history instead)
      const int64_t rh = r0 >= 2 ? r0 - 2 : 0;
      const double *hsrc = p.x + (rh + (lane >> 3 & 1)) * p.ldx + c0 + 2 * (lane & 7);
      if (lane < 16) dma16(hsrc, lds0 + (unsigned)(hist - smem) + s * kHist);
    };
    auto prepare_tile = [&](int64_t t) {
      char *xs = smem + (int)(t % kSlots) * kSlot + lane_off;
      const char *hs = hist + (int)(t % kSlots) * kHist + cl * 8;
      const int adj1 = (q == 0) ? 16 : 0, adj2 = (q < 2) ? 16 : 0;
      const char *x_d0 = xs + q * kStep;
      const char *x_d1[2] = {xs + (q - 1) * kStep - adj1, xs + (q - 1) * kStep};   // [j odd]
      const char *x_d2[2] = {xs + (q - 2) * kStep - adj2, xs + (q - 2) * kStep};
      double x0[16], x1[16], x2[16];
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
        // x[-2], x[-1] relative to this tile: the landed history rows, or the bank's history at the start of the stream
        const bool stream_start = tile_row(t) == 0;
        const double g2 = *reinterpret_cast<const double *>(hs), g1 = *reinterpret_cast<const double *>(hs + 128);
        const double pm1 = stream_start ? hh1 : g1, pm2 = stream_start ? hh2 : g2;
        const double s0 = *reinterpret_cast<const double *>(xs + ALZ_EOFF(0));
        const double s1 = *reinterpret_cast<const double *>(xs + ALZ_EOFF(1));
        const double s2 = *reinterpret_cast<const double *>(xs + ALZ_EOFF(2));
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
      // in place: every read of the tile is done (one wave, program order), p goes to unpadded rows of the same slot
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) *reinterpret_cast<double *>(xs + (4 * j + q) * kStep) = acc[j];
    };
    auto store_tile = [&](int64_t t) {
      const char *ys = smem + (int)(t % kSlots) * kSlot;
      double *yt = yg + tile_row(t) * p.ldy;
      dbl2 v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j) v[j] = *reinterpret_cast<const dbl2 *>(ys + j * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < kChunks; ++j) store16(yt + j * y_chunk, v[j]);
    };
    // the input history the bank keeps for the next block: the last two x rows of the block (owner of the last chunk)
    const bool owns_last = ((K - 1) % W) == w;
    for (int t = 0; t < kDmaLead && t < TOT; ++t) queue_tile(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prepare_tile(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int64_t n_iv = TOT + kRecLag + 2;
    for (int64_t i = 0; i < n_iv; ++i) {
      const int64_t ts = i - kRecLag - 1;
      if (ts >= 0 && ts < TOT) store_tile(ts);
      if (i + kDmaLead < TOT) queue_tile(i + kDmaLead);
      if (i + 1 < TOT) {
        // issued after tile i + 1's transfers: three more tiles (9 each) and the stores of the last three intervals
        // (8 each, once tiles are being stored); the last tiles of the sequence simply wait for everything
        if (i + kDmaLead < TOT) {
          if (ts >= 2) asm volatile("s_waitcnt vmcnt(51)" ::: "memory");
          else if (ts == 1) asm volatile("s_waitcnt vmcnt(43)" ::: "memory");
          else if (ts == 0) asm volatile("s_waitcnt vmcnt(35)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (owns_last && i + 1 == TOT - 1 && q == 3) {        // (before the tile is overwritten with p)
          const char *xs = smem + (int)((i + 1) % kSlots) * kSlot + lane_off;
          if (p.nb > 1) p.xh[0 * p.channels + c] = *reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 1));
          if (p.nb > 2) p.xh[1 * p.channels + c] = *reinterpret_cast<const double *>(xs + ALZ_EOFF(T - 2));
        }
        prepare_tile(i + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ------------------------------ ZS (wave 2) and REC (wave 0): the recurrence ------------------------------
    const bool is_rec = wave == 0;
    double na1 = 0, na2 = 0;
    if (PA & 1u) na1 = -p.a[1 * p.n_sets + set];
    if (PA & 2u) na2 = -p.a[2 * p.n_sets + set];
    const double m11 = p.power[0 * p.channels + c], m12 = p.power[1 * p.channels + c];
    const double m21 = p.power[2 * p.channels + c], m22 = p.power[3 * p.channels + c];
    double base1 = (p.na > 1) ? p.yh[0 * p.channels + c] : 0.0;      // REC: the state its next chunk is derived from
    double base2 = (p.na > 2) ? p.yh[1 * p.channels + c] : 0.0;      // (the bank's state, then its own end states)
    asm volatile("" : "+v"(na1), "+v"(na2), "+v"(base1), "+v"(base2));
    double m1 = 0.0, m2 = 0.0, t2 = 0.0;
    bool gave_up = false;
    double kp0 = 0.0, kp1 = 0.0, kp2 = 0.0;                  // p of the previous tile's rows T-3, T-2, T-1 (lagging groups)
    const int lag = is_rec ? kRecLag : 0;
    const int64_t n_iv = TOT + kRecLag + 2;
    __builtin_amdgcn_s_barrier();
    for (int64_t i = 0; i < n_iv; ++i) {
      const int64_t t = i - lag;                             // the tile this wave works on
      if (t >= 0 && t < TOT) {
        char *cur = smem + (int)(t % kSlots) * kSlot + lane_off - q * kStep;
        const bool chunk_start = (t % NT) == 0;
        double s1 = 0.0, s2 = 0.0;                           // the state row 0 of a new chunk starts from
        if (chunk_start && is_rec) {
          // true state of chunk j: S <- M S + z over the chunks between the base state and j
          const int64_t seq = t / NT, j = (int64_t)w + seq * W;
          if (seq > 0) {                                     // base = this wave's own end state of chunk j - W (group 0 has it)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            base1 = *reinterpret_cast<const double *>(exch + cl * 8);
            base2 = *reinterpret_cast<const double *>(exch + 128 + cl * 8);
          }
          const int64_t first = seq > 0 ? j - W + 1 : 0;     // z_first .. z_{j-1}
          s1 = base1; s2 = base2;
          for (int64_t jj = first; jj < j; ++jj) {
            unsigned long long v1 = kSentinel, v2 = kSentinel;
            const unsigned long long *src = zg + jj * 32 + cl;
            int spins = gave_up ? kSpinCap : 0;
            while (spins < kSpinCap) {
              v1 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              v2 = __hip_atomic_load(src + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (v1 != kSentinel && v2 != kSentinel) break;
              ++spins;
              __builtin_amdgcn_s_sleep(8);
            }
            if (spins >= kSpinCap) { *p.err = 1; v1 = 0; v2 = 0; gave_up = true; }   // (results are garbage from here on; no further waits)
            const double z1 = __longlong_as_double((long long)v1), z2 = __longlong_as_double((long long)v2);
            const double n1 = __builtin_fma(m11, s1, __builtin_fma(m12, s2, z1));
            const double n2 = __builtin_fma(m21, s1, __builtin_fma(m22, s2, z2));
            s1 = n1; s2 = n2;
          }
        }
        // this tile's last three p rows, for the lagging groups' first steps of the NEXT tile (REC overwrites them)
        const double nk0 = *reinterpret_cast<const double *>(cur + (q + T - 3) * kStep);
        const double nk1 = *reinterpret_cast<const double *>(cur + (q + T - 2) * kStep);
        const double nk2 = *reinterpret_cast<const double *>(cur + (q + T - 1) * kStep);
        constexpr int NCH = T / 8;
        double pr[3][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double v = *reinterpret_cast<const double *>(cur + (u < q ? q : u) * kStep);
          const int jx = u + 3 - q;                          // (u < q: row T + u - q of the previous tile)
          pr[0][u] = (u < 3 && u < q) ? (jx <= 0 ? kp0 : jx == 1 ? kp1 : kp2) : v;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) pr[1][u] = *reinterpret_cast<const double *>(cur + (8 + u) * kStep);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          if (k + 2 < NCH) {
#pragma unroll
            for (int u = 0; u < 8; ++u) pr[(k + 2) % 3][u] = *reinterpret_cast<const double *>(cur + ((k + 2) * 8 + u) * kStep);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (k == 0 && u <= 3 && chunk_start) {
              // row 0 of the chunk is this lane group's step q: (re)start from the chunk's initial state there;
              // at the very start of the stream the lagging groups have nothing to do before that step
              const bool start = u == q;
              m1 = start ? s1 : m1;
              m2 = start ? s2 : m2;
              t2 = start ? na2 * s2 : t2;
            }
            double acc = pr[k % 3][u];
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
            if (is_rec && (u & 3) == 3) *reinterpret_cast<double *>(cur + (k * 8 + u) * kStep) = acc;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        kp0 = nk0; kp1 = nk1; kp2 = nk2;
        if ((t % NT) == NT - 1 && q == 0) {
          // group 0 has just finished the chunk: its (m1, m2) is the chunk's end state
          const int64_t j = (int64_t)w + (t / NT) * W;
          if (is_rec) {
            *reinterpret_cast<double *>(exch + cl * 8) = m1;
            *reinterpret_cast<double *>(exch + 128 + cl * 8) = m2;
            if (j == K - 1) {                                // the bank's state after the block
              if (p.na > 1) p.yh[0 * p.channels + c] = m1;
              if (p.na > 2) p.yh[1 * p.channels + c] = m2;
            }
          } else {
            __hip_atomic_store(zg + j * 32 + cl, (unsigned long long)__double_as_longlong(m1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(zg + j * 32 + 16 + cl, (unsigned long long)__double_as_longlong(m2), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#undef ALZ_EOFF
}

typedef void (*look_fn)(LArgs);
static look_fn pick_look(unsigned pb, unsigned pa) {
#define ALZ_PAT(PB_, PA_) if (pb == PB_ && pa == PA_) return (look_fn)k_look<PB_, PA_>;
  ALZ_PAT(1, 1) ALZ_PAT(3, 1) ALZ_PAT(1, 3) ALZ_PAT(3, 3) ALZ_PAT(5, 3) ALZ_PAT(7, 3) ALZ_PAT(1, 2)
#undef ALZ_PAT
  return nullptr;
}

// One-pass time-parallel run of a biquad-class section over whole 512-sample chunks of a time-major block.
// `power` = the section's M = A^512 per channel ([4][channels], alz_scan.hip); `zbuf` (>= groups * chunks * 32 doubles)
// and `err` are scratch of the handle.  *done_samples: the whole chunks covered (0: not this kernel's shape).
int launch_look(const SectionDev &sec, const BlockIO &io, hipStream_t stream, const double *power, double *zbuf,
                uint64_t zbuf_bytes, int *err, int64_t *done_samples, const char **kernel_name) {
  *done_samples = 0;
  if (!(sec.nb <= 3 && sec.na <= 3 && sec.na >= 2 && sec.uniform) || sec.any_div || io.fused) return ALZ_OK;
  const bool tm = io.sxc == 1 && io.syc == 1;
  if (!tm || io.map_input || io.pre_op) return ALZ_OK;
  const int64_t C = io.channels, L = kNT * 64;
  if (C % 16 || io.c_first != 0 || io.c_count != C || io.x == io.y) return ALZ_OK;
  if ((((uintptr_t)io.x | (uintptr_t)io.y) & 15) || ((io.sxn | io.syn) & 1)) return ALZ_OK;
  const int64_t K = io.n / L, groups = C / 16;
  int W = (int)(256 / groups);                               // every workgroup of the launch resident: one per CU
  if (W > K) W = (int)K;
  if (W < 2 || groups > 128 || K < 4) return ALZ_OK;
  if ((uint64_t)groups * K * 32 * sizeof(double) > zbuf_bytes) return ALZ_OK;
  look_fn fn = pick_look(sec.present_b, sec.present_a);
  if (!fn) return ALZ_OK;
  LArgs p;
  p.x = io.x; p.y = io.y; p.ldx = io.sxn; p.ldy = io.syn; p.n_chunks = K; p.channels = C;
  p.n_inputs = io.mode == ALZ_BANK_OUTER ? io.n_inputs : 0; p.n_sets = io.n_sets; p.workers = W;
  p.nb = sec.nb; p.na = sec.na; p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.yh = sec.yh;
  p.power = power; p.z = (unsigned long long *)zbuf; p.err = err;
  ALZ_HIP_CHECK(hipMemsetAsync(zbuf, 0xFF, (size_t)groups * K * 32 * sizeof(double), stream));
  const size_t lds = (size_t)kSlots * kSlot + (size_t)kSlots * kHist + 256;
  const int rc = ensure_dynamic_lds((const void *)fn, (int)lds);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)(groups * W)), dim3(192), lds, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *done_samples = K * L;
  *kernel_name = "k_look";
  return ALZ_OK;
}

}  // namespace alz
