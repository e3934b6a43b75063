// alz_mid.hip -- the shapes between "biquad" and "long FIR" as streaming kernels (round 6).
//
// Replaces LinearFilter.__call__ (reference audiolazy/lazy_filters.py:141-264) for ONE-section filters that the two-pole
// streaming kernels (alz_wave.hip) do not take and that used to run lane-per-channel (k_masked / k_generic, alz_iir.hip):
//   * IIR sections of order 3 .. 8 with dense coefficients -- ZFilter(butter(...)) as the reference's own
//     examples/butterworth_with_noise.py:52-67 builds them (orders 4 and 6), all-pole filters (lpc synthesis, 1 / A(z));
//   * a short dense numerator + ONE far tap in front of one or two poles -- maverage.recursive(size)
//     (lazy_analysis.py:569-591: (1 - z^-size) / (size (1 - z^-1)), nb = size + 1 with two taps, na = 2).
//
// k_duo's split, generalised: a workgroup owns 16 channels of a time-major block; per 64-sample tile
//   AUX    queues the tile DMA (global_load_lds, three tiles ahead) and forms the feed-forward sums p[n] = sum_k b_k x[n-k]
//          of a whole tile time-parallel: lane (q, channel) owns samples 4 j + q and reads every row it needs ONCE into
//          registers (the x ring is one contiguous run of 128-byte rows -- a guard copy of the ring's last 8 rows sits in
//          front of its first slot, so a tile's reach into its predecessor never wraps); the far tap comes from up to
//          four tiles back in the same ring (masked ring addresses: 16 reads per tile);
//   REC    runs only the serial part  y[n] = (..((p[n] + (-a1) y[n-1]) + (-a2) y[n-2]) ..) + (-aK) y[n-K]  -- the reference's
//          left-to-right sum, separately rounded -- on 16 channels x 4 copies skewed by one step (one ds_write_b64
//          stores four finished rows, as in k_duo); its products (-a_k) y[n-k], k >= 2, do not wait for y[n-1], but the K
//          additions do: the chain is one multiply + K additions per sample and cannot be shortened without changing the
//          doubles (floating-point addition does not re-associate);
//   STORE  writes finished tiles with 1 KiB stores.
// One barrier per tile.  Bound: the recurrence chain for K >= 3 (channels x clock / ((K + 1) x ~7.5 cycles): order 6 at 4096
// channels cannot pass ~185 Gsamples/s = 0.37 of the HBM roof bit-exactly); HBM for maverage.recursive (one pole).
// Time-major rows, a0 == 1, whole 64-sample tiles (the ragged rest continues on the lane-per-channel kernels from the
// state this kernel leaves), in place too (the far tap is read from the ring in LDS, not from memory).
#include "alz_common.h"

namespace alz {

namespace {

constexpr int kMT = 64, kMG = 16, kMTile = 8192, kMGuard = 1024, kMPRing = 3, kMYRing = 2;

struct MArgs {
  const double *x;
  double *y;
  int64_t ldx, ldy, n_tiles;
  int64_t channels, n_inputs, n_sets;
  int map_input;
  int far_delay;               // FAR instantiations: the delay S of the far numerator tap (9 <= S <= 256)
  unsigned pb_mask;            // bit k: numerator tap k is present (a zero tap is absent from the sum: lazy_filters.py:205-206); all ones when dense
  int ns;                      // x ring slots (a power of two: 4, or 8 with a far tap)
  const double *b, *a;
  double *xh, *yh;             // the bank's state [taps - 1][channels]
  int tile_pace;               // the common tile clock (alz_common.h pace_wait; 0: free-running)
};

template <bool NT>
__device__ __forceinline__ void mid_dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef double mdbl2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void mid_store16(double *gdst, mdbl2 v) {
  if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(gdst), "v"(v) : "memory");
}
// at most n vector-memory operations outstanding (n rounded DOWN to a literal the instruction can carry: waits longer, never shorter)
__device__ __forceinline__ void mid_wait_vm(int n) {
  if (n >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if (n >= 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// NB dense numerator taps b_0 .. b_{NB-1} (+ one far tap b_S when FAR), K dense feedback taps a_1 .. a_K
template <int NB, int K, bool FAR, bool NT>
__global__ __launch_bounds__(192) void k_mid(MArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int T = kMT, G = kMG;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cl = lane & 15, q = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * G, c = c0 + cl;
  const int64_t in0 = (p.n_inputs && p.map_input) ? c0 % p.n_inputs : c0;
  const int64_t set = p.n_inputs ? c / p.n_inputs : ((p.n_sets == 1) ? 0 : c);
  const int64_t nt = p.n_tiles;
  const int NS = p.ns;
  char *guard = smem;                                    // rows -8 .. -1 in front of ring slot 0
  char *xring = smem + kMGuard;                          // NS slots of 64 rows x 128 bytes, contiguous
  char *pring = xring + NS * kMTile;
  char *yring = pring + kMPRing * kMTile;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  if (wave == 1) {
    // ------------------------------------------ AUX ------------------------------------------
    const int row = lane / 8, cp = lane % 8;
    const double *xg = p.x + (int64_t)row * p.ldx + in0 + 2 * cp;
    const int64_t x_chunk = 8 * p.ldx, x_tile = (int64_t)T * p.ldx;
    double bc[NB], bfar = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) bc[k] = p.b[(int64_t)k * p.n_sets + set];
    if constexpr (FAR) bfar = p.b[(int64_t)p.far_delay * p.n_sets + set];
#pragma unroll
    for (int k = 0; k < NB; ++k) asm volatile("" : "+v"(bc[k]));
    asm volatile("" : "+v"(bfar));
    const int S = FAR ? p.far_delay : 0;
    const unsigned ring_mask = (unsigned)(NS * kMTile - 1);
    // the delay line before the block: row -1-k = xh[k], into the guard (dense taps) and, for the far tap, into the tail of the ring
    {
      const int hx = FAR ? S : NB - 1;
      for (int k = q; k < hx; k += 4) {
        const double v = p.xh[(int64_t)k * p.channels + c];
        if (k < 8) *reinterpret_cast<double *>(guard + (7 - k) * 128 + cl * 8) = v;
        if constexpr (FAR) *reinterpret_cast<double *>(xring + (((unsigned)(-(k + 1)) * 128u) & ring_mask) + cl * 8) = v;
      }
    }
    auto dmas_of = [&](int64_t t) -> int { return 8 + (((int)t & (NS - 1)) == NS - 1 ? 1 : 0); };
    auto queue_tile = [&](int64_t t) {
      const int s = (int)t & (NS - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) mid_dma16<NT>(xg + t * x_tile + j * x_chunk, lds0 + kMGuard + s * kMTile + j * 1024);
      if (s == NS - 1) mid_dma16<NT>(xg + t * x_tile + 7 * x_chunk, lds0);          // the ring's last 8 rows again, in front of slot 0
    };
    auto feed_forward = [&](int64_t t) {
      const int s = (int)t & (NS - 1);
      const char *xs = xring + s * kMTile + q * 128 + cl * 8;        // row q of the tile; row q + i at xs + i * 128 (i >= -8: guard / previous slot)
      char *ps = pring + (int)(t % kMPRing) * kMTile + q * 128 + cl * 8;
      // rows q + i, i = 4 j - k: each needed row read once
      double xr[61 + NB - 1];
#pragma unroll
      for (int i = -(NB - 1); i <= 60; ++i) {
        bool need = false;
#pragma unroll
        for (int k = 0; k < NB; ++k) need |= ((i + k) % 4 == 0) && (i + k >= 0) && (i + k <= 60);
        if (need) xr[i + NB - 1] = *reinterpret_cast<const double *>(xs + i * 128);
      }
      double xf[16];
      if constexpr (FAR) {
        const unsigned base = (unsigned)((int)(t * T) + q - S) * 128u;                 // row 64 t + q - S of the stream, as a ring offset
#pragma unroll
        for (int j = 0; j < 16; ++j) xf[j] = *reinterpret_cast<const double *>(xring + ((base + (unsigned)j * 512u) & ring_mask) + cl * 8);
      }
      if (p.pb_mask == (1u << NB) - 1u) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          double acc = bc[0] * xr[4 * j + NB - 1];
#pragma unroll
          for (int k = 1; k < NB; ++k) acc = acc + bc[k] * xr[4 * j - k + NB - 1];
          if constexpr (FAR) acc = acc + bfar * xf[j];
          *reinterpret_cast<double *>(ps + j * 512) = acc;
        }
      } else {
        // zero taps inside the numerator (a band-pass butter's b = [b0, 0, -2 b0, 0, b0]): absent from the sum, like the
        // reference's generated statement; the pattern is the bank's (uniform), so the tests are scalar
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          double acc = -0.0;                                             // additive identity: the first present term initialises the sum
#pragma unroll
          for (int k = 0; k < NB; ++k)
            if (p.pb_mask >> k & 1u) acc = acc + bc[k] * xr[4 * j - k + NB - 1];
          if constexpr (FAR) acc = acc + bfar * xf[j];
          *reinterpret_cast<double *>(ps + j * 512) = acc;
        }
      }
    };
    int queued = 0;
    for (int t = 0; t < 3 && t < nt; ++t) { queue_tile(t); queued += dmas_of(t); }
    mid_wait_vm(queued - dmas_of(0));                                                // tile 0 has landed
    feed_forward(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long pace0 = p.tile_pace > 0 ? (long long)wall_clock64() : 0;

    long long pace_shift = 0;
    for (int64_t i = 0; i < nt; ++i) {
      if (p.tile_pace > 0) pace_wait(pace0, i, p.tile_pace, pace_shift);
      if (i + 3 < nt) queue_tile(i + 3);
      if (i + 1 < nt) {
        int after = 0;                                                               // transfers issued after tile i + 1's
        if (i + 2 < nt) after += dmas_of(i + 2);
        if (i + 3 < nt) after += dmas_of(i + 3);
        mid_wait_vm(after);
        feed_forward(i + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // the input history for the next block: the block's last rows, still in the ring (written after every read of the old one)
    {
      const int hx = FAR ? S : NB - 1;
      const unsigned last = (unsigned)((int)(nt * T) - 1) * 128u;
      for (int k = q; k < hx; k += 4) {
        double v;
        if (FAR || nt * T - 1 - k >= 0) v = *reinterpret_cast<const double *>(xring + ((last - (unsigned)k * 128u) & ring_mask) + cl * 8);
        else v = *reinterpret_cast<const double *>(guard + (8 - (k + 1 - (int)(nt * T))) * 128 + cl * 8);
        p.xh[(int64_t)k * p.channels + c] = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (wave == 2) {
    // ------------------------------------------ STORE ------------------------------------------
    const int row = lane / 8, cp = lane % 8;
    double *yg = p.y + (int64_t)row * p.ldy + c0 + 2 * cp;
    const int64_t y_chunk = 8 * p.ldy, y_tile = (int64_t)T * p.ldy;
    auto store_tile = [&](int64_t t) {
      const char *ys = yring + (int)(t % kMYRing) * kMTile;
      double *yt = yg + t * y_tile;
      mdbl2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const mdbl2 *>(ys + j * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) mid_store16<NT>(yt + j * y_chunk, v[j]);
    };
    __builtin_amdgcn_s_barrier();
    for (int64_t i = 0; i < nt; ++i) {
      if (i >= 1) store_tile(i - 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    store_tile(nt - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ------------------------------------------ REC ------------------------------------------
    double na[K], m[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      na[k] = -p.a[(int64_t)(k + 1) * p.n_sets + set];
      m[k] = p.yh[(int64_t)k * p.channels + c];
      asm volatile("" : "+v"(na[k]), "+v"(m[k]));
    }
    __builtin_amdgcn_s_barrier();                                                   // p of tile 0 is ready
    int ps_cur = 0, ps_prv = kMPRing - 1, ys_cur = 0;
    for (int64_t i = 0; i < nt; ++i) {
      // copy q runs q steps behind copy 0: at step u it takes sample u - q (u < q: the previous tile's last rows)
      const char *cur = pring + ps_cur * kMTile + cl * 8 - q * 128;
      const char *prv = pring + ps_prv * kMTile + cl * 8 + (T - q) * 128;
      char *wr = yring + ys_cur * kMTile + cl * 8 - q * 128;
      ps_prv = ps_cur;
      ps_cur = (ps_cur + 1 == kMPRing) ? 0 : ps_cur + 1;
      ys_cur = (ys_cur + 1 == kMYRing) ? 0 : ys_cur + 1;
      double pr[2][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const char *src = (u < 3 && u < q) ? prv : cur;
        pr[0][u] = *reinterpret_cast<const double *>(src + u * 128);
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) pr[(g + 1) & 1][u] = *reinterpret_cast<const double *>(cur + ((g + 1) * 8 + u) * 128);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          double acc = pr[g & 1][u];
#pragma unroll
          for (int k = 0; k < K; ++k) acc = acc + na[k] * m[k];
          if (g == 0 && u < 3 && i == 0) {
            const bool on = u >= q;                                                  // start of the stream: copy q holds its state until step q
#pragma unroll
            for (int k = K - 1; k > 0; --k) m[k] = on ? m[k - 1] : m[k];
            m[0] = on ? acc : m[0];
          } else {
#pragma unroll
            for (int k = K - 1; k > 0; --k) m[k] = m[k - 1];
            m[0] = acc;
          }
          if ((u & 3) == 3) *reinterpret_cast<double *>(wr + (g * 8 + u) * 128) = acc;   // rows u - 3 .. u, one per copy
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                                 // y of tile i done, p of tile i + 1 ready
    }
    if (lane < G) {
#pragma unroll
      for (int k = 0; k < K; ++k) p.yh[(int64_t)k * p.channels + c] = m[k];
    }
  }
}

typedef void (*mid_fn)(MArgs);
template <int NB, int K, bool FAR>
mid_fn mid_nt(bool nt) { return nt ? (mid_fn)k_mid<NB, K, FAR, true> : (mid_fn)k_mid<NB, K, FAR, false>; }

mid_fn pick_mid(int nb, int k, bool far, bool nt) {
#define ALZ_MID(NB_, K_) if (!far && nb == NB_ && k == K_) return mid_nt<NB_, K_, false>(nt);
  // order n with a full numerator (butter, cheby, ...: nb == na), n = 3 .. 8
  ALZ_MID(4, 3) ALZ_MID(5, 4) ALZ_MID(6, 5) ALZ_MID(7, 6) ALZ_MID(8, 7) ALZ_MID(9, 8)
  // all-pole filters (1 / A(z): LPC synthesis), order 3 .. 8
  ALZ_MID(1, 3) ALZ_MID(1, 4) ALZ_MID(1, 5) ALZ_MID(1, 6) ALZ_MID(1, 7) ALZ_MID(1, 8)
  // longer numerators in front of one or two poles
  ALZ_MID(4, 1) ALZ_MID(5, 1) ALZ_MID(6, 1) ALZ_MID(7, 1) ALZ_MID(8, 1) ALZ_MID(9, 1)
  ALZ_MID(4, 2) ALZ_MID(5, 2) ALZ_MID(6, 2) ALZ_MID(7, 2) ALZ_MID(8, 2) ALZ_MID(9, 2)
#undef ALZ_MID
#define ALZ_MIDF(NB_, K_) if (far && nb == NB_ && k == K_) return mid_nt<NB_, K_, true>(nt);
  // a far tap behind a short numerator: maverage.recursive (b0, b_size | a1) and its relatives
  ALZ_MIDF(1, 1) ALZ_MIDF(1, 2) ALZ_MIDF(2, 1) ALZ_MIDF(2, 2)
#undef ALZ_MIDF
  return nullptr;
}

}  // namespace

// One section whose shape is one of pick_mid's, on a time-major block: the whole 64-sample tiles of all channels.
// *done_samples / *done_channels: what was covered (0: not this kernel's shape -- nothing launched).
int launch_mid(const SectionDev &sec, const BlockIO &io, hipStream_t stream, int64_t *done_samples, int64_t *done_channels,
               const char **kernel_name) {
  *done_samples = 0;
  *done_channels = 0;
  if (!sec.uniform || sec.any_div || sec.na < 2 || sec.na > 9) return ALZ_OK;
  if (ALZ_TUNE("ALZ_MID_OFF", 0)) return ALZ_OK;                        // (tuning builds: round 5's lane-per-channel kernels, for A/B timing)
  if (sec.nb <= 3 && sec.na <= 3) return ALZ_OK;                          // (the two-pole streaming kernels' shapes)
  if (!(io.sxc == 1 && io.syc == 1) || io.c_first != 0 || io.c_count != io.channels || io.channels % kMG) return ALZ_OK;
  if (io.pre_op) return ALZ_OK;                                          // (the opt-in FMA mode ALLOWS contraction; this kernel simply does not use it)
  if ((((uintptr_t)io.x | (uintptr_t)io.y) & 15) || ((io.sxn | io.syn) & 1)) return ALZ_OK;
  if (io.mode == ALZ_BANK_OUTER && io.map_input && io.n_inputs % kMG) return ALZ_OK;
  const int K = sec.na - 1;
  if (sec.present_a != (1u << K) - 1u) return ALZ_OK;                      // dense feedback a_1 .. a_K
  // numerator: dense b_0 .. b_{NB-1}, optionally one far tap behind it
  int nbd = 0, far_delay = 0;
  bool far = false;
  unsigned pb_mask = 0;
  if (sec.nb <= 9 && sec.present_b == (1u << sec.nb) - 1u) {
    nbd = sec.nb;
    pb_mask = sec.present_b;
  } else if (sec.nb >= 4 && sec.nb <= 9 && (sec.present_b >> (sec.nb - 1) & 1u)) {
    nbd = sec.nb;                                                          // zero taps inside a short numerator (highest tap present)
    pb_mask = sec.present_b;
  } else if (sec.n_ff >= 2 && sec.n_ff <= 3) {
    nbd = sec.n_ff - 1;
    for (int j = 0; j < nbd; ++j)
      if (sec.tap_b[j] != j) return ALZ_OK;
    far_delay = sec.tap_b[nbd];
    if (far_delay != sec.nb - 1 || far_delay < 9 || far_delay > 256) return ALZ_OK;
    far = true;
    pb_mask = (1u << nbd) - 1u;
  } else {
    return ALZ_OK;
  }
  const int64_t tiles = io.n / kMT;
  if (tiles < 1) return ALZ_OK;
  const bool nt = io.stream_once != 0;
  mid_fn fn = pick_mid(nbd, K, far, nt);
  if (!fn) return ALZ_OK;
  MArgs p;
  p.x = io.x; p.y = io.y; p.ldx = io.sxn; p.ldy = io.syn; p.n_tiles = tiles; p.channels = io.channels;
  p.n_inputs = io.mode == ALZ_BANK_OUTER ? io.n_inputs : 0; p.n_sets = io.n_sets; p.map_input = io.map_input;
  p.far_delay = far_delay; p.ns = far ? 8 : 4; p.pb_mask = pb_mask;
  p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.yh = sec.yh;
  // (tuning builds: the workgroups on one tile clock)
  p.tile_pace = tile_pace16((io.channels / kMG) * (long long)kMG * kMT * 16ll, ALZ_TUNE("ALZ_MID_PACE_GBPS", 0));
  const size_t lds = (size_t)kMGuard + (size_t)p.ns * kMTile + (size_t)(kMPRing + kMYRing) * kMTile;
  const int rc = ensure_dynamic_lds((const void *)fn, (int)lds);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)(io.channels / kMG)), dim3(192), lds, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *done_samples = tiles * kMT;
  *done_channels = io.channels;
  *kernel_name = far ? "k_mid<far tap>" : "k_mid";
  return ALZ_OK;
}

}  // namespace alz
