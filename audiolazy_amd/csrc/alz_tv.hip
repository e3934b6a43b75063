// alz_tv.hip -- time-varying linear filters: coefficients that are series, one value per sample.
//
// Replaces the Stream-coefficient branch of LinearFilter.__call__ (reference
// audiolazy/lazy_filters.py:197-224): a coefficient that is an iterable contributes the term
// ``next(b_k) * d_k`` (numerator) or ``-next(a_k) * m_k`` (denominator) -- one coefficient value
// consumed per output sample -- in the same left-to-right sum as the constant terms (numerator
// delays ascending, then denominator delays ascending; zero constants absent; ``/ a0`` when
// a0 != 1, ``-( )`` when a0 == -1, :236-240).  examples/formants.py and the Stream-argument forms
// of resonator / lowpass / highpass (:1179-1495) produce such filters.
//
// Lane = channel; every lane walks the block serially with the last NB-1 inputs and NA-1 outputs
// in registers.  The coefficient values of a batch of samples are loaded ahead of the dependent
// steps (they do not depend on the recurrence); a shared series (stride_c == 0) is one broadcast
// load per wave.  A single stream (channels == 1, the reference's own use) takes k_tv_one, where
// the lanes prefetch in time instead.  Arithmetic: separately rounded mul/add in the reference's
// order (bit-exact).
#include <cstdlib>
#include "alz_common.h"

#include <type_traits>

namespace alz {

static constexpr int kTvMax = 17;  // taps per side the register windows hold (delays 0 .. 16)

struct TvSide {
  int kind[kTvMax];               // 0 absent, 1 constant, 2 series
  double value[kTvMax];
  const double *series[kTvMax];
  int64_t sn[kTvMax], sc[kTvMax];
  int negated[kTvMax];            // denominator series already holds -a_k[n]
};

struct TvArgs {
  const double *x;
  double *y;
  int64_t n, sxn, sxc, syn, syc, channels;
  TvSide b, a;                    // a.kind[0] is unused (the gain is separate)
  int nb, na;
  int gain_mode;                  // 0 none, 1 divide by gain, 2 negate
  double gain;
  double *xh, *yh;                // [k * channels + c] = x[-1-k] / y[-1-k], updated in place
  double zero;
  int n_terms;
};

// PB / PA: compile-time presence masks (bit k <=> b_k present; bit k-1 <=> a_k present) for the
// curated biquad-class patterns -- absent taps then cost nothing and present ones need no
// "is it there" select; kTvAny leaves presence to the run-time descriptors.
static constexpr unsigned kTvAny = ~0u;

template <int NB, int NA, int B, unsigned PB = kTvAny, unsigned PA = kTvAny>
__global__ __launch_bounds__(64) void k_tv(TvArgs p) {
  constexpr bool STATIC = PB != kTvAny;
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.channels) return;
  double d[NB], m[NA];            // d[k] = x[n-k], m[k] = y[n-k]  (d[0], m[0] are scratch)
#pragma unroll
  for (int k = 1; k < NB; ++k) d[k] = (k < p.nb) ? p.xh[(int64_t)(k - 1) * p.channels + c] : 0.0;
#pragma unroll
  for (int k = 1; k < NA; ++k) m[k] = (k < p.na) ? p.yh[(int64_t)(k - 1) * p.channels + c] : 0.0;
  // The tap descriptors are wave-uniform run-time values.  They are read once, and the step
  // below picks constant / series / absent with selects: straight-line code, no branch per tap.
  double cb[NB], nca[NA];         // constants (denominator already negated)
  bool b_on[NB], b_ser[NB], a_on[NA], a_ser[NA];
#pragma unroll
  for (int k = 0; k < NB; ++k) { cb[k] = p.b.value[k]; b_on[k] = p.b.kind[k] != 0; b_ser[k] = p.b.kind[k] == 2; }
#pragma unroll
  for (int k = 0; k < NA; ++k) { nca[k] = -p.a.value[k]; a_on[k] = k > 0 && p.a.kind[k] != 0; a_ser[k] = k > 0 && p.a.kind[k] == 2; }
  const bool divide = p.gain_mode == 1, negate = p.gain_mode == 2, no_terms = p.n_terms == 0;
  const double gain = p.gain, zero = p.zero;
  const double *xc = p.x + c * p.sxc;
  double *yc = p.y + c * p.syc;

  for (int64_t n0 = 0; n0 < p.n; n0 += B) {
    const int cnt = (p.n - n0 < B) ? (int)(p.n - n0) : B;
    // everything the batch needs from memory, issued before the dependent steps
    double xv[B], sb[NB][B], sa[NA][B];
#pragma unroll
    for (int u = 0; u < B; ++u) xv[u] = (u < cnt) ? xc[(n0 + u) * p.sxn] : 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
#pragma unroll
      for (int u = 0; u < B; ++u) sb[k][u] = 0.0;
      if ((!STATIC || ((PB >> k) & 1u)) && b_ser[k]) {
        const double *s = p.b.series[k] + c * p.b.sc[k];
#pragma unroll
        for (int u = 0; u < B; ++u) sb[k][u] = (u < cnt) ? s[(n0 + u) * p.b.sn[k]] : 0.0;
      }
    }
#pragma unroll
    for (int k = 1; k < NA; ++k) {
#pragma unroll
      for (int u = 0; u < B; ++u) sa[k][u] = 0.0;
      if ((!STATIC || ((PA >> (k - 1)) & 1u)) && a_ser[k]) {
        const double *s = p.a.series[k] + c * p.a.sc[k];
        const bool neg = p.a.negated[k] != 0;
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const double v = (u < cnt) ? s[(n0 + u) * p.a.sn[k]] : 0.0;
          sa[k][u] = neg ? v : -v;
        }
      }
    }
    auto steps = [&](auto div_tag) {
    constexpr bool DIV = decltype(div_tag)::value;   // keeps the long division sequence out of the common path
#pragma unroll
    for (int u = 0; u < B; ++u) {
      d[0] = xv[u];
      double acc = -0.0;           // additive identity: the first present term initialises the sum
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if constexpr (STATIC) {
          if ((PB >> k) & 1u) acc = acc + (b_ser[k] ? sb[k][u] : cb[k]) * d[k];
        } else {
          const double coef = b_ser[k] ? sb[k][u] : cb[k];
          const double s = acc + coef * d[k];
          acc = b_on[k] ? s : acc;
        }
      }
#pragma unroll
      for (int k = 1; k < NA; ++k) {
        if constexpr (STATIC) {
          if ((PA >> (k - 1)) & 1u) acc = acc + (a_ser[k] ? sa[k][u] : nca[k]) * m[k];
        } else {
          const double coef = a_ser[k] ? sa[k][u] : nca[k];
          const double s = acc + coef * m[k];
          acc = a_on[k] ? s : acc;
        }
      }
      if constexpr (DIV) acc = acc / gain;
      acc = negate ? -acc : acc;
      if constexpr (!STATIC) acc = no_terms ? zero : acc;
      if (u < cnt) {
        yc[(n0 + u) * p.syn] = acc;
#pragma unroll
        for (int k = NA - 1; k > 1; --k) m[k] = m[k - 1];
        if (NA > 1) m[1] = acc;
#pragma unroll
        for (int k = NB - 1; k > 0; --k) d[k] = d[k - 1];
      }
    }
    };
    if (divide) steps(std::true_type{});
    else steps(std::false_type{});
  }
#pragma unroll
  for (int k = 1; k < NB; ++k)
    if (k < p.nb) p.xh[(int64_t)(k - 1) * p.channels + c] = d[k];
#pragma unroll
  for (int k = 1; k < NA; ++k)
    if (k < p.na) p.yh[(int64_t)(k - 1) * p.channels + c] = m[k];
}

// ---------------------------------------------------------------------------
// k_tvp<PB, PA, B>: k_tv for the curated biquad-class presence patterns, rebuilt around the cost of
// a step.  One wave per 64 channels has its SIMD to itself and issues about one instruction per
// five cycles, so a step costs what it issues: k_tv's step carries per-step "is the batch this
// long" branches, per-step constant / series selects and address arithmetic from spilled SGPRs.
// Here
//  * full batches and the ragged tail are separate instantiations of the batch body (no per-step
//    bounds tests in the full one), fully unrolled, so the delay lines rotate by renaming;
//  * the coefficient values of a batch are put in registers BEFORE the dependent steps by one
//    wave-uniform branch per tap and batch: a constant is copied, a series is loaded with vector
//    loads (running pointers, no multiplies) -- also a series shared by the bank, all lanes at one
//    address: scalar loads from the constant address space were tried for those and cost far more
//    than they saved (33 vs 55 Gsamples/s at 4096 channels);
//  * the input rows of the NEXT batch are requested before the steps of the current one;
//  * denominator terms are subtracted (acc - a_k[n] * y[n-k] is acc + (-a_k[n]) * y[n-k] bit for
//    bit; the negation is a source modifier), so nothing is negated per value except a series the
//    host already negated (ALZ_TV_NEGATED, the int-0 rule), which is negated back;
//  * the gain mode (none / negate / divide) picks one of three step bodies per batch.
// Same terms, same order, same roundings as k_tv.  Also tried and not kept: look-ahead
// for the series too (two full register sets), compiler-untracked look-ahead loads with a counted
// s_waitcnt, shared series staged in SGPRs, a second wave per block touching rows ahead, narrower waves.
// ---------------------------------------------------------------------------
#ifndef ALZ_TVP_B
#define ALZ_TVP_B 16   // samples per batch (8 and 12 measured slower, 24 equal)
#endif
typedef const double __attribute__((address_space(4))) *tv_uniform_t;
typedef double tv_vec8 __attribute__((ext_vector_type(8)));
typedef tv_vec8 tv_vec8_u __attribute__((aligned(8)));        // a series starts wherever the batch does
typedef const tv_vec8_u __attribute__((address_space(4))) *tv_uniform8_t;

template <bool FULL, int B>
__device__ __forceinline__ void tvp_fill(const TvSide &sd, int k, int64_t c, int64_t n0, int cnt, bool negate_back,
                                         double (&out)[B]) {
  if (sd.kind[k] != 2) {
#pragma unroll
    for (int u = 0; u < B; ++u) out[u] = sd.value[k];
  } else {                                   // a series: per channel, or one for the bank (channel stride 0)
    const char *q = (const char *)(sd.series[k] + c * sd.sc[k] + n0 * sd.sn[k]);
    const int64_t step = sd.sn[k] * 8;
#pragma unroll
    for (int u = 0; u < B; ++u) {
      out[u] = (FULL || u < cnt) ? *(const double *)q : 0.0;
      q += step;
    }
  }
  if (negate_back && sd.kind[k] == 2) {
#pragma unroll
    for (int u = 0; u < B; ++u) out[u] = -out[u];
  }
}

template <unsigned PB, unsigned PA, int B>
__global__ __launch_bounds__(64) void k_tvp(TvArgs p) {
  constexpr int NB = 3, NA = 3;
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.channels) return;
  double d[NB], m[NA];            // d[k] = x[n-k], m[k] = y[n-k]  (d[0], m[0] are scratch)
#pragma unroll
  for (int k = 1; k < NB; ++k) d[k] = (k < p.nb) ? p.xh[(int64_t)(k - 1) * p.channels + c] : 0.0;
#pragma unroll
  for (int k = 1; k < NA; ++k) m[k] = (k < p.na) ? p.yh[(int64_t)(k - 1) * p.channels + c] : 0.0;
  const double gain = p.gain;
  const char *xp = (const char *)(p.x + c * p.sxc);
  char *yp = (char *)(p.y + c * p.syc);
  const int64_t sx8 = p.sxn * 8, sy8 = p.syn * 8;

  auto load_x = [&](auto full_tag, int64_t n0, int cnt, double (&xv)[B]) {
    constexpr bool FULL = decltype(full_tag)::value;
    const char *q = xp + n0 * sx8;
#pragma unroll
    for (int u = 0; u < B; ++u) {
      xv[u] = (FULL || u < cnt) ? *(const double *)q : 0.0;
      q += sx8;
    }
  };
  auto batch = [&](auto full_tag, int64_t n0, int cnt, double (&xv)[B], auto look_ahead) {
    constexpr bool FULL = decltype(full_tag)::value;
    double cb[NB][B], ca[NA][B];
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if ((PB >> k) & 1u) tvp_fill<FULL, B>(p.b, k, c, n0, cnt, false, cb[k]);
#pragma unroll
    for (int k = 1; k < NA; ++k)
      if ((PA >> (k - 1)) & 1u) tvp_fill<FULL, B>(p.a, k, c, n0, cnt, p.a.negated[k] != 0, ca[k]);
    look_ahead();
    auto steps = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;         // 0 none, 1 divide, 2 negate
      char *q = yp + n0 * sy8;
#pragma unroll
      for (int u = 0; u < B; ++u) {
        d[0] = xv[u];
        double acc = -0.0;         // additive identity: the first present term initialises the sum
#pragma unroll
        for (int k = 0; k < NB; ++k)
          if ((PB >> k) & 1u) acc = acc + cb[k][u] * d[k];
#pragma unroll
        for (int k = 1; k < NA; ++k)
          if ((PA >> (k - 1)) & 1u) acc = acc - ca[k][u] * m[k];
        if constexpr (MODE == 1) acc = acc / gain;
        if constexpr (MODE == 2) acc = -acc;
        if (FULL || u < cnt) {
          *(double *)q = acc;
          q += sy8;
#pragma unroll
          for (int k = NA - 1; k > 1; --k) m[k] = m[k - 1];
          m[1] = acc;
#pragma unroll
          for (int k = NB - 1; k > 0; --k) d[k] = d[k - 1];
        }
      }
    };
    if (p.gain_mode == 0) steps(std::integral_constant<int, 0>{});
    else if (p.gain_mode == 1) steps(std::integral_constant<int, 1>{});
    else steps(std::integral_constant<int, 2>{});
  };
  // The input rows of the NEXT full batch are requested before the steps of the current one (two
  // register sets that swap roles), unconditionally -- the last batch is requested twice: behind a
  // branch the compiler's waitcnt pass would have to drain every outstanding load at the join.
  // A per-channel coefficient series is still loaded by the batch that uses it (its loads come
  // before the look-ahead in program order, so waiting for them leaves the look-ahead in flight).
  const int64_t nfull = p.n / B;
  if (nfull > 0) {
    double xa[B], xb[B];
    load_x(std::true_type{}, 0, B, xa);
    for (int64_t i = 0;;) {
      batch(std::true_type{}, i * B, B, xa, [&] { load_x(std::true_type{}, ((i + 1 < nfull) ? i + 1 : nfull - 1) * B, B, xb); });
      if (++i >= nfull) break;
      batch(std::true_type{}, i * B, B, xb, [&] { load_x(std::true_type{}, ((i + 1 < nfull) ? i + 1 : nfull - 1) * B, B, xa); });
      if (++i >= nfull) break;
    }
  }
  if (nfull * B < p.n) {
    double xt[B];
    load_x(std::false_type{}, nfull * B, (int)(p.n - nfull * B), xt);
    batch(std::false_type{}, nfull * B, (int)(p.n - nfull * B), xt, [] {});
  }
#pragma unroll
  for (int k = 1; k < NB; ++k)
    if (k < p.nb) p.xh[(int64_t)(k - 1) * p.channels + c] = d[k];
#pragma unroll
  for (int k = 1; k < NA; ++k)
    if (k < p.na) p.yh[(int64_t)(k - 1) * p.channels + c] = m[k];
}

// ---------------------------------------------------------------------------
// k_tv_one: a single stream.  One wave; the 64 lanes fetch 64 consecutive samples of x and of
// every coefficient series with one coalesced load each, one batch ahead of the recurrence, so the
// memory latency is off the serial path.  The recurrence itself is computed by all lanes alike
// (the operands of step u are broadcast with v_readlane from lane u); lane u keeps y[u] and the
// batch is stored with one coalesced store.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double lane_value(double v, int lane) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)bits, lane);
  const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// branch-free pick between two doubles with a wave-uniform all-ones / all-zeros mask
__device__ __forceinline__ double pick(unsigned long long mask, double yes, double no) {
  const unsigned long long a = (unsigned long long)__double_as_longlong(yes);
  const unsigned long long b = (unsigned long long)__double_as_longlong(no);
  return __longlong_as_double((long long)((a & mask) | (b & ~mask)));
}

// a load the compiler does not track: the wave waits for it by hand (wait_loaded) so that the
// next batch can stay in flight across the whole recurrence of the current one
__device__ __forceinline__ double load_untracked(const double *ptr) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}

template <int NB, int NA, unsigned PB = kTvAny, unsigned PA = kTvAny>
__global__ __launch_bounds__(64) void k_tv_one(TvArgs p) {
  constexpr bool STATIC = PB != kTvAny;     // compile-time presence masks, as in k_tv
  const int lane = threadIdx.x;
  constexpr int kLoads = NB + NA;          // loads per batch: x, NB numerator and NA-1 denominator series
  // Untracked (inline-asm) loads only in the small variant.  Above 256 registers hipcc parks
  // values in AGPRs right after the asm statement and reuses the VGPR; a load still in flight
  // would then land in a register that already holds something else (seen as a memory
  // aperture violation with the 17-tap variant).
  constexpr bool kUntracked = NB <= 3 && NA <= 3;
  double d[NB], m[NA];
#pragma unroll
  for (int k = 1; k < NB; ++k) d[k] = (k < p.nb) ? p.xh[k - 1] : 0.0;
#pragma unroll
  for (int k = 1; k < NA; ++k) m[k] = (k < p.na) ? p.yh[k - 1] : 0.0;
  double cb[NB], nca[NA];
  unsigned long long b_on[NB], b_ser[NB], a_on[NA], a_ser[NA], a_neg[NA];   // masks
  // (taps that are not series carry series = x, stride 0 from the host: every fetch below is an
  // unconditional load from a valid address, and no pointer table has to live in registers)
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    cb[k] = p.b.value[k];
    b_on[k] = p.b.kind[k] != 0 ? ~0ull : 0ull;
    b_ser[k] = p.b.kind[k] == 2 ? ~0ull : 0ull;
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    nca[k] = -p.a.value[k];
    a_on[k] = (k > 0 && p.a.kind[k] != 0) ? ~0ull : 0ull;
    a_ser[k] = (k > 0 && p.a.kind[k] == 2) ? ~0ull : 0ull;
    a_neg[k] = p.a.negated[k] != 0 ? ~0ull : 0ull;
  }
  const bool divide = p.gain_mode == 1;
  const unsigned long long negate = p.gain_mode == 2 ? ~0ull : 0ull, no_terms = p.n_terms == 0 ? ~0ull : 0ull;
  const double gain = p.gain, zero = p.zero;

  struct Batch { double x, sb[NB], sa[NA]; };
  auto fetch = [&](int64_t n0, Batch &t) {
    int64_t i = n0 + lane;
    i = i < p.n ? i : p.n - 1;             // past the end: re-read the last sample (never used)
    if constexpr (kUntracked) {
      t.x = load_untracked(p.x + i * p.sxn);
#pragma unroll
      for (int k = 0; k < NB; ++k) t.sb[k] = load_untracked(p.b.series[k] + i * p.b.sn[k]);
#pragma unroll
      for (int k = 1; k < NA; ++k) t.sa[k] = load_untracked(p.a.series[k] + i * p.a.sn[k]);
    } else {
      t.x = p.x[i * p.sxn];
#pragma unroll
      for (int k = 0; k < NB; ++k) t.sb[k] = p.b.series[k][i * p.b.sn[k]];
#pragma unroll
      for (int k = 1; k < NA; ++k) t.sa[k] = p.a.series[k][i * p.a.sn[k]];
    }
  };
  // the batch fetched before the most recent one has landed once at most kLoads loads (the most
  // recent batch) plus the y store in between are outstanding; tie the registers to the wait
  auto wait_loaded = [&](Batch &t) {
    if constexpr (!kUntracked) return;       // hipcc waits for its own loads
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads) : "memory");
    asm volatile("" : "+v"(t.x));
#pragma unroll
    for (int k = 0; k < NB; ++k) asm volatile("" : "+v"(t.sb[k]));
#pragma unroll
    for (int k = 1; k < NA; ++k) asm volatile("" : "+v"(t.sa[k]));
  };
  auto recur = [&](const Batch &t, int64_t n0, auto div_tag) {
    constexpr bool DIV = decltype(div_tag)::value;   // the division sequence must not sit in the other loop
    const int cnt = (p.n - n0 < 64) ? (int)(p.n - n0) : 64;
    double ybuf = 0.0;
    for (int u = 0; u < cnt; ++u) {
      d[0] = lane_value(t.x, u);
      double acc = -0.0;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (STATIC && !((PB >> k) & 1u)) continue;
        const double coef = pick(b_ser[k], lane_value(t.sb[k], u), cb[k]);
        if (STATIC) acc = acc + coef * d[k];
        else acc = pick(b_on[k], acc + coef * d[k], acc);
      }
#pragma unroll
      for (int k = 1; k < NA; ++k) {
        if (STATIC && !((PA >> (k - 1)) & 1u)) continue;
        const double sv = lane_value(t.sa[k], u);
        const double coef = pick(a_ser[k], pick(a_neg[k], sv, -sv), nca[k]);
        if (STATIC) acc = acc + coef * m[k];
        else acc = pick(a_on[k], acc + coef * m[k], acc);
      }
      if constexpr (DIV) acc = acc / gain;
      acc = pick(negate, -acc, acc);
      if (!STATIC) acc = pick(no_terms, zero, acc);
      ybuf = (lane == u) ? acc : ybuf;
#pragma unroll
      for (int k = NA - 1; k > 1; --k) m[k] = m[k - 1];
      if (NA > 1) m[1] = acc;
#pragma unroll
      for (int k = NB - 1; k > 0; --k) d[k] = d[k - 1];
    }
    if (lane < cnt) p.y[(n0 + lane) * p.syn] = ybuf;
  };

  Batch A, B;
  fetch(0, A);
  for (int64_t n0 = 0; n0 < p.n; n0 += 128) {
    fetch(n0 + 64, B);
    wait_loaded(A);
    if (divide) recur(A, n0, std::true_type{});
    else recur(A, n0, std::false_type{});
    fetch(n0 + 128, A);
    wait_loaded(B);
    if (n0 + 64 < p.n) {
      if (divide) recur(B, n0 + 64, std::true_type{});
      else recur(B, n0 + 64, std::false_type{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
#pragma unroll
    for (int k = 1; k < NB; ++k)
      if (k < p.nb) p.xh[k - 1] = d[k];
#pragma unroll
    for (int k = 1; k < NA; ++k)
      if (k < p.na) p.yh[k - 1] = m[k];
  }
}

}  // namespace alz

extern "C" {

int alz_tv_process_dev(int nb, const alz_tv_tap_t *b, int na, const alz_tv_tap_t *a, int64_t channels,
                       const double *x_dev, double *y_dev, int64_t n, int layout, int64_t ldx, int64_t ldy,
                       double *xh_dev, double *yh_dev, double zero, int device, void *stream) {
  if (!b || !a || !x_dev || !y_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (nb < 1 || na < 1 || channels < 1 || n < 0) return alz::fail(ALZ_E_ARG, "nb, na, channels >= 1 and n >= 0 required");
  if (nb > alz::kTvMax || na > alz::kTvMax)
    return alz::fail(ALZ_E_UNSUPPORTED, "time-varying filters are limited to 17 taps per side");
  if ((nb > 1 && !xh_dev) || (na > 1 && !yh_dev)) return alz::fail(ALZ_E_ARG, "history arrays required");
  if (layout != ALZ_TIME_MAJOR && layout != ALZ_CHAN_MAJOR) return alz::fail(ALZ_E_ARG, "unknown layout");
  if (a[0].series_dev) return alz::fail(ALZ_E_ARG, "a0 must be constant (normalise a series gain on the host)");
  if (a[0].value == 0.0) return alz::fail(ALZ_E_ZERO_GAIN, "Invalid filter gain");
  alz::TvArgs p;
  p.x = x_dev; p.y = y_dev; p.n = n; p.channels = channels;
  if (layout == ALZ_TIME_MAJOR) { p.sxn = ldx; p.sxc = 1; p.syn = ldy; p.syc = 1; }
  else { p.sxn = 1; p.sxc = ldx; p.syn = 1; p.syc = ldy; }
  p.nb = nb; p.na = na; p.xh = xh_dev; p.yh = yh_dev; p.zero = zero; p.n_terms = 0;
  for (int k = 0; k < alz::kTvMax; ++k) {
    const alz_tv_tap_t *tb = k < nb ? &b[k] : nullptr, *ta = (k < na && k > 0) ? &a[k] : nullptr;
    p.b.kind[k] = !tb ? 0 : tb->series_dev ? 2 : (tb->value != 0.0 ? 1 : 0);
    const bool bs = p.b.kind[k] == 2;
    p.b.value[k] = tb ? tb->value : 0.0; p.b.series[k] = bs ? tb->series_dev : x_dev;
    p.b.sn[k] = bs ? tb->stride_n : 0; p.b.sc[k] = bs ? tb->stride_c : 0;
    p.a.kind[k] = !ta ? 0 : ta->series_dev ? 2 : (ta->value != 0.0 ? 1 : 0);
    const bool as = p.a.kind[k] == 2;
    p.a.value[k] = ta ? ta->value : 0.0; p.a.series[k] = as ? ta->series_dev : x_dev;
    p.a.sn[k] = as ? ta->stride_n : 0; p.a.sc[k] = as ? ta->stride_c : 0;
    p.a.negated[k] = (as && (ta->flags & ALZ_TV_NEGATED)) ? 1 : 0; p.b.negated[k] = 0;
    p.n_terms += (p.b.kind[k] != 0) + (p.a.kind[k] != 0);
  }
  p.gain = a[0].value;
  p.gain_mode = p.gain == 1.0 ? 0 : p.gain == -1.0 ? 2 : 1;
  alz::note_kernel("");
  if (n == 0) return ALZ_OK;
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  // a bank steered by control streams (every series shared by the channels): the two-wave kernel
  // takes the full tiles, the lane-per-channel kernels below continue with the ragged rest
  static const bool duo_off = ALZ_TUNE("ALZ_TV_DUO", 1) == 0;
  if (!duo_off && channels > 1 && nb <= 3 && na <= 3 && p.gain_mode == 0) {
    int kind[5], negated[5];
    double value[5];
    const double *series[5];
    bool ok = true;
    for (int t = 0; t < 5; ++t) {
      const alz::TvSide &sd = t < 3 ? p.b : p.a;
      const int k = t < 3 ? t : t - 2;
      kind[t] = sd.kind[k]; value[t] = sd.value[k]; series[t] = sd.series[k]; negated[t] = sd.negated[k];
      if (sd.kind[k] == 2 && (sd.sc[k] != 0 || sd.sn[k] != 1)) ok = false;
    }
    int64_t done = 0;
    if (ok) {
      const int rc = alz::launch_tvduo(p.x, p.y, p.n, ldx, ldy, layout == ALZ_CHAN_MAJOR, channels, nb, na, kind, value,
                                       series, negated, xh_dev, yh_dev, (hipStream_t)stream, &done);
      if (rc) { if (prev != device) (void)hipSetDevice(prev); return rc; }
    }
    if (done > 0) {
      alz::note_kernel("k_tvduo", true);
      p.x += done * p.sxn; p.y += done * p.syn; p.n -= done;
      for (int k = 0; k < alz::kTvMax; ++k) {
        if (p.b.kind[k] == 2) p.b.series[k] += done * p.b.sn[k];
        if (p.a.kind[k] == 2) p.a.series[k] += done * p.a.sn[k];
      }
      if (p.n == 0) { if (prev != device) (void)hipSetDevice(prev); return ALZ_OK; }
    }
  }
  // per-channel coefficient series on time-major rows (``repeat(ndarray)``-style coefficient Streams): the three-wave
  // kernel takes the full tiles, the lane-per-channel kernels below continue with the ragged rest
  if (ALZ_TUNE("ALZ_TV_PC", 1) && channels > 1 && nb <= 3 && na <= 3 && p.gain_mode == 0 && layout == ALZ_TIME_MAJOR) {
    int kind[5], negated[5];
    double value[5];
    const double *series[5];
    int64_t sld[5];
    bool ok = true, any = false;
    for (int t = 0; t < 5; ++t) {
      const alz::TvSide &sd = t < 3 ? p.b : p.a;
      const int k = t < 3 ? t : t - 2;
      kind[t] = sd.kind[k]; value[t] = sd.value[k]; series[t] = sd.kind[k] == 2 ? sd.series[k] : nullptr;
      negated[t] = sd.negated[k]; sld[t] = sd.sn[k];
      if (sd.kind[k] == 2) {
        any = true;
        if (sd.sc[k] != 1 || sd.sn[k] < channels) ok = false;     // one coefficient per channel, rows at least a bank wide
      }
    }
    int64_t done = 0;
    if (ok && any) {
      const int rc = alz::launch_tvpc(p.x, p.y, p.n, ldx, ldy, channels, nb, na, kind, value, series, sld, negated,
                                      xh_dev, yh_dev, (hipStream_t)stream, &done);
      if (rc) { if (prev != device) (void)hipSetDevice(prev); return rc; }
    }
    if (done > 0) {
      alz::note_kernel("k_tvpc", true);
      p.x += done * p.sxn; p.y += done * p.syn; p.n -= done;
      for (int k = 0; k < alz::kTvMax; ++k) {
        if (p.b.kind[k] == 2) p.b.series[k] += done * p.b.sn[k];
        if (p.a.kind[k] == 2) p.a.series[k] += done * p.a.sn[k];
      }
      if (p.n == 0) { if (prev != device) (void)hipSetDevice(prev); return ALZ_OK; }
    }
  }
  const unsigned grid = (unsigned)((channels + 63) / 64);
  if (channels == 1 && nb <= 3 && na <= 3) {
    unsigned pb = 0, pa = 0;
    for (int k = 0; k < 3; ++k) pb |= (p.b.kind[k] != 0) << k;
    for (int k = 1; k < 3; ++k) pa |= (p.a.kind[k] != 0) << (k - 1);
    void (*fn)(alz::TvArgs) = alz::k_tv_one<3, 3>;
#define ALZ_TV_PAT(PB_, PA_) if (pb == PB_ && pa == PA_) fn = alz::k_tv_one<3, 3, PB_, PA_>;
    ALZ_TV_PAT(1, 1) ALZ_TV_PAT(3, 1) ALZ_TV_PAT(1, 3) ALZ_TV_PAT(3, 3) ALZ_TV_PAT(5, 3) ALZ_TV_PAT(7, 3)
    ALZ_TV_PAT(1, 2) ALZ_TV_PAT(1, 0) ALZ_TV_PAT(2, 0) ALZ_TV_PAT(3, 0) ALZ_TV_PAT(4, 0) ALZ_TV_PAT(7, 0)
#undef ALZ_TV_PAT
    hipLaunchKernelGGL(fn, dim3(1), dim3(64), 0, (hipStream_t)stream, p);
    alz::note_kernel("k_tv_one<3,3>", true);
  }
  else if (channels == 1) {
    hipLaunchKernelGGL((alz::k_tv_one<alz::kTvMax, alz::kTvMax>), dim3(1), dim3(64), 0, (hipStream_t)stream, p);
    alz::note_kernel("k_tv_one", true);
  }
  else if (nb <= 3 && na <= 3) {
    unsigned pb = 0, pa = 0;
    for (int k = 0; k < 3; ++k) pb |= (p.b.kind[k] != 0) << k;
    for (int k = 1; k < 3; ++k) pa |= (p.a.kind[k] != 0) << (k - 1);
    void (*fn)(alz::TvArgs) = alz::k_tv<3, 3, 16>;
#define ALZ_TV_PAT(PB_, PA_) \
  if (pb == PB_ && pa == PA_) fn = alz::k_tvp<PB_, PA_, ALZ_TVP_B>;
    ALZ_TV_PAT(1, 1) ALZ_TV_PAT(3, 1) ALZ_TV_PAT(1, 3) ALZ_TV_PAT(3, 3) ALZ_TV_PAT(5, 3) ALZ_TV_PAT(7, 3)
    ALZ_TV_PAT(1, 2) ALZ_TV_PAT(1, 0) ALZ_TV_PAT(2, 0) ALZ_TV_PAT(3, 0) ALZ_TV_PAT(4, 0) ALZ_TV_PAT(7, 0)
#undef ALZ_TV_PAT
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64), 0, (hipStream_t)stream, p);
    alz::note_kernel(fn == alz::k_tv<3, 3, 16> ? "k_tv<3,3>" : "k_tvp", true);
  }
  else {
    hipLaunchKernelGGL((alz::k_tv<alz::kTvMax, alz::kTvMax, 2>), dim3(grid), dim3(64), 0, (hipStream_t)stream, p);
    alz::note_kernel("k_tv", true);
  }
  hipError_t e = hipGetLastError();
  if (prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) return alz::fail(ALZ_E_HIP, std::string("k_tv launch: ") + hipGetErrorString(e));
  return ALZ_OK;
}

}  // extern "C"
