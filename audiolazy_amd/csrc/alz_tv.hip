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
// load per wave.  Arithmetic: separately rounded mul/add in the reference's order (bit-exact).
#include "alz_common.h"

namespace alz {

static constexpr int kTvMax = 9;   // taps per side the register windows hold (delays 0 .. 8)

struct TvSide {
  int kind[kTvMax];               // 0 absent, 1 constant, 2 series
  double value[kTvMax];
  const double *series[kTvMax];
  int64_t sn[kTvMax], sc[kTvMax];
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

template <int NB, int NA, int B>
__global__ __launch_bounds__(64) void k_tv(TvArgs p) {
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.channels) return;
  double d[NB], m[NA];            // d[k] = x[n-k], m[k] = y[n-k]  (d[0], m[0] are scratch)
#pragma unroll
  for (int k = 1; k < NB; ++k) d[k] = (k < p.nb) ? p.xh[(int64_t)(k - 1) * p.channels + c] : 0.0;
#pragma unroll
  for (int k = 1; k < NA; ++k) m[k] = (k < p.na) ? p.yh[(int64_t)(k - 1) * p.channels + c] : 0.0;
  double cb[NB], nca[NA];         // constants (denominator already negated)
#pragma unroll
  for (int k = 0; k < NB; ++k) cb[k] = p.b.value[k];
#pragma unroll
  for (int k = 0; k < NA; ++k) nca[k] = -p.a.value[k];
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
      if (p.b.kind[k] == 2) {
        const double *s = p.b.series[k] + c * p.b.sc[k];
#pragma unroll
        for (int u = 0; u < B; ++u) sb[k][u] = (u < cnt) ? s[(n0 + u) * p.b.sn[k]] : 0.0;
      }
    }
#pragma unroll
    for (int k = 1; k < NA; ++k) {
      if (p.a.kind[k] == 2) {
        const double *s = p.a.series[k] + c * p.a.sc[k];
#pragma unroll
        for (int u = 0; u < B; ++u) sa[k][u] = (u < cnt) ? -s[(n0 + u) * p.a.sn[k]] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {
      if (u < cnt) {
        d[0] = xv[u];
        double acc = -0.0;         // additive identity: the first present term initialises the sum
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (p.b.kind[k] == 1) acc = acc + cb[k] * d[k];
          else if (p.b.kind[k] == 2) acc = acc + sb[k][u] * d[k];
        }
#pragma unroll
        for (int k = 1; k < NA; ++k) {
          if (p.a.kind[k] == 1) acc = acc + nca[k] * m[k];
          else if (p.a.kind[k] == 2) acc = acc + sa[k][u] * m[k];
        }
        if (p.gain_mode == 1) acc = acc / p.gain;
        else if (p.gain_mode == 2) acc = -acc;
        if (p.n_terms == 0) acc = p.zero;
        yc[(n0 + u) * p.syn] = acc;
#pragma unroll
        for (int k = NA - 1; k > 1; --k) m[k] = m[k - 1];
        if (NA > 1) m[1] = acc;
#pragma unroll
        for (int k = NB - 1; k > 0; --k) d[k] = d[k - 1];
      }
    }
  }
#pragma unroll
  for (int k = 1; k < NB; ++k)
    if (k < p.nb) p.xh[(int64_t)(k - 1) * p.channels + c] = d[k];
#pragma unroll
  for (int k = 1; k < NA; ++k)
    if (k < p.na) p.yh[(int64_t)(k - 1) * p.channels + c] = m[k];
}

}  // namespace alz

extern "C" {

int alz_tv_process_dev(int nb, const alz_tv_tap_t *b, int na, const alz_tv_tap_t *a, int64_t channels,
                       const double *x_dev, double *y_dev, int64_t n, int layout, int64_t ldx, int64_t ldy,
                       double *xh_dev, double *yh_dev, double zero, int device, void *stream) {
  if (!b || !a || !x_dev || !y_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (nb < 1 || na < 1 || channels < 1 || n < 0) return alz::fail(ALZ_E_ARG, "nb, na, channels >= 1 and n >= 0 required");
  if (nb > alz::kTvMax || na > alz::kTvMax)
    return alz::fail(ALZ_E_UNSUPPORTED, "time-varying filters are limited to 9 taps per side");
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
    p.b.value[k] = tb ? tb->value : 0.0; p.b.series[k] = tb ? tb->series_dev : nullptr;
    p.b.sn[k] = tb ? tb->stride_n : 0; p.b.sc[k] = tb ? tb->stride_c : 0;
    p.a.kind[k] = !ta ? 0 : ta->series_dev ? 2 : (ta->value != 0.0 ? 1 : 0);
    p.a.value[k] = ta ? ta->value : 0.0; p.a.series[k] = ta ? ta->series_dev : nullptr;
    p.a.sn[k] = ta ? ta->stride_n : 0; p.a.sc[k] = ta ? ta->stride_c : 0;
    p.n_terms += (p.b.kind[k] != 0) + (p.a.kind[k] != 0);
  }
  p.gain = a[0].value;
  p.gain_mode = p.gain == 1.0 ? 0 : p.gain == -1.0 ? 2 : 1;
  if (n == 0) return ALZ_OK;
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  const unsigned grid = (unsigned)((channels + 63) / 64);
  if (nb <= 3 && na <= 3)
    hipLaunchKernelGGL((alz::k_tv<3, 3, 8>), dim3(grid), dim3(64), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((alz::k_tv<alz::kTvMax, alz::kTvMax, 4>), dim3(grid), dim3(64), 0, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  if (prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) return alz::fail(ALZ_E_HIP, std::string("k_tv launch: ") + hipGetErrorString(e));
  return ALZ_OK;
}

}  // extern "C"
