// alz_iir.hip -- Direct-Form-I section kernels, one wavefront lane per channel.
//
// Replaces the generated per-sample loop of LinearFilter.__call__
// (reference audiolazy/lazy_filters.py:197-257) for many channels at once:
//
//     m0 = b0*d0 + b1*d1 + ... + (-a1)*m1 + (-a2)*m2 + ...     (left to right)
//     m0 = (m0) / a0  when a0 != 1 ;  yield m0 ; shift m's ; shift d's
//
// Bit-exactness rules (the whole file is built with -ffp-contract=off):
//   * separately rounded v_mul_f64 / v_add_f64, never v_fma_f64;
//   * a coefficient equal to zero is ABSENT from the sum (lazy_filters.py:209,
//     :223), which matters for signed zeros and non-finite samples;
//   * "1 * d" / "-1 * d" special cases (:205-208, :219-222) are the same doubles
//     as the products, so they need no code;
//   * division by a0 is the IEEE f64 division (x / 1 == x, x / -1 == -x, so one
//     code path covers :233-237), compiled out when every a0 == 1.
//
// Three kernel families:
//   k_small<PB,PA,DIV>   nb,na <= 3, zero pattern uniform over the bank and
//                        folded at compile time (state in VGPRs);
//   k_masked<NB,NA>      per-lane zero pattern, taps padded up to (NB,NA),
//                        select instead of branch (state in VGPRs);
//   k_generic            any nb/na, history read back from HBM (slow catch-all).
#include "alz_common.h"

namespace alz {

static constexpr int kUnroll = 8;

struct KArgs {
  const double *x;
  double *y;
  int64_t n, sxn, sxc, syn, syc;
  int64_t channels, n_inputs, n_sets;
  int64_t c_first, c_end;  // channel range of this launch
  int mode;
  int map_input;  // OUTER mode, first section: channel c reads input c % n_inputs
  int nb, na;
  const double *b, *a;
  double *xh, *yh;
  double zero;
  unsigned long long x_and;   // k_small: mask applied to every input sample (all ones, or abs: sign bit cleared)
};

__device__ __forceinline__ void lane_ids(const KArgs &p, int64_t c, int64_t &in, int64_t &set) {
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
}

// ---------------------------------------------------------------------------
// k_small: compile-time tap pattern.  PB bit k <=> b_k present (k = 0..2),
// PA bit k-1 <=> a_k present (k = 1..2).
// ---------------------------------------------------------------------------
template <unsigned PB, unsigned PA, bool DIV>
__device__ __forceinline__ double small_step(double d0, double d1, double d2, double m1, double m2,
                                             double b0, double b1, double b2, double na1,
                                             double na2, double a0) {
  double acc = 0.0;
  bool first = true;
  // the `first` flag folds at compile time: the first present term initialises
  // the accumulator, exactly like the first operand of the generated expression
  if constexpr (PB & 1u) { acc = b0 * d0; first = false; }
  if constexpr (PB & 2u) { const double t = b1 * d1; acc = first ? t : acc + t; first = false; }
  if constexpr (PB & 4u) { const double t = b2 * d2; acc = first ? t : acc + t; first = false; }
  if constexpr (PA & 1u) { const double t = na1 * m1; acc = first ? t : acc + t; first = false; }
  if constexpr (PA & 2u) { const double t = na2 * m2; acc = first ? t : acc + t; first = false; }
  if constexpr (DIV) acc = acc / a0;
  return acc;
}

template <unsigned PB, unsigned PA, bool DIV>
__global__ __launch_bounds__(64) void k_small(KArgs p) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  int64_t in, set;
  lane_ids(p, c, in, set);
  constexpr int NBH = (PB & 4u) ? 2 : (PB & 2u) ? 1 : 0;  // history depth actually read
  constexpr int NAH = (PA & 2u) ? 2 : (PA & 1u) ? 1 : 0;

  double b0 = 0, b1 = 0, b2 = 0, na1 = 0, na2 = 0, a0 = 1;
  if (PB & 1u) b0 = p.b[0 * p.n_sets + set];
  if (PB & 2u) b1 = p.b[1 * p.n_sets + set];
  if (PB & 4u) b2 = p.b[2 * p.n_sets + set];
  a0 = p.a[set];
  if (PA & 1u) na1 = -p.a[1 * p.n_sets + set];
  if (PA & 2u) na2 = -p.a[2 * p.n_sets + set];

  // history registers: the bank's state depth is nb-1 / na-1 (may exceed what
  // the present taps read; the extra slots are still shifted like the reference)
  double d1 = (p.nb > 1) ? p.xh[0 * p.channels + c] : 0.0;
  double d2 = (p.nb > 2) ? p.xh[1 * p.channels + c] : 0.0;
  double m1 = (p.na > 1) ? p.yh[0 * p.channels + c] : 0.0;
  double m2 = (p.na > 2) ? p.yh[1 * p.channels + c] : 0.0;
  (void)NBH; (void)NAH;

  const double *xp = p.x + in * p.sxc;
  double *yp = p.y + c * p.syc;
  int64_t n = 0;
  for (; n + kUnroll <= p.n; n += kUnroll) {
    double xv[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) xv[u] = xp[(n + u) * p.sxn];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const double d0 = __longlong_as_double(__double_as_longlong(xv[u]) & (long long)p.x_and);
      const double m0 = small_step<PB, PA, DIV>(d0, d1, d2, m1, m2, b0, b1, b2, na1, na2, a0);
      yp[(n + u) * p.syn] = m0;
      m2 = m1; m1 = m0; d2 = d1; d1 = d0;
    }
  }
  for (; n < p.n; ++n) {
    const double d0 = __longlong_as_double(__double_as_longlong(xp[n * p.sxn]) & (long long)p.x_and);
    const double m0 = small_step<PB, PA, DIV>(d0, d1, d2, m1, m2, b0, b1, b2, na1, na2, a0);
    yp[n * p.syn] = m0;
    m2 = m1; m1 = m0; d2 = d1; d1 = d0;
  }
  if (p.nb > 1) p.xh[0 * p.channels + c] = d1;
  if (p.nb > 2) p.xh[1 * p.channels + c] = d2;
  if (p.na > 1) p.yh[0 * p.channels + c] = m1;
  if (p.na > 2) p.yh[1 * p.channels + c] = m2;
}

// ---------------------------------------------------------------------------
// k_masked: per-lane zero pattern.  acc starts at -0.0 (x + -0.0 == x for every
// double, so the first present term initialises it exactly) and a term is kept
// only where its coefficient is non-zero.
// ---------------------------------------------------------------------------
template <int NB, int NA>
__global__ __launch_bounds__(64) void k_masked(KArgs p) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  int64_t in, set;
  lane_ids(p, c, in, set);

  double bc[NB], nac[NA], d[NB], m[NA];
  bool bz[NB], az[NA];
  int nterms = 0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    bc[k] = (k < p.nb) ? p.b[(int64_t)k * p.n_sets + set] : 0.0;
    bz[k] = bc[k] != 0.0;
    nterms += bz[k];
    d[k] = (k >= 1 && k < p.nb) ? p.xh[(int64_t)(k - 1) * p.channels + c] : 0.0;
  }
  const double a0 = p.a[set];
#pragma unroll
  for (int k = 1; k < NA; ++k) {
    const double av = (k < p.na) ? p.a[(int64_t)k * p.n_sets + set] : 0.0;
    nac[k] = -av;
    az[k] = av != 0.0;
    nterms += az[k];
    m[k] = (k < p.na) ? p.yh[(int64_t)(k - 1) * p.channels + c] : 0.0;
  }
  const bool all_zero = nterms == 0;
  const bool div = a0 != 1.0;

  const double *xp = p.x + in * p.sxc;
  double *yp = p.y + c * p.syc;
  for (int64_t n = 0; n < p.n; ++n) {
    d[0] = xp[n * p.sxn];
    double acc = -0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const double s = acc + bc[k] * d[k];
      acc = bz[k] ? s : acc;
    }
#pragma unroll
    for (int k = 1; k < NA; ++k) {
      const double s = acc + nac[k] * m[k];
      acc = az[k] ? s : acc;
    }
    if (div) acc = acc / a0;
    if (all_zero) acc = p.zero;
    yp[n * p.syn] = acc;
#pragma unroll
    for (int k = NA - 1; k > 1; --k) m[k] = m[k - 1];
    if (NA > 1) m[1] = acc;
#pragma unroll
    for (int k = NB - 1; k > 0; --k) d[k] = d[k - 1];
  }
#pragma unroll
  for (int k = 1; k < NB; ++k)
    if (k < p.nb) p.xh[(int64_t)(k - 1) * p.channels + c] = d[k];
#pragma unroll
  for (int k = 1; k < NA; ++k)
    if (k < p.na) p.yh[(int64_t)(k - 1) * p.channels + c] = m[k];
}

// ---------------------------------------------------------------------------
// k_generic: any order.  History comes from the block itself (HBM / caches) or,
// for times before the block, from the state arrays.  x and y must not alias.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_generic(KArgs p, double *xh_new, double *yh_new) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  int64_t in, set;
  lane_ids(p, c, in, set);
  const double *xp = p.x + in * p.sxc;
  double *yp = p.y + c * p.syc;
  const double a0 = p.a[set];
  const bool div = a0 != 1.0;
  int nterms = 0;
  for (int k = 0; k < p.nb; ++k) nterms += p.b[(int64_t)k * p.n_sets + set] != 0.0;
  for (int k = 1; k < p.na; ++k) nterms += p.a[(int64_t)k * p.n_sets + set] != 0.0;

  for (int64_t n = 0; n < p.n; ++n) {
    double acc = -0.0;
    for (int k = 0; k < p.nb; ++k) {
      const double bk = p.b[(int64_t)k * p.n_sets + set];
      if (bk == 0.0) continue;
      const int64_t i = n - k;
      const double v = (i >= 0) ? xp[i * p.sxn] : p.xh[(-i - 1) * p.channels + c];
      acc = acc + bk * v;
    }
    for (int k = 1; k < p.na; ++k) {
      const double ak = p.a[(int64_t)k * p.n_sets + set];
      if (ak == 0.0) continue;
      const int64_t i = n - k;
      const double v = (i >= 0) ? yp[i * p.syn] : p.yh[(-i - 1) * p.channels + c];
      acc = acc + (-ak) * v;
    }
    if (div) acc = acc / a0;
    if (nterms == 0) acc = p.zero;
    yp[n * p.syn] = acc;
  }
  // new histories (written to separate arrays; the caller swaps them in)
  for (int k = 0; k < p.nb - 1; ++k) {
    const int64_t i = p.n - 1 - k;
    xh_new[(int64_t)k * p.channels + c] = (i >= 0) ? xp[i * p.sxn] : p.xh[(-i - 1) * p.channels + c];
  }
  for (int k = 0; k < p.na - 1; ++k) {
    const int64_t i = p.n - 1 - k;
    yh_new[(int64_t)k * p.channels + c] = (i >= 0) ? yp[i * p.syn] : p.yh[(-i - 1) * p.channels + c];
  }
}

// rows [0, rows) x channels [c_first, c_first + c_count) of a [rows][channels] state slab
__global__ void k_copy_state(double *dst, const double *src, int64_t rows, int64_t channels,
                             int64_t c_first, int64_t c_count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c_count) return;
  const int64_t at = (i / c_count) * channels + c_first + i % c_count;
  dst[at] = src[at];
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
typedef void (*small_fn)(KArgs);

template <unsigned PB, unsigned PA>
static small_fn pick_div(bool div) {
  return div ? (small_fn)k_small<PB, PA, true> : (small_fn)k_small<PB, PA, false>;
}

template <unsigned PB>
static small_fn pick_pa(unsigned pa, bool div) {
  switch (pa) {
    case 0: return pick_div<PB, 0>(div);
    case 1: return pick_div<PB, 1>(div);
    case 2: return pick_div<PB, 2>(div);
    default: return pick_div<PB, 3>(div);
  }
}

static small_fn pick_small(unsigned pb, unsigned pa, bool div) {
  switch (pb) {
    case 0: return pick_pa<0>(pa, div);
    case 1: return pick_pa<1>(pa, div);
    case 2: return pick_pa<2>(pa, div);
    case 3: return pick_pa<3>(pa, div);
    case 4: return pick_pa<4>(pa, div);
    case 5: return pick_pa<5>(pa, div);
    case 6: return pick_pa<6>(pa, div);
    default: return pick_pa<7>(pa, div);
  }
}

// scratch for k_generic's new-state arrays is owned by the caller (api): the
// section's xh/yh point at arrays with room for a second copy right behind the
// first (see Bank::alloc in alz_api.hip).
int launch_section(const SectionDev &sec, const BlockIO &io, hipStream_t stream,
                   const char **kernel_name) {
  KArgs p;
  p.x = io.x; p.y = io.y; p.n = io.n;
  p.sxn = io.sxn; p.sxc = io.sxc; p.syn = io.syn; p.syc = io.syc;
  p.channels = io.channels; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.c_first = io.c_first; p.c_end = io.c_first + io.c_count;
  p.mode = io.mode; p.map_input = io.map_input; p.nb = sec.nb; p.na = sec.na;
  p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.yh = sec.yh; p.zero = io.zero;
  p.x_and = io.pre_op == ALZ_MAP_ABS ? 0x7fffffffffffffffull : ~0ull;
  const dim3 grid((unsigned)((io.c_count + 63) / 64)), block(64);
  const bool nonempty = (sec.present_b | sec.present_a) != 0;
  if (io.pre_op && !(io.pre_op == ALZ_MAP_ABS && sec.nb <= 3 && sec.na <= 3 && sec.uniform && nonempty))
    return fail(ALZ_E_UNSUPPORTED, "input map reached a kernel that does not fuse it");   // (alz_api.hip maps first)

  if (sec.nb <= 3 && sec.na <= 3 && sec.uniform && nonempty) {
    small_fn fn = pick_small(sec.present_b, sec.present_a, sec.any_div);
    hipLaunchKernelGGL(fn, grid, block, 0, stream, p);
    *kernel_name = "k_small";
  } else if (sec.nb <= 3 && sec.na <= 3) {
    hipLaunchKernelGGL((k_masked<3, 3>), grid, block, 0, stream, p);
    *kernel_name = "k_masked<3,3>";
  } else if (sec.nb <= 8 && sec.na <= 3) {
    hipLaunchKernelGGL((k_masked<8, 3>), grid, block, 0, stream, p);
    *kernel_name = "k_masked<8,3>";
  } else if (sec.nb <= 16 && sec.na <= 9) {
    hipLaunchKernelGGL((k_masked<16, 9>), grid, block, 0, stream, p);
    *kernel_name = "k_masked<16,9>";
  } else {
    bool taken = false;
    int rc = launch_sparse(sec, io, stream, &taken, kernel_name);
    if (rc) return rc;
    if (taken) return ALZ_OK;
    if (io.x == io.y) return fail(ALZ_E_ARG, "k_fir / k_generic cannot run in place");
    rc = launch_fir(sec, io, stream, &taken, kernel_name);
    if (rc) return rc;
    if (taken) return ALZ_OK;
    const int64_t nx = (int64_t)(sec.nb - 1) * io.channels;
    const int64_t ny = (int64_t)(sec.na - 1) * io.channels;
    double *xh_new = sec.xh + nx, *yh_new = sec.yh + ny;
    hipLaunchKernelGGL(k_generic, grid, block, 0, stream, p, xh_new, yh_new);
    // only the channels this launch covered have a new history (the others' rows are stale)
    const int64_t cx = (int64_t)(sec.nb - 1) * io.c_count, cy = (int64_t)(sec.na - 1) * io.c_count;
    if (cx > 0)
      hipLaunchKernelGGL(k_copy_state, dim3((unsigned)((cx + 255) / 256)), dim3(256), 0, stream,
                         sec.xh, xh_new, (int64_t)(sec.nb - 1), io.channels, io.c_first, io.c_count);
    if (cy > 0)
      hipLaunchKernelGGL(k_copy_state, dim3((unsigned)((cy + 255) / 256)), dim3(256), 0, stream,
                         sec.yh, yh_new, (int64_t)(sec.na - 1), io.channels, io.c_first, io.c_count);
    *kernel_name = "k_generic";
  }
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

}  // namespace alz
