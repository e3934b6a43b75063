// alz_comb.hip -- sparse long-delay sections: comb filters and their relatives.
//
// Replaces LinearFilter.__call__ (reference audiolazy/lazy_filters.py:141-264) for filters such
// as comb.fb / comb.tau  y[n] = x[n] + alpha*y[n-D]  (lazy_filters.py:1090-1147), comb.ff
// (:1150-1173), their linearize()d fractional-delay forms (two adjacent taps, :339-373) and
// karplus_strong (lazy_synth.py:624-657): a handful of non-zero taps at large delays.  The
// reference's generated loop shifts all D memory variables on every sample (:254-255); here the
// delay line is simply the block itself -- y[n-D] is read back from the output rows written D
// steps earlier (L2 / Infinity Cache), times before the block come from the state arrays.
//
// Because every feedback delay is >= kMinDelay rows, a batch of 8 consecutive rows has no
// dependence inside it: the loads of a batch are all issued first and the recurrence is not
// latency-bound.  Arithmetic is the same bit-exact DF-I sum (ascending numerator delays, then
// ascending denominator delays, separately rounded mul/add, absent taps absent).
// Time-major blocks, lane = channel (512-byte coalesced rows), x and y distinct.
#include "alz_common.h"

namespace alz {

static constexpr int kMaxTaps = 8;      // per side
static constexpr int kMinDelay = 16;    // smallest feedback delay this kernel accepts
static constexpr int kBatch = 8;        // rows per batch (< kMinDelay)

struct SArgs {
  const double *x;
  double *y;
  int64_t n, sxn, syn;
  int64_t channels, n_inputs, n_sets;
  int64_t c_first, c_end;
  int mode, map_input;
  int nb, na;
  int nfb, nff;                 // number of present taps
  int kb[kMaxTaps], ka[kMaxTaps];
  const double *b, *a;
  const double *xh, *yh;        // histories: xh[k*C + c] = x[-1-k]
};

__global__ __launch_bounds__(64) void k_sparse(SArgs p) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  int64_t in, set;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
  double bc[kMaxTaps], nac[kMaxTaps];
#pragma unroll
  for (int j = 0; j < kMaxTaps; ++j) {
    bc[j] = (j < p.nff) ? p.b[(int64_t)p.kb[j] * p.n_sets + set] : 0.0;
    nac[j] = (j < p.nfb) ? -p.a[(int64_t)p.ka[j] * p.n_sets + set] : 0.0;
  }
  const double *xc = p.x + in;
  double *yc = p.y + c;
  const double *xhc = p.xh + c;
  const double *yhc = p.yh + c;

  auto xval = [&](int64_t t) -> double { return t >= 0 ? xc[t * p.sxn] : xhc[(-t - 1) * p.channels]; };
  auto yval = [&](int64_t t) -> double { return t >= 0 ? yc[t * p.syn] : yhc[(-t - 1) * p.channels]; };

  for (int64_t n0 = 0; n0 < p.n; n0 += kBatch) {
    const int rows = (p.n - n0 < kBatch) ? (int)(p.n - n0) : kBatch;
    double xv[kMaxTaps][kBatch], yv[kMaxTaps][kBatch];
    // every load of the batch first (no feedback delay is shorter than the batch)
#pragma unroll
    for (int j = 0; j < kMaxTaps; ++j) {
      if (j < p.nff) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) xv[j][u] = (u < rows) ? xval(n0 + u - p.kb[j]) : 0.0;
      }
      if (j < p.nfb) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) yv[j][u] = (u < rows) ? yval(n0 + u - p.ka[j]) : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      if (u < rows) {
        double acc = -0.0;   // additive identity: the first present term initialises the sum
#pragma unroll
        for (int j = 0; j < kMaxTaps; ++j)
          if (j < p.nff) acc = acc + bc[j] * xv[j][u];
#pragma unroll
        for (int j = 0; j < kMaxTaps; ++j)
          if (j < p.nfb) acc = acc + nac[j] * yv[j][u];
        yc[(n0 + u) * p.syn] = acc;
      }
    }
  }
}

// histories after the block, for both delay lines (written to the spare halves of the slabs)
__global__ void k_sparse_state(SArgs p, double *xh_new, double *yh_new) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? c % p.n_inputs : c;
  for (int k = (int)blockIdx.y; k < p.nb - 1; k += (int)gridDim.y) {
    const int64_t t = p.n - 1 - k;
    xh_new[(int64_t)k * p.channels + c] = (t >= 0) ? p.x[t * p.sxn + in] : p.xh[(-t - 1) * p.channels + c];
  }
  for (int k = (int)blockIdx.y; k < p.na - 1; k += (int)gridDim.y) {
    const int64_t t = p.n - 1 - k;
    yh_new[(int64_t)k * p.channels + c] = (t >= 0) ? p.y[t * p.syn + c] : p.yh[(-t - 1) * p.channels + c];
  }
}

__global__ void k_copy_rows(double *dst, const double *src, int64_t count, int64_t stride, int rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  for (int k = (int)blockIdx.y; k < rows; k += (int)gridDim.y) dst[(int64_t)k * stride + i] = src[(int64_t)k * stride + i];
}

// Sparse section with every feedback delay >= kMinDelay, time-major, x != y, a0 == 1, uniform
// zero pattern.  The host-side tap lists come from the coefficient scan done at create time.
int launch_sparse(const SectionDev &sec, const BlockIO &io, hipStream_t stream, bool *taken,
                  const char **kernel_name) {
  *taken = false;
  if (sec.n_ff < 0 || sec.n_ff > kMaxTaps || sec.n_fb > kMaxTaps || sec.n_fb < 1) return ALZ_OK;
  if (sec.n_ff + sec.n_fb == 0 || !sec.uniform || sec.any_div) return ALZ_OK;
  if (sec.tap_a[0] < kMinDelay) return ALZ_OK;
  if (!(io.sxc == 1 && io.syc == 1) || io.x == io.y) return ALZ_OK;
  SArgs p;
  p.x = io.x; p.y = io.y; p.n = io.n; p.sxn = io.sxn; p.syn = io.syn;
  p.channels = io.channels; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.c_first = io.c_first; p.c_end = io.c_first + io.c_count;
  p.mode = io.mode; p.map_input = io.map_input;
  p.nb = sec.nb; p.na = sec.na; p.nff = sec.n_ff; p.nfb = sec.n_fb;
  for (int j = 0; j < kMaxTaps; ++j) { p.kb[j] = sec.tap_b[j]; p.ka[j] = sec.tap_a[j]; }
  p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.yh = sec.yh;
  const unsigned gx = (unsigned)((io.c_count + 63) / 64);
  hipLaunchKernelGGL(k_sparse, dim3(gx), dim3(64), 0, stream, p);
  const int64_t nx = (int64_t)(sec.nb - 1) * io.channels, ny = (int64_t)(sec.na - 1) * io.channels;
  double *xh_new = sec.xh + nx, *yh_new = sec.yh + ny;
  hipLaunchKernelGGL(k_sparse_state, dim3(gx, 16), dim3(64), 0, stream, p, xh_new, yh_new);
  const unsigned gc = (unsigned)((io.c_count + 255) / 256);
  if (sec.nb > 1)
    hipLaunchKernelGGL(k_copy_rows, dim3(gc, 16), dim3(256), 0, stream, sec.xh + io.c_first,
                       xh_new + io.c_first, io.c_count, io.channels, sec.nb - 1);
  if (sec.na > 1)
    hipLaunchKernelGGL(k_copy_rows, dim3(gc, 16), dim3(256), 0, stream, sec.yh + io.c_first,
                       yh_new + io.c_first, io.c_count, io.channels, sec.na - 1);
  ALZ_HIP_CHECK(hipGetLastError());
  *taken = true;
  *kernel_name = "k_sparse";
  return ALZ_OK;
}

}  // namespace alz
