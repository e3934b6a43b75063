// alz_lev.hip -- levinson_durbin in the reference's own dense form (lazy_lpc.py:115-136), bit-identical.
// (Its own translation unit: the fully unrolled O(order^3) kernels take a while to compile.)
#include "alz_common.h"

namespace alz {

// k_levinson_dense<P>: levinson_durbin exactly as the reference computes it (lazy_lpc.py:115-136) -- with
// its DENSE inner products  inner(a, b) = sum(acdata[|i-j|] * a_i * b_j for i.. for j..)  evaluated in
// the reference's order, O(order^3) per frame instead of the O(order^2) recursion of k_levinson_lane:
//   A = 1;  for m = 1..order:  B = A(1/z) z^-m;  A -= inner(A, z^-m) / inner(B, B) * B;   error = inner(A, A)
// so coefficients and error are bit-identical to the reference's (opt-in: ALZ_LPC_DENSE).  One lane per
// frame, everything in registers (P = order + 1 is a template parameter, all loops unrolled).  Terms
// whose factor is an exact zero are skipped where that cannot change the sum (x + (+-0) == x for the
// running sum, which starts at +0 and can never be -0); `la` mirrors the reference's dense numlist
// length, which shrinks when the top coefficient cancels to exactly zero.
template <int P>
__global__ __launch_bounds__(64) void k_levinson_dense(const double *__restrict__ r_in, int64_t n_frames, int n_lags,
                                                        double *__restrict__ coefs, double *__restrict__ err,
                                                        int *__restrict__ status) {
  const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (f >= n_frames) return;
  constexpr int order = P - 1;
  double ac[P], A[P], B[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    ac[i] = (i < n_lags) ? r_in[f * n_lags + i] : 0.0;      // zero-extended lags (lazy_lpc.py:117-118)
    A[i] = 0.0;
    B[i] = 0.0;
  }
  A[0] = 1.0;
  int la = 1, st = ALZ_OK;
#pragma unroll
  for (int m = 1; m <= order; ++m) {
    // B = A(1/z) * z**-m: B[m - i] = A[i] for the la dense coefficients of A, zero elsewhere
#pragma unroll
    for (int t = 0; t <= m; ++t) B[t] = (m - t < la) ? A[m - t] : 0.0;
    // inner(A, z**-m): of the (i, j) terms only j = m has a non-zero b_j (= 1)
    double num = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) num = (i < la) ? num + (ac[m - i] * A[i]) * 1.0 : num;
    double den = 0.0;
#pragma unroll
    for (int i = 0; i <= m; ++i) {
#pragma unroll
      for (int j = 0; j <= m; ++j) den = den + (ac[i > j ? i - j : j - i] * B[i]) * B[j];
    }
    bool ok = st == ALZ_OK;
    if (ok && den == 0.0) {                                   // ZeroDivisionError -> ParCorError (:132-133)
      st = ALZ_E_PARCOR;
      ok = false;
    }
    const double k = ok ? num / den : 0.0;
#pragma unroll
    for (int i = 0; i <= m; ++i) A[i] = ok ? A[i] - k * B[i] : A[i];
    if (ok) {
      la = m + 1;
#pragma unroll
      for (int t = m; t >= 1; --t)
        if (la == t + 1 && A[t] == 0.0) la = t;               // Poly drops exact-zero terms: the dense list shrinks
    }
  }
  double e = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) {
#pragma unroll
    for (int j = 0; j < P; ++j) e = (i < la && j < la) ? e + (ac[i > j ? i - j : j - i] * A[i]) * A[j] : e;
  }
#pragma unroll
  for (int i = 0; i < P; ++i) coefs[f * P + i] = (i < la) ? A[i] : 0.0;
  err[f] = e;
  status[f] = st;
}

// any order up to 63: the same arithmetic with run-time loops (arrays in scratch memory)
__global__ __launch_bounds__(64) void k_levinson_dense_any(const double *__restrict__ r_in, int64_t n_frames, int n_lags,
                                                            int order, double *__restrict__ coefs,
                                                            double *__restrict__ err, int *__restrict__ status) {
  const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (f >= n_frames) return;
  double ac[64], A[64], B[64];
  for (int i = 0; i <= order; ++i) {
    ac[i] = (i < n_lags) ? r_in[f * n_lags + i] : 0.0;
    A[i] = 0.0;
    B[i] = 0.0;
  }
  A[0] = 1.0;
  int la = 1, st = ALZ_OK;
  for (int m = 1; m <= order && st == ALZ_OK; ++m) {
    for (int t = 0; t <= m; ++t) B[t] = (m - t < la) ? A[m - t] : 0.0;
    double num = 0.0;
    for (int i = 0; i < la && i < m; ++i) num = num + (ac[m - i] * A[i]) * 1.0;
    double den = 0.0;
    for (int i = 0; i <= m; ++i)
      for (int j = 0; j <= m; ++j) den = den + (ac[i > j ? i - j : j - i] * B[i]) * B[j];
    if (den == 0.0) {
      st = ALZ_E_PARCOR;
      break;
    }
    const double k = num / den;
    for (int i = 0; i <= m; ++i) A[i] = A[i] - k * B[i];
    la = m + 1;
    while (la > 1 && A[la - 1] == 0.0) --la;
  }
  double e = 0.0;
  for (int i = 0; i < la; ++i)
    for (int j = 0; j < la; ++j) e = e + (ac[i > j ? i - j : j - i] * A[i]) * A[j];
  for (int i = 0; i <= order; ++i) coefs[f * (order + 1) + i] = (i < la) ? A[i] : 0.0;
  err[f] = e;
  status[f] = st;
}

int launch_levinson_dense(const double *r, int64_t n_frames, int n_lags, int order, double *coefs, double *err,
                                 int *status, hipStream_t st) {
  if (order > 63) return fail(ALZ_E_UNSUPPORTED, "levinson: order > 63 is outside the engine's gate");
  if (n_frames == 0) return ALZ_OK;
  const dim3 grid((unsigned)((n_frames + 63) / 64)), block(64);
  switch (order) {
#define ALZ_LD(P_) case P_ - 1: hipLaunchKernelGGL(k_levinson_dense<P_>, grid, block, 0, st, r, n_frames, n_lags, coefs, err, status); break;
    ALZ_LD(2) ALZ_LD(3) ALZ_LD(4) ALZ_LD(5) ALZ_LD(7) ALZ_LD(9) ALZ_LD(11) ALZ_LD(13) ALZ_LD(17)
#undef ALZ_LD
    default:
      hipLaunchKernelGGL(k_levinson_dense_any, grid, block, 0, st, r, n_frames, n_lags, order, coefs, err, status);
  }
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

}  // namespace alz
