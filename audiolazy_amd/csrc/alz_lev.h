// alz_lev.h -- levinson_durbin in the reference's own dense form (lazy_lpc.py:115-136) on P = order + 1 lags
// held in registers; shared by k_levinson_dense<P> (alz_lev.hip: lags from memory) and
// k_acorr_stage<P, 2> (alz_lpc.hip: lags straight from the lane's accumulators, lpc.kautocor in ONE launch).
#pragma once
#include "alz_common.h"

namespace alz {

// The reference (lazy_lpc.py:115-136):
//   A = 1;  for m = 1..order:  B = A(1/z) z^-m;  A -= inner(A, z^-m) / inner(B, B) * B;   error = inner(A, A)
// with  inner(a, b) = sum(acdata[|i-j|] * a_i * b_j for i.. for j..)  evaluated densely, i outer, j inner.
// Coefficients and error come out bit-identical to the reference's.  Terms whose factor is an exact zero are
// skipped where that cannot change the sum (x + (+-0) == x for a running sum that starts at +0 and can
// therefore never be -0): B_0 = A_m is always zero when step m starts (the dense list of A has at most m
// entries), so row 0 / column 0 of inner(B, B) are left out; `la` mirrors the reference's dense numlist
// length, which shrinks when the top coefficient cancels to exactly zero.
template <int P>
__device__ __forceinline__ void levinson_dense_regs(const double (&ac)[P], double (&A)[P], double &e, int &st) {
  constexpr int order = P - 1;
  double B[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    A[i] = 0.0;
    B[i] = 0.0;
  }
  A[0] = 1.0;
  int la = 1;
  st = ALZ_OK;
#pragma unroll
  for (int m = 1; m <= order; ++m) {
    // B = A(1/z) * z**-m: B[m - i] = A[i] for the la dense coefficients of A, zero elsewhere (B[0] = 0)
#pragma unroll
    for (int t = 1; t <= m; ++t) B[t] = (m - t < la) ? A[m - t] : 0.0;
    // inner(A, z**-m): of the (i, j) terms only j = m has a non-zero b_j (= 1)
    double num = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) num = (i < la) ? num + (ac[m - i] * A[i]) * 1.0 : num;
    double den = 0.0;
#pragma unroll
    for (int i = 1; i <= m; ++i) {
#pragma unroll
      for (int j = 1; j <= m; ++j) den = den + (ac[i > j ? i - j : j - i] * B[i]) * B[j];
    }
    bool ok = st == ALZ_OK;
    if (ok && den == 0.0) {                                   // ZeroDivisionError -> ParCorError (:132-133)
      st = ALZ_E_PARCOR;
      ok = false;
    }
    const double k = ok ? num / den : 0.0;
#pragma unroll
    for (int i = 0; i <= m; ++i) A[i] = ok ? A[i] - k * B[i] : A[i];   // (B[0] == 0: k * 0 kept, a NaN k must still poison A[0])
    if (ok) {
      la = m + 1;
#pragma unroll
      for (int t = m; t >= 1; --t)
        if (la == t + 1 && A[t] == 0.0) la = t;               // Poly drops exact-zero terms: the dense list shrinks
    }
  }
  e = 0.0;
#pragma unroll
  for (int i = 0; i < P; ++i) {
#pragma unroll
    for (int j = 0; j < P; ++j) e = (i < la && j < la) ? e + (ac[i > j ? i - j : j - i] * A[i]) * A[j] : e;
  }
#pragma unroll
  for (int i = 0; i < P; ++i) A[i] = (i < la) ? A[i] : 0.0;
}

}  // namespace alz
