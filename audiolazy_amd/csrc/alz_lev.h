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
// Round 6: the recursion step in two forms.  `lev_step_generic` is the per-lane form (every lane carries its own dense
// length `la` and status: selects on every term).  In all but pathological frames la == m when step m starts and no
// denominator is zero, for EVERY lane of the wave -- `lev_step_regular` assumes that, runs the same operations in the same
// order without a single select (the per-lane form spent a quarter of its vector instructions on v_cndmask / v_cmp and ran
// its multiplies inside ~120 EXEC-masked regions), and checks the assumption with two wave-wide votes; a wave in which any
// lane deviates (den == 0: ParCorError; a top coefficient that cancels to exactly zero) repeats that step -- nothing has
// been written yet -- and continues in the per-lane form.  Same doubles either way.
template <int P, int M>
__device__ __forceinline__ void lev_step_generic(const double (&ac)[P], double (&A)[P], double (&B)[P], int &la, int &st) {
  constexpr int m = M;
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

// ALZ_LEV_PIPE 1 (A/B builds only; measured SLOWER, profiles/NOTES_r06.md 8): the regular form's sums as three-stage
// pipelines with the instruction order pinned -- term t + 2's first product, term t + 1's second product and term t's
// addition issued in turn, so that no instruction uses a result younger than three instructions (hipcc emits every term of
// inner(B, B) as  v_mul -> v_mul -> v_add  through one temporary).  Same products and additions in the same order.  On
// configs[4]'s 65 536 frames (one wave per SIMD) it took 78.4 us against 75.1 for the plain loops, with MORE wait cycles
// (SQ_WAIT_ANY +19 %) and 227 VGPRs instead of 162; at 2^20 frames the two are equal.
#ifndef ALZ_LEV_PIPE
#define ALZ_LEV_PIPE 0
#endif
#define ALZ_LEV_PIN() __builtin_amdgcn_sched_barrier(0)
// s + sum over i = LO .. HI, j = LO .. HI (i outer) of (ac[|i - j|] * V[i]) * V[j], added in that order
template <int P, int LO, int HI>
__device__ __forceinline__ double lev_quad(const double (&ac)[P], const double (&V)[P], double s) {
#if ALZ_LEV_PIPE
  constexpr int n = HI - LO + 1, N = n * n;
  double q[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int t = -2; t < N; ++t) {
    if (t + 2 < N) {
      const int i = LO + (t + 2) / n, j = LO + (t + 2) % n;
      q[(t + 2) % 3] = ac[i > j ? i - j : j - i] * V[i];
      ALZ_LEV_PIN();
    }
    if (t + 1 >= 0 && t + 1 < N) {
      const int j = LO + (t + 1) % n;
      q[(t + 1) % 3] = q[(t + 1) % 3] * V[j];
      ALZ_LEV_PIN();
    }
    if (t >= 0) {
      s = s + q[t % 3];
      ALZ_LEV_PIN();
    }
  }
#else
#pragma unroll
  for (int i = LO; i <= HI; ++i) {
#pragma unroll
    for (int j = LO; j <= HI; ++j) s = s + (ac[i > j ? i - j : j - i] * V[i]) * V[j];
  }
#endif
  return s;
}

// Entry: la == M and st == ALZ_OK on every active lane.  Returns whether that still holds for step M + 1.
template <int P, int M>
__device__ __forceinline__ bool lev_step_regular(const double (&ac)[P], double (&A)[P], double (&B)[P], int &la, int &st) {
  constexpr int m = M;
#pragma unroll
  for (int t = 1; t <= m; ++t) B[t] = A[m - t];
  double num = 0.0;
#if ALZ_LEV_PIPE
  {
    double pr[2] = {0.0, 0.0};
#pragma unroll
    for (int t = -1; t < m; ++t) {
      if (t + 1 < m) {
        pr[(t + 1) % 2] = (ac[m - (t + 1)] * A[t + 1]) * 1.0;
        ALZ_LEV_PIN();
      }
      if (t >= 0) {
        num = num + pr[t % 2];
        ALZ_LEV_PIN();
      }
    }
  }
#else
#pragma unroll
  for (int i = 0; i < m; ++i) num = num + (ac[m - i] * A[i]) * 1.0;
#endif
  const double den = lev_quad<P, 1, M>(ac, B, 0.0);
  if (__builtin_amdgcn_ballot_w64(den == 0.0) != 0) {       // some lane: ParCorError -- the step again, lane by lane
    la = m;
    st = ALZ_OK;
    lev_step_generic<P, M>(ac, A, B, la, st);
    return false;
  }
  const double k = num / den;
#if ALZ_LEV_PIPE
  // A[i] -= k B[i]: the products four at a time in front of their subtractions
#pragma unroll
  for (int i0 = 0; i0 <= m; i0 += 4) {
    double kb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u <= m) kb[u] = k * B[i0 + u];
    ALZ_LEV_PIN();
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u <= m) A[i0 + u] = A[i0 + u] - kb[u];
    ALZ_LEV_PIN();
  }
#else
#pragma unroll
  for (int i = 0; i <= m; ++i) A[i] = A[i] - k * B[i];
#endif
  if (__builtin_amdgcn_ballot_w64(A[m] == 0.0) == 0) return true;
  la = m + 1;                                               // some lane's top coefficient cancelled: lengths per lane from here on
  st = ALZ_OK;
#pragma unroll
  for (int t = m; t >= 1; --t)
    if (la == t + 1 && A[t] == 0.0) la = t;
  return false;
}

template <int P, int M>
__device__ __forceinline__ void lev_steps(const double (&ac)[P], double (&A)[P], double (&B)[P], int &la, int &st, bool &regular) {
  if constexpr (M <= P - 1) {
    if (regular) regular = lev_step_regular<P, M>(ac, A, B, la, st);
    else lev_step_generic<P, M>(ac, A, B, la, st);
    lev_steps<P, M + 1>(ac, A, B, la, st, regular);
  }
}

template <int P>
__device__ __forceinline__ void levinson_dense_regs(const double (&ac)[P], double (&A)[P], double &e, int &st) {
  double B[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    A[i] = 0.0;
    B[i] = 0.0;
  }
  A[0] = 1.0;
  int la = 1;
  st = ALZ_OK;
  bool regular = true;                                       // wave-uniform
  lev_steps<P, 1>(ac, A, B, la, st, regular);
  e = 0.0;
  if (regular) {                                             // la == P on every lane
    e = lev_quad<P, 0, P - 1>(ac, A, 0.0);
  } else {
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
      for (int j = 0; j < P; ++j) e = (i < la && j < la) ? e + (ac[i > j ? i - j : j - i] * A[i]) * A[j] : e;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) A[i] = (i < la) ? A[i] : 0.0;
  }
}

#undef ALZ_LEV_PIN

}  // namespace alz
