// alz_lev.hip -- levinson_durbin in the reference's own dense form (lazy_lpc.py:115-136), bit-identical.
// (Its own translation unit: the fully unrolled O(order^3) kernels take a while to compile.)
#include "alz_lev.h"

namespace alz {

// k_levinson_dense<P>: levinson_durbin exactly as the reference computes it (lazy_lpc.py:115-136), one lane per
// frame, everything in registers (P = order + 1 is a template parameter, all loops unrolled): see alz_lev.h.
template <int P>
__global__ __launch_bounds__(64) void k_levinson_dense(const double *__restrict__ r_in, int64_t n_frames, int n_lags,
                                                        double *__restrict__ coefs, double *__restrict__ err,
                                                        int *__restrict__ status) {
  const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (f >= n_frames) return;
  double ac[P], A[P], e;
  int st;
#pragma unroll
  for (int i = 0; i < P; ++i) ac[i] = (i < n_lags) ? r_in[f * n_lags + i] : 0.0;      // zero-extended lags (lazy_lpc.py:117-118)
  levinson_dense_regs<P>(ac, A, e, st);
#pragma unroll
  for (int i = 0; i < P; ++i) coefs[f * P + i] = A[i];
  err[f] = e;
  status[f] = st;
}

// any order: the same arithmetic with run-time loops.  BIG = false: order <= 63, arrays in scratch memory;
// BIG = true: arrays in a workspace of 3 * (order + 1) doubles per frame (lpc(blk, order >= 100) takes the
// Levinson-Durbin route in the reference, lazy_lpc.py:176-180: rare, one frame at a time, never fast).
template <bool BIG>
__global__ __launch_bounds__(64) void k_levinson_dense_any(const double *__restrict__ r_in, int64_t n_frames, int n_lags,
                                                            int order, double *__restrict__ coefs,
                                                            double *__restrict__ err, int *__restrict__ status,
                                                            double *__restrict__ ws) {
  const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (f >= n_frames) return;
  double loc[BIG ? 1 : 3 * 64];
  double *ac = BIG ? ws + f * 3 * (int64_t)(order + 1) : loc;
  double *A = ac + (BIG ? order + 1 : 64), *B = A + (BIG ? order + 1 : 64);
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
  if (n_frames == 0) return ALZ_OK;
  const dim3 grid((unsigned)((n_frames + 63) / 64)), block(64);
  switch (order) {
#define ALZ_LD(P_) case P_ - 1: hipLaunchKernelGGL(k_levinson_dense<P_>, grid, block, 0, st, r, n_frames, n_lags, coefs, err, status); break;
    ALZ_LD(2) ALZ_LD(3) ALZ_LD(4) ALZ_LD(5) ALZ_LD(7) ALZ_LD(9) ALZ_LD(11) ALZ_LD(13) ALZ_LD(17)
#undef ALZ_LD
    default:
      if (order <= 63) {
        hipLaunchKernelGGL(k_levinson_dense_any<false>, grid, block, 0, st, r, n_frames, n_lags, order, coefs, err, status,
                           (double *)nullptr);
      } else {
        double *ws = nullptr;
        if (hipMallocAsync((void **)&ws, (size_t)n_frames * 3 * (order + 1) * sizeof(double), st) != hipSuccess)
          return fail(ALZ_E_NOMEM, "levinson: workspace allocation failed");
        hipLaunchKernelGGL(k_levinson_dense_any<true>, grid, block, 0, st, r, n_frames, n_lags, order, coefs, err, status, ws);
        (void)hipFreeAsync(ws, st);
      }
  }
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

}  // namespace alz
