// alz_lpc.hip -- batched lpc.kautocor: autocorrelation + Levinson-Durbin per frame.
//
// Replaces, for a batch of frames,
//   acorr(blk, order)            reference audiolazy/lazy_analysis.py:277-312
//   levinson_durbin(acdata, ord) reference audiolazy/lazy_lpc.py:52-136
//   lpc.kautocor(blk, order)     reference audiolazy/lazy_lpc.py:229-272
//
// Mapping: a frame owns a SLOT of 32 (order <= 31) or 64 lanes of a wavefront;
// lane i of the slot owns lag i / coefficient i.  The frame is staged once in
// LDS (coalesced HBM read), then every lag lane walks it left to right --
// the same summation order as the reference's sum(...), so acorr is bit-exact
// (file built with -ffp-contract=off).  Levinson-Durbin then runs across the
// lanes of the slot with cross-lane shuffles: the standard O(order^2)
// recursion, mathematically the reference's update
//   A -= inner(A, z**-m) / inner(B, B) * B      (lazy_lpc.py:128-131)
// with inner(B, B) carried as the running prediction error.
#include "alz_common.h"

namespace alz {

template <int SLOT>
__device__ __forceinline__ double slot_sum(double v) {
#pragma unroll
  for (int off = SLOT / 2; off > 0; off >>= 1) v = v + __shfl_xor(v, off, SLOT);
  return v;
}

template <int SLOT>
__global__ __launch_bounds__(64) void k_lpc(const double *__restrict__ sig, int64_t n_frames,
                                            int frame_len, int64_t hop, int order,
                                            double *__restrict__ coefs, double *__restrict__ err,
                                            int *__restrict__ status, double *__restrict__ r_out,
                                            int from_r) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *smem = reinterpret_cast<double *>(smem_raw);
  constexpr int FPW = 64 / SLOT;  // frames per wave
  const int lane = threadIdx.x;
  const int slot = lane / SLOT, i = lane % SLOT;
  const int64_t f0 = (int64_t)blockIdx.x * FPW;

  // stage the wave's frames in LDS, coalesced (from_r: "frames" are ready-made lag lists
  // r[0..order] and the autocorrelation pass is skipped -- levinson_durbin alone)
  for (int s = 0; s < FPW; ++s) {
    const int64_t f = f0 + s;
    if (f >= n_frames) break;
    const double *src = sig + f * hop;
    for (int n = lane; n < frame_len; n += 64) smem[s * frame_len + n] = src[n];
  }
  __syncthreads();

  const int64_t f = f0 + slot;
  const bool live = f < n_frames;
  const double *fr = smem + slot * frame_len;

  // acorr: lane i sums lag i, left to right (lazy_analysis.py:311-312)
  double r = 0.0;
  if (from_r) {
    if (live && i <= order && i < frame_len) r = fr[i];   // short lag lists are zero-extended (lazy_lpc.py:117-118)
  } else if (live && i <= order) {
    const int cnt = frame_len - i;
    for (int n = 0; n < cnt; ++n) r = r + fr[n] * fr[n + i];
  }
  if (r_out) {
    if (live && i <= order) r_out[f * (order + 1) + i] = r;
    if (!coefs) return;
  }

  // Levinson-Durbin across the slot's lanes
  double a = (i == 0) ? 1.0 : 0.0;
  double E = __shfl(r, 0, SLOT);
  int st = ALZ_OK;
  for (int m = 1; m <= order; ++m) {
    const int src = (m - i) & (SLOT - 1);
    const double rr = __shfl(r, src, SLOT);   // r[m - i]
    const double ar = __shfl(a, src, SLOT);   // a[m - i]
    const double num = slot_sum<SLOT>((i < m) ? a * rr : 0.0);
    if (E == 0.0) st = ALZ_E_PARCOR;          // inner(B, B) == 0, lazy_lpc.py:132-133
    const double k = (st == ALZ_OK) ? -(num / E) : 0.0;
    if (i >= 1 && i < m) a = a + k * ar;
    if (i == m) a = k;
    E = E * (1.0 - k * k);
  }
  if (live) {
    if (i <= order) coefs[f * (order + 1) + i] = a;
    if (i == 0) {
      err[f] = E;
      status[f] = st;
    }
  }
}

static int launch_lpc(const double *sig, int64_t n_frames, int frame_len, int64_t hop, int order,
                      double *coefs, double *err, int *status, double *r_out, int from_r,
                      hipStream_t st) {
  if (n_frames < 0 || frame_len < 1 || hop < 0 || order < 0)
    return fail(ALZ_E_ARG, "lpc: bad frame geometry");
  if (order > 63) return fail(ALZ_E_UNSUPPORTED, "lpc: order > 63 is outside the engine's gate");
  if (n_frames == 0) return ALZ_OK;
  const int slot = order <= 31 ? 32 : 64;
  const int fpw = 64 / slot;
  const size_t lds = (size_t)fpw * frame_len * sizeof(double);
  if (lds > 160 * 1024) return fail(ALZ_E_UNSUPPORTED, "lpc: frame does not fit the 160 KiB LDS");
  const dim3 grid((unsigned)((n_frames + fpw - 1) / fpw)), block(64);
  if (slot == 32) {
    if (lds > 64 * 1024)
      ALZ_HIP_CHECK(hipFuncSetAttribute((const void *)k_lpc<32>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lpc<32>, grid, block, lds, st, sig, n_frames, frame_len, hop, order, coefs,
                       err, status, r_out, from_r);
  } else {
    if (lds > 64 * 1024)
      ALZ_HIP_CHECK(hipFuncSetAttribute((const void *)k_lpc<64>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lpc<64>, grid, block, lds, st, sig, n_frames, frame_len, hop, order, coefs,
                       err, status, r_out, from_r);
  }
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

}  // namespace alz

extern "C" {

int alz_lpc_kautocor_dev(const double *sig_dev, int64_t n_frames, int frame_len, int64_t hop,
                         int order, double *coefs_dev, double *err_dev, int *status_dev, int device,
                         void *stream) {
  if (!sig_dev || !coefs_dev || !err_dev || !status_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  int rc = alz::launch_lpc(sig_dev, n_frames, frame_len, hop, order, coefs_dev, err_dev, status_dev,
                           nullptr, 0, (hipStream_t)stream);
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

int alz_levinson_dev(const double *r_dev, int64_t n_frames, int n_lags, int order,
                     double *coefs_dev, double *err_dev, int *status_dev, int device, void *stream) {
  if (!r_dev || !coefs_dev || !err_dev || !status_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (n_lags < 1) return alz::fail(ALZ_E_ARG, "levinson: need at least lag 0");
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  int rc = alz::launch_lpc(r_dev, n_frames, n_lags, n_lags, order, coefs_dev, err_dev, status_dev,
                           nullptr, 1, (hipStream_t)stream);
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

int alz_acorr_dev(const double *sig_dev, int64_t n_frames, int frame_len, int64_t hop, int max_lag,
                  double *r_dev, int device, void *stream) {
  if (!sig_dev || !r_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  int rc = alz::launch_lpc(sig_dev, n_frames, frame_len, hop, max_lag, nullptr, nullptr, nullptr,
                           r_dev, 0, (hipStream_t)stream);
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

}  // extern "C"
