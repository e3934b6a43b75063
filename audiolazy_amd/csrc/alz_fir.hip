// alz_fir.hip -- long tapped-delay-line FIR, time-parallel, bit-exact.
//
// Replaces LinearFilter.__call__ (reference audiolazy/lazy_filters.py:141-264) for the
// feedback-free case  y[n] = (b0*x[n] + b1*x[n-1] + ... + b_{nb-1}*x[n-nb+1]) / a0  with
// many taps (BASELINE config 3: 256 taps x 8192 channels).  No recurrence, so time is
// parallel too: a wave owns 64 adjacent channels (lane = channel, 512-byte coalesced rows of
// the time-major block) and a run of TB output rows; it keeps R = 16 output accumulators per
// lane in VGPRs and walks the taps in ascending order in blocks of K = 8, so every output sees
// exactly the reference's left-to-right sum  ((b0*d0 + b1*d1) + b2*d2) + ...  with separately
// rounded multiply and add (file built with -ffp-contract=off).  acc starts at -0.0, which is
// the additive identity for every double, and zero taps are skipped (they are absent from the
// reference's expression, lazy_filters.py:209).
//
// Taps shared by the whole bank (n_sets == 1) are wave-uniform and live in SGPRs (scalar loads);
// per-channel taps are one coalesced vector load per tap.  The x rows a tap block needs
// (R + K - 1 of them) come from L1/L2: neighbouring time blocks of the same channel group re-read
// each other's rows, HBM sees each row about once.
//
// Bound: FP64 issue, not HBM: 2*nb - 1 f64 ops per output sample (511 for 256 taps) at one op
// per ~4.5 cycles per SIMD.
#include "alz_common.h"

namespace alz {

static constexpr int kFirR = 32;   // outputs per lane held in registers
static constexpr int kFirK = 16;   // taps per block
static constexpr int kFirTB = 256; // output rows per wave

struct FArgs {
  const double *x;
  double *y;
  int64_t n, sxn, syn;     // time-major: channel stride is 1
  int64_t channels, n_inputs, n_sets;
  int64_t c_first, c_end;
  int mode, map_input;
  int nb;
  const double *b, *a;     // b[k * n_sets + set], a[set] (a0 only)
  const double *xh;        // xh[k * channels + c] = x[-1-k]
  int div;                 // some a0 != 1
  double zero;             // what an all-zero tap set yields (lazy_filters.py:227-231)
};

template <bool SHARED>
__global__ __launch_bounds__(64) void k_fir(FArgs p) {
  const int lane = threadIdx.x;
  int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + lane;
  const bool live = c < p.c_end;
  if (!live) c = p.c_end - 1;  // clamp: keep the wave's loads in bounds, mask its stores
  int64_t in, set;
  if (p.mode == ALZ_BANK_OUTER) {
    in = p.map_input ? c % p.n_inputs : c;
    set = c / p.n_inputs;
  } else {
    in = c;
    set = (p.n_sets == 1) ? 0 : c;
  }
  const double a0 = p.a[set];
  const int64_t tb0 = (int64_t)blockIdx.y * kFirTB;
  // shared taps are staged once in LDS, zero-padded to a multiple of the tap block: every later
  // tap read is one broadcast ds_read with no bounds test (a zero tap is skipped anyway)
  extern __shared__ __attribute__((aligned(16))) double tap_lds[];
  if constexpr (SHARED) {
    const int padded = ((p.nb + kFirK - 1) / kFirK) * kFirK;
    for (int k = lane; k < padded; k += 64) tap_lds[k] = (k < p.nb) ? p.b[k] : 0.0;
    __syncthreads();
  }
  bool all_zero = false;
  if constexpr (!SHARED) {   // a channel whose taps are all zero yields `zero`
    int nz = 0;
    for (int k = 0; k < p.nb; ++k) nz += p.b[(int64_t)k * p.n_sets + set] != 0.0;
    all_zero = nz == 0;
  }

  for (int sub = 0; sub < kFirTB; sub += kFirR) {
    const int64_t t0 = tb0 + sub;
    if (t0 >= p.n) break;
    double acc[kFirR];
#pragma unroll
    for (int r = 0; r < kFirR; ++r) acc[r] = -0.0;

    // x window: xw[j] = row t0 - kb - (K-1) + j, j = 0 .. R+K-2.  Going to the next tap block
    // moves the window K rows into the past: R-1 rows are kept (register moves), K are loaded.
    double xw[kFirR + kFirK - 1];
    auto load_row = [&](int64_t t) -> double {
      if (t > p.n - 1) t = p.n - 1;                         // past the block: value is never used
      int64_t hk = -t - 1;                                  // before the stream: history row
      if (hk > p.nb - 2) hk = p.nb - 2;                     // beyond the delay line: never used
      const double *src = (t >= 0) ? p.x + t * p.sxn + in : p.xh + (hk < 0 ? 0 : hk) * p.channels + c;
      return *src;
    };
#pragma unroll
    for (int j = kFirK; j < kFirR + kFirK - 1; ++j) xw[j] = load_row(t0 - (kFirK - 1) + j);
    for (int kb = 0; kb < p.nb; kb += kFirK) {
      if (kb > 0) {
#pragma unroll
        for (int j = kFirR + kFirK - 2; j >= kFirK; --j) xw[j] = xw[j - kFirK];
      }
      const int64_t tb = t0 - kb - (kFirK - 1);
      if (tb >= 0) {                                        // whole row group inside the block
        const double *r0 = p.x + tb * p.sxn + in;
#pragma unroll
        for (int j = 0; j < kFirK; ++j) xw[j] = r0[j * p.sxn];
      } else {                                              // reaches into the history rows
#pragma unroll
        for (int j = 0; j < kFirK; ++j) xw[j] = load_row(tb + j);
      }
      if constexpr (SHARED) {
        double bk[kFirK];                                   // wave-uniform taps: broadcast LDS reads
#pragma unroll
        for (int kk = 0; kk < kFirK; ++kk) bk[kk] = tap_lds[kb + kk];
#pragma unroll
        for (int kk = 0; kk < kFirK; ++kk) {
          if (bk[kk] == 0.0) continue;                      // absent from the reference's sum
#pragma unroll
          for (int r = 0; r < kFirR; ++r) acc[r] = acc[r] + bk[kk] * xw[r - kk + (kFirK - 1)];
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < kFirK; ++kk) {
          const int k = kb + kk;
          if (k >= p.nb) break;
          const double bk = p.b[(int64_t)k * p.n_sets + set];
          const bool nz = bk != 0.0;
#pragma unroll
          for (int r = 0; r < kFirR; ++r) {
            const double s = acc[r] + bk * xw[r - kk + (kFirK - 1)];
            acc[r] = nz ? s : acc[r];
          }
        }
      }
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < kFirR; ++r) {
        const int64_t t = t0 + r;
        if (t < p.n) p.y[t * p.syn + c] = all_zero ? p.zero : (p.div ? acc[r] / a0 : acc[r]);
      }
    }
  }
}

// new input history after the block: xh_new[k] = x[n-1-k], or the old history when the
// block was shorter than the delay line
__global__ void k_fir_state(FArgs p, double *xh_new) {
  const int64_t c = p.c_first + (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.c_end) return;
  const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? c % p.n_inputs : c;
  for (int k = (int)blockIdx.y; k < p.nb - 1; k += (int)gridDim.y) {
    const int64_t t = p.n - 1 - k;
    xh_new[(int64_t)k * p.channels + c] = (t >= 0) ? p.x[t * p.sxn + in] : p.xh[(-t - 1) * p.channels + c];
  }
}

__global__ void k_copy_doubles(double *dst, const double *src, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = src[i];
}

// Feedback-free section with more taps than the register kernels take, time-major block,
// x and y distinct.  Returns ALZ_OK with *taken = false when the shape is not this kernel's.
int launch_fir(const SectionDev &sec, const BlockIO &io, hipStream_t stream, bool *taken,
               const char **kernel_name) {
  *taken = false;
  if (sec.na != 1 || sec.nb < 2) return ALZ_OK;
  if (!(io.sxc == 1 && io.syc == 1)) return ALZ_OK;        // time-major only
  if (io.x == io.y) return ALZ_OK;
  if ((sec.present_b) == 0) return ALZ_OK;                 // all-zero filter: k_generic yields `zero`
  FArgs p;
  p.x = io.x; p.y = io.y; p.n = io.n; p.sxn = io.sxn; p.syn = io.syn;
  p.channels = io.channels; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.c_first = io.c_first; p.c_end = io.c_first + io.c_count;
  p.mode = io.mode; p.map_input = io.map_input;
  p.nb = sec.nb; p.b = sec.b; p.a = sec.a; p.xh = sec.xh; p.div = sec.any_div ? 1 : 0;
  p.zero = io.zero;
  const unsigned gx = (unsigned)((io.c_count + 63) / 64);
  const unsigned gy = (unsigned)((io.n + kFirTB - 1) / kFirTB);
  if (gy > 65535u) return ALZ_OK;  // block longer than the grid's y range: caller falls back
  const size_t tap_bytes = (size_t)((sec.nb + kFirK - 1) / kFirK) * kFirK * sizeof(double);
  if (sec.shared_sets && tap_bytes > 48 * 1024) return ALZ_OK;   // absurdly long: let k_generic have it
  if (sec.shared_sets)
    hipLaunchKernelGGL(k_fir<true>, dim3(gx, gy), dim3(64), tap_bytes, stream, p);
  else
    hipLaunchKernelGGL(k_fir<false>, dim3(gx, gy), dim3(64), 0, stream, p);
  // histories: into the spare half of the state slab, then over the live half
  const int64_t nx = (int64_t)(sec.nb - 1) * io.channels;
  double *xh_new = sec.xh + nx;
  hipLaunchKernelGGL(k_fir_state, dim3(gx, 8), dim3(64), 0, stream, p, xh_new);
  // only the channels of this launch were refreshed in xh_new: copy just those back
  if (io.c_first == 0 && io.c_count == io.channels) {
    hipLaunchKernelGGL(k_copy_doubles, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream,
                       sec.xh, xh_new, nx);
  } else {
    for (int k = 0; k < sec.nb - 1; ++k)
      hipLaunchKernelGGL(k_copy_doubles, dim3((unsigned)((io.c_count + 255) / 256)), dim3(256), 0,
                         stream, sec.xh + (int64_t)k * io.channels + io.c_first,
                         xh_new + (int64_t)k * io.channels + io.c_first, io.c_count);
  }
  ALZ_HIP_CHECK(hipGetLastError());
  *taken = true;
  *kernel_name = sec.shared_sets ? "k_fir<shared>" : "k_fir<per-channel>";
  return ALZ_OK;
}

}  // namespace alz
