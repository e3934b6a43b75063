// alz_io.hip -- the data formats either side of the filter path, and the ordered mixdown.
//
//  * k_mix        ParallelFilter.__call__'s sum ((y0 + y1) + y2) ... over the outputs of the
//                 filters fed with the same input (reference audiolazy/lazy_filters.py:1048-1054),
//                 taken over the coefficient sets of an OUTER bank's output block.
//  * k_mix_tracks Streamix (reference audiolazy/lazy_stream.py:633-724): tracks entering at their
//                 own start samples, summed in the order they were added.
//  * k_pcm_decode WavStream's sample conversion (reference audiolazy/lazy_wav.py:58-130):
//                 little-endian 8/16/24/32-bit PCM -> float64, v / 2**(bits-1) (8-bit data is
//                 unsigned, v - 128), or the stored integer itself with keep.
//  * k_pcm_encode chunks (reference audiolazy/lazy_io.py:44-128): float64 samples packed in one
//                 of struct's homogeneous formats b/B/h/H/i/I/f/d, either byte order.
//
// All three are pure streaming kernels (HBM-bound, no reuse): lane = consecutive samples, 16 or
// 32 bytes of output per lane, no LDS.  The divisions are by powers of two and the sum order is
// the reference's, so the results are bit-identical to CPython's.
#include "alz_common.h"

namespace alz {

// ------------------------------------------------------------------------------- mixdown
struct MixArgs {
  const double *y;
  double *out;
  int64_t n_sets, n_inputs, n;
  int64_t inner;       // extent of the fastest (contiguous) axis of out
  int64_t total;       // n * n_inputs
  int64_t y_outer, y_set, o_outer;   // strides in doubles
};

// One output element per lane.  Time-major: inner = input, outer = time, a set is n_inputs
// further along the row.  Channel-major: inner = time, outer = input, a set is n_inputs rows down.
__global__ __launch_bounds__(256) void k_mix(MixArgs p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.total) return;
  const int64_t outer = i / p.inner, in = i - outer * p.inner;
  const double *src = p.y + outer * p.y_outer + in;
  double acc = src[0];
  int64_t s = 1;
  for (; s + 8 <= p.n_sets; s += 8) {   // the loads of 8 sets in flight, the sum still in order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(s + u) * p.y_set];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = acc + v[u];
  }
  for (; s < p.n_sets; ++s) acc = acc + src[s * p.y_set];
  p.out[outer * p.o_outer + in] = acc;
}

// Streamix (reference lazy_stream.py:633-724): tracks that start at different samples, summed
// in the order they were added:  data = zero; for snd in playing: data += next(snd).
static constexpr int kMixTracks = 24;   // tracks per launch (more: further launches continue the sum)

struct TrackArgs {
  const double *track[kMixTracks];
  int64_t start[kMixTracks], length[kMixTracks];
  int n_tracks, first;      // first launch: the sum starts from `zero`, later ones from out[n]
  double zero;
  double *out;
  int64_t n_out;
};

__global__ __launch_bounds__(256) void k_mix_tracks(TrackArgs p) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= p.n_out) return;
  double acc = p.first ? p.zero : p.out[n];
#pragma unroll 4
  for (int k = 0; k < p.n_tracks; ++k) {
    const int64_t i = n - p.start[k];
    if (i >= 0 && i < p.length[k]) acc = acc + p.track[k][i];
  }
  p.out[n] = acc;
}

// ---------------------------------------------------------------------------- PCM decode
// 4 samples per lane: 4 / 8 / 12 / 16 input bytes -> 32 output bytes.
template <int BITS>
__device__ __forceinline__ double pcm_value(const unsigned char *q, int keep) {
  int v;
  if (BITS == 8) {
    v = keep ? (int)q[0] : (int)q[0] - 128;             // the only unsigned width
  } else if (BITS == 16) {
    v = (int)(short)((unsigned)q[0] | ((unsigned)q[1] << 8));
  } else if (BITS == 24) {
    v = (int)(((unsigned)q[0] << 8) | ((unsigned)q[1] << 16) | ((unsigned)q[2] << 24)) >> 8;
  } else {
    v = (int)((unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24));
  }
  const double d = (double)v;
  return keep ? d : d * (1.0 / (double)(1u << (BITS - 1)));   // exact: a power of two
}

template <int BITS>
__global__ __launch_bounds__(256) void k_pcm_decode(const unsigned char *raw, double *out, int64_t n,
                                                     int keep) {
  constexpr int B = BITS / 8;
  const int64_t q0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (q0 >= n) return;
  if (q0 + 4 <= n) {
    // the lane's 4*B bytes as aligned dwords (raw is 4-byte aligned, 4*B is a multiple of 4)
    unsigned w[B];
    const unsigned *rw = (const unsigned *)(raw + q0 * B);
#pragma unroll
    for (int j = 0; j < B; ++j) w[j] = rw[j];
    unsigned char bytes[4 * B];
#pragma unroll
    for (int j = 0; j < 4 * B; ++j) bytes[j] = (unsigned char)(w[j >> 2] >> (8 * (j & 3)));
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = pcm_value<BITS>(bytes + u * B, keep);
    double2 *dst = (double2 *)(out + q0);
    dst[0] = make_double2(v[0], v[1]);
    dst[1] = make_double2(v[2], v[3]);
  } else {
    for (int64_t q = q0; q < n; ++q) out[q] = pcm_value<BITS>(raw + q * B, keep);
  }
}

// ---------------------------------------------------------------------------- PCM encode
enum { kBadNotInteger = 1, kBadRange = 2, kBadFloatOverflow = 4 };

struct EncArgs {
  const double *in;
  unsigned char *out;
  int64_t n;
  int width;          // bytes per item
  int kind;           // 0 signed int, 1 unsigned int, 2 float32, 3 float64
  int big_endian;
  int *bad;
};

__device__ __forceinline__ unsigned long long enc_bits(const EncArgs &p, double x, int &bad) {
  if (p.kind == 3) return (unsigned long long)__double_as_longlong(x);
  if (p.kind == 2) {
    const float f = (float)x;   // round to nearest even, like struct's 'f'
    if (isinf(f) && !isinf(x)) bad |= kBadFloatOverflow;   // OverflowError in struct.pack
    return (unsigned long long)__float_as_uint(f);
  }
  if (!(x == trunc(x))) { bad |= kBadNotInteger; return 0; }   // also catches NaN
  const int bits = 8 * p.width;
  const double lo = p.kind == 0 ? -ldexp(1.0, bits - 1) : 0.0;
  const double hi = p.kind == 0 ? ldexp(1.0, bits - 1) - 1.0 : ldexp(1.0, bits) - 1.0;
  if (x < lo || x > hi) { bad |= kBadRange; return 0; }
  return (unsigned long long)(long long)x;   // two's complement, truncated to the width below
}

__global__ __launch_bounds__(256) void k_pcm_encode(EncArgs p) {
  const int64_t q0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (q0 >= p.n) return;
  int bad = 0;
  const int cnt = (p.n - q0 < 4) ? (int)(p.n - q0) : 4;
  unsigned char *dst = p.out + q0 * p.width;
  for (int u = 0; u < cnt; ++u) {
    const unsigned long long bits = enc_bits(p, p.in[q0 + u], bad);
    for (int j = 0; j < p.width; ++j) {
      const int sh = p.big_endian ? 8 * (p.width - 1 - j) : 8 * j;
      dst[u * p.width + j] = (unsigned char)(bits >> sh);
    }
  }
  if (bad) atomicOr(p.bad, bad);
}

// fast little-endian paths: 4 items per lane written as one aligned vector
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_pcm_encode_le(const double *in, T *out, int64_t n, int *bad_out) {
  const int64_t q0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (q0 + 4 > n) return;   // the tail goes through k_pcm_encode
  const double2 a = ((const double2 *)(in + q0))[0], b = ((const double2 *)(in + q0))[1];
  const double x[4] = {a.x, a.y, b.x, b.y};
  int bad = 0;
  T v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (KIND == 2) {
      const float f = (float)x[u];
      if (isinf(f) && !isinf(x[u])) bad |= kBadFloatOverflow;
      v[u] = (T)f;
    } else {
      constexpr int bits = 8 * (int)sizeof(T);
      const double lo = -(double)(1ll << (bits - 1)), hi = (double)((1ll << (bits - 1)) - 1);
      if (!(x[u] == trunc(x[u]))) bad |= kBadNotInteger;
      else if (x[u] < lo || x[u] > hi) bad |= kBadRange;
      v[u] = bad ? (T)0 : (T)(long long)x[u];
    }
  }
  struct alignas(4 * sizeof(T)) Pack { T v[4]; } pk = {{v[0], v[1], v[2], v[3]}};
  *(Pack *)(out + q0) = pk;
  if (bad) atomicOr(bad_out, bad);
}

struct DeviceScope {
  int prev = 0, dev = 0;
  bool ok = true;
  explicit DeviceScope(int device) : dev(device) {
    if (hipGetDevice(&prev) != hipSuccess) ok = false;
    if (ok && prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceScope() { if (ok && prev != dev) (void)hipSetDevice(prev); }
};

}  // namespace alz

extern "C" {

int alz_mix_dev(const double *y_dev, int64_t n_sets, int64_t n_inputs, int64_t n, int layout, int64_t ldy,
                int64_t ldo, double *out_dev, int device, void *stream) {
  if (!y_dev || !out_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (n_sets < 1 || n_inputs < 1 || n < 0) return alz::fail(ALZ_E_ARG, "n_sets, n_inputs >= 1 and n >= 0 required");
  if (layout != ALZ_TIME_MAJOR && layout != ALZ_CHAN_MAJOR) return alz::fail(ALZ_E_ARG, "unknown layout");
  if (n == 0) return ALZ_OK;
  alz::MixArgs p;
  p.y = y_dev; p.out = out_dev; p.n_sets = n_sets; p.n_inputs = n_inputs; p.n = n;
  p.total = n * n_inputs;
  if (layout == ALZ_TIME_MAJOR) {
    if (ldy < n_sets * n_inputs || ldo < n_inputs) return alz::fail(ALZ_E_ARG, "row pitch smaller than the row");
    p.inner = n_inputs; p.y_outer = ldy; p.y_set = n_inputs; p.o_outer = ldo;
  } else {
    if (ldy < n || ldo < n) return alz::fail(ALZ_E_ARG, "row pitch smaller than the row");
    p.inner = n; p.y_outer = ldy; p.y_set = n_inputs * ldy; p.o_outer = ldo;
  }
  alz::DeviceScope scope(device);
  if (!scope.ok) return alz::fail(ALZ_E_HIP, "hipSetDevice failed");
  const unsigned grid = (unsigned)((p.total + 255) / 256);
  hipLaunchKernelGGL(alz::k_mix, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

int alz_mix_tracks_dev(int n_tracks, const double *const *tracks_dev, const int64_t *starts, const int64_t *lengths,
                       double zero, int64_t n_out, double *out_dev, int device, void *stream) {
  if (n_tracks < 0 || n_out < 0) return alz::fail(ALZ_E_ARG, "negative count");
  if (!out_dev || (n_tracks > 0 && (!tracks_dev || !starts || !lengths))) return alz::fail(ALZ_E_ARG, "NULL argument");
  for (int k = 0; k < n_tracks; ++k)
    if (!tracks_dev[k] || starts[k] < 0 || lengths[k] < 0) return alz::fail(ALZ_E_ARG, "bad track");
  if (n_out == 0) return ALZ_OK;
  alz::DeviceScope scope(device);
  if (!scope.ok) return alz::fail(ALZ_E_HIP, "hipSetDevice failed");
  const unsigned grid = (unsigned)((n_out + 255) / 256);
  int done = 0;
  do {   // at least one launch: with no track at all the output is `zero` everywhere
    alz::TrackArgs p;
    p.n_tracks = (n_tracks - done < alz::kMixTracks) ? n_tracks - done : alz::kMixTracks;
    for (int k = 0; k < alz::kMixTracks; ++k) {
      const bool on = k < p.n_tracks;
      p.track[k] = on ? tracks_dev[done + k] : nullptr;
      p.start[k] = on ? starts[done + k] : 0;
      p.length[k] = on ? lengths[done + k] : 0;
    }
    p.first = done == 0; p.zero = zero; p.out = out_dev; p.n_out = n_out;
    hipLaunchKernelGGL(alz::k_mix_tracks, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    done += p.n_tracks;
  } while (done < n_tracks);
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

int alz_pcm_decode_dev(const void *raw_dev, int bits, int keep, int64_t n_samples, double *out_dev, int device,
                       void *stream) {
  if (!raw_dev || !out_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (n_samples < 0) return alz::fail(ALZ_E_ARG, "negative sample count");
  if (bits != 8 && bits != 16 && bits != 24 && bits != 32)
    return alz::fail(ALZ_E_UNSUPPORTED, "bits per sample must be 8, 16, 24 or 32");
  if (((uintptr_t)raw_dev & 3) || ((uintptr_t)out_dev & 15))
    return alz::fail(ALZ_E_ARG, "raw must be 4-byte aligned and out 16-byte aligned");
  if (n_samples == 0) return ALZ_OK;
  alz::DeviceScope scope(device);
  if (!scope.ok) return alz::fail(ALZ_E_HIP, "hipSetDevice failed");
  const unsigned grid = (unsigned)(((n_samples + 3) / 4 + 255) / 256);
  const unsigned char *raw = (const unsigned char *)raw_dev;
  hipStream_t st = (hipStream_t)stream;
  switch (bits) {
    case 8: hipLaunchKernelGGL(alz::k_pcm_decode<8>, dim3(grid), dim3(256), 0, st, raw, out_dev, n_samples, keep); break;
    case 16: hipLaunchKernelGGL(alz::k_pcm_decode<16>, dim3(grid), dim3(256), 0, st, raw, out_dev, n_samples, keep); break;
    case 24: hipLaunchKernelGGL(alz::k_pcm_decode<24>, dim3(grid), dim3(256), 0, st, raw, out_dev, n_samples, keep); break;
    default: hipLaunchKernelGGL(alz::k_pcm_decode<32>, dim3(grid), dim3(256), 0, st, raw, out_dev, n_samples, keep); break;
  }
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

int alz_pcm_encode_dev(const double *in_dev, int64_t n, int dfmt, int big_endian, void *out_dev, int *flags_dev,
                       int device, void *stream) {
  if (!in_dev || !out_dev || !flags_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (n < 0) return alz::fail(ALZ_E_ARG, "negative item count");
  alz::EncArgs p;
  p.in = in_dev; p.out = (unsigned char *)out_dev; p.n = n; p.big_endian = big_endian ? 1 : 0; p.bad = flags_dev;
  switch (dfmt) {
    case 'b': p.width = 1; p.kind = 0; break;
    case 'B': p.width = 1; p.kind = 1; break;
    case 'h': p.width = 2; p.kind = 0; break;
    case 'H': p.width = 2; p.kind = 1; break;
    case 'i': case 'l': p.width = 4; p.kind = 0; break;
    case 'I': case 'L': p.width = 4; p.kind = 1; break;
    case 'f': p.width = 4; p.kind = 2; break;
    case 'd': p.width = 8; p.kind = 3; break;
    default: return alz::fail(ALZ_E_UNSUPPORTED, "dfmt must be one of b B h H i I l L f d");
  }
  if (n == 0) return ALZ_OK;
  alz::DeviceScope scope(device);
  if (!scope.ok) return alz::fail(ALZ_E_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(((n + 3) / 4 + 255) / 256);
  const bool aligned = !((uintptr_t)in_dev & 15) && !((uintptr_t)out_dev & 15);
  const bool fast = aligned && !p.big_endian && (dfmt == 'h' || dfmt == 'i' || dfmt == 'l' || dfmt == 'f');
  if (dfmt == 'd' && !p.big_endian) {   // the items as they are
    ALZ_HIP_CHECK(hipMemcpyAsync(out_dev, in_dev, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    return ALZ_OK;
  }
  if (fast) {
    if (dfmt == 'h')
      hipLaunchKernelGGL((alz::k_pcm_encode_le<short, 0>), dim3(grid), dim3(256), 0, st, in_dev, (short *)out_dev, n, flags_dev);
    else if (dfmt == 'f')
      hipLaunchKernelGGL((alz::k_pcm_encode_le<float, 2>), dim3(grid), dim3(256), 0, st, in_dev, (float *)out_dev, n, flags_dev);
    else
      hipLaunchKernelGGL((alz::k_pcm_encode_le<int, 0>), dim3(grid), dim3(256), 0, st, in_dev, (int *)out_dev, n, flags_dev);
    const int64_t done = n & ~(int64_t)3;
    if (done < n) {   // ragged tail
      alz::EncArgs t = p;
      t.in = in_dev + done; t.out = p.out + done * p.width; t.n = n - done;
      hipLaunchKernelGGL(alz::k_pcm_encode, dim3(1), dim3(256), 0, st, t);
    }
  } else {
    hipLaunchKernelGGL(alz::k_pcm_encode, dim3(grid), dim3(256), 0, st, p);
  }
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

}  // extern "C"
