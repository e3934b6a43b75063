// alz_map.hip -- elementwise stages either side of the filter, on the device.
//
// The reference writes them as lazy per-sample Stream expressions around a filter call:
// ``abs(Stream(sig))`` / ``Stream(sig) ** 2`` / ``... ** .5`` in envelope (audiolazy/lazy_analysis.py
// :440-520), ``clip`` (:619-647), and the operator table every Stream carries (audiolazy/lazy_stream.py
// :47-71: + - * / with a number or another Stream, unary - and abs).  Each op below is the same IEEE
// binary64 operation CPython performs on the same operands, so the result is bit-identical -- with
// one stated exception: ALZ_MAP_SQUARE is x * x (correctly rounded), whereas the reference's
// ``x ** 2`` is libm's pow(x, 2.0), which differs from it in the last bit for about one sample in a
// thousand; it is offered as an opt-in, never substituted silently.
//
// Pure streaming kernels: no reuse, no LDS; a lane moves 16-byte pieces, four in flight.
// Algorithmic bytes: 8 read per operand + 8 written per sample.
#include "alz_common.h"

namespace alz {

typedef double dbl2 __attribute__((ext_vector_type(2)));

struct MArgs {
  const double *x, *y;
  double *out;
  int64_t n;
  double p0, p1;
  int *flags;
};

template <int OP>
__device__ __forceinline__ double map_one(double x, double y, double p0, double p1, int &flag) {
  if constexpr (OP == ALZ_MAP_ABS) return __builtin_fabs(x);
  if constexpr (OP == ALZ_MAP_NEG) return -x;
  if constexpr (OP == ALZ_MAP_SQRT) {
    // CPython's ``v ** .5`` is libm pow(v, .5): (-0.0) ** .5 == +0.0 and (-inf) ** .5 == +inf (C99 pow), where
    // sqrt would give -0.0 and NaN; a negative finite item is complex in Python 3 -> flagged
    if (x == -__builtin_inf()) return __builtin_inf();
    if (x < 0.0) flag |= ALZ_MAP_DOMAIN;
    return __builtin_sqrt(x + 0.0);               // correctly rounded (OCML), == pow(v, .5) of libm on [0, inf]
  }
  if constexpr (OP == ALZ_MAP_SQUARE) return x * x;
  if constexpr (OP == ALZ_MAP_MUL) return x * p0;
  if constexpr (OP == ALZ_MAP_ADD) return x + p0;
  if constexpr (OP == ALZ_MAP_SUB) return x - p0;
  if constexpr (OP == ALZ_MAP_RSUB) return p0 - x;
  if constexpr (OP == ALZ_MAP_DIV) return x / p0;
  if constexpr (OP == ALZ_MAP_RDIV) {
    if (x == 0.0) flag |= ALZ_MAP_ZERODIV;        // float division by zero raises in Python
    return p0 / x;
  }
  // clip's three rules (lazy_analysis.py:638-647): the one-sided forms keep an item only when it is
  // strictly inside (so a NaN becomes the limit), the two-sided form replaces it only when it is
  // strictly outside (a NaN passes)
  if constexpr (OP == ALZ_MAP_CLIP) return x > p1 ? p1 : (x < p0 ? p0 : x);
  if constexpr (OP == ALZ_MAP_CLIP_HIGH) return x < p1 ? x : p1;
  if constexpr (OP == ALZ_MAP_CLIP_LOW) return x > p0 ? x : p0;
  if constexpr (OP == ALZ_MAP_ADD2) return x + y;
  if constexpr (OP == ALZ_MAP_SUB2) return x - y;
  if constexpr (OP == ALZ_MAP_MUL2) return x * y;
  if constexpr (OP == ALZ_MAP_DIV2) {
    if (y == 0.0) flag |= ALZ_MAP_ZERODIV;
    return x / y;
  }
  return x;
}

template <int OP>
constexpr bool map_is_binary() { return OP >= ALZ_MAP_ADD2; }

// pieces of 16 bytes; `wide` says all three pointers are 16-byte aligned
template <int OP, bool WIDE>
__global__ __launch_bounds__(256) void k_map(MArgs p) {
  constexpr int U = 4;
  int flag = 0;
  if constexpr (WIDE) {
    const int64_t pieces = p.n >> 1;
    const dbl2 *x2 = reinterpret_cast<const dbl2 *>(p.x);
    const dbl2 *y2 = reinterpret_cast<const dbl2 *>(p.y);
    dbl2 *o2 = reinterpret_cast<dbl2 *>(p.out);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < pieces; i0 += stride * U) {
      dbl2 a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < pieces) {
          a[u] = x2[i];
          if constexpr (map_is_binary<OP>()) b[u] = y2[i];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < pieces) {
          dbl2 r;
          r.x = map_one<OP>(a[u].x, map_is_binary<OP>() ? b[u].x : 0.0, p.p0, p.p1, flag);
          r.y = map_one<OP>(a[u].y, map_is_binary<OP>() ? b[u].y : 0.0, p.p0, p.p1, flag);
          o2[i] = r;
        }
      }
    }
    if ((p.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const int64_t i = p.n - 1;
      p.out[i] = map_one<OP>(p.x[i], map_is_binary<OP>() ? p.y[i] : 0.0, p.p0, p.p1, flag);
    }
  } else {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += stride)
      p.out[i] = map_one<OP>(p.x[i], map_is_binary<OP>() ? p.y[i] : 0.0, p.p0, p.p1, flag);
  }
  if (flag && p.flags) atomicOr(p.flags, flag);
}

typedef void (*map_fn)(MArgs);

template <bool WIDE>
static map_fn pick_map(int op) {
  switch (op) {
#define ALZ_OP(O) case O: return (map_fn)k_map<O, WIDE>;
    ALZ_OP(ALZ_MAP_ABS) ALZ_OP(ALZ_MAP_NEG) ALZ_OP(ALZ_MAP_SQRT) ALZ_OP(ALZ_MAP_SQUARE)
    ALZ_OP(ALZ_MAP_MUL) ALZ_OP(ALZ_MAP_ADD) ALZ_OP(ALZ_MAP_SUB) ALZ_OP(ALZ_MAP_RSUB)
    ALZ_OP(ALZ_MAP_DIV) ALZ_OP(ALZ_MAP_RDIV)
    ALZ_OP(ALZ_MAP_CLIP) ALZ_OP(ALZ_MAP_CLIP_HIGH) ALZ_OP(ALZ_MAP_CLIP_LOW)
    ALZ_OP(ALZ_MAP_ADD2) ALZ_OP(ALZ_MAP_SUB2) ALZ_OP(ALZ_MAP_MUL2) ALZ_OP(ALZ_MAP_DIV2)
#undef ALZ_OP
    default: return nullptr;
  }
}

int launch_map(int op, const double *x, const double *y, double p0, double p1, int64_t n, double *out,
               int *flags, hipStream_t stream) {
  if (n == 0) return ALZ_OK;
  const bool binary = op >= ALZ_MAP_ADD2;
  const bool wide = ((((uintptr_t)x | (uintptr_t)out) | (binary ? (uintptr_t)y : 0)) & 15) == 0;
  map_fn fn = wide ? pick_map<true>(op) : pick_map<false>(op);
  if (!fn) return fail(ALZ_E_ARG, "alz_map_dev: unknown op");
  MArgs p;
  p.x = x; p.y = y; p.out = out; p.n = n; p.p0 = p0; p.p1 = p1; p.flags = flags;
  const int64_t work = wide ? ((n >> 1) + 4 * 256 - 1) / (4 * 256) : (n + 255) / 256;
  int64_t blocks = work < 1 ? 1 : work;
  if (blocks > 8192) blocks = 8192;                 // grid-stride beyond that: 32 workgroups per CU
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

// OUTER bank with few inputs: x [n, n_inputs] -> xe with one column (time-major) / row (channel-major)
// per channel, channel c holding input c % n_inputs, so that the streaming kernels can read it like a
// diagonal bank's input (a gammatone bank on ONE stream is 256 channels that all read the same samples)
__global__ __launch_bounds__(256) void k_expand(const double *x, double *xe, int64_t n, int64_t channels, int64_t n_inputs,
                                                int64_t sxn, int64_t sxc, int64_t sen, int64_t sec, int time_major) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * channels) return;
  const int64_t t = time_major ? i / channels : i % n, c = time_major ? i % channels : i / n;   // the fast index is contiguous in xe
  xe[t * sen + c * sec] = x[t * sxn + (c % n_inputs) * sxc];
}

int launch_expand(const double *x, double *xe, int64_t n, int64_t channels, int64_t n_inputs, int64_t sxn, int64_t sxc,
                  int64_t sen, int64_t sec, hipStream_t stream) {
  const int64_t total = n * channels;
  if (total == 0) return ALZ_OK;
  hipLaunchKernelGGL(k_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, xe, n, channels, n_inputs,
                     sxn, sxc, sen, sec, sec == 1 ? 1 : 0);
  ALZ_HIP_CHECK(hipGetLastError());
  return ALZ_OK;
}

}  // namespace alz

extern "C" int alz_map_dev(int op, const double *x_dev, const double *y_dev, double p0, double p1, int64_t n,
                           double *out_dev, int *flags_dev, int device, void *stream) {
  if (!x_dev || !out_dev) return alz::fail(ALZ_E_ARG, "NULL argument");
  if (n < 0) return alz::fail(ALZ_E_ARG, "negative length");
  if (op >= ALZ_MAP_ADD2 && !y_dev) return alz::fail(ALZ_E_ARG, "binary op needs a second block");
  if (op == ALZ_MAP_CLIP && p1 < p0)
    return alz::fail(ALZ_E_ARG, "Higher clipping limit is smaller than lower one");   // lazy_analysis.py:643-644
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  const int rc = alz::launch_map(op, x_dev, y_dev, p0, p1, n, out_dev, flags_dev, (hipStream_t)stream);
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}
