/* How often does libm's pow(x, 2.0) -- what CPython computes for `x ** 2` (floatobject.c float_pow -> pow()) -- differ
 * from the correctly rounded product x * x?  Sweeps N pseudo-random doubles: uniform(-1, 1) audio samples, and a second
 * pass with random exponents over the whole normal range.  Build: gcc -O2 -fno-builtin-pow tools/pow2_sweep.c -lm
 * (-fno-builtin-pow: the compiler must not fold pow(x, 2.0) into x * x itself).  Reference lines:
 * audiolazy/lazy_analysis.py:440-465, 496-520 (envelope.rms / envelope.squared: `Stream(sig) ** 2`). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t next(void) {           /* xorshift128+ */
  uint64_t a = s[0], b = s[1];
  s[0] = b; a ^= a << 23; a ^= a >> 17; a ^= b ^ (b >> 26); s[1] = a;
  return a + b;
}

int main(int argc, char **argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
  uint64_t bad_audio = 0, bad_wide = 0;
  double worst = 0.0, ex = 0.0;
  for (uint64_t i = 0; i < n; ++i) {
    const double x = (double)(int64_t)next() * (1.0 / 9223372036854775808.0);     /* uniform(-1, 1), 63 random bits */
    const double p = pow(x, 2.0), q = x * x;
    if (memcmp(&p, &q, 8)) { ++bad_audio; const double r = fabs(p - q) / q; if (r > worst) { worst = r; ex = x; } }
  }
  for (uint64_t i = 0; i < n / 4; ++i) {
    uint64_t bits = next();
    const uint64_t e = 1023 - 400 + (bits >> 52) % 800;                          /* exponents -400 .. +399: x * x stays normal */
    bits = (bits & 0x800FFFFFFFFFFFFFull) | (e << 52);
    double x; memcpy(&x, &bits, 8);
    const double p = pow(x, 2.0), q = x * x;
    if (memcmp(&p, &q, 8)) ++bad_wide;
  }
  printf("pow(x, 2.0) != x * x: %llu of %llu uniform(-1, 1) samples (%.3g; worst relative difference %.3g at x = %a), "
         "%llu of %llu wide-range samples\n", (unsigned long long)bad_audio, (unsigned long long)n, (double)bad_audio / (double)n,
         worst, ex, (unsigned long long)bad_wide, (unsigned long long)(n / 4));
  return 0;
}
