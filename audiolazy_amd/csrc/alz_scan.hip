// alz_scan.hip -- time-parallel execution of a biquad-class section: chunked state propagation.
//
// The reference's generator (audiolazy/lazy_filters.py:251-257) is one serial chain per channel:
// y[n] needs y[n-1].  On the GPU that makes throughput = channels x step rate, so a
// NARROW bank -- 512 channels, one GPU's share of configs[1] sharded over 8 -- leaves the chip idle:
// 32 workgroups at ~30 cycles per step.  This mode (opt-in: alz_bank_set_time_parallel) cuts the
// time axis of a block into K chunks of L samples and runs them as K x C independent lanes:
//
//   prep    x[jL-1], x[jL-2] -> the input history of chunk j (exact: it is just the block);
//   pass 1  every chunk from a ZERO output state, no stores: its end state z_j = (y[L-1], y[L-2])
//           (the streaming kernels of alz_wave.hip, store-less instantiation, 8 B/sample read);
//   fix     per channel, serially over the K chunks: S_0 = the bank's state, S_{j+1} = M S_j + z_j
//           with M = A^L, A = [[-a1, -a2], [1, 0]] -- the recurrence is linear, so the true state at
//           a chunk boundary is the zero-state end state plus the propagated initial state;
//   pass 2  every chunk again from its true initial state S_j, with stores: the ordinary kernels,
//           8 B/sample read + 8 B/sample written.
//
// M is not formed by matrix powers: its columns are the end states of the homogeneous recurrence
// (zero input) started from (1, 0) and (0, 1), run for L steps with the kernels' own arithmetic and
// cached per (section, L) on the bank handle.
//
// NOT bit-identical to the reference: inside a chunk every sample is the same DF-I statement, but
// S_j carries the rounding of a different summation order.  Contract 1e-6 normalised; measured
// <= 1e-12 on the configs[1] bank and ~1e-9 on resonator.z_exp(50 Hz, 1 Hz) (tests/test_gpu_scan.py).
// Algorithmic bytes stay 16 per channel-sample; HBM traffic is 24 (the block is read twice).
#include "alz_common.h"

namespace alz {

struct ScanArgs {
  const double *x;
  int64_t sxn, sxc;
  int64_t C, n_inputs, n_sets;
  int mode, map_input;
  int nb, na;
  int64_t L, K;
  const double *a;
  double *xh, *yh;      // the bank's state [taps-1][C]
  double *vxh, *vyh;    // per-chunk state [taps-1][K*C]
  double *power;        // [4][C]: M11 M12 M21 M22
};

// input history of every chunk; zero output state for pass 1
__global__ __launch_bounds__(256) void k_scan_prep(ScanArgs p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t V = p.K * p.C;
  if (i >= V) return;
  const int64_t j = i / p.C, c = i - j * p.C;
  const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? c % p.n_inputs : c;
  for (int k = 0; k < p.nb - 1; ++k) {
    const int64_t t = j * p.L - 1 - k;
    p.vxh[(int64_t)k * V + i] = t >= 0 ? p.x[t * p.sxn + in * p.sxc] : p.xh[(-t - 1) * p.C + c];
  }
  for (int k = 0; k < p.na - 1; ++k) p.vyh[(int64_t)k * V + i] = 0.0;
}

// columns of M = A^L: homogeneous recurrence from (y[-1], y[-2]) = (1, 0) and (0, 1)
__global__ __launch_bounds__(64) void k_scan_power(ScanArgs p) {
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.C) return;
  const int64_t set = p.mode == ALZ_BANK_OUTER ? c / p.n_inputs : (p.n_sets == 1 ? 0 : c);
  const double na1 = -p.a[1 * p.n_sets + set];
  const double na2 = p.na > 2 ? -p.a[2 * p.n_sets + set] : 0.0;
  double u1 = 1.0, u2 = 0.0, v1 = 0.0, v2 = 1.0;
  for (int64_t n = 0; n < p.L; ++n) {
    const double yu = na1 * u1 + na2 * u2;
    const double yv = na1 * v1 + na2 * v2;
    u2 = u1; u1 = yu;
    v2 = v1; v1 = yv;
  }
  p.power[0 * p.C + c] = u1;   // M11: y[L-1] from (1, 0)
  p.power[1 * p.C + c] = v1;   // M12: y[L-1] from (0, 1)
  p.power[2 * p.C + c] = u2;   // M21: y[L-2] from (1, 0)
  p.power[3 * p.C + c] = v2;   // M22
}

// S_{j+1} = M S_j + z_j per channel; vyh holds z_j on entry and S_j (chunk j's true initial state) on exit
__global__ __launch_bounds__(64) void k_scan_fix(ScanArgs p) {
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.C) return;
  const int64_t V = p.K * p.C;
  const bool two = p.na > 2;
  const double m11 = p.power[0 * p.C + c], m12 = p.power[1 * p.C + c];
  const double m21 = p.power[2 * p.C + c], m22 = p.power[3 * p.C + c];
  double s1 = p.yh[0 * p.C + c], s2 = two ? p.yh[1 * p.C + c] : 0.0;
  constexpr int B = 8;                      // chunk states fetched ahead of the dependent chain
  for (int64_t j0 = 0; j0 < p.K; j0 += B) {
    double z1[B], z2[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int64_t j = j0 + u < p.K ? j0 + u : p.K - 1;
      z1[u] = p.vyh[0 * V + j * p.C + c];
      z2[u] = two ? p.vyh[1 * V + j * p.C + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {
      if (j0 + u < p.K) {
        const int64_t j = j0 + u;
        p.vyh[0 * V + j * p.C + c] = s1;
        if (two) p.vyh[1 * V + j * p.C + c] = s2;
        const double n1 = __builtin_fma(m11, s1, __builtin_fma(m12, s2, z1[u]));
        const double n2 = __builtin_fma(m21, s1, __builtin_fma(m22, s2, z2[u]));
        s1 = n1;
        s2 = n2;
      }
    }
  }
}

// the last chunk's end state is the bank's state after the block
__global__ __launch_bounds__(256) void k_scan_finish(ScanArgs p) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= p.C) return;
  const int64_t V = p.K * p.C, last = (p.K - 1) * p.C + c;
  for (int k = 0; k < p.nb - 1; ++k) p.xh[(int64_t)k * p.C + c] = p.vxh[(int64_t)k * V + last];
  for (int k = 0; k < p.na - 1; ++k) p.yh[(int64_t)k * p.C + c] = p.vyh[(int64_t)k * V + last];
}

static int grow_scratch(double **ptr, uint64_t *have, uint64_t need) {
  if (*have >= need) return ALZ_OK;
  if (*ptr) (void)hipFree(*ptr);
  *ptr = nullptr;
  *have = 0;
  if (hipMalloc((void **)ptr, need) != hipSuccess) return fail(ALZ_E_NOMEM, "hipMalloc failed (time-parallel scratch)");
  *have = need;
  return ALZ_OK;
}

int launch_scan(const SectionDev &sec, int section_index, const BlockIO &io, hipStream_t stream,
                int64_t chunk_len, ScanScratch *scratch, int64_t *done_samples, const char **kernel_name) {
  *done_samples = 0;
  if (!(sec.nb <= 3 && sec.na <= 3 && sec.uniform) || sec.any_div) return ALZ_OK;
  if ((sec.present_b | sec.present_a) == 0) return ALZ_OK;
  if (io.c_first != 0 || io.c_count != io.channels) return ALZ_OK;
  const int64_t C = io.channels;
  if (C % 16) return ALZ_OK;
  // chunk length: a multiple of the longest tile (64 samples); by default short enough that
  // chunks x channels fill the chip (>= 65536 lanes: one 64-lane wave per SIMD)
  int64_t L = chunk_len;
  if (L <= 0) {
    const int64_t k_target = (65536 + C - 1) / C;
    L = io.n / (k_target > 0 ? k_target : 1);
  }
  L = L / 64 * 64;
  if (L < 256) L = 256;
  const int64_t K = io.n / L;
  if (K < 2) return ALZ_OK;
  const int64_t V = K * C;

  const uint64_t vbytes = (uint64_t)2 * V * sizeof(double);
  uint64_t have_x = scratch->v_bytes, have_y = scratch->v_bytes;
  int rc = grow_scratch(&scratch->vxh, &have_x, vbytes);
  if (rc) return rc;
  rc = grow_scratch(&scratch->vyh, &have_y, vbytes);
  if (rc) return rc;
  scratch->v_bytes = have_x < have_y ? have_x : have_y;
  const bool feedback = sec.na > 1;

  ScanArgs p;
  p.x = io.x; p.sxn = io.sxn; p.sxc = io.sxc;
  p.C = C; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets; p.mode = io.mode; p.map_input = io.map_input;
  p.nb = sec.nb; p.na = sec.na; p.L = L; p.K = K; p.a = sec.a;
  p.xh = sec.xh; p.yh = sec.yh; p.vxh = scratch->vxh; p.vyh = scratch->vyh;
  if (feedback) {
    uint64_t have_p = scratch->power_bytes;
    rc = grow_scratch(&scratch->power, &have_p, (uint64_t)4 * C * sizeof(double));
    if (rc) return rc;
    if (have_p != scratch->power_bytes) scratch->power_len = 0;
    scratch->power_bytes = have_p;
  }
  p.power = scratch->power;

  WaveChunks ch;
  ch.n_chunks = K; ch.chunk_len = L; ch.vxh = scratch->vxh; ch.vyh = scratch->vyh;
  bool taken = false;
  const char *inner = "";
  hipLaunchKernelGGL(k_scan_prep, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, stream, p);
  if (feedback) {
    ch.nostore = true;
    rc = launch_wave_chunks(sec, io, stream, ch, &taken, &inner);
    if (rc) return rc;
    if (!taken) return ALZ_OK;            // (prep only touched scratch)
    if (scratch->power_len != L || scratch->power_section != section_index) {
      hipLaunchKernelGGL(k_scan_power, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, stream, p);
      scratch->power_len = L;
      scratch->power_section = section_index;
    }
    hipLaunchKernelGGL(k_scan_fix, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, stream, p);
  }
  ch.nostore = false;
  rc = launch_wave_chunks(sec, io, stream, ch, &taken, &inner);
  if (rc) return rc;
  if (!taken) {
    if (feedback) return fail(ALZ_E_HIP, "time-parallel replay launch refused after the zero-state pass");
    return ALZ_OK;
  }
  hipLaunchKernelGGL(k_scan_finish, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *done_samples = K * L;
  *kernel_name = inner[2] == 'd' ? "k_scan(k_duo<16>)" : inner[7] == '6' ? "k_scan(k_wave<64>)" : "k_scan(k_wave<16>)";
  return ALZ_OK;
}

}  // namespace alz
