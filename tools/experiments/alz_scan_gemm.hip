// EXPERIMENT (not part of the shipped library): alz_scan.hip with the zero-state pass of the fused time-parallel cascade as dot
// products with the cascade's impulse responses (k_cgemm_tables / k_cgemm_dot / k_cgemm_reduce, ALZ_CSCAN_GEMM=1 in a -DALZ_TUNING build).
// Built INSTEAD of csrc/alz_scan.hip by tools/build_cscan_gemm.sh; DESIGN.md section 7, cscan_gemm_prototype.py.

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
  // One pass where the shape allows it (time-major block, a recursive section): 512-sample chunks resident in LDS, the
  // block read once (alz_look.hip) -- 260 Gsamples/s with 16 B/sample of traffic at 512 channels x 2^20 against 228
  // with 24 for the three-launch form below (profiles/NOTES_r03.md).  ALZ_TP_ONE_PASS asks for it; ALZ_TP_AUTO takes
  // it when its workgroups (one per CU: 16 channels x up to 16 chunks in flight) fill most of the chip, i.e. from
  // about 200 channels up; narrower banks fill the chip better as chunks x channels lanes of the three-launch form.
  bool one_pass = chunk_len == ALZ_TP_ONE_PASS;
  if (chunk_len == ALZ_TP_AUTO && sec.na > 1 && io.sxc == 1 && io.syc == 1) {
    int dev = 0, cus = 0;
    ALZ_HIP_CHECK(hipGetDevice(&dev));
    ALZ_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t groups = C / 16, Kl = io.n / kLookChunk;
    int64_t wk = groups > 0 ? cus / groups : 0;
    wk = wk > 16 ? 16 : wk;
    wk = wk > Kl ? Kl : wk;
    one_pass = wk >= 2 && 4 * groups * wk >= 3 * (int64_t)cus;
  }
  if (chunk_len < 0) chunk_len = 0;
  if (one_pass && sec.na > 1 && io.n >= 4 * kLookChunk && io.sxc == 1 && io.syc == 1) {
    const int64_t groups = C / 16, Kl = io.n / kLookChunk;
    // (a wait that ran out in an earlier launch is reported by alz_api.hip's take_look_error at every entry point)
    if (!scratch->look_err) {
      if (hipHostMalloc((void **)&scratch->look_err, 64, hipHostMallocDefault) != hipSuccess)
        return fail(ALZ_E_NOMEM, "hipHostMalloc failed (time-parallel scratch)");
      *scratch->look_err = 0;
    }
    const uint64_t zneed = (uint64_t)groups * Kl * 32 * sizeof(double);
    uint64_t have_z = scratch->zbuf_bytes, have_p = scratch->power_bytes;
    int rc2 = grow_scratch(&scratch->zbuf, &have_z, zneed);
    if (rc2) return rc2;
    scratch->zbuf_bytes = have_z;
    rc2 = grow_scratch(&scratch->power, &have_p, (uint64_t)4 * C * sizeof(double));
    if (rc2) return rc2;
    if (have_p != scratch->power_bytes) scratch->power_len = 0;
    scratch->power_bytes = have_p;
    ScanArgs pw;
    pw.x = io.x; pw.sxn = io.sxn; pw.sxc = io.sxc; pw.C = C; pw.n_inputs = io.n_inputs; pw.n_sets = io.n_sets;
    pw.mode = io.mode; pw.map_input = io.map_input; pw.nb = sec.nb; pw.na = sec.na; pw.L = kLookChunk; pw.K = Kl;
    pw.a = sec.a; pw.xh = sec.xh; pw.yh = sec.yh; pw.vxh = nullptr; pw.vyh = nullptr; pw.power = scratch->power;
    const bool fresh = scratch->power_len != kLookChunk || scratch->power_section != section_index;
    if (fresh) hipLaunchKernelGGL(k_scan_power, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, stream, pw);
    int64_t done = 0;
    rc2 = launch_look(sec, io, stream, scratch->power, scratch->zbuf, zneed, scratch->look_err, &done, kernel_name);
    if (rc2) return rc2;
    if (fresh) {                      // (the matrix is valid whether or not the kernel took the block)
      scratch->power_len = kLookChunk;
      scratch->power_section = section_index;
    }
    if (done > 0) {
      ALZ_HIP_CHECK(hipGetLastError());
      *done_samples = done;
      *kernel_name = "k_scan(k_look)";
      return ALZ_OK;
    }
  }
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

// ---------------------------------------------------------------------------
// Time-parallel execution of a whole fused cascade (a gammatone band: four biquad-class sections).
//
// The reference's own filterbank shape is ONE signal through every band (lazy_auditory.py:158-218,
// examples/gammatone_plots.py:47): 256 bands x 1 stream is 256 serial chains.  Section by section the mode
// above costs four zero-state passes + four replays over an input that was first expanded to a column per
// band (12 x the algorithmic traffic).  Here the cascade stays fused (alz_casc.hip) and the chunks of the
// time axis become its channels: a channel-major block [S, N] read as [S * K, L] IS an OUTER bank on S * K
// input rows, and its output [B, S * K, L] IS the block [B * S, N] -- no expansion, no copies.
//
//   prep    chunk j > 0 starts from zero outputs; section 0's input history is the block itself (exact),
//           chunk 0 starts from the bank's state;
//   pass 1  the fused cascade over all chunks, no stores: end states.  Chunk 0's is the true S_1;
//   fix     per real channel, serially over the chunks: S_{j+1} = M S_j + z_j, S = the two last outputs of every
//           section (the next section's input history is the same two numbers), M (2 nsec x 2 nsec) = the
//           zero-input response over L steps, columns from unit states, cached per chunk length;
//   pass 2  the fused cascade again from the true states, with stores.
//
// Traffic: the input is read twice (S x N doubles, tiny next to the output), the output written once:
// 8 + 16 / B bytes per output sample against 8 + 8 / B algorithmic.
// ---------------------------------------------------------------------------
struct CScanArgs {
  const double *x;
  int64_t ldx;                 // elements between input rows (channel-major)
  int64_t C, n_inputs, n_sets; // C = real channels of the bank
  int mode, map_input, nsec;
  int nb[4], na[4];
  const double *b[4], *a[4];
  double *xh[4], *yh[4];       // the bank's state [taps-1][C]
  double *vxh[4], *vyh[4];     // per-chunk state [taps-1][C * K], slot real * K + chunk
  int64_t L, K;
  double *power;               // M[r][e] at power[(r * 8 + e) * C + c]
#ifdef ALZ_TUNING
  int first_is_z;              // (experimental zero-state pass) slot 0 of vyh holds z_0, not S_1: the fix starts at chunk 0
#endif
};

__global__ __launch_bounds__(256) void k_cscan_prep(CScanArgs p) {
  const int64_t vc = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t V = p.K * p.C;
  if (vc >= V) return;
  const int64_t real = vc / p.K, j = vc - real * p.K;
  const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? real % p.n_inputs : real;
  for (int s = 0; s < p.nsec; ++s) {
    for (int k = 0; k < p.nb[s] - 1; ++k) {
      double v = 0.0;
      if (j == 0) v = p.xh[s][(int64_t)k * p.C + real];
      else if (s == 0) v = p.x[in * p.ldx + j * p.L - 1 - k];
      p.vxh[s][(int64_t)k * V + vc] = v;
    }
    for (int k = 0; k < p.na[s] - 1; ++k) p.vyh[s][(int64_t)k * V + vc] = j == 0 ? p.yh[s][(int64_t)k * p.C + real] : 0.0;
  }
}

// column e of M: the cascade with zero input from the unit state e (state r = 2 s + k: output y_s[-1-k])
__global__ __launch_bounds__(64) void k_cscan_power(CScanArgs p) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int ND = 2 * p.nsec;
  if (i >= p.C * ND) return;
  const int64_t c = i / ND;
  const int e = (int)(i - c * ND);
  const int64_t set = p.mode == ALZ_BANK_OUTER ? c / p.n_inputs : (p.n_sets == 1 ? 0 : c);
  double b0[4], b1[4], b2[4], na1[4], na2[4], y1[4], y2[4];
  for (int s = 0; s < 4; ++s) {
    const bool on = s < p.nsec;
    // section 0 sees a zero input: its numerator does not matter here
    b0[s] = (on && s > 0 && p.nb[s] > 0) ? p.b[s][0 * p.n_sets + set] : 0.0;
    b1[s] = (on && s > 0 && p.nb[s] > 1) ? p.b[s][1 * p.n_sets + set] : 0.0;
    b2[s] = (on && s > 0 && p.nb[s] > 2) ? p.b[s][2 * p.n_sets + set] : 0.0;
    na1[s] = (on && p.na[s] > 1) ? -p.a[s][1 * p.n_sets + set] : 0.0;
    na2[s] = (on && p.na[s] > 2) ? -p.a[s][2 * p.n_sets + set] : 0.0;
    y1[s] = (e == 2 * s) ? 1.0 : 0.0;
    y2[s] = (e == 2 * s + 1) ? 1.0 : 0.0;
  }
  for (int64_t n = 0; n < p.L; ++n) {
    double xin = 0.0, x1 = 0.0, x2 = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double o1 = y1[s], o2 = y2[s];
      const double y = b0[s] * xin + b1[s] * x1 + b2[s] * x2 + na1[s] * o1 + na2[s] * o2;
      y2[s] = o1;
      y1[s] = y;
      xin = y; x1 = o1; x2 = o2;
    }
  }
  for (int s = 0; s < p.nsec; ++s) {
    p.power[((int64_t)(2 * s) * 8 + e) * p.C + c] = y1[s];
    p.power[((int64_t)(2 * s + 1) * 8 + e) * p.C + c] = y2[s];
  }
}

// S_{j+1} = M S_j + z_j per real channel.  On entry vyh holds the end states of pass 1 (slot 0: the true S_1,
// slots j > 0: z_j); on exit every slot holds its chunk's true initial state (slot 0: the bank's).
// Eight lanes per channel, one per state row: lane r keeps row r of M, walks the channel's chunks in order (they are
// contiguous in memory, [k][channel * K + j]) with the z of the next block of eight chunks in flight, and the eight
// lanes of a channel exchange their state components through LDS once per chunk.  (One lane per channel with the
// whole matrix in registers issued 14 scattered stores and 8 scattered loads per chunk from each of 64 lanes: 586 us
// per call at one chunk of look-ahead, 254 us at eight -- more than either cascade pass; profiles/NOTES_r03.md.)
__global__ __launch_bounds__(64) void k_cscan_fix(CScanArgs p) {
  __shared__ double xs[64];
  const int lane = threadIdx.x, r = lane & 7, g = lane >> 3;
  int64_t c = (int64_t)blockIdx.x * 8 + g;
  const bool live_c = c < p.C;
  if (!live_c) c = p.C - 1;                                  // (keeps the wave's LDS exchange uniform; nothing is stored)
  const int64_t V = p.K * p.C, v0 = c * p.K;
  const int ND = 2 * p.nsec;
  constexpr int B = 8;
  const int s = r >> 1, k = r & 1;
  const bool on = r < ND && k < p.na[s < p.nsec ? s : 0] - 1;
  double M[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) M[e] = (r < ND && e < ND) ? p.power[((int64_t)r * 8 + e) * p.C + c] : 0.0;
  const double *zsrc = on ? p.vyh[s] + (int64_t)k * V + v0 : p.power;
  double *ydst = (on && live_c) ? p.vyh[s] + (int64_t)k * V + v0 : nullptr;
  // the next section's input history is this section's output history
  double *xdst = (r < ND && live_c && s + 1 < p.nsec && k < p.nb[s + 1] - 1) ? p.vxh[s + 1] + (int64_t)k * V + v0 : nullptr;
  double S = on ? zsrc[0] : 0.0;                             // end state of chunk 0 = the true S_1 (row r of it)
  int64_t j_first = 1;
#ifdef ALZ_TUNING
  if (p.first_is_z) {                                        // slot 0 holds z_0: the recursion starts from the bank's state
    S = on ? p.yh[s][(int64_t)k * p.C + c] : 0.0;
    j_first = 0;
  }
#endif
  if (ydst && j_first == 1) ydst[0] = p.yh[s][(int64_t)k * p.C + c];   // chunk 0 replays from the bank's state (with z_0 in slot 0 the chain's first step writes it, AFTER z_0 has been fetched)
  auto fetch = [&](int64_t j0, double (&z)[B]) {             // z_r of chunks j0 .. j0 + B - 1 (clamped)
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int64_t j = j0 + u < p.K ? j0 + u : p.K - 1;
      z[u] = on ? zsrc[j] : 0.0;
    }
  };
  auto chain = [&](int64_t j0, const double (&z)[B]) {
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int64_t j = j0 + u;
      if (j < p.K) {                                         // (uniform)
        if (ydst) ydst[j] = S;
        if (xdst) xdst[j] = S;
        xs[lane] = S;
        __builtin_amdgcn_wave_barrier();
        double acc = z[u];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_fma(M[e], xs[g * 8 + e], acc);
        __builtin_amdgcn_wave_barrier();
        S = acc;
      }
    }
  };
  // chunks 1 .. K - 1 in blocks of B, two register sets
  double za[B], zb[B];
  fetch(j_first, za);
  for (int64_t j0 = j_first; j0 < p.K; j0 += 2 * B) {
    fetch(j0 + B, zb);
    chain(j0, za);
    fetch(j0 + 2 * B, za);
    chain(j0 + B, zb);
  }
}

// the last chunk's end state (left by pass 2) is the bank's state after the block
__global__ __launch_bounds__(256) void k_cscan_finish(CScanArgs p) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= p.C) return;
  const int64_t V = p.K * p.C, last = c * p.K + p.K - 1;
  for (int s = 0; s < p.nsec; ++s) {
    for (int k = 0; k < p.nb[s] - 1; ++k) p.xh[s][(int64_t)k * p.C + c] = p.vxh[s][(int64_t)k * V + last];
    for (int k = 0; k < p.na[s] - 1; ++k) p.yh[s][(int64_t)k * p.C + c] = p.vyh[s][(int64_t)k * V + last];
  }
}

#ifdef ALZ_TUNING
// ---------------------------------------------------------------------------
// Experimental, tuning builds only (ALZ_CSCAN_GEMM=1): the zero-state pass as dot products (DESIGN.md section 7,
// tools/experiments/cscan_gemm_prototype.py).  A chunk's zero-state end state z_j[s][k] = y_s[L - 1 - k] is linear in
// the chunk's input: sum_m h_s[L - 1 - k - m] x_j[m] + e1[s][k] x_j[-1] + e2[s][k] x_j[-2], h_s = impulse response of
// sections 0 .. s, e1 / e2 = the responses to a unit sample in section 0's input history.  8 fused multiply-adds per
// sample where the cascade pass issues ~28 instructions.  OUTER banks on ONE input row only (the reference's own
// filterbank shape); the bank's state must be self-consistent (true after reset and after every block).
// ---------------------------------------------------------------------------
struct CGemmTab {
  double *hr = nullptr, *edge = nullptr;   // hr[(c * L + m) * 4 + s] = h_s[L - 1 - m];  edge[((c * 2 + q) * 4 + s) * 2 + k]
  double *part = nullptr;                  // partial sums of the split dot products
  uint64_t hr_bytes = 0, edge_bytes = 0, part_bytes = 0;
  int64_t L = 0, C = 0;
  const void *key = nullptr;
};
static CGemmTab g_cgemm;

__global__ __launch_bounds__(64) void k_cgemm_tables(CScanArgs p, double *__restrict__ hr, double *__restrict__ edge) {
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.C) return;
  const int64_t set = p.mode == ALZ_BANK_OUTER ? c / p.n_inputs : (p.n_sets == 1 ? 0 : c);
  double b0[4], b1[4], b2[4], na1[4], na2[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bool on = s < p.nsec;
    b0[s] = (on && p.nb[s] > 0) ? p.b[s][0 * p.n_sets + set] : 0.0;
    b1[s] = (on && p.nb[s] > 1) ? p.b[s][1 * p.n_sets + set] : 0.0;
    b2[s] = (on && p.nb[s] > 2) ? p.b[s][2 * p.n_sets + set] : 0.0;
    na1[s] = (on && p.na[s] > 1) ? -p.a[s][1 * p.n_sets + set] : 0.0;
    na2[s] = (on && p.na[s] > 2) ? -p.a[s][2 * p.n_sets + set] : 0.0;
  }
  for (int run = 0; run < 3; ++run) {       // 0: impulse at n = 0;  1: x[-1] = 1;  2: x[-2] = 1
    double y1[4] = {0.0, 0.0, 0.0, 0.0}, y2[4] = {0.0, 0.0, 0.0, 0.0};
    double xa = run == 1 ? 1.0 : 0.0, xb = run == 2 ? 1.0 : 0.0;     // section 0's input history x[n-1], x[n-2]
    for (int64_t n = 0; n < p.L; ++n) {
      double xin = (run == 0 && n == 0) ? 1.0 : 0.0, x1 = xa, x2 = xb;
      xb = xa;
      xa = xin;
      double out[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double o1 = y1[s], o2 = y2[s];
        const double y = b0[s] * xin + b1[s] * x1 + b2[s] * x2 + na1[s] * o1 + na2[s] * o2;
        y2[s] = o1;
        y1[s] = y;
        out[s] = y;
        xin = y; x1 = o1; x2 = o2;
      }
      if (run == 0) {
        double *dst = hr + ((int64_t)c * p.L + (p.L - 1 - n)) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[s] = s < p.nsec ? out[s] : 0.0;
      }
    }
    if (run > 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        edge[(((int64_t)c * 2 + (run - 1)) * 4 + s) * 2 + 0] = s < p.nsec ? y1[s] : 0.0;
        edge[(((int64_t)c * 2 + (run - 1)) * 4 + s) * 2 + 1] = s < p.nsec ? y2[s] : 0.0;
      }
    }
  }
}

// lane = chunk j (64 per wave), NS consecutive real channels per wave, the sum over the chunk split SPLIT ways (grid z:
// 2048 waves for 256 bands x 256 chunks, two per SIMD); the responses are wave-uniform (scalar loads, the next pair of
// samples' worth requested before the current one is used), every lane reads its own chunk of the one input row.
// Partial sums go to part[(seg * 8 + 2 s + k) * V + c * K + j]; k_cgemm_reduce adds them in segment order.
template <int NS, int SPLIT>
__global__ __launch_bounds__(64) void k_cgemm_dot(CScanArgs p, const double *__restrict__ hr, const double *__restrict__ edge,
                                                  double *__restrict__ part) {
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t c0 = (int64_t)blockIdx.y * NS;
  const int seg = blockIdx.z;
  const int64_t Ls = p.L / SPLIT, m0 = seg * Ls, m1 = m0 + Ls;
  const double *xrow = p.x + j * p.L;
  double acc[NS][4][2];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[a][s][0] = acc[a][s][1] = 0.0;
  double xprev = seg > 0 ? xrow[m0 - 1] : 0.0;
  dbl2 v = *reinterpret_cast<const dbl2 *>(xrow + m0);
  double h[NS][8];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int q = 0; q < 8; ++q) h[a][q] = hr[((c0 + a) * p.L + m0) * 4 + q];
  for (int64_t m = m0; m < m1; m += 2) {
    const int64_t mn = m + 2 < m1 ? m + 2 : m;                 // (the last step requests its own pair again)
    const dbl2 vn = *reinterpret_cast<const dbl2 *>(xrow + mn);
    double hn[NS][8];
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
      for (int q = 0; q < 8; ++q) hn[a][q] = hr[((c0 + a) * p.L + mn) * 4 + q];
#pragma unroll
    for (int a = 0; a < NS; ++a) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[a][s][0] = __builtin_fma(h[a][s], v.x, acc[a][s][0]);
        acc[a][s][1] = __builtin_fma(h[a][s], xprev, acc[a][s][1]);
        acc[a][s][0] = __builtin_fma(h[a][4 + s], v.y, acc[a][s][0]);
        acc[a][s][1] = __builtin_fma(h[a][4 + s], v.x, acc[a][s][1]);
      }
    }
    xprev = v.y;
    v = vn;
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
      for (int q = 0; q < 8; ++q) h[a][q] = hn[a][q];
  }
  const int64_t V = p.K * p.C;
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    const int64_t c = c0 + a;
    double xm1 = 0.0, xm2 = 0.0;
    if (seg == 0) {                                            // the two samples before the chunk: one segment adds them
      xm1 = j > 0 ? xrow[-1] : (p.nb[0] > 1 ? p.xh[0][0 * p.C + c] : 0.0);
      xm2 = j > 0 ? xrow[-2] : (p.nb[0] > 2 ? p.xh[0][1 * p.C + c] : 0.0);
    }
    const double *e = edge + c * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        double z = acc[a][s][k];
        z = __builtin_fma(e[(0 * 4 + s) * 2 + k], xm1, z);
        z = __builtin_fma(e[(1 * 4 + s) * 2 + k], xm2, z);
        part[((int64_t)(seg * 8 + 2 * s + k)) * V + c * p.K + j] = z;
      }
    }
  }
}

// The direct-load form with the scalar loads of the NEXT pair of samples issued before the current pair's arithmetic
// (CGEMM_ASM=1): hand-placed s_load_dwordx16 + s_waitcnt, because the compiler waits for a scalar load where it issues it.
// NS x 16 SGPRs per buffer, two buffers: NS = 2 fits the SGPR file.
template <int NS, int SPLIT>
__global__ __launch_bounds__(64) void k_cgemm_dot_asm(CScanArgs p, const double *__restrict__ hr, const double *__restrict__ edge,
                                                      double *__restrict__ part) {
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  typedef double dbl8 __attribute__((ext_vector_type(8)));
  static_assert(NS == 2, "two buffers of NS x 16 SGPRs");
  const int lane = threadIdx.x;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t c0 = (int64_t)blockIdx.y * NS;
  const int seg = blockIdx.z;
  const int64_t Ls = p.L / SPLIT, m0 = seg * Ls, m1 = m0 + Ls;
  const double *xrow = p.x + j * p.L;
  double acc[NS][4][2];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[a][s][0] = acc[a][s][1] = 0.0;
  double xprev = seg > 0 ? xrow[m0 - 1] : 0.0;
  const double *ha = hr + ((c0 + 0) * p.L + m0) * 4, *hb = hr + ((c0 + 1) * p.L + m0) * 4;   // (wave-uniform)
  dbl8 ca, cb, na, nb;
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(ca), "=&s"(cb) : "s"(ha), "s"(hb) : "memory");
  dbl2 v = *reinterpret_cast<const dbl2 *>(xrow + m0);
  for (int64_t m = m0; m < m1; m += 2) {
    const int64_t step = m + 2 < m1 ? 8 : 0;                    // (the last step requests its own pair again)
    ha += step; hb += step;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(na), "=&s"(nb) : "s"(ha), "s"(hb) : "memory");
    const dbl2 vn = *reinterpret_cast<const dbl2 *>(xrow + (m + 2 < m1 ? m + 2 : m));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0][s][0] = __builtin_fma(ca[s], v.x, acc[0][s][0]);
      acc[0][s][1] = __builtin_fma(ca[s], xprev, acc[0][s][1]);
      acc[0][s][0] = __builtin_fma(ca[4 + s], v.y, acc[0][s][0]);
      acc[0][s][1] = __builtin_fma(ca[4 + s], v.x, acc[0][s][1]);
      acc[1][s][0] = __builtin_fma(cb[s], v.x, acc[1][s][0]);
      acc[1][s][1] = __builtin_fma(cb[s], xprev, acc[1][s][1]);
      acc[1][s][0] = __builtin_fma(cb[4 + s], v.y, acc[1][s][0]);
      acc[1][s][1] = __builtin_fma(cb[4 + s], v.x, acc[1][s][1]);
    }
    xprev = v.y;
    v = vn;
    // the requested pair has landed; tying the buffers to the wait keeps their copies behind it
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb) : : "memory");
    ca = na;
    cb = nb;
  }
  const int64_t V = p.K * p.C;
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    const int64_t c = c0 + a;
    double xm1 = 0.0, xm2 = 0.0;
    if (seg == 0) {
      xm1 = j > 0 ? xrow[-1] : (p.nb[0] > 1 ? p.xh[0][0 * p.C + c] : 0.0);
      xm2 = j > 0 ? xrow[-2] : (p.nb[0] > 2 ? p.xh[0][1 * p.C + c] : 0.0);
    }
    const double *e = edge + c * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        double z = acc[a][s][k];
        z = __builtin_fma(e[(0 * 4 + s) * 2 + k], xm1, z);
        z = __builtin_fma(e[(1 * 4 + s) * 2 + k], xm2, z);
        part[((int64_t)(seg * 8 + 2 * s + k)) * V + c * p.K + j] = z;
      }
    }
  }
}

// The same sums with the input staged through LDS (CGEMM_LDS=1): a lane reading its own row costs the vector cache one
// line access per lane and instruction (64 cycles per 16-byte load, NOTES_r04.md section 2); here every 16 samples of the
// wave's 64 rows arrive as eight 1 KiB global_load_lds transfers (8 rows x 128 B each: 8 line accesses), XOR-swizzled on
// the global side so that the lanes' 16-byte reads hit distinct banks -- k_acorr_stage's scheme (alz_lpc.hip).
__device__ __forceinline__ void cg_dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int NS, int SPLIT>
__global__ __launch_bounds__(64) void k_cgemm_dot_lds(CScanArgs p, const double *__restrict__ hr, const double *__restrict__ edge,
                                                      double *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char cg_smem[];
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  constexpr int RING = 2;
  const int lane = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * 64, j = j0 + lane;
  const int64_t c0 = (int64_t)blockIdx.y * NS;
  const int seg = blockIdx.z;
  const int64_t Ls = p.L / SPLIT, m0 = seg * Ls;
  const int nsteps = (int)(Ls / 16);
  const double *xrow = p.x + j * p.L;
  const unsigned lds0 = (unsigned)(uintptr_t)cg_smem;
  auto queue = [&](int st) {
    const unsigned slot = lds0 + (unsigned)(st % RING) * 8192u;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = 8 * t + lane / 8;
      const int piece = (lane % 8) ^ (row & 7);
      cg_dma16(p.x + (j0 + row) * p.L + m0 + 16 * st + 2 * piece, slot + t * 1024);
    }
  };
  double acc[NS][4][2];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[a][s][0] = acc[a][s][1] = 0.0;
  double xprev = seg > 0 ? xrow[m0 - 1] : 0.0;
  queue(0);
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) {
      queue(st + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char *slot = cg_smem + (st % RING) * 8192 + lane * 128;
    double cur[16];
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {
      const dbl2 v = *reinterpret_cast<const dbl2 *>(slot + ((pc ^ (lane & 7)) * 16));
      cur[2 * pc] = v.x;
      cur[2 * pc + 1] = v.y;
    }
    const int64_t m = m0 + 16 * st;
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const double *h = hr + ((c0 + a) * p.L + m + u) * 4;       // (wave-uniform)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double h0 = h[s], h1 = h[4 + s];
          acc[a][s][0] = __builtin_fma(h0, cur[u], acc[a][s][0]);
          acc[a][s][1] = __builtin_fma(h0, xprev, acc[a][s][1]);
          acc[a][s][0] = __builtin_fma(h1, cur[u + 1], acc[a][s][0]);
          acc[a][s][1] = __builtin_fma(h1, cur[u], acc[a][s][1]);
        }
      }
      xprev = cur[u + 1];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                            // (the slot's reads have landed before it is refilled)
  }
  const int64_t V = p.K * p.C;
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    const int64_t c = c0 + a;
    double xm1 = 0.0, xm2 = 0.0;
    if (seg == 0) {
      xm1 = j > 0 ? xrow[-1] : (p.nb[0] > 1 ? p.xh[0][0 * p.C + c] : 0.0);
      xm2 = j > 0 ? xrow[-2] : (p.nb[0] > 2 ? p.xh[0][1 * p.C + c] : 0.0);
    }
    const double *e = edge + c * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        double z = acc[a][s][k];
        z = __builtin_fma(e[(0 * 4 + s) * 2 + k], xm1, z);
        z = __builtin_fma(e[(1 * 4 + s) * 2 + k], xm2, z);
        part[((int64_t)(seg * 8 + 2 * s + k)) * V + c * p.K + j] = z;
      }
    }
  }
}

// Both remedies together (CGEMM_LDS=2; run once with the round's last seconds: correct, 315.9 Gsamples/s = the best direct form): rows through LDS as in
// k_cgemm_dot_lds, the responses of the next pair of samples requested by hand before the current pair's arithmetic as in
// k_cgemm_dot_asm.  (The compiler's own lgkmcnt waits for the LDS reads do not know of the scalar loads in flight; they only
// become more conservative by them: a count reached with extra operations outstanding needs more completions, never fewer.)
template <int NS, int SPLIT>
__global__ __launch_bounds__(64) void k_cgemm_dot_lds_asm(CScanArgs p, const double *__restrict__ hr, const double *__restrict__ edge,
                                                          double *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char cg_smem2[];
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  typedef double dbl8 __attribute__((ext_vector_type(8)));
  static_assert(NS == 2, "two buffers of NS x 16 SGPRs");
  constexpr int RING = 2;
  const int lane = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * 64, j = j0 + lane;
  const int64_t c0 = (int64_t)blockIdx.y * NS;
  const int seg = blockIdx.z;
  const int64_t Ls = p.L / SPLIT, m0 = seg * Ls;
  const int nsteps = (int)(Ls / 16);
  const double *xrow = p.x + j * p.L;
  const unsigned lds0 = (unsigned)(uintptr_t)cg_smem2;
  auto queue = [&](int st) {
    const unsigned slot = lds0 + (unsigned)(st % RING) * 8192u;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = 8 * t + lane / 8;
      const int piece = (lane % 8) ^ (row & 7);
      cg_dma16(p.x + (j0 + row) * p.L + m0 + 16 * st + 2 * piece, slot + t * 1024);
    }
  };
  double acc[NS][4][2];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[a][s][0] = acc[a][s][1] = 0.0;
  double xprev = seg > 0 ? xrow[m0 - 1] : 0.0;
  const double *ha = hr + ((c0 + 0) * p.L + m0) * 4, *hb = hr + ((c0 + 1) * p.L + m0) * 4;   // (wave-uniform)
  dbl8 ca, cb, na, nb;
  queue(0);
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(ca), "=&s"(cb) : "s"(ha), "s"(hb) : "memory");
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) {
      queue(st + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char *slot = cg_smem2 + (st % RING) * 8192 + lane * 128;
    double cur[16];
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {
      const dbl2 v = *reinterpret_cast<const dbl2 *>(slot + ((pc ^ (lane & 7)) * 16));
      cur[2 * pc] = v.x;
      cur[2 * pc + 1] = v.y;
    }
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      const int adv = (u == 14 && st + 1 == nsteps) ? 0 : 8;       // (the very last pair requests itself again)
      ha += adv; hb += adv;
      asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(na), "=&s"(nb) : "s"(ha), "s"(hb) : "memory");
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0][s][0] = __builtin_fma(ca[s], cur[u], acc[0][s][0]);
        acc[0][s][1] = __builtin_fma(ca[s], xprev, acc[0][s][1]);
        acc[0][s][0] = __builtin_fma(ca[4 + s], cur[u + 1], acc[0][s][0]);
        acc[0][s][1] = __builtin_fma(ca[4 + s], cur[u], acc[0][s][1]);
        acc[1][s][0] = __builtin_fma(cb[s], cur[u], acc[1][s][0]);
        acc[1][s][1] = __builtin_fma(cb[s], xprev, acc[1][s][1]);
        acc[1][s][0] = __builtin_fma(cb[4 + s], cur[u + 1], acc[1][s][0]);
        acc[1][s][1] = __builtin_fma(cb[4 + s], cur[u], acc[1][s][1]);
      }
      xprev = cur[u + 1];
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb) : : "memory");
      ca = na;
      cb = nb;
    }
  }
  const int64_t V = p.K * p.C;
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    const int64_t c = c0 + a;
    double xm1 = 0.0, xm2 = 0.0;
    if (seg == 0) {
      xm1 = j > 0 ? xrow[-1] : (p.nb[0] > 1 ? p.xh[0][0 * p.C + c] : 0.0);
      xm2 = j > 0 ? xrow[-2] : (p.nb[0] > 2 ? p.xh[0][1 * p.C + c] : 0.0);
    }
    const double *e = edge + c * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        double z = acc[a][s][k];
        z = __builtin_fma(e[(0 * 4 + s) * 2 + k], xm1, z);
        z = __builtin_fma(e[(1 * 4 + s) * 2 + k], xm2, z);
        part[((int64_t)(seg * 8 + 2 * s + k)) * V + c * p.K + j] = z;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_cgemm_reduce(CScanArgs p, const double *__restrict__ part, int split) {
  const int64_t V = p.K * p.C;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 8 * V) return;
  const int r = (int)(idx / V);
  const int64_t vc = idx - (int64_t)r * V;
  const int s = r >> 1, k = r & 1;
  if (s >= p.nsec) return;
  double z = part[(int64_t)r * V + vc];
  for (int seg = 1; seg < split; ++seg) z = z + part[((int64_t)seg * 8 + r) * V + vc];
  p.vyh[s][(int64_t)k * V + vc] = z;
}

static int cgemm_zero_state_pass(const CScanArgs &p, ScanScratch *scratch, bool fresh, hipStream_t stream) {
#ifndef CGEMM_NS
#define CGEMM_NS 4
#endif
#ifndef CGEMM_SPLIT
#define CGEMM_SPLIT 8
#endif
  constexpr int NS = CGEMM_NS, SPLIT = CGEMM_SPLIT;
  if (p.C % NS != 0 || p.L % (2 * SPLIT) != 0) return fail(ALZ_E_UNSUPPORTED, "dot-product zero-state pass: shape");
  const int64_t V = p.K * p.C;
  const uint64_t hr_need = (uint64_t)p.C * p.L * 4 * sizeof(double), edge_need = (uint64_t)p.C * 16 * sizeof(double);
  const uint64_t part_need = (uint64_t)SPLIT * 8 * V * sizeof(double);
  const bool realloc = g_cgemm.hr_bytes < hr_need || g_cgemm.edge_bytes < edge_need;
  int rc = grow_scratch(&g_cgemm.hr, &g_cgemm.hr_bytes, hr_need);
  if (rc) return rc;
  rc = grow_scratch(&g_cgemm.edge, &g_cgemm.edge_bytes, edge_need);
  if (rc) return rc;
  rc = grow_scratch(&g_cgemm.part, &g_cgemm.part_bytes, part_need);
  if (rc) return rc;
  if (fresh || realloc || g_cgemm.key != (const void *)scratch || g_cgemm.L != p.L || g_cgemm.C != p.C) {
    hipLaunchKernelGGL(k_cgemm_tables, dim3((unsigned)((p.C + 63) / 64)), dim3(64), 0, stream, p, g_cgemm.hr, g_cgemm.edge);
    g_cgemm.key = scratch; g_cgemm.L = p.L; g_cgemm.C = p.C;
  }
#if defined(CGEMM_ASM) && CGEMM_ASM
  hipLaunchKernelGGL((k_cgemm_dot_asm<NS, SPLIT>), dim3((unsigned)(p.K / 64), (unsigned)(p.C / NS), SPLIT), dim3(64), 0, stream, p,
                     (const double *)g_cgemm.hr, (const double *)g_cgemm.edge, g_cgemm.part);
#elif defined(CGEMM_LDS) && CGEMM_LDS == 2
  if (p.L % (16 * SPLIT) != 0) return fail(ALZ_E_UNSUPPORTED, "dot-product zero-state pass (LDS form): shape");
  hipLaunchKernelGGL((k_cgemm_dot_lds_asm<NS, SPLIT>), dim3((unsigned)(p.K / 64), (unsigned)(p.C / NS), SPLIT), dim3(64), 2 * 8192, stream, p,
                     (const double *)g_cgemm.hr, (const double *)g_cgemm.edge, g_cgemm.part);
#elif defined(CGEMM_LDS) && CGEMM_LDS
  if (p.L % (16 * SPLIT) != 0) return fail(ALZ_E_UNSUPPORTED, "dot-product zero-state pass (LDS form): shape");
  hipLaunchKernelGGL((k_cgemm_dot_lds<NS, SPLIT>), dim3((unsigned)(p.K / 64), (unsigned)(p.C / NS), SPLIT), dim3(64), 2 * 8192, stream, p,
                     (const double *)g_cgemm.hr, (const double *)g_cgemm.edge, g_cgemm.part);
#else
  hipLaunchKernelGGL((k_cgemm_dot<NS, SPLIT>), dim3((unsigned)(p.K / 64), (unsigned)(p.C / NS), SPLIT), dim3(64), 0, stream, p,
                     (const double *)g_cgemm.hr, (const double *)g_cgemm.edge, g_cgemm.part);
#endif
  hipLaunchKernelGGL(k_cgemm_reduce, dim3((unsigned)((8 * V + 255) / 256)), dim3(256), 0, stream, p, (const double *)g_cgemm.part, SPLIT);
  return hipGetLastError() == hipSuccess ? ALZ_OK : fail(ALZ_E_HIP, "k_cgemm_dot launch failed");
}
#endif  // ALZ_TUNING

int launch_scan_cascade(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream, int64_t chunk_len,
                        ScanScratch *scratch, bool *taken, const char **kernel_name) {
  *taken = false;
  if (nsec < 2 || nsec > 4) return ALZ_OK;
  const bool cm = io.sxn == 1 && io.syn == 1;
  if (!cm) return ALZ_OK;
  for (int s = 0; s < nsec; ++s) {
    // every section keeps two outputs of state; only the first may look further back into its input
    if (secs[s].na != 3 || secs[s].any_div || !secs[s].uniform) return ALZ_OK;
    // biquad-class numerators only: gammatone.sampled's first section (8 taps of +-1e3 with heavy cancellation,
    // SURVEY.md 8a) makes the chunk-state recursion lose ten digits (2e-6 measured against 2e-11 for slaney); it
    // stays on the section-by-section mode
    if (secs[s].nb > 3 || secs[s].nb < 1) return ALZ_OK;
  }
  const int64_t C = io.channels;
  // chunks: a multiple of 64 per real channel (a 64-lane group = 64 chunks of one channel) that divides the
  // block into whole 16-sample tiles; by default enough of them to fill the chip (>= 1024 groups of 64)
  int64_t K = 0;
  if (chunk_len > 0) {
    if (io.n % chunk_len == 0) K = io.n / chunk_len;
  } else {
    int64_t want = (65536 + C - 1) / C;
    want = (want + 63) / 64 * 64;
    for (int64_t k = want; k >= 64; k -= 64)
      if (io.n % k == 0 && (io.n / k) % 16 == 0 && io.n / k >= 256) { K = k; break; }
  }
  if (K < 64 || K % 64 != 0) return ALZ_OK;
  const int64_t L = io.n / K;
  if (L % 16 != 0 || L < 64) return ALZ_OK;
  const int64_t V = K * C;

  // scratch: [section][x | y][2 or nb-1][V] + M
  uint64_t need = 0;
  for (int s = 0; s < nsec; ++s) need += (uint64_t)((secs[s].nb - 1) + (secs[s].na - 1)) * V * sizeof(double);
  uint64_t have = scratch->v_bytes;
  int rc = grow_scratch(&scratch->vxh, &have, need);
  if (rc) return rc;
  scratch->v_bytes = have;
  uint64_t have_p = scratch->power_bytes;
  rc = grow_scratch(&scratch->power, &have_p, (uint64_t)64 * C * sizeof(double));
  if (rc) return rc;
  if (have_p != scratch->power_bytes) scratch->power_len = 0;
  scratch->power_bytes = have_p;

  CScanArgs p;
  CascChunks ch;
  ch.n_chunks = K; ch.chunk_len = L;
  p.x = io.x; p.ldx = io.sxc; p.C = C; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.mode = io.mode; p.map_input = io.map_input; p.nsec = nsec; p.L = L; p.K = K; p.power = scratch->power;
  double *cur = scratch->vxh;
  for (int s = 0; s < 4; ++s) {
    const SectionDev &d = secs[s < nsec ? s : 0];
    p.nb[s] = d.nb; p.na[s] = d.na; p.b[s] = d.b; p.a[s] = d.a; p.xh[s] = d.xh; p.yh[s] = d.yh;
    p.vxh[s] = p.vyh[s] = nullptr;
    if (s < nsec) {
      p.vxh[s] = cur; cur += (int64_t)(d.nb - 1) * V;
      p.vyh[s] = cur; cur += (int64_t)(d.na - 1) * V;
    }
    ch.vxh[s] = p.vxh[s]; ch.vyh[s] = p.vyh[s];
  }
  const char *inner = "";
  hipLaunchKernelGGL(k_cscan_prep, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, stream, p);
  bool ok = false;
  const bool fresh_power = scratch->power_len != L || scratch->power_section != -2;
  bool dot_pass = false;
#ifdef ALZ_TUNING
  p.first_is_z = 0;
  dot_pass = ALZ_TUNE("ALZ_CSCAN_GEMM", 0) != 0 && io.mode == ALZ_BANK_OUTER && io.n_inputs == 1 && io.map_input &&
             C % 2 == 0 && L % 8 == 0 && (((uintptr_t)io.x) & 15) == 0;
  if (dot_pass) {
    rc = cgemm_zero_state_pass(p, scratch, fresh_power, stream);
    if (rc) return rc;
    p.first_is_z = 1;
  }
#endif
  if (!dot_pass) {
    ch.nostore = true;
    rc = launch_cascade_chunks(secs, nsec, io, stream, ch, &ok, &inner);
    if (rc) return rc;
    if (!ok) return ALZ_OK;               // (prep only touched scratch)
  }
  if (fresh_power) {
    hipLaunchKernelGGL(k_cscan_power, dim3((unsigned)((C * 2 * nsec + 63) / 64)), dim3(64), 0, stream, p);
    scratch->power_len = L;
    scratch->power_section = -2;          // (-2: this slot holds a cascade's matrix)
  }
  hipLaunchKernelGGL(k_cscan_fix, dim3((unsigned)((C + 7) / 8)), dim3(64), 0, stream, p);
  ch.nostore = false;
  rc = launch_cascade_chunks(secs, nsec, io, stream, ch, &ok, &inner);
  if (rc) return rc;
  if (!ok) return fail(ALZ_E_HIP, "time-parallel cascade: replay launch refused after the zero-state pass");
  hipLaunchKernelGGL(k_cscan_finish, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, p);
  ALZ_HIP_CHECK(hipGetLastError());
  *taken = true;
  *kernel_name = inner[2] == 'p' ? "k_cscan(k_pipe)" : "k_cscan(k_casc)";
  return ALZ_OK;
}

}  // namespace alz
