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

// One pass where the shape allows it (a recursive section, whole 512-sample chunks, either layout since round 5):
// 512-sample chunks resident in LDS, the block read once (alz_look.hip) -- 260 Gsamples/s with 16 B/sample of traffic at
// 512 channels x 2^20 against 228 with 24 for the three-launch form (profiles/NOTES_r03.md).  ALZ_TP_ONE_PASS asks for
// it; ALZ_TP_AUTO takes it when its workgroups (one per CU: 16 channels x up to 16 chunks in flight) fill most of the
// chip, i.e. from about 200 channels up; narrower banks fill the chip better as chunks x channels lanes of the
// three-launch form.
int device_cus() {
  thread_local int cached_dev = -1, cached = 0;      // (per thread: handles of different devices may be driven from different threads)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != cached_dev) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cached = cus;
    cached_dev = dev;
  }
  return cached;
}

unsigned *convoy_ring(hipStream_t stream) {
  constexpr int kRings = 8;                           // launches in flight that may each own a ring
  constexpr size_t kRingBytes = 64 * 64;
  thread_local int cached_dev = -1;
  thread_local char *rings = nullptr;
  thread_local unsigned next = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (dev != cached_dev || !rings) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    char *w = nullptr;
    if (hipMalloc((void **)&w, kRings * kRingBytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // (kept for the process)
    rings = w; cached_dev = dev; next = 0;
  }
  char *ring = rings + (size_t)(next++ % kRings) * kRingBytes;
  if (hipMemsetAsync(ring, 0, kRingBytes, stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return reinterpret_cast<unsigned *>(ring);
}

int device_wall_clock_khz() {
  thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != cached_dev) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) { (void)hipGetLastError(); khz = 0; }
    cached = khz;
    cached_dev = dev;
  }
  return cached;
}

bool scan_takes_one_pass(const SectionDev &sec, const BlockIO &io, int64_t chunk_len) {
  if (!(sec.nb <= 3 && sec.na <= 3 && sec.uniform) || sec.any_div || sec.na < 2) return false;
  if (io.c_first != 0 || io.c_count != io.channels || io.channels % 16) return false;
  if (io.n < 4 * kLookChunk) return false;
  const int cus = device_cus();
  bool one_pass = chunk_len == ALZ_TP_ONE_PASS;          // (kTpThreeLaunch: the engine's chunk length, never this form)
  if (chunk_len == ALZ_TP_AUTO) {
    const int64_t groups = io.channels / 16, Kl = io.n / kLookChunk;
    int64_t wk = groups > 0 ? cus / groups : 0;
    wk = wk > 16 ? 16 : wk;
    wk = wk > Kl ? Kl : wk;
    one_pass = wk >= 2 && 4 * groups * wk >= 3 * (int64_t)cus;
    // the launching call synchronises with a one-pass launch (alz_bank_set_look_check, default): ~60 us per block, more than
    // a short block's kernels take in either form -- the three-launch form needs no such wait
    if (io.look_sync && Kl < 128) one_pass = false;
  }
  return one_pass && look_takes(sec, io, cus);
}

int launch_scan(const SectionDev &sec, int section_index, const BlockIO &io, hipStream_t stream,
                int64_t chunk_len, ScanScratch *scratch, int64_t *done_samples, const char **kernel_name) {
  *done_samples = 0;
  if (!(sec.nb <= 3 && sec.na <= 3 && sec.uniform) || sec.any_div) return ALZ_OK;
  if ((sec.present_b | sec.present_a) == 0) return ALZ_OK;
  if (io.c_first != 0 || io.c_count != io.channels) return ALZ_OK;
  const int64_t C = io.channels;
  if (C % 16) return ALZ_OK;
  const bool one_pass = scan_takes_one_pass(sec, io, chunk_len);
  if (chunk_len < 0) chunk_len = 0;
  if (one_pass) {
    const int64_t groups = C / 16, Kl = io.n / kLookChunk;
    // (a wait that ran out in an earlier launch is reported by alz_api.hip's take_look_error at every entry point)
    if (!scratch->look_err) {
      if (hipHostMalloc((void **)&scratch->look_err, 64, hipHostMallocDefault) != hipSuccess)
        return fail(ALZ_E_NOMEM, "hipHostMalloc failed (time-parallel scratch)");
      for (int k = 0; k < kLookErrWords; ++k) scratch->look_err[k] = 0;
    }
    const uint64_t zneed = look_scratch_bytes(groups, Kl);
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
  // (the three-launch form below reads the block as it is: an input map fused into the reads is the one-pass kernel's and
  // the serial kernels' only -- the caller then finishes the section with those)
  if (io.pre_op) return ALZ_OK;
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
  int64_t ldx;                 // elements between input rows (channel-major) / between samples (time-major)
  int64_t C, n_inputs, n_sets; // C = real channels of the bank
  int mode, map_input, nsec;
  int nb[4], na[4];
  const double *b[4], *a[4];
  double *xh[4], *yh[4];       // the bank's state [taps-1][C]
  double *vxh[4], *vyh[4];     // per-chunk state [taps-1][C * K]
  int64_t L, K;
  double *power;               // M[r][e] at power[(r * 8 + e) * C + c]
  int slot_tm;                 // state slot of (real channel c, chunk j): 0 -> c * K + j (a 64-lane group of the cascade kernel =
                               // 64 chunks of one channel), 1 -> j * C + c (chunk-major: 64 channels of one chunk -- time-major
                               // blocks, and channel-major ones whose single input stream is broadcast)
  int x_tm;                    // the block is time-major (x[t * ldx + in]) rather than channel-major (x[in * ldx + t])
  int first_is_z;              // slot 0 of vyh holds z_0 (the dot-product zero-state pass): the fix starts at chunk 0 from the bank's state
  int xcd_map;                 // k_cdot3: workgroups that share a band group (its slice of the response table) on ONE XCD
};

__device__ __forceinline__ int64_t cs_slot(const CScanArgs &p, int64_t c, int64_t j) { return p.slot_tm ? j * p.C + c : c * p.K + j; }
// sample t of input row `in`
__device__ __forceinline__ int64_t cs_xat(const CScanArgs &p, int64_t in, int64_t t) { return p.x_tm ? t * p.ldx + in : in * p.ldx + t; }

__global__ __launch_bounds__(256) void k_cscan_prep(CScanArgs p) {
  const int64_t vc = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t V = p.K * p.C;
  if (vc >= V) return;
  int64_t real, j;
  if (p.slot_tm) { j = vc / p.C; real = vc - j * p.C; } else { real = vc / p.K; j = vc - real * p.K; }
  const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? real % p.n_inputs : real;
  for (int s = 0; s < p.nsec; ++s) {
    for (int k = 0; k < p.nb[s] - 1; ++k) {
      double v = 0.0;
      if (j == 0) v = p.xh[s][(int64_t)k * p.C + real];
      else if (s == 0) v = p.x[cs_xat(p, in, j * p.L - 1 - k)];
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

// S_{j+1} = M S_j + z_j per real channel, serially over the chunks; S = (y_s[-1], y_s[-2]) of every section (the next
// section's input history is the same two numbers).  On entry vyh holds the end states of the zero-state pass -- from the
// cascade pass slot 0 is the true S_1 and slots j > 0 are z_j; from the dot-product pass (first_is_z) every slot is z_j and
// the recursion starts from the bank's state.  On exit every slot holds its chunk's true initial state (vyh[s], and
// vxh[s + 1]: the same numbers), and the BANK's state is the state after the last chunk, S_K = M S_{K-1} + z_{K-1} (with
// section 0's input history from the block's last samples): no separate finish launch, and it equals what the replay of
// the last chunk will leave to within the rounding every chunk boundary carries.
//
// A DPP row of 16 lanes per channel (four channels per wave): lane r < 8 keeps row r of M and component r of S; the
// product needs every component in every lane -- v_mov_b32_dpp row_newbcast:e, two per component and step, no LDS round
// trip (round 3's form, eight lanes per channel exchanging through LDS: ~800 cycles per chunk, 96 us for 256 chunks).
// Lanes 8 .. 15 mirror 0 .. 7 and take the vxh stores off them.  Chunks go in blocks of eight: the z of the next block
// are in flight while this one is chained, a block's states leave as 16-byte stores where the slots of consecutive
// chunks are adjacent (channel-major).
template <int E>
__device__ __forceinline__ double cs_bcast(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x150 + E, 0xf, 0xf, false);          // row_newbcast:E
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + E, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

template <bool SLOT_TM>
__global__ __launch_bounds__(64) void k_cscan_fix(CScanArgs p) {
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x, r = lane & 7, g = lane >> 4;
  const bool mirror = (lane & 8) != 0;
  int64_t c = (int64_t)blockIdx.x * 4 + g;
  const bool live_c = c < p.C;
  if (!live_c) c = p.C - 1;                                  // (keeps EXEC full for the DPP moves; nothing is stored)
  const int64_t V = p.K * p.C;
  const int ND = 2 * p.nsec;
  constexpr int B = 8;
  const int s = r >> 1, k = r & 1;
  const bool row_on = r < ND && k < p.na[s < p.nsec ? s : 0] - 1;
  double M[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) M[e] = (r < ND && e < ND) ? p.power[((int64_t)r * 8 + e) * p.C + c] : 0.0;
  const int64_t sj = SLOT_TM ? p.C : 1, base = SLOT_TM ? c : c * p.K;
  const double *zsrc = row_on ? p.vyh[s] + (int64_t)k * V + base : nullptr;
  // lanes 0 .. 7 store the chunk's output state, lanes 8 .. 15 the next section's input history (the same numbers)
  double *dst = nullptr;
  if (live_c && row_on && !mirror) dst = p.vyh[s] + (int64_t)k * V + base;
  if (live_c && r < ND && mirror && s + 1 < p.nsec && k < p.nb[s + 1] - 1) dst = p.vxh[s + 1] + (int64_t)k * V + base;
  const double bank_y = row_on ? p.yh[s][(int64_t)k * p.C + c] : 0.0;
  // what slot 0 gets: the bank's state.  (After the cascade pass the next section's input history of chunk 0 is the
  // bank's own x history -- k_cscan_prep put it there for that pass, and a set_state may have made it differ from y.)
  double first = bank_y;
  if (!p.first_is_z && mirror && dst) first = p.xh[s + 1][(int64_t)k * p.C + c];
  double S = bank_y;
  auto fetch = [&](int64_t j0, double (&z)[B]) {
    if (j0 >= p.K) j0 = p.K - B;                             // (the look-ahead past the end reads the last block again)
    if constexpr (!SLOT_TM) {
#pragma unroll
      for (int u = 0; u < B; u += 2) {
        dbl2 v = {0.0, 0.0};
        if (zsrc) v = *reinterpret_cast<const dbl2 *>(zsrc + j0 + u);
        z[u] = v.x; z[u + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int u = 0; u < B; ++u) z[u] = zsrc ? zsrc[(j0 + u) * sj] : 0.0;
    }
  };
  auto chain = [&](int64_t j0, const double (&z)[B]) {
    double out[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const bool head = j0 + u == 0;
      out[u] = head ? first : S;
      const double s0 = cs_bcast<0>(S), s1 = cs_bcast<1>(S), s2 = cs_bcast<2>(S), s3 = cs_bcast<3>(S);
      const double s4 = cs_bcast<4>(S), s5 = cs_bcast<5>(S), s6 = cs_bcast<6>(S), s7 = cs_bcast<7>(S);
      double a0 = __builtin_fma(M[0], s0, z[u]);
      double a1 = M[1] * s1;
      a0 = __builtin_fma(M[2], s2, a0);
      a1 = __builtin_fma(M[3], s3, a1);
      a0 = __builtin_fma(M[4], s4, a0);
      a1 = __builtin_fma(M[5], s5, a1);
      a0 = __builtin_fma(M[6], s6, a0);
      a1 = __builtin_fma(M[7], s7, a1);
      const double next = a0 + a1;
      S = (head && !p.first_is_z) ? z[u] : next;             // (cascade pass: slot 0 already is the true S_1)
    }
    if (dst) {
      if constexpr (!SLOT_TM) {
#pragma unroll
        for (int u = 0; u < B; u += 2) {
          dbl2 v = {out[u], out[u + 1]};
          *reinterpret_cast<dbl2 *>(dst + j0 + u) = v;
        }
      } else {
#pragma unroll
        for (int u = 0; u < B; ++u) dst[(j0 + u) * sj] = out[u];
      }
    }
  };
  double za[B], zb[B];
  fetch(0, za);
  for (int64_t j0 = 0; j0 < p.K; j0 += 2 * B) {               // (K is a multiple of 64)
    fetch(j0 + B, zb);
    chain(j0, za);
    fetch(j0 + 2 * B, za);
    chain(j0 + B, zb);
  }
  // the bank's state after the block
  if (live_c && row_on && !mirror) p.yh[s][(int64_t)k * p.C + c] = S;
  if (live_c && r < ND && mirror && s + 1 < p.nsec && k < p.nb[s + 1] - 1) p.xh[s + 1][(int64_t)k * p.C + c] = S;
  if (live_c && !mirror && r < p.nb[0] - 1 && r < 2) {        // section 0's input history: the block's last samples (exact)
    const int64_t in = (p.mode == ALZ_BANK_OUTER && p.map_input) ? c % p.n_inputs : c;
    p.xh[0][(int64_t)r * p.C + c] = p.x[cs_xat(p, in, p.K * p.L - 1 - r)];
  }
}

// ---------------------------------------------------------------------------
// The zero-state pass as dot products (round 4's experiment, shipped in round 5).  A chunk's zero-state end state is
// linear in the chunk's input:
//   z_j[s][k] = y_s[L - 1 - k] = sum_m h_s[L - 1 - k - m] x_j[m] + e1[s][k] x_j[-1] + e2[s][k] x_j[-2],
// h_s the impulse response of sections 0 .. s, e1 / e2 the responses to a unit sample in section 0's input history --
// 8 fused multiply-adds per input sample and band where the cascade pass issues ~28 instructions and keeps nothing but
// its last state.  The tables (one row of 4 doubles per set and tap, 32 B x sets x L) are built once per bank and chunk
// length by running the cascade's own arithmetic over a unit impulse.  OUTER banks that read their input by input index
// (the reference's filterbank shape: lazy_auditory.py:158-218 through examples/gammatone_plots.py:47); the bank's state
// must be self-consistent (section s + 1's input history = section s's output history: true after reset and after every
// block, not after an arbitrary set_state -- the handle keeps a flag) because chunk 0 goes through S_1 = M S_0 + z_0
// like every other chunk.
//
// k_cdot: lane = chunk (64 consecutive chunks of one input row per wave: every lane reads its own chunk by 16-byte
// loads), NS bands per workgroup with their responses wave-uniform in SGPRs, the sum over the chunk split over the SPLIT
// waves of the workgroup, whose partial sums meet in LDS and are added in segment order (deterministic) -- no partial
// sums in HBM, no reduction launch.  Wave 0 also leaves section 0's input history of every chunk (the block itself:
// exact) in vxh[0], which was k_cscan_prep's job.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cdot_tables(CScanArgs p, double *__restrict__ hr, double *__restrict__ edge) {
  const int64_t set = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (set >= p.n_sets) return;
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
      if (run == 0) {                       // hr[(set L + m) 4 + s] = h_s[L - 1 - m]: the weight of x[m] in y_s[L - 1]
        double *dst = hr + ((int64_t)set * p.L + (p.L - 1 - n)) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[s] = s < p.nsec ? out[s] : 0.0;
      }
    }
    if (run > 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        edge[(((int64_t)set * 2 + (run - 1)) * 4 + s) * 2 + 0] = s < p.nsec ? y1[s] : 0.0;
        edge[(((int64_t)set * 2 + (run - 1)) * 4 + s) * 2 + 1] = s < p.nsec ? y2[s] : 0.0;
      }
    }
  }
}

// what both forms of the dot pass do with a wave's sums: segment 0 adds the edge terms (the two samples before the chunk,
// which it also leaves in vxh[0] for the replay), the workgroup's partial sums meet in LDS and are added in segment order
template <int NS, int SPLIT>
__device__ __forceinline__ void cdot_finish(const CScanArgs &p, const double *__restrict__ edge, double (&acc)[NS][4][2], double *cd_part,
                                            int lane, int seg, int64_t j, int64_t in, int64_t set0, const double *xrow) {
  const int64_t V = p.K * p.C;
  if (seg == 0) {
    // the two samples before the chunk -- from the block, or the bank's input history for chunk 0 -- enter through the
    // edge responses, and are what the replay of this chunk starts section 0 from
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      const int64_t c = (set0 + a) * p.n_inputs + in;
      double xm1 = 0.0, xm2 = 0.0;
      if (p.nb[0] > 1) xm1 = j > 0 ? xrow[-1] : p.xh[0][0 * p.C + c];
      if (p.nb[0] > 2) xm2 = j > 0 ? xrow[-2] : p.xh[0][1 * p.C + c];
      const int64_t slot = cs_slot(p, c, j);
      if (p.nb[0] > 1) p.vxh[0][0 * V + slot] = xm1;
      if (p.nb[0] > 2) p.vxh[0][1 * V + slot] = xm2;
      const double *e = edge + (set0 + a) * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          double z = acc[a][s][k];
          z = __builtin_fma(e[(0 * 4 + s) * 2 + k], xm1, z);
          z = __builtin_fma(e[(1 * 4 + s) * 2 + k], xm2, z);
          acc[a][s][k] = z;
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int k = 0; k < 2; ++k) cd_part[((seg * NS + a) * 8 + 2 * s + k) * 64 + lane] = acc[a][s][k];
  __syncthreads();
  constexpr int QPW = NS * 8 / SPLIT;                          // sums per wave
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int q = seg * QPW + i, a = q >> 3, s = (q >> 1) & 3, k = q & 1;   // (wave-uniform)
    if (s >= p.nsec) continue;
    double z = cd_part[((0 * NS + a) * 8 + 2 * s + k) * 64 + lane];
#pragma unroll
    for (int sg = 1; sg < SPLIT; ++sg) z = z + cd_part[((sg * NS + a) * 8 + 2 * s + k) * 64 + lane];
    const int64_t c = (set0 + a) * p.n_inputs + in;
    p.vyh[s][(int64_t)k * V + cs_slot(p, c, j)] = z;
  }
}

template <int NS, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void k_cdot(CScanArgs p, const double *__restrict__ hr, const double *__restrict__ edge) {
  extern __shared__ __attribute__((aligned(16))) double cd_part[];   // [SPLIT][NS * 8][64]
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  static_assert((NS * 8) % SPLIT == 0, "the final sums are shared out over the waves");
  const int lane = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t ngrp = p.n_sets / NS;
  const int64_t in = (int64_t)blockIdx.y / ngrp, set0 = ((int64_t)blockIdx.y - in * ngrp) * NS;
  const int64_t Ls = p.L / SPLIT, m0 = seg * Ls, m1 = m0 + Ls;
  const double *xrow = p.x + in * p.ldx + j * p.L;            // (channel-major rows, or ONE contiguous time-major column)
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
    for (int q = 0; q < 8; ++q) h[a][q] = hr[((set0 + a) * p.L + m0) * 4 + q];
  for (int64_t m = m0; m < m1; m += 2) {
    const int64_t mn = m + 2 < m1 ? m + 2 : m;                 // (the last step requests its own pair again)
    const dbl2 vn = *reinterpret_cast<const dbl2 *>(xrow + mn);
    double hn[NS][8];
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
      for (int q = 0; q < 8; ++q) hn[a][q] = hr[((set0 + a) * p.L + mn) * 4 + q];
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
  cdot_finish<NS, SPLIT>(p, edge, acc, cd_part, lane, seg, j, in, set0, xrow);
}

// k_cdot3 (round 5, second form): the same sums in the same order -- identical doubles -- with the operand streams laid
// out for the memory system instead of for the compiler.  k_cdot's counters (profiles/r05_cdot_pmc.txt): its waves WAIT 69 %
// of their cycles (VALU active 22 %), every 16-byte row load is a vector-cache miss of its own (TCP_TCC_READ_REQ =
// TCP_TOTAL_CACHE_ACCESSES: the eight loads that share a 128-byte line are eight L2 requests, iterations apart), and the
// four s_load_dwordx16 of a pair of samples are waited for where they are issued.  Here
//   * a lane reads its row a whole 128-byte line at a time: eight 16-byte loads issued together for 16 samples, the next
//     16 samples' loads in flight while this step's 512 FMAs run (two register sets);
//   * the responses come in HALF steps -- bands 0 - 1 of a pair, then bands 2 - 3 -- through two sets of 32 SGPRs: while
//     one half's 32 FMAs run, the other half's two s_load_dwordx16 are in flight (a full double buffer of the 64 SGPRs
//     a pair of samples needs for four bands does not fit the SGPR file);
//   * hipcc's scheduler would hoist the FMAs above the requests (and wait at once): sched_barriers pin the order.
#define ALZ_SLOAD2(d0, d1, p0, p1) \
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(d0), "=&s"(d1) : "s"(p0), "s"(p1) : "memory")
#define ALZ_SWAIT2(d0, d1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1) : : "memory")
#define ALZ_XLOAD(dst, ptr, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=&v"(dst) : "v"(ptr) : "memory")
template <int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void k_cdot3(CScanArgs p, const double *__restrict__ hr, const double *__restrict__ edge) {
  extern __shared__ __attribute__((aligned(16))) double cd_part3[];  // [SPLIT][NS * 8][64]
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  typedef double dbl8 __attribute__((ext_vector_type(8)));
  constexpr int NS = 4;
  static_assert((NS * 8) % SPLIT == 0, "the final sums are shared out over the waves");
  const int lane = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Workgroups are dealt to the eight XCDs by their linear index mod 8 (x fastest): with the launch's natural order the four chunk
  // groups of a band group sit on four XCDs and each pulls that band group's slice of the response table through its own L2.
  // xcd_map (grid.y a multiple of 8): XCD e takes the band groups y = e (mod 8), every chunk group of them -- a slice is fetched
  // by one XCD.
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (p.xcd_map) {
    const unsigned id = bx + gridDim.x * by, per = gridDim.y >> 3, loc = id >> 3;
    by = (id & 7u) + 8u * (loc % per);
    bx = loc / per;
  }
  const int64_t j = (int64_t)bx * 64 + lane;
  const int64_t ngrp = p.n_sets / NS;
  const int64_t in = (int64_t)by / ngrp, set0 = ((int64_t)by - in * ngrp) * NS;
  const int64_t Ls = p.L / SPLIT, m0 = seg * Ls;
  const int nsteps = (int)(Ls / 16);
  const double *xrow = p.x + in * p.ldx + j * p.L;
  double acc[NS][4][2];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[a][s][0] = acc[a][s][1] = 0.0;
  double xprev = seg > 0 ? xrow[m0 - 1] : 0.0;
  asm volatile("" : "+v"(xprev));                               // (waited for here: no vmcnt wait of hipcc's inside the loop)
  // the responses of bands set0 .. set0 + 3 at sample m0: 8 doubles (64 bytes) per pair of samples and band
  const double *h0 = hr + ((set0 + 0) * p.L + m0) * 4, *h1 = hr + ((set0 + 1) * p.L + m0) * 4;
  const double *h2 = hr + ((set0 + 2) * p.L + m0) * 4, *h3 = hr + ((set0 + 3) * p.L + m0) * 4;
  dbl8 a0, a1, b0, b1;                                          // bands 0 - 1 / bands 2 - 3 of the pair in work
  dbl2 xa[8], xb[8];
  auto load_line = [&](dbl2 (&d)[8], const double *src) {
    ALZ_XLOAD(d[0], src, 0); ALZ_XLOAD(d[1], src, 16); ALZ_XLOAD(d[2], src, 32); ALZ_XLOAD(d[3], src, 48);
    ALZ_XLOAD(d[4], src, 64); ALZ_XLOAD(d[5], src, 80); ALZ_XLOAD(d[6], src, 96); ALZ_XLOAD(d[7], src, 112);
  };
  auto landed = [&](dbl2 (&d)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : : "memory");
  };
#define ALZ_HALF(A, C0, C1, V)                                                     \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                  \
    acc[A][s][0] = __builtin_fma(C0[s], V.x, acc[A][s][0]);                        \
    acc[A][s][1] = __builtin_fma(C0[s], xprev, acc[A][s][1]);                      \
    acc[A][s][0] = __builtin_fma(C0[4 + s], V.y, acc[A][s][0]);                    \
    acc[A][s][1] = __builtin_fma(C0[4 + s], V.x, acc[A][s][1]);                    \
    acc[A + 1][s][0] = __builtin_fma(C1[s], V.x, acc[A + 1][s][0]);                \
    acc[A + 1][s][1] = __builtin_fma(C1[s], xprev, acc[A + 1][s][1]);              \
    acc[A + 1][s][0] = __builtin_fma(C1[4 + s], V.y, acc[A + 1][s][0]);            \
    acc[A + 1][s][1] = __builtin_fma(C1[4 + s], V.x, acc[A + 1][s][1]);            \
  }
  // (hipcc may sink a half's FMAs below the requests that follow them -- nothing but data flow orders arithmetic against
  // an asm statement -- and then keeps the OLD response buffers alive in spilled copies: first build, 579 SGPR spills.  An
  // empty asm that "uses" the half's sixteen sums pins the FMAs in front of it.)
#define ALZ_PIN(A)                                                                 \
  asm volatile("" : "+v"(acc[A][0][0]), "+v"(acc[A][0][1]), "+v"(acc[A][1][0]), "+v"(acc[A][1][1]), "+v"(acc[A][2][0]),       \
               "+v"(acc[A][2][1]), "+v"(acc[A][3][0]), "+v"(acc[A][3][1]), "+v"(acc[A + 1][0][0]), "+v"(acc[A + 1][0][1]),    \
               "+v"(acc[A + 1][1][0]), "+v"(acc[A + 1][1][1]), "+v"(acc[A + 1][2][0]), "+v"(acc[A + 1][2][1]),               \
               "+v"(acc[A + 1][3][0]), "+v"(acc[A + 1][3][1]));
  // 16 samples from the register set CUR (the table has one spare pair behind its end: the last pair's look-ahead)
#define ALZ_STEP(CUR)                                                              \
  _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                  \
    const dbl2 v = CUR[q];                                                         \
    ALZ_SWAIT2(a0, a1);                      /* bands 0 - 1 of this pair have landed */ \
    ALZ_SLOAD2(b0, b1, h2, h3);              /* bands 2 - 3 of this pair */        \
    __builtin_amdgcn_sched_barrier(0);                                             \
    ALZ_HALF(0, a0, a1, v)                                                         \
    ALZ_PIN(0)                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                             \
    h0 += 8; h1 += 8;                                                              \
    ALZ_SWAIT2(b0, b1);                      /* bands 2 - 3 have landed */         \
    ALZ_SLOAD2(a0, a1, h0, h1);              /* bands 0 - 1 of the NEXT pair */    \
    __builtin_amdgcn_sched_barrier(0);                                             \
    ALZ_HALF(2, b0, b1, v)                                                         \
    ALZ_PIN(2)                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                             \
    h2 += 8; h3 += 8;                                                              \
    xprev = v.y;                                                                   \
  }
  load_line(xa, xrow + m0);
  ALZ_SLOAD2(a0, a1, h0, h1);
  landed(xa);
  for (int t = 0; t < nsteps; t += 2) {
    // (an odd step count re-reads the last line into the spare set: loaded, never used)
    load_line(xb, xrow + m0 + 16 * (int64_t)(t + 1 < nsteps ? t + 1 : t));
    __builtin_amdgcn_sched_barrier(0);
    ALZ_STEP(xa)
    landed(xb);
    if (t + 1 < nsteps) {
      load_line(xa, xrow + m0 + 16 * (int64_t)(t + 2 < nsteps ? t + 2 : t + 1));
      __builtin_amdgcn_sched_barrier(0);
      ALZ_STEP(xb)
      landed(xa);
    }
  }
  ALZ_SWAIT2(a0, a1);                                           // (the request that was issued behind the last pair)
  cdot_finish<NS, SPLIT>(p, edge, acc, cd_part3, lane, seg, j, in, set0, xrow);
}
#undef ALZ_HALF
#undef ALZ_PIN
#undef ALZ_STEP
#undef ALZ_SLOAD2
#undef ALZ_SWAIT2
#undef ALZ_XLOAD

int launch_scan_cascade(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream, int64_t chunk_len,
                        ScanScratch *scratch, bool state_consistent, bool *taken, const char **kernel_name) {
  *taken = false;
  if (nsec < 2 || nsec > 4) return ALZ_OK;
  const bool cm = io.sxn == 1 && io.syn == 1;
  // time-major blocks: the 64 lanes of a cascade group are 64 adjacent channels of ONE chunk (rows of 512 contiguous
  // bytes in the output), so the bank's channels must come in whole groups, and the input either has a column per
  // group lane (inputs % 64 == 0) or is one stream every band reads (broadcast by scalar loads in k_casc)
  const bool by_input = io.mode == ALZ_BANK_OUTER && io.map_input;
  const bool tm = !cm && io.sxc == 1 && io.syc == 1 && io.channels % 64 == 0 &&
                  (by_input ? (io.n_inputs == 1 || io.n_inputs % 64 == 0) : true);
  if (!cm && !tm) return ALZ_OK;
  for (int s = 0; s < nsec; ++s) {
    // every section keeps two outputs of state; only the first may look further back into its input
    if (secs[s].na != 3 || secs[s].any_div || !secs[s].uniform) return ALZ_OK;
    // biquad-class numerators only: gammatone.sampled's first section (8 taps of +-1e3 with heavy cancellation,
    // SURVEY.md 8a) makes the chunk-state recursion lose ten digits (2e-6 measured against 2e-11 for slaney); it
    // stays on the section-by-section mode
    if (secs[s].nb > 3 || secs[s].nb < 1) return ALZ_OK;
  }
  const int64_t C = io.channels;
  // chunks: a multiple of 64 per real channel (channel-major: a 64-lane group = 64 chunks of one channel; the fix kernel
  // walks them in blocks of 16) that divides the block into whole 16-sample tiles; by default enough of them to fill
  // the chip (>= 1024 groups of 64)
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
  // chunk-major virtual channels: always for time-major blocks; for channel-major ones when ONE input stream feeds every
  // band (then no input tiles travel at all: k_casc<bc>) and the chunks fill the chip with single-wave cascades
  ch.chunk_major = !cm || (by_input && io.n_inputs == 1 && C % 64 == 0 && V >= 65536);
  p.x = io.x; p.ldx = cm ? io.sxc : io.sxn; p.C = C; p.n_inputs = io.n_inputs; p.n_sets = io.n_sets;
  p.mode = io.mode; p.map_input = io.map_input; p.nsec = nsec; p.L = L; p.K = K; p.power = scratch->power;
  p.x_tm = cm ? 0 : 1;
  p.first_is_z = 0;
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
  bool ok = false;
  // will the fused cascade kernel take the replay?  Asked before anything is launched: the fix kernel writes the bank's
  // state ahead of the replay, and a pattern without a fused instantiation must leave the block to the other paths
  // untouched (round 5's fuzzer found the missing question: tools/fuzz_timeparallel.py)
  ch.nostore = false;
  ch.probe = true;
  rc = launch_cascade_chunks(secs, nsec, io, stream, ch, &ok, &inner);
  if (rc) return rc;
  if (!ok && cm && ch.chunk_major) {      // (no broadcast instantiation for this pattern: the chunks of a channel as lanes)
    ch.chunk_major = false;
    rc = launch_cascade_chunks(secs, nsec, io, stream, ch, &ok, &inner);
    if (rc) return rc;
  }
  if (!ok) return ALZ_OK;
  ch.probe = false;
  p.slot_tm = ch.chunk_major ? 1 : 0;
  const bool fresh_power = scratch->power_len != L || scratch->power_section != -2;

  // the zero-state pass: dot products with the cascade's impulse responses where the shape offers them (an OUTER bank
  // reading by input index from rows of chunks: channel-major, or the one contiguous column of a one-stream time-major
  // block), else the cascade kernel itself without stores
  // (measured and not kept, profiles/NOTES_r05.md 2: two bands per workgroup with both operand streams requested one step
  // ahead by hand-placed s_load / global_load and sched_barriers -- 305 against 353 Gsamples/s: half the bands per wave is
  // twice the row loads, and those cost the vector cache a line access per lane)
  constexpr int NS = 4, SPLIT = 8;
  const bool dot_pass = state_consistent && by_input && io.x != io.y && io.n_sets % NS == 0 && L % (2 * SPLIT) == 0 &&
                        (((uintptr_t)io.x) & 15) == 0 && (cm ? (p.ldx % 2 == 0) : (io.n_inputs == 1 && p.ldx == 1)) &&
                        (uint64_t)(K / 64) <= 65535u && (uint64_t)(io.n_sets / NS) * (uint64_t)io.n_inputs <= 65535u;
  if (dot_pass) {
    // (+ one pair of samples: k_cdot3 requests the responses one pair ahead, also behind the last set's last pair)
    const uint64_t hr_need = ((uint64_t)io.n_sets * L + 2) * 4 * sizeof(double), edge_need = (uint64_t)io.n_sets * 16 * sizeof(double);
    uint64_t have_h = scratch->hr_bytes, have_e = scratch->edge_bytes;
    rc = grow_scratch(&scratch->hr, &have_h, hr_need);
    if (rc) return rc;
    rc = grow_scratch(&scratch->edge, &have_e, edge_need);
    if (rc) return rc;
    const bool fresh_tab = have_h != scratch->hr_bytes || have_e != scratch->edge_bytes || scratch->tab_len != L;
    scratch->hr_bytes = have_h; scratch->edge_bytes = have_e;
    if (fresh_tab) {
      hipLaunchKernelGGL(k_cdot_tables, dim3((unsigned)((io.n_sets + 63) / 64)), dim3(64), 0, stream, p, scratch->hr, scratch->edge);
      scratch->tab_len = L;
    }
    const int lds = SPLIT * NS * 8 * 64 * (int)sizeof(double);
    p.first_is_z = 1;
#ifndef ALZ_CDOT_V3
#define ALZ_CDOT_V3 1
#endif
    // (k_cdot3 wants whole 128-byte lines per lane and step: segments of a multiple of 16 samples, 128-byte aligned chunks)
    const bool lines = ALZ_TUNE("ALZ_CDOT_V3", ALZ_CDOT_V3) != 0 && L % (16 * SPLIT) == 0 && (((uintptr_t)io.x) & 127) == 0 && (io.n_inputs == 1 || p.ldx % 16 == 0);
    const void *dot_fn = lines ? (const void *)k_cdot3<SPLIT> : (const void *)k_cdot<NS, SPLIT>;
    rc = ensure_dynamic_lds(dot_fn, lds);
    if (rc) return rc;
    p.xcd_map = (((io.n_sets / NS) * io.n_inputs) % 8 == 0 && ALZ_TUNE("ALZ_CDOT_XCD", 1) != 0) ? 1 : 0;
    if (lines)
      hipLaunchKernelGGL((k_cdot3<SPLIT>), dim3((unsigned)(K / 64), (unsigned)((io.n_sets / NS) * io.n_inputs)), dim3(64 * SPLIT),
                         lds, stream, p, (const double *)scratch->hr, (const double *)scratch->edge);
    else
      hipLaunchKernelGGL((k_cdot<NS, SPLIT>), dim3((unsigned)(K / 64), (unsigned)((io.n_sets / NS) * io.n_inputs)), dim3(64 * SPLIT),
                         lds, stream, p, (const double *)scratch->hr, (const double *)scratch->edge);
  } else {
    hipLaunchKernelGGL(k_cscan_prep, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, stream, p);
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
  if (!p.slot_tm) hipLaunchKernelGGL(k_cscan_fix<false>, dim3((unsigned)((C + 3) / 4)), dim3(64), 0, stream, p);
  else hipLaunchKernelGGL(k_cscan_fix<true>, dim3((unsigned)((C + 3) / 4)), dim3(64), 0, stream, p);
  ch.nostore = false;
  rc = launch_cascade_chunks(secs, nsec, io, stream, ch, &ok, &inner);
  if (rc) return rc;
  if (!ok) return fail(ALZ_E_HIP, dot_pass ? "time-parallel cascade: the fused cascade kernel refused the replay after the dot-product pass"
                                           : "time-parallel cascade: replay launch refused after the zero-state pass");
  ALZ_HIP_CHECK(hipGetLastError());
  *taken = true;
  const bool bc = inner[2] == 'c' && inner[6] == '<';       // "k_casc<bc>": the broadcast-input instantiation
  *kernel_name = dot_pass ? (inner[2] == 'p' ? "k_cscan(k_cdot+k_pipe)" : bc ? "k_cscan(k_cdot+k_casc<bc>)" : "k_cscan(k_cdot+k_casc)")
                          : (inner[2] == 'p' ? "k_cscan(k_pipe)" : bc ? "k_cscan(k_casc<bc>)" : "k_cscan(k_casc)");
  return ALZ_OK;
}

}  // namespace alz
