// alz_common.h -- shared declarations of libalzhip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/alz.h"

// Ablation switches (timing experiments that produce WRONG output) exist only in builds made with
// -DALZ_ABLATE (tools/variant builds loaded through ALZ_LIBRARY); in the shipped library the tests
// below are compile-time false and ALZ_WAVE_DEBUG in the environment does nothing.
#ifdef ALZ_ABLATE
#define ALZ_DBG(p, bit) (((p).dbg & (bit)) != 0)
#define ALZ_DBG_ENV() (getenv("ALZ_WAVE_DEBUG") ? atoi(getenv("ALZ_WAVE_DEBUG")) : 0)
#else
#define ALZ_DBG(p, bit) false
#define ALZ_DBG_ENV() 0
#endif

// Tuning overrides read from the environment exist only in -DALZ_TUNING builds (tools/variants, loaded through
// ALZ_LIBRARY for A/B runs); the shipped library carries the measured defaults and no run-time knobs.
#ifdef ALZ_TUNING
#include <stdlib.h>
#define ALZ_TUNE(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define ALZ_TUNE(name, dflt) (dflt)
#endif

namespace alz {

#if defined(__HIPCC__)
// An experiment hook (round 6, profiles/NOTES_r06.md 8.7; reachable in -DALZ_TUNING builds only): workgroup g starts g * ticks of
// the 100 MHz clock late.  The workgroups of a recurrence kernel leave their start together and advance at the same step rate, so on
// a channel-major block whose rows lie 2^k bytes apart the whole chip touches the same row offset at every moment; the same launch
// takes 12.3 or 16 ms depending on where its two blocks lie.  Breaking the lock-step does NOT change that (same buffers, stagger
// 0 / 0.2 / 0.8 us per workgroup: 15.96 / 16.06 / 16.25 ms in a slow placement, 12.30 / 12.34 / 12.48 in a fast one): it is the
// physical placement.  Kept so that the measurement can be repeated.
__device__ __forceinline__ void stagger_start(unsigned ticks, unsigned group) {
  if (ticks != 0u) {
    const long long until = (long long)wall_clock64() + (long long)group * (long long)ticks;
    while ((long long)wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
  }
}
// The common tile clock (round 6, alz_wave.hip launch_wave_impl: profiles/r06_duo_tilepace*.log).  A time-major block streams whole rows
// only while the workgroups of a launch touch the same rows together; free-running they drift apart.  The wave that requests a
// workgroup's tiles therefore asks for tile i no earlier than i x pace after ITS OWN start (pace16: 1/16 ticks of the 100 MHz clock per
// tile, from tile_pace16 below; 0: free-running).  A wave that is late does not wait: a slower box or a shared GPU degrades to the
// free-running rate, not below it.
// (`pace`: bits 0 - 19 the pace in 1/16 ticks, bits 20 - 30 FORGIVE in ticks, tuning builds only.  FORGIVE > 0: a wave more than that
// behind its schedule does not run free until it has caught up -- its schedule restarts from now (`shift`), so it still requests no
// more than a tile per pace.  Measured (profiles/r06_pace7_forgiving_clock.log, 8 / 40 / 150 ticks against the fixed clock, three
// kernels, rates below and above the knee): no difference anywhere -- what the clock buys is the workgroups being on the SAME rows,
// not a limit on the request rate.  A clock that shifts all workgroups by the largest lateness any of them reports (one word per
// launch, scalar atomic max: r06_pace6_feedback_ratchet.log) was worse than no clock: the maximum over 256 workgroups of every
// transient only ever grows.  So: a fixed clock, with margin.)
__device__ __forceinline__ void pace_wait(long long t0, long long i, int pace, long long &shift) {
  const int pace16 = pace & 0xFFFFF, forgive = pace >> 20;
  const long long due = t0 + ((i * (long long)pace16) >> 4) + shift;
  const long long now = (long long)wall_clock64();
  if (now < due) {
    if (due - now > (1ll << 18)) return;       // (no tile step of these kernels is 2.6 ms: a schedule that far ahead is not one to keep)
    do __builtin_amdgcn_s_sleep(1); while ((long long)wall_clock64() < due);
  } else if (forgive > 0 && now - due > forgive) {
    shift += now - due;
  }
}
// (experiment, tuning builds, k_duo only; profiles/r06_convoy.log, r06_convoy2.log.  Result: it does hold the order without a rate --
// one-pole banks 368 - 374 Gsamples/s at Q = 16, S = 2 ... 8 against the shipped clock's 359 and the best clock's 375 -- but every
// checkpoint is a scalar load's latency in the wave that requests the tiles: the FMA bank, whose helper wave is its limit, gets 337 -
// 342 against the clock's 350 (and 316 - 321 with clock AND convoy), 8192 channels gain nothing (257 - 286 against 350), short
// intervals are ruinous (Q = 2: 73) and the one-pole figure is not monotonic in Q (Q = 32: 323 - 332, Q = 64: 297).  Added to the clock
// it does stop the one-pole banks' collapse at rates past the knee (355 - 359 at 6000 - 7000 GB/s instead of 290) but not the others'.
// With the checkpoint in the storing wave instead (cfg bit 16; the per-tile barrier holds the workgroup; r06_convoy3.log) the FMA bank
// does reach 355 - 360 at Q = 8, S = 2 ... 4 -- and the one-pole banks fall to 317 - 359, 6144 channels to 134 - 250 against the clock's
// 282, 8192 channels stay at the free-running 239 - 280: which (Q, S) works depends on the shape.  Not shipped.)
// A CONVOY instead of a clock -- no rate to choose.  Every Q-th tile is a checkpoint; a wave announces the
// checkpoints it reaches in a ring of 64 counters (scalar atomic add) and passes checkpoint c only when ALL `groups` workgroups have
// reached checkpoint c - S, so no workgroup is more than (S + 1) Q tiles ahead of the slowest.  cfg = Q | S << 8.  The wait is bounded:
// a wave that has waited 100 us stops synchronising for the rest of the launch (`off`).)
__device__ __forceinline__ void convoy_sync(unsigned *ring, int cfg, unsigned groups, long long i, bool &off) {
  const int Q = cfg & 255, S = (cfg >> 8) & 255;
  if (off || (i % Q) != 0) return;
  const long long c = i / Q;
  const unsigned one = 1u;
  asm volatile("s_atomic_add %0, %1, 0x0" : : "s"(one), "s"(ring + (c & 63) * 16) : "memory");
  if (c < S) return;
  const long long w = c - S;
  const unsigned target = groups * (unsigned)(w / 64 + 1);
  const unsigned *slot = ring + (w & 63) * 16;
  long long t0 = 0;
  for (;;) {
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(slot) : "memory");
    if (v >= target) return;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) t0 = now;
    else if (now - t0 > 10000) { off = true; return; }
    __builtin_amdgcn_s_sleep(2);
  }
}
#endif
// the counter ring of one convoy launch on `stream` (64 counters, a 64-byte line each), zeroed in stream order; nullptr: none (alz_scan.hip)
unsigned *convoy_ring(hipStream_t stream);
// pace of a launch whose workgroups together move bytes_per_step per tile step, at gbps (GB/s; <= 0: no pacing): 1/16 ticks of 10 ns
// (the ticks are those of the device's wall clock -- s_memrealtime: 100 MHz on gfx950, asked of the runtime per device; a device that
// does not say runs free: a mis-scaled clock would THROTTLE, the waits are not bounded)
int device_wall_clock_khz();
inline int tile_pace16(long long bytes_per_step, int gbps) {
  const long long khz = gbps > 0 ? device_wall_clock_khz() : 0;
  const int pace16 = khz > 0 ? (int)((bytes_per_step * 16ll * khz / 1000ll + gbps * 500ll) / (gbps * 1000ll)) : 0;
  return pace16 > 0 && pace16 < (1 << 20) ? (pace16 | (ALZ_TUNE("ALZ_PACE_FORGIVE", 0) << 20)) : 0;
}

// Workgroups of a launch that run TOGETHER when a CU holds one of them: all of them up to the CU count; of a larger launch the CU count
// when its rounds are full ones (the last round at least 3/4 of a full one: each round starts where the one before -- kept in step by
// the clock -- ends, and paces itself from its own start), 0 otherwise (a short last round would be throttled to a clock sized for a
// full one: 320 and 512 groups on a clock for all of them lost 20 - 37 %, profiles/r06_pace2.log).
inline long long paced_groups_one_per_cu(long long groups, int cus) {
  if (groups <= cus) return groups;
  const long long last = groups % cus;
  return (last == 0 || 4 * last >= 3ll * cus) ? cus : 0;
}

// thread-local last-error message (alz_last_error)
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
// thread-local name of what the last handle-less call (alz_lpc_*, alz_acorr_dev, alz_levinson_dev*, alz_tv_process_dev)
// launched (alz_last_kernel): a diagnostic, like alz_bank_last_kernel for banks
void note_kernel(const std::string &name, bool append = false);

#define ALZ_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess)                                                               \
      return ::alz::fail(ALZ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for kernel `fn` on the CURRENT device, remembered
// per (kernel, device, size) under a mutex: a process may drive several devices and several
// threads may create banks at once (alz_api.hip).
int ensure_dynamic_lds(const void *fn, int bytes);

// One cascaded section as the kernels see it.  Device arrays are tap-major so
// that lane == channel reads are coalesced:
//   b[k * n_sets + set], a[k * n_sets + set]
//   xh[k * channels + c] = input  of this section at time -1-k   (k < nb-1)
//   yh[k * channels + c] = output of this section at time -1-k   (k < na-1)
// Pacing slab of k_fir_ring's chains (alz_fir.hip): start stamps of the runs of the launch in flight, owned by the bank,
// one per section; grown on demand, never shrunk.  Stamps carry the launch's epoch, so the slab is never cleared.
struct FirChains {
  unsigned long long *flags = nullptr;
  size_t len = 0;
  unsigned epoch = 0;
};

struct SectionDev {
  FirChains *chains = nullptr;   // nullptr: k_fir_ring keeps its interleaved mapping
  int nb, na;
  const double *b, *a;
  double *xh, *yh;
  unsigned present_b;  // bit k set: b_k != 0 for at least one set
  unsigned present_a;  // bit k-1 set: a_k != 0 for at least one set (k >= 1)
  bool uniform;        // every set has the same zero pattern
  bool any_div;        // some a_0 != 1
  bool shared_sets;    // n_sets == 1
  // sparse view: delays of the taps that are non-zero for at least one set (ascending);
  // n_ff / n_fb = -1 when there are more than 8 of them
  int n_ff, n_fb;
  int tap_b[8], tap_a[8];
};

// Block description handed to the launchers.
struct BlockIO {
  const double *x;
  double *y;
  int64_t n;          // samples per channel
  int64_t sxn, sxc;   // x element strides (time, channel)
  int64_t syn, syc;   // y element strides
  int64_t channels;   // output channels of the bank (state array stride)
  int64_t c_first, c_count;  // channel range this launch covers
  int64_t n_inputs;   // input channels
  int64_t n_sets;     // coefficient sets
  int mode;           // ALZ_BANK_DIAGONAL / ALZ_BANK_OUTER
  int map_input;      // OUTER mode: this launch reads the n_inputs input channels
  double zero;        // what an all-zero section yields (lazy_filters.py:227-231)
  int fused;          // opt-in FMA contraction in the streaming kernels (not bit-exact)
  int pre_op;         // elementwise stage fused into this section's input reads (ALZ_MAP_ABS) or 0;
                      // honoured by launch_wave and the k_small path of launch_section only
  int look_sync = 0;    // the process call will wait for a one-pass time-parallel launch (ALZ_LOOK_CHECK_CALL): short blocks are
                      // then better off in three launches, which nobody has to wait for (scan_takes_one_pass, ALZ_TP_AUTO)
  int stream_once = 0;  // the block is large, read once and its result not read again by this call (single-section
                      // bank): k_duo then moves it with non-temporal loads and stores
};

// alz_iir.hip: any section shape, channels [c_first, c_first + c_count)
int launch_section(const SectionDev &sec, const BlockIO &io, hipStream_t stream,
                   const char **kernel_name);
// alz_casc.hip: a whole cascade (2..4 sections of a fused pattern) in one pass; reports the
// samples / channels it covered (full 16-sample tiles of full 64-channel groups)
int launch_cascade(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream,
                   int64_t *done_samples, int64_t *done_channels, const char **kernel_name);
// the same over n_chunks chunks of the time axis at once (time-parallel mode, channel-major blocks): per-section
// state arrays of [taps - 1][n_chunks * io.channels], slot real_channel * n_chunks + chunk; `nostore` runs the
// cascade for its end states only.  *taken = false: nothing launched.
struct CascChunks {
  int64_t n_chunks, chunk_len;
  bool nostore;
  bool chunk_major = false;  // slots j * channels + c and 64-channel groups inside ONE chunk (always for time-major blocks; for
                             // channel-major ones with a single input stream) instead of c * n_chunks + j
  bool probe = false;        // report whether the launch would take the block, launch nothing
  double *vxh[4], *vyh[4];
};
int launch_cascade_chunks(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream,
                          const CascChunks &ch, bool *taken, const char **kernel_name);
// alz_comb.hip: sparse sections whose feedback delays are all long (comb filters): the step kernels k_comb_tm /
// k_comb_cm (either layout, in place when the numerator is b0 alone) or k_sparse (time-major, x != y); *taken says
// whether the shape was one of theirs
int launch_sparse(const SectionDev &sec, const BlockIO &io, hipStream_t stream, bool *taken,
                  const char **kernel_name);
// whether launch_sparse runs this section IN PLACE on this block (io.x == io.y; the step kernels k_comb_tm / k_comb_cm
// with the single numerator tap b0): the caller then needs no copy of the block
bool comb_takes_in_place(const SectionDev &sec, const BlockIO &io);
// alz_fir.hip: long feedback-free sections on time-major blocks (x != y); *taken says whether
// the shape was this kernel's
int launch_fir(const SectionDev &sec, const BlockIO &io, hipStream_t stream, bool *taken,
               const char **kernel_name);
// alz_mid.hip: one-section IIR filters of order 3 .. 8 (dense coefficients) and short numerators with one far tap
// (maverage.recursive) as a three-wave streaming kernel on time-major blocks; reports what it covered like launch_wave
int launch_mid(const SectionDev &sec, const BlockIO &io, hipStream_t stream, int64_t *done_samples, int64_t *done_channels,
               const char **kernel_name);
// alz_wave.hip: the streaming kernel takes the full tiles of the full channel groups it
// can and reports how much that was; the caller finishes the rest with launch_section
int launch_wave(const SectionDev &sec, const BlockIO &io, hipStream_t stream,
                int64_t *done_samples, int64_t *done_channels, const char **kernel_name);
// the same kernels over n_chunks chunks of the time axis at once (blockIdx.y = chunk): chunk j covers
// samples [j * chunk_len, (j + 1) * chunk_len) of io.x / io.y and keeps its state in slot
// j * io.channels + c of vxh / vyh (arrays of [taps - 1][n_chunks * io.channels]); `nostore` runs the
// recurrence for its end state only.  *taken = false (nothing launched) when the shape is not one
// the streaming kernels cover completely.
struct WaveChunks {
  int64_t n_chunks, chunk_len;
  bool nostore;
  double *vxh, *vyh;
};
int launch_wave_chunks(const SectionDev &sec, const BlockIO &io, hipStream_t stream, const WaveChunks &ch,
                       bool *taken, const char **kernel_name);
// alz_scan.hip: time-parallel execution of one biquad-class section (chunked state propagation:
// zero-state pass, per-channel scan of the chunk states, replay); opt-in, not bit-exact.
struct ScanScratch {                 // owned by the bank handle, grown on demand
  double *vxh = nullptr, *vyh = nullptr;
  uint64_t v_bytes = 0;
  double *power = nullptr;           // [4][channels] transition matrix A^chunk_len per channel
  uint64_t power_bytes = 0;
  int64_t power_len = 0;             // chunk length the cached matrix belongs to (0: none)
  int power_section = -1;
  double *hr = nullptr, *edge = nullptr;   // k_cdot: the cascade's impulse responses per set and tap, and the edge responses
  uint64_t hr_bytes = 0, edge_bytes = 0;
  int64_t tab_len = 0;               // chunk length the tables belong to (0: none)
  double *zbuf = nullptr;            // k_look: published chunk end states
  uint64_t zbuf_bytes = 0;
  int *look_err = nullptr;           // k_look: kLookErrWords words of pinned host memory, one per wait site (alz_look.hip W_*): the kernel
                                     // sets the word of a bounded wait that ran out
};
// alz_tvduo.hip: time-varying biquad-class filter with bank-wide coefficient series, streaming kernel (recurrence, feed-forward and store waves)
int launch_tvduo(const double *x, double *y, int64_t n, int64_t ldx, int64_t ldy, int cm, int64_t channels, int nb, int na,
                 const int *kind, const double *value, const double *const *series, const int *negated,
                 double *xh, double *yh, hipStream_t stream, int64_t *done_samples);
// alz_tvpc.hip: time-varying biquad-class filter with PER-CHANNEL coefficient series (time-major rows), three-wave kernel
int launch_tvpc(const double *x, double *y, int64_t n, int64_t ldx, int64_t ldy, int64_t channels, int nb, int na,
                const int *kind, const double *value, const double *const *series, const int64_t *series_ld,
                const int *negated, double *xh, double *yh, hipStream_t stream, int64_t *done_samples);
// alz_map.hip: one elementwise op over n contiguous doubles (see alz_map_dev)
int launch_map(int op, const double *x, const double *y, double p0, double p1, int64_t n, double *out,
               int *flags, hipStream_t stream);
int launch_levinson_dense(const double *r, int64_t n_frames, int n_lags, int order, double *coefs, double *err,
                          int *status, hipStream_t st);
int launch_expand(const double *x, double *xe, int64_t n, int64_t channels, int64_t n_inputs, int64_t sxn, int64_t sxc,
                  int64_t sen, int64_t sec, hipStream_t stream);
// alz_look.hip: the time-parallel mode of one biquad-class section in ONE pass (chunks resident in LDS, chunk states
// through global memory); see the file header
int launch_look(const SectionDev &sec, const BlockIO &io, hipStream_t stream, const double *power, double *zbuf,
                uint64_t zbuf_bytes, int *err, int64_t *done_samples, const char **kernel_name);
constexpr int64_t kLookChunk = 512;
uint64_t look_scratch_bytes(int64_t groups, int64_t chunks);   // what launch_look needs in `zbuf` for that many 16-channel groups and chunks
constexpr int kLookErrWords = 16;
constexpr int64_t kTpThreeLaunch = -3;   // internal chunk_len value: the engine's chunk length, never the one-pass form (the re-run of a block
                                         // on which the one-pass kernel gave up)
// names of the wait sites, for messages (index = word of look_err)
const char *look_wait_name(int site);
// compute units of the CURRENT device (asked once per device and thread)
int device_cus();
// whether launch_look would take this section and block (shape only; `cus` = the device's CU count)
bool look_takes(const SectionDev &sec, const BlockIO &io, int cus);
// whether launch_scan would send this section and block to the one-pass form for the handle's chunk-length setting
bool scan_takes_one_pass(const SectionDev &sec, const BlockIO &io, int64_t chunk_len);
// time-parallel execution of a whole fused cascade (see alz_scan.hip) on a channel-major block, or a time-major one whose
// channels come in whole groups of 64; *taken = false when the shape is not covered (nothing written but scratch).
// state_consistent: every section's input history equals its predecessor's output history (true after reset and after
// every block) -- what the dot-product zero-state pass needs for chunk 0.
int launch_scan_cascade(const SectionDev *secs, int nsec, const BlockIO &io, hipStream_t stream, int64_t chunk_len,
                        ScanScratch *scratch, bool state_consistent, bool *taken, const char **kernel_name);
int launch_scan(const SectionDev &sec, int section_index, const BlockIO &io, hipStream_t stream,
                int64_t chunk_len, ScanScratch *scratch, int64_t *done_samples, const char **kernel_name);

#ifdef __HIPCC__
// Progress counters in LDS, one writer each.  The LDS executes a wave's operations in the order it issued them, so a
// counter written after the data (or after the reads that free a slot) needs no wait in between, and a reader that has
// seen the counter sees the data; the compiler is held to the same order by the empty asm statements.
__device__ __forceinline__ void publish(int *flag, int value, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
// `site`: which wait this is (alz_look.hip's W_* codes) -- a wait that runs out sets err[site], so the host can say which
// of the protocol's dependencies did not arrive
__device__ __forceinline__ void await(const int *flag, int need, int &cap, int *err, int site = 0) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > cap) { if (err) err[site] = 1; cap = 0; break; }   // (cannot happen: every wait points to earlier work; a wave
  }                                                        //  that gave up once no longer waits at all)
  asm volatile("" ::: "memory");
}

#endif

}  // namespace alz
