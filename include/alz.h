/*
 * alz.h -- C ABI of libalzhip.so, the MI355X (gfx950) blocked stream-filter engine.
 *
 * The reference (AudioLazy 0.6.1dev, pure Python) has no FFI: its boundary for
 * this path is the callable protocol "a filter is any callable that receives an
 * iterable and returns a Stream" (audiolazy/lazy_filters.py:975-978, 1033-1036)
 * with the signature __call__(seq, memory=None, zero=0.) (:141, :840).  Each
 * entry point below names the reference lines whose *execution* it replaces;
 * the Python mirror of the callable protocol lives in audiolazy_amd/ and binds
 * these symbols through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns an int status (ALZ_OK or a negative ALZ_E_*);
 *     alz_last_error() returns a thread-local message for the last failure.
 *   - plain pointers and sizes only; all samples, coefficients and state are
 *     IEEE binary64 (the reference computes in CPython floats).
 *   - "dev" pointers are device (HBM) pointers, "host" pointers host memory.
 *     Data buffers are caller-owned; coefficients are copied at create time.
 *   - a handle is bound to one device and is not thread-safe; distinct handles
 *     may be used from distinct threads.  Calls are asynchronous with respect
 *     to the hipStream_t passed as `stream` (NULL = the default stream).
 *   - arithmetic: Direct Form I in the reference's generated term order with
 *     separately rounded multiply and add (no FMA), zero coefficients absent
 *     from the sum, division by a0 only when a0 != 1 -- results are
 *     bit-identical to the reference generator (lazy_filters.py:197-257).
 */
#ifndef ALZ_H
#define ALZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALZ_VERSION 320 /* 0.4.0: + alz_bank_set_look_check, alz_bank_look_stats (the one-pass time-parallel kernel verified on the call that launched it); 0.3.2: time-parallel mode for time-major cascades, one-pass form in both layouts / in place / with the |x| map (same entry points); 0.3.1: + alz_lag_matrix_dev; 0.3.0: + alz_comm_* (direct RCCL), one-launch ALZ_LPC_DENSE, any LPC order; 0.2.1: + alz_levinson_dev_ex, ALZ_LPC_DENSE (0.2.0: alz_map_dev, alz_bank_set_input_map, alz_bank_set_time_parallel, alz_lpc_kautocor_dev_ex) */

/* status codes; the Python shim re-raises the reference's exception types */
#define ALZ_OK 0
#define ALZ_E_ARG (-1)         /* bad argument                      -> ValueError */
#define ALZ_E_NONCAUSAL (-2)   /* lazy_filters.py:165-168           -> ValueError("Non-causal filter") */
#define ALZ_E_ZERO_GAIN (-3)   /* lazy_filters.py:177-178           -> ZeroDivisionError("Invalid filter gain") */
#define ALZ_E_PARCOR (-4)      /* lazy_lpc.py:132-133               -> ParCorError */
#define ALZ_E_HIP (-5)         /* HIP runtime failure               -> RuntimeError */
#define ALZ_E_NOMEM (-6)       /* allocation failure                -> MemoryError */
#define ALZ_E_UNSUPPORTED (-7) /* shape outside the engine's gate   -> NotImplementedError */

/* sample layouts: element (n, c) of a block of n samples x C channels */
#define ALZ_TIME_MAJOR 0 /* [N, C]: addr = n*ld + c   (the reference's vector-valued-sample rows) */
#define ALZ_CHAN_MAJOR 1 /* [C, N]: addr = c*ld + n   (one Stream per row) */

/* bank modes */
#define ALZ_BANK_DIAGONAL 0 /* channel c: input c, coefficient set c (n_sets == n_inputs) */
#define ALZ_BANK_OUTER 1    /* channel c = set*n_inputs + input: every coefficient set on
                               every input (a filterbank; n_sets == 1 is "shared taps") */

typedef struct alz_bank alz_bank_t;

/* ---- library / device ---------------------------------------------------- */
int alz_version(void);
const char *alz_last_error(void);
/* Name of the kernel(s) the last handle-less call of this thread launched (alz_lpc_kautocor_dev*, alz_acorr_dev,
 * alz_levinson_dev*, alz_tv_process_dev): a diagnostic like alz_bank_last_kernel; bench.py and the tests read the
 * kernel names from here instead of assuming them. */
const char *alz_last_kernel(void);
int alz_device_count(int *count);
/* device memory helpers so a caller needs no other HIP binding */
int alz_malloc(int device, uint64_t bytes, void **dev_ptr);
int alz_free(int device, void *dev_ptr);
int alz_memcpy_h2d(int device, void *dst_dev, const void *src_host, uint64_t bytes);
int alz_memcpy_d2h(int device, void *dst_host, const void *src_dev, uint64_t bytes);
int alz_device_sync(int device);

/* ---- filter bank ---------------------------------------------------------
 * Replaces the execution of LinearFilter.__call__ (lazy_filters.py:141-264),
 * ZFilter.__call__ (:840-889) and CascadeFilter.__call__ (:988-990) for
 * `channels` independent streams at once.
 *
 * A bank is `n_sections` cascaded sections (CascadeFilter order); section s has
 * nb[s] numerator taps b_0..b_{nb-1} and na[s] denominator taps a_0..a_{na-1}
 * (the reference's numlist / denlist, lazy_filters.py:55-67).  Coefficients:
 *   b: [n_sets, sum(nb)]   a: [n_sets, sum(na)]   row-major, sections concatenated.
 * Errors: ALZ_E_ZERO_GAIN if any a_0 == 0; ALZ_E_ARG on bad sizes.
 */
int alz_bank_create(int64_t n_sets, int64_t n_inputs, int mode, int n_sections,
                    const int *nb, const int *na, const double *b_host,
                    const double *a_host, int device, alz_bank_t **out);
int alz_bank_destroy(alz_bank_t *h);
int alz_bank_channels(const alz_bank_t *h, int64_t *channels);

/* State = the reference generator's local variables (lazy_filters.py:243-250):
 *   xh[c][hx(s) + k] = d_{k+1} of section s = its input  at time -1-k
 *   yh[c][hy(s) + k] = m_{k+1} of section s = its output at time -1-k
 * with hx(s) = sum_{t<s}(nb[t]-1), hy(s) = sum_{t<s}(na[t]-1); host arrays
 * [channels, sum(nb-1)] and [channels, sum(na-1)], row-major.
 * alz_bank_reset == memory=None: every d and m is `zero` (:185-186, :247-250).
 * The state persists across process calls, so consecutive blocks equal one
 * continuous reference run.                                                   */
int alz_bank_reset(alz_bank_t *h, double zero);
int alz_bank_set_state(alz_bank_t *h, const double *xh_host, const double *yh_host);
int alz_bank_get_state(alz_bank_t *h, double *xh_host, double *yh_host);

/* Filter one block of n samples per channel.
 *   x: n_inputs channels, y: `channels` channels, both in `layout`;
 *   ldx / ldy: leading dimension in elements (TIME_MAJOR: >= channel count,
 *   CHAN_MAJOR: >= n).  y may alias x only in DIAGONAL mode with ldx == ldy.
 * process_host stages through device buffers owned by the handle.
 * Device scratch is allocated on first need by the call that needs it and kept
 * by the handle (chunk states of the time-parallel mode; for a long feedback-
 * free section on a big time-major block, a slab of one 8-byte start stamp per
 * 48 rows and 64 channels -- 1/384 of the block -- through which the waves that
 * share input rows pace each other; nothing is READ through it: same doubles). */
int alz_bank_process_dev(alz_bank_t *h, const double *x_dev, double *y_dev,
                         int64_t n, int layout, int64_t ldx, int64_t ldy,
                         void *stream);
int alz_bank_process_host(alz_bank_t *h, const double *x_host, double *y_host,
                          int64_t n, int layout, int64_t ldx, int64_t ldy);
int alz_bank_sync(alz_bank_t *h);
/* Opt-in fused mode: allow v_fma_f64 contraction -- in the streaming biquad kernel (half the
 * recurrence chain) and in the shared-tap FIR kernel (one instruction per tap instead of two: the
 * throughput mode of a feedback-free section, same ascending tap order).  Results are then NOT
 * bit-identical to the reference generator (normalised differences ~1e-13, contract 1e-6);
 * default off.  No reference counterpart (CPython floats never fuse).  The mode ALLOWS the
 * contraction: where the default kernel is the faster one (a time-major biquad bank of 4096 - 5120
 * channels, bound by its helper wave rather than by the recurrence) the engine keeps it.   */
int alz_bank_set_fused(alz_bank_t *h, int on);
/* Opt-in time-parallel mode for narrow banks (the reference's generator is one serial chain per
 * channel, lazy_filters.py:251-257, so a bank of a few hundred channels cannot fill the GPU):
 * biquad-class sections (nb, na <= 3, a0 == 1) are run as chunks of `chunk_len` samples in
 * parallel -- zero-state pass, per-channel propagation of the chunk states through the section's
 * transition matrix, replay from the true states.  chunk_len: 0 = off (default), -1 = chosen by
 * the engine, > 0 = that many samples (rounded down to a multiple of 64).  Every sample is still
 * the DF-I statement, but the chunk states carry a different rounding: results are NOT
 * bit-identical (<= 1e-6 normalised by contract, ~1e-12 typical, ~1e-9 for poles at radius
 * 0.9999).  Sections the mode does not cover run as usual.
 * The engine's choice (ALZ_TP_AUTO) includes the ONE-PASS form for a biquad-class section (either layout, in place,
 * and with the ALZ_MAP_ABS input map read through the kernel -- round 5; time-major only before): 512-sample chunks
 * stay in LDS between the zero-state sums and the replay, so the block is read once (16 bytes of HBM traffic per
 * sample instead of 24; same numerics).  It is taken when its workgroups fill most of the chip (from about 200
 * channels up to 2048); ALZ_TP_ONE_PASS (-2) asks for it on any shape it covers, a positive chunk_len always means
 * the three-launch form.
 * Cascades of 2 - 4 sections with two poles and at most three numerator taps each (gammatone.slaney / klapuri,
 * lazy_auditory.py:184-218) run FUSED in this mode: the chunks of the time axis become the fused cascade kernel's
 * channels; the zero-state pass is a dot product of every chunk with the cascade's impulse responses when the bank is
 * an OUTER bank reading by input index and its state is self-consistent (every section's input history = its
 * predecessor's output history: after reset, after any processed block, after a set_state whose rows say so --
 * otherwise the cascade kernel itself makes that pass).  Channel-major blocks, and time-major ones (the reference's
 * vector-valued samples) when the bank's channels are a multiple of 64 and its inputs are one stream or a multiple of
 * 64.  Measured <= 3e-11 against the oracle (tests: 1e-9).  A cascade whose FIRST section has more than three
 * numerator taps (gammatone.sampled: eight taps of +-1e3 that cancel) is split -- numerator exact and time-parallel,
 * its recursion serial and exact, the remaining sections fused and chunked in place: measured 1e-8 at a 50 Hz band,
 * which is that form's bar (tests/test_gpu_outer_narrow.py).                                                       */
#define ALZ_TP_AUTO (-1)
#define ALZ_TP_ONE_PASS (-2)
int alz_bank_set_time_parallel(alz_bank_t *h, int64_t chunk_len);
/* The one-pass form is the one kernel of the library whose workgroups wait for each other (chunk states travel from
 * workgroup to workgroup while the block is in flight); it is launched cooperatively -- all its workgroups fit on the
 * device at once -- but foreign work holding compute units can still keep some of them from starting, so its waits are
 * BOUNDED and a launch can give up (having written a bad block and a bad bank state).  What the handle does then:
 *   ALZ_LOOK_CHECK_CALL (default): the process call that launched the kernel keeps a copy of the bank's state
 *     (one small launch), waits for its own work on `stream`, and if the kernel gave up puts the state back and
 *     processes the block again with the three-launch form of the same mode (same numerics): the caller never sees
 *     it (alz_bank_last_kernel shows it, alz_bank_look_stats counts it).  An IN-PLACE block cannot be processed
 *     again -- that call fails with ALZ_E_HIP, the bank's state as before the call.  Costs the call its asynchrony.
 *   ALZ_LOOK_CHECK_DEFERRED: the call stays asynchronous; a give-up is reported (ALZ_E_HIP, naming the waits that
 *     ran out) by the NEXT entry point of the handle that hands results over or takes a block (process, sync,
 *     get_state), and the bank must be reset / set_state'd by the caller.
 * No reference counterpart (the reference is one serial generator per channel, lazy_filters.py:251-257).            */
#define ALZ_LOOK_CHECK_DEFERRED 0
#define ALZ_LOOK_CHECK_CALL 1
int alz_bank_set_look_check(alz_bank_t *h, int mode);
/* Counters of the one-pass kernel on this handle since it was created: launches, launches that gave up, blocks
 * processed again because of that, and the waits that ran out in the last such launch (bit k = wait site k of
 * csrc/alz_look.hip's W_* list).  Any pointer may be NULL.  Diagnostic (tests/test_gpu_look_soak.py).             */
int alz_bank_look_stats(const alz_bank_t *h, int64_t *launches, int64_t *gave_up, int64_t *reruns, unsigned *last_sites);

/* Name of the kernel variant the last process call dispatched to (diagnostic;
 * tests use it to prove the fast paths are the ones exercised).              */
const char *alz_bank_last_kernel(const alz_bank_t *h);

/* ---- LPC -------------------------------------------------------------------
 * Replaces lpc.kautocor (lazy_lpc.py:229-272) = acorr(blk, order)
 * (lazy_analysis.py:277-312) + levinson_durbin (lazy_lpc.py:52-136) for a batch
 * of frames: frame f = sig[f*hop .. f*hop + frame_len).
 *   coefs [n_frames, order+1] (numlist of the analysis FIR, coefs[.][0] == 1),
 *   err   [n_frames]          (the filter's .error attribute),
 *   status[n_frames]          ALZ_OK or ALZ_E_PARCOR (zero-energy frame; the
 *                             reference raises ParCorError, lazy_lpc.py:132-133).
 * Floating point: the lags are bit-exact; Levinson-Durbin is the standard O(order^2) recursion
 * instead of the reference's dense inner products (tests hold it to 1e-9 normalised error, measured
 * ~1e-16).  alz_lpc_kautocor_dev_ex with ALZ_LPC_DENSE is the bit-identical form. */
int alz_lpc_kautocor_dev(const double *sig_dev, int64_t n_frames, int frame_len,
                         int64_t hop, int order, double *coefs_dev,
                         double *err_dev, int *status_dev, int device, void *stream);
/* The same with options.  ALZ_LPC_FUSED: the autocorrelation sums use one fused multiply-add per term
 * instead of a separately rounded multiply and add (the kernel is bound by FP64 issue: half the
 * instructions).  Same ascending order per lag; the lags are then NOT bit-identical to
 * lazy_analysis.py:311-312 (relative differences ~1e-16; contract 1e-6).  Needs frames of >= 32
 * samples, an even hop, >= 16384 frames and a curated order (8, 10, 12, 16, 20, 24, 32); other shapes
 * run the exact kernels whatever the flag says.  No reference counterpart (CPython never fuses). */
#define ALZ_LPC_FUSED 1
/* ALZ_LPC_DENSE: Levinson-Durbin with the reference's own dense inner products in the reference's
 * order (lazy_lpc.py:121-131: inner(a, b) = sum(acdata[|i-j|] * a_i * b_j ...), O(order^3) per frame)
 * on lags from the exact autocorrelation kernels: coefficients AND error bit-identical to
 * lpc.kautocor / levinson_durbin on every frame, at about 1.5 x the time of the default path. */
#define ALZ_LPC_DENSE 2
int alz_lpc_kautocor_dev_ex(const double *sig_dev, int64_t n_frames, int frame_len,
                            int64_t hop, int order, double *coefs_dev,
                            double *err_dev, int *status_dev, int flags, int device, void *stream);
/* levinson_durbin alone (lazy_lpc.py:52-136) on ready-made lag lists r [n_frames, n_lags];
 * n_lags <= order is zero-extended like the reference does (:117-118). */
int alz_levinson_dev(const double *r_dev, int64_t n_frames, int n_lags, int order,
                     double *coefs_dev, double *err_dev, int *status_dev, int device,
                     void *stream);
/* The same with options: flags = ALZ_LPC_DENSE for the reference's dense, bit-identical form. */
int alz_levinson_dev_ex(const double *r_dev, int64_t n_frames, int n_lags, int order,
                        double *coefs_dev, double *err_dev, int *status_dev, int flags, int device,
                        void *stream);
/* acorr alone (lazy_analysis.py:277-312): r [n_frames, max_lag+1] */
int alz_acorr_dev(const double *sig_dev, int64_t n_frames, int frame_len,
                  int64_t hop, int max_lag, double *r_dev, int device, void *stream);
/* lag_matrix (lazy_analysis.py:315-342) for every frame: phi [n_frames, max_lag+1, max_lag+1], cell (j, i) =
 * sum(blk[n - i] * blk[n - j] for n in max_lag .. frame_len - 1), added left to right from 0: the doubles of the
 * reference.  These are the statistics lpc.covar (lazy_lpc.py:275-294) and lpc.kcovar (:297-340) start from; their
 * small dense solves stay on the host (audiolazy_amd/lpc.py), like lpc.nautocor's.  max_lag >= frame_len is
 * ALZ_E_ARG (the reference's ValueError("Block length should be higher than order"), :337-338). */
int alz_lag_matrix_dev(const double *sig_dev, int64_t n_frames, int frame_len,
                       int64_t hop, int max_lag, double *phi_dev, int device, void *stream);

/* ---- either side of the path: mixdown and sample formats ---------------------------------- */
/* ParallelFilter.__call__ (lazy_filters.py:1048-1054): the outputs of the filters fed with the
 * same input summed ((y_0 + y_1) + y_2) ..., over the n_sets coefficient sets of an OUTER bank's
 * output block y (channel = set * n_inputs + input, same layouts and pitches as
 * alz_bank_process_dev) into out (n_inputs channels).  The order is the reference's, so the sum is
 * bit-identical to it. */
int alz_mix_dev(const double *y_dev, int64_t n_sets, int64_t n_inputs, int64_t n, int layout,
                int64_t ldy, int64_t ldo, double *out_dev, int device, void *stream);
/* Streamix (lazy_stream.py:633-724) for tracks that are already arrays: track k (device pointer
 * tracks_dev[k], lengths[k] samples) enters the mix at output sample starts[k];
 * out[n] = ((zero + t_a[n - s_a]) + t_b[n - s_b]) ... over the tracks playing at n, in the order
 * given (= the order they were added; the reference sums ``data += next(snd)`` over its playing
 * list).  tracks_dev / starts / lengths are HOST arrays of n_tracks entries. */
int alz_mix_tracks_dev(int n_tracks, const double *const *tracks_dev, const int64_t *starts,
                       const int64_t *lengths, double zero, int64_t n_out, double *out_dev,
                       int device, void *stream);
/* WavStream's sample conversion (lazy_wav.py:58-61, 110-128): n_samples little-endian PCM items of
 * 8 (unsigned), 16, 24 or 32 bits, in file order (interleaved channels = a time-major block), to
 * float64: v / 2**(bits-1) with 8-bit data re-centred by -128, or the stored integer when keep != 0.
 * raw_dev 4-byte aligned, out_dev 16-byte aligned. */
int alz_pcm_decode_dev(const void *raw_dev, int bits, int keep, int64_t n_samples,
                       double *out_dev, int device, void *stream);
/* chunks (lazy_io.py:44-128): n float64 items packed as struct format character dfmt
 * ('b' 'B' 'h' 'H' 'i' 'I' 'l' 'L' 'f' 'd'), little endian unless big_endian != 0.  *flags_dev
 * (an int the caller zeroes) receives ALZ_PCM_NOT_INTEGER / ALZ_PCM_RANGE / ALZ_PCM_FLOAT_OVERFLOW
 * bits for the items struct.pack would refuse (struct.error / OverflowError in the reference). */
#define ALZ_PCM_NOT_INTEGER 1
#define ALZ_PCM_RANGE 2
#define ALZ_PCM_FLOAT_OVERFLOW 4
int alz_pcm_encode_dev(const double *in_dev, int64_t n, int dfmt, int big_endian, void *out_dev,
                       int *flags_dev, int device, void *stream);

/* ---- elementwise stages around the filter ----------------------------------------------------- */
/* The lazy per-sample Stream expressions the reference wraps around a filter call: the operator
 * table of every Stream (lazy_stream.py:47-71: + - * / with a number or with another Stream, unary
 * minus, abs), ``clip`` (lazy_analysis.py:619-647) and the squaring / root of ``envelope``
 * (lazy_analysis.py:440-520), on a block of n contiguous float64 items: out[i] = op(x[i], ...).
 * Every op is the IEEE operation CPython performs on the same operands (bit-identical results);
 * ALZ_MAP_SQUARE is x * x, which is NOT what the reference's ``x ** 2`` (libm pow) returns for about
 * one sample in a thousand (last bit) -- use it knowingly.  p0 / p1: the scalar operand, or the
 * low / high clipping limits.  y_dev: second block of the two-operand ops (NULL otherwise).
 * out_dev may alias x_dev.  *flags_dev (an int the caller zeroes, or NULL) collects
 * ALZ_MAP_ZERODIV (a zero divisor: Python raises ZeroDivisionError) and ALZ_MAP_DOMAIN (root of a
 * negative item: Python returns a complex number). */
#define ALZ_MAP_ABS 1       /* abs(x)                                  */
#define ALZ_MAP_NEG 2       /* -x                                      */
#define ALZ_MAP_SQRT 3      /* x ** .5                                 */
#define ALZ_MAP_SQUARE 4    /* x * x   (see above)                     */
#define ALZ_MAP_MUL 5       /* x * p0  (== p0 * x)                     */
#define ALZ_MAP_ADD 6       /* x + p0  (== p0 + x)                     */
#define ALZ_MAP_SUB 7       /* x - p0                                  */
#define ALZ_MAP_RSUB 8      /* p0 - x                                  */
#define ALZ_MAP_DIV 9       /* x / p0                                  */
#define ALZ_MAP_RDIV 10     /* p0 / x                                  */
#define ALZ_MAP_CLIP 11     /* clip(x, low = p0, high = p1)            */
#define ALZ_MAP_CLIP_HIGH 12 /* clip(x, None, high = p1)               */
#define ALZ_MAP_CLIP_LOW 13 /* clip(x, low = p0, None)                 */
#define ALZ_MAP_ADD2 20     /* x + y                                   */
#define ALZ_MAP_SUB2 21     /* x - y                                   */
#define ALZ_MAP_MUL2 22     /* x * y                                   */
#define ALZ_MAP_DIV2 23     /* x / y                                   */
#define ALZ_MAP_ZERODIV 1
#define ALZ_MAP_DOMAIN 2
int alz_map_dev(int op, const double *x_dev, const double *y_dev, double p0, double p1, int64_t n,
                double *out_dev, int *flags_dev, int device, void *stream);
/* The same stage as part of a bank: op (0 = none, ALZ_MAP_ABS, ALZ_MAP_NEG or ALZ_MAP_SQUARE) is
 * applied to every input sample before the first section sees it -- ``filt(abs(sig))``, the shape of
 * envelope.abs (lazy_analysis.py:468-493).  For a bank of one biquad-class section ALZ_MAP_ABS is
 * fused into the kernels' input reads (no extra pass over the block); other shapes and ops run one
 * streaming pass first.  The input history of the bank then holds mapped samples. */
int alz_bank_set_input_map(alz_bank_t *h, int op);

/* ---- time-varying filters ------------------------------------------------------------------ */
/* One coefficient of a time-varying filter: a constant, or a series with one value per sample
 * (lazy_filters.py:202-204, 214-216: ``next(b_k) * d_k`` / ``-next(a_k) * m_k``). */
typedef struct alz_tv_tap {
  double value;             /* the constant (ignored when series_dev != NULL); 0 = term absent */
  const double *series_dev; /* coefficient value for sample n at series_dev[n*stride_n + c*stride_c] */
  int64_t stride_n;
  int64_t stride_c;         /* 0: one series shared by every channel */
  int64_t flags;            /* ALZ_TV_NEGATED: a denominator series that already holds -a_k[n] */
} alz_tv_tap_t;
/* The reference negates a denominator coefficient before the product (``-next(a_k) * m_k``).  For
 * a Python int 0 that is 0, not -0.0, and the sign of a zero product can surface in the output; a
 * host that wants the reference's result for integer-valued coefficient streams negates them
 * itself (in Python arithmetic) and sets this flag. */
#define ALZ_TV_NEGATED 1
/* LinearFilter.__call__ with Stream coefficients (lazy_filters.py:141-264) on one block of
 * `channels` independent streams: b[0..nb-1], a[0..na-1] (a[0] constant and non-zero: the
 * reference normalises a series a0 away first, :166-174); nb, na <= 17.  xh_dev / yh_dev hold the
 * input / output histories [k * channels + c] = x[-1-k] / y[-1-k] (nb-1 / na-1 rows) and are
 * updated in place, so consecutive blocks continue one stream; `zero` is the output of a filter
 * with no terms (:227-231).  Same layouts and pitches as alz_bank_process_dev. */
int alz_tv_process_dev(int nb, const alz_tv_tap_t *b, int na, const alz_tv_tap_t *a,
                       int64_t channels, const double *x_dev, double *y_dev, int64_t n, int layout,
                       int64_t ldx, int64_t ldy, double *xh_dev, double *yh_dev, double zero,
                       int device, void *stream);

/* ---- the optional downstream collective, on RCCL (no reference counterpart: the reference is one process,
 * audiolazy/lazy_stream.py:114; SURVEY.md 8e / BASELINE.json north_star: "a single RCCL gather over xGMI only when
 * a downstream mix needs every channel on one device").  Filtering itself never exchanges anything.  One process per
 * GPU; rank 0 calls alz_comm_unique_id and hands the 128 bytes to the other ranks through its launcher; every rank
 * then calls alz_comm_create (collective).  librccl.so is resolved at run time; without it these calls return
 * ALZ_E_UNSUPPORTED and nothing else in the library is affected. */
typedef struct alz_comm alz_comm_t;
#define ALZ_COMM_ID_BYTES 128
int alz_comm_unique_id(void *id_out /* ALZ_COMM_ID_BYTES */);
int alz_comm_create(int device, int world, int rank, const void *id /* ALZ_COMM_ID_BYTES */, alz_comm_t **out);
int alz_comm_destroy(alz_comm_t *c);
/* every rank contributes `count` doubles; rank `root` (every rank when root < 0) receives world * count doubles in
 * rank order: the [C / G, N] channel-major shards of a block become the [C, N] block.  Asynchronous on `stream`. */
int alz_comm_gather(alz_comm_t *c, const double *send_dev, double *recv_dev, int64_t count, int root, void *stream);
/* element-wise sum over the ranks of `count` doubles to `root` (to every rank when root < 0): the mix of per-rank
 * partial mixes (ParallelFilter / Streamix over shards).  The order of the additions is the collective's. */
int alz_comm_sum(alz_comm_t *c, const double *send_dev, double *recv_dev, int64_t count, int root, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ALZ_H */
