// alz_api.hip -- the C ABI of libalzhip.so (see include/alz.h): handles, state,
// host/device block entry points.  No CPU compute path exists in this library:
// every process call launches HIP kernels or fails.
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <new>
#include <tuple>

#include "alz_common.h"

namespace alz {

static thread_local std::string g_err;

static thread_local std::string g_kernel;
void set_error(const std::string &msg) { g_err = msg; }
void note_kernel(const std::string &name, bool append) {
  if (append && !g_kernel.empty()) g_kernel += " + " + name;
  else g_kernel = name;
}
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

int ensure_dynamic_lds(const void *fn, int bytes) {
  static std::mutex mu;
  static std::map<std::tuple<const void *, int>, int> done;   // (kernel, device) -> largest size set
  int dev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int &have = done[std::make_tuple(fn, dev)];
  if (have >= bytes) return ALZ_OK;
  ALZ_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
  return ALZ_OK;
}

}  // namespace alz

using alz::fail;

struct alz_bank {
  int device = 0;
  int64_t n_sets = 0, n_inputs = 0, channels = 0;
  int mode = 0;
  int n_sections = 0;
  std::vector<int> nb, na;
  std::vector<int64_t> hx_off, hy_off;  // per-section offsets into the host state rows
  int64_t thx = 0, thy = 0;             // sum(nb-1), sum(na-1)
  // device arrays, one allocation each
  double *b_dev = nullptr, *a_dev = nullptr;
  double *xh_dev = nullptr, *yh_dev = nullptr;  // 2x capacity (k_generic's new-state copy)
  std::vector<alz::SectionDev> sec;
  std::vector<alz::FirChains> fir_chains;   // per section (sized once: the sections point into it)
  double zero = 0.0;
  int fused = 0;
  int input_map = 0;                    // ALZ_MAP_ABS / NEG / SQUARE applied to every input sample, or 0
  double *map_in = nullptr;             // the mapped input block when the stage is not fused into a kernel
  uint64_t map_in_bytes = 0;
  double *expand_in = nullptr;          // OUTER bank with few inputs: the input with one column / row per channel
  uint64_t expand_in_bytes = 0;
  int64_t time_parallel = 0;            // 0 off (default), -1 automatic chunk length, > 0 chunk length
  // every section's input history is its predecessor's output history: true after reset and after every processed
  // block, and after a set_state whose rows say so (checked there) -- the dot-product zero-state pass of the fused
  // time-parallel cascade chains chunk 0 like every other chunk and needs it (alz_scan.hip)
  bool state_consistent = true;
  std::vector<alz::ScanScratch> scan;   // per section: chunk states and the cached transition matrices
  // a first section with a long numerator as TWO sections that compute the same doubles: its numerator alone (feedback-
  // free) and 1 / its denominator (b = [1]), on the section's own state slabs -- for the time-parallel mode (process_dev)
  bool has_split = false;
  alz::SectionDev split_fir, split_pole;
  double *ones_dev = nullptr;
  // staging for process_host and for out-of-place generic sections
  double *stage_x = nullptr, *stage_y = nullptr, *scratch = nullptr;
  uint64_t stage_x_bytes = 0, stage_y_bytes = 0, scratch_bytes = 0;
  const char *last_kernel = "";
  std::string last_kernels;
  // process_host pipeline: copy-in / compute / copy-out streams and their hand-over events
  hipStream_t host_streams[3] = {nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> host_events;
  // section pipeline of narrow cascades (process_dev): one stream per section, events per (section, chunk)
  hipStream_t sec_streams[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> sec_events;
  // the one-pass time-parallel kernel (alz_look.hip) and its bounded waits -- alz_bank_set_look_check
  int look_check = ALZ_LOOK_CHECK_CALL;   // verify on the call that launched it (and re-run the block) / report at the next entry point
  bool look_launched = false;             // this process call launched the one-pass kernel
  bool no_one_pass = false;               // the re-run of a block on which it gave up: three-launch form
  double *state_snap = nullptr;           // the bank's state before the call (xh_dev then yh_dev), for that re-run
  uint64_t state_snap_bytes = 0;
  int64_t look_launches = 0, look_gave_up = 0, look_reruns = 0;
  unsigned look_last_sites = 0;           // bit W_*: the waits that ran out in the last launch that gave up
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int grow(double **ptr, uint64_t *have, uint64_t need) {
  if (*have >= need) return ALZ_OK;
  if (*ptr) (void)hipFree(*ptr);
  *ptr = nullptr;
  *have = 0;
  if (hipMalloc((void **)ptr, need) != hipSuccess)
    return fail(ALZ_E_NOMEM, "hipMalloc failed for " + std::to_string(need) + " bytes");
  *have = need;
  return ALZ_OK;
}

// host [rows = channels][cols] (row-major)  <->  device [cols][channels]
void transpose_to_dev_order(const double *src, std::vector<double> &dst, int64_t rows, int64_t cols) {
  dst.resize((size_t)(rows * cols));
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t k = 0; k < cols; ++k) dst[(size_t)(k * rows + r)] = src[r * cols + k];
}

// two arrays in one small launch (the bank's state kept / put back around a one-pass time-parallel launch)
__global__ __launch_bounds__(256) void k_state_copy(double *d1, const double *s1, int64_t n1, double *d2, const double *s2, int64_t n2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n1) d1[i] = s1[i];
  else if (i - n1 < n2) d2[i - n1] = s2[i - n1];
}

}  // namespace

extern "C" {

int alz_version(void) { return ALZ_VERSION; }

const char *alz_last_error(void) { return alz::g_err.c_str(); }

const char *alz_last_kernel(void) { return alz::g_kernel.c_str(); }

int alz_device_count(int *count) {
  if (!count) return fail(ALZ_E_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(ALZ_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return ALZ_OK;
}

int alz_malloc(int device, uint64_t bytes, void **dev_ptr) {
  if (!dev_ptr) return fail(ALZ_E_ARG, "dev_ptr is NULL");
  DeviceGuard g(device);
  if (!g.ok) return fail(ALZ_E_HIP, "hipSetDevice failed");
  if (hipMalloc(dev_ptr, bytes ? bytes : 8) != hipSuccess) return fail(ALZ_E_NOMEM, "hipMalloc failed");
  return ALZ_OK;
}

int alz_free(int device, void *dev_ptr) {
  DeviceGuard g(device);
  if (!g.ok) return fail(ALZ_E_HIP, "hipSetDevice failed");
  ALZ_HIP_CHECK(hipFree(dev_ptr));
  return ALZ_OK;
}

int alz_memcpy_h2d(int device, void *dst_dev, const void *src_host, uint64_t bytes) {
  DeviceGuard g(device);
  if (!g.ok) return fail(ALZ_E_HIP, "hipSetDevice failed");
  ALZ_HIP_CHECK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return ALZ_OK;
}

int alz_memcpy_d2h(int device, void *dst_host, const void *src_dev, uint64_t bytes) {
  DeviceGuard g(device);
  if (!g.ok) return fail(ALZ_E_HIP, "hipSetDevice failed");
  ALZ_HIP_CHECK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return ALZ_OK;
}

int alz_device_sync(int device) {
  DeviceGuard g(device);
  if (!g.ok) return fail(ALZ_E_HIP, "hipSetDevice failed");
  ALZ_HIP_CHECK(hipDeviceSynchronize());
  return ALZ_OK;
}

int alz_bank_create(int64_t n_sets, int64_t n_inputs, int mode, int n_sections, const int *nb,
                    const int *na, const double *b_host, const double *a_host, int device,
                    alz_bank_t **out) {
  if (!out) return fail(ALZ_E_ARG, "out is NULL");
  *out = nullptr;
  if (n_sets < 1 || n_inputs < 1 || n_sections < 1 || !nb || !na || !b_host || !a_host)
    return fail(ALZ_E_ARG, "alz_bank_create: bad sizes or NULL arrays");
  if (mode != ALZ_BANK_DIAGONAL && mode != ALZ_BANK_OUTER) return fail(ALZ_E_ARG, "bad bank mode");
  if (mode == ALZ_BANK_DIAGONAL && n_sets != n_inputs && n_sets != 1)
    return fail(ALZ_E_ARG, "DIAGONAL bank needs n_sets == n_inputs (or 1 shared set)");
  int64_t tb = 0, ta = 0;
  for (int s = 0; s < n_sections; ++s) {
    if (nb[s] < 1 || na[s] < 1) return fail(ALZ_E_ARG, "every section needs nb >= 1 and na >= 1");
    tb += nb[s];
    ta += na[s];
  }
  // a0 == 0 -> ZeroDivisionError("Invalid filter gain"), lazy_filters.py:177-178
  {
    int64_t off = 0;
    for (int s = 0; s < n_sections; ++s) {
      for (int64_t q = 0; q < n_sets; ++q)
        if (a_host[q * ta + off] == 0.0) return fail(ALZ_E_ZERO_GAIN, "Invalid filter gain");
      off += na[s];
    }
  }

  alz_bank *h = new (std::nothrow) alz_bank();
  if (!h) return fail(ALZ_E_NOMEM, "out of host memory");
  h->device = device;
  h->n_sets = n_sets;
  h->n_inputs = n_inputs;
  h->mode = mode;
  h->channels = (mode == ALZ_BANK_OUTER) ? n_sets * n_inputs : n_inputs;
  h->n_sections = n_sections;
  h->nb.assign(nb, nb + n_sections);
  h->na.assign(na, na + n_sections);
  for (int s = 0; s < n_sections; ++s) {
    h->hx_off.push_back(h->thx);
    h->hy_off.push_back(h->thy);
    h->thx += nb[s] - 1;
    h->thy += na[s] - 1;
  }

  DeviceGuard g(device);
  if (!g.ok) {
    delete h;
    return fail(ALZ_E_HIP, "hipSetDevice failed");
  }
  // coefficients, tap-major on the device: section s, tap k, set q at
  //   b_dev[(boff(s) + k) * n_sets + q]
  std::vector<double> bt, at;
  transpose_to_dev_order(b_host, bt, n_sets, tb);
  transpose_to_dev_order(a_host, at, n_sets, ta);
  bt.resize(bt.size() + 16, 0.0);   // k_fir_ring reads whole 8-tap blocks with scalar loads: keep them in bounds
  const uint64_t xs = (uint64_t)(h->thx > 0 ? h->thx : 1) * h->channels * 8 * 2;
  const uint64_t ys = (uint64_t)(h->thy > 0 ? h->thy : 1) * h->channels * 8 * 2;
  if (hipMalloc((void **)&h->b_dev, bt.size() * 8) != hipSuccess ||
      hipMalloc((void **)&h->a_dev, at.size() * 8) != hipSuccess ||
      hipMalloc((void **)&h->xh_dev, xs) != hipSuccess ||
      hipMalloc((void **)&h->yh_dev, ys) != hipSuccess) {
    alz_bank_destroy(h);
    return fail(ALZ_E_NOMEM, "hipMalloc failed while creating the bank");
  }
  if (hipMemcpy(h->b_dev, bt.data(), bt.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->a_dev, at.data(), at.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(h->xh_dev, 0, xs) != hipSuccess || hipMemset(h->yh_dev, 0, ys) != hipSuccess) {
    alz_bank_destroy(h);
    return fail(ALZ_E_HIP, "coefficient upload failed");
  }

  // Section descriptors.  Each section's state occupies a [taps-1][channels]
  // slab (x2 for k_generic's new-state copy), slabs laid out back to back.
  int64_t boff = 0, aoff = 0;
  h->fir_chains.resize((size_t)n_sections);
  for (int s = 0; s < n_sections; ++s) {
    alz::SectionDev d;
    d.chains = &h->fir_chains[(size_t)s];
    d.nb = nb[s];
    d.na = na[s];
    d.b = h->b_dev + boff * n_sets;
    d.a = h->a_dev + aoff * n_sets;
    d.xh = h->xh_dev + 2 * h->hx_off[s] * h->channels;
    d.yh = h->yh_dev + 2 * h->hy_off[s] * h->channels;
    d.present_b = d.present_a = 0;
    d.uniform = true;
    d.any_div = false;
    d.shared_sets = n_sets == 1;
    for (int k = 0; k < nb[s]; ++k) {
      int64_t nz = 0;
      for (int64_t q = 0; q < n_sets; ++q) nz += b_host[q * tb + boff + k] != 0.0;
      if (nz && k < 32) d.present_b |= 1u << k;
      if (nz && nz != n_sets) d.uniform = false;
      if (nz && k >= 32) d.present_b |= 0x80000000u;
    }
    for (int k = 1; k < na[s]; ++k) {
      int64_t nz = 0;
      for (int64_t q = 0; q < n_sets; ++q) nz += a_host[q * ta + aoff + k] != 0.0;
      if (nz && k <= 32) d.present_a |= 1u << (k - 1);
      if (nz && nz != n_sets) d.uniform = false;
      if (nz && k > 32) d.present_a |= 0x80000000u;
    }
    for (int64_t q = 0; q < n_sets; ++q) d.any_div |= a_host[q * ta + aoff] != 1.0;
    d.n_ff = d.n_fb = 0;
    for (int j = 0; j < 8; ++j) d.tap_b[j] = d.tap_a[j] = 0;
    for (int k = 0; k < nb[s]; ++k) {
      bool nz = false;
      for (int64_t q = 0; q < n_sets && !nz; ++q) nz = b_host[q * tb + boff + k] != 0.0;
      if (nz) {
        if (d.n_ff >= 0 && d.n_ff < 8) d.tap_b[d.n_ff++] = k;
        else d.n_ff = -1;
      }
    }
    for (int k = 1; k < na[s]; ++k) {
      bool nz = false;
      for (int64_t q = 0; q < n_sets && !nz; ++q) nz = a_host[q * ta + aoff + k] != 0.0;
      if (nz) {
        if (d.n_fb >= 0 && d.n_fb < 8) d.tap_a[d.n_fb++] = k;
        else d.n_fb = -1;
      }
    }
    h->sec.push_back(d);
    boff += nb[s];
    aoff += na[s];
  }
  h->scan.resize((size_t)n_sections + 2);   // [n_sections]: the fused time-parallel cascade's own scratch; [n_sections + 1]: the same for the split form
  {
    // y = ((sum_k b_k x[n-k]) + (-a1) y1) + (-a2) y2 is w = sum_k b_k x[n-k] followed by y = (1 w + (-a1) y1) + (-a2) y2:
    // 1 * w is w, every other operation is the same -- the two sections give the section's own doubles
    const alz::SectionDev &s0 = h->sec[0];
    if (n_sections >= 2 && n_sections <= 4 && s0.nb > 3 && s0.na == 3 && !s0.any_div && s0.uniform && s0.present_a == 3u &&
        hipMalloc((void **)&h->ones_dev, (size_t)n_sets * 8) == hipSuccess) {
      std::vector<double> ones((size_t)n_sets, 1.0);
      if (hipMemcpy(h->ones_dev, ones.data(), ones.size() * 8, hipMemcpyHostToDevice) == hipSuccess) {
        h->split_fir = s0;
        h->split_fir.na = 1; h->split_fir.a = h->ones_dev; h->split_fir.present_a = 0; h->split_fir.n_fb = 0;
        h->split_pole = s0;
        h->split_pole.nb = 1; h->split_pole.b = h->ones_dev; h->split_pole.present_b = 1u;
        h->split_pole.n_ff = 1; h->split_pole.tap_b[0] = 0;
        h->has_split = true;
      }
    }
    (void)hipGetLastError();
  }
  *out = h;
  return ALZ_OK;
}

int alz_bank_destroy(alz_bank_t *h) {
  if (!h) return ALZ_OK;
  DeviceGuard g(h->device);
  if (h->b_dev) (void)hipFree(h->b_dev);
  if (h->ones_dev) (void)hipFree(h->ones_dev);
  if (h->a_dev) (void)hipFree(h->a_dev);
  if (h->xh_dev) (void)hipFree(h->xh_dev);
  if (h->yh_dev) (void)hipFree(h->yh_dev);
  if (h->stage_x) (void)hipFree(h->stage_x);
  if (h->stage_y) (void)hipFree(h->stage_y);
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->map_in) (void)hipFree(h->map_in);
  if (h->expand_in) (void)hipFree(h->expand_in);
  if (h->state_snap) (void)hipFree(h->state_snap);
  for (alz::FirChains &fc : h->fir_chains)
    if (fc.flags) (void)hipFree(fc.flags);
  for (hipStream_t st : h->host_streams)
    if (st) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : h->host_events) (void)hipEventDestroy(ev);
  for (hipStream_t st : h->sec_streams)
    if (st) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : h->sec_events) (void)hipEventDestroy(ev);
  for (alz::ScanScratch &sc : h->scan) {
    if (sc.vxh) (void)hipFree(sc.vxh);
    if (sc.vyh) (void)hipFree(sc.vyh);
    if (sc.power) (void)hipFree(sc.power);
    if (sc.zbuf) (void)hipFree(sc.zbuf);
    if (sc.hr) (void)hipFree(sc.hr);
    if (sc.edge) (void)hipFree(sc.edge);
    if (sc.look_err) (void)hipHostFree(sc.look_err);
  }
  delete h;
  return ALZ_OK;
}

int alz_bank_channels(const alz_bank_t *h, int64_t *channels) {
  if (!h || !channels) return fail(ALZ_E_ARG, "NULL argument");
  *channels = h->channels;
  return ALZ_OK;
}

static int put_state(alz_bank *h, const double *xh_host, const double *yh_host) {
  // host rows [channels][thx] -> per-section device slabs [nb-1][channels]
  const int64_t C = h->channels;
  // self-consistent: x history of section s + 1 == y history of section s, as far as both are kept (a partial
  // update -- one of the two arrays missing -- cannot be checked against what the device holds: not consistent)
  h->state_consistent = true;
  for (int s = 0; s + 1 < h->n_sections && h->state_consistent; ++s) {
    const int64_t kx = h->nb[s + 1] - 1, ky = h->na[s] - 1;
    if (kx <= 0) continue;
    if (!xh_host || !yh_host || ky < kx) { h->state_consistent = false; break; }
    for (int64_t c = 0; c < C && h->state_consistent; ++c)
      for (int64_t k = 0; k < kx; ++k) {
        const double xv = xh_host[c * h->thx + h->hx_off[s + 1] + k], yv = yh_host[c * h->thy + h->hy_off[s] + k];
        if (memcmp(&xv, &yv, sizeof(double)) != 0) { h->state_consistent = false; break; }
      }
  }
  std::vector<double> tmp;
  for (int s = 0; s < h->n_sections; ++s) {
    const int64_t kx = h->nb[s] - 1, ky = h->na[s] - 1;
    if (kx > 0 && xh_host) {
      tmp.resize((size_t)(kx * C));
      for (int64_t c = 0; c < C; ++c)
        for (int64_t k = 0; k < kx; ++k) tmp[(size_t)(k * C + c)] = xh_host[c * h->thx + h->hx_off[s] + k];
      ALZ_HIP_CHECK(hipMemcpy(h->sec[s].xh, tmp.data(), tmp.size() * 8, hipMemcpyHostToDevice));
    }
    if (ky > 0 && yh_host) {
      tmp.resize((size_t)(ky * C));
      for (int64_t c = 0; c < C; ++c)
        for (int64_t k = 0; k < ky; ++k) tmp[(size_t)(k * C + c)] = yh_host[c * h->thy + h->hy_off[s] + k];
      ALZ_HIP_CHECK(hipMemcpy(h->sec[s].yh, tmp.data(), tmp.size() * 8, hipMemcpyHostToDevice));
    }
  }
  return ALZ_OK;
}

int alz_bank_reset(alz_bank_t *h, double zero) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  DeviceGuard g(h->device);
  ALZ_HIP_CHECK(hipDeviceSynchronize());
  h->zero = zero;
  std::vector<double> xh((size_t)(h->channels * (h->thx > 0 ? h->thx : 1)), zero);
  std::vector<double> yh((size_t)(h->channels * (h->thy > 0 ? h->thy : 1)), zero);
  return put_state(h, xh.data(), yh.data());
}

int alz_bank_set_state(alz_bank_t *h, const double *xh_host, const double *yh_host) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  DeviceGuard g(h->device);
  ALZ_HIP_CHECK(hipDeviceSynchronize());
  return put_state(h, xh_host, yh_host);
}

// The one-pass time-parallel kernel (alz_look.hip) reports a wait that ran out in a word of pinned host memory per
// section.  Every entry point that hands results to the caller looks at it -- after its synchronisation where it has
// one -- so a bad block is never delivered silently (round-3 review: only the NEXT one-pass call used to look).
static unsigned take_look_sites(alz_bank_t *h) {
  unsigned sites = 0;
  for (alz::ScanScratch &sc : h->scan)
    if (sc.look_err)
      for (int k = 0; k < alz::kLookErrWords; ++k)
        if (((volatile int *)sc.look_err)[k] != 0) {
          sc.look_err[k] = 0;
          sites |= 1u << k;
        }
  if (sites) {
    h->look_gave_up += 1;
    h->look_last_sites = sites;
  }
  return sites;
}
static std::string look_site_names(unsigned sites) {
  std::string out;
  for (int k = 0; k < alz::kLookErrWords; ++k)
    if (sites >> k & 1u) out += (out.empty() ? "" : ", ") + std::string(alz::look_wait_name(k));
  return out;
}
static int take_look_error(alz_bank_t *h) {
  const unsigned sites = take_look_sites(h);
  // (the kernel had advanced the bank's state by then: the block cannot simply be processed again)
  return sites ? fail(ALZ_E_HIP, "time-parallel mode: the one-pass kernel of an EARLIER process call on this bank gave up "
                                 "waiting for another workgroup (waits that ran out: " + look_site_names(sites) + "): the block "
                                 "that call wrote and the bank's state are invalid, "
                                 "and the call that reports this has processed nothing -- reset() or set_state() the bank, "
                                 "then process from the last good block again")
               : ALZ_OK;
}

int alz_bank_get_state(alz_bank_t *h, double *xh_host, double *yh_host) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  DeviceGuard g(h->device);
  ALZ_HIP_CHECK(hipDeviceSynchronize());
  if (const int lrc = take_look_error(h)) return lrc;
  const int64_t C = h->channels;
  std::vector<double> tmp;
  for (int s = 0; s < h->n_sections; ++s) {
    const int64_t kx = h->nb[s] - 1, ky = h->na[s] - 1;
    if (kx > 0 && xh_host) {
      tmp.resize((size_t)(kx * C));
      ALZ_HIP_CHECK(hipMemcpy(tmp.data(), h->sec[s].xh, tmp.size() * 8, hipMemcpyDeviceToHost));
      for (int64_t c = 0; c < C; ++c)
        for (int64_t k = 0; k < kx; ++k) xh_host[c * h->thx + h->hx_off[s] + k] = tmp[(size_t)(k * C + c)];
    }
    if (ky > 0 && yh_host) {
      tmp.resize((size_t)(ky * C));
      ALZ_HIP_CHECK(hipMemcpy(tmp.data(), h->sec[s].yh, tmp.size() * 8, hipMemcpyDeviceToHost));
      for (int64_t c = 0; c < C; ++c)
        for (int64_t k = 0; k < ky; ++k) yh_host[c * h->thy + h->hy_off[s] + k] = tmp[(size_t)(k * C + c)];
    }
  }
  return ALZ_OK;
}

static int process_dev_impl(alz_bank_t *h, const double *x_dev, double *y_dev, int64_t n, int layout,
                            int64_t ldx, int64_t ldy, void *stream);

int alz_bank_process_dev(alz_bank_t *h, const double *x_dev, double *y_dev, int64_t n, int layout,
                         int64_t ldx, int64_t ldy, void *stream) {
  // The one-pass time-parallel kernel (alz_look.hip) is the one kernel of the library whose workgroups wait for each
  // other; its waits are bounded, and a launch that gave up (its workgroups were not all resident in time: foreign work
  // on the device) has written a bad block and a bad state.  ALZ_LOOK_CHECK_CALL (default): this call notices -- it keeps
  // a copy of the bank's state, waits for its own work on `stream`, and on a give-up restores the state and processes the
  // block again with the three-launch form of the same mode (same numerics), so the caller never sees it; an in-place
  // block cannot be processed again (its input is gone): the call FAILS, with the state as it was before the call.
  bool snap = false;
  int64_t nx = 0, ny = 0;
  if (h && h->look_check == ALZ_LOOK_CHECK_CALL && (h->time_parallel == ALZ_TP_AUTO || h->time_parallel == ALZ_TP_ONE_PASS) &&
      x_dev && y_dev && n >= 4 * alz::kLookChunk && h->channels % 16 == 0) {
    DeviceGuard g(h->device);
    nx = (h->thx > 0 ? h->thx : 1) * h->channels * 2;
    ny = (h->thy > 0 ? h->thy : 1) * h->channels * 2;
    if (g.ok && grow(&h->state_snap, &h->state_snap_bytes, (uint64_t)(nx + ny) * 8) == ALZ_OK) {
      hipLaunchKernelGGL(k_state_copy, dim3((unsigned)((nx + ny + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->state_snap,
                         h->xh_dev, nx, h->state_snap + nx, h->yh_dev, ny);
      snap = hipGetLastError() == hipSuccess;
    }
  }
  if (h) h->look_launched = false;
  int rc = process_dev_impl(h, x_dev, y_dev, n, layout, ldx, ldy, stream);
  if (rc == ALZ_OK && h->look_launched && h->look_check == ALZ_LOOK_CHECK_CALL) {
    DeviceGuard g(h->device);
    ALZ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const unsigned sites = take_look_sites(h);
    if (sites) {
      const std::string what = "time-parallel mode: the one-pass kernel gave up waiting for another workgroup (waits that ran out: " +
                               look_site_names(sites) + ")";
      if (!snap) return fail(ALZ_E_HIP, what + " and no copy of the bank's state could be kept: the block and the bank's state are invalid");
      hipLaunchKernelGGL(k_state_copy, dim3((unsigned)((nx + ny + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->xh_dev,
                         h->state_snap, nx, h->yh_dev, h->state_snap + nx, ny);
      ALZ_HIP_CHECK(hipGetLastError());
      if (x_dev == y_dev)
        return fail(ALZ_E_HIP, what + " on an IN-PLACE block: its contents are invalid and its input is gone; the bank's state is "
                               "what it was before this call -- process the block again from a copy of its input");
      const std::string first = h->last_kernels;
      h->no_one_pass = true;
      rc = process_dev_impl(h, x_dev, y_dev, n, layout, ldx, ldy, stream);
      h->no_one_pass = false;
      h->look_reruns += 1;
      h->last_kernels = first + "  [gave up: " + look_site_names(sites) + "; block processed again:] " + h->last_kernels;
    }
  }
  // two samples through every section make each section's input history its predecessor's output history
  if (rc == ALZ_OK && h && n >= 2) h->state_consistent = true;
  return rc;
}

static int process_dev_impl(alz_bank_t *h, const double *x_dev, double *y_dev, int64_t n, int layout,
                            int64_t ldx, int64_t ldy, void *stream) {
  if (!h || !x_dev || !y_dev) return fail(ALZ_E_ARG, "NULL argument");
  if (n < 0) return fail(ALZ_E_ARG, "negative block length");
  if (layout != ALZ_TIME_MAJOR && layout != ALZ_CHAN_MAJOR) return fail(ALZ_E_ARG, "bad layout");
  if (layout == ALZ_TIME_MAJOR && (ldx < h->n_inputs || ldy < h->channels))
    return fail(ALZ_E_ARG, "TIME_MAJOR leading dimension smaller than the channel count");
  if (layout == ALZ_CHAN_MAJOR && (ldx < n || ldy < n))
    return fail(ALZ_E_ARG, "CHAN_MAJOR leading dimension smaller than the block length");
  if (x_dev == y_dev && (h->mode != ALZ_BANK_DIAGONAL || ldx != ldy))
    return fail(ALZ_E_ARG, "in-place processing needs DIAGONAL mode and ldx == ldy");
  h->last_kernels.clear();
  h->last_kernel = "";
  if (const int lrc = take_look_error(h)) return lrc;      // (an EARLIER block of this handle: read without a sync)
  if (n == 0) return ALZ_OK;
  DeviceGuard g(h->device);
  if (!g.ok) return fail(ALZ_E_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  // (the re-run of a block on which the one-pass kernel gave up: the same mode without that form)
  const int64_t tp_mode = (h->no_one_pass && h->time_parallel < 0) ? alz::kTpThreeLaunch : h->time_parallel;

  // the elementwise input stage: |x| rides on the section's own input reads where the kernels can
  // do that for free (one biquad-class section, the streaming kernels and k_small); anything else
  // gets the mapped block from one streaming pass (16 B/sample more traffic)
  int pre_fused = 0;
  bool map_pass = false;
  if (h->input_map) {
    const alz::SectionDev &s0 = h->sec[0];
    bool fusable = h->input_map == ALZ_MAP_ABS && h->n_sections == 1 && s0.nb <= 3 && s0.na <= 3 &&
                   s0.uniform && (s0.present_b | s0.present_a) != 0 && !h->fused;
    if (fusable && h->time_parallel != 0) {
      // time-parallel mode: only the one-pass kernel reads through the map (alz_look.hip); where the three-launch form
      // will run, the mapped block comes from a streaming pass as before
      alz::BlockIO probe;
      probe.n = n; probe.pre_op = ALZ_MAP_ABS; probe.channels = h->channels; probe.n_inputs = h->n_inputs; probe.n_sets = h->n_sets;
      probe.mode = h->mode; probe.zero = h->zero; probe.fused = h->fused; probe.stream_once = 0;
      probe.look_sync = h->look_check == ALZ_LOOK_CHECK_CALL ? 1 : 0;
      probe.x = x_dev; probe.y = y_dev;
      probe.sxn = layout == ALZ_TIME_MAJOR ? ldx : 1; probe.sxc = layout == ALZ_TIME_MAJOR ? 1 : ldx;
      probe.syn = layout == ALZ_TIME_MAJOR ? ldy : 1; probe.syc = layout == ALZ_TIME_MAJOR ? 1 : ldy;
      probe.map_input = (h->mode == ALZ_BANK_OUTER && (h->n_inputs % 64) == 0) ? 1 : 0;
      probe.c_first = 0; probe.c_count = h->channels;
      fusable = h->mode != ALZ_BANK_OUTER && alz::scan_takes_one_pass(s0, probe, tp_mode);
    }
    if (fusable) {
      pre_fused = ALZ_MAP_ABS;
    } else {
      const uint64_t extent = layout == ALZ_TIME_MAJOR ? (uint64_t)((n - 1) * ldx + h->n_inputs)
                                                       : (uint64_t)((h->n_inputs - 1) * ldx + n);
      int rc = grow(&h->map_in, &h->map_in_bytes, extent * 8);
      if (rc) return rc;
      rc = alz::launch_map(h->input_map, x_dev, nullptr, 0.0, 0.0, (int64_t)extent, h->map_in, nullptr, st);
      if (rc) return rc;
      x_dev = h->map_in;
      map_pass = true;
    }
  }

  alz::BlockIO io;
  io.n = n;
  io.pre_op = 0;
  io.channels = h->channels;
  io.n_inputs = h->n_inputs;
  io.n_sets = h->n_sets;
  io.mode = h->mode;
  io.zero = h->zero;
  io.fused = h->fused;
  io.look_sync = h->look_check == ALZ_LOOK_CHECK_CALL ? 1 : 0;
  // (blocks processed in place are not `stream_once` here; k_duo's launcher gives them non-temporal tiles itself where that pays:
  // alz_wave.hip, nt_in_place)
  io.stream_once = (h->n_sections == 1 && x_dev != y_dev && (uint64_t)n * (uint64_t)h->channels * 8u >= (256ull << 20)) ? 1 : 0;
  int64_t sxn = layout == ALZ_TIME_MAJOR ? ldx : 1, sxc = layout == ALZ_TIME_MAJOR ? 1 : ldx;
  const int64_t syn = layout == ALZ_TIME_MAJOR ? ldy : 1, syc = layout == ALZ_TIME_MAJOR ? 1 : ldy;
  std::string last_noted;
  auto note = [&](const char *k) {
    h->last_kernel = k;
    if (last_noted == k && h->last_kernels.size() > 96) return;   // (a chunked run repeats its kernels: keep the record short)
    last_noted = k;
    if (h->last_kernels.size() > 400) return;
    if (!h->last_kernels.empty()) h->last_kernels += "+";
    h->last_kernels += k;
  };
  if (map_pass) note("k_map");          // (the input map as a streaming pass of its own)

  // opt-in time-parallel mode, whole fused cascade at once (channel-major blocks): the chunks of the time axis
  // become the cascade kernel's channels, read straight from the un-expanded input (alz_scan.hip)
  // (time-major blocks -- the reference's vector-valued samples -- since round 5: their chunks are groups of 64
  // adjacent channels, so the bank's channels must be a multiple of 64; launch_scan_cascade decides)
  if (h->time_parallel != 0 && h->n_sections >= 2 && x_dev != y_dev) {
    io.n = n;
    io.x = x_dev; io.y = y_dev;
    io.sxn = sxn; io.sxc = sxc; io.syn = syn; io.syc = syc;
    io.map_input = h->mode == ALZ_BANK_OUTER;
    io.c_first = 0; io.c_count = h->channels;
    bool taken = false;
    const char *name = "";
    const int rc = alz::launch_scan_cascade(h->sec.data(), h->n_sections, io, st, h->time_parallel < 0 ? 0 : h->time_parallel,
                                            &h->scan[(size_t)h->n_sections], h->state_consistent, &taken, &name);
    if (rc) return rc;
    if (taken) {
      h->state_consistent = true;
      note(name);
      return ALZ_OK;
    }
  }

  // An OUTER bank whose inputs are fewer than a workgroup's channels (a gammatone bank on one stream:
  // every band reads the same samples) cannot feed the streaming / pipeline kernels by input index.
  // Give it one input column (row) per channel first: one streaming pass the size of the output, after
  // which every kernel reads it like a diagonal bank's input.
  int outer_by_input = h->mode == ALZ_BANK_OUTER;
  if (outer_by_input && (h->n_inputs % 64) != 0 && h->channels >= 64 && x_dev != y_dev) {
    const int64_t lde = layout == ALZ_TIME_MAJOR ? ((h->channels + 1) & ~(int64_t)1) : ((n + 1) & ~(int64_t)1);
    const uint64_t extent = layout == ALZ_TIME_MAJOR ? (uint64_t)n * lde : (uint64_t)h->channels * lde;
    // (a block as large as the output; when the device cannot spare it the bank simply keeps reading by
    // input index through the lane-per-channel kernels: slower, same doubles)
    if (grow(&h->expand_in, &h->expand_in_bytes, extent * 8) == ALZ_OK) {
      const int64_t sen = layout == ALZ_TIME_MAJOR ? lde : 1, sec = layout == ALZ_TIME_MAJOR ? 1 : lde;
      const int rc = alz::launch_expand(x_dev, h->expand_in, n, h->channels, h->n_inputs, sxn, sxc, sen, sec, st);
      if (rc) return rc;
      x_dev = h->expand_in;
      sxn = sen;
      sxc = sec;
      outer_by_input = 0;
    } else {
      (void)hipGetLastError();
    }
  }
  // extent of y in elements, for the out-of-place copy some sections need
  const uint64_t y_extent = layout == ALZ_TIME_MAJOR ? (uint64_t)((n - 1) * ldy + h->channels)
                                                      : (uint64_t)((h->channels - 1) * ldy + n);

  // Section by section over channels [c_first, c_first + c_count) and samples
  // [t_first, t_first + t_count): the streaming kernel takes what it can (only when the range
  // is the whole bank), the lane-per-channel kernels finish the ragged rest.
  auto run_sections_on = [&](int64_t c_first, int64_t c_count, int64_t t_first, int64_t t_count, int s_lo, int s_hi,
                             hipStream_t st) -> int {
    const bool whole = c_first == 0 && c_count == h->channels;
    for (int s = s_lo; s < s_hi; ++s) {
      const alz::SectionDev &sec = h->sec[s];
      const bool generic = !(sec.nb <= 16 && sec.na <= 9);
      io.n = t_count;
      io.y = y_dev + t_first * syn;
      io.syn = syn;
      io.syc = syc;
      if (s == 0) {
        io.x = x_dev + t_first * sxn;
        io.sxn = sxn;
        io.sxc = sxc;
        io.map_input = outer_by_input;
        io.pre_op = pre_fused;
      } else {
        io.x = io.y;
        io.sxn = syn;
        io.sxc = syc;
        io.map_input = 0;
        io.pre_op = 0;
      }
      io.c_first = c_first;
      io.c_count = c_count;
      if (generic && io.x == io.y && !alz::comb_takes_in_place(sec, io)) {
        // k_fir / k_generic read their history from the block, so they cannot overwrite it
        int rc = grow(&h->scratch, &h->scratch_bytes, y_extent * 8);
        if (rc) return rc;
        ALZ_HIP_CHECK(hipMemcpyAsync(h->scratch, y_dev, y_extent * 8, hipMemcpyDeviceToDevice, st));
        io.x = h->scratch + t_first * syn;
      }
      int64_t done_n = 0, done_c = 0;
      const char *name = "";
      io.c_first = c_first;
      io.c_count = c_count;
      int rc = ALZ_OK;
      if (h->time_parallel != 0 && whole && !generic) {
        // opt-in time-parallel mode: whole chunks of the whole bank; the ragged rest continues
        // serially from the state the replay pass left
        rc = alz::launch_scan(sec, s, io, st, tp_mode, &h->scan[(size_t)s],
                              &done_n, &name);
        if (rc) return rc;
        if (done_n > 0) done_c = c_count;
        if (done_n > 0 && strcmp(name, "k_scan(k_look)") == 0) {
          h->look_launched = true;
          h->look_launches += 1;
        }
      }
      if (done_c == 0 && whole) rc = alz::launch_mid(sec, io, st, &done_n, &done_c, &name);
      if (rc) return rc;
      if (done_c == 0 && !generic && whole) rc = alz::launch_wave(sec, io, st, &done_n, &done_c, &name);
      if (rc) return rc;
      if (done_c > 0) note(name);
      if (done_c < c_count) {  // channels the streaming kernel did not take, whole time range
        io.c_first = c_first + done_c;
        io.c_count = c_count - done_c;
        rc = alz::launch_section(sec, io, st, &name);
        if (rc) return rc;
        note(name);
      }
      if (done_c > 0 && done_n < t_count) {  // ragged time tail of the streamed channels
        io.c_first = c_first;
        io.c_count = done_c;
        io.x += done_n * io.sxn;
        io.y += done_n * io.syn;
        io.n = t_count - done_n;
        rc = alz::launch_section(sec, io, st, &name);
        if (rc) return rc;
        note(name);
      }
    }
    return ALZ_OK;
  };
  auto run_sections = [&](int64_t c_first, int64_t c_count, int64_t t_first, int64_t t_count) -> int {
    return run_sections_on(c_first, c_count, t_first, t_count, 0, h->n_sections, st);
  };

  // Narrow cascades: section pipeline over chunks of the time axis.  A cascade on a few hundred channels is a
  // handful of fused-cascade workgroups on a 256-CU chip, each advancing one sample per ~60 ns (256 bands on one
  // signal, the reference's own filterbank shape: 4 workgroups, 4.2 Gsamples/s), and run section after section
  // it is four serial passes.  Here section s works on chunk j while section s + 1 works on chunk j - 1, each
  // section on its own stream with the two-wave streaming kernel (16 channels per workgroup, the recurrence wave
  // at its ~30-cycle step): the sections overlap, every sample is still produced by the same kernels in the same
  // order -- bit-exact -- and the intermediate chunks are re-read from L2 / Infinity Cache.  The block costs
  // (chunks + sections - 1) chunk times instead of sections x chunks.
  auto pipelined = [&]() -> int {
    if (h->n_sections < 2 || h->n_sections > 4 || h->time_parallel != 0 || x_dev == y_dev) return 0;
    if (h->channels % 16 != 0 || h->channels >= ALZ_TUNE("ALZ_SECPIPE_MAX", 2048)) return 0;
    for (int s = 0; s < h->n_sections; ++s) {
      const alz::SectionDev &sec = h->sec[s];
      if (!(sec.nb <= 3 && sec.na <= 3 && sec.uniform) || (sec.present_b | sec.present_a) == 0) return 0;
    }
    if (outer_by_input && (h->n_inputs % 16) != 0) return 0;   // (the streaming kernel reads whole 16-input groups)
    int64_t chunk = ALZ_TUNE("ALZ_SECPIPE_CHUNK", 16384);
    chunk = chunk / 64 * 64;
    if (chunk < 64 || n < 4 * chunk) return 0;
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    const int ns = h->n_sections;
    // the LAST section runs on the caller's stream, the others on streams of the handle: a process has four hardware
    // queues by default (GPU_MAX_HW_QUEUES) and the caller's stream holds one -- with four streams of our own two
    // sections shared a queue and took turns (rocprofv3 trace, profiles/NOTES_r03.md)
    for (int s = 0; s + 1 < ns; ++s)
      if (!h->sec_streams[s] && hipStreamCreateWithFlags(&h->sec_streams[s], hipStreamNonBlocking) != hipSuccess) return 0;
    auto stream_of = [&](int s) { return s + 1 < ns ? h->sec_streams[s] : st; };
    const size_t need = (size_t)ns * n_chunks + 1;
    while (h->sec_events.size() < need) {
      hipEvent_t ev;
      // (device-scope release: the waiting streams are on this device; a system-scope release per chunk would
      // flush more than the hand-over needs)
      if (hipEventCreateWithFlags(&ev, ALZ_TUNE("ALZ_SECPIPE_EVFLAGS", (int)(hipEventDisableTiming | hipEventReleaseToDevice))) != hipSuccess)
        return 0;
      h->sec_events.push_back(ev);
    }
    auto ev_of = [&](int s, int64_t j) { return h->sec_events[(size_t)(j * ns + s)]; };
    hipEvent_t start = h->sec_events[need - 1];
    if (hipEventRecord(start, st) != hipSuccess) return -1000;         // everything queued on `st` so far (input map, expansion)
    for (int s = 0; s + 1 < ns; ++s)
      if (hipStreamWaitEvent(h->sec_streams[s], start, 0) != hipSuccess) return -1000;
    // an error in the middle leaves work queued on the side streams: the caller's stream is made to wait for it
    // before the error is returned, so that "everything this call queued is ordered before what the caller queues
    // next on `st`" holds on the error paths too
    auto bail = [&](int code) -> int {
      for (int s = 0; s + 1 < ns; ++s)
        if (hipEventRecord(ev_of(s, 0), h->sec_streams[s]) == hipSuccess) (void)hipStreamWaitEvent(st, ev_of(s, 0), 0);
      return code;
    };
    for (int64_t j = 0; j < n_chunks; ++j) {
      const int64_t t0 = j * chunk, tn = (t0 + chunk <= n) ? chunk : n - t0;
      for (int s = 0; s < ns; ++s) {
        hipStream_t ss = stream_of(s);
        if (s > 0 && !ALZ_TUNE("ALZ_SECPIPE_NOWAIT", 0) && hipStreamWaitEvent(ss, ev_of(s - 1, j), 0) != hipSuccess) return bail(-1000);
        const int rc = run_sections_on(0, h->channels, t0, tn, s, s + 1, ss);
        if (rc) return bail(rc);
        if (hipEventRecord(ev_of(s, j), ss) != hipSuccess) return bail(-1000);
      }
    }
    // (the caller's stream ends with the last section's last chunk; every earlier section finished before that
    // chunk could start)
    h->last_kernels += "  [section pipeline: " + std::to_string(ns) + " streams x " + std::to_string(n_chunks) + " chunks]";
    return 1;
  };

  // fused cascade: all sections in one pass over the full tiles of the full channel groups;
  // the ragged remainder (and every other cascade) goes section by section
  {
    const int took = pipelined();
    if (took < 0) return took == -1000 ? fail(ALZ_E_HIP, "section pipeline: stream / event call failed") : took;
    if (took > 0) return ALZ_OK;
  }
  int64_t fused_n = 0, fused_c = 0;
  if (h->n_sections >= 2 && x_dev != y_dev && h->time_parallel == 0) {
    io.n = n;
    io.x = x_dev; io.y = y_dev;
    io.sxn = sxn; io.sxc = sxc; io.syn = syn; io.syc = syc;
    io.map_input = outer_by_input;
    io.c_first = 0; io.c_count = h->channels;
    const char *name = "";
    int rc = alz::launch_cascade(h->sec.data(), h->n_sections, io, st, &fused_n, &fused_c, &name);
    if (rc) return rc;
    if (fused_c > 0) note(name);
  }
  // Opt-in time-parallel mode on a cascade whose FIRST section the fused form does not take (gammatone.sampled,
  // reference lazy_auditory.py:158-181: eight numerator taps of +-1e3 that cancel -- carried through the fused
  // chunk-state recursion they cost ten digits).  The section is split (see alz_bank_create): its numerator runs
  // feedback-free and exact (k_fir_cm: lanes over time), its recursion serially and exactly over that output, and
  // the other sections, biquad-class, fused and time-parallel over the result, in place.
  if (fused_c == 0 && h->time_parallel != 0 && h->has_split && layout == ALZ_CHAN_MAJOR && x_dev != y_dev) {
    io.n = n;
    io.x = x_dev; io.y = y_dev;
    io.sxn = sxn; io.sxc = sxc; io.syn = syn; io.syc = syc;
    io.map_input = outer_by_input; io.pre_op = 0;
    io.c_first = 0; io.c_count = h->channels;
    bool ft = false;
    const char *name = "";
    int rc = alz::launch_fir(h->split_fir, io, st, &ft, &name);
    if (rc) return rc;
    if (ft) {
      note(name);
      // the section's recursion, serially and exactly, in place (chunked, even with the numerator out of it, its state
      // recursion measured 2e-6 on the 50 Hz band -- poles at radius 0.997, 0.4 degrees apart: not within the contract)
      io.x = y_dev; io.sxn = syn; io.sxc = syc; io.map_input = 0;
      rc = alz::launch_section(h->split_pole, io, st, &name);
      if (rc) return rc;
      note(name);
      bool taken = false;
      rc = alz::launch_scan_cascade(h->sec.data() + 1, h->n_sections - 1, io, st, h->time_parallel < 0 ? 0 : h->time_parallel,
                                    &h->scan[(size_t)h->n_sections + 1], false, &taken, &name);
      if (rc) return rc;
      if (taken) {
        note(name);
        return ALZ_OK;
      }
      return run_sections_on(0, h->channels, 0, n, 1, h->n_sections, st);
    }
  }
  if (fused_c == 0) return run_sections(0, h->channels, 0, n);
  if (fused_c < h->channels) {
    int rc = run_sections(fused_c, h->channels - fused_c, 0, n);
    if (rc) return rc;
  }
  if (fused_n < n) return run_sections(0, fused_c, fused_n, n - fused_n);
  return ALZ_OK;
}

int alz_bank_process_host(alz_bank_t *h, const double *x_host, double *y_host, int64_t n, int layout,
                          int64_t ldx, int64_t ldy) {
  if (!h || !x_host || !y_host) return fail(ALZ_E_ARG, "NULL argument");
  if (n <= 0) return n == 0 ? ALZ_OK : fail(ALZ_E_ARG, "negative block length");
  if (layout != ALZ_TIME_MAJOR && layout != ALZ_CHAN_MAJOR) return fail(ALZ_E_ARG, "bad layout");
  DeviceGuard g(h->device);
  if (!g.ok) return fail(ALZ_E_HIP, "hipSetDevice failed");
  // Stage through device buffers with an EVEN leading dimension and 16-byte aligned rows, so
  // the streaming kernels (16-byte DMA pieces) apply whatever pitch the host arrays have.
  const int64_t in_cols = layout == ALZ_TIME_MAJOR ? h->n_inputs : n;
  const int64_t out_cols = layout == ALZ_TIME_MAJOR ? h->channels : n;
  const int64_t in_rows = layout == ALZ_TIME_MAJOR ? n : h->n_inputs;
  const int64_t out_rows = layout == ALZ_TIME_MAJOR ? n : h->channels;
  if (ldx < in_cols || ldy < out_cols) return fail(ALZ_E_ARG, "leading dimension too small");
  // (a single column needs no padding: no 16-byte-piece kernel takes a one-channel bank, and a
  // contiguous vector keeps the copies one-dimensional and the time-parallel FIR kernel eligible)
  const int64_t lsx = in_cols == 1 ? 1 : (in_cols + 1) & ~(int64_t)1;
  const int64_t lsy = out_cols == 1 ? 1 : (out_cols + 1) & ~(int64_t)1;
  int rc = grow(&h->stage_x, &h->stage_x_bytes, (uint64_t)in_rows * lsx * 8);
  if (rc) return rc;
  rc = grow(&h->stage_y, &h->stage_y_bytes, (uint64_t)out_rows * lsy * 8);
  if (rc) return rc;
  // Large blocks: pin the caller's arrays for the call and run copy-in, kernels and copy-out as a
  // three-stage pipeline over chunks of the time axis (pinned, double-buffered staging).
  // The link then carries both directions at once and at the pinned rate; small blocks (the filter
  // call protocol's 4096-sample blocks) keep the plain synchronous path, whose latency is lower.
  static const int pipe_env = ALZ_TUNE("ALZ_HOST_PIPE", 1);
  const uint64_t total_bytes = ((uint64_t)in_rows * in_cols + (uint64_t)out_rows * out_cols) * 8;
  if (pipe_env && total_bytes >= ((uint64_t)64 << 20) && n >= 8192) {
    const uint64_t x_extent = ((uint64_t)(in_rows - 1) * ldx + in_cols) * 8;
    const uint64_t y_extent = ((uint64_t)(out_rows - 1) * ldy + out_cols) * 8;
    const hipError_t rx = hipHostRegister((void *)x_host, x_extent, hipHostRegisterDefault);
    const bool x_ok = rx == hipSuccess || rx == hipErrorHostMemoryAlreadyRegistered;
    hipError_t ry = hipErrorUnknown;
    if (x_ok) ry = hipHostRegister((void *)y_host, y_extent, hipHostRegisterDefault);
    const bool y_ok = ry == hipSuccess || ry == hipErrorHostMemoryAlreadyRegistered;
    (void)hipGetLastError();
    if (x_ok && y_ok) {
      int64_t n_chunks = (int64_t)(total_bytes >> 26);            // ~64 MiB of traffic per chunk
      if (n_chunks < 2) n_chunks = 2;
      if (n_chunks > 16) n_chunks = 16;
      int64_t step = ((n + n_chunks - 1) / n_chunks + 1023) / 1024 * 1024;
      n_chunks = (n + step - 1) / step;
      rc = ALZ_OK;
      for (int i = 0; i < 3 && rc == ALZ_OK; ++i)
        if (!h->host_streams[i] && hipStreamCreateWithFlags(&h->host_streams[i], hipStreamNonBlocking) != hipSuccess)
          rc = fail(ALZ_E_HIP, "hipStreamCreate failed");
      while (rc == ALZ_OK && (int64_t)h->host_events.size() < 2 * n_chunks) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) rc = fail(ALZ_E_HIP, "hipEventCreate failed");
        else h->host_events.push_back(ev);
      }
      hipStream_t s_in = h->host_streams[0], s_run = h->host_streams[1], s_out = h->host_streams[2];
      const bool tm = layout == ALZ_TIME_MAJOR;
      for (int64_t k = 0; k < n_chunks && rc == ALZ_OK; ++k) {
        const int64_t t0 = k * step, len = (t0 + step <= n ? step : n - t0);
        const double *src = tm ? x_host + t0 * ldx : x_host + t0;
        double *dst = tm ? h->stage_x + t0 * lsx : h->stage_x + t0;
        if (hipMemcpy2DAsync(dst, (size_t)lsx * 8, src, (size_t)ldx * 8, (size_t)(tm ? in_cols : len) * 8,
                             (size_t)(tm ? len : in_rows), hipMemcpyHostToDevice, s_in) != hipSuccess ||
            hipEventRecord(h->host_events[2 * k], s_in) != hipSuccess ||
            hipStreamWaitEvent(s_run, h->host_events[2 * k], 0) != hipSuccess) {
          rc = fail(ALZ_E_HIP, "host pipeline: copy-in failed");
          break;
        }
        rc = alz_bank_process_dev(h, tm ? h->stage_x + t0 * lsx : h->stage_x + t0,
                                  tm ? h->stage_y + t0 * lsy : h->stage_y + t0, len, layout, lsx, lsy, s_run);
        if (rc) break;
        double *out = tm ? y_host + t0 * ldy : y_host + t0;
        const double *res = tm ? h->stage_y + t0 * lsy : h->stage_y + t0;
        if (hipEventRecord(h->host_events[2 * k + 1], s_run) != hipSuccess ||
            hipStreamWaitEvent(s_out, h->host_events[2 * k + 1], 0) != hipSuccess ||
            hipMemcpy2DAsync(out, (size_t)ldy * 8, res, (size_t)lsy * 8, (size_t)(tm ? out_cols : len) * 8,
                             (size_t)(tm ? len : out_rows), hipMemcpyDeviceToHost, s_out) != hipSuccess)
          rc = fail(ALZ_E_HIP, "host pipeline: copy-out failed");
      }
      const hipError_t e1 = hipStreamSynchronize(s_in), e2 = hipStreamSynchronize(s_run), e3 = hipStreamSynchronize(s_out);
      if (rx == hipSuccess) (void)hipHostUnregister((void *)x_host);
      if (ry == hipSuccess) (void)hipHostUnregister((void *)y_host);
      if (rc == ALZ_OK && (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess))
        rc = fail(ALZ_E_HIP, "host pipeline: stream synchronisation failed");
      if (rc == ALZ_OK) rc = take_look_error(h);
      return rc;
    }
    if (rx == hipSuccess) (void)hipHostUnregister((void *)x_host);     // y could not be pinned: plain path
  }
  ALZ_HIP_CHECK(hipMemcpy2D(h->stage_x, (size_t)lsx * 8, x_host, (size_t)ldx * 8, (size_t)in_cols * 8,
                            (size_t)in_rows, hipMemcpyHostToDevice));
  rc = alz_bank_process_dev(h, h->stage_x, h->stage_y, n, layout, lsx, lsy, nullptr);
  if (rc) return rc;
  ALZ_HIP_CHECK(hipStreamSynchronize(nullptr));
  if (const int lrc = take_look_error(h)) return lrc;
  ALZ_HIP_CHECK(hipMemcpy2D(y_host, (size_t)ldy * 8, h->stage_y, (size_t)lsy * 8, (size_t)out_cols * 8,
                            (size_t)out_rows, hipMemcpyDeviceToHost));
  return ALZ_OK;
}

int alz_bank_set_fused(alz_bank_t *h, int on) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  h->fused = on ? 1 : 0;
  return ALZ_OK;
}

int alz_bank_set_input_map(alz_bank_t *h, int op) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  if (op != 0 && op != ALZ_MAP_ABS && op != ALZ_MAP_NEG && op != ALZ_MAP_SQUARE)
    return fail(ALZ_E_ARG, "input map must be 0, ALZ_MAP_ABS, ALZ_MAP_NEG or ALZ_MAP_SQUARE");
  h->input_map = op;
  return ALZ_OK;
}

int alz_bank_set_time_parallel(alz_bank_t *h, int64_t chunk_len) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  if (chunk_len < ALZ_TP_ONE_PASS)
    return fail(ALZ_E_ARG, "chunk length must be -2 (one pass), -1 (automatic), 0 (off) or positive");
  h->time_parallel = chunk_len;
  return ALZ_OK;
}

int alz_bank_sync(alz_bank_t *h) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  DeviceGuard g(h->device);
  ALZ_HIP_CHECK(hipDeviceSynchronize());
  return take_look_error(h);
}

int alz_bank_set_look_check(alz_bank_t *h, int mode) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  if (mode != ALZ_LOOK_CHECK_CALL && mode != ALZ_LOOK_CHECK_DEFERRED) return fail(ALZ_E_ARG, "mode must be ALZ_LOOK_CHECK_CALL or ALZ_LOOK_CHECK_DEFERRED");
  h->look_check = mode;
  return ALZ_OK;
}

int alz_bank_look_stats(const alz_bank_t *h, int64_t *launches, int64_t *gave_up, int64_t *reruns, unsigned *last_sites) {
  if (!h) return fail(ALZ_E_ARG, "NULL handle");
  if (launches) *launches = h->look_launches;
  if (gave_up) *gave_up = h->look_gave_up;
  if (reruns) *reruns = h->look_reruns;
  if (last_sites) *last_sites = h->look_last_sites;
  return ALZ_OK;
}

const char *alz_bank_last_kernel(const alz_bank_t *h) { return h ? h->last_kernels.c_str() : ""; }

}  // extern "C"
