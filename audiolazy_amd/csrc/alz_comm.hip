// alz_comm.hip -- the one optional collective of the engine, bound to RCCL directly.
//
// Filtering shards with no exchange (channels / input streams / frames per GPU); only a
// DOWNSTREAM consumer that needs every channel on one device -- or their mix -- needs a collective, and it is
// a single one: all-gather / gather of the [C / G, N] shards, or a sum of the per-rank mixes (all-reduce /
// reduce).  audiolazy_amd/sharding.py offers it through torch.distributed; these entry points offer the same to
// a caller that uses the C ABI alone (alz_malloc + alz_bank_process_dev): one process per GPU, the 128-byte
// unique id of rank 0 handed to the others by whatever launcher started them.
//
// librccl.so is resolved at run time (dlopen), not linked: a process that never gathers does not load it, and a
// process that already has RCCL (PyTorch-ROCm bundles one) gets that copy -- two RCCL runtimes in one process
// do not share their device state.  The reference has no counterpart (single process, lazy_stream.py:114).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "alz_common.h"

namespace {

// the part of rccl.h this file needs (ROCm 7.2: NCCL_UNIQUE_ID_BYTES 128, ncclFloat64 = 8, ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef void *comm_t;
typedef int (*get_unique_id_fn)(UniqueId *);
typedef int (*comm_init_rank_fn)(comm_t *, int, UniqueId, int);
typedef int (*comm_destroy_fn)(comm_t);
typedef const char *(*get_error_string_fn)(int);
typedef int (*all_gather_fn)(const void *, void *, size_t, int, comm_t, hipStream_t);
typedef int (*gather_fn)(const void *, void *, size_t, int, int, comm_t, hipStream_t);
typedef int (*all_reduce_fn)(const void *, void *, size_t, int, int, comm_t, hipStream_t);
typedef int (*reduce_fn)(const void *, void *, size_t, int, int, int, comm_t, hipStream_t);
constexpr int kFloat64 = 8, kSum = 0;

struct Rccl {
  void *lib = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  get_error_string_fn get_error_string = nullptr;
  all_gather_fn all_gather = nullptr;
  gather_fn gather = nullptr;
  all_reduce_fn all_reduce = nullptr;
  reduce_fn reduce = nullptr;
  std::string why;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already mapped first (PyTorch's), then the system's
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
      if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    if (!r.lib)
      for (const char *n : names)
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!r.lib) {
      const char *e = dlerror();   // (the call clears the error: read it once)
      r.why = std::string("librccl.so not found: ") + (e ? e : "");
      return;
    }
#define ALZ_SYM(field, name)                                             \
  r.field = (decltype(r.field))dlsym(r.lib, name);                       \
  if (!r.field && r.why.empty()) r.why = std::string("librccl.so lacks ") + name;
    ALZ_SYM(get_unique_id, "ncclGetUniqueId")
    ALZ_SYM(comm_init_rank, "ncclCommInitRank")
    ALZ_SYM(comm_destroy, "ncclCommDestroy")
    ALZ_SYM(get_error_string, "ncclGetErrorString")
    ALZ_SYM(all_gather, "ncclAllGather")
    ALZ_SYM(gather, "ncclGather")
    ALZ_SYM(all_reduce, "ncclAllReduce")
    ALZ_SYM(reduce, "ncclReduce")
#undef ALZ_SYM
  });
  return &r;
}

int rccl_fail(const char *what, int code) {
  Rccl *r = rccl();
  return alz::fail(ALZ_E_HIP, std::string(what) + ": " + (r->get_error_string ? r->get_error_string(code) : "RCCL error") +
                                  " (" + std::to_string(code) + ")");
}

}  // namespace

struct alz_comm {
  comm_t comm = nullptr;
  int device = 0, world = 1, rank = 0;
};

extern "C" {

int alz_comm_unique_id(void *id_out) {
  if (!id_out) return alz::fail(ALZ_E_ARG, "NULL argument");
  Rccl *r = rccl();
  if (!r->why.empty()) return alz::fail(ALZ_E_UNSUPPORTED, r->why);
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, sizeof(id.internal));
  return ALZ_OK;
}

int alz_comm_create(int device, int world, int rank, const void *id128, alz_comm_t **out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return alz::fail(ALZ_E_ARG, "bad communicator geometry");
  Rccl *r = rccl();
  if (!r->why.empty()) return alz::fail(ALZ_E_UNSUPPORTED, r->why);
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) ALZ_HIP_CHECK(hipSetDevice(device));
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  alz_comm *c = new alz_comm;
  c->device = device; c->world = world; c->rank = rank;
  const int rc = r->comm_init_rank(&c->comm, world, id, rank);     // (collective: every rank calls it)
  if (prev != device) (void)hipSetDevice(prev);
  if (rc) {
    delete c;
    return rccl_fail("ncclCommInitRank", rc);
  }
  *out = c;
  return ALZ_OK;
}

int alz_comm_destroy(alz_comm_t *c) {
  if (!c) return ALZ_OK;
  Rccl *r = rccl();
  if (c->comm && r->comm_destroy) (void)r->comm_destroy(c->comm);
  delete c;
  return ALZ_OK;
}

int alz_comm_gather(alz_comm_t *c, const double *send_dev, double *recv_dev, int64_t count, int root, void *stream) {
  if (!c || !send_dev || count < 0) return alz::fail(ALZ_E_ARG, "bad gather arguments");
  if (root >= c->world) return alz::fail(ALZ_E_ARG, "gather: root outside the communicator");
  if ((root < 0 || root == c->rank) && !recv_dev) return alz::fail(ALZ_E_ARG, "gather: the receiving rank needs recv_dev");
  Rccl *r = rccl();
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != c->device) ALZ_HIP_CHECK(hipSetDevice(c->device));
  const int rc = root < 0 ? r->all_gather(send_dev, recv_dev, (size_t)count, kFloat64, c->comm, (hipStream_t)stream)
                          : r->gather(send_dev, recv_dev, (size_t)count, kFloat64, root, c->comm, (hipStream_t)stream);
  if (prev != c->device) (void)hipSetDevice(prev);
  return rc ? rccl_fail(root < 0 ? "ncclAllGather" : "ncclGather", rc) : ALZ_OK;
}

int alz_comm_sum(alz_comm_t *c, const double *send_dev, double *recv_dev, int64_t count, int root, void *stream) {
  if (!c || !send_dev || count < 0) return alz::fail(ALZ_E_ARG, "bad reduction arguments");
  if (root >= c->world) return alz::fail(ALZ_E_ARG, "sum: root outside the communicator");
  if ((root < 0 || root == c->rank) && !recv_dev) return alz::fail(ALZ_E_ARG, "sum: the receiving rank needs recv_dev");
  Rccl *r = rccl();
  int prev = 0;
  ALZ_HIP_CHECK(hipGetDevice(&prev));
  if (prev != c->device) ALZ_HIP_CHECK(hipSetDevice(c->device));
  const int rc = root < 0 ? r->all_reduce(send_dev, recv_dev, (size_t)count, kFloat64, kSum, c->comm, (hipStream_t)stream)
                          : r->reduce(send_dev, recv_dev, (size_t)count, kFloat64, kSum, root, c->comm, (hipStream_t)stream);
  if (prev != c->device) (void)hipSetDevice(prev);
  return rc ? rccl_fail(root < 0 ? "ncclAllReduce" : "ncclReduce", rc) : ALZ_OK;
}

}  // extern "C"
