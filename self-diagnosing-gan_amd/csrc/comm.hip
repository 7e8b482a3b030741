// Data-parallel exchange behind the C ABI: an explicit context (RCCL communicator of one process per GPU) and the two
// collectives the path needs -- SURVEY §8(b) "minimum exports" diagan_ctx / diagan_allreduce_grads /
// diagan_allgather_logits.
//
// Reference side: DistributedDataParallel's gradient averaging (stylegan2/train_ffhq.py:572-585) and the per-sample logit
// gather `concat_all_gather` (stylegan2/train_ffhq.py:150-161); here ONE all-reduce (SUM; the 1/W is folded into the
// Adam kernel's gradient read) of a network's flat gradient slab per update, and ONE all-gather of the contiguous logit
// shards per snapshot, both enqueued on the caller's stream (so a whole update can be captured in a hipGraph).
//
// RCCL is resolved with dlopen at context creation: the library has no link-time dependency on it, builds and loads on a
// box without RCCL, and single-process runs never touch it.  The unique id travels through the host launcher
// (diagan/trainer/distributed.py broadcasts the 128 bytes with torch.distributed's store-backed object collective).
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(rccl_comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, rccl_comm, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
};

Rccl g_rccl;

// enum values of rccl.h (ncclDataType_t / ncclRedOp_t)
constexpr int kInt8 = 0, kFloat32 = 7, kFloat64 = 8, kSum = 0;

int load_rccl() {
  if (g_rccl.handle) return DIAGAN_OK;
  // torch's own copy first when it is already mapped (one RCCL per process), then the ROCm one
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (h) break;
  }
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) return diagan::set_err(DIAGAN_EUNSUP, "comm: librccl.so not found (%s)", dlerror());
  Rccl r;
  r.handle = h;
  r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  r.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
  r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.all_gather || !r.error_string)
    return diagan::set_err(DIAGAN_EUNSUP, "comm: librccl.so lacks a required entry point");
  g_rccl = r;
  return DIAGAN_OK;
}

#define DG_RCCL(call)                                                                              \
  do {                                                                                             \
    int _r = (call);                                                                               \
    if (_r != 0) return diagan::set_err(DIAGAN_EHIP, "%s: %s", #call, g_rccl.error_string(_r));    \
  } while (0)

}  // namespace

struct diagan_ctx {
  rccl_comm comm;
  int rank, world, device;
};

// 128 bytes that identify one communicator; rank 0 makes them, every rank passes them to diagan_ctx_create
DIAGAN_API int diagan_comm_unique_id(void* id128) {
  DG_REQUIRE(id128, "comm_unique_id: null buffer");
  int rc = load_rccl();
  if (rc != DIAGAN_OK) return rc;
  rccl_unique_id id;
  DG_RCCL(g_rccl.get_unique_id(&id));
  memcpy(id128, id.internal, 128);
  return DIAGAN_OK;
}

DIAGAN_API int diagan_ctx_create(diagan_ctx** out, const void* id128, int rank, int world, int device) {
  DG_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world && device >= 0, "ctx_create: bad arguments");
  int rc = load_rccl();
  if (rc != DIAGAN_OK) return rc;
  DG_HIP(hipSetDevice(device));
  rccl_unique_id id;
  memcpy(id.internal, id128, 128);
  diagan_ctx* c = new diagan_ctx{nullptr, rank, world, device};
  int r = g_rccl.comm_init_rank(&c->comm, world, id, rank);
  if (r != 0) {
    delete c;
    return diagan::set_err(DIAGAN_EHIP, "ncclCommInitRank: %s", g_rccl.error_string(r));
  }
  *out = c;
  return DIAGAN_OK;
}

DIAGAN_API int diagan_ctx_destroy(diagan_ctx* c) {
  if (!c) return DIAGAN_OK;
  if (c->comm) g_rccl.comm_destroy(c->comm);
  delete c;
  return DIAGAN_OK;
}

DIAGAN_API int diagan_ctx_rank(const diagan_ctx* c) { return c ? c->rank : -1; }
DIAGAN_API int diagan_ctx_world(const diagan_ctx* c) { return c ? c->world : -1; }

// in-place SUM of a flat fp32 gradient slab over all ranks, on `stream` (the mean's 1/W is applied by diagan_adam_step)
DIAGAN_API int diagan_allreduce_grads(diagan_ctx* c, float* slab, int64_t n, void* stream) {
  DG_REQUIRE(c && slab && n > 0, "allreduce_grads: bad arguments");
  DG_RCCL(g_rccl.all_reduce(slab, slab, (size_t)n, kFloat32, kSum, c->comm, (hipStream_t)stream));
  return DIAGAN_OK;
}

// recv[r * n_per_rank .. (r+1) * n_per_rank) = rank r's `send` (elem_bytes 4: float32 logits, 8: float64 record rows);
// values are copied, never summed: index assignment stays bit-exact
DIAGAN_API int diagan_allgather_logits(diagan_ctx* c, const void* send, void* recv, int64_t n_per_rank, int elem_bytes,
                                       void* stream) {
  DG_REQUIRE(c && send && recv && n_per_rank > 0 && (elem_bytes == 4 || elem_bytes == 8 || elem_bytes == 1),
             "allgather_logits: bad arguments");
  const int dt = elem_bytes == 4 ? kFloat32 : (elem_bytes == 8 ? kFloat64 : kInt8);
  DG_RCCL(g_rccl.all_gather(send, recv, (size_t)n_per_rank, dt, c->comm, (hipStream_t)stream));
  return DIAGAN_OK;
}
