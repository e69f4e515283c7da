// C1: the one collective of the path -- sum of the per-shard Gram matrices (and of the residual's two scalars)
// over the GPUs of a node, RCCL all-reduce over xGMI.  The reference has no distributed code; the Python host
// (aggforce_amd/distributed.py) goes through torch.distributed (backend "nccl" = RCCL).  These entry points give
// a host WITHOUT torch (path B) the same step on the C ABI: one process per GPU, rank 0 makes a
// unique id and hands it to the others by whatever channel the host has (file, MPI, socket), every rank creates
// its communicator, then calls aggf_allreduce_sum between aggf_gram and aggf_eq_qp_solve.
//
// RCCL is loaded on first use (dlopen of librccl.so.1): single-GPU users never pay for, or depend on, it.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "aggf_common.h"

namespace aggf {

constexpr int RCCL_ID_BYTES = 128;       // NCCL_UNIQUE_ID_BYTES
struct RcclId { char internal[RCCL_ID_BYTES]; };
typedef void* rccl_comm_t;
typedef int (*fn_get_id)(RcclId*);
typedef int (*fn_init_rank)(rccl_comm_t*, int, RcclId, int);
typedef int (*fn_destroy)(rccl_comm_t);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_version)(int*);
// ncclSum, ncclFloat32, ncclFloat64 and the by-value 128-byte id of rccl.h, NCCL API 2.x (unchanged since 2.0); the
// version of the library that was actually loaded is checked below, so a future ABI break is refused, not miscalled
constexpr int RCCL_SUM = 0, RCCL_F32 = 7, RCCL_F64 = 8;
constexpr int RCCL_MIN_VERSION = 20000, RCCL_MAX_VERSION = 29999;  // ncclGetVersion: major * 10000 + minor * 100 + patch (>= 2.9)

struct Rccl {
  void* handle = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_errstr errstr = nullptr;
  int version = 0;
};

static Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (r.handle) {
      r.get_id = (fn_get_id)dlsym(r.handle, "ncclGetUniqueId");
      r.init_rank = (fn_init_rank)dlsym(r.handle, "ncclCommInitRank");
      r.destroy = (fn_destroy)dlsym(r.handle, "ncclCommDestroy");
      r.allreduce = (fn_allreduce)dlsym(r.handle, "ncclAllReduce");
      r.errstr = (fn_errstr)dlsym(r.handle, "ncclGetErrorString");
      fn_version ver = (fn_version)dlsym(r.handle, "ncclGetVersion");
      if (ver && ver(&r.version) != 0) r.version = 0;
      // versions before 2.9 report major * 1000 + minor * 100 + patch
      if (r.version > 0 && r.version < 10000) r.version = (r.version / 1000) * 10000 + r.version % 1000;
    }
  });
  if (!(r.handle && r.get_id && r.init_rank && r.destroy && r.allreduce)) return nullptr;
  if (r.version < RCCL_MIN_VERSION || r.version > RCCL_MAX_VERSION) return nullptr;  // constants above are for API 2.x
  return &r;
}

static int rccl_fail(Rccl* r, int rc, const char* what) {
  return fail(AGGF_ERR_COMM, "%s failed: %s", what, (r && r->errstr) ? r->errstr(rc) : "RCCL error");
}

}  // namespace aggf

using namespace aggf;

extern "C" int aggf_comm_unique_id(void* id_out, size_t id_bytes) {
  if (!id_out || id_bytes < (size_t)RCCL_ID_BYTES) return fail(AGGF_ERR_ARG, "aggf_comm_unique_id: need %d bytes", RCCL_ID_BYTES);
  Rccl* r = rccl();
  if (!r) return fail(AGGF_ERR_COMM, "aggf_comm_unique_id: librccl.so.1 could not be loaded");
  RcclId id;
  const int rc = r->get_id(&id);
  if (rc) return rccl_fail(r, rc, "ncclGetUniqueId");
  memcpy(id_out, &id, RCCL_ID_BYTES);
  return AGGF_OK;
}

extern "C" int aggf_comm_init(const void* id, size_t id_bytes, int32_t rank, int32_t world, void** comm_out) {
  if (!id || id_bytes < (size_t)RCCL_ID_BYTES || !comm_out || world <= 0 || rank < 0 || rank >= world)
    return fail(AGGF_ERR_ARG, "aggf_comm_init: bad argument");
  Rccl* r = rccl();
  if (!r) return fail(AGGF_ERR_COMM, "aggf_comm_init: librccl.so.1 could not be loaded");
  RcclId uid;
  memcpy(&uid, id, RCCL_ID_BYTES);
  rccl_comm_t c = nullptr;
  const int rc = r->init_rank(&c, world, uid, rank);
  if (rc) return rccl_fail(r, rc, "ncclCommInitRank");
  *comm_out = c;
  return AGGF_OK;
}

extern "C" int aggf_comm_destroy(void* comm) {
  if (!comm) return AGGF_OK;
  Rccl* r = rccl();
  if (!r) return fail(AGGF_ERR_COMM, "aggf_comm_destroy: librccl.so.1 could not be loaded");
  const int rc = r->destroy((rccl_comm_t)comm);
  return rc ? rccl_fail(r, rc, "ncclCommDestroy") : AGGF_OK;
}

extern "C" int aggf_allreduce_sum(void* buf, int64_t count, int dtype, void* comm, void* stream_v) {
  if (!buf || !comm || count < 0) return fail(AGGF_ERR_ARG, "aggf_allreduce_sum: bad argument");
  if (dtype != AGGF_F32 && dtype != AGGF_F64) return fail(AGGF_ERR_ARG, "aggf_allreduce_sum: bad dtype");
  if (count == 0) return AGGF_OK;
  Rccl* r = rccl();
  if (!r) return fail(AGGF_ERR_COMM, "aggf_allreduce_sum: librccl.so.1 could not be loaded");
  const int rc = r->allreduce(buf, buf, (size_t)count, dtype == AGGF_F64 ? RCCL_F64 : RCCL_F32, RCCL_SUM,
                              (rccl_comm_t)comm, (hipStream_t)stream_v);
  return rc ? rccl_fail(r, rc, "ncclAllReduce") : AGGF_OK;
}
