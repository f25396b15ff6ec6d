// Single-proof sharding over the GPUs of one node: the built-in RCCL transport and
// lmn_ctx_set_shard(_rccl) / clear_shard.  The sharded stages themselves live with the phases they split (commit.cpp,
// prove.cpp).
#include <dlfcn.h>

#include "prover_internal.h"

namespace lmn {

// Built-in transport: RCCL over xGMI, bound at run time (the library has no link-time dependency on librccl, so a
// single-GPU deployment never loads it).  One communicator per context, collectives enqueued on the prover's stream.
// The handful of RCCL entry points used are declared here (NCCL's stable C ABI) instead of including rccl.h, so that the
// test-only emulation build carries the same transport code: tests/emu/stub_rccl.cpp stands in for librccl there
// (LMN_RCCL_LIB names the library to load) and runs unique-id exchange, per-rank communicator initialisation, group
// batching and the collectives themselves with world 2 / 4 / 8 on a machine without GPUs.
typedef struct lmnNcclComm* lmnNcclComm_t;
struct lmnNcclUniqueId {
  char internal[128];
};
constexpr int LMN_NCCL_UINT8 = 1;   // ncclUint8 / ncclChar
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(lmnNcclUniqueId*) = nullptr;
  int (*CommInitRank)(lmnNcclComm_t*, int, lmnNcclUniqueId, int) = nullptr;
  int (*CommDestroy)(lmnNcclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, lmnNcclComm_t, void*) = nullptr;
  int (*Send)(const void*, size_t, int, int, lmnNcclComm_t, void*) = nullptr;
  int (*Recv)(void*, size_t, int, int, lmnNcclComm_t, void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  static RcclApi& get() {
    static RcclApi api = [] {
      RcclApi a;
      const char* env = getenv("LMN_RCCL_LIB");
      if (env && *env) {
        a.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
      } else {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
          a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
          if (a.handle) break;
        }
      }
      if (a.handle) {
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
        a.Send = (decltype(a.Send))dlsym(a.handle, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(a.handle, "ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
      }
      return a;
    }();
    if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather)
      throw LmnError(LMN_ERR_NO_DEVICE, "librccl could not be loaded (needed for lmn_ctx_set_shard_rccl; LMN_RCCL_LIB overrides "
                                        "the library name)");
    return api;
  }
};
struct RcclTransport {
  lmnNcclComm_t comm = nullptr;
  uint32_t rank = 0, world = 1;
  static int all_gather(void* user, void* buf, size_t bytes, void* stream) {
    RcclTransport* t = (RcclTransport*)user;
    return RcclApi::get().AllGather((const char*)buf + (size_t)t->rank * bytes, buf, bytes, LMN_NCCL_UINT8, t->comm, stream);
  }
  static int group_begin(void*) { return RcclApi::get().GroupStart(); }
  static int group_end(void*) { return RcclApi::get().GroupEnd(); }
  // grouped point-to-point: xGMI links are point-to-point, every peer pair moves its part over its own link
  static int all_to_all(void* user, const void* send, const size_t* so, const size_t* sb, void* recv, const size_t* ro,
                        const size_t* rb, void* stream) {
    RcclTransport* t = (RcclTransport*)user;
    RcclApi& api = RcclApi::get();
    int rc = api.GroupStart();
    for (uint32_t p = 0; p < t->world && rc == 0; ++p) {
      if (sb[p]) rc = api.Send((const char*)send + so[p], sb[p], LMN_NCCL_UINT8, (int)p, t->comm, stream);
      if (rc == 0 && rb[p]) rc = api.Recv((char*)recv + ro[p], rb[p], LMN_NCCL_UINT8, (int)p, t->comm, stream);
    }
    const int rc_end = api.GroupEnd();
    return rc ? rc : rc_end;
  }
};
void rccl_unique_id(uint8_t* out) {
  static_assert(sizeof(lmnNcclUniqueId) <= LMN_RCCL_ID_BYTES, "ncclUniqueId larger than the ABI slot");
  lmnNcclUniqueId id;
  if (RcclApi::get().GetUniqueId(&id) != 0) throw LmnError(LMN_ERR_INTERNAL, "ncclGetUniqueId failed");
  memset(out, 0, LMN_RCCL_ID_BYTES);
  memcpy(out, &id, sizeof id);
}
void Context::set_shard_rccl(uint32_t rank, uint32_t world, uint32_t fri_min_log, const uint8_t* id_bytes) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  {
    lmn_collective probe{nullptr, &RcclTransport::all_gather, nullptr, nullptr, nullptr};
    check_shard_args(rank, world, fri_min_log, &probe);  // a rejected call leaves the current sharding untouched
  }
  clear_shard();
  lmnNcclUniqueId id;
  memcpy(&id, id_bytes, sizeof id);
  RcclTransport* t = new RcclTransport();
  t->rank = rank;
  t->world = world;
  if (RcclApi::get().CommInitRank(&t->comm, (int)world, id, (int)rank) != 0) {
    delete t;
    throw LmnError(LMN_ERR_INTERNAL, "ncclCommInitRank failed");
  }
  RcclApi& api = RcclApi::get();
  const bool can_group = api.GroupStart && api.GroupEnd;
  lmn_collective c{t, &RcclTransport::all_gather, can_group ? &RcclTransport::group_begin : nullptr,
                   can_group ? &RcclTransport::group_end : nullptr,
                   can_group && api.Send && api.Recv ? &RcclTransport::all_to_all : nullptr};
  try {
    set_shard(rank, world, fri_min_log, &c);
  } catch (...) {
    RcclApi::get().CommDestroy(t->comm);
    delete t;
    throw;
  }
  shard_.rccl = t;
}
void rccl_release(void* p) {
  RcclTransport* t = (RcclTransport*)p;
  if (t->comm) RcclApi::get().CommDestroy(t->comm);
  delete t;
}

void Context::check_shard_args(uint32_t rank, uint32_t world, uint32_t fri_min_log, const lmn_collective* coll) {
  if (world == 0 || (world & (world - 1)) || world > 8 || rank >= world)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "shard: world must be 1, 2, 4 or 8 and rank < world");
  if (!coll || !coll->all_gather) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "shard: missing all_gather");
  int g = 0;
  while ((1u << g) < world) ++g;
  if (fri_min_log == 0) fri_min_log = 16;
  // a split quotient column / FRI layer needs at least 4 rows per rank
  if ((int)fri_min_log < g + 1 || fri_min_log > 30) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "shard: bad fri_min_log");
}

void Context::set_shard(uint32_t rank, uint32_t world, uint32_t fri_min_log, const lmn_collective* coll) {
  check_shard_args(rank, world, fri_min_log, coll);
  int g = 0;
  while ((1u << g) < world) ++g;
  if (fri_min_log == 0) fri_min_log = 16;
  lmn_sync(stream_);
  void* keep = shard_.rccl;
  shard_ = Shard{};
  shard_.rccl = keep;
  shard_.active = true;
  shard_.rank = rank;
  shard_.world = world;
  shard_.g = g;
  shard_.fri_min_log = (int)fri_min_log;
  shard_.coll = *coll;
}

void Context::clear_shard() {
  lmn_sync(stream_);
  if (shard_.rccl) rccl_release(shard_.rccl);
  shard_ = Shard{};
}

}  // namespace lmn
