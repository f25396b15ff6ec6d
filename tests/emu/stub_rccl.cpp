// TEST-ONLY stand-in for librccl (loaded through LMN_RCCL_LIB by the emulation build): the handful of NCCL entry points
// the library's built-in transport binds (prover.cpp RcclApi), implemented over POSIX shared memory between the rank
// processes of one machine.  "Device" memory of the emulation build is host memory, streams do not exist: every call
// completes before it returns.  It lets the CPU suite run lmn_ctx_set_shard_rccl - unique-id exchange, one communicator
// per rank, grouped all-gathers, grouped send / recv (the all-to-all) - with 2, 4 and 8 ranks.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t REGION = 96u << 20;   // per-rank outbox (sparse: only touched pages exist)
constexpr int MAX_WORLD = 8;
struct Dir {
  uint64_t off[MAX_WORLD], bytes[MAX_WORLD];
};
struct Shared {
  std::atomic<int> arrived, generation, joined;
  Dir dir[MAX_WORLD];
};
struct Comm {
  int rank, world;
  Shared* sh;
  char* data;   // world regions
  size_t map_bytes;
  char name[64];
};
struct Op {
  bool send;
  const void* src;
  void* dst;
  size_t bytes;
  int peer;
  Comm* comm;
};
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

void barrier(Comm* c) {
  const int gen = c->sh->generation.load();
  if (c->sh->arrived.fetch_add(1) + 1 == c->world) {
    c->sh->arrived.store(0);
    c->sh->generation.fetch_add(1);
  } else {
    while (c->sh->generation.load() == gen) usleep(50);
  }
}

int run_ops(std::vector<Op>& ops) {
  if (ops.empty()) return 0;
  Comm* c = ops[0].comm;
  // phase 1: everything this rank sends goes to its outbox, one directory entry per destination (sends to one peer are
  // concatenated in call order, as are the matching receives)
  uint64_t at = 0;
  Dir& d = c->sh->dir[c->rank];
  for (int p = 0; p < c->world; ++p) {
    d.off[p] = at;
    d.bytes[p] = 0;
    for (auto& o : ops)
      if (o.send && o.peer == p) {
        if (at + o.bytes > REGION) return 2;
        memcpy(c->data + (size_t)c->rank * REGION + at, o.src, o.bytes);
        at += o.bytes;
        d.bytes[p] += o.bytes;
      }
  }
  barrier(c);
  int rc = 0;
  for (int p = 0; p < c->world; ++p) {
    uint64_t got = 0;
    const Dir& pd = c->sh->dir[p];
    for (auto& o : ops)
      if (!o.send && o.peer == p) {
        if (got + o.bytes > pd.bytes[c->rank]) {
          rc = 3;   // a receive without a matching send
          break;
        }
        memcpy(o.dst, c->data + (size_t)p * REGION + pd.off[c->rank] + got, o.bytes);
        got += o.bytes;
      }
    if (got != pd.bytes[c->rank]) rc = rc ? rc : 4;   // a send nobody received
  }
  barrier(c);
  return rc;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(char* id /* 128 bytes */) {
  memset(id, 0, 128);
  unsigned long long r = 0;
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f || fread(&r, sizeof r, 1, f) != 1) return 1;
  fclose(f);
  snprintf(id, 64, "/lmn_stub_rccl_%016llx", r);
  return 0;
}

struct IdByValue {
  char internal[128];
};
int ncclCommInitRank(void** comm_out, int world, IdByValue id, int rank) {
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return 1;
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  // several communicators may be created from one id (one per prover context, in the same order on every rank): the
  // n-th of them gets its own shared-memory object
  static std::atomic<int> seq{0};
  char base[40];
  memcpy(base, id.internal, 39);
  base[39] = 0;
  snprintf(c->name, sizeof c->name, "%s_%d", base, seq.fetch_add(1));
  c->map_bytes = 4096 + (size_t)world * REGION;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 1;
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) return 1;
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return 1;
  c->sh = (Shared*)m;            // a fresh shm object is zero-filled: counters start at 0
  c->data = (char*)m + 4096;
  c->sh->joined.fetch_add(1);
  while (c->sh->joined.load() < world) usleep(100);   // like ncclCommInitRank: returns when every rank has joined
  barrier(c);
  if (rank == 0) shm_unlink(c->name);                 // everybody has it mapped
  *comm_out = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  munmap((void*)c->sh, c->map_bytes);
  delete c;
  return 0;
}

int ncclGroupStart() {
  ++g_group_depth;
  return 0;
}
int ncclGroupEnd() {
  if (g_group_depth <= 0) return 1;
  if (--g_group_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void*) {
  if (dtype != 1 && dtype != 0) return 1;
  g_ops.push_back({true, buf, nullptr, count, peer, (Comm*)comm});
  if (g_group_depth == 0) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
  }
  return 0;
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void*) {
  if (dtype != 1 && dtype != 0) return 1;
  g_ops.push_back({false, nullptr, buf, count, peer, (Comm*)comm});
  if (g_group_depth == 0) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
  }
  return 0;
}

// all-gather = every rank sends its part to every rank
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void*) {
  Comm* c = (Comm*)comm;
  if (dtype != 1 && dtype != 0) return 1;
  for (int p = 0; p < c->world; ++p) {
    g_ops.push_back({true, send, nullptr, count, p, c});
    g_ops.push_back({false, nullptr, (char*)recv + (size_t)p * count, count, p, c});
  }
  if (g_group_depth == 0) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
  }
  return 0;
}

}  // extern "C"
