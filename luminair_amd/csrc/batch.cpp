// Runtime of the lock-step batch build (batch.h) and its C entry points lmn_batch_* (include/luminair_hip.h).
// Only compiled into libluminair_hip_batch.so (-DLMN_BATCH).
#ifdef LMN_BATCH
#include "../../include/luminair_hip_batch.h"

#include <sys/mman.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <vector>

#include "capi_internal.h"

// Sanitizer builds of the emulated batch library (tests/emu/build_emu.sh batch asan | tsan): every switch between a worker
// thread's scheduler context and a member fiber is announced, as tests/emu/emu_runtime.cpp does for the kernel fibers.
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define LMN_BATCH_ASAN 1
#endif
#if defined(__SANITIZE_THREAD__)
#include <sanitizer/tsan_interface.h>
#define LMN_BATCH_TSAN 1
#endif

struct lmn_batch;
namespace lmn {

thread_local BatchGroup* tls_batch_group = nullptr;
thread_local int tls_batch_member = 0;
// A thread that launches outside any batch (the caller of lmn_batch_create / lmn_batch_prove, an lmn_prove_submit worker,
// plain lmn_prove) gets a group of one.  It is released when the thread ends - except on the main thread, whose
// thread-local destructors run while the HIP runtime may already be shutting down.
struct SoloGroupHolder {
  BatchGroup* g = nullptr;
  ~SoloGroupHolder();
};
static thread_local SoloGroupHolder tls_solo;
struct BounceOut {   // device -> pageable host memory: lands in a page-locked slot, copied out after the group's wait
  void* dst;
  const void* slot;
  size_t bytes;
};
static std::vector<BounceOut>& bounce_out_list();   // of the calling member

constexpr size_t BATCH_HALF_BYTES = 12u << 20;    // launch tables + copy lists of one synchronisation epoch
constexpr size_t BATCH_BOUNCE_TOTAL = 256u << 20; // page-locked bounce memory of a group, split over members x 2 epoch halves
constexpr size_t BATCH_BOUNCE_MAX = 1u << 20;     // larger pageable transfers are issued by the member itself
constexpr size_t BATCH_MEMBER_COPIES = 4096;

void batch_check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw LmnError(-100, std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

struct SpinGuard {
  std::atomic_flag& f;
  explicit SpinGuard(std::atomic_flag& f_) : f(f_) {
    while (f.test_and_set(std::memory_order_acquire)) {
      // lock-step members arrive together: back off between attempts instead of hammering the line
      for (int k = 0; k < 64; ++k) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
  }
  ~SpinGuard() { f.clear(std::memory_order_release); }
};

// ---- page-locked host ranges (the copy kernel can address them directly)
struct PinnedRange {
  uintptr_t base, size;
  intptr_t dev_delta;
};
static std::atomic_flag g_pinned_lock = ATOMIC_FLAG_INIT;
static std::vector<PinnedRange> g_pinned;
static std::atomic<uint64_t> g_pinned_version{1};   // bumped whenever the registry changes
void batch_register_pinned(const void* p, size_t bytes) {
  void* d = nullptr;
  intptr_t delta = 0;
  if (hipHostGetDevicePointer(&d, const_cast<void*>(p), 0) == hipSuccess && d) delta = (intptr_t)((uintptr_t)d - (uintptr_t)p);
  SpinGuard lk(g_pinned_lock);
  g_pinned.push_back({(uintptr_t)p, bytes, delta});
  g_pinned_version.fetch_add(1, std::memory_order_release);
}
void batch_unregister_pinned(const void* p) {
  SpinGuard lk(g_pinned_lock);
  for (size_t i = 0; i < g_pinned.size(); ++i)
    if (g_pinned[i].base == (uintptr_t)p) {
      g_pinned.erase(g_pinned.begin() + i);
      g_pinned_version.fetch_add(1, std::memory_order_release);
      return;
    }
}
static bool pinned_device_address(const void* p, size_t n, uint64_t* out) {
  // lock-step members ask at the same moment: each thread keeps its own copy of the (rarely changing) registry and
  // takes the lock only to refresh it
  static thread_local std::vector<PinnedRange> mine;
  static thread_local uint64_t mine_version = 0;
  const uint64_t ver = g_pinned_version.load(std::memory_order_acquire);
  if (mine_version != ver) {
    SpinGuard lk(g_pinned_lock);
    mine = g_pinned;
    mine_version = g_pinned_version.load(std::memory_order_relaxed);
  }
  for (auto& r : mine)
    if ((uintptr_t)p >= r.base && (uintptr_t)p + n <= r.base + r.size) {
      *out = (uint64_t)((uintptr_t)p + r.dev_delta);
      return true;
    }
  return false;
}

SoloGroupHolder::~SoloGroupHolder() {
  if (!g) return;
  if ((long)syscall(SYS_gettid) != (long)getpid()) {
    batch_group_release(*g);
    delete g;
  }
  g = nullptr;
}

// `solo`: a group of one never batches transfers (batch_copy declines outside a batch) and may simply drain its stream
// when its launch table is full: 1 MiB of table halves and no bounce memory instead of 24 + 24 + 8 MiB per thread.
void batch_group_init(BatchGroup& g, int slots, bool solo) {
  g.slots = slots;
  g.n_alloc = slots;
  g.half_bytes = solo ? (size_t)(512u << 10) : BATCH_HALF_BYTES;
  batch_check_hip(hipHostMalloc((void**)&g.host, 2 * g.half_bytes, hipHostMallocDefault), "hipHostMalloc");
  batch_check_hip(hipMalloc((void**)&g.dev, 2 * g.half_bytes), "hipMalloc");
  batch_register_pinned(g.host, 2 * g.half_bytes);
  g.bounce_bytes = solo ? (size_t)4096 : std::min<size_t>(4u << 20, std::max<size_t>(256u << 10, BATCH_BOUNCE_TOTAL / (2 * (size_t)slots)));
  batch_check_hip(hipHostMalloc((void**)&g.bounce, (size_t)slots * 2 * g.bounce_bytes, hipHostMallocDefault), "hipHostMalloc");
  batch_register_pinned(g.bounce, (size_t)slots * 2 * g.bounce_bytes);
  g.member = new BatchMember[slots];
  for (int i = 0; i < slots; ++i) {
    g.member[i].copies = new BatchCopy[BATCH_MEMBER_COPIES];
    g.member[i].copy_cap = BATCH_MEMBER_COPIES;
    g.member[i].bounce_out = new std::vector<BounceOut>();
  }
}
void batch_group_release(BatchGroup& g) {
  if (g.host) {
    batch_unregister_pinned(g.host);
    (void)hipHostFree(g.host);
  }
  if (g.bounce) {
    batch_unregister_pinned(g.bounce);
    (void)hipHostFree(g.bounce);
  }
  if (g.dev) (void)hipFree(g.dev);
  if (g.member) {
    for (int i = 0; i < g.n_alloc; ++i) {
      delete[] g.member[i].copies;
      delete static_cast<std::vector<BounceOut>*>(g.member[i].bounce_out);
    }
    delete[] g.member;
  }
  g.host = g.dev = g.bounce = nullptr;
  g.member = nullptr;
}

static std::vector<BounceOut>& bounce_out_list() {
  static thread_local std::vector<BounceOut> solo;
  BatchGroup* g = tls_batch_group;
  if (!g) return solo;
  return *static_cast<std::vector<BounceOut>*>(g->member[tls_batch_member].bounce_out);
}

BatchGroup& batch_current_group() {
  if (tls_batch_group) return *tls_batch_group;
  if (!tls_solo.g) {   // a thread outside any batch: a group of one that launches on the caller's stream
    BatchGroup* g = new BatchGroup();
    g->solo = true;
    batch_group_init(*g, 1, true);
    g->member[0].active = true;
    tls_solo.g = g;
  }
  return *tls_solo.g;
}

static void stream_wait(hipStream_t s) {
  for (;;) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) batch_check_hip(e, "hipStreamQuery");
#if defined(__x86_64__)
    for (int k = 0; k < 32; ++k) __builtin_ia32_pause();
#endif
  }
}

// Members are FIBERS: a handful of OS threads each run several members, switching at every rendezvous.  (One spinning
// thread per member was the first design: beyond ~32 members the waiting threads alone exhaust the CPU time a
// container is entitled to and the batch time explodes - measured on the MI355X boxes, tools/small_proof_batch.py.)
// Switching between a worker's scheduler and its member fibers: ucontext.  glibc's swapcontext saves and restores the signal mask -
// one rt_sigprocmask system call per switch, on a lock (sighand->siglock) that all threads of the process share - and that was the
// suspect for the collapse of several groups' throughput with the number of worker threads (end of round 6).  Measured with a switch
// of our own (-DLMN_BATCH_FIBER_ASM, x86-64: callee-saved registers, the two floating-point control words, the stack pointer; byte-
// identical proofs on the emulated and the GPU batch library): no difference - 3 groups of 192 with 8 / 16 / 24 workers each 42 / 24 /
// 13 k proofs/s either way (gpurun_out/r11m).  Not the cause; the experiment build stays.
#if defined(__x86_64__) && defined(LMN_BATCH_FIBER_ASM)
#define LMN_FIBER_ASM 1
struct FiberCtx {
  void* sp = nullptr;
};
extern "C" void lmn_fiber_switch(FiberCtx* from, FiberCtx* to);
#if !defined(__HIP_DEVICE_COMPILE__)
asm(R"(
  .text
  .p2align 4
  .globl lmn_fiber_switch
  .type lmn_fiber_switch,@function
lmn_fiber_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
  .size lmn_fiber_switch,.-lmn_fiber_switch
)");
#endif
// a context that starts `entry` on the given stack at its first switch; `entry` must never return
static void fiber_ctx_prepare(FiberCtx& c, void* stack, size_t bytes, void (*entry)()) {
  uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
  uint64_t* sp = reinterpret_cast<uint64_t*>(top - 72);   // control words | r15 r14 r13 r12 rbx rbp | entry | (no caller)
  sp[0] = 0x1f80ull | (0x037full << 32);                   // default MXCSR, default x87 control word
  for (int k = 1; k <= 6; ++k) sp[k] = 0;
  sp[7] = (uint64_t)(uintptr_t)entry;                      // popped by the switch's ret: rsp = top - 8, as right after a call
  sp[8] = 0;                                               // where a caller's return address would be: the unwinder stops here
  c.sp = sp;
}
#else
typedef ucontext_t FiberCtx;
#endif

struct BatchFiber {
  FiberCtx ctx;
  void* stack = nullptr;
  lmn_batch* batch = nullptr;
  int member = 0;
  bool done = true;
  bool waiting = false;
  uint64_t waiting_gen = 0;
#ifdef LMN_BATCH_ASAN
  void* fake_stack = nullptr;
#endif
#ifdef LMN_BATCH_TSAN
  void* tsan_fiber = nullptr;
#endif
};
static thread_local FiberCtx tls_sched_ctx;
static thread_local BatchFiber* tls_fiber = nullptr;
#ifdef LMN_BATCH_ASAN
static thread_local void* tls_sched_fake = nullptr;
static thread_local const void* tls_sched_bottom = nullptr;
static thread_local size_t tls_sched_size = 0;
#endif
#ifdef LMN_BATCH_TSAN
static thread_local void* tls_sched_tsan = nullptr;
#endif
// member fiber -> its worker thread's scheduler (`last`: the fiber is finished) and back
static void fiber_yield(BatchFiber* f) {
#ifdef LMN_BATCH_ASAN
  __sanitizer_start_switch_fiber(&f->fake_stack, tls_sched_bottom, tls_sched_size);
#endif
#ifdef LMN_BATCH_TSAN
  __tsan_switch_to_fiber(tls_sched_tsan, 0);
#endif
#ifdef LMN_FIBER_ASM
  lmn_fiber_switch(&f->ctx, &tls_sched_ctx);
#else
  swapcontext(&f->ctx, &tls_sched_ctx);
#endif
#ifdef LMN_BATCH_ASAN
  __sanitizer_finish_switch_fiber(f->fake_stack, &tls_sched_bottom, &tls_sched_size);
#endif
}

static void wait_generation(BatchGroup& g, uint64_t gen) {
  if (tls_fiber) {
    BatchFiber* f = tls_fiber;
    f->waiting = true;
    f->waiting_gen = gen;
    while (g.generation.load(std::memory_order_acquire) == gen) fiber_yield(f);
    f->waiting = false;
    return;
  }
  for (uint32_t spins = 0; g.generation.load(std::memory_order_acquire) == gen; ++spins) {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if (spins > (1u << 22)) std::this_thread::yield();
  }
}

// one workgroup per entry: copy (or fill) up to BATCH_COPY_CHUNK bytes; the entry list itself sits in page-locked host memory
__global__ void k_batch_copy(const BatchCopy* __restrict__ entries) {
  const BatchCopy c = entries[blockIdx.x];
  if (c.src == 0) {
    unsigned char* d = reinterpret_cast<unsigned char*>(c.dst);
    for (uint32_t i = threadIdx.x; i < c.bytes; i += blockDim.x) d[i] = (unsigned char)c.fill;
  } else if (((c.dst | c.src | c.bytes) & 15u) == 0) {
    const uint4* sp = reinterpret_cast<const uint4*>(c.src);
    uint4* dp = reinterpret_cast<uint4*>(c.dst);
    for (uint32_t i = threadIdx.x; i < c.bytes / 16; i += blockDim.x) dp[i] = sp[i];
  } else if (((c.dst | c.src | c.bytes) & 3u) == 0) {
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(c.src);
    uint32_t* dp = reinterpret_cast<uint32_t*>(c.dst);
    for (uint32_t i = threadIdx.x; i < c.bytes / 4; i += blockDim.x) dp[i] = sp[i];
  } else {
    const unsigned char* sp = reinterpret_cast<const unsigned char*>(c.src);
    unsigned char* dp = reinterpret_cast<unsigned char*>(c.dst);
    for (uint32_t i = threadIdx.x; i < c.bytes; i += blockDim.x) dp[i] = sp[i];
  }
}

// `bytes` of the current epoch's launch-table half; called by the last arriver only (every member has added the same
// amounts to its own tbl_off, so the offsets agree)
static size_t table_alloc(BatchGroup& g, size_t& off, uint64_t epoch, size_t bytes) {
  const size_t at = (off + 255) & ~(size_t)255;
  if (at + bytes + g.flush_top[epoch & 1] > g.half_bytes)
    throw LmnError(LMN_ERR_INTERNAL, "batch table memory exhausted between two synchronisations");
  off = at + bytes;
  return (epoch & 1) * g.half_bytes + at;
}

// all members' pending transfers in one launch (they precede the rendezvous in every member's program order)
static void flush_copies(BatchGroup& g, BatchMember& me) {
  size_t n = 0;
  for (int i = 0; i < g.slots; ++i)
    if (g.member[i].present) n += g.member[i].n_copies;
  if (n == 0) return;
  // the list is taken from the TOP of this epoch's half of the table memory, so that the members' deterministic
  // launch-table offsets (growing from the bottom) are not disturbed
  const size_t bytes = (n * sizeof(BatchCopy) + 255) & ~(size_t)255;
  size_t& top = g.flush_top[me.epoch & 1];
  if (me.tbl_off + top + bytes + 4096 > g.half_bytes)
    throw LmnError(LMN_ERR_INTERNAL, "batch table memory exhausted between two synchronisations");
  top += bytes;
  unsigned char* at = g.host + (me.epoch & 1) * g.half_bytes + g.half_bytes - top;
  BatchCopy* list = reinterpret_cast<BatchCopy*>(at);
  size_t k = 0;
  for (int i = 0; i < g.slots; ++i) {
    BatchMember& m = g.member[i];
    if (!m.present) continue;
    memcpy(list + k, m.copies, m.n_copies * sizeof(BatchCopy));
    k += m.n_copies;
    m.n_copies = 0;
  }
  hipLaunchKernelGGL(k_batch_copy, dim3((unsigned)n), dim3(256), 0, g.stream, (const BatchCopy*)list);
  batch_check_hip(hipGetLastError(), "batched copy");
  g.copy_launches++;
}

static void push_copy(BatchMember& m, uint64_t dst, uint64_t src, size_t n, uint32_t fill) {
  for (size_t o = 0; o < n; o += BATCH_COPY_CHUNK) {
    if (m.n_copies == m.copy_cap) throw LmnError(LMN_ERR_INTERNAL, "too many pending batch transfers");
    const uint32_t len = (uint32_t)(n - o < BATCH_COPY_CHUNK ? n - o : BATCH_COPY_CHUNK);
    m.copies[m.n_copies++] = BatchCopy{dst + o, src ? src + o : 0, len, fill};
  }
}

// A transfer that cannot be batched (a pageable buffer larger than the bounce slot) is issued by the member itself, at
// once.  Its earlier transfers are still waiting in the member's list for the next rendezvous: issue those first, in
// order, so that program order holds (e.g. a device-to-device copy followed by the download of the same buffer).
static void flush_member_direct(BatchGroup& g, BatchMember& m) {
  for (uint32_t i = 0; i < m.n_copies; ++i) {
    const BatchCopy& c = m.copies[i];
    if (c.src == 0)
      batch_check_hip(hipMemsetAsync(reinterpret_cast<void*>(c.dst), (int)(c.fill & 0xffu), c.bytes, g.stream), "hipMemsetAsync");
    else
      batch_check_hip(hipMemcpyAsync(reinterpret_cast<void*>(c.dst), reinterpret_cast<const void*>(c.src), c.bytes,
                                     hipMemcpyDefault, g.stream), "hipMemcpyAsync");
  }
  m.n_copies = 0;
}

bool batch_copy(void* dst, const void* src, size_t n, int dir) {
  BatchGroup* gp = tls_batch_group;
  if (!gp) return false;                 // outside a batch: the caller's own hipMemcpyAsync
  if (n == 0) return true;
  BatchGroup& g = *gp;
  BatchMember& m = g.member[tls_batch_member];
  uint64_t d = (uint64_t)(uintptr_t)dst, s = (uint64_t)(uintptr_t)src;
  auto bounce = [&](size_t bytes) -> unsigned char* {
    const size_t at = (m.bounce_off + 63) & ~(size_t)63;
    if (bytes > BATCH_BOUNCE_MAX || at + bytes > g.bounce_bytes) return nullptr;
    m.bounce_off = at + bytes;
    return g.bounce + ((size_t)tls_batch_member * 2 + (m.epoch & 1)) * g.bounce_bytes + at;
  };
  if (dir == 0 && !pinned_device_address(src, n, &s)) {          // pageable source: through a page-locked slot
    unsigned char* slot = bounce(n);
    if (!slot) {
      flush_member_direct(g, m);
      SpinGuard lk(g.lock);
      g.direct_copies++;
      return false;
    }
    memcpy(slot, src, n);
    s = (uint64_t)(uintptr_t)slot;
  } else if (dir == 1 && !pinned_device_address(dst, n, &d)) {   // pageable destination: copied out after the wait
    unsigned char* slot = bounce(n);
    if (!slot) {
      flush_member_direct(g, m);
      SpinGuard lk(g.lock);
      g.direct_copies++;
      return false;
    }
    d = (uint64_t)(uintptr_t)slot;
    bounce_out_list().push_back({dst, slot, n});
  }
  if (dir == 3)
    push_copy(m, d, 0, n, (uint32_t)(uintptr_t)src);
  else
    push_copy(m, d, s, n, 0);
  return true;
}

static void throw_if_failed(BatchGroup& g) {
  const int f = g.failed.load();
  if (f == 1)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "lmn_batch_prove: the pies of a batch must have identical shapes (the members' "
                                             "kernel sequences diverged)");
  if (f) throw LmnError(LMN_ERR_INTERNAL, "lmn_batch_prove: a batched launch failed");
}

// the last arriver (or a member that leaves while the others wait) completes the rendezvous for everybody.
// `me`: a member that is present (its view of kind / offsets is the group's).  g.lock is NOT held: nobody else moves.
static void complete_rendezvous(BatchGroup& g, BatchMember& me) {
  try {
    if (!g.failed.load()) {
      // every present member must want the same thing
      for (int i = 0; i < g.slots && !g.failed.load(); ++i) {
        BatchMember& m = g.member[i];
        if (!m.present) continue;
        if (m.kind != me.kind || (me.kind == 1 && (m.fn != me.fn || m.bx != me.bx || m.by != me.by || m.smem != me.smem ||
                                                   m.slot_bytes != me.slot_bytes)))
          g.failed.store(1);
      }
    }
    if (!g.failed.load()) {
      flush_copies(g, me);
      if (me.kind == 1) {
        g.region = me.region;
        g.grid = dim3(0, 0, 1);
        for (int i = 0; i < g.slots; ++i) {
          BatchMember& m = g.member[i];
          BatchSlotHdr* h = reinterpret_cast<BatchSlotHdr*>(g.host + g.region + (size_t)i * me.slot_bytes);
          if (!m.present) {
            h->active = 0;           // a member that left: its slot holds stale bytes
            continue;
          }
          // grids may differ in extent (the decommitment gather has one workgroup per queried run): launch the largest,
          // the trampoline drops the workgroups outside a member's own grid
          if (m.gx > g.grid.x) g.grid.x = m.gx;
          if (m.gy > g.grid.y) g.grid.y = m.gy;
        }
        g.block = dim3(me.bx, me.by, 1);
        g.smem = me.smem;
        g.slot_bytes = me.slot_bytes;
        me.do_launch(g);
        g.launches++;
      } else if (me.kind == 2) {
        stream_wait(g.stream);
        g.flush_top[me.epoch & 1] = 0;
        g.syncs++;
      }
    }
  } catch (...) {
    g.failed.store(2);
  }
  for (int i = 0; i < g.slots; ++i) g.member[i].present = false;
  g.state.store(g.state.load(std::memory_order_relaxed) & ~0xffffffffull, std::memory_order_release);   // nobody else moves now
  g.generation.fetch_add(1, std::memory_order_release);
}

// arrive; returns after the rendezvous is complete
static inline uint64_t now_ns() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void arrive(BatchGroup& g, BatchMember& me) {
  const uint64_t gen = g.generation.load(std::memory_order_acquire);
  const uint64_t t = now_ns();
  if (me.t_resume) me.ns_busy += t - me.t_resume;
  me.t_arrive = t;
  me.present = true;
  const uint64_t st = g.state.fetch_add(1, std::memory_order_acq_rel) + 1;
  if ((uint32_t)st >= (uint32_t)(st >> 32)) {
    uint64_t first = t;
    for (int i = 0; i < g.slots; ++i)
      if (g.member[i].present && g.member[i].t_arrive < first) first = g.member[i].t_arrive;
    g.ns_skew += t - first;                // first to last arrival: imbalance of the members' own host work
    complete_rendezvous(g, me);
    g.ns_leader += now_ns() - t;           // launch / wait issued for everybody
  } else {
    wait_generation(g, gen);
  }
  me.t_resume = now_ns();
  throw_if_failed(g);
}

unsigned char* batch_launch_begin(BatchGroup& g, const void* fn, void (*do_launch)(BatchGroup&), dim3 grid, dim3 block,
                                  size_t smem, size_t slot_bytes, hipStream_t stream) {
  throw_if_failed(g);
  BatchMember& me = g.member[tls_batch_group ? tls_batch_member : 0];
  if (g.solo) {
    g.stream = stream;
    if (((me.tbl_off + 255) & ~(size_t)255) + slot_bytes + 4096 > g.half_bytes) {   // a solo thread may simply drain
      stream_wait(g.stream);
      me.tbl_off = 0;
    }
  }
  const size_t region = table_alloc(g, me.tbl_off, me.epoch, slot_bytes * (size_t)g.slots);
  me.region = region;   // (every member computes the same value; the group's copy is set by the last arriver: TSan, round 6)
  me.kind = 1;
  me.fn = fn;
  me.do_launch = do_launch;
  me.gx = grid.x;
  me.gy = grid.y;
  me.bx = block.x;
  me.by = block.y;
  me.smem = smem;
  me.slot_bytes = slot_bytes;
  return g.host + region + (size_t)(tls_batch_group ? tls_batch_member : 0) * slot_bytes;
}

void batch_launch_end(BatchGroup& g) { arrive(g, g.member[tls_batch_group ? tls_batch_member : 0]); }

void batch_sync(hipStream_t s) {
  BatchGroup& g = batch_current_group();
  BatchMember& me = g.member[tls_batch_group ? tls_batch_member : 0];
  if (g.solo) g.stream = s;
  throw_if_failed(g);
  me.kind = 2;
  arrive(g, me);
  me.epoch++;
  me.tbl_off = 0;
  me.bounce_off = 0;
  // device -> pageable host transfers of this member: their slots live in the bounce half of the epoch just finished,
  // which this member does not write again before it has passed the NEXT wait
  auto& outs = bounce_out_list();
  for (auto& b : outs) memcpy(b.dst, b.slot, b.bytes);
  outs.clear();
}

void batch_leave() {
  BatchGroup* g = tls_batch_group;
  bounce_out_list().clear();
  if (!g) return;
  BatchMember& me = g->member[tls_batch_member];
  if (me.t_resume) me.ns_busy += now_ns() - me.t_resume;
  {
    SpinGuard lk(g->lock);
    g->ns_busy += me.ns_busy;
  }
  me.ns_busy = 0;
  me.t_resume = 0;
  me.active = false;
  me.present = false;
  me.n_copies = 0;
  const uint64_t st = g->state.fetch_sub(1ull << 32, std::memory_order_acq_rel) - (1ull << 32);
  if ((st >> 32) > 0 && (uint32_t)st >= (uint32_t)(st >> 32)) {   // the others were all waiting for this member
    for (int i = 0; i < g->slots; ++i)
      if (g->member[i].present) {
        complete_rendezvous(*g, g->member[i]);
        break;
      }
  }
}

}  // namespace lmn

// ------------------------------------------------------------------------------------------------ C entry points
struct lmn_batch {
  int device = 0;
  uint32_t slots = 0, n_threads = 1;
  lmn::BatchGroup group;
  std::vector<lmn::Context*> ctx;
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  uint64_t job_id = 0;
  bool quit = false;
  uint32_t n = 0, pending = 0;
  const lmn_table* const* tables = nullptr;
  size_t n_tables = 0;
  const lmn_settings* settings = nullptr;
  std::vector<std::vector<uint8_t>> out;
  std::vector<int> rc;
  std::vector<std::string> err;
  std::string last_error;
  std::mutex call_mu;   // one lmn_batch_prove at a time
};

static void run_member(lmn_batch* b, uint32_t me) {
  int rc = LMN_OK;
  std::string msg;
  try {
    b->out[me] = b->ctx[me]->prove(b->tables[me], b->n_tables, b->settings);
  } catch (const LmnError& e) {
    rc = (e.code == -100 || (e.code <= -1 && e.code >= -10)) ? e.code : LMN_ERR_INTERNAL;
    msg = e.what();
  } catch (const std::bad_alloc&) {
    rc = LMN_ERR_OUT_OF_MEMORY;
    msg = "host allocation failed";
  } catch (const std::exception& e) {
    rc = LMN_ERR_INTERNAL;
    msg = e.what();
  }
  lmn::batch_leave();   // normal end or failure: the remaining members no longer wait for this one
  b->rc[me] = rc;
  b->err[me] = msg;
}

static void fiber_entry() {
  lmn::BatchFiber* f = lmn::tls_fiber;
#ifdef LMN_BATCH_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &lmn::tls_sched_bottom, &lmn::tls_sched_size);
#endif
  run_member(f->batch, (uint32_t)f->member);
  f->done = true;
  // returning resumes uc_link = the scheduler
#ifdef LMN_BATCH_ASAN
  __sanitizer_start_switch_fiber(nullptr, lmn::tls_sched_bottom, lmn::tls_sched_size);
#endif
#ifdef LMN_BATCH_TSAN
  // not by returning: this frame's function-exit event would be recorded after the switch, i.e. popped from the SCHEDULER's
  // shadow stack - one entry too many per finished member, until the worker thread's own frames underflow it
  __tsan_switch_to_fiber(lmn::tls_sched_tsan, 0);
#endif
#ifdef LMN_FIBER_ASM
  lmn::lmn_fiber_switch(&f->ctx, &lmn::tls_sched_ctx);   // never resumed (the entry of a prepared context must not return)
  __builtin_unreachable();
#elif defined(LMN_BATCH_TSAN)
  setcontext(&lmn::tls_sched_ctx);
#endif
}

constexpr size_t BATCH_FIBER_STACK = 1u << 20;
// fiber stacks come from mmap with an inaccessible page below them: running off the end of a stack faults instead of
// overwriting whatever the heap had put next to a malloc'ed block
static void* fiber_stack_alloc() {
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  void* p = mmap(nullptr, BATCH_FIBER_STACK + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
  if (p == MAP_FAILED) throw std::bad_alloc();
  (void)mprotect(p, page, PROT_NONE);
  return static_cast<char*>(p) + page;
}
static void fiber_stack_free(void* stack) {
  if (!stack) return;
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  (void)munmap(static_cast<char*>(stack) - page, BATCH_FIBER_STACK + page);
}

// worker w of T runs members w, w + T, w + 2T, ... of every batch as fibers
static void batch_worker(lmn_batch* b, uint32_t w) {
  uint64_t seen = 0;
  (void)hipSetDevice(b->device);
  std::vector<lmn::BatchFiber> fibers;
  for (;;) {
    uint32_t n;
    {
      std::unique_lock<std::mutex> lk(b->m);
      b->cv_job.wait(lk, [&] { return b->quit || b->job_id != seen; });
      if (b->quit) break;
      seen = b->job_id;
      n = b->n;
    }
    uint32_t mine = 0;
    for (uint32_t m = w; m < n; m += b->n_threads) ++mine;
    if (mine == 0) continue;
    if (fibers.size() < mine) fibers.resize(mine);
    lmn::tls_batch_group = &b->group;
    uint32_t k = 0;
    for (uint32_t m = w; m < n; m += b->n_threads, ++k) {
      lmn::BatchFiber& f = fibers[k];
      if (!f.stack) f.stack = fiber_stack_alloc();
#ifdef LMN_FIBER_ASM
      lmn::fiber_ctx_prepare(f.ctx, f.stack, BATCH_FIBER_STACK, fiber_entry);
#else
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack;
      f.ctx.uc_stack.ss_size = BATCH_FIBER_STACK;
      f.ctx.uc_link = &lmn::tls_sched_ctx;
#endif
      f.batch = b;
      f.member = (int)m;
      f.done = false;
      f.waiting = false;
#ifndef LMN_FIBER_ASM
      makecontext(&f.ctx, fiber_entry, 0);
#endif
#ifdef LMN_BATCH_TSAN
      lmn::tls_sched_tsan = __tsan_get_current_fiber();
      if (f.tsan_fiber) __tsan_destroy_fiber(f.tsan_fiber);
      f.tsan_fiber = __tsan_create_fiber(0);
#endif
    }
    uint32_t left = mine;
    uint32_t idle_pauses = 1u;
    while (left) {
      bool progress = false;
      for (uint32_t i = 0; i < mine; ++i) {
        lmn::BatchFiber& f = fibers[i];
        if (f.done) continue;
        if (f.waiting && b->group.generation.load(std::memory_order_acquire) == f.waiting_gen) continue;
        lmn::tls_fiber = &f;
        lmn::tls_batch_member = f.member;
#ifdef LMN_BATCH_ASAN
        __sanitizer_start_switch_fiber(&lmn::tls_sched_fake, f.stack, BATCH_FIBER_STACK);
#endif
#ifdef LMN_BATCH_TSAN
        __tsan_switch_to_fiber(f.tsan_fiber, 0);
#endif
#ifdef LMN_FIBER_ASM
        lmn::lmn_fiber_switch(&lmn::tls_sched_ctx, &f.ctx);
#else
        swapcontext(&lmn::tls_sched_ctx, &f.ctx);
#endif
#ifdef LMN_BATCH_ASAN
        __sanitizer_finish_switch_fiber(lmn::tls_sched_fake, nullptr, nullptr);
#endif
        lmn::tls_fiber = nullptr;
        progress = true;
        if (f.done) --left;
      }
      // nothing runnable: back off (1 .. 64 pauses) - a worker that polls at full speed takes issue slots from the hardware
      // thread next to it, which may be the one whose member everybody is waiting for
      if (!progress) {
#if defined(__x86_64__)
        for (uint32_t k = 0; k < idle_pauses; ++k) __builtin_ia32_pause();
#endif
        if (idle_pauses < 64u) idle_pauses *= 2u;
      } else {
        idle_pauses = 1u;
      }
    }
    lmn::tls_batch_group = nullptr;
    {
      std::lock_guard<std::mutex> lk(b->m);
      b->pending -= mine;
      if (b->pending == 0) b->cv_done.notify_all();
    }
  }
  for (auto& f : fibers) {
#ifdef LMN_BATCH_TSAN
    if (f.tsan_fiber) __tsan_destroy_fiber(f.tsan_fiber);
#endif
    fiber_stack_free(f.stack);
  }
}

extern "C" {

int lmn_batch_create(int device, const lmn_config* cfg, uint32_t slots, lmn_batch** out) {
  if (!out || slots == 0 || slots > 256) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  lmn_config c;
  if (cfg) c = *cfg; else lmn_default_config(&c);
  lmn_batch* b = new lmn_batch();
  try {
    b->device = device;
    b->slots = slots;
    for (uint32_t i = 0; i < slots; ++i) b->ctx.push_back(new lmn::Context(device, c));
    LMN_HIP_CHECK(hipSetDevice(device));
    LMN_HIP_CHECK(hipStreamCreateWithFlags(&b->group.stream, hipStreamNonBlocking));
    lmn::batch_group_init(b->group, (int)slots);
    for (auto* x : b->ctx) x->adopt_stream(b->group.stream);
    b->out.resize(slots);
    b->rc.assign(slots, 0);
    b->err.resize(slots);
    const char* env_t = getenv("LMN_BATCH_THREADS");
    b->n_threads = std::max<uint32_t>(1u, std::min<uint32_t>(slots, env_t ? (uint32_t)atoi(env_t) : 8u));
    for (uint32_t i = 0; i < b->n_threads; ++i) b->workers.emplace_back(batch_worker, b, i);
  } catch (const LmnError& e) {
    const int code = e.code;
    lmn_batch_destroy(b);
    return (code == -100 || (code <= -1 && code >= -10)) ? code : LMN_ERR_INTERNAL;
  } catch (...) {
    lmn_batch_destroy(b);
    return LMN_ERR_INTERNAL;
  }
  *out = b;
  return LMN_OK;
}

void lmn_batch_destroy(lmn_batch* b) {
  if (!b) return;
  {
    std::lock_guard<std::mutex> lk(b->m);
    b->quit = true;
  }
  b->cv_job.notify_all();
  for (auto& t : b->workers)
    if (t.joinable()) t.join();
  (void)hipSetDevice(b->device);
  if (b->group.stream) (void)hipStreamSynchronize(b->group.stream);
  for (auto* x : b->ctx) delete x;
  lmn::batch_group_release(b->group);
  if (b->group.stream) (void)hipStreamDestroy(b->group.stream);
  delete b;
}

const char* lmn_batch_last_error(const lmn_batch* b) { return b ? b->last_error.c_str() : "null batch"; }

int lmn_batch_prove(lmn_batch* b, uint32_t n, const lmn_table* const* tables, size_t n_tables, const lmn_settings* settings,
                    uint8_t** proofs, size_t* lens, int* rcs) {
  if (!b || !tables || !proofs || !lens || n == 0 || n > b->slots || n_tables == 0) return LMN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> call(b->call_mu);
  for (uint32_t i = 0; i < n; ++i) {
    proofs[i] = nullptr;
    lens[i] = 0;
    if (rcs) rcs[i] = LMN_OK;
    if (!tables[i]) return LMN_ERR_INVALID_ARGUMENT;
    for (size_t t = 0; t < n_tables; ++t)
      if (tables[i][t].kind != tables[0][t].kind || tables[i][t].n_rows != tables[0][t].n_rows) {
        b->last_error = "lmn_batch_prove: the pies of a batch must have the same table kinds and row counts";
        return LMN_ERR_INVALID_ARGUMENT;
      }
  }
  {
    // Per-context set-up that only SOME members would do (a context that has not proved this shape yet builds its
    // twiddle tables: uploads and waits the other members do not make) happens here, on the calling thread, before the
    // members start in lock-step: a batch may use more slots, or a larger shape, than the batches before it.
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->group.stream);   // a failed batch may have left work behind
    try {
      for (uint32_t i = 0; i < n; ++i) b->ctx[i]->prepare_for(tables[i], n_tables);
    } catch (const LmnError& e) {
      b->last_error = e.what();
      return (e.code == -100 || (e.code <= -1 && e.code >= -10)) ? e.code : LMN_ERR_INTERNAL;
    } catch (const std::bad_alloc&) {   // (host-built twiddle vectors: nothing may cross the extern "C" boundary)
      b->last_error = "lmn_batch_prove: out of host memory while preparing the contexts";
      return LMN_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
      b->last_error = e.what();
      return LMN_ERR_INTERNAL;
    } catch (...) {
      b->last_error = "lmn_batch_prove: unknown exception while preparing the contexts";
      return LMN_ERR_INTERNAL;
    }
  }
  {
    std::lock_guard<std::mutex> lk(b->m);
    b->group.slots = (int)n;
    b->group.state.store((uint64_t)n << 32);
    b->group.failed.store(0);
    b->group.flush_top[0] = b->group.flush_top[1] = 0;
    for (uint32_t i = 0; i < b->slots; ++i) {
      lmn::BatchMember& m = b->group.member[i];
      m.active = i < n;
      m.present = false;
      m.n_copies = 0;
      m.tbl_off = 0;
      m.epoch = 0;
      m.bounce_off = 0;
    }
    b->n = n;
    b->pending = n;
    b->tables = tables;
    b->n_tables = n_tables;
    b->settings = settings;
    b->job_id++;
  }
  b->cv_job.notify_all();
  {
    std::unique_lock<std::mutex> lk(b->m);
    b->cv_done.wait(lk, [&] { return b->pending == 0; });
  }
  int first = LMN_OK;
  for (uint32_t i = 0; i < n; ++i) {
    if (rcs) rcs[i] = b->rc[i];
    if (b->rc[i] != LMN_OK) {
      if (first == LMN_OK) {
        first = b->rc[i];
        b->last_error = b->err[i];
      }
      continue;
    }
    uint8_t* p = (uint8_t*)malloc(b->out[i].size() ? b->out[i].size() : 1);
    if (!p) return LMN_ERR_OUT_OF_MEMORY;
    memcpy(p, b->out[i].data(), b->out[i].size());
    proofs[i] = p;
    lens[i] = b->out[i].size();
  }
  return first;
}

uint64_t lmn_batch_counter(const lmn_batch* b, int which) {
  if (!b) return 0;
  switch (which) {
    case 0: return b->group.launches;
    case 1: return b->group.syncs;
    case 2: return b->group.copy_launches;
    case 3: return b->group.direct_copies;
    case 4: return b->group.ns_skew;
    case 5: return b->group.ns_leader;
    default: return b->group.ns_busy;
  }
}

}  // extern "C"
#endif  // LMN_BATCH
