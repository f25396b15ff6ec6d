// Lock-step batching of SMALL proofs (LMN_BATCH build = libluminair_hip_batch.so; include/luminair_hip.h lmn_batch_*).
//
// A proof whose tables have at most a few thousand rows - BASELINE config 4
// (/root/reference/examples/black-schole-nn/src/main.rs:61-103) and the reference's own benchmark shape
// (/root/reference/crates/graph/benches/ops.rs:92-166, 32x32 tensors) - is bound by its ~60 kernel launches and 6 host
// round trips, not by the work in them.  This build runs B proofs of IDENTICAL SHAPE in lock-step: B host threads execute
// the unchanged Context::prove on their own contexts (own arena, own transcript), all on ONE stream, and every
// LMN_LAUNCH is a rendezvous at which the last arriver launches a single trampoline kernel whose blockIdx.z selects the
// member: each member's kernel arguments travel through a device-side table instead of the kernel-argument segment.  The
// kernel bodies are the product's own kernels compiled as device functions (LMN_KERNEL = __device__ void), so a batched
// proof is byte-identical to lmn_prove's by construction (tests/test_gpu_parity.py checks it).  Host synchronisations are
// rendezvous too: one stream wait per batch instead of one per proof.
#pragma once
#ifdef LMN_BATCH
#ifndef LMN_EMU
#include <hip/hip_runtime.h>
#endif   // (the emulation build's stand-ins for the HIP calls used here: platform.h)

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>

namespace lmn {

struct BatchSlotHdr {
  uint32_t active, gx, gy, pad;
};
constexpr size_t BATCH_SLOT_ALIGN = 16;

// kernel parameters of one member, laid out as a plain aggregate in the table
template <class... P>
struct BatchPack;
template <>
struct BatchPack<> {};
template <class H, class... T>
struct BatchPack<H, T...> {
  H h;
  BatchPack<T...> t;
};
template <class H, class... T, class A, class... R>
inline void batch_pack_fill(BatchPack<H, T...>& p, A&& a, R&&... r) {
  p.h = H(a);
  if constexpr (sizeof...(T) > 0) batch_pack_fill(p.t, std::forward<R>(r)...);
}
template <auto Fn, class... Got>
__device__ __forceinline__ void batch_pack_apply(const BatchPack<>&, const Got&... got) {
  Fn(got...);
}
template <auto Fn, class H, class... T, class... Got>
__device__ __forceinline__ void batch_pack_apply(const BatchPack<H, T...>& p, const Got&... got) {
  batch_pack_apply<Fn>(p.t, got..., p.h);
}

template <auto Fn, class... P>
__global__ void lmn_batch_tramp(const unsigned char* table, uint32_t slot_bytes) {
  const unsigned char* slot = table + (size_t)blockIdx.z * slot_bytes;
  const BatchSlotHdr* h = reinterpret_cast<const BatchSlotHdr*>(slot);
  if (!h->active || blockIdx.x >= h->gx || blockIdx.y >= h->gy) return;   // block-uniform
  if constexpr (sizeof...(P) == 0) {
    Fn();
  } else {
    batch_pack_apply<Fn>(*reinterpret_cast<const BatchPack<P...>*>(slot + sizeof(BatchSlotHdr)));
  }
}

// Small transfers of the members are not issued one by one (B members x ~16 hipMemcpyAsync per proof on one stream
// serialise inside the runtime): each becomes an entry of a list that the last arriver of the next rendezvous hands to
// ONE copy kernel.  Host memory is read / written by that kernel directly when it is page-locked; pageable host
// memory goes through a page-locked bounce slot of the group's table memory.
struct BatchCopy {
  uint64_t dst, src;   // device-visible addresses
  uint32_t bytes;
  uint32_t fill;       // bytes of `value` to store instead of copying when src == 0
};
constexpr uint32_t BATCH_COPY_CHUNK = 16u << 10;   // one workgroup per chunk

// per-member state of the rendezvous in progress; written by the member alone, read by the last arriver
struct alignas(128) BatchMember {
  const void* fn = nullptr;      // trampoline instantiation the member wants to launch
  void (*do_launch)(struct BatchGroup&) = nullptr;
  size_t tbl_off = 0;            // the member's own view of the launch-table offset (identical on all members)
  uint64_t epoch = 0;            // synchronisations the member has passed
  uint32_t gx = 0, gy = 0, bx = 0, by = 0;
  size_t smem = 0, slot_bytes = 0;
  size_t region = 0;             // the member's own view of the launch's table region (identical on all members)
  int kind = 0;                  // 1 launch, 2 wait
  bool present = false;          // arrived at the rendezvous in progress
  uint64_t t_arrive = 0, t_resume = 0, ns_busy = 0;   // instrumentation: the member's own host time between rendezvous
  bool active = false;           // still part of the group
  BatchCopy* copies = nullptr;   // the member's pending transfers
  size_t n_copies = 0, copy_cap = 0;
  size_t bounce_off = 0;         // bump offset inside the member's page-locked bounce area (reset every epoch)
  void* bounce_out = nullptr;    // device -> pageable host transfers to finish after the next wait
};

// one group = the members of a lock-step batch (a thread outside any group is a group of one)
struct BatchGroup {
  int slots = 1;                 // z extent of every launch (members that left keep an inactive slot)
  int n_alloc = 1;               // members allocated (slots of the largest batch)
  hipStream_t stream{};
  bool solo = false;
  // (members still taking part) << 32 | (members arrived at the rendezvous in progress): one wait-free fetch_add per
  // arrival - lock-step members arrive within a microsecond of each other, a lock here is a convoy
  // (own cache line: every arrival writes it - next to `generation`, which every waiting worker thread polls, each arrival
  // took the line away from all of them: the more worker threads in the process, the slower the rendezvous; end of round 6)
  alignas(128) std::atomic<uint64_t> state{1ull << 32};
  alignas(128) std::atomic_flag lock = ATOMIC_FLAG_INIT;   // counters only
  // written once per rendezvous by the last arriver, polled by everybody else
  alignas(128) std::atomic<uint64_t> generation{0};
  std::atomic<int> failed{0};    // members diverged (different kernel / shape): every member throws at its next rendezvous
  alignas(128)
  // Table memory, mirrored in page-locked host and device memory, in two halves used by alternating synchronisation
  // epochs.  Every member executes the same launch sequence, so each computes the table offsets itself (tbl_off):
  // no shared allocator on the launch path.
  unsigned char* host = nullptr;
  unsigned char* dev = nullptr;
  size_t half_bytes = 0;         // launch tables of one epoch
  size_t flush_top[2] = {0, 0};  // copy lists of the epoch, taken from the top of its half
  uint64_t epoch = 0;
  unsigned char* bounce = nullptr;   // page-locked: slots x 2 halves x bounce_bytes
  size_t bounce_bytes = 0;
  BatchMember* member = nullptr;
  void (*do_launch)(BatchGroup&) = nullptr;
  dim3 grid, block;
  size_t smem = 0, region = 0, slot_bytes = 0;
  uint64_t launches = 0, syncs = 0, copy_launches = 0, direct_copies = 0;
  uint64_t ns_skew = 0, ns_leader = 0, ns_busy = 0;   // where a batch's host time goes (lmn_batch_counter 4, 5, 6)
};
// a transfer of a group member: true = taken over (runs before the next launch / wait of the group), false = the
// caller issues it itself (solo thread, or too large for the bounce memory)
bool batch_copy(void* dst, const void* src, size_t n, int dir /*0 h2d, 1 d2h, 2 d2d, 3 memset(dst, (int)src)*/);
void batch_register_pinned(const void* p, size_t bytes);
void batch_unregister_pinned(const void* p);
extern thread_local BatchGroup* tls_batch_group;
extern thread_local int tls_batch_member;
BatchGroup& batch_current_group();                       // the thread's group (creates the solo group on first use)
void batch_group_init(BatchGroup& g, int slots, bool solo = false);         // table / bounce memory for `slots` members
void batch_group_release(BatchGroup& g);
// a member's slot (header + pack) of the launch it is about to take part in
unsigned char* batch_launch_begin(BatchGroup& g, const void* fn, void (*do_launch)(BatchGroup&), dim3 grid, dim3 block,
                                  size_t smem, size_t slot_bytes, hipStream_t stream);
void batch_launch_end(BatchGroup& g);                    // arrive; the last arriver launches for everybody
void batch_sync(hipStream_t s);                          // rendezvous + one stream wait for the whole group
void batch_leave();                                      // the calling member stops taking part (normal end or exception)
void batch_check_hip(hipError_t e, const char* what);

template <auto Fn, class... P>
struct BatchLauncher {
  static void launch(BatchGroup& g) {
    if (g.smem > 64 * 1024)
      batch_check_hip(hipFuncSetAttribute((const void*)lmn_batch_tramp<Fn, P...>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024), "hipFuncSetAttribute");
    // LMN_BATCH_ARGS=host: the trampoline reads the members' arguments straight from the page-locked table (one PCIe read
    // per wave) instead of from a device copy made by an extra hipMemcpyAsync per launch
    static const bool from_host = getenv("LMN_BATCH_ARGS") && !strcmp(getenv("LMN_BATCH_ARGS"), "host");
    const unsigned char* table = (from_host ? g.host : g.dev) + g.region;
    if (!from_host)
      batch_check_hip(hipMemcpyAsync(g.dev + g.region, g.host + g.region, g.slot_bytes * (size_t)g.slots,
                                     hipMemcpyHostToDevice, g.stream), "argument table upload");
    hipLaunchKernelGGL((lmn_batch_tramp<Fn, P...>), dim3(g.grid.x, g.grid.y, (unsigned)g.slots), g.block, g.smem, g.stream,
                       table, (uint32_t)g.slot_bytes);
    batch_check_hip(hipGetLastError(), "batched launch");
  }
};

template <auto Fn, class... P, class... A>
inline void batch_launch_typed(void (*)(P...), dim3 grid, dim3 block, size_t smem, hipStream_t stream, A&&... args) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
  BatchGroup& g = batch_current_group();
  constexpr size_t raw = sizeof(BatchSlotHdr) + (sizeof...(P) ? sizeof(BatchPack<P...>) : 0);
  constexpr size_t slot_bytes = (raw + BATCH_SLOT_ALIGN - 1) / BATCH_SLOT_ALIGN * BATCH_SLOT_ALIGN;
  static_assert(alignof(BatchPack<P...>) <= BATCH_SLOT_ALIGN, "pack alignment");
  unsigned char* slot = batch_launch_begin(g, (const void*)lmn_batch_tramp<Fn, P...>, &BatchLauncher<Fn, P...>::launch, grid,
                                           block, smem, slot_bytes, stream);
  BatchSlotHdr hdr{1u, grid.x, grid.y, 0u};
  memcpy(slot, &hdr, sizeof hdr);
  if constexpr (sizeof...(P) > 0) {
    BatchPack<P...> pk;
    batch_pack_fill(pk, std::forward<A>(args)...);
    memcpy(slot + sizeof(BatchSlotHdr), &pk, sizeof pk);
  }
  batch_launch_end(g);
}
template <auto Fn, class... A>
inline void batch_launch(dim3 grid, dim3 block, size_t smem, hipStream_t s, A&&... args) {
  batch_launch_typed<Fn>(static_cast<decltype(Fn)>(nullptr), grid, block, smem, s, std::forward<A>(args)...);
}

}  // namespace lmn
#endif  // LMN_BATCH
