// prove(): MI355X replacement for /root/reference/crates/prover/src/prover.rs:28-319.
// The Fiat-Shamir channel, proof assembly and (tiny) decommitment bookkeeping run on the host;
// every per-row / per-coefficient pass runs in the gfx950 kernels of kernels.hip on trace data
// that stays resident in HBM from the first transpose to the last query gather.
#include "prover.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>

namespace lmn {

// ------------------------------------------------------------------------------------ components
// Column layouts / relation wiring: crates/air/src/components/{add,mul,recip,inputs}/{table,component}.rs
static const ComponentSpec kSpecs[] = {
    // kind, n_cols, is_last, n_rel, rel_mult, rel_val, rel_id, n_local, rel_elems, rel_neg, rel_pre, n_pre, pre_id, n_pad, pad_col, pad_val
    {LMN_KIND_ADD, 15, 4, 3, {12, 13, 14}, {9, 10, 11}, {1, 2, 0}, 6, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_MUL, 16, 4, 3, {13, 14, 15}, {9, 10, 11}, {1, 2, 0}, 7, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_RECIP, 13, 3, 2, {11, 12}, {7, 8}, {1, 0}, 5, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_INPUTS, 7, 2, 1, {6}, {5}, {0}, 3, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    // constraint forms fully visible in the reference (no numerair helper):
    {LMN_KIND_SUM_REDUCE, 14, 3, 2, {12, 13}, {7, 8}, {1, 0}, 7, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},   // sum_reduce/component.rs:36-110
    {LMN_KIND_MAX_REDUCE, 15, 3, 2, {13, 14}, {7, 8}, {1, 0}, 9, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},   // max_reduce/component.rs
    {LMN_KIND_CONTIGUOUS, 11, 3, 2, {9, 10}, {7, 8}, {1, 0}, 4, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},    // contiguous/component.rs
    // numerair's eval_fixed_sqrt / eval_fixed_rem are un-vendored: natural fixed-point identities (unpinned)
    {LMN_KIND_SQRT, 13, 3, 2, {11, 12}, {7, 8}, {1, 0}, 5, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_REM, 16, 4, 3, {13, 14, 15}, {9, 10, 11}, {1, 2, 0}, 6, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    // less_than/component.rs:48-185; padding row less_than/table.rs:47-72 (rhs=1, out=4096, diff=1, limb0=1)
    {LMN_KIND_LESS_THAN, 22, 4, 7, {18, 19, 20, 21, 21, 21, 21}, {9, 10, 11, 14, 15, 16, 17}, {1, 2, 0, -1, -1, -1, -1}, 9,
     {0, 0, 0, 1, 1, 1, 1}, {0}, {0}, 0, {0, 0}, 4, {10, 11, 12, 14}, {1u, 4096u, 1u, 1u}},
    // lookups/range_check/component.rs: (-multiplicity, [range_check_8_column_0])
    {LMN_KIND_RANGE_CHECK_LOOKUP, 1, -1, 1, {0}, {0}, {-1}, 0, {ELEMS_RANGE_CHECK}, {1}, {1}, 1, {PRE_RANGE_CHECK, 0}, 0, {0}, {0}},
    // sin/component.rs:50-122 (exp2, log2 alike): node relations on input/out + LUT relation (lookup_mult, [input, out])
    {LMN_KIND_SIN, 12, 3, 3, {9, 10, 11}, {7, 8, 7}, {1, 0, 8}, 4, {0, 0, ELEMS_SIN}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_EXP2, 12, 3, 3, {9, 10, 11}, {7, 8, 7}, {1, 0, 8}, 4, {0, 0, ELEMS_EXP2}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_LOG2, 12, 3, 3, {9, 10, 11}, {7, 8, 7}, {1, 0, 8}, 4, {0, 0, ELEMS_LOG2}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    // lookups/sin/component.rs:40-59: (-multiplicity, [lut_0, lut_1]) over the two preprocessed columns
    {LMN_KIND_SIN_LOOKUP, 1, -1, 1, {0}, {0}, {1}, 0, {ELEMS_SIN}, {1}, {1}, 2, {PRE_SIN0, PRE_SIN0 + 1}, 0, {0}, {0}},
    {LMN_KIND_EXP2_LOOKUP, 1, -1, 1, {0}, {0}, {1}, 0, {ELEMS_EXP2}, {1}, {1}, 2, {PRE_EXP20, PRE_EXP20 + 1}, 0, {0}, {0}},
    {LMN_KIND_LOG2_LOOKUP, 1, -1, 1, {0}, {0}, {1}, 0, {ELEMS_LOG2}, {1}, {1}, 2, {PRE_LOG20, PRE_LOG20 + 1}, 0, {0}, {0}},
};
const ComponentSpec* component_spec(int kind) {
  for (auto& s : kSpecs)
    if (s.kind == kind) return &s;
  return nullptr;
}

ConstraintLayout constraint_layout(const ComponentSpec& sp, uint32_t flags) {
  ConstraintLayout L;
  L.n_kernel = sp.n_local + sp.n_rel;
  // kernel slot 1 is the eval_fixed_* constraint of Mul / Recip / Sqrt / Rem; Mul's kernel slot 2 is its zero slot
  bool drop_slot2 = false, extra_after1 = false, neg1 = false;
  switch (sp.kind) {
    case LMN_KIND_MUL: drop_slot2 = (flags & LMN_PV_MUL_ONE_SLOT) != 0; break;
    case LMN_KIND_RECIP: extra_after1 = (flags & LMN_PV_RECIP_TWO_SLOTS) != 0; neg1 = (flags & LMN_PV_RECIP_NEG) != 0; break;
    case LMN_KIND_SQRT: extra_after1 = (flags & LMN_PV_SQRT_TWO_SLOTS) != 0; neg1 = (flags & LMN_PV_SQRT_NEG) != 0; break;
    case LMN_KIND_REM: extra_after1 = (flags & LMN_PV_REM_TWO_SLOTS) != 0; neg1 = (flags & LMN_PV_REM_NEG) != 0; break;
    default: break;
  }
  int p = 0;
  for (int k = 0; k < L.n_kernel; ++k) {
    L.neg[k] = k == 1 && neg1;
    if (k == 2 && drop_slot2) {
      L.proto_index[k] = -1;
      continue;
    }
    L.proto_index[k] = p++;
    if (k == 1 && extra_after1) ++p;   // the helper's second (zero) slot
  }
  L.n_protocol = p;
  return L;
}

RelElems draw_relation_elements(Channel& channel, uint32_t protocol_flags) {
  RelElems e;
  auto draw = [&](int set) {
    std::vector<QM31> d = channel.draw_felts(2);
    if (set >= 0) {
      e.z[set] = d[0];
      e.alpha[set] = d[1];
      e.drawn[set] = true;
    }
  };
  draw(ELEMS_NODE);
  draw(ELEMS_SIN);  // the KAT era drew a single LUT relation; HEAD: sin, exp2, log2, range_check
  if (protocol_flags & LMN_PV_LUT_DRAWS4) {
    draw(ELEMS_EXP2);
    draw(ELEMS_LOG2);
    draw(ELEMS_RANGE_CHECK);
  }
  return e;
}

std::vector<int> assign_preprocessed(std::vector<Instance>& inst) {
  int log_of[N_PRE_IDS];
  for (int& l : log_of) l = -1;
  for (auto& ci : inst)
    for (int k = 0; k < ci.spec->n_pre; ++k) log_of[ci.spec->pre_id[k]] = ci.log_size;
  std::vector<int> order;
  for (int id = 0; id < N_PRE_IDS; ++id)
    if (log_of[id] >= 0) order.push_back(id);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return log_of[a] > log_of[b]; });
  int pos[N_PRE_IDS];
  std::vector<int> logs;
  for (size_t i = 0; i < order.size(); ++i) {
    pos[order[i]] = (int)i;
    logs.push_back(log_of[order[i]]);
  }
  for (auto& ci : inst)
    for (int k = 0; k < ci.spec->n_pre; ++k) ci.pre_idx[k] = pos[ci.spec->pre_id[k]];
  return logs;
}

// ------------------------------------------------------------------------------------ arena
Arena::~Arena() {
  if (base_) lmn_dev_free(base_);
}
void Arena::reserve(size_t bytes) {
  if (bytes <= cap_) return;
  if (base_) lmn_dev_free(base_);
  base_ = nullptr;
  cap_ = 0;
  try {
    base_ = (char*)lmn_dev_malloc(bytes);
  } catch (const LmnError& e) {
    throw LmnError(LMN_ERR_OUT_OF_MEMORY, std::string("device arena allocation failed: ") + e.what());
  }
  cap_ = bytes;
  off_ = 0;
}
void* Arena::alloc_bytes(size_t bytes) {
  size_t a = (off_ + 255) & ~(size_t)255;
  if (a + bytes > cap_) throw LmnError(LMN_ERR_OUT_OF_MEMORY, "device arena exhausted");
  off_ = a + bytes;
  return base_ + a;
}

// ------------------------------------------------------------------------------------ timing helper
struct TimedSpan {
  lmn_event_t a, b;
  int cat;  // index into accumulators
};
struct EventLog {
  std::vector<lmn_event_t> pool;
  size_t used = 0;
  std::vector<TimedSpan> spans;
  lmn_event_t get() {
    if (used == pool.size()) pool.push_back(lmn_event_create());
    return pool[used++];
  }
  void reset() {
    used = 0;
    spans.clear();
  }
  ~EventLog() {
    for (auto e : pool) lmn_event_destroy(e);
  }
};
static EventLog* g_log(Context* c);

enum Cat {
  C_TOTAL = 0, C_TRANSPOSE, C_MAIN_COMMIT, C_LOGUP, C_INTER_COMMIT, C_COMPOSITION, C_COMP_COMMIT, C_OODS, C_QUOT,
  C_FRI, C_DECOMMIT, C_FFT, C_MERKLE, C_MERKLE_FUSED, C_N
};

struct StageTimer {
  Context* ctx;
  EventLog* log;
  lmn_stream_t s;
  int cat;
  lmn_event_t a{};
  bool on;
  // Event records are not free (each costs a few microseconds between dependent kernels), so they are
  // only taken when profiling was requested for this context (lmn_set_profiling).
  StageTimer(Context* c, EventLog* l, lmn_stream_t st, int cat_) : ctx(c), log(l), s(st), cat(cat_), on(c->profiling) {
    if (!on) return;
    a = log->get();
    lmn_event_record(a, s);
  }
  ~StageTimer() {
    if (!on) return;
    lmn_event_t b = log->get();
    lmn_event_record(b, s);
    log->spans.push_back({a, b, cat});
  }
};

// one EventLog per context (owned through an opaque pointer to keep prover.h free of event types)
static EventLog* g_log(Context* c) { return static_cast<EventLog*>(c->event_log); }

// ------------------------------------------------------------------------------------ context
static void rccl_release(void* transport);  // single-proof sharding section below
static void release_host_scratch(void* p);  // decommit-planning storage, defined with HostScratch
void rccl_unique_id(uint8_t* out);

Context::Context(int device, const lmn_config& c) : cfg(c), device_(device) {
  // validate before acquiring anything: a throwing constructor does not run the destructor
  if (cfg.log_blowup != 1) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "only log_blowup = 1 is supported");
  if (cfg.n_queries == 0 || cfg.n_queries > 1024) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad n_queries");
  if (cfg.log_last_layer > 10) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad log_last_layer");
  if (cfg.pow_bits > 40) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad pow_bits");
  if (cfg.protocol_variant & ~LMN_PV_ALL) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad protocol_variant (unknown LMN_PV_* bits)");
  if (cfg.fp_scale != 12) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "only fp_scale = 12 is supported");
#ifndef LMN_EMU
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw LmnError(LMN_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "device index out of range");
  LMN_HIP_CHECK(hipSetDevice(device));
  {
    // Contexts get stream priorities round-robin over the device's range.  Concurrent provers that start together
    // (a service's worker pool, bench.py after its drain) otherwise tend to stay in lock-step: their latency-bound
    // FRI tails coincide and the chip idles ~0.7 ms per round.  With staggered priorities the contexts fall into a
    // pipeline instead - measured on 20-proof regions with 4 contexts: {488, 471, 432, 429, 386, 485} proofs/s
    // without, {466, 485, 486, 461, 486, 489} with; long runs and solo latency unchanged.
    // LMN_STREAM_PRIO_CYCLE=0 switches it off.
    static const bool cycle = !(getenv("LMN_STREAM_PRIO_CYCLE") && atoi(getenv("LMN_STREAM_PRIO_CYCLE")) == 0);
    static std::atomic<int> counter{0};
    int lo = 0, hi = 0;
    if (cycle && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
      const int span = lo - hi + 1;                     // lo = least priority (numerically greatest)
      const int prio = hi + (counter.fetch_add(1) % span);
      LMN_HIP_CHECK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, prio));
    } else {
      LMN_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    }
  }
  if (getenv("LMN_FRI_OVERLAP") && atoi(getenv("LMN_FRI_OVERLAP")) != 0) {
    LMN_HIP_CHECK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
    ev_fork_ = lmn_event_create_sync();
    ev_join_ = lmn_event_create_sync();
    have_stream2_ = true;
  }
#else
  stream_ = 0;
#endif
  event_log = new EventLog();
  pin_cap_ = 32u << 20;
  pin_base_ = (char*)lmn_host_alloc_pinned(pin_cap_);
  {
    const uint32_t zero[2] = {0u, 0u};
    bad_flag_ = (uint32_t*)lmn_dev_malloc(8);   // [0]: prove's non-canonical-word verdict, [1]: trace_lut's range verdict
    lmn_h2d(bad_flag_, zero, 8, stream_);
    lmn_sync(stream_);
  }
}

Context::~Context() {
#ifndef LMN_EMU
  (void)hipSetDevice(device_);
  (void)hipStreamSynchronize(stream_);
#endif
  delete static_cast<EventLog*>(event_log);
  event_log = nullptr;
  release_host_scratch(host_scratch);
  host_scratch = nullptr;
  if (shard_.rccl) rccl_release(shard_.rccl);
  shard_.rccl = nullptr;
  for (void* p : tw_allocs_) lmn_dev_free(p);
  if (bad_flag_) lmn_dev_free(bad_flag_);
  if (pin_base_) lmn_host_free_pinned(pin_base_);
#ifndef LMN_EMU
  if (have_stream2_) {
    (void)hipStreamSynchronize(stream2_);
    (void)hipStreamDestroy(stream2_);
    lmn_event_destroy(ev_fork_);
    lmn_event_destroy(ev_join_);
  }
  if (owns_stream_) (void)hipStreamDestroy(stream_);
#endif
}

#ifdef LMN_BATCH
void Context::adopt_stream(lmn_stream_t s) {
  LMN_HIP_CHECK(hipSetDevice(device_));
  LMN_HIP_CHECK(hipStreamSynchronize(stream_));
  if (owns_stream_) LMN_HIP_CHECK(hipStreamDestroy(stream_));
  stream_ = s;
  owns_stream_ = false;
}
#endif

void* Context::pin_alloc(size_t bytes) {
  size_t a = (pin_off_ + 63) & ~(size_t)63;
  if (a + bytes > pin_cap_) throw LmnError(LMN_ERR_OUT_OF_MEMORY, "pinned staging buffer exhausted");
  pin_off_ = a + bytes;
  return pin_base_ + a;
}
void* Context::stage_upload(const void* host, size_t bytes) {
  void* d = arena_.alloc_bytes(bytes ? bytes : 4);
  if (bytes == 0) return d;
  void* p = pin_alloc(bytes);
  memcpy(p, host, bytes);
  lmn_h2d(d, p, bytes, stream_);
  return d;
}
const void* Context::stage_download(const void* dev, size_t bytes) {
  void* p = pin_alloc(bytes ? bytes : 4);
  if (bytes) lmn_d2h(p, dev, bytes, stream_);
  return p;
}
void Context::fetch_root_async(DevMerkle& m) {
  if (m.layers.empty() || !m.layers[0]) return;
  m.root_pinned = (const uint32_t*)stage_download(m.layers[0], 32);
}

void* Context::upload(const void* host, size_t bytes) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  void* d = lmn_dev_malloc(bytes);
  lmn_h2d(d, host, bytes, stream_);
  lmn_sync(stream_);
  return d;
}
void Context::device_free(void* p) { lmn_dev_free(p); }
void* Context::device_alloc(size_t bytes) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  return lmn_dev_malloc(bytes ? bytes : 4);
}
void Context::upload_to(const void* host, size_t bytes, void* dst) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  lmn_h2d(dst, host, bytes, stream_);
  lmn_sync(stream_);  // the host buffer is borrowed only for the duration of the call
}
void Context::device_copy(void* dst, const void* src, size_t bytes) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  lmn_d2d(dst, src, bytes, stream_);  // stream-ordered, no wait
}
void Context::download(const void* device, void* host, size_t bytes) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  lmn_d2h(host, device, bytes, stream_);
  lmn_sync(stream_);
}

static TraceNode trace_node(const lmn_node_info& info) {
  auto m31 = [](int64_t v) { return (uint32_t)(((v % (int64_t)P31) + (int64_t)P31) % (int64_t)P31); };
  TraceNode nd{};
  nd.node_id = info.node_id;
  nd.lhs_id = info.input_ids[0];
  nd.rhs_id = info.input_ids[1];
  nd.lhs_mult = m31(info.input_mults[0]);
  nd.rhs_mult = m31(info.input_mults[1]);
  nd.out_mult = info.is_final_output ? 0u : m31(info.num_consumers);
  return nd;
}

// `LuminairSumReduce::process_trace` (prim.rs:1450-1565) on a contiguous (front, dim, back) device tensor
void Context::trace_reduce(bool is_max, const int32_t* input, uint64_t front, uint64_t dim, uint64_t back,
                           const lmn_node_info& info, uint32_t* rows, uint64_t row_offset, int32_t* out) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (front == 0 || dim == 0 || back == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (front * back * dim >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tensor too large");
  launch_trace_reduce(is_max, input, front, dim, back, trace_node(info), rows + row_offset * (is_max ? 15ull : 14ull), out,
                      stream_);  // stream-ordered with every later call on this context (lmn_prove, lmn_download)
}

// `process_trace` of one Add / Mul / Recip node on device tensors (prim.rs:967-1013, :1090-1139, :388-431)
static TraceView trace_view(const lmn_view* v, uint64_t n) {
  TraceView t{};
  if (!v) return t;
  if (v->ndim < 1 || v->ndim > 4) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: ndim must be 1..4");
  uint64_t prod = 1;
  t.ndim = v->ndim;
  for (uint32_t k = 0; k < v->ndim; ++k) {
    if (v->shape[k] == 0) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: empty dimension");
    t.shape[k] = v->shape[k];
    t.strides[k] = v->strides[k];
    prod *= v->shape[k];
  }
  if (v->offset < 0) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: negative offset");
  t.offset = v->offset;
  if (prod != n) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: shape does not match the element count");
  return t;
}

// `process_trace` of a Sin / Exp2 / Log2 node on a device tensor; fills the LUT multiplicity column too
void Context::trace_lut(uint32_t kind, const int32_t* input, const lmn_view* view, uint64_t n, const lmn_node_info& info,
                        const uint32_t* lut_col1, const lmn_range* ranges, uint32_t n_ranges, uint32_t* mult, uint32_t* rows,
                        uint64_t row_offset, int32_t* out) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (kind != LMN_KIND_SIN && kind != LMN_KIND_EXP2 && kind != LMN_KIND_LOG2)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: kind must be Sin, Exp2 or Log2");
  if (n == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (n >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad sizes");
  if (!ranges || n_ranges == 0 || n_ranges > (uint32_t)LUT_MAX_RANGES)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: 1..16 value ranges");
  LutRanges rg{};
  rg.n = (int)n_ranges;
  uint64_t base = 0;
  for (uint32_t k = 0; k < n_ranges; ++k) {
    // ascending, disjoint, inside the Fixed<12> range the LUT generator accepts (coalesce_ranges' output)
    if (ranges[k].hi < ranges[k].lo || ranges[k].lo <= -(1ll << 30) || ranges[k].hi >= (1ll << 30) ||
        (k > 0 && ranges[k].lo <= ranges[k - 1].hi))
      throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: ranges must be ascending, disjoint and inside (-2^30, 2^30)");
    rg.lo[k] = (int32_t)ranges[k].lo;
    rg.hi[k] = (int32_t)ranges[k].hi;
    rg.base[k] = (uint32_t)base;
    base += (uint64_t)(ranges[k].hi - ranges[k].lo + 1);
    if (base > (1ull << 26)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: LUT larger than 2^26 rows");
  }
  const TraceView tv = trace_view(view, n);
  // bad_flag_[1] is zero between calls; the kernel sets it when an input misses every range
  launch_trace_lut(input, tv, n, trace_node(info), lut_col1, rg, mult, rows + row_offset * 12ull, out, bad_flag_ + 1, stream_);
  uint32_t err = 0;
  lmn_d2h(&err, bad_flag_ + 1, 4, stream_);
  lmn_sync(stream_);
  if (err) {
    const uint32_t zero = 0u;
    lmn_h2d(bad_flag_ + 1, &zero, 4, stream_);
    lmn_sync(stream_);
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: an input value lies outside the LUT's range");
  }
}

void Context::trace_elementwise(uint32_t kind, const int32_t* lhs, const lmn_view* lv, const int32_t* rhs,
                                const lmn_view* rv, uint64_t n, const lmn_node_info& info, uint32_t* rows,
                                uint64_t row_offset, int32_t* out, uint32_t* aux) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  const ComponentSpec* sp = component_spec((int)kind);
  const bool binary = kind == LMN_KIND_ADD || kind == LMN_KIND_MUL || kind == LMN_KIND_REM || kind == LMN_KIND_LESS_THAN;
  const bool unary = kind == LMN_KIND_RECIP || kind == LMN_KIND_SQRT || kind == LMN_KIND_CONTIGUOUS || kind == LMN_KIND_INPUTS;
  if (!sp || !(binary || unary))
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_elementwise: not an elementwise kind");
  if (binary && !rhs) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_elementwise: missing right operand");
  if (kind == LMN_KIND_LESS_THAN && !aux)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_elementwise: LessThan needs the range-check multiplicity table");
  if (n == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (n >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tensor too large");
  const TraceNode nd = trace_node(info);
  const TraceView tlv = trace_view(lv, n), trv = trace_view(rv, n);
  launch_trace_elementwise((int)kind, lhs, tlv, rhs, trv, n, nd, rows + row_offset * (uint64_t)sp->n_cols, out, aux,
                           stream_);  // stream-ordered with every later call on this context
}

// `LuminairContiguous::process_trace` in the reference's own row rule (prim.rs:229-301): max(in_size, out_size) rows
void Context::trace_contiguous(const int32_t* input, uint64_t in_size, const lmn_view* view, uint64_t out_size,
                               const lmn_node_info& info, uint32_t* rows, uint64_t row_offset, int32_t* out) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (in_size == 0 || out_size == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (in_size >= (1ull << 31) || out_size >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tensor too large");
  TraceNode nd = trace_node(info);
  nd.phys_n = in_size;
  nd.out_n = out_size;
  const TraceView tv = trace_view(view, out_size);
  if (view)  // every element the view addresses must lie inside the buffer
    for (uint64_t corner = 0; corner < (1ull << view->ndim); ++corner) {
      int64_t off = view->offset;
      for (uint32_t k = 0; k < view->ndim; ++k)
        if (corner >> k & 1) off += (int64_t)(view->shape[k] - 1) * view->strides[k];
      if (off < 0 || (uint64_t)off >= in_size) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "contiguous: the view leaves the input buffer");
    }
  else if (out_size > in_size)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "contiguous: output larger than the input buffer without a view");
  const uint64_t n = std::max(in_size, out_size);
  launch_trace_elementwise(LMN_KIND_CONTIGUOUS, input, tv, nullptr, TraceView{}, n, nd, rows + row_offset * 11ull, out, nullptr,
                           stream_);
}

// Twiddle tables for every canonic domain up to 2^max_domain_log (computed
// once per context and cached across proofs, instead of once per proof as prover.rs:38-42 does).
//   Y[m][h] = y(half_coset_m.at(bitrev(h, m-1))), h < 2^(m-1)      (layer 0 of domain m)
//   X[k][h] = x(half_coset_k.at(bitrev(h, k-2))), h < 2^(k-2)      (layer 1 of domain k)
// Layer i >= 1 of domain m is X[m-i+1] (doubling a canonic half coset gives the next smaller one).
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
// Twiddle tables are a function of the domain size alone: one set per device, built on the device (k_twiddles) and shared by
// every context of the process.  The registry holds weak references - the last context that goes away frees the tables -
// and a context that needs a larger domain than the current set builds a new one (the contexts still using the old one
// keep it alive).  The batch library's members run in lock-step (no member may skip launches another one makes) and the
// emulation build has no device: both keep one host-built set per context (below).
namespace {
struct TwiddleSet {
  int device = 0, max_log = 0;
  void* slab = nullptr;
  std::vector<uint32_t*> Y, X, iY, iX, Y2, X2, iY2, iX2;
  ~TwiddleSet() {
    if (slab) {
      (void)hipSetDevice(device);
      (void)hipFree(slab);
    }
  }
};
std::mutex g_tw_mu;
std::map<int, std::weak_ptr<TwiddleSet>> g_tw;
}  // namespace

void Context::ensure_twiddles(int M) {
  if (M <= tw_max_log_) return;
  if (M > MAX_LOG - 2) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace too large");
  std::lock_guard<std::mutex> lk(g_tw_mu);
  std::shared_ptr<TwiddleSet> set = g_tw[device_].lock();
  if (!set || set->max_log < M) {
    set = std::make_shared<TwiddleSet>();
    set->device = device_;
    set->max_log = M;
    // 8 tables per size (y and x coordinate; value / inverse / both doubled), each at a 256-byte boundary
    auto slot = [](uint64_t words) { return (words + 63) & ~(uint64_t)63; };
    uint64_t total = 0;
    for (int m = 1; m <= M; ++m) total += 4 * slot(1ull << (m - 1)) + (m >= 2 ? 4 * slot(1ull << (m - 2)) : 0);
    set->slab = lmn_dev_malloc(total * 4);
    uint32_t* at = (uint32_t*)set->slab;
    auto take = [&](uint64_t words) {
      uint32_t* p = at;
      at += slot(words);
      return p;
    };
    for (auto* v : {&set->Y, &set->X, &set->iY, &set->iX, &set->Y2, &set->X2, &set->iY2, &set->iX2}) v->assign(M + 1, nullptr);
    for (int m = M; m >= 1; --m) {
      // half coset of CanonicCoset(m): initial index 2^(30-m), step 2^(32-m), 2^(m-1) points
      TwGen g{};
      const Pt init = pt_of_index(1u << (30 - m));
      g.ix = init.x;
      g.iy = init.y;
      Pt st = pt_of_index(m >= 2 ? (1u << (32 - m)) : 0u);
      for (int k = 0; k < 30; ++k) {
        g.sx[k] = st.x;
        g.sy[k] = st.y;
        st = pt_double(st);
      }
      const uint64_t half = 1ull << (m - 1);
      set->Y[m] = take(half);
      set->iY[m] = take(half);
      set->Y2[m] = take(half);
      set->iY2[m] = take(half);
      launch_twiddles(m - 1, g, 0, set->Y[m], set->iY[m], set->Y2[m], set->iY2[m], stream_);
      if (m >= 2) {
        const uint64_t quarter = 1ull << (m - 2);
        set->X[m] = take(quarter);
        set->iX[m] = take(quarter);
        set->X2[m] = take(quarter);
        set->iX2[m] = take(quarter);
        launch_twiddles(m - 2, g, 1, set->X[m], set->iX[m], set->X2[m], set->iX2[m], stream_);
      }
    }
    lmn_sync(stream_);   // every stream of the process may read the tables from here on
    g_tw[device_] = set;
  }
  twY_ = set->Y;
  twX_ = set->X;
  itwY_ = set->iY;
  itwX_ = set->iX;
  twY2_ = set->Y2;
  twX2_ = set->X2;
  itwY2_ = set->iY2;
  itwX2_ = set->iX2;
  tw_max_log_ = set->max_log;
  tw_shared_ = set;
}
#else
void Context::ensure_twiddles(int M) {
  if (M <= tw_max_log_) return;
  if (M > MAX_LOG - 2) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace too large");
  for (void* p : tw_allocs_) lmn_dev_free(p);
  tw_allocs_.clear();
  twY_.assign(M + 1, nullptr);
  twX_.assign(M + 1, nullptr);
  itwY_.assign(M + 1, nullptr);
  itwX_.assign(M + 1, nullptr);
  twY2_.assign(M + 1, nullptr);
  twX2_.assign(M + 1, nullptr);
  itwY2_.assign(M + 1, nullptr);
  itwX2_.assign(M + 1, nullptr);
  auto batch_inverse = [](const std::vector<uint32_t>& v) {
    std::vector<uint32_t> pre(v.size()), out(v.size());
    uint32_t acc = 1;
    for (size_t i = 0; i < v.size(); ++i) {
      pre[i] = acc;
      acc = m_mul(acc, v[i]);
    }
    uint32_t inv = m_inv(acc);
    for (size_t i = v.size(); i-- > 0;) {
      out[i] = m_mul(inv, pre[i]);
      inv = m_mul(inv, v[i]);
    }
    return out;
  };
  auto up = [&](const std::vector<uint32_t>& v) {
    uint32_t* d = (uint32_t*)lmn_dev_malloc(v.size() * 4);
    tw_allocs_.push_back(d);
    lmn_h2d(d, v.data(), v.size() * 4, stream_);
    lmn_sync(stream_);
    return d;
  };
  auto up2 = [&](std::vector<uint32_t> v) {   // doubled entries (TwPtrs::d)
    for (auto& x : v) x *= 2u;
    return up(v);
  };
  for (int m = 1; m <= M; ++m) {
    // half coset of CanonicCoset(m): initial index 2^(30-m), step 2^(32-m), 2^(m-1) points
    uint32_t half = 1u << (m - 1);
    Pt cur = pt_of_index(1u << (30 - m));
    Pt step = pt_of_index(m >= 2 ? (1u << (32 - m)) : 0u);
    std::vector<Pt> pts(half);
    for (uint32_t j = 0; j < half; ++j) {
      pts[j] = cur;
      cur = pt_add(cur, step);
    }
    std::vector<uint32_t> Y(half);
    for (uint32_t h = 0; h < half; ++h) Y[h] = pts[bit_reverse(h, m - 1)].y;
    twY_[m] = up(Y);
    itwY_[m] = up(batch_inverse(Y));
    twY2_[m] = up2(Y);
    itwY2_[m] = up2(batch_inverse(Y));
    if (m >= 2) {
      uint32_t quarter = 1u << (m - 2);
      std::vector<uint32_t> X(quarter);
      for (uint32_t h = 0; h < quarter; ++h) X[h] = pts[bit_reverse(h, m - 2)].x;
      twX_[m] = up(X);
      itwX_[m] = up(batch_inverse(X));
      twX2_[m] = up2(X);
      itwX2_[m] = up2(batch_inverse(X));
    }
  }
  tw_max_log_ = M;
}
#endif

TwPtrs Context::tw(int m) const {
  TwPtrs t{};
  t.l[0] = twY_[m];
  t.d[0] = twY2_[m];
  for (int i = 1; i < m; ++i) {
    t.l[i] = twX_[m - i + 1];
    t.d[i] = twX2_[m - i + 1];
  }
  return t;
}
TwPtrs Context::itw(int m) const {
  TwPtrs t{};
  t.l[0] = itwY_[m];
  t.d[0] = itwY2_[m];
  for (int i = 1; i < m; ++i) {
    t.l[i] = itwX_[m - i + 1];
    t.d[i] = itwX2_[m - i + 1];
  }
  return t;
}

// ------------------------------------------------------------------------------------ timed launches
void Context::merkle_layer_timed(const uint32_t* prev, const uint32_t* const* cols, int ncols, uint32_t size,
                                 uint32_t* out) {
  launch_merkle_layer(prev, cols, ncols, size, out, stream_);
  timings.merkle_launches++;
  timings.merkle_bytes += (uint64_t)size * (4ull * ncols + 32ull + (prev ? 64ull : 0ull));
  timings.merkle_compressions += (uint64_t)size * std::max<uint64_t>(1, ((prev ? 16 : 0) + (uint64_t)ncols + 15) / 16);
}

// Merkle tree over columns sorted by size (descending, stable): SURVEY.md Appendix A.4.
// Levels are produced by fused subtree launches: a start level (children hashes and/or its own
// columns) plus up to 8 following levels that have no columns of their own.
void Context::build_merkle_levels(std::vector<uint32_t*>& layers, int max_log,
                                  const std::vector<std::vector<const uint32_t*>>& per_level, DevChannel* ch,
                                  QM31* alpha_out, uint32_t* root_copy, const MerkleFold* fold, std::vector<MerkleCut>* cuts) {
  bool chan_done = false;
  auto layer = [&](int l) {   // storage of level l, allocated when the first launch writes it
    if (!layers[l]) layers[l] = arena_.alloc_words((size_t)8 << l);
    return layers[l];
  };
  if (fold && (max_log <= 10 || per_level[max_log].size() != 4))
    throw LmnError(LMN_ERR_INTERNAL, "merkle: a folded leaf level needs a 4-column tree of more than 2^10 leaves");
  {
    StageTimer t(this, g_log(this), stream_, C_MERKLE);
    const uint32_t* prev = nullptr;
    int level = max_log;
    MerkleFold below{};   // the leaf level of a tree whose next level has columns too: hashed by that level's launch
    // from 2^19 leaves on (below that the launches are latency-bound and the separate leaf launch is the cheaper form);
    // LMN_MERKLE_BELOW_MIN_LOG lowers the threshold for the emulation tests
    const char* below_env = getenv("LMN_MERKLE_BELOW_MIN_LOG");
    const int below_min_log = below_env ? std::max(12, atoi(below_env)) : 19;
    // runs of contiguous equal-size columns of a level; false if there are more than MERKLE_MAX_SEG of them
    auto make_segs = [&](int lv, MerkleSegs& sg) {
      int nseg = 0;
      for (auto* c : per_level[lv]) {
        if (nseg > 0 && c == sg.base[nseg - 1] + ((uint64_t)sg.n[nseg - 1] << lv)) {
          sg.n[nseg - 1]++;
        } else if (nseg < MERKLE_MAX_SEG) {
          sg.base[nseg] = c;
          sg.n[nseg] = 1;
          ++nseg;
        } else {
          return false;
        }
      }
      return true;
    };
    while (level >= 0) {
      auto& lc = per_level[level];
      if (level == wait_before_level_) {   // this level's columns were produced on the second stream
        lmn_stream_wait_event(stream_, wait_before_level_ev_);
        wait_before_level_ = -1;
      }
      if (cuts && !fold && !prev && level == max_log && level >= below_min_log && !lc.empty() && lc.size() <= 8 &&
          !per_level[level - 1].empty()) {
        MerkleSegs sl{}, snext{};
        if (make_segs(level, sl) && sl.n[0] == (int)lc.size() && make_segs(level - 1, snext)) {
          below.below = lc[0];
          below.below_ncols = (int)lc.size();
          cuts->push_back({level, 1, nullptr, sl, (int)lc.size()});   // a node of this level = the hash of its leaf
          timings.merkle_fused_bytes += ((uint64_t)1 << level) * (4ull * lc.size() + 32ull);
          timings.merkle_fused_compressions += (uint64_t)1 << level;
          timings.merkle_bytes += ((uint64_t)1 << level) * (4ull * lc.size() + 32ull);
          timings.merkle_compressions += (uint64_t)1 << level;
          level -= 1;
          continue;
        }
      }
      MerkleSegs sg{};
      const bool seg_ok = make_segs(level, sg);
      if (!seg_ok) {
        // rare scattered level: pointer-table kernel, one level per launch
        const uint32_t** dptrs = (const uint32_t**)stage_upload(lc.data(), lc.size() * sizeof(void*));
        merkle_layer_timed(prev, dptrs, (int)lc.size(), 1u << level, layer(level));
        prev = layers[level];
        level -= 1;
        continue;
      }
      int plain = 0;
      while (level - plain - 1 >= 0 && per_level[level - plain - 1].empty()) ++plain;
      MerkleLevels outs{};
      int nfused;
      bool over_leaves = false;
      if (level <= 10) {
        nfused = std::min(plain, 10);
        for (int l = 0; l <= nfused; ++l) outs.p[l] = layer(level - l);
        bool to_root = level - nfused == 0;
        launch_merkle_small(prev, sg, (int)lc.size(), 1u << level, outs, nfused, to_root ? ch : nullptr, alpha_out,
                            root_copy, stream_);
        if (to_root && ch) chan_done = true;
      } else {
        nfused = std::min(std::min(plain, MERKLE_MAX_FUSED), level - 10);
        // per-lane subtree depth: only as deep as still leaves >= 2^17 lanes (latency-bound below that)
        int sub = std::max(0, std::min(std::min(MERKLE_MAX_SUB, nfused), level - 17));
        if (const char* e = getenv("LMN_MERKLE_SUB")) sub = std::min(std::min(atoi(e), nfused), MERKLE_MAX_SUB);
        nfused = std::min(nfused, sub + 8);
        // the `sub` levels a lane reduces in registers are not written when the caller can recompute what it needs of them
        const int skip = cuts ? sub : 0;
        for (int l = 0; l < skip; ++l) {
          if (layers[level - l]) throw LmnError(LMN_ERR_INTERNAL, "merkle: a level to be skipped already has storage");
          outs.p[l] = nullptr;
        }
        for (int l = skip; l <= nfused; ++l) outs.p[l] = layer(level - l);
        over_leaves = below.below != nullptr;
        if (skip) cuts->push_back({level, skip, prev, sg, (int)lc.size(), below.below, below.below_ncols});
        StageTimer tf(this, g_log(this), stream_, C_MERKLE_FUSED);
        launch_merkle_fused(prev, sg, (int)lc.size(), 1u << level, outs, sub, nfused, stream_,
                            over_leaves ? &below : (level == max_log ? fold : nullptr));
        below = MerkleFold{};
        timings.merkle_fused_launches++;
        // a folded leaf level also reads the pair it folds (32 B) and writes the layer (16 B) instead of reading it (16 B)
        if (fold && level == max_log) timings.merkle_fused_bytes += ((uint64_t)1 << level) * 32ull;
        const bool kids = prev || over_leaves;   // SURVEY's byte formula: as if the children's hashes were read
        uint64_t words = (kids ? 16 : 0) + lc.size();
        timings.merkle_fused_bytes += ((uint64_t)1 << level) * (4ull * lc.size() + 32ull + (kids ? 64ull : 0ull));
        timings.merkle_fused_compressions += ((uint64_t)1 << level) * std::max<uint64_t>(1, (words + 15) / 16);
        for (int l = 1; l <= nfused; ++l) {
          timings.merkle_fused_bytes += ((uint64_t)1 << (level - l)) * 96ull;
          timings.merkle_fused_compressions += (uint64_t)1 << (level - l);
        }
      }
      timings.merkle_launches++;
      const bool had_kids = prev || over_leaves;
      timings.merkle_bytes += ((uint64_t)1 << level) * (4ull * lc.size() + 32ull + (had_kids ? 64ull : 0ull));
      for (int l = 1; l <= nfused; ++l) timings.merkle_bytes += ((uint64_t)1 << (level - l)) * 96ull;
      {
        uint64_t words = (had_kids ? 16 : 0) + lc.size();
        timings.merkle_compressions += ((uint64_t)1 << level) * std::max<uint64_t>(1, (words + 15) / 16);
        for (int l = 1; l <= nfused; ++l) timings.merkle_compressions += (uint64_t)1 << (level - l);
      }
      prev = layers[level - nfused];
      level -= nfused + 1;
    }
  }
  if (ch && !chan_done) launch_chan_mix_root_draw(ch, layers[0], alpha_out, root_copy, stream_);
}

void Context::build_merkle(DevMerkle& m, const std::vector<ColRef>& cols_sorted, DevChannel* ch, QM31* alpha_out,
                           uint32_t* root_copy, bool sharded, const MerkleFold* fold) {
  if (fold && sharded) throw LmnError(LMN_ERR_INTERNAL, "merkle: folded leaf levels are not sharded");
  m.max_log = cols_sorted.empty() ? 0 : cols_sorted[0].log;
  m.layers.assign(m.max_log + 1, nullptr);
  m.cuts.clear();
  m.g = 0;
  if (cols_sorted.empty()) {
    m.root = b2_hash_words(nullptr, 0);
    return;
  }
  if (!sharded) {
    std::vector<std::vector<const uint32_t*>> per_level(m.max_log + 1);
    for (auto& c : cols_sorted) {
      if (c.sharded) throw LmnError(LMN_ERR_INTERNAL, "merkle: sharded column in a replicated tree");
      per_level[c.log].push_back(c.ptr);
    }
    build_merkle_levels(m.layers, m.max_log, per_level, ch, alpha_out, root_copy, fold, merkle_cut_ ? &m.cuts : nullptr);
    return;
  }
  // Sharded tree (SURVEY.md §8e stage C/D): the aligned block of rows [rank * 2^(k-g), (rank+1) * 2^(k-g)) of every
  // column of log size k is the leaf data of subtree `rank` below level g.  Hash that subtree here, all-gather the
  // world subtree roots (32 B each) and hash the top g levels identically on every rank.
  const int g = shard_.g;
  m.g = g;
  const int loc_log = m.max_log - g;
  if (loc_log < 0) throw LmnError(LMN_ERR_INTERNAL, "merkle: tree smaller than the shard count");
  std::vector<std::vector<const uint32_t*>> per_level(loc_log + 1);
  for (auto& c : cols_sorted) {
    if (c.log < g) throw LmnError(LMN_ERR_INTERNAL, "merkle: column smaller than the shard count");
    per_level[c.log - g].push_back(c.sharded ? c.ptr : c.ptr + ((uint64_t)shard_.rank << (c.log - g)));
  }
  uint32_t* level_g = arena_.alloc_words((size_t)8 << g);  // node r = root of rank r's subtree
  std::vector<uint32_t*> loc(loc_log + 1, nullptr);
  loc[0] = level_g + 8ull * shard_.rank;
  for (int l = loc_log; l >= 1; --l) loc[l] = arena_.alloc_words((size_t)8 << l);
  build_merkle_levels(loc, loc_log, per_level, nullptr, nullptr, nullptr);
  for (int l = 1; l <= loc_log; ++l) m.layers[l + g] = loc[l];
  m.layers[g] = level_g;
  gather_columns(level_g, 0, 1, 8);
  if (g == 0) {
    if (ch) launch_chan_mix_root_draw(ch, level_g, alpha_out, root_copy, stream_);
    return;
  }
  for (int l = g - 1; l >= 0; --l) m.layers[l] = arena_.alloc_words((size_t)8 << l);
  MerkleLevels outs{};
  for (int l = 0; l <= g - 1; ++l) outs.p[l] = m.layers[g - 1 - l];
  MerkleSegs none{};
  StageTimer t(this, g_log(this), stream_, C_MERKLE);
  launch_merkle_small(level_g, none, 0, 1u << (g - 1), outs, g - 1, ch, alpha_out, root_copy, stream_);
  timings.merkle_launches++;
  timings.merkle_compressions += (1ull << g) - 1;
}

// In-place all-gather of column blocks through the shard's collective (RCCL over xGMI, or the caller's callback).
void Context::gather_columns(uint32_t* base, uint64_t col_stride, int ncols, uint64_t words_per_rank) {
  if (!shard_.active) throw LmnError(LMN_ERR_INTERNAL, "gather without a shard");
  const bool group = ncols > 1 && shard_.coll.group_begin && shard_.coll.group_end;
  if (group && shard_.coll.group_begin(shard_.coll.user) != 0) throw LmnError(LMN_ERR_INTERNAL, "shard group_begin failed");
  int rc = 0;
  timings.shard_gather_bytes += (uint64_t)ncols * words_per_rank * 4 * (shard_.world - 1);
  timings.shard_gather_calls += (uint32_t)ncols;
  for (int c = 0; c < ncols && rc == 0; ++c)
    rc = shard_.coll.all_gather(shard_.coll.user, base + (uint64_t)c * col_stride, (size_t)words_per_rank * 4,
                                (void*)(uintptr_t)stream_);
  // an open group is always closed, also when one of its calls failed
  if (group && shard_.coll.group_end(shard_.coll.user) != 0 && rc == 0) rc = -1;
  if (rc != 0) throw LmnError(LMN_ERR_INTERNAL, "shard all_gather failed (code " + std::to_string(rc) + ")");
}

bool Context::shard_all_to_all() const {
  static const bool off = getenv("LMN_SHARD_A2A") && atoi(getenv("LMN_SHARD_A2A")) == 0;   // ablation: replicated interpolation
  return shard_.active && shard_.world > 1 && shard_.coll.all_to_all != nullptr && !off;
}

// column-parallel interpolation pays where the transforms are throughput-bound; small columns stay replicated (two more
// collectives would cost more than the few microseconds of butterflies).  LMN_SHARD_A2A_MIN_LOG lowers the bar (tests).
bool Context::shard_a2a_columns(int log_size) const {
  static const int min_log = getenv("LMN_SHARD_A2A_MIN_LOG") ? atoi(getenv("LMN_SHARD_A2A_MIN_LOG")) : 13;
  return shard_all_to_all() && cfg.log_blowup == 1 && log_size >= min_log && log_size >= 4;
}
// Row-parallel front end: every rank transposes and computes the logup fractions of its row block only; the blocks go to
// the columns' owners by an all-to-all (the reverse of stage B).  Three more collectives per component: worth it for the
// big tables (BASELINE config 5: 2^23 rows), not at 2^20.  LMN_SHARD_ROWS_MIN_LOG lowers the bar (tests).
bool Context::shard_rows_front(int log_size) const {
  static const int min_log = getenv("LMN_SHARD_ROWS_MIN_LOG") ? atoi(getenv("LMN_SHARD_ROWS_MIN_LOG")) : 22;
  return shard_a2a_columns(log_size) && log_size >= min_log && ((1ull << log_size) >> shard_.g) >= 64;
}

Context::CommitOut Context::interpolate_for_commit(uint32_t* coeffs, const uint32_t* evals, int ncols, int log_size,
                                                   int halo_first, bool evals_row_blocks) {
  const uint64_t n = 1ull << log_size;
  CommitOut out;
  StageTimer t(this, g_log(this), stream_, C_FFT);
  if (!shard_.active && cfg.log_blowup == 1 && fft_interp_extend_supported(log_size)) {
    timings.fft_bytes += (uint64_t)ncols * 8ull * n;
    timings.fft_butterflies += (uint64_t)ncols * (n / 2) * (uint64_t)log_size;
    uint32_t* lde = arena_.alloc_words((size_t)ncols * 2 * n);
    timings.fft_launches += launch_interp_extend(coeffs, n, evals, n, lde, 2 * n, ncols, log_size, itw(log_size),
                                                 tw(log_size + 1), stream_);
    timings.fft_bytes += (uint64_t)ncols * 12ull * n;                       // the extension: 4n read + 8n written
    timings.fft_butterflies += (uint64_t)ncols * n * (uint64_t)log_size;  // 2^(k+1)/2 * k (the top layer is the identity)
    out.lde = lde;
    out.stride = 2 * n;
    return out;
  }
  const int g = shard_.g;
  const uint64_t L = 2 * n, Lb = L >> g;
  if (evals_row_blocks && !shard_a2a_columns(log_size)) throw LmnError(LMN_ERR_INTERNAL, "row-block evaluations without stage A");
  if (shard_a2a_columns(log_size)) {
    const uint32_t G = shard_.world, me = shard_.rank;
    for (uint32_t r = 0; r <= G; ++r) out.first[r] = (int)((uint64_t)r * ncols / G);
    const int c0 = out.first[me], nm = out.first[me + 1] - c0;
    size_t so[8], sb[8], ro[8], rb[8];
    if (evals_row_blocks) {
      // ---- the row blocks of this rank's columns come in from every rank (block p of column c from rank p)
      const uint64_t nb = n >> g;
      uint32_t* blocks = arena_.alloc_words((size_t)std::max(nm, 1) * n);
      for (uint32_t p = 0; p < G; ++p) {
        so[p] = (size_t)out.first[p] * nb * 4;
        sb[p] = (size_t)(out.first[p + 1] - out.first[p]) * nb * 4;
        ro[p] = (size_t)p * nm * nb * 4;
        rb[p] = (size_t)nm * nb * 4;
      }
      if (shard_.coll.all_to_all(shard_.coll.user, evals, so, sb, blocks, ro, rb, (void*)(uintptr_t)stream_) != 0)
        throw LmnError(LMN_ERR_INTERNAL, "shard all_to_all failed");
      timings.shard_a2a_bytes += (uint64_t)nm * (n - nb) * 4;
      timings.shard_a2a_calls++;
      launch_unpack_blocks(blocks, coeffs + (uint64_t)c0 * n, n, (uint32_t)nb, nm, (int)G, stream_);
      evals = coeffs;   // stage A continues in place
    }
    // ---- stage A: this rank's share of the columns, interpolated and extended over ALL rows
    uint32_t* full = arena_.alloc_words((size_t)std::max(nm, 1) * L);
    if (nm > 0) {
      timings.fft_bytes += (uint64_t)nm * 20ull * n;
      timings.fft_butterflies += (uint64_t)nm * (n / 2 + n) * (uint64_t)log_size;
      if (fft_interp_extend_supported(log_size)) {
        timings.fft_launches += launch_interp_extend(coeffs + (uint64_t)c0 * n, n, evals + (uint64_t)c0 * n, n, full, L, nm,
                                                     log_size, itw(log_size), tw(log_size + 1), stream_);
      } else {
        timings.fft_launches += launch_ifft(coeffs + (uint64_t)c0 * n, n, evals + (uint64_t)c0 * n, n, nm, log_size, itw(log_size), stream_);
        timings.fft_launches += launch_fft(full, L, coeffs + (uint64_t)c0 * n, n, log_size, nm, log_size + 1, tw(log_size + 1), stream_);
      }
    }
    // ---- stage B: row block s of every own column goes to rank s
    uint32_t* sendbuf = arena_.alloc_words((size_t)std::max(nm, 1) * L);
    uint32_t* lde = arena_.alloc_words((size_t)ncols * Lb);
    PackSel sel{};
    for (uint32_t s = 0; s < G; ++s) sel.blk[s][0] = s;
    launch_pack_blocks(full, L, sendbuf, (uint32_t)Lb, nm, 1, (int)G, sel, stream_);
    for (uint32_t p = 0; p < G; ++p) {
      so[p] = (size_t)p * nm * Lb * 4;
      sb[p] = (size_t)nm * Lb * 4;
      ro[p] = (size_t)out.first[p] * Lb * 4;
      rb[p] = (size_t)(out.first[p + 1] - out.first[p]) * Lb * 4;
    }
    if (shard_.coll.all_to_all(shard_.coll.user, sendbuf, so, sb, lde, ro, rb, (void*)(uintptr_t)stream_) != 0)
      throw LmnError(LMN_ERR_INTERNAL, "shard all_to_all failed");
    timings.shard_a2a_bytes += (uint64_t)(ncols - nm) * Lb * 4;
    timings.shard_a2a_calls++;
    if (halo_first >= 0) {
      // the mask offset -1 of the last logup column group reads the previous trace row, which under bit reversal lies
      // in block rev(rev(b)+1) (odd storage indices) or rev(rev(b)-1) (even ones) of the same 4 columns
      auto nb = [&](uint32_t b, int h) {
        const uint32_t rbv = bit_reverse(b, g);
        return bit_reverse((h == 0 ? rbv + 1 : rbv + G - 1) & (G - 1), g);
      };
      const int h0 = std::max(halo_first, c0), h1 = std::min(halo_first + 4, c0 + nm), nh = std::max(0, h1 - h0);
      uint32_t* hsend = arena_.alloc_words((size_t)std::max(nh, 1) * 2 * Lb * G);
      uint32_t* hrecv = arena_.alloc_words((size_t)4 * 2 * Lb);
      PackSel hs{};
      for (uint32_t s = 0; s < G; ++s) {
        hs.blk[s][0] = nb(s, 0);
        hs.blk[s][1] = nb(s, 1);
      }
      if (nh > 0) launch_pack_blocks(full + (uint64_t)(h0 - c0) * L, L, hsend, (uint32_t)Lb, nh, 2, (int)G, hs, stream_);
      for (uint32_t p = 0; p < G; ++p) {
        const int q0 = std::max(halo_first, out.first[p]), q1 = std::min(halo_first + 4, out.first[p + 1]);
        const int nq = std::max(0, q1 - q0);
        so[p] = (size_t)p * nh * 2 * Lb * 4;
        sb[p] = (size_t)nh * 2 * Lb * 4;
        ro[p] = (size_t)std::max(0, q0 - halo_first) * 2 * Lb * 4;
        rb[p] = (size_t)nq * 2 * Lb * 4;
      }
      if (shard_.coll.all_to_all(shard_.coll.user, hsend, so, sb, hrecv, ro, rb, (void*)(uintptr_t)stream_) != 0)
        throw LmnError(LMN_ERR_INTERNAL, "shard all_to_all failed");
      timings.shard_a2a_bytes += (uint64_t)(4 - nh) * 2 * Lb * 4;
      timings.shard_a2a_calls++;
      out.halo = arena_.alloc_words(4 * L);
      for (int j = 0; j < 4; ++j)
        for (int h = 0; h < 2; ++h)
          lmn_d2d(out.halo + (uint64_t)j * L + (uint64_t)nb(me, h) * Lb, hrecv + ((uint64_t)j * 2 + h) * Lb, Lb * 4, stream_);
    }
    out.lde = lde;
    out.stride = Lb;
    out.sharded = true;
    out.owned = true;
    return out;
  }
  timings.fft_bytes += (uint64_t)ncols * 8ull * n;
  timings.fft_butterflies += (uint64_t)ncols * (n / 2) * (uint64_t)log_size;
  timings.fft_launches += launch_ifft(coeffs, n, evals, n, ncols, log_size, itw(log_size), stream_);
  return out;
}

// columns hold coefficients; produce LDE evaluations (contiguous runs of equal size share launches).  With a
// shard set, only this rank's aligned block of rows of every LDE is evaluated (launch_fft_block: the top
// log2(world) layers collapse to a world-point combination at fixed row, the rest runs inside the block).
void Context::lde_and_merkle(DevTree& tree) {
  const int lb = (int)cfg.log_blowup;
  const bool sh = shard_.active;
  const int g = sh ? shard_.g : 0;
  size_t i = 0;
  while (i < tree.cols.size()) {
    if (tree.cols[i].lde) {  // produced together with the interpolation (interpolate_for_commit)
      ++i;
      continue;
    }
    size_t j = i;
    int log = tree.cols[i].log_size;
    uint64_t n = 1ull << log;
    while (j < tree.cols.size() && !tree.cols[j].lde && tree.cols[j].log_size == log &&
           tree.cols[j].coeffs == tree.cols[i].coeffs + (j - i) * n)
      ++j;
    int ncols = (int)(j - i);
    uint64_t L = (n << lb) >> g;  // rows held here
    uint32_t* lde = arena_.alloc_words((size_t)ncols * L);
    {
      StageTimer t(this, g_log(this), stream_, C_FFT);
      if (g == 0)
        timings.fft_launches += launch_fft(lde, L, tree.cols[i].coeffs, n, log, ncols, log + lb, tw(log + lb), stream_);
      else
        timings.fft_launches += launch_fft_block(lde, L, tree.cols[i].coeffs, n, log, ncols, log + lb, g, shard_.rank,
                                                 tw(log + lb), stream_);
      timings.fft_bytes += (uint64_t)ncols * (4ull * n + 4ull * L);
      timings.fft_butterflies += (uint64_t)ncols * (L / 2) * (uint64_t)(log + lb);
    }
    for (int c = 0; c < ncols; ++c) {
      tree.cols[i + c].lde = lde + (uint64_t)c * L;
      tree.cols[i + c].sharded = sh;
    }
    i = j;
  }
  std::vector<ColRef> sorted;
  for (auto& c : tree.cols) sorted.push_back({c.lde, c.log_size + lb, c.sharded});
  std::stable_sort(sorted.begin(), sorted.end(), [](auto& a, auto& b) { return a.log > b.log; });
  build_merkle(tree.merkle, sorted, nullptr, nullptr, nullptr, sh);
  fetch_root_async(tree.merkle);
}

// ------------------------------------------------------------------------------------ host-side AIR at a point
static QM31 qsub1(QM31 a) { return q_sub_m(a, 1u); }
static QM31 one_minus(QM31 a) { return q_sub(q_one(), a); }

// local constraints at a point, in `evaluate` order (crates/air/src/components/*/component.rs)
static std::vector<QM31> local_constraints(int kind, const std::vector<QM31>& c) {
  std::vector<QM31> out;
  if (kind == LMN_KIND_ADD || kind == LMN_KIND_MUL) {
    QM31 is_last = c[4], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    if (kind == LMN_KIND_ADD) {
      out.push_back(q_sub(c[11], q_add(c[9], c[10])));
    } else {
      out.push_back(q_sub(q_mul(c[9], c[10]), q_add(q_mul_m(c[11], 4096u), c[12])));
      out.push_back(q_zero());
    }
    out.push_back(q_mul(not_last, q_sub(c[5], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[6], c[1])));
    out.push_back(q_mul(not_last, q_sub(c[7], c[2])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[8], c[3]))));
  } else if (kind == LMN_KIND_RECIP) {
    QM31 is_last = c[3], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_sub(q_sqr(c[10]), q_add(q_mul(c[7], c[8]), c[9])));
    out.push_back(q_mul(not_last, q_sub(c[4], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[1])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[6], c[2]))));
  } else if (kind == LMN_KIND_SQRT) {
    QM31 is_last = c[3], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_sub(q_mul(c[7], c[10]), q_add(q_sqr(c[8]), c[9])));
    out.push_back(q_mul(not_last, q_sub(c[4], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[1])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[6], c[2]))));
  } else if (kind == LMN_KIND_REM) {
    QM31 is_last = c[4], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_sub(c[9], q_add(q_mul(c[10], c[12]), c[11])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[6], c[1])));
    out.push_back(q_mul(not_last, q_sub(c[7], c[2])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[8], c[3]))));
  } else if (kind == LMN_KIND_RANGE_CHECK_LOOKUP || kind == LMN_KIND_SIN_LOOKUP || kind == LMN_KIND_EXP2_LOOKUP ||
             kind == LMN_KIND_LOG2_LOOKUP) {
    // no local constraints
  } else if (kind == LMN_KIND_LESS_THAN) {
    QM31 is_last = c[4], not_last = one_minus(is_last), borrow = c[13];
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_mul(borrow, qsub1(borrow)));
    out.push_back(q_sub(c[11], q_mul_m(one_minus(borrow), 4096u)));
    out.push_back(q_sub(q_add(c[9], c[12]), c[10]));  // - borrow * (2^31 - 1) == 0 in M31
    out.push_back(q_sub(c[12], q_add(q_add(q_mul_m(c[17], 1u << 24), q_mul_m(c[16], 1u << 16)),
                                     q_add(q_mul_m(c[15], 1u << 8), c[14]))));
    out.push_back(q_mul(not_last, q_sub(c[5], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[6], c[1])));
    out.push_back(q_mul(not_last, q_sub(c[7], c[2])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[8], c[3]))));
  } else if (kind == LMN_KIND_INPUTS) {
    QM31 is_last = c[2], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_mul(not_last, q_sub(c[3], c[0])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[4], c[1]))));
  } else {  // SumReduce / MaxReduce / Contiguous / Sin / Exp2 / Log2 share the id/idx prefix (columns 0..6)
    QM31 is_last = c[3], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    if (kind == LMN_KIND_SUM_REDUCE) {
      QM31 ils = c[11];
      out.push_back(q_mul(ils, qsub1(ils)));
      out.push_back(q_sub(c[10], q_add(c[9], c[7])));
      out.push_back(q_mul(q_sub(c[8], c[10]), ils));
    } else if (kind == LMN_KIND_MAX_REDUCE) {
      QM31 ils = c[11], im = c[12];
      out.push_back(q_mul(ils, qsub1(ils)));
      out.push_back(q_mul(im, qsub1(im)));
      out.push_back(q_mul(im, q_sub(c[10], c[7])));
      out.push_back(q_mul(one_minus(im), q_sub(c[10], c[9])));
      out.push_back(q_mul(q_sub(c[8], c[10]), ils));
    }
    out.push_back(q_mul(not_last, q_sub(c[4], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[1])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[6], c[2]))));
  }
  return out;
}

QM31 eval_composition_at_point(const std::vector<Instance>& inst,
                               const std::vector<std::vector<std::vector<QM31>>>& sv, QPt oods, const RelElems& elems,
                               QM31 comp_alpha, uint32_t protocol_flags) {
  QM31 acc = q_zero();
  for (auto& ci : inst) {
    const ComponentSpec* sp = ci.spec;
    std::vector<QM31> main(sp->n_cols);
    for (int c = 0; c < sp->n_cols; ++c) main[c] = sv[1][ci.main_start + c][0];
    std::vector<QM31> cons = local_constraints(sp->kind, main);
    QM31 prev = q_zero();
    QM31 shift = q_mul_m(ci.claimed, m_inv((uint32_t)((1ull << ci.log_size) % P31)));
    for (int j = 0; j < sp->n_rel; ++j) {
      const auto* cols = &sv[2][ci.inter_start + 4 * j];
      auto cell = [&](int idx) { return sp->rel_pre[j] ? sv[0][ci.pre_idx[idx]][0] : main[idx]; };
      const int es = sp->rel_elems[j];
      QM31 den = q_sub(cell(sp->rel_val[j]), elems.z[es]);
      if (sp->rel_id[j] >= 0) den = q_add(den, q_mul(elems.alpha[es], cell(sp->rel_id[j])));
      QM31 num = sp->rel_neg[j] ? q_neg(main[sp->rel_mult[j]]) : main[sp->rel_mult[j]];
      QM31 cur, diff;
      if (j < sp->n_rel - 1) {
        cur = q_from_partial_evals(cols[0][0], cols[1][0], cols[2][0], cols[3][0]);
        diff = q_sub(cur, prev);
      } else {
        QM31 prev_row = q_from_partial_evals(cols[0][0], cols[1][0], cols[2][0], cols[3][0]);
        cur = q_from_partial_evals(cols[0][1], cols[1][1], cols[2][1], cols[3][1]);
        diff = q_add(q_sub(q_sub(cur, prev_row), prev), shift);
      }
      cons.push_back(q_sub(q_mul(diff, den), num));
      prev = cur;
    }
    QM31 x = oods.x;
    for (int k = 0; k < ci.log_size - 1; ++k) x = q_sub_m(q_add(q_sqr(x), q_sqr(x)), 1u);
    QM31 zinv = q_inv(x);
    // kernel-slot values -> the protocol's constraint list (constraint-form bits: slots added / dropped, signs)
    const ConstraintLayout L = constraint_layout(*sp, protocol_flags);
    std::vector<QM31> proto(L.n_protocol, q_zero());
    for (int k = 0; k < L.n_kernel; ++k)
      if (L.proto_index[k] >= 0) proto[L.proto_index[k]] = L.neg[k] ? q_neg(cons[k]) : cons[k];
    for (auto& c : proto) acc = q_add(q_mul(acc, comp_alpha), q_mul(c, zinv));
  }
  return acc;
}

// ------------------------------------------------------------------------------------ OODS evaluation
std::vector<QM31> Context::eval_at_points(const std::vector<EvalJob>& jobs, const std::vector<QPt>& points,
                                          int max_log, bool split) {
  const int np = (int)points.size();
  const uint32_t lo_n = 1u << EVAL_LB;
  const int hi_bits = max_log > EVAL_LB ? max_log - EVAL_LB : 0;
  const uint32_t hi_n = 1u << hi_bits;
  const int nmaps = std::max(max_log, EVAL_LB);
  // mappings per point: y, x, pi(x), pi^2(x), ...  (tables themselves are expanded on the device)
  std::vector<QM31> maps((size_t)np * nmaps);
  for (int p = 0; p < np; ++p) {
    QM31* mp = &maps[(size_t)p * nmaps];
    mp[0] = points[p].y;
    mp[1] = points[p].x;
    QM31 cur = points[p].x;
    for (int k = 2; k < nmaps; ++k) {
      cur = q_sub_m(q_add(q_sqr(cur), q_sqr(cur)), 1u);
      mp[k] = cur;
    }
  }
  int max_chunks = eval_num_chunks(max_log);
  EvalJob* d_jobs = upload_vec(jobs);
  QM31* d_maps = upload_vec(maps);
  QM31* d_lo = (QM31*)arena_.alloc_bytes((size_t)np * lo_n * sizeof(QM31));
  QM31* d_hi = (QM31*)arena_.alloc_bytes((size_t)np * hi_n * sizeof(QM31));
  QM31* d_part = (QM31*)arena_.alloc_bytes(jobs.size() * (size_t)max_chunks * sizeof(QM31));
  split = split && shard_.active && shard_.world > 1;
  const uint32_t W = split ? shard_.world : 1u, R = split ? shard_.rank : 0u;
  const size_t nj = jobs.size();
  QM31* d_out = (QM31*)arena_.alloc_bytes((size_t)W * nj * sizeof(QM31));   // slot r: rank r's partial sums
  launch_eval_tables(d_maps, nmaps, np, d_lo, d_hi, hi_n, hi_bits, stream_);
  launch_eval_at_point(d_jobs, (int)nj, d_lo, d_hi, hi_n, max_log, d_part, max_chunks, stream_, R, W);
  launch_eval_reduce(d_jobs, (int)nj, d_part, max_chunks, d_out + (size_t)R * nj, stream_);
  if (split) gather_columns((uint32_t*)d_out, 0, 1, nj * 4);
  const QM31* res = (const QM31*)stage_download(d_out, (size_t)W * nj * sizeof(QM31));
  lmn_sync(stream_);
  std::vector<QM31> out(res, res + nj);
  for (uint32_t r = 1; r < W; ++r)
    for (size_t j = 0; j < nj; ++j) out[j] = q_add(out[j], res[(size_t)r * nj + j]);
  return out;
}

// ------------------------------------------------------------------------------------ decommit planning
// A run of device words to fetch.  owner < 0: every rank holds it; otherwise only rank `owner` does (row-block
// sharded column or Merkle layer) and ptr is meaningful on that rank alone.
struct Ref {
  const uint32_t* ptr;
  uint32_t len;
  int owner;
  int job = -1;   // >= 0: not in memory - Merkle node to recompute (index into the plan's MerkleRecompute list), ptr is null
};
static Ref col_ref(const ColRef& c, uint64_t row, int g) {
  if (!c.sharded) return {c.ptr + row, 1, -1};
  const int sh = c.log - g;
  return {c.ptr + (row & ((1ull << sh) - 1)), 1, (int)(row >> sh)};
}
static Ref node_ref(const DevMerkle& m, int layer, uint64_t node, std::vector<MerkleRecompute>& jobs) {
  if (!m.layers[layer]) {   // a level its launch kept in registers (MerkleCut)
    for (auto& c : m.cuts)
      if (layer <= c.start_log && layer > c.start_log - c.depth) {
        jobs.push_back({c.prev, c.sg, c.ncols, 1u << c.start_log, c.below, c.below_ncols, (uint32_t)node, c.start_log - layer, 0u});
        return {nullptr, 8, -1, (int)jobs.size() - 1};
      }
    throw LmnError(LMN_ERR_INTERNAL, "merkle: layer without storage");
  }
  if (m.g == 0 || layer <= m.g) return {m.layers[layer] + node * 8, 8, -1};
  const int sh = layer - m.g;
  return {m.layers[layer] + (node & ((1ull << sh) - 1)) * 8, 8, (int)(node >> sh)};
}

// Decommitment plan of one tree / FRI layer, and the per-context scratch that keeps the plans' storage alive across
// proofs (the planning runs on the host between the last FRI sync and the gather launch, i.e. on the critical
// path of the proof's latency: no allocations there after the first proof).
struct DecommitPlan {
  std::vector<Ref> fri_wit, queried, hash_wit, col_wit;
  void clear() {
    fri_wit.clear();
    queried.clear();
    hash_wit.clear();
    col_wit.clear();
  }
};
struct HostScratch {
  std::vector<DecommitPlan> plans;
  size_t used = 0;
  std::vector<GatherEntry> entries;
  std::vector<MerkleRecompute> jobs;
  std::vector<std::pair<int, uint32_t>> runs;
  std::vector<ColRef> cols;
  DecommitPlan& next() {
    if (used == plans.size()) plans.emplace_back();
    DecommitPlan& p = plans[used++];
    p.clear();
    return p;
  }
};

static void release_host_scratch(void* p) { delete static_cast<HostScratch*>(p); }

// MerkleProver::decommit: emits device references in output order
static void plan_merkle_decommit(const DevMerkle& m, const std::vector<ColRef>& cols_sorted, int g,
                                 const std::map<int, std::vector<uint32_t>>& queries, std::vector<Ref>& queried,
                                 std::vector<Ref>& hash_wit, std::vector<Ref>& col_wit, std::vector<MerkleRecompute>& jobs) {
  size_t pos = 0;
  std::vector<uint32_t> last, total;
  last.reserve(16);
  total.reserve(16);
  for (int log = m.max_log; log >= 0; --log) {
    size_t start = pos;
    while (pos < cols_sorted.size() && cols_sorted[pos].log == log) ++pos;
    bool have_prev = log < m.max_log;
    static const std::vector<uint32_t> kNone;
    auto it = queries.find(log);
    const std::vector<uint32_t>& colq = it != queries.end() ? it->second : kNone;
    size_t pi = 0, ci = 0;
    total.clear();
    while (pi < last.size() || ci < colq.size()) {
      uint32_t node;
      if (pi < last.size() && ci < colq.size())
        node = std::min(last[pi] / 2, colq[ci]);
      else if (pi < last.size())
        node = last[pi] / 2;
      else
        node = colq[ci];
      if (have_prev) {
        if (pi < last.size() && last[pi] == 2 * node)
          ++pi;
        else
          hash_wit.push_back(node_ref(m, log + 1, 2ull * node, jobs));
        if (pi < last.size() && last[pi] == 2 * node + 1)
          ++pi;
        else
          hash_wit.push_back(node_ref(m, log + 1, 2ull * node + 1, jobs));
      }
      bool is_q = ci < colq.size() && colq[ci] == node;
      if (is_q) ++ci;
      for (size_t c = start; c < pos; ++c) (is_q ? queried : col_wit).push_back(col_ref(cols_sorted[c], node, g));
      total.push_back(node);
    }
    last.swap(total);
  }
}

static std::vector<uint32_t> fold_positions(const std::vector<uint32_t>& p, int n) {
  std::vector<uint32_t> out;
  for (auto v : p) {
    uint32_t q = v >> n;
    if (out.empty() || out.back() != q) out.push_back(q);
  }
  return out;
}

// compute_decommitment_positions_and_witness_evals with fold_step = 1; `cols` = the 4 coordinate columns
static void plan_fri_witness(const ColRef (&cols)[4], int g, const std::vector<uint32_t>& qpos,
                             std::vector<uint32_t>& dec_pos, std::vector<Ref>& wit) {
  size_t i = 0;
  while (i < qpos.size()) {
    uint32_t start = (qpos[i] >> 1) << 1;
    std::vector<uint32_t> subset;
    while (i < qpos.size() && ((qpos[i] >> 1) << 1) == start) subset.push_back(qpos[i++]);
    for (uint32_t pos = start; pos < start + 2; ++pos) {
      dec_pos.push_back(pos);
      if (std::find(subset.begin(), subset.end(), pos) != subset.end()) continue;
      for (int k = 0; k < 4; ++k) wit.push_back(col_ref(cols[k], pos, g));
    }
  }
}

// optional host-side wall-clock marks (LMN_HOST_PROFILE=1), printed to stderr
struct HostMarks {
  bool on = getenv("LMN_HOST_PROFILE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  void mark(const char* what) {
    if (!on) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[host] %-28s +%8.1f us  (t=%8.1f)\n", what,
            std::chrono::duration<double, std::micro>(now - last).count(),
            std::chrono::duration<double, std::micro>(now - t0).count());
    last = now;
  }
};

// ------------------------------------------------------------------------------------ FRI quotients
// Host side of `accumulate_quotients` for the columns of one LDE size: ColumnSampleBatch::new_vec groups
// the (column, point, value) samples by point in first-appearance order; per sample the line through
// (p.y, v) and (conj p.y, conj v) gives coefficients a, b, c scaled by alpha^k.
QuotientArgs Context::make_quotient_args(int ls, const std::vector<const uint32_t*>& cols,
                                         const std::vector<std::vector<std::pair<int, QM31>>>& samples,
                                         const std::vector<QPt>& points, QM31 quot_alpha, bool alloc_out) {
  std::vector<int> batch_point;
  std::vector<std::vector<std::pair<int, QM31>>> batch_cols;
  for (size_t c = 0; c < cols.size(); ++c)
    for (auto& sm : samples[c]) {
      size_t b = 0;
      while (b < batch_point.size() && batch_point[b] != sm.first) ++b;
      if (b == batch_point.size()) {
        batch_point.push_back(sm.first);
        batch_cols.emplace_back();
      }
      batch_cols[b].push_back({(int)c, sm.second});
    }
  if (batch_point.size() > (size_t)QUOT_MAX_BATCH) throw LmnError(LMN_ERR_INTERNAL, "too many sample batches");
  QuotientArgs a{};
  a.log_size = ls;
  a.nbatch = (int)batch_point.size();
  std::vector<int> col_idx;
  std::vector<QM31> coeff_c;
  for (size_t b = 0; b < batch_point.size(); ++b) {
    QPt pt = points[batch_point[b]];
    a.batch_start[b] = (int)col_idx.size();
    QM31 alpha = q_one(), A = q_zero(), B = q_zero();
    for (auto& cv : batch_cols[b]) {
      alpha = q_mul(alpha, quot_alpha);
      QM31 val = cv.second;
      QM31 la = q_sub(q_conj(val), val);
      QM31 lc = q_sub(q_conj(pt.y), pt.y);
      QM31 lbb = q_sub(q_mul(val, lc), q_mul(la, pt.y));
      A = q_add(A, q_mul(alpha, la));
      B = q_add(B, q_mul(alpha, lbb));
      col_idx.push_back(cv.first);
      coeff_c.push_back(q_mul(alpha, lc));
    }
    a.A[b] = A;
    a.B[b] = B;
    a.batch_coeff[b] = q_pow(quot_alpha, batch_cols[b].size());
    a.prx[b] = {pt.x.a, pt.x.b};
    a.pix[b] = {pt.x.c, pt.x.d};
    a.pry[b] = {pt.y.a, pt.y.b};
    a.piy[b] = {pt.y.c, pt.y.d};
  }
  a.batch_start[batch_point.size()] = (int)col_idx.size();
  if (col_idx.size() > (size_t)QUOT_MAX_ENTRIES) throw LmnError(LMN_ERR_INTERNAL, "too many column samples");
  std::vector<QuotEntry> entries(col_idx.size());
  for (size_t k = 0; k < col_idx.size(); ++k) entries[k] = {cols[col_idx[k]], coeff_c[k]};
  a.entries = upload_vec(entries);
  a.tw_y = twY_[ls];
  a.tw_x = ls >= 2 ? twX_[ls] : nullptr;
  a.row0 = 0;
  a.log_rows = ls;
  a.out_stride = 1ull << ls;
  a.out = alloc_out ? arena_.alloc_words(4ull << ls) : nullptr;
  return a;
}

#ifdef LMN_BATCH
void Context::prepare_for(const lmn_table* tables, size_t n_tables) {
  LMN_HIP_CHECK(hipSetDevice(device_));
  if (!tables) return;
  int max_log = 0;
  for (size_t t = 0; t < n_tables; ++t) {
    if (tables[t].n_rows == 0 || tables[t].n_rows > (1ull << 26)) return;
    int ls = 4;
    while ((1ull << ls) < tables[t].n_rows) ++ls;
    max_log = std::max(max_log, ls);
  }
  const int max_lde = max_log + 1 + (int)cfg.log_blowup;
  if (max_lde > MAX_LOG - 2) return;
  ensure_twiddles(max_lde);
}
#endif

// ------------------------------------------------------------------------------------ prove
std::vector<uint8_t> Context::prove(const lmn_table* tables, size_t n_tables, const lmn_settings* settings) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (!tables || n_tables == 0) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "no trace tables");
  const int lb = (int)cfg.log_blowup;
  const int n_slots = claim_slots(cfg.protocol_variant);
  HostMarks hm;
  EventLog* log = g_log(this);
  log->reset();
  memset(&timings, 0, sizeof timings);
#ifndef LMN_EMU
  // LMN_FRI_OVERLAP: a previous proof that failed between the fork and the join may have left a kernel of the second
  // stream writing the arena this proof is about to reset
  if (have_stream2_) lmn_sync(stream2_);
#endif
  wait_before_level_ = -1;
  // big trees are stored without the levels their fused launches keep in registers (MerkleCut); sharded proofs and the
  // level-2 ops (whose handles expose every layer) keep whole trees
  struct CutScope {
    bool& flag;
    ~CutScope() { flag = false; }
  } cut_scope{merkle_cut_};
  merkle_cut_ = !shard_.active && getenv("LMN_MERKLE_FULL") == nullptr;

  // ---- validate + size
  struct TableInfo {
    const ComponentSpec* spec;
    uint64_t n_rows;
    int log_size;
    const uint32_t* rows;
    bool on_device;
  };
  std::vector<TableInfo> infos;
  int max_log = 0;
  int prev_kind = -1;
  size_t words = 0;
  // a sharded context holds only its row block of every LDE / Merkle layer / large FRI layer
  const int size_g = shard_.active ? shard_.g : 0;
  auto row_split = [&](uint64_t w) { return size_g ? (w >> size_g) + 8192 : w; };
  for (size_t t = 0; t < n_tables; ++t) {
    const lmn_table& tb = tables[t];
    const ComponentSpec* sp = component_spec((int)tb.kind);
    if (!sp) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "unsupported component kind " + std::to_string(tb.kind));
    if ((int)tb.kind >= n_slots)
      throw LmnError(LMN_ERR_INVALID_ARGUMENT, "component kind has no claim slot in this protocol variant");
    if ((int)tb.kind <= prev_kind)
      throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tables must be in gen_trace order (ascending kind, no duplicates)");
    prev_kind = (int)tb.kind;
    if (tb.n_rows == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
    if (!tb.rows) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "null rows pointer");
    if (tb.n_rows > (1ull << 26)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace table has more than 2^26 rows");
    uint64_t size = 16;
    while (size < tb.n_rows) size <<= 1;
    int ls = 0;
    while ((1ull << ls) < size) ++ls;
    if (ls + lb + 2 > MAX_LOG - 2) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace too large");
    infos.push_back({sp, tb.n_rows, ls, tb.rows, (tb.flags & LMN_TABLE_ROWS_ON_DEVICE) != 0});
    max_log = std::max(max_log, ls);
    uint64_t cells = (uint64_t)(sp->n_cols + 4 * sp->n_rel) << ls;
    words += cells * 2 + row_split(cells << lb);               // evals + coeffs + lde (this rank's row block)
    if (!infos.back().on_device) words += tb.n_rows * sp->n_cols;  // staging
    words += (uint64_t)sp->n_pre * ((2ull << ls) + row_split(2ull << ls));  // preprocessed columns: evals + coeffs + lde
    words += (4ull << ls) * 2;                                 // logup temps
    words += (4ull << (ls + 1)) * 3;                           // per-size composition scratch
    if (shard_.active) words += (4ull << (ls + 1)) + 4096;     // halo rows of the last logup column group
    if (shard_rows_front(ls))                                  // received row blocks, gathered logup sums, scan output, coefficients
      words += (uint64_t)(sp->n_cols + 8 * sp->n_rel + 16) << ls;
    if (shard_all_to_all()) {                                  // own columns over all rows, packed copy, halo exchange
      const uint64_t G = shard_.world;
      words += 2 * (((uint64_t)sp->n_cols / G + 1) + ((uint64_t)(4 * sp->n_rel) / G + 1)) * (2ull << ls) + 6 * (2ull << ls);
    }
  }
  const int comp_log = max_log + 1;
  const int max_lde = comp_log + lb;
  words += (4ull << comp_log) * 2 + row_split(4ull << max_lde);  // composition values/coeffs + lde
  if (shard_all_to_all()) words += 2ull << max_lde;              // one composition column over all rows + its packed copy
  words += row_split((4ull << max_lde) * 3);                   // quotient columns (all sizes) + fri layers
  if (merkle_cut_ && max_lde >= 21)
    // trees without their register levels (MerkleCut): 1/8 of the nodes of a tree of 2^20 leaves and more
    // (bounds: up to four trees whose leaf level has more than 8 columns next to a smaller component's - that level is a
    // launch of its own and stays whole - and 1/8 + the block tops of everything else)
    words += 4 * (8ull << max_lde) + 7 * (3ull << max_lde) + (16ull << 20);
  else
    words += row_split(7 * (16ull << max_lde));                // merkle trees (4 trace + fri first + inner)
  if (shard_.active) words += 64ull << std::min(max_lde, std::max(shard_.fri_min_log, 12) + 2);  // replicated small FRI layers + their trees
  words += (16u << 20);                                        // slack: tables, partials, gather buffers
  ensure_twiddles(max_lde);
  arena_.reserve(words * 4);
  arena_.reset();
  pin_off_ = 0;

  Channel channel(cfg.protocol_variant);
  Proof proof;
  proof.claim.assign(n_slots, -1);
  proof.interaction_claim.assign(n_slots, {false, q_zero()});
  proof.pow_bits = cfg.pow_bits;
  proof.log_blowup = cfg.log_blowup;
  proof.log_last_layer = cfg.log_last_layer;
  proof.n_queries = cfg.n_queries;

  StageTimer* total_timer = new StageTimer(this, log, stream_, C_TOTAL);
  std::unique_ptr<StageTimer> total_guard(total_timer);

  // ---- PHASE 0: preprocessed trace (prover.rs:54-59): empty tree (root = blake2s("")) unless a lookup
  // component is present.  Columns in PreProcessedTrace order (preprocessed.rs:157-179: sin, exp2, log2
  // LUT pairs from the settings, then the 8-bit range check whose row r holds r), stable-sorted by size
  // descending (PreProcessedTrace::new).
  DevTree tree0;
  std::vector<Instance> inst;
  for (auto& ti : infos) {
    Instance ci{};
    ci.spec = ti.spec;
    ci.log_size = ti.log_size;
    inst.push_back(ci);
  }
  std::vector<uint32_t*> pre_evals;  // tree-0 columns on their trace domain (logup denominators)
  {
    uint32_t present = 0;
    const lmn_lut* lut_of[3] = {nullptr, nullptr, nullptr};
    for (auto& ti : infos) {
      if (ti.spec->kind == LMN_KIND_SIN_LOOKUP) present |= LMN_LOOKUP_SIN;
      if (ti.spec->kind == LMN_KIND_EXP2_LOOKUP) present |= LMN_LOOKUP_EXP2;
      if (ti.spec->kind == LMN_KIND_LOG2_LOOKUP) present |= LMN_LOOKUP_LOG2;
      if (ti.spec->kind == LMN_KIND_RANGE_CHECK_LOOKUP) present |= LMN_LOOKUP_RANGE_CHECK;
    }
    if (settings && (settings->has_lookups & ~present))
      throw LmnError(LMN_ERR_INVALID_ARGUMENT, "settings announce a lookup whose table is not in the pie");
    if (settings && settings->n_luts) {
      if (!settings->luts) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "null luts pointer");
      for (uint32_t i = 0; i < settings->n_luts; ++i) {
        const lmn_lut& l = settings->luts[i];
        if (l.kind > LMN_LUT_LOG2 || !l.col0 || !l.col1 || lut_of[l.kind])
          throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad or duplicate LUT in settings");
        lut_of[l.kind] = &l;
      }
    }
    std::vector<int> logs = assign_preprocessed(inst);
    tree0.cols.resize(logs.size());
    pre_evals.resize(logs.size(), nullptr);
    for (auto& ci : inst) {
      const ComponentSpec* sp = ci.spec;
      for (int k = 0; k < sp->n_pre; ++k) {
        const uint64_t n = 1ull << ci.log_size;
        uint32_t* evals;
        if (sp->pre_id[k] == PRE_RANGE_CHECK) {
          if (ci.log_size != 8) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "RangeCheckLookup table must have exactly 256 rows");
          std::vector<uint32_t> lut(n);
          for (uint32_t r = 0; r < n; ++r) lut[r] = r;
          evals = upload_vec(lut);
        } else {
          const lmn_lut* l = lut_of[sp->pre_id[k] / 2];
          if (!l) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "lookup table present but settings carry no LUT columns for it");
          if ((int)l->log_size != ci.log_size)
            throw LmnError(LMN_ERR_INVALID_ARGUMENT, "lookup table rows must match the LUT column size");
          const uint32_t* src = (sp->pre_id[k] & 1) ? l->col1 : l->col0;
          for (uint64_t r = 0; r < n; ++r)
            if (src[r] >= P31) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "LUT value is not a canonical M31");
          evals = arena_.alloc_words(n);
          lmn_h2d(evals, src, n * 4, stream_);
        }
        uint32_t* coeffs = arena_.alloc_words(n);
        launch_ifft(coeffs, n, evals, n, 1, ci.log_size, itw(ci.log_size), stream_);
        tree0.cols[ci.pre_idx[k]] = {ci.log_size, coeffs, nullptr};
        pre_evals[ci.pre_idx[k]] = evals;
      }
    }
    if (!tree0.cols.empty()) {
      lde_and_merkle(tree0);
      lmn_sync(stream_);
      tree0.merkle.finish_root();
    } else {
      build_merkle(tree0.merkle, {});
    }
  }
  channel.mix_root(tree0.merkle.root);

  // ---- PHASE 1: main trace (prover.rs:70-179)
  DevTree tree1;
  // Persistent device word the transposes write when a table holds a word that is not a canonical M31.  Unsharded proofs
  // never reset it: every proof has its own mark (>= 2) and only that value counts, so the accepting and the rejecting
  // path issue the same launches / copies / waits - what the lock-step batch library needs from its members (a
  // rejected pie leaves its batch alone and the slot stays usable).  Sharded proofs gather the word across ranks, whose
  // counters are unrelated: their mark is 1 and the rejecting path clears it.
  uint32_t* d_bad = bad_flag_;
  if (++bad_epoch_ < 2u) bad_epoch_ = 2u;
  const uint32_t bad_mark = shard_.active ? 1u : bad_epoch_;
  const uint32_t* h_bad = nullptr;
  bool any_rows_front = false;
  {
    StageTimer st(this, log, stream_, C_TRANSPOSE);
    for (size_t t = 0; t < infos.size(); ++t) {
      auto& ti = infos[t];
      uint64_t n = 1ull << ti.log_size;
      const bool rows_front = shard_rows_front(ti.log_size);
      // row-parallel front end of a sharded proof: only this rank's block of the (padded) rows is transposed - and, for
      // host tables, uploaded
      const uint64_t nb = rows_front ? n >> shard_.g : n, blk0 = rows_front ? (uint64_t)shard_.rank * nb : 0;
      const uint64_t up0 = std::min<uint64_t>(blk0, ti.n_rows), up1 = std::min<uint64_t>(blk0 + nb, ti.n_rows);
      const uint32_t* d_rows = ti.rows;
      if (!ti.on_device) {
        uint32_t* stg = arena_.alloc_words(std::max<uint64_t>(up1 - up0, 1) * ti.spec->n_cols);
        if (up1 > up0) lmn_h2d(stg, ti.rows + up0 * ti.spec->n_cols, (up1 - up0) * ti.spec->n_cols * 4, stream_);
        d_rows = stg - up0 * ti.spec->n_cols;   // indexed by table row: only rows [up0, up1) are ever read
      }
      uint32_t* evals = arena_.alloc_words((size_t)ti.spec->n_cols * nb);
      PadRow pad{};
      if (ti.spec->is_last_col >= 0) pad.v[ti.spec->is_last_col] = 1u;
      for (int k = 0; k < ti.spec->n_pad; ++k) pad.v[ti.spec->pad_col[k]] = ti.spec->pad_val[k];
      launch_transpose_pad_rows(d_rows, ti.n_rows, ti.spec->n_cols, ti.log_size, evals, nb, blk0, nb, pad, d_bad, stream_, bad_mark);
      inst[t].trace_evals = evals;
      inst[t].rows_sharded = rows_front;
      any_rows_front = any_rows_front || rows_front;
      proof.claim[ti.spec->kind] = ti.log_size;
    }
  }
  {
    StageTimer st(this, log, stream_, C_MAIN_COMMIT);
    int off = 0;
    for (auto& ci : inst) {
      uint64_t n = 1ull << ci.log_size;
      int nc = ci.spec->n_cols;
      uint32_t* coeffs = arena_.alloc_words((size_t)nc * n);
      const CommitOut co = interpolate_for_commit(coeffs, ci.trace_evals, nc, ci.log_size, -1, ci.rows_sharded);
      ci.main_start = off;
      off += nc;
      for (int c = 0; c < nc; ++c)
        tree1.cols.push_back({ci.log_size, coeffs + (uint64_t)c * n, co.lde ? co.lde + (uint64_t)c * co.stride : nullptr,
                              co.sharded, co.owner_of(c)});
    }
    for (int k = 0; k < n_slots; ++k)  // LuminairClaim::mix_into (crates/air/src/lib.rs:52-104)
      if (proof.claim[k] >= 0) channel.mix_u64((uint64_t)proof.claim[k]);
    lde_and_merkle(tree1);
    uint32_t n_flags = 1;
    if (any_rows_front) {   // every rank has only looked at its own rows: the ranks must agree on the verdict
      n_flags = shard_.world;
      uint32_t* flags = arena_.alloc_words(n_flags);
      lmn_d2d(flags + shard_.rank, d_bad, 4, stream_);
      gather_columns(flags, 0, 1, 1);
      h_bad = (const uint32_t*)stage_download(flags, 4 * n_flags);
    } else {
      h_bad = (const uint32_t*)stage_download(d_bad, 4);
    }
    lmn_sync(stream_);
    bool bad_any = false;
    for (uint32_t k = 0; k < n_flags; ++k) bad_any = bad_any || h_bad[k] == bad_mark;
    if (bad_any) {
      if (shard_.active) {
        const uint32_t zero = 0u;
        lmn_h2d(d_bad, &zero, 4, stream_);
        lmn_sync(stream_);
      }
      throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace table holds a word that is not a canonical M31 (>= 2^31-1)");
    }
    tree1.merkle.finish_root();
    channel.mix_root(tree1.merkle.root);
  }
  hm.mark("sync1: root1 mixed");

  // ---- PHASE 2: interaction trace (prover.rs:186-298)
  const RelElems elems = draw_relation_elements(channel, cfg.protocol_variant);
  DevTree tree2;
  {
    StageTimer st(this, log, stream_, C_LOGUP);
    int off = 0;
    for (auto& ci : inst) {
      const ComponentSpec* sp = ci.spec;
      uint64_t n = 1ull << ci.log_size;
      int nic = 4 * sp->n_rel;
      if (ci.rows_sharded) {
        // ---- row-parallel logup: fractions and running sums of this rank's row block; the claimed sum and the
        // coset-order prefix sum of the last column need all rows - 16 bytes per rank and 16 bytes per row are gathered
        const uint32_t G = shard_.world, me = shard_.rank;
        const uint64_t nb = n >> shard_.g, row0 = (uint64_t)me * nb;
        uint32_t* iblk = arena_.alloc_words((size_t)nic * nb);
        QM31* last_full = (QM31*)arena_.alloc_bytes(n * sizeof(QM31));
        LogupArgs a{};
        a.k = sp->n_rel;
        for (int j = 0; j < sp->n_rel; ++j) {
          const int es = sp->rel_elems[j];
          if (!elems.drawn[es])
            throw LmnError(LMN_ERR_INVALID_ARGUMENT, "component needs relation elements this protocol variant does not draw");
          auto column = [&](int idx) -> const uint32_t* {
            return sp->rel_pre[j] ? pre_evals[ci.pre_idx[idx]] + row0 : ci.trace_evals + (uint64_t)idx * nb;
          };
          a.val[j] = column(sp->rel_val[j]);
          a.id[j] = sp->rel_id[j] >= 0 ? column(sp->rel_id[j]) : nullptr;
          a.mult[j] = ci.trace_evals + (uint64_t)sp->rel_mult[j] * nb;
          a.neg[j] = sp->rel_neg[j];
          a.z[j] = elems.z[es];
          a.alpha[j] = elems.alpha[es];
        }
        a.inter = iblk;
        a.last_tmp = last_full + row0;
        const int nbk = logup_num_blocks((uint32_t)nb);
        a.partials = arena_.alloc_words((size_t)nbk * 4);
        a.n = (uint32_t)nb;
        launch_logup_fracs(a, stream_);
        QM31* local = (QM31*)arena_.alloc_bytes(2 * sizeof(QM31));
        launch_logup_reduce(a.partials, nbk, 1u, local, stream_);             // local[0] = sum over this rank's rows
        uint32_t* slots = arena_.alloc_words(4 * (size_t)G);
        lmn_d2d(slots + 4 * me, local, sizeof(QM31), stream_);
        gather_columns(slots, 0, 1, 4);
        QM31* d_cs = (QM31*)arena_.alloc_bytes(2 * sizeof(QM31));
        launch_logup_reduce(slots, (int)G, m_inv((uint32_t)(n % P31)), d_cs, stream_);   // claimed sum, shift
        gather_columns((uint32_t*)last_full, 0, 1, nb * 4);
        uint32_t* scan_out = arena_.alloc_words(4 * n);
        QM31* bsums = (QM31*)arena_.alloc_bytes((size_t)logup_scan_num_blocks(ci.log_size) * sizeof(QM31));
        launch_logup_scan(last_full, d_cs, ci.log_size, scan_out, bsums, stream_);
        for (int k = 0; k < 4; ++k)
          lmn_d2d(iblk + (uint64_t)(nic - 4 + k) * nb, scan_out + (uint64_t)k * n + row0, nb * 4, stream_);
        ci.d_claimed_shift = d_cs;
        ci.inter_start = off;
        off += nic;
        uint32_t* icoeffs = arena_.alloc_words((size_t)nic * n);
        const CommitOut co = interpolate_for_commit(icoeffs, iblk, nic, ci.log_size, nic - 4, true);
        ci.halo = co.halo;
        for (int c = 0; c < nic; ++c)
          tree2.cols.push_back({ci.log_size, icoeffs + (uint64_t)c * n, co.lde + (uint64_t)c * co.stride, co.sharded, co.owner_of(c)});
        continue;
      }
      uint32_t* ievals = arena_.alloc_words((size_t)nic * n);
      LogupArgs a{};
      a.k = sp->n_rel;
      for (int j = 0; j < sp->n_rel; ++j) {
        const int es = sp->rel_elems[j];
        if (!elems.drawn[es])
          throw LmnError(LMN_ERR_INVALID_ARGUMENT, "component needs relation elements this protocol variant does not draw");
        auto column = [&](int idx) -> const uint32_t* {
          return sp->rel_pre[j] ? pre_evals[ci.pre_idx[idx]] : ci.trace_evals + (uint64_t)idx * n;
        };
        a.val[j] = column(sp->rel_val[j]);
        a.id[j] = sp->rel_id[j] >= 0 ? column(sp->rel_id[j]) : nullptr;
        a.mult[j] = ci.trace_evals + (uint64_t)sp->rel_mult[j] * n;
        a.neg[j] = sp->rel_neg[j];
        a.z[j] = elems.z[es];
        a.alpha[j] = elems.alpha[es];
      }
      a.inter = ievals;
      a.last_tmp = (QM31*)arena_.alloc_bytes(n * sizeof(QM31));
      int nb = logup_num_blocks((uint32_t)n);
      a.partials = arena_.alloc_words((size_t)nb * 4);
      a.n = (uint32_t)n;
      launch_logup_fracs(a, stream_);
      QM31* d_cs = (QM31*)arena_.alloc_bytes(2 * sizeof(QM31));
      uint32_t n_inv = m_inv((uint32_t)(n % P31));
      launch_logup_reduce(a.partials, nb, n_inv, d_cs, stream_);
      QM31* bsums = (QM31*)arena_.alloc_bytes((size_t)logup_scan_num_blocks(ci.log_size) * sizeof(QM31));
      launch_logup_scan(a.last_tmp, d_cs, ci.log_size, ievals + (uint64_t)(nic - 4) * n, bsums, stream_);
      ci.d_claimed_shift = d_cs;
      ci.inter_start = off;
      off += nic;
      // interaction evals -> coefficients in place, registered as tree-2 columns
      const CommitOut co = interpolate_for_commit(ievals, ievals, nic, ci.log_size, nic - 4);
      ci.halo = co.halo;
      for (int c = 0; c < nic; ++c)
        tree2.cols.push_back({ci.log_size, ievals + (uint64_t)c * n, co.lde ? co.lde + (uint64_t)c * co.stride : nullptr,
                              co.sharded, co.owner_of(c)});
    }
  }
  {
    // commit the interaction tree first (it does not depend on the transcript), then fetch the
    // claimed sums and the root with a single synchronisation
    StageTimer st(this, log, stream_, C_INTER_COMMIT);
    lde_and_merkle(tree2);
    std::vector<const QM31*> cs(inst.size());
    for (size_t i = 0; i < inst.size(); ++i)
      cs[i] = (const QM31*)stage_download(inst[i].d_claimed_shift, 2 * sizeof(QM31));
    lmn_sync(stream_);
    tree2.merkle.finish_root();
    for (size_t i = 0; i < inst.size(); ++i) {
      inst[i].claimed = cs[i][0];
      proof.interaction_claim[inst[i].spec->kind] = {true, cs[i][0]};
    }
  }
  for (int k = 0; k < n_slots; ++k)
    if (proof.interaction_claim[k].first) channel.mix_felts({proof.interaction_claim[k].second});
  channel.mix_root(tree2.merkle.root);
  hm.mark("sync2: claims+root2 mixed");

  // ---- stwo::prover::prove (prover.rs:312): composition polynomial
  const QM31 comp_alpha = channel.draw_felt();
  int n_total = 0;
  for (auto& ci : inst) n_total += constraint_layout(*ci.spec, cfg.protocol_variant).n_protocol;
  std::vector<QM31> powers(n_total);
  powers[0] = q_one();
  for (int k = 1; k < n_total; ++k) powers[k] = q_mul(powers[k - 1], comp_alpha);
  DevTree tree3;
  {
    StageTimer st(this, log, stream_, C_COMPOSITION);
    std::map<int, uint32_t*> sub;  // eval log -> 4 x 2^e accumulation buffer
    const int sg = shard_.active ? shard_.g : 0;
    int k0 = 0;
    for (auto& ci : inst) {
      int e = ci.log_size + 1;
      uint64_t E = 1ull << e;
      bool first = sub.find(e) == sub.end();
      if (first) sub[e] = arena_.alloc_words(4 * E);
      CompositionArgs a{};
      a.kind = ci.spec->kind;
      a.log_size = ci.log_size;
      a.eval_log = e;
      a.main = tree1.cols[ci.main_start].lde;
      a.inter = tree2.cols[ci.inter_start].lde;
      a.row0 = shard_.rank << (e - sg);
      a.n_rows = (uint32_t)(E >> sg);
      a.stride = E >> sg;
      const int last_group = 4 * (ci.spec->n_rel - 1);
      if (sg == 0) {
        a.prev_last = a.inter + (uint64_t)last_group * E;
      } else {
        // The mask offset -1 of the last logup column group reads other row blocks: under bit reversal the previous
        // row of block b lies in block rev(rev(b)+1) (odd storage indices) or rev(rev(b)-1) (even ones).  Evaluate
        // those two blocks of the group's 4 columns here as well, straight from the coefficients.
        if (ci.halo) {   // arrived with the interaction commit's all-to-all
          a.prev_last = ci.halo;
        } else {
        uint32_t* halo = arena_.alloc_words(4 * E);
        const uint32_t G = 1u << sg, rb = bit_reverse(shard_.rank, sg);
        const uint32_t nb[2] = {bit_reverse((rb + 1) & (G - 1), sg), bit_reverse((rb + G - 1) & (G - 1), sg)};
        for (int h = 0; h < (nb[0] == nb[1] ? 1 : 2); ++h) {
          StageTimer t(this, log, stream_, C_FFT);
          timings.fft_launches += launch_fft_block(halo + (uint64_t)nb[h] * (E >> sg), E,
                                                   tree2.cols[ci.inter_start + last_group].coeffs, 1ull << ci.log_size,
                                                   ci.log_size, 4, e, sg, nb[h], tw(e), stream_);
        }
        a.prev_last = halo;
        }
      }
      a.out = sub[e];
      a.accumulate = first ? 0 : 1;
      a.z = elems.z[ELEMS_NODE];
      a.alpha = elems.alpha[ELEMS_NODE];
      for (int j = 0; j < ci.spec->n_rel; ++j)
        if (ci.spec->rel_elems[j] != ELEMS_NODE) {
          a.z2 = elems.z[ci.spec->rel_elems[j]];
          a.alpha2 = elems.alpha[ci.spec->rel_elems[j]];
        }
      a.pre = ci.pre_idx[0] >= 0 ? tree0.cols[ci.pre_idx[0]].lde : nullptr;
      a.pre2 = ci.pre_idx[1] >= 0 ? tree0.cols[ci.pre_idx[1]].lde : nullptr;
      a.claimed_shift = ci.d_claimed_shift;
      const ConstraintLayout L = constraint_layout(*ci.spec, cfg.protocol_variant);
      for (int k = 0; k < L.n_kernel; ++k) {
        a.coeff[k] = L.proto_index[k] < 0 ? q_zero() : powers[n_total - 1 - (k0 + L.proto_index[k])];
        if (L.neg[k]) a.coeff[k] = q_neg(a.coeff[k]);
      }
      k0 += L.n_protocol;
      for (int b = 0; b < 2; ++b) {
        Pt p = domain_point(e, (uint32_t)b << ci.log_size);
        uint32_t x = p.x;
        for (int k = 0; k < ci.log_size - 1; ++k) x = m_sub(m_dbl(m_sqr(x)), 1u);
        a.zinv[b] = m_inv(x);
      }
      launch_composition(a, stream_);
    }
    // sharded: every rank evaluated its row block of each per-size accumulator; make them whole everywhere (the
    // one bulk exchange of the proof: 16 B per eval-domain row in total) before the interpolation
    if (shard_.active)
      for (auto& kv : sub) gather_columns(kv.second, 1ull << kv.first, 4, (1ull << kv.first) >> sg);
    // DomainEvaluationAccumulator::finalize: fold smaller sizes into larger ones
    uint32_t* cur = nullptr;  // coefficients, 4 x 2^cur_log
    CommitOut comp_out;
    int cur_log = 0;
    for (auto& kv : sub) {
      int e = kv.first;
      uint64_t E = 1ull << e;
      uint32_t* vals = kv.second;
      if (cur) {
        uint32_t* ext = arena_.alloc_words(4 * E);
        StageTimer t(this, log, stream_, C_FFT);
        timings.fft_launches += launch_fft(ext, E, cur, 1ull << cur_log, cur_log, 4, e, tw(e), stream_);
        timings.fft_bytes += 4ull * (4ull << cur_log) + 4ull * 4ull * E;
        timings.fft_butterflies += 4ull * (E / 2) * (uint64_t)e;
        launch_secure_add(vals, ext, 4 * E, stream_);
      }
      if (e == comp_log) {
        comp_out = interpolate_for_commit(vals, vals, 4, e);   // the last (largest) size: the committed polynomial
      } else {
        StageTimer t(this, log, stream_, C_FFT);
        timings.fft_launches += launch_ifft(vals, E, vals, E, 4, e, itw(e), stream_);
        timings.fft_bytes += 4ull * 8ull * E;
        timings.fft_butterflies += 4ull * (E / 2) * (uint64_t)e;
      }
      cur = vals;
      cur_log = e;
    }
    if (cur_log != comp_log) throw LmnError(LMN_ERR_INTERNAL, "composition size mismatch");
    for (int k = 0; k < 4; ++k)
      tree3.cols.push_back({comp_log, cur + ((uint64_t)k << comp_log), comp_out.lde ? comp_out.lde + (uint64_t)k * comp_out.stride : nullptr,
                            comp_out.sharded, comp_out.owner_of(k)});
  }
  {
    StageTimer st(this, log, stream_, C_COMP_COMMIT);
    lde_and_merkle(tree3);
    lmn_sync(stream_);
    tree3.merkle.finish_root();
    channel.mix_root(tree3.merkle.root);
  }
  hm.mark("sync3: root3 mixed");
  DevTree* trees[4] = {&tree0, &tree1, &tree2, &tree3};
  for (auto* t : trees) proof.commitments.push_back(t->merkle.root);

  // ---- OODS point + mask points
  QM31 tt = channel.draw_felt();
  QM31 t2 = q_sqr(tt);
  QM31 tinv = q_inv(q_add_m(t2, 1u));
  QPt oods{q_mul(q_sub(q_one(), t2), tinv), q_mul(q_add(tt, tt), tinv)};
  std::vector<QPt> points{oods};
  std::map<int, int> prev_point_of_log;
  for (auto& ci : inst) {
    if (prev_point_of_log.count(ci.log_size)) continue;
    Pt stp = pt_of_index((0x80000000u - subgroup_gen_index(ci.log_size)) & 0x7fffffffu);  // -step
    prev_point_of_log[ci.log_size] = (int)points.size();
    points.push_back(qpt_add_m(oods, stp));
  }
  // sample point indices per tree/column, in sampled_values order
  std::vector<std::vector<std::vector<int>>> spoints(4);
  spoints[0].assign(tree0.cols.size(), {0});
  spoints[1].assign(tree1.cols.size(), {0});
  spoints[2].assign(tree2.cols.size(), {0});
  spoints[3].assign(4, {0});
  for (auto& ci : inst) {
    int nic = 4 * ci.spec->n_rel;
    for (int c = nic - 4; c < nic; ++c) spoints[2][ci.inter_start + c] = {prev_point_of_log[ci.log_size], 0};
  }
  std::vector<std::vector<std::vector<QM31>>> sampled(4);
  {
    StageTimer st(this, log, stream_, C_OODS);
    std::vector<EvalJob> jobs;
    for (int t = 0; t < 4; ++t)
      for (size_t c = 0; c < trees[t]->cols.size(); ++c)
        for (int p : spoints[t][c]) jobs.push_back({trees[t]->cols[c].coeffs, trees[t]->cols[c].log_size, p, trees[t]->cols[c].owner});
    std::vector<QM31> vals = eval_at_points(jobs, points, comp_log, /*split=*/true);
    size_t k = 0;
    for (int t = 0; t < 4; ++t) {
      sampled[t].resize(trees[t]->cols.size());
      for (size_t c = 0; c < trees[t]->cols.size(); ++c)
        for (size_t p = 0; p < spoints[t][c].size(); ++p) sampled[t][c].push_back(vals[k++]);
    }
  }
  proof.sampled_values = sampled;
  hm.mark("sync4: oods values on host");
  {
    std::vector<QM31> flat;
    for (auto& t : sampled)
      for (auto& c : t)
        for (auto& v : c) flat.push_back(v);
    channel.mix_felts(flat);
  }
  // sanity check of stwo::prover::prove: composition OODS eval must match the AIR at the samples
  {
    QM31 lhs = q_from_partial_evals(sampled[3][0][0], sampled[3][1][0], sampled[3][2][0], sampled[3][3][0]);
    QM31 rhs = eval_composition_at_point(inst, sampled, oods, elems, comp_alpha, cfg.protocol_variant);
    if (!q_eq(lhs, rhs) && !LMN_ABLATED(~0u)) throw LmnError(LMN_ERR_CONSTRAINTS, "ProverError(ConstraintsNotSatisfied)");
  }

  // ---- FRI quotients, one secure column per LDE size (descending)
  const QM31 quot_alpha = channel.draw_felt();
  hm.mark("sampled mixed, oods check, alpha drawn");
  // sharding of the FRI part: a quotient column / FRI layer of more than 2^fri_T rows is split into row blocks
  // (pair folds stay inside a block: rows 2i and 2i+1 are adjacent in bit-reversed order); smaller ones are
  // all-gathered once and finished identically on every rank
  const bool sh = shard_.active;
  const int g = sh ? shard_.g : 0;
  const int fri_T = std::max(shard_.fri_min_log, (int)cfg.log_last_layer + lb);
  auto sharded_log = [&](int lg) { return sh && lg > fri_T; };
  struct FlatCol {
    const uint32_t* lde;  // all rows, or this rank's block of them (sharded proof)
    int lde_log;
    std::vector<std::pair<int, QM31>> samples;  // (point index, value)
  };
  std::vector<FlatCol> flat;
  for (int t = 0; t < 4; ++t)
    for (size_t c = 0; c < trees[t]->cols.size(); ++c) {
      FlatCol f{trees[t]->cols[c].lde, trees[t]->cols[c].log_size + lb, {}};
      for (size_t p = 0; p < spoints[t][c].size(); ++p) f.samples.push_back({spoints[t][c][p], sampled[t][c][p]});
      flat.push_back(f);
    }
  std::set<int, std::greater<int>> size_set;
  for (auto& f : flat) size_set.insert(f.lde_log);
  std::vector<int> sizes(size_set.begin(), size_set.end());
  struct Quot {
    int log;
    uint32_t* vals;  // 4 x 2^log, or 4 x 2^(log-g) (this rank's rows) when sharded
    bool sharded;
  };
  std::vector<Quot> quots;
  {
    StageTimer st(this, log, stream_, C_QUOT);
    for (int ls : sizes) {
      std::vector<const FlatCol*> cols;
      for (auto& f : flat)
        if (f.lde_log == ls) cols.push_back(&f);
      std::vector<const uint32_t*> ptrs;
      std::vector<std::vector<std::pair<int, QM31>>> smp;
      for (auto* c : cols) {
        ptrs.push_back(c->lde);
        smp.push_back(c->samples);
      }
      QuotientArgs a = make_quotient_args(ls, ptrs, smp, points, quot_alpha, !sh);
      uint32_t* vals = a.out;
      const bool qs = sharded_log(ls);
      const uint64_t L = 1ull << ls, Lb = L >> g;
      if (sh) {
        a.row0 = shard_.rank << (ls - g);
        a.log_rows = ls - g;
        if (qs) {
          vals = arena_.alloc_words(4 * Lb);
          a.out = vals;
          a.out_stride = Lb;
        } else {
          vals = arena_.alloc_words(4 * L);
          a.out = vals + a.row0;
          a.out_stride = L;
        }
      }
      // unsharded proofs with two LDE sizes: the second (smaller) size is computed on the second stream, next to the
      // leaf hashing of the first size's quotient columns; build_merkle_levels waits for it before level `ls`
      const bool overlap = have_stream2_ && !sh && sizes.size() == 2 && ls == sizes[1];
      if (overlap) {
        lmn_event_record(ev_fork_, stream_);            // everything enqueued so far (incl. the entry-table upload)
        lmn_stream_wait_event(stream2_, ev_fork_);
        launch_quotients(a, stream2_);
        lmn_event_record(ev_join_, stream2_);
        wait_before_level_ev_ = ev_join_;
        wait_before_level_ = ls;
      } else {
        launch_quotients(a, stream_);
      }
      if (sh && !qs) gather_columns(vals, L, 4, Lb);
      quots.push_back({ls, vals, qs});
    }
  }

  hm.mark("quotients enqueued");
  // ---- FRI commit
  struct FriLayer {
    int log;
    uint32_t* vals;  // 4 x 2^log (line evaluation), or this rank's 4 x 2^(log-g) rows when sharded
    bool sharded;
    DevMerkle merkle;
  };
  auto secure_cols = [&](const uint32_t* vals, int lg, bool s, std::vector<ColRef>& out) {
    const uint64_t stride = s ? (1ull << (lg - g)) : (1ull << lg);
    for (int k = 0; k < 4; ++k) out.push_back({vals + (uint64_t)k * stride, lg, s});
  };
  DevMerkle first_merkle;
  std::vector<ColRef> first_cols;
  std::vector<FriLayer> inner;
  std::vector<QM31> last_vals;
  int last_log = 0;
  {
    StageTimer st(this, log, stream_, C_FRI);
    for (auto& q : quots) secure_cols(q.vals, q.log, q.sharded, first_cols);
    // The FRI commit loop runs without host round trips: a device-resident copy of the channel
    // mixes each layer root and draws the folding alpha; the host replays the same steps afterwards.
    int ls0 = quots[0].log;
    const int last_size_log = (int)cfg.log_last_layer + lb;
    const int max_layers = ls0 + 1;
    DevChannel hc{};
    memcpy(hc.digest, channel.digest().w, 32);
    hc.n_sent = 0;
    hc.variant = (cfg.protocol_variant & LMN_PV_DRAW_CTR_U32) ? 1u : 0u;   // the only encoding the FRI loop's channel ops depend on
    DevChannel* d_ch = (DevChannel*)stage_upload(&hc, sizeof hc);
    QM31* d_alphas = (QM31*)arena_.alloc_bytes((size_t)max_layers * sizeof(QM31));
    uint32_t* d_roots = arena_.alloc_words((size_t)max_layers * 8);
    int n_roots = 0;
    build_merkle(first_merkle, first_cols, d_ch, d_alphas + n_roots, d_roots + 8 * n_roots, sharded_log(ls0));
    ++n_roots;
    // fold of a (whole or row-block) source into a (whole or row-block) destination one size smaller; a sharded
    // source folds its own pairs only: into its own block of a sharded destination, or into its rows of a whole
    // one (all-gathered by the caller afterwards).  Line domain of log L: x-coordinates of CanonicCoset(L+1)'s half coset.
    auto fold = [&](bool circle, uint32_t* dst, bool dst_s, const uint32_t* src, int src_log, bool src_s, const QM31* alpha,
                    int accumulate) {
      const uint32_t* itw = circle ? itwY_[src_log] : itwX_[src_log + 1];
      if (!src_s) {
        if (circle)
          launch_fold_circle_into_line(dst, src, 1u << src_log, itw, alpha, accumulate, stream_);
        else
          launch_fold_line(dst, src, 1u << src_log, itw, alpha, stream_);
        return;
      }
      const uint32_t src_len = 1u << (src_log - g);
      const uint32_t off = shard_.rank << (src_log - 1 - g);  // first folded row (= first twiddle) of this block
      uint32_t* d = dst_s ? dst : dst + off;
      const uint64_t dstride = dst_s ? 0 : (1ull << (src_log - 1));
      if (circle)
        launch_fold_circle_into_line(d, src, src_len, itw + off, alpha, accumulate, stream_, dstride);
      else
        launch_fold_line(d, src, src_len, itw + off, alpha, stream_, dstride);
    };
    auto layer_alloc = [&](int lg, bool s) { return arena_.alloc_words(s ? (4ull << (lg - g)) : (4ull << lg)); };
    int layer_log = ls0 - 1;
    bool lay_sh = sharded_log(layer_log);
    uint32_t* layer = layer_alloc(layer_log, lay_sh);
    // Unsharded proofs leave the fold that produces a layer PENDING, so that the layer's own leaf hashing can
    // compute it (MerkleFold): one launch and one pass over the layer less.  It is materialised by the plain fold
    // kernel instead whenever something else needs the values first (a quotient column that joins the layer, the
    // single-block FRI tail, a layer too small for the fused kernel).
    struct PendingFold {
      bool on = false, circle = false;
      const uint32_t* src = nullptr;
      int src_log = 0;
      const QM31* alpha = nullptr;
      const uint32_t* join = nullptr;   // a quotient column of the source's size that joins the layer (circle fold, accumulated)
    } pend;
    auto materialise = [&](uint32_t* dst) {
      if (!pend.on) return;
      fold(pend.circle, dst, false, pend.src, pend.src_log, false, pend.alpha, 0);
      if (pend.join) fold(true, dst, false, pend.join, pend.src_log, false, pend.alpha, 1);
      pend.on = false;
    };
    static const bool fuse_folds = getenv("LMN_NO_FOLD_FUSION") == nullptr;
    if (!sh && fuse_folds) {
      pend = {true, true, quots[0].vals, ls0, d_alphas + (n_roots - 1)};
    } else {
      fold(true, layer, lay_sh, quots[0].vals, ls0, quots[0].sharded, d_alphas + (n_roots - 1), 0);
      if (quots[0].sharded && !lay_sh) gather_columns(layer, 1ull << layer_log, 4, (1ull << layer_log) >> g);
    }
    size_t qi = 1;
    while (layer_log > last_size_log) {
      if (pend.on && layer_log <= 10) materialise(layer);
      if (!lay_sh && layer_log <= 10 && qi == quots.size()) {
        // all remaining layers fit one block: commit + fold them in a single launch
        int n_tail = layer_log - last_size_log;
        std::vector<FriTailLayer> tl(n_tail);
        for (int li = 0; li < n_tail; ++li) {
          int L = layer_log - li;
          FriLayer fl;
          fl.log = L;
          fl.vals = layer;
          fl.sharded = false;
          fl.merkle.max_log = L;
          fl.merkle.layers.assign(L + 1, nullptr);
          for (int l = 0; l <= L; ++l) fl.merkle.layers[l] = arena_.alloc_words((size_t)8 << l);
          uint32_t* next = arena_.alloc_words(4ull << (L - 1));
          tl[li].vals = layer;
          tl[li].next = next;
          tl[li].itw = itwX_[L + 1];
          for (int l = 0; l <= L; ++l) tl[li].merkle[l] = fl.merkle.layers[l];
          inner.push_back(fl);
          layer = next;
        }
        FriTailLayer* d_tl = upload_vec(tl);
        {
          StageTimer t(this, log, stream_, C_MERKLE);
          launch_fri_tail(d_ch, d_tl, n_tail, layer_log, d_alphas + n_roots, d_roots + 8 * n_roots, stream_);
        }
        for (int li = 0; li < n_tail; ++li) timings.merkle_compressions += (2ull << (layer_log - li));
        n_roots += n_tail;
        layer_log = last_size_log;
        break;
      }
      FriLayer fl;
      fl.log = layer_log;
      fl.vals = layer;
      fl.sharded = lay_sh;
      std::vector<ColRef> lc;
      secure_cols(layer, layer_log, lay_sh, lc);
      if (pend.on) {
        MerkleFold mf{pend.src, pend.circle ? itwY_[pend.src_log] : itwX_[pend.src_log + 1], pend.alpha, layer};
        if (pend.join) {
          mf.src2 = pend.join;
          mf.itw2 = itwY_[pend.src_log];
        }
        build_merkle(fl.merkle, lc, d_ch, d_alphas + n_roots, d_roots + 8 * n_roots, false, &mf);
        pend.on = false;
      } else {
        build_merkle(fl.merkle, lc, d_ch, d_alphas + n_roots, d_roots + 8 * n_roots, lay_sh);
      }
      ++n_roots;
      const QM31* d_alpha = d_alphas + (n_roots - 1);
      const int next_log = layer_log - 1;
      const bool next_sh = sharded_log(next_log);
      uint32_t* next = layer_alloc(next_log, next_sh);
      const bool joins = qi < quots.size() && quots[qi].log - 1 == next_log;
      const bool fuse_joins = getenv("LMN_NO_JOIN_FUSION") == nullptr;   // (read per proof: the tests toggle it)
      if (!sh && fuse_folds && next_log > 10 && (!joins || (fuse_joins && !quots[qi].sharded))) {
        pend = {true, false, layer, layer_log, d_alpha, joins ? quots[qi].vals : nullptr};
        if (joins) ++qi;   // (quotient sizes are distinct: at most one column joins a layer)
      } else {
        fold(false, next, next_sh, layer, layer_log, lay_sh, d_alpha, 0);
      }
      inner.push_back(fl);
      while (qi < quots.size() && quots[qi].log - 1 == next_log) {
        fold(true, next, next_sh, quots[qi].vals, quots[qi].log, quots[qi].sharded, d_alpha, 1);
        ++qi;
      }
      // a quotient column of this size is sharded exactly when the layer is, so one test covers both sources
      if (lay_sh && !next_sh) gather_columns(next, 1ull << next_log, 4, (1ull << next_log) >> g);
      layer = next;
      layer_log = next_log;
      lay_sh = next_sh;
    }
    materialise(layer);  // a last layer larger than the fused threshold (log_last_layer > 9) is still pending
    hm.mark("fri enqueued");
    // one sync: roots + alphas back, then replay the transcript on the host channel
    const uint32_t* h_roots = (const uint32_t*)stage_download(d_roots, (size_t)n_roots * 32);
    const QM31* h_alphas = (const QM31*)stage_download(d_alphas, (size_t)n_roots * sizeof(QM31));
    if (qi != quots.size()) throw LmnError(LMN_ERR_INTERNAL, "FRI: unconsumed columns");
    last_log = layer_log;
    const uint32_t* raw = (const uint32_t*)stage_download(layer, (size_t)16 << last_log);
    lmn_sync(stream_);
    {
      uint32_t n = 1u << last_log;
      for (uint32_t i = 0; i < n; ++i) last_vals.push_back({raw[i], raw[n + i], raw[2 * n + i], raw[3 * n + i]});
    }
    for (int r = 0; r < n_roots; ++r) {
      Hash32 root;
      memcpy(root.w, &h_roots[(size_t)r * 8], 32);
      if (r == 0)
        first_merkle.root = root;
      else
        inner[r - 1].merkle.root = root;
      channel.mix_root(root);
      QM31 a = channel.draw_felt();
      if (!q_eq(a, h_alphas[r])) throw LmnError(LMN_ERR_INTERNAL, "device/host transcript divergence in FRI");
    }
  }
  hm.mark("fri synced+replayed");
  // last layer: interpolate the line evaluation (bit-reversed over LineDomain(half_odds(last_log)))
  {
    std::vector<std::vector<QM31>> chunks{last_vals};
    int dlog = last_log;
    // x-coordinates of the current line domain in bit-reversed order
    auto line_xs = [&](int lg) {
      std::vector<uint32_t> xs(1u << lg);
      uint32_t init = 1u << (31 - (lg + 2)), step = lg >= 1 ? (1u << (31 - lg)) : 0u;
      for (uint32_t i = 0; i < (1u << lg); ++i) xs[i] = pt_of_index(init + bit_reverse(i, lg) * step).x;
      return xs;
    };
    while (dlog > 0) {
      std::vector<uint32_t> xs = line_xs(dlog);
      std::vector<std::vector<QM31>> nc;
      for (auto& ch : chunks) {
        std::vector<QM31> f0, f1;
        for (size_t i = 0; i < ch.size() / 2; ++i) {
          QM31 a = ch[2 * i], b = ch[2 * i + 1];
          f0.push_back(q_add(a, b));
          f1.push_back(q_mul_m(q_sub(a, b), m_inv(xs[2 * i])));
        }
        nc.push_back(f0);
        nc.push_back(f1);
      }
      chunks.swap(nc);
      // after halving, the remaining domain is the doubled line domain
      dlog -= 1;
    }
    uint32_t n = 1u << last_log;
    uint32_t ninv = m_inv(n % P31);
    std::vector<QM31> coeffs(n);
    for (uint32_t idx = 0; idx < n; ++idx) {
      uint32_t j = 0;
      for (int k = 0; k < last_log; ++k) j |= ((idx >> (last_log - 1 - k)) & 1u) << k;
      coeffs[j] = q_mul_m(chunks[idx][0], ninv);
    }
    uint32_t bound = 1u << cfg.log_last_layer;
    for (uint32_t j = bound; j < n; ++j)
      if (!q_is_zero(coeffs[j]) && !LMN_ABLATED(~0u)) throw LmnError(LMN_ERR_INTERNAL, "FRI: invalid last-layer degree");
    coeffs.resize(bound);
    proof.last_layer_coeffs = coeffs;
    proof.last_layer_log_size = cfg.log_last_layer;
    channel.mix_felts(coeffs);
  }

  hm.mark("last layer");
  // ---- proof of work + queries
  proof.proof_of_work = channel.grind(cfg.pow_bits);
  channel.mix_u64(proof.proof_of_work);
  const int ls0 = quots[0].log;
  std::vector<uint32_t> queries;
  {
    std::set<uint32_t> qs;
    uint64_t cnt = 0;
    const uint32_t mask = (1u << ls0) - 1u;
    while (cnt < cfg.n_queries) {
      Hash32 r = channel.draw_random_words();
      for (int i = 0; i < 8 && cnt < cfg.n_queries; ++i, ++cnt) qs.insert(r.w[i] & mask);
    }
    queries.assign(qs.begin(), qs.end());
  }
  std::map<int, std::vector<uint32_t>> pos_by_log;
  for (int ls : sizes) pos_by_log[ls] = fold_positions(queries, ls0 - ls);

  hm.mark("pow+queries");
  // ---- decommitment: plan device references, gather once, distribute
  {
    StageTimer st(this, log, stream_, C_DECOMMIT);
    typedef DecommitPlan Plan;
    if (!host_scratch) host_scratch = new HostScratch();
    HostScratch& hs = *static_cast<HostScratch*>(host_scratch);
    hs.used = 0;
    hs.jobs.clear();
    hs.plans.reserve(inner.size() + 5);  // plans are handed out by reference: no reallocation while planning
    std::vector<Plan>& plans = hs.plans;  // [first, inner..., tree0..3]
    {
      Plan& p = hs.next();
      std::map<int, std::vector<uint32_t>> dec;
      for (size_t qk = 0; qk < quots.size(); ++qk) {
        const ColRef(&c4)[4] = *reinterpret_cast<const ColRef(*)[4]>(&first_cols[4 * qk]);
        plan_fri_witness(c4, g, pos_by_log[quots[qk].log], dec[quots[qk].log], p.fri_wit);
      }
      std::vector<Ref> dummy;
      plan_merkle_decommit(first_merkle, first_cols, g, dec, dummy, p.hash_wit, p.col_wit, hs.jobs);
    }
    std::vector<uint32_t> lq = fold_positions(queries, 1);
    for (auto& fl : inner) {
      Plan& p = hs.next();
      std::map<int, std::vector<uint32_t>> dec;
      std::vector<ColRef>& lc = hs.cols;
      lc.clear();
      secure_cols(fl.vals, fl.log, fl.sharded, lc);
      const ColRef(&c4)[4] = *reinterpret_cast<const ColRef(*)[4]>(lc.data());
      plan_fri_witness(c4, g, lq, dec[fl.log], p.fri_wit);
      std::vector<Ref> dummy;
      plan_merkle_decommit(fl.merkle, lc, g, dec, dummy, p.hash_wit, p.col_wit, hs.jobs);
      lq = fold_positions(lq, 1);
    }
    for (auto* t : trees) {
      Plan& p = hs.next();
      std::vector<ColRef>& sorted = hs.cols;
      sorted.clear();
      std::map<int, std::vector<uint32_t>> qmap;
      sorted.reserve(t->cols.size());
      for (auto& c : t->cols) {
        sorted.push_back({c.lde, c.log_size + lb, c.sharded});
        if (!qmap.count(c.log_size + lb)) qmap[c.log_size + lb] = pos_by_log[c.log_size + lb];
      }
      std::stable_sort(sorted.begin(), sorted.end(), [](auto& a, auto& b) { return a.log > b.log; });
      plan_merkle_decommit(t->merkle, sorted, g, qmap, p.queried, p.hash_wit, p.col_wit, hs.jobs);
    }
    // Every rank plans the same list; it fetches the runs it holds into its own slot of the output buffer, the
    // slots are all-gathered (a few KB per rank) and each run is then read from its owner's slot.
    std::vector<GatherEntry>& entries = hs.entries;
    entries.clear();
    std::vector<std::pair<int, uint32_t>>& runs = hs.runs;  // (owner, len) in output order
    runs.clear();
    uint32_t out_words = 0;
    auto add_refs = [&](const std::vector<Ref>& refs) {
      for (auto& r : refs) {
        if (r.job >= 0)
          hs.jobs[r.job].dst_off = out_words;   // unsharded proofs only: one output slot
        else if (r.owner < 0 || r.owner == (int)shard_.rank)
          entries.push_back({arena_.word_offset(r.ptr), r.len, out_words});
        if (sh) runs.push_back({r.owner, r.len});
        out_words += r.len;
      }
    };
    for (size_t k = 0; k < hs.used; ++k) {
      Plan& p = plans[k];
      add_refs(p.fri_wit);
      add_refs(p.queried);
      add_refs(p.hash_wit);
      add_refs(p.col_wit);
    }
    const uint32_t* gathered = nullptr;
    std::vector<uint32_t> merged;
    if (out_words) {
      const uint32_t slots = sh ? shard_.world : 1u, mine = sh ? shard_.rank : 0u;
      for (auto& e : entries) e.dst_off += mine * out_words;
      // the entry table is read once, one entry per lane: the kernel takes it straight from pinned host memory
      GatherEntry* d_e = (GatherEntry*)pin_alloc((entries.size() + 1) * sizeof(GatherEntry));
      memcpy(d_e, entries.data(), entries.size() * sizeof(GatherEntry));
      if (!hs.jobs.empty() && sh) throw LmnError(LMN_ERR_INTERNAL, "sharded proofs keep whole trees");
      MerkleRecompute* d_j = (MerkleRecompute*)pin_alloc((hs.jobs.size() + 1) * sizeof(MerkleRecompute));
      memcpy(d_j, hs.jobs.data(), hs.jobs.size() * sizeof(MerkleRecompute));
      uint32_t* d_o = arena_.alloc_words((size_t)slots * out_words);
      hm.mark("decommit planned");
      if (hm.on) fprintf(stderr, "[host] decommit: %zu runs gathered, %zu tree nodes recomputed\n", entries.size(), hs.jobs.size());
      launch_gather(arena_.base_words(), d_e, (uint32_t)entries.size(), d_j, (uint32_t)hs.jobs.size(), d_o, stream_);
      if (sh) gather_columns(d_o, 0, 1, out_words);
      gathered = (const uint32_t*)stage_download(d_o, (size_t)slots * out_words * 4);
      lmn_sync(stream_);
      if (sh) {
        merged.resize(out_words);
        uint32_t at = 0;
        for (auto& r : runs) {
          const uint32_t slot = r.first < 0 ? mine : (uint32_t)r.first;
          memcpy(&merged[at], gathered + (size_t)slot * out_words + at, (size_t)r.second * 4);
          at += r.second;
        }
        gathered = merged.data();
      }
    }
    size_t g = 0;
    auto take_q = [&](size_t nrefs) {
      std::vector<QM31> v;
      v.reserve(nrefs / 4);
      for (size_t i = 0; i < nrefs / 4; ++i) {
        v.push_back({gathered[g], gathered[g + 1], gathered[g + 2], gathered[g + 3]});
        g += 4;
      }
      return v;
    };
    auto take_u32 = [&](size_t n) {
      std::vector<uint32_t> v(gathered + g, gathered + g + n);
      g += n;
      return v;
    };
    auto take_hashes = [&](size_t n) {
      std::vector<Hash32> v(n);
      for (size_t i = 0; i < n; ++i) {
        memcpy(v[i].w, &gathered[g], 32);
        g += 8;
      }
      return v;
    };
    size_t pi = 0;
    auto fill_layer = [&](FriLayerProof& lp, const Hash32& root) {
      Plan& p = plans[pi++];
      lp.fri_witness = take_q(p.fri_wit.size());
      take_u32(p.queried.size());
      lp.decommitment.hash_witness = take_hashes(p.hash_wit.size());
      lp.decommitment.column_witness = take_u32(p.col_wit.size());
      lp.commitment = root;
    };
    fill_layer(proof.first_layer, first_merkle.root);
    proof.inner_layers.resize(inner.size());
    for (size_t i = 0; i < inner.size(); ++i) fill_layer(proof.inner_layers[i], inner[i].merkle.root);
    for (int t = 0; t < 4; ++t) {
      Plan& p = plans[pi++];
      take_q(p.fri_wit.size());
      proof.queried_values.push_back(take_u32(p.queried.size()));
      Decommitment d;
      d.hash_witness = take_hashes(p.hash_wit.size());
      d.column_witness = take_u32(p.col_wit.size());
      proof.decommitments.push_back(d);
    }
  }
  hm.mark("decommit done");
  total_guard.reset();
  lmn_sync(stream_);

  // ---- timings
  float acc[C_N] = {0};
  for (auto& sp : log->spans) acc[sp.cat] += lmn_event_elapsed_ms(sp.a, sp.b);
  timings.total_ms = acc[C_TOTAL];
  timings.transpose_ms = acc[C_TRANSPOSE];
  timings.main_commit_ms = acc[C_MAIN_COMMIT];
  timings.logup_ms = acc[C_LOGUP];
  timings.interaction_commit_ms = acc[C_INTER_COMMIT];
  timings.composition_ms = acc[C_COMPOSITION];
  timings.composition_commit_ms = acc[C_COMP_COMMIT];
  timings.oods_ms = acc[C_OODS];
  timings.quotients_ms = acc[C_QUOT];
  timings.fri_ms = acc[C_FRI];
  timings.decommit_ms = acc[C_DECOMMIT];
  timings.fft_ms = acc[C_FFT];
  timings.merkle_ms = acc[C_MERKLE];
  timings.merkle_fused_ms = acc[C_MERKLE_FUSED];
  return proof_to_bincode(proof);
}

// ------------------------------------------------------------------------------------ single-proof sharding
}  // namespace lmn
#include <dlfcn.h>
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
static void rccl_release(void* p) {
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

// ------------------------------------------------------------------------------------ level-2 ops
constexpr uint32_t OP_MAX_LOG = 26;  // largest column a level-2 op accepts (as lmn_prove: 2^26 rows)
static void check_op_log(uint32_t log_size, const char* what) {
  if (log_size > OP_MAX_LOG) throw LmnError(LMN_ERR_INVALID_ARGUMENT, std::string(what) + ": log size above 26");
}
void Context::set_device() {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
}
void Context::reset_event_log() { g_log(this)->reset(); }
// every op starts from an empty device arena AND an empty pinned staging buffer
void Context::begin_op() {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  arena_.reset();
  pin_off_ = 0;
}

void Context::op_interpolate(uint32_t* cols, uint32_t ncols, uint32_t log_size) {
  check_op_log(log_size, "interpolate");
  ensure_twiddles((int)log_size);
  size_t bytes = ((size_t)ncols << log_size) * 4;
  arena_.reserve(bytes + (1u << 20));
  begin_op();
  uint32_t* d = arena_.alloc_words((size_t)ncols << log_size);
  lmn_h2d(d, cols, bytes, stream_);
  launch_ifft(d, 1ull << log_size, d, 1ull << log_size, (int)ncols, (int)log_size, itw((int)log_size), stream_);
  lmn_d2h(cols, d, bytes, stream_);
  lmn_sync(stream_);
}

void Context::op_evaluate(const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                          uint32_t* out) {
  if (log_coeffs > log_domain) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "log_coeffs > log_domain");
  check_op_log(log_domain, "evaluate");
  ensure_twiddles((int)log_domain);
  size_t in_w = (size_t)ncols << log_coeffs, out_w = (size_t)ncols << log_domain;
  arena_.reserve((in_w + out_w) * 4 + (1u << 20));
  begin_op();
  uint32_t* d_in = arena_.alloc_words(in_w);
  uint32_t* d_out = arena_.alloc_words(out_w);
  lmn_h2d(d_in, coeffs, in_w * 4, stream_);
  launch_fft(d_out, 1ull << log_domain, d_in, 1ull << log_coeffs, (int)log_coeffs, (int)ncols, (int)log_domain,
             tw((int)log_domain), stream_);
  lmn_d2h(out, d_out, out_w * 4, stream_);
  lmn_sync(stream_);
}

void Context::op_evaluate_block(const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                                uint32_t log_blocks, uint32_t block, uint32_t* out) {
  if (log_coeffs > log_domain) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "log_coeffs > log_domain");
  if (log_blocks < 1 || log_blocks > 3 || log_blocks >= log_domain || block >= (1u << log_blocks))
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad block specification");
  check_op_log(log_domain, "evaluate_block");
  ensure_twiddles((int)log_domain);
  const uint32_t lb = log_domain - log_blocks;
  size_t in_w = (size_t)ncols << log_coeffs, out_w = (size_t)ncols << lb;
  arena_.reserve((in_w + out_w) * 4 + (1u << 20));
  begin_op();
  uint32_t* d_in = arena_.alloc_words(in_w);
  uint32_t* d_out = arena_.alloc_words(out_w);
  lmn_h2d(d_in, coeffs, in_w * 4, stream_);
  launch_fft_block(d_out, 1ull << lb, d_in, 1ull << log_coeffs, (int)log_coeffs, (int)ncols, (int)log_domain,
                   (int)log_blocks, block, tw((int)log_domain), stream_);
  lmn_d2h(out, d_out, out_w * 4, stream_);
  lmn_sync(stream_);
}

void Context::op_merkle_root(const uint32_t* const* cols, const uint32_t* log_sizes, uint32_t ncols, uint8_t root[32]) {
  size_t words = 0;
  uint32_t max_log = 0;
  for (uint32_t c = 0; c < ncols; ++c) {
    check_op_log(log_sizes[c], "merkle_root");
    words += 1ull << log_sizes[c];
    max_log = std::max(max_log, log_sizes[c]);
  }
  arena_.reserve((words + (16ull << max_log)) * 4 + (1u << 20));
  begin_op();
  std::vector<ColRef> sorted;
  for (uint32_t c = 0; c < ncols; ++c) {
    uint32_t* d = arena_.alloc_words(1ull << log_sizes[c]);
    lmn_h2d(d, cols[c], (4ull << log_sizes[c]), stream_);
    sorted.push_back({d, (int)log_sizes[c], false});
  }
  std::stable_sort(sorted.begin(), sorted.end(), [](auto& a, auto& b) { return a.log > b.log; });
  g_log(this)->reset();
  DevMerkle m;
  build_merkle(m, sorted);
  fetch_root_async(m);
  lmn_sync(stream_);
  m.finish_root();
  memcpy(root, m.root.w, 32);
}

void Context::op_eval_at_point(const uint32_t* coeffs, uint32_t log_size, const uint32_t pt[8], uint32_t out[4]) {
  check_op_log(log_size, "eval_at_point");
  arena_.reserve((4ull << log_size) + (8u << 20));
  begin_op();
  uint32_t* d = arena_.alloc_words(1ull << log_size);
  lmn_h2d(d, coeffs, 4ull << log_size, stream_);
  QPt p{{pt[0], pt[1], pt[2], pt[3]}, {pt[4], pt[5], pt[6], pt[7]}};
  std::vector<QM31> r = eval_at_points({{d, (int)log_size, 0}}, {p}, (int)log_size);
  out[0] = r[0].a;
  out[1] = r[0].b;
  out[2] = r[0].c;
  out[3] = r[0].d;
}

// QuotientOps::accumulate_quotients for the columns of one LDE size
void Context::op_accumulate_quotients(uint32_t log_size, const uint32_t* const* cols, uint32_t ncols,
                                      const uint32_t* sample_col, const uint32_t* sample_point, const uint32_t* sample_values,
                                      uint32_t nsamples, const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4],
                                      uint32_t* out) {
  if (log_size < 2 || log_size > 26) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad log_size");
  ensure_twiddles((int)log_size);
  const uint64_t L = 1ull << log_size;
  arena_.reserve(((uint64_t)ncols + 4) * L * 4 + (8u << 20));
  begin_op();
  std::vector<const uint32_t*> d_cols(ncols);
  for (uint32_t c = 0; c < ncols; ++c) {
    uint32_t* d = arena_.alloc_words(L);
    lmn_h2d(d, cols[c], L * 4, stream_);
    d_cols[c] = d;
  }
  std::vector<QPt> pts(npoints);
  for (uint32_t p = 0; p < npoints; ++p) {
    const uint32_t* w = points_xy + 8 * p;
    pts[p] = {{w[0], w[1], w[2], w[3]}, {w[4], w[5], w[6], w[7]}};
  }
  std::vector<std::vector<std::pair<int, QM31>>> smp(ncols);
  for (uint32_t i = 0; i < nsamples; ++i) {
    if (sample_col[i] >= ncols || sample_point[i] >= npoints) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad sample index");
    const uint32_t* v = sample_values + 4 * i;
    smp[sample_col[i]].push_back({(int)sample_point[i], QM31{v[0], v[1], v[2], v[3]}});
  }
  QuotientArgs a = make_quotient_args((int)log_size, d_cols, smp, pts, QM31{alpha[0], alpha[1], alpha[2], alpha[3]});
  launch_quotients(a, stream_);
  lmn_d2h(out, a.out, 16 * L, stream_);
  lmn_sync(stream_);
}

// FriOps::fold_line (circle == 0) / FriOps::fold_circle_into_line (circle == 1; dst is accumulated:
// dst = dst * alpha^2 + fold(src), as the FRI commit loop does)
void Context::op_fold(int circle, uint32_t* dst, const uint32_t* src, uint32_t log_src, const uint32_t alpha[4]) {
  if (log_src < 1 || log_src > OP_MAX_LOG) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad log size");
  ensure_twiddles((int)log_src + 1);
  const uint64_t L = 1ull << log_src;
  arena_.reserve(6 * L * 4 + (1u << 20));
  begin_op();
  uint32_t* d_src = arena_.alloc_words(4 * L);
  uint32_t* d_dst = arena_.alloc_words(2 * L);
  lmn_h2d(d_src, src, 16 * L, stream_);
  std::vector<QM31> av{QM31{alpha[0], alpha[1], alpha[2], alpha[3]}};
  QM31* d_alpha = upload_vec(av);
  if (circle) {
    lmn_h2d(d_dst, dst, 8 * L, stream_);
    launch_fold_circle_into_line(d_dst, d_src, (uint32_t)L, itwY_[log_src], d_alpha, 1, stream_);
  } else {
    launch_fold_line(d_dst, d_src, (uint32_t)L, itwX_[log_src + 1], d_alpha, stream_);
  }
  lmn_d2h(dst, d_dst, 8 * L, stream_);
  lmn_sync(stream_);
}

// tiled FFT vs one-layer-per-launch kernels on pseudo-random data (device-side differential check)
void Context::op_fft_selftest(uint32_t log_size, uint32_t ncols) {
  check_op_log(log_size, "fft_selftest");
  ensure_twiddles((int)log_size + 1);
  size_t w = (size_t)ncols << log_size;
  arena_.reserve(w * 32 + (1u << 20));
  begin_op();
  std::vector<uint32_t> h(w);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    v = (uint32_t)(st >> 33) % P31;
  }
  uint32_t* a = arena_.alloc_words(w);
  uint32_t* b = arena_.alloc_words(w);
  uint64_t n = 1ull << log_size;
  for (int inverse = 0; inverse < 2; ++inverse) {
    lmn_h2d(a, h.data(), w * 4, stream_);
    lmn_h2d(b, h.data(), w * 4, stream_);
    if (inverse) {
      launch_ifft(a, n, a, n, (int)ncols, (int)log_size, itw((int)log_size), stream_);
      launch_fft_simple(b, n, (int)ncols, (int)log_size, itw((int)log_size), true, stream_);
    } else {
      launch_fft(a, n, a, n, (int)log_size, (int)ncols, (int)log_size, tw((int)log_size), stream_);
      launch_fft_simple(b, n, (int)ncols, (int)log_size, tw((int)log_size), false, stream_);
    }
    std::vector<uint32_t> ra(w), rb(w);
    lmn_d2h(ra.data(), a, w * 4, stream_);
    lmn_d2h(rb.data(), b, w * 4, stream_);
    lmn_sync(stream_);
    for (size_t i = 0; i < w; ++i)
      if (ra[i] != rb[i])
        throw LmnError(LMN_ERR_INTERNAL, std::string("fft selftest mismatch (inverse=") + std::to_string(inverse) +
                                             ") at word " + std::to_string(i));
  }
  // the fused interpolate + extend path of the commitments against the separate transforms
  if (fft_interp_extend_supported((int)log_size) && (int)log_size + 1 <= tw_max_log_) {
    begin_op();
    uint32_t* ev = arena_.alloc_words(w);
    uint32_t* co = arena_.alloc_words(w);
    uint32_t* lde = arena_.alloc_words(2 * w);
    uint32_t* co2 = arena_.alloc_words(w);
    uint32_t* lde2 = arena_.alloc_words(2 * w);
    lmn_h2d(ev, h.data(), w * 4, stream_);
    launch_interp_extend(co, n, ev, n, lde, 2 * n, (int)ncols, (int)log_size, itw((int)log_size), tw((int)log_size + 1), stream_);
    lmn_d2d(co2, ev, w * 4, stream_);
    launch_fft_simple(co2, n, (int)ncols, (int)log_size, itw((int)log_size), true, stream_);
    launch_extend(co2, n, (int)log_size, lde2, 2 * n, (int)log_size + 1, (int)ncols, stream_);
    launch_fft_simple(lde2, 2 * n, (int)ncols, (int)log_size + 1, tw((int)log_size + 1), false, stream_);
    std::vector<uint32_t> ra(3 * w), rb(3 * w);
    lmn_d2h(ra.data(), co, w * 4, stream_);
    lmn_d2h(ra.data() + w, lde, 2 * w * 4, stream_);
    lmn_d2h(rb.data(), co2, w * 4, stream_);
    lmn_d2h(rb.data() + w, lde2, 2 * w * 4, stream_);
    lmn_sync(stream_);
    for (size_t i = 0; i < 3 * w; ++i)
      if (ra[i] != rb[i])
        throw LmnError(LMN_ERR_INTERNAL, std::string("fft selftest mismatch (interpolate + extend) at word ") + std::to_string(i));
  }
}

}  // namespace lmn
