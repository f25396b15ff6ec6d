// Prover context: device arena, page-locked staging, uploads / downloads, and the twiddle tables
// (/root/reference/crates/prover/src/prover.rs:38-42 recomputes them per proof; here one set
// per device, built on the device and shared by the contexts of a process).
#include "prover_internal.h"

namespace lmn {

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

// ------------------------------------------------------------------------------------ context

Context::Context(int device, const lmn_config& c) : cfg(c), device_(device) {
  // validate before acquiring anything: a throwing constructor does not run the destructor
  if (cfg.log_blowup < 1 || cfg.log_blowup > 3) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "log_blowup must be 1, 2 or 3");
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
    // LMN_CU_SPLIT=P[:1] (experiment of round 6, docs/SWITCHES.md): context k's stream runs on partition k mod P of the CUs
    // only (P = 2, 4, 8; contiguous mask bits, or ":1" = bit i belongs to partition (i mod 8) * P / 8) - do launches of
    // different proofs overlap better side by side on parts of the chip than queued behind each other on all of it?
    static const char* split_env = getenv("LMN_CU_SPLIT");
    const int parts = split_env ? atoi(split_env) : 0;
    if (parts == 2 || parts == 4 || parts == 8) {
      static std::atomic<int> split_counter{0};
      const int part = split_counter.fetch_add(1) % parts;
      const bool by_xcd = strstr(split_env, ":1") != nullptr;
      hipDeviceProp_t prop;
      LMN_HIP_CHECK(hipGetDeviceProperties(&prop, device));
      const int n_cu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
      for (int i = 0; i < n_cu; ++i) {
        const int owner = by_xcd ? (i % 8) * parts / 8 : (int)((int64_t)i * parts / n_cu);
        if (owner == part) mask[(size_t)i / 32] |= 1u << (i % 32);
      }
      LMN_HIP_CHECK(hipExtStreamCreateWithCUMask(&stream_, (uint32_t)mask.size(), mask.data()));
    } else
    if (cycle && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
      const int span = lo - hi + 1;                     // lo = least priority (numerically greatest)
      const int prio = hi + (counter.fetch_add(1) % span);
      LMN_HIP_CHECK(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, prio));
    } else {
      LMN_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    }
  }
#ifndef LMN_BATCH
  // LMN_FRI_OVERLAP (experiment; default off): 1 = always, 2 = while this is the process's only proof in flight
  fri_overlap_mode_ = getenv("LMN_FRI_OVERLAP") ? std::max(0, std::min(2, atoi(getenv("LMN_FRI_OVERLAP")))) : 0;
#endif
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

std::atomic<int> g_proofs_in_flight{0};

// Experiment switch: the smaller FRI quotient column on a second stream next to the leaf hashing of the larger one's.  It
// helped a solo proof in round 3 (-50 us) and cost 7 % of throughput; with round 4's fused first layer it costs a solo
// proof 60 - 100 us as well (round 5, gpu_session_r8k: 2.37 - 2.46 vs 2.33 - 2.37 ms), also in the "only while no other proof
// of the process is in flight" form (mode 2).  Off by default; the stream is created on first use.
bool Context::second_stream_wanted() {
#if defined(LMN_EMU) || defined(LMN_BATCH)
  return false;
#else
  if (fri_overlap_mode_ == 0 || (fri_overlap_mode_ == 2 && g_proofs_in_flight.load(std::memory_order_relaxed) != 1)) return false;
  if (!have_stream2_) {
    LMN_HIP_CHECK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
    ev_fork_ = lmn_event_create_sync();
    ev_join_ = lmn_event_create_sync();
    have_stream2_ = true;
  }
  return true;
#endif
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
void Context::stage_group_begin(size_t reserve_bytes) {
  if (grp_pin_) throw LmnError(LMN_ERR_INTERNAL, "upload group already open");
  grp_pin_ = (char*)pin_alloc(reserve_bytes);
  grp_dev_ = (char*)arena_.alloc_bytes(reserve_bytes);
  grp_cap_ = reserve_bytes;
  grp_off_ = 0;
}
void Context::stage_group_end() {
  if (!grp_pin_) return;
  if (grp_off_) lmn_h2d(grp_dev_, grp_pin_, grp_off_, stream_);
  grp_pin_ = grp_dev_ = nullptr;
  grp_cap_ = grp_off_ = 0;
}
void* Context::stage_upload(const void* host, size_t bytes) {
  if (grp_pin_ && bytes) {
    const size_t at = (grp_off_ + 63) & ~(size_t)63;
    if (at + bytes <= grp_cap_) {
      memcpy(grp_pin_ + at, host, bytes);
      grp_off_ = at + bytes;
      return grp_dev_ + at;
    }
  }
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

// Twiddle tables for every canonic domain up to 2^max_domain_log (computed
// once per context and cached across proofs, instead of once per proof as prover.rs:38-42 does).
//   Y[m][h] = y(half_coset_m.at(bitrev(h, m-1))), h < 2^(m-1)      (layer 0 of domain m)
//   X[k][h] = x(half_coset_k.at(bitrev(h, k-2))), h < 2^(k-2)      (layer 1 of domain k)
// Layer i >= 1 of domain m is X[m-i+1] (doubling a canonic half coset gives the next smaller one).
#if !defined(LMN_BATCH)
// Twiddle tables are a function of the domain size alone: one set per device, built on the device (k_twiddles) and shared by
// every context of the process.  The registry holds weak references - the last context that goes away frees the tables -
// and a context that needs a larger domain than the current set builds a new one (the contexts still using the old one
// keep it alive).  The batch library's members run in lock-step (no member may skip launches another one makes): it keeps
// one host-built set per context (below).  The emulation build shares the registry code, so that the thread sanitizer
// run of the CPU suite sees contexts of several host threads racing for it (tests/test_sanitizers.py).
namespace {
struct TwiddleSet {
  int device = 0, max_log = 0;
  void* slab = nullptr;
  std::vector<uint32_t*> Y, X, iY, iX, Y2, X2, iY2, iX2;
  ~TwiddleSet() {
    if (slab) {
#ifdef LMN_EMU
      lmn_dev_free(slab);
#else
      (void)hipSetDevice(device);
      (void)hipFree(slab);
#endif
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
  // the tables are rebuilt from here on: a build that throws midway (allocation, upload) must not leave the old size in
  // place, or a later smaller proof would take the early return above and launch with freed / null table pointers
  tw_max_log_ = 0;
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

}  // namespace lmn
