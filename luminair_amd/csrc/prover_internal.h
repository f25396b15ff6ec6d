// Internals shared by the translation units of the prover (context / commit / prove phases / decommitment / sharding /
// level-2 ops): event-based stage timing, decommitment planning storage, host-side wall-clock marks.  Not part of any
// boundary; prover.h is what capi.cpp / level2.cpp / verifier.cpp see.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>

#include "prover.h"

namespace lmn {

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
inline EventLog* g_log(Context* c) { return static_cast<EventLog*>(c->event_log); }


// ------------------------------------------------------------------------------------ decommit planning (decommit.cpp)
// A run of device words to fetch.  owner < 0: every rank holds it; otherwise only rank `owner` does (row-block
// sharded column or Merkle layer) and ptr is meaningful on that rank alone.
struct Ref {
  const uint32_t* ptr;
  uint32_t len;
  int owner;
  int job = -1;   // >= 0: not in memory - Merkle node to recompute (index into the plan's MerkleRecompute list), ptr is null
};
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

void release_host_scratch(void* p);
void plan_merkle_decommit(const DevMerkle& m, const std::vector<ColRef>& cols_sorted, int g,
                          const std::map<int, std::vector<uint32_t>>& queries, std::vector<Ref>& queried,
                          std::vector<Ref>& hash_wit, std::vector<Ref>& col_wit, std::vector<MerkleRecompute>& jobs);
std::vector<uint32_t> fold_positions(const std::vector<uint32_t>& p, int n);
void plan_fri_witness(const ColRef (&cols)[4], int g, const std::vector<uint32_t>& qpos,
                             std::vector<uint32_t>& dec_pos, std::vector<Ref>& wit);

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


// proofs between entry and exit of Context::prove in this process (context.cpp)
extern std::atomic<int> g_proofs_in_flight;

// single-proof sharding transport (shard.cpp)
void rccl_release(void* transport);
void rccl_unique_id(uint8_t* out);

}  // namespace lmn
