// prove(): MI355X replacement for /root/reference/crates/prover/src/prover.rs:28-319.
// The Fiat-Shamir channel, proof assembly and (tiny) decommitment bookkeeping run on the host;
// every per-row / per-coefficient pass runs in the gfx950 kernels on trace data
// that stays resident in HBM from the first transpose to the last query gather.
#include "prove_run.h"

#include <exception>

namespace lmn {

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
// The phases of one proof, in transcript order; each is a Context member in the file named.
std::vector<uint8_t> Context::prove(const lmn_table* tables, size_t n_tables, const lmn_settings* settings) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (!tables || n_tables == 0) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "no trace tables");
  struct InFlight {
    InFlight() { g_proofs_in_flight.fetch_add(1, std::memory_order_relaxed); }
    ~InFlight() { g_proofs_in_flight.fetch_sub(1, std::memory_order_relaxed); }
  } in_flight;
  // A proof that ends in an exception may leave launches in flight that read plans from, or write results to, the
  // page-locked staging memory the next proof reuses from its start: drain the stream on that way out (not in a
  // lock-step batch, whose waits are rendezvous of all members), and close an upload group left open.
  struct FailureDrain {
    Context* c;
    int pending = std::uncaught_exceptions();
    ~FailureDrain() {
      if (std::uncaught_exceptions() <= pending) return;
      c->grp_pin_ = c->grp_dev_ = nullptr;
      c->grp_cap_ = c->grp_off_ = 0;
#if defined(LMN_BATCH) && !defined(LMN_EMU)
      // decided at run time: the batch library also exports the solo entry points (lmn_prove / lmn_prove_submit), whose
      // threads are outside any lock-step group and must drain like the main library's
      if (tls_batch_group == nullptr) (void)hipStreamSynchronize(c->stream_);
#elif !defined(LMN_EMU)
      (void)hipStreamSynchronize(c->stream_);
#endif
    }
  } failure_drain{this};
  ProofRun r(cfg.protocol_variant);
  r.tables = tables;
  r.n_tables = n_tables;
  r.settings = settings;
  r.lb = (int)cfg.log_blowup;
  r.n_slots = claim_slots(cfg.protocol_variant);
  r.log = g_log(this);
  r.log->reset();
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

  // Device-resident Fiat-Shamir of the commitment phases - sharded proofs included since round 6 (every rank runs the same
  // steps behind the all-gather of the subtree roots); LMN_HOST_FS=1 keeps the transcript on the host (round 1-4
  // behaviour, for A/B measurements)
  r.dev_fs = getenv("LMN_HOST_FS") == nullptr;
  run_setup(r);               // prove.cpp: validate the tables, size the arena, twiddles, transcript
  run_preprocessed(r);        // phase_trace.cpp: tree 0 (LUT columns)                      prover.rs:54-59
  run_main_trace(r);          // phase_trace.cpp: transpose + commit, claim mixed           prover.rs:70-179
  run_interaction(r);         // phase_logup.cpp: relation draws, logup columns + commit    prover.rs:186-298
  run_composition(r);         // phase_composition.cpp: constraint quotients + commit       prover.rs:312 (stwo::prover::prove)
  run_oods(r);                // phase_oods.cpp: OODS point, sampled values, self-check
  run_quotients(r);           // phase_oods.cpp: FRI quotient columns
  run_fri_commit(r);          // phase_fri.cpp: layers with the device-resident channel, last layer
  run_queries(r);             // phase_decommit.cpp: proof of work, query positions
  run_decommit(r);            // phase_decommit.cpp: one gather launch, proof assembly
  return run_finish(r);
}

void Context::run_setup(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  const lmn_table* tables = r.tables;
  const size_t n_tables = r.n_tables;
  // ---- validate + size
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
    // A column of 2^(ls + lb) evaluations joins the FRI line layer of 2^(ls + lb - 1); stwo's commit loop ends at the
    // last layer of 2^(log_last_layer + lb) values and asserts that every column has been consumed by then: a table
    // of 2^log_last_layer rows or fewer cannot be proved under this PcsConfig (the reference panics)
    if (ls <= (int)cfg.log_last_layer)
      throw LmnError(LMN_ERR_INVALID_ARGUMENT, "a table needs more than 2^log_last_layer rows (after padding)");
    infos.push_back({sp, tb.n_rows, ls, tb.rows, (tb.flags & LMN_TABLE_ROWS_ON_DEVICE) != 0});
    max_log = std::max(max_log, ls);
    uint64_t cells = (uint64_t)(sp->n_cols + 4 * sp->n_rel) << ls;
    words += cells * 2 + row_split(cells << lb);               // evals + coeffs + lde (this rank's row block)
    if (!infos.back().on_device) words += tb.n_rows * sp->n_cols;  // staging
    words += (uint64_t)sp->n_pre * ((2ull << ls) + row_split(2ull << ls));  // preprocessed columns: evals + coeffs + lde
    words += (4ull << ls) * 2;                                 // logup temps
    words += (4ull << (ls + 1)) * 3;                           // per-size composition scratch
    if (lb != 1) words += (uint64_t)(sp->n_cols + 4 * sp->n_rel + sp->n_pre) << (ls + 1);   // columns on the constraint domain
    if (shard_.active) words += (4ull << (ls + 1)) + 4096;     // halo rows of the last logup column group
    if (shard_rows_front(ls))                                  // received row blocks, gathered logup sums, scan output, coefficients
      words += (uint64_t)(sp->n_cols + 8 * sp->n_rel + 16) << ls;
    if (shard_all_to_all()) {                                  // own columns over all rows, packed copy, halo exchange
      const uint64_t G = shard_.world;
      words += 2 * (((uint64_t)sp->n_cols / G + 1) + ((uint64_t)(4 * sp->n_rel) / G + 1)) * (2ull << ls) + 6 * (2ull << ls);
    }
  }
  comp_log = max_log + 1;
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

  proof.claim.assign(n_slots, -1);
  proof.interaction_claim.assign(n_slots, {false, q_zero()});
  proof.pow_bits = cfg.pow_bits;
  proof.log_blowup = cfg.log_blowup;
  proof.log_last_layer = cfg.log_last_layer;
  proof.n_queries = cfg.n_queries;

  r.total_guard.reset(new StageTimer(this, log, stream_, C_TOTAL));
}

std::vector<uint8_t> Context::run_finish(ProofRun& r) {
  LMN_RUN_ALIASES(r);
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
  std::vector<uint8_t> bytes = proof_to_bincode(proof);
  hm.mark("serialized");
  return bytes;
}

}  // namespace lmn
