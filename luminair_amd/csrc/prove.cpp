// prove(): MI355X replacement for /root/reference/crates/prover/src/prover.rs:28-319.
// The Fiat-Shamir channel, proof assembly and (tiny) decommitment bookkeeping run on the host;
// every per-row / per-coefficient pass runs in the gfx950 kernels on trace data
// that stays resident in HBM from the first transpose to the last query gather.
#include "prover_internal.h"

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

}  // namespace lmn
