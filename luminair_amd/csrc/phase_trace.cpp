// prove(), phases 0 and 1: the preprocessed tree (LUT columns, /root/reference/crates/prover/src/prover.rs:54-59) and the
// main trace - pad + AoS -> SoA, interpolate, extend, commit (prover.rs:70-179; add/witness.rs:33-108).
#include "prove_run.h"

namespace lmn {

void Context::run_preprocessed(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  const lmn_settings* settings = r.settings;
  // ---- PHASE 0: preprocessed trace (prover.rs:54-59): empty tree (root = blake2s("")) unless a lookup
  // component is present.  Columns in PreProcessedTrace order (preprocessed.rs:157-179: sin, exp2, log2
  // LUT pairs from the settings, then the 8-bit range check whose row r holds r), stable-sorted by size
  // descending (PreProcessedTrace::new).
  for (auto& ti : infos) {
    Instance ci{};
    ci.spec = ti.spec;
    ci.log_size = ti.log_size;
    inst.push_back(ci);
  }
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
}

void Context::run_main_trace(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  // ---- PHASE 1: main trace (prover.rs:70-179)
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
      PadRow pad{};
      if (ti.spec->is_last_col >= 0) pad.v[ti.spec->is_last_col] = 1u;
      for (int k = 0; k < ti.spec->n_pad; ++k) pad.v[ti.spec->pad_col[k]] = ti.spec->pad_val[k];
      inst[t].rows_dev = d_rows;
      inst[t].rows_n = ti.n_rows;
      inst[t].pad = pad;
      // LMN_ROWS_FUSION=1 (a switch, off by default): big unsharded tables get no transpose launch - the first pass of the
      // interpolation reads the rows itself (fft_fixed.hip k_fft_rows_fx) and nothing is stored column-major before it;
      // logup fractions then read the table's rows.  Measured (docs/HISTORY.md, round 6): skipping the transpose outright is
      // worth 5 % proofs/s, but the pass that absorbs it costs what the two launches cost (54 us against 27 + 27 solo;
      // +0.4 % under load: inside the noise) - byte-identical proofs, kept for the next idea.  Components with
      // preprocessed (LUT) columns keep the plain path; LMN_ROWS_FUSION_MIN_LOG (default 18) lowers the size threshold.
      const char* fmin = getenv("LMN_ROWS_FUSION_MIN_LOG");
      const char* fon = getenv("LMN_ROWS_FUSION");
      const bool try_fused = fon && atoi(fon) != 0 && !shard_.active && lb == 1 && ti.spec->n_pre == 0 &&
                             fft_interp_extend_supported(ti.log_size) && ti.log_size >= (fmin ? std::max(13, atoi(fmin)) : 18) &&
                             getenv("LMN_NO_FFT_FIXED") == nullptr;
      if (try_fused) {
        inst[t].trace_evals = nullptr;
      } else {
        uint32_t* evals = arena_.alloc_words((size_t)ti.spec->n_cols * nb);
        launch_transpose_pad_rows(d_rows, ti.n_rows, ti.spec->n_cols, ti.log_size, evals, nb, blk0, nb, pad, d_bad, stream_, bad_mark);
        inst[t].trace_evals = evals;
      }
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
      CommitOut co;
      bool fused = false;
      if (!ci.trace_evals) {
        uint32_t* lde = arena_.alloc_words((size_t)nc * 2 * n);
        {
          StageTimer t(this, log, stream_, C_FFT);
          fused = launch_interp_extend_rows(coeffs, n, ci.rows_dev, ci.rows_n, ci.pad, d_bad, bad_mark, lde, 2 * n, nc, ci.log_size,
                                            itw(ci.log_size), tw(ci.log_size + 1), stream_);
        }
        if (fused) {
          timings.fft_launches += 3;
          timings.fft_bytes += (uint64_t)nc * 20ull * n;                         // as interpolate_for_commit counts the two transforms
          timings.fft_butterflies += (uint64_t)nc * (n / 2) * (uint64_t)ci.log_size + (uint64_t)nc * n * (uint64_t)ci.log_size;
          co.lde = lde;
          co.stride = 2 * n;
        } else {
          throw LmnError(LMN_ERR_INTERNAL, "main trace: no row pass for a size run_main_trace had chosen it for");
        }
      }
      if (!fused) co = interpolate_for_commit(coeffs, ci.trace_evals, nc, ci.log_size, -1, ci.rows_sharded);
      ci.main_start = off;
      off += nc;
      for (int c = 0; c < nc; ++c)
        tree1.cols.push_back({ci.log_size, coeffs + (uint64_t)c * n, co.lde ? co.lde + (uint64_t)c * co.stride : nullptr,
                              co.sharded, co.owner_of(c)});
    }
    for (int k = 0; k < n_slots; ++k)  // LuminairClaim::mix_into (crates/air/src/lib.rs:52-104)
      if (proof.claim[k] >= 0) channel.mix_u64((uint64_t)proof.claim[k]);
    r.bad_mark = bad_mark;
    if (r.dev_fs) {
      // no wait: the launch that produces the root also mixes it and draws the relation elements (ChanStep kind 1); the
      // host replays the step in run_oods, where the non-canonical-word verdict is read as well
      r.d_chan = (DevChannel*)arena_.alloc_bytes(sizeof(DevChannel));
      r.d_report = (DevReport*)arena_.alloc_bytes(sizeof(DevReport));
      ChanStep step{};
      step.kind = 1;
      memcpy(step.start.digest, channel.digest().w, 32);
      step.start.n_sent = 0;
      step.start.variant = (cfg.protocol_variant & LMN_PV_DRAW_CTR_U32) ? 1u : 0u;
      step.bad_word = d_bad;
      int sets[CHAN_N_ELEMS];
      const int n_draws = relation_draw_sets(cfg.protocol_variant, sets);
      if (n_draws < 1 || n_draws > CHAN_N_ELEMS) throw LmnError(LMN_ERR_INTERNAL, "relation draws: bad count");
      step.sets.n = n_draws;
      for (int i = 0; i < n_draws; ++i) step.sets.set[i] = sets[i];
      step.rep = r.d_report;
      lde_and_merkle(tree1, false, r.d_chan, &step);
      for (int i = 0; i < n_draws; ++i)
        if (sets[i] >= 0) elems.drawn[sets[i]] = true;
      hm.mark("main trace enqueued (device transcript)");
      return;
    }
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
}

}  // namespace lmn
