// Commitment of a batch of columns (`tree_builder.extend_evals` + `commit`, /root/reference/crates/prover/src/prover.rs:56-59,
// 179,298; crates/air/src/utils.rs:112-128): interpolation, low-degree extension, Blake2s Merkle tree in fused subtree
// launches, and their sharded forms (column-parallel stage A, all-to-all stage B, row-block stage C).
#include "prover_internal.h"

namespace lmn {

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
                                  QM31* alpha_out, uint32_t* root_copy, const MerkleFold* fold, std::vector<MerkleCut>* cuts,
                                  const ChanStep* step) {
  bool chan_done = false;
  // LMN_CHAN_STEP_SEPARATE=1 (measurements, tests): a commitment phase's transcript step as a launch of its own behind the tree
  const bool step_inside = step && getenv("LMN_CHAN_STEP_SEPARATE") == nullptr;
  const ChanStep* d_step = nullptr;   // the kernels read the plan from page-locked memory (one load per lane)
  if (step) {
    check_chan_step(*step);
    ChanStep* p = (ChanStep*)pin_alloc(sizeof(ChanStep));
    *p = *step;
    d_step = p;
  }
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
        const bool with_ch = to_root && ch && (!step || step_inside);
        launch_merkle_small(prev, sg, (int)lc.size(), 1u << level, outs, nfused, with_ch ? ch : nullptr, alpha_out,
                            root_copy, stream_, with_ch ? d_step : nullptr, with_ch && step ? step->kind : 0);
        if (with_ch) chan_done = true;
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
  if (ch && !chan_done) {
    if (step)
      launch_chan_step(ch, d_step, step->kind, layers[0], stream_);
    else
      launch_chan_mix_root_draw(ch, layers[0], alpha_out, root_copy, stream_);
  }
}

void Context::build_merkle(DevMerkle& m, const std::vector<ColRef>& cols_sorted, DevChannel* ch, QM31* alpha_out,
                           uint32_t* root_copy, bool sharded, const MerkleFold* fold, const ChanStep* step) {
  if (fold && sharded) throw LmnError(LMN_ERR_INTERNAL, "merkle: folded leaf levels are not sharded");
  if (step && (!ch || cols_sorted.empty())) throw LmnError(LMN_ERR_INTERNAL, "merkle: a transcript step needs a tree and a channel");
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
    build_merkle_levels(m.layers, m.max_log, per_level, ch, alpha_out, root_copy, fold, merkle_cut_ ? &m.cuts : nullptr, step);
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
  // the transcript step that consumes the root (a commitment phase's ChanStep, or a FRI layer's mix_root + draw) runs behind
  // the gather on every rank alike: inside the launch that hashes the top g levels, or as a launch of its own
  const ChanStep* d_step = nullptr;
  if (step) {
    check_chan_step(*step);
    ChanStep* p = (ChanStep*)pin_alloc(sizeof(ChanStep));
    *p = *step;
    d_step = p;
  }
  if (g == 0) {
    if (ch && step)
      launch_chan_step(ch, d_step, step->kind, level_g, stream_);
    else if (ch)
      launch_chan_mix_root_draw(ch, level_g, alpha_out, root_copy, stream_);
    return;
  }
  for (int l = g - 1; l >= 0; --l) m.layers[l] = arena_.alloc_words((size_t)8 << l);
  MerkleLevels outs{};
  for (int l = 0; l <= g - 1; ++l) outs.p[l] = m.layers[g - 1 - l];
  MerkleSegs none{};
  StageTimer t(this, g_log(this), stream_, C_MERKLE);
  launch_merkle_small(level_g, none, 0, 1u << (g - 1), outs, g - 1, ch, alpha_out, root_copy, stream_, d_step,
                      step ? step->kind : 0);
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
void Context::lde_and_merkle(DevTree& tree, bool fetch_root, DevChannel* step_ch, const ChanStep* step) {
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
  build_merkle(tree.merkle, sorted, step ? step_ch : nullptr, nullptr, nullptr, sh, nullptr, step);
  if (fetch_root) fetch_root_async(tree.merkle);   // (device-resident transcript: the root travels with the DevReport)
}

}  // namespace lmn
