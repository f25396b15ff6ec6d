// prove(), inside stwo::prover::prove: proof of work, query positions, and the decommitment of every tree and FRI layer
// through one planned gather launch; assembles the proof container.
#include "prove_run.h"

namespace lmn {

void Context::run_queries(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  // ---- proof of work + queries
  proof.proof_of_work = channel.grind(cfg.pow_bits);
  channel.mix_u64(proof.proof_of_work);
  const int ls0 = quots[0].log;
  queries.clear();
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
  pos_by_log.clear();
  for (int ls : sizes) pos_by_log[ls] = fold_positions(queries, ls0 - ls);

  hm.mark("pow+queries");
}

void Context::run_decommit(ProofRun& r) {
  LMN_RUN_ALIASES(r);
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
      secure_columns(fl.vals, fl.log, fl.sharded, g, lc);
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
      if (!entries.empty()) memcpy(d_e, entries.data(), entries.size() * sizeof(GatherEntry));
      if (!hs.jobs.empty() && sh) throw LmnError(LMN_ERR_INTERNAL, "sharded proofs keep whole trees");
      MerkleRecompute* d_j = (MerkleRecompute*)pin_alloc((hs.jobs.size() + 1) * sizeof(MerkleRecompute));
      if (!hs.jobs.empty()) memcpy(d_j, hs.jobs.data(), hs.jobs.size() * sizeof(MerkleRecompute));   // (memcpy from a null vector: UB even for 0 bytes)
      // an unsharded proof's gather writes straight to page-locked memory: nothing to download behind it
      uint32_t* d_o = sh ? arena_.alloc_words((size_t)slots * out_words) : (uint32_t*)result_block((size_t)out_words * 4);
      hm.mark("decommit planned");
      if (hm.on) fprintf(stderr, "[host] decommit: %zu runs gathered, %zu tree nodes recomputed\n", entries.size(), hs.jobs.size());
      launch_gather(arena_.base_words(), d_e, (uint32_t)entries.size(), d_j, (uint32_t)hs.jobs.size(), d_o, stream_);
      if (sh) gather_columns(d_o, 0, 1, out_words);
      gathered = sh ? (const uint32_t*)stage_download(d_o, (size_t)slots * out_words * 4) : d_o;
      lmn_sync(stream_);
      hm.mark("gathered");
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
    // (runs behind the proof's last wait: whole runs are copied, not words)
    auto take_q = [&](size_t nrefs) {
      const QM31* b = reinterpret_cast<const QM31*>(gathered + g);
      g += nrefs / 4 * 4;
      return std::vector<QM31>(b, b + nrefs / 4);
    };
    auto take_u32 = [&](size_t n) {
      std::vector<uint32_t> v(gathered + g, gathered + g + n);
      g += n;
      return v;
    };
    auto skip = [&](size_t n) { g += n; };
    auto take_hashes = [&](size_t n) {
      const Hash32* b = reinterpret_cast<const Hash32*>(gathered + g);
      g += 8 * n;
      return std::vector<Hash32>(b, b + n);
    };
    size_t pi = 0;
    auto fill_layer = [&](FriLayerProof& lp, const Hash32& root) {
      Plan& p = plans[pi++];
      lp.fri_witness = take_q(p.fri_wit.size());
      skip(p.queried.size());
      lp.decommitment.hash_witness = take_hashes(p.hash_wit.size());
      lp.decommitment.column_witness = take_u32(p.col_wit.size());
      lp.commitment = root;
    };
    fill_layer(proof.first_layer, first_merkle.root);
    proof.inner_layers.resize(inner.size());
    for (size_t i = 0; i < inner.size(); ++i) fill_layer(proof.inner_layers[i], inner[i].merkle.root);
    for (int t = 0; t < 4; ++t) {
      Plan& p = plans[pi++];
      skip(p.fri_wit.size());
      proof.queried_values.push_back(take_u32(p.queried.size()));
      Decommitment d;
      d.hash_witness = take_hashes(p.hash_wit.size());
      d.column_witness = take_u32(p.col_wit.size());
      proof.decommitments.push_back(d);
    }
  }
  hm.mark("decommit done");
  r.total_guard.reset();
  lmn_sync(stream_);
}

}  // namespace lmn
