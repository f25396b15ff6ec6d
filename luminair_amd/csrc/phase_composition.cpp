// prove(), inside stwo::prover::prove (/root/reference/crates/prover/src/prover.rs:312): the composition polynomial - every
// component's constraint quotients accumulated per evaluation-domain size, interpolated and committed as tree 3.
#include "prove_run.h"

namespace lmn {

void Context::run_composition(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  // ---- stwo::prover::prove (prover.rs:312): composition polynomial
  // (device-resident transcript: the randomness was drawn and every coefficient laid out by the ChanStep of kind 2)
  int n_total = 0;
  for (auto& ci : inst) n_total += constraint_layout(*ci.spec, cfg.protocol_variant).n_protocol;
  std::vector<QM31> powers(std::max(n_total, 1));
  if (!r.dev_fs) {
    comp_alpha = channel.draw_felt();
    powers[0] = q_one();
    for (int k = 1; k < n_total; ++k) powers[k] = q_mul(powers[k - 1], comp_alpha);
  }
  size_t inst_index = 0;
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
      const uint32_t* pre_e[2] = {ci.pre_idx[0] >= 0 ? tree0.cols[ci.pre_idx[0]].lde : nullptr,
                                  ci.pre_idx[1] >= 0 ? tree0.cols[ci.pre_idx[1]].lde : nullptr};
      if (lb != 1) {
        // The constraints are evaluated on the domain of log size log_size + 1 (max_constraint_log_degree_bound,
        // add/component.rs:33-35).  At blow-up 2 that IS the committed LDE; at larger blow-ups it is not (canonic cosets of
        // different sizes are disjoint), so the component's columns are evaluated there from their coefficients.
        // (a sharded proof evaluates its own row block only; blow-ups above 2 keep every column's coefficients on every rank:
        // shard_a2a_columns)
        auto on_eval_domain = [&](const uint32_t* coeffs, int ncols) {
          const uint64_t rows = E >> sg;
          uint32_t* ev = arena_.alloc_words((size_t)ncols * rows);
          StageTimer t(this, log, stream_, C_FFT);
          if (sg == 0)
            timings.fft_launches += launch_fft(ev, E, coeffs, 1ull << ci.log_size, ci.log_size, ncols, e, tw(e), stream_);
          else
            timings.fft_launches += launch_fft_block(ev, rows, coeffs, 1ull << ci.log_size, ci.log_size, ncols, e, sg,
                                                     shard_.rank, tw(e), stream_);
          timings.fft_bytes += (uint64_t)ncols * (4ull << ci.log_size) + (uint64_t)ncols * 4ull * rows;
          timings.fft_butterflies += (uint64_t)ncols * (rows / 2) * (uint64_t)e;
          return (const uint32_t*)ev;
        };
        a.main = on_eval_domain(tree1.cols[ci.main_start].coeffs, ci.spec->n_cols);
        a.inter = on_eval_domain(tree2.cols[ci.inter_start].coeffs, 4 * ci.spec->n_rel);
        for (int k = 0; k < 2; ++k)
          if (ci.pre_idx[k] >= 0) pre_e[k] = on_eval_domain(tree0.cols[ci.pre_idx[k]].coeffs, 1);
      }
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
      a.pre = pre_e[0];
      a.pre2 = pre_e[1];
      a.claimed_shift = ci.d_claimed_shift;
      const ConstraintLayout L = constraint_layout(*ci.spec, cfg.protocol_variant);
      if (r.dev_fs) {
        a.d_coeff = r.d_coeff + 16 * inst_index;
        a.d_elems = &r.d_report->elems;
        a.es2 = ELEMS_NODE;
        for (int j = 0; j < ci.spec->n_rel; ++j)
          if (ci.spec->rel_elems[j] != ELEMS_NODE) a.es2 = ci.spec->rel_elems[j];
      } else {
        for (int k = 0; k < L.n_kernel; ++k) {
          a.coeff[k] = L.proto_index[k] < 0 ? q_zero() : powers[n_total - 1 - (k0 + L.proto_index[k])];
          if (L.neg[k]) a.coeff[k] = q_neg(a.coeff[k]);
        }
      }
      ++inst_index;
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
    if (r.dev_fs) {
      ChanStep step;
      plan_oods_step(r, step);
      lde_and_merkle(tree3, false, r.d_chan, &step);   // no wait: the launch that produces root 3 draws the OODS point
      hm.mark("composition enqueued (device transcript)");
      return;
    }
    lde_and_merkle(tree3);
    lmn_sync(stream_);
    tree3.merkle.finish_root();
    channel.mix_root(tree3.merkle.root);
  }
  hm.mark("sync3: root3 mixed");
  for (auto* t : trees) proof.commitments.push_back(t->merkle.root);
}

}  // namespace lmn
