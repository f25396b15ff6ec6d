// prove(), phase 2: the interaction trace (/root/reference/crates/prover/src/prover.rs:186-298;
// add/witness.rs:126-167 and siblings): relation element draws, logup fractions and prefix sums, commit, claimed sums.
#include "prove_run.h"

namespace lmn {

void Context::run_interaction(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  // ---- PHASE 2: interaction trace (prover.rs:186-298)
  if (!r.dev_fs) elems = draw_relation_elements(channel, cfg.protocol_variant);   // else: drawn on the device, `drawn` set in run_main_trace
  const DevElems* d_elems = r.dev_fs ? &r.d_report->elems : nullptr;
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
          a.es[j] = es;
          a.d_elems = d_elems;
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
        if (!ci.trace_evals) {
          // the transpose ran inside the interpolation (run_main_trace): the cells come from the table's rows
          if (sp->rel_pre[j]) throw LmnError(LMN_ERR_INTERNAL, "logup: a preprocessed relation of a table without column-major evaluations");
          a.rows = ci.rows_dev;
          a.row_words = (uint32_t)sp->n_cols;
          a.n_real = (uint32_t)ci.rows_n;
          a.vcol[j] = sp->rel_val[j];
          a.icol[j] = sp->rel_id[j];
          a.mcol[j] = sp->rel_mult[j];
          a.pad_val[j] = ci.pad.v[sp->rel_val[j]];
          a.pad_id[j] = sp->rel_id[j] >= 0 ? ci.pad.v[sp->rel_id[j]] : 0u;
          a.pad_mult[j] = ci.pad.v[sp->rel_mult[j]];
        } else {
          a.val[j] = column(sp->rel_val[j]);
          a.id[j] = sp->rel_id[j] >= 0 ? column(sp->rel_id[j]) : nullptr;
          a.mult[j] = ci.trace_evals + (uint64_t)sp->rel_mult[j] * n;
        }
        a.neg[j] = sp->rel_neg[j];
        a.z[j] = elems.z[es];
        a.alpha[j] = elems.alpha[es];
        a.es[j] = es;
        a.d_elems = d_elems;
      }
      a.inter = ievals;
      a.last_tmp = (QM31*)arena_.alloc_bytes(n * sizeof(QM31));
      int nb = logup_num_blocks((uint32_t)n);
      a.partials = arena_.alloc_words((size_t)nb * 4);
      a.n = (uint32_t)n;
      launch_logup_fracs(a, stream_);
      QM31* d_cs = (QM31*)arena_.alloc_bytes(2 * sizeof(QM31));
      uint32_t n_inv = m_inv((uint32_t)(n % P31));
      // claimed sum and shift come out of the scan's own block totals (no reduction launch over a.partials in front)
      QM31* bsums = (QM31*)arena_.alloc_bytes((size_t)logup_scan_num_blocks(ci.log_size) * sizeof(QM31));
      launch_logup_scan(a.last_tmp, d_cs, ci.log_size, ievals + (uint64_t)(nic - 4) * n, bsums, stream_, true, n_inv);
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
    if (r.dev_fs) {
      // no wait: the device mixes the claimed sums and the root, draws the composition randomness and lays out every
      // component's constraint coefficients (ChanStep kind 2)
      ChanStep step{};
      step.kind = 2;
      ChanCoeffPlan& plan = step.coeff;
      if (inst.size() > (size_t)CHAN_MAX_INST) throw LmnError(LMN_ERR_INTERNAL, "more components than claim slots");
      plan.n_inst = (int)inst.size();
      int k0 = 0;
      for (size_t i = 0; i < inst.size(); ++i) {
        const ConstraintLayout L = constraint_layout(*inst[i].spec, cfg.protocol_variant);
        plan.claimed[i] = inst[i].d_claimed_shift;
        plan.k0[i] = (int16_t)k0;
        plan.n_kernel[i] = (int8_t)L.n_kernel;
        for (int k = 0; k < 16; ++k) plan.proto_index[i][k] = (int8_t)(k < L.n_kernel ? L.proto_index[k] : -1);
        uint16_t neg = 0;
        for (int k = 0; k < L.n_kernel; ++k) neg |= (uint16_t)(L.neg[k] ? 1u << k : 0u);
        plan.neg[i] = neg;
        k0 += L.n_protocol;
      }
      plan.n_total = k0;
      r.d_coeff = (QM31*)arena_.alloc_bytes(inst.size() * 16 * sizeof(QM31));
      step.coeff_out = r.d_coeff;
      step.rep = r.d_report;
      lde_and_merkle(tree2, false, r.d_chan, &step);   // the launch that produces root 2 makes the step (ChanStep kind 2)
      hm.mark("interaction trace enqueued (device transcript)");
      return;
    }
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
}

}  // namespace lmn
