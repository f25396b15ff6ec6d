// prove(), inside stwo::prover::prove: the OODS point and the sampled values of every committed polynomial (with the
// prover's own check of the composition identity), then the FRI quotient columns, one per LDE size.
#include "prove_run.h"

namespace lmn {

// The host's first wait of a proof with the device-resident transcript is over: replay the steps the k_chan_* kernels
// made - root 1, relation draws, claimed sums, root 2, composition randomness, root 3, OODS draw - on the host channel
// from what came back in the DevReport, and insist that every draw agrees.
void Context::replay_device_transcript(ProofRun& r, const std::function<void(QM31)>& set_points) {
  LMN_RUN_ALIASES(r);
  const DevReport& rep = *r.h_report;
  if (rep.bad == r.bad_mark)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace table holds a word that is not a canonical M31 (>= 2^31-1)");
  auto diverged = [](const char* what) {
    return LmnError(LMN_ERR_INTERNAL, std::string("device/host transcript divergence at ") + what);
  };
  memcpy(tree1.merkle.root.w, rep.roots[0], 32);
  memcpy(tree2.merkle.root.w, rep.roots[1], 32);
  memcpy(tree3.merkle.root.w, rep.roots[2], 32);
  channel.mix_root(tree1.merkle.root);
  hm.mark("replay: root1 mixed");
  elems = draw_relation_elements(channel, cfg.protocol_variant);
  for (int e = 0; e < N_ELEMS; ++e)
    if (elems.drawn[e] && (!q_eq(elems.z[e], rep.elems.z[e]) || !q_eq(elems.alpha[e], rep.elems.alpha[e])))
      throw diverged("the relation elements");
  for (size_t i = 0; i < inst.size(); ++i) {
    inst[i].claimed = rep.claimed[i];
    proof.interaction_claim[inst[i].spec->kind] = {true, rep.claimed[i]};
  }
  for (int k = 0; k < n_slots; ++k)
    if (proof.interaction_claim[k].first) channel.mix_felts({proof.interaction_claim[k].second});
  channel.mix_root(tree2.merkle.root);
  comp_alpha = channel.draw_felt();
  if (!q_eq(comp_alpha, rep.comp_alpha)) throw diverged("the composition randomness");
  channel.mix_root(tree3.merkle.root);
  for (auto* t : trees) proof.commitments.push_back(t->merkle.root);
  const QM31 tt = channel.draw_felt();
  if (!q_eq(tt, rep.t)) throw diverged("the OODS point");
  set_points(tt);
  hm.mark("replay: transcript up to the OODS point");
}

// ---- OODS point + mask points: point 0 = the OODS point, then per trace size the point one trace step before it
void Context::plan_sample_points(ProofRun& r) {
  if (!r.neg_step.empty()) return;
  r.neg_step.assign(1, Pt{1u, 0u});
  for (auto& ci : r.inst) {
    if (r.prev_point_of_log.count(ci.log_size)) continue;
    r.prev_point_of_log[ci.log_size] = (int)r.neg_step.size();
    r.neg_step.push_back(pt_of_index((0x80000000u - subgroup_gen_index(ci.log_size)) & 0x7fffffffu));  // -step
  }
}

// sample point indices per tree / column in sampled_values order, and the evaluation jobs they make (needs the four trees'
// columns with their coefficient pointers: called once the composition polynomial has its place)
void Context::plan_eval_jobs(ProofRun& r) {
  if (!r.spoints.empty()) return;
  LMN_RUN_ALIASES(r);
  plan_sample_points(r);
  spoints.assign(4, {});
  spoints[0].assign(tree0.cols.size(), {0});
  spoints[1].assign(tree1.cols.size(), {0});
  spoints[2].assign(tree2.cols.size(), {0});
  spoints[3].assign(4, {0});
  for (auto& ci : inst) {
    int nic = 4 * ci.spec->n_rel;
    for (int c = nic - 4; c < nic; ++c) spoints[2][ci.inter_start + c] = {r.prev_point_of_log[ci.log_size], 0};
  }
  r.eval_jobs.clear();
  for (int t = 0; t < 4; ++t)
    for (size_t c = 0; c < trees[t]->cols.size(); ++c)
      for (int p : spoints[t][c])
        r.eval_jobs.push_back({trees[t]->cols[c].coeffs, trees[t]->cols[c].log_size, p, trees[t]->cols[c].owner});
}

// Device-resident transcript: the launch that produces the composition tree's root draws the OODS point and expands the
// sample points' mappings (ChanStep kind 3); everything the device-resident steps produced comes back in one block that
// step writes to page-locked memory, valid after the wait inside run_oods's eval_at_points.
void Context::plan_oods_step(ProofRun& r, ChanStep& step) {
  plan_sample_points(r);
  const std::vector<Pt>& neg_step = r.neg_step;
  if (neg_step.size() > (size_t)CHAN_MAX_POINTS) throw LmnError(LMN_ERR_INTERNAL, "more sample points than the device transcript plans for");
  step = ChanStep{};
  step.kind = 3;
  ChanOodsPlan& plan = step.oods;
  plan.n_points = (int)neg_step.size();
  plan.n_maps = std::max(r.comp_log, EVAL_LB);
  for (size_t p = 0; p < neg_step.size(); ++p) {
    plan.step_x[p] = neg_step[p].x;
    plan.step_y[p] = neg_step[p].y;
  }
  r.d_maps = (QM31*)arena_.alloc_bytes((size_t)plan.n_points * plan.n_maps * sizeof(QM31));
  DevReport* h_rep = (DevReport*)result_block(sizeof(DevReport));
  r.h_report = h_rep;
  step.maps_out = r.d_maps;
  step.rep_host = reinterpret_cast<uint32_t*>(h_rep);
  step.rep = r.d_report;
  // the evaluation kernels' job table goes to device memory through the step's workgroup (its lanes fetch it from
  // page-locked memory next to the plan): no transfer in front of the evaluation
  plan_eval_jobs(r);
  const size_t bytes = r.eval_jobs.size() * sizeof(EvalJob);
  static_assert(sizeof(EvalJob) % 4 == 0, "the job table is copied word by word");
  EvalJob* pinned = (EvalJob*)pin_alloc(bytes);
  memcpy(pinned, r.eval_jobs.data(), bytes);
  r.d_eval_jobs = (EvalJob*)arena_.alloc_bytes(bytes);
  step.copy_src = reinterpret_cast<const uint32_t*>(pinned);
  step.copy_dst = reinterpret_cast<uint32_t*>(r.d_eval_jobs);
  step.copy_words = (uint32_t)(bytes / 4);
}

void Context::run_oods(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  plan_sample_points(r);
  std::map<int, int>& prev_point_of_log = r.prev_point_of_log;
  const std::vector<Pt>& neg_step = r.neg_step;
  auto set_points = [&](QM31 tt) {
    QM31 t2 = q_sqr(tt);
    QM31 tinv = q_inv(q_add_m(t2, 1u));
    oods = QPt{q_mul(q_sub(q_one(), t2), tinv), q_mul(q_add(tt, tt), tinv)};
    points.assign(1, oods);
    for (size_t p = 1; p < neg_step.size(); ++p) points.push_back(qpt_add_m(oods, neg_step[p]));
  };
  if (!r.dev_fs) set_points(channel.draw_felt());   // (device-resident transcript: run_composition's plan_oods_step)
  plan_eval_jobs(r);   // (device-resident transcript: already made by plan_oods_step)
  sampled.assign(4, {});
  if (!shard_.active) {
    // the host is about to wait for the sampled values and is a millisecond ahead of the device: lay out the FRI commit
    // loop's buffers now instead of behind the wait (plan_fri_layout needs the LDE sizes only)
    int lo = 1 << 30, hi = 0;
    for (int t = 0; t < 4; ++t)
      for (auto& c : trees[t]->cols) {
        lo = std::min(lo, c.log_size + lb);
        hi = std::max(hi, c.log_size + lb);
      }
    if (hi > 0) plan_fri_layout(r, hi, lo);
  }
  {
    StageTimer st(this, log, stream_, C_OODS);
    std::vector<QM31> vals = eval_at_points(r.eval_jobs, points, comp_log, /*split=*/true, r.dev_fs ? r.d_maps : nullptr,
                                            (int)neg_step.size(), r.dev_fs ? r.d_eval_jobs : nullptr);
    if (r.dev_fs) replay_device_transcript(r, set_points);
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
}

void Context::run_quotients(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  // ---- FRI quotients, one secure column per LDE size (descending)
  quot_alpha = channel.draw_felt();
  hm.mark("sampled mixed, oods check, alpha drawn");
  // sharding of the FRI part: a quotient column / FRI layer of more than 2^fri_T rows is split into row blocks
  // (pair folds stay inside a block: rows 2i and 2i+1 are adjacent in bit-reversed order); smaller ones are
  // all-gathered once and finished identically on every rank
  sh = shard_.active;
  g = sh ? shard_.g : 0;
  fri_T = std::max(shard_.fri_min_log, (int)cfg.log_last_layer + lb);
  auto sharded_log = [&](int lg) { return r.sharded_log(lg); };
  struct FlatCol {
    const uint32_t* lde;  // all rows, or this rank's block of them (sharded proof)
    int lde_log;
    std::vector<std::pair<int, QM31>> samples;  // (point index, value)
  };
  std::vector<FlatCol> flat;
  flat.reserve(tree0.cols.size() + tree1.cols.size() + tree2.cols.size() + tree3.cols.size());
  for (int t = 0; t < 4; ++t)
    for (size_t c = 0; c < trees[t]->cols.size(); ++c) {
      FlatCol f{trees[t]->cols[c].lde, trees[t]->cols[c].log_size + lb, {}};
      for (size_t p = 0; p < spoints[t][c].size(); ++p) f.samples.push_back({spoints[t][c][p], sampled[t][c][p]});
      flat.push_back(f);
    }
  std::set<int, std::greater<int>> size_set;
  for (auto& f : flat) size_set.insert(f.lde_log);
  sizes.assign(size_set.begin(), size_set.end());
  {
    StageTimer st(this, log, stream_, C_QUOT);
    // Everything this phase and the FRI commit loop read from tables on the device - the sample entries of each LDE size,
    // the channel's start state, the tail's layer table - goes up in ONE transfer ahead of the first quotient kernel
    // (each transfer is a blit launch of ~5 us in the proof's one stream)
    size_t n_samples = 0;
    for (auto& f : flat) n_samples += f.samples.size();
    stage_group_begin(n_samples * sizeof(QuotEntry) + 64 * sizes.size() + 4096);
    std::vector<QuotientArgs> qargs;
    for (int ls : sizes) {
      std::vector<const uint32_t*> ptrs;
      std::vector<std::vector<std::pair<int, QM31>>> smp;
      ptrs.reserve(flat.size());
      smp.reserve(flat.size());
      for (auto& f : flat)
        if (f.lde_log == ls) {   // (each column has one size: its samples move on)
          ptrs.push_back(f.lde);
          smp.push_back(std::move(f.samples));
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
      qargs.push_back(a);
      quots.push_back({ls, vals, qs});
    }
    plan_fri_buffers(r);
    hm.mark("quotient + FRI tables built");
    stage_group_end();
    hm.mark("quotient + FRI tables uploaded");
    for (size_t k = 0; k < sizes.size(); ++k) {
      const int ls = sizes[k];
      const QuotientArgs& a = qargs[k];
      // LMN_FRI_OVERLAP (experiment): unsharded proofs with two LDE sizes compute the second (smaller) size on the second
      // stream, next to the leaf hashing of the first size's quotient columns; build_merkle_levels waits for it before level `ls`
      const bool overlap = !sh && sizes.size() == 2 && ls == sizes[1] && second_stream_wanted();
      if (overlap) {
        lmn_event_record(ev_fork_, stream_);            // everything enqueued so far (incl. the table upload)
        lmn_stream_wait_event(stream2_, ev_fork_);
        launch_quotients(a, stream2_);
        lmn_event_record(ev_join_, stream2_);
        wait_before_level_ev_ = ev_join_;
        wait_before_level_ = ls;
      } else {
        launch_quotients(a, stream_);
      }
      if (sh && !quots[k].sharded) gather_columns(quots[k].vals, 1ull << ls, 4, (1ull << ls) >> g);
    }
  }

  hm.mark("quotients enqueued");
}

}  // namespace lmn
