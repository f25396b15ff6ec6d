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

void Context::set_sample_points(ProofRun& r, QM31 tt) {
  QM31 t2 = q_sqr(tt);
  QM31 tinv = q_inv(q_add_m(t2, 1u));
  r.oods = QPt{q_mul(q_sub(q_one(), t2), tinv), q_mul(q_add(tt, tt), tinv)};
  r.points.assign(1, r.oods);
  for (size_t p = 1; p < r.neg_step.size(); ++p) r.points.push_back(qpt_add_m(r.oods, r.neg_step[p]));
}

// stwo::prover::prove's sanity check: the composition polynomial's OODS value must match the AIR at the sampled values
void Context::check_composition_identity(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  QM31 lhs = q_from_partial_evals(sampled[3][0][0], sampled[3][1][0], sampled[3][2][0], sampled[3][3][0]);
  QM31 rhs = eval_composition_at_point(inst, sampled, oods, elems, comp_alpha, cfg.protocol_variant);
  if (!q_eq(lhs, rhs) && !LMN_ABLATED(~0u)) throw LmnError(LMN_ERR_CONSTRAINTS, "ProverError(ConstraintsNotSatisfied)");
}

// The quotient kernels' tables without the host (kernels.h QuotPrepPlan): the sample batches of every LDE size are a
// function of the trees' layout alone (ColumnSampleBatch::new_vec groups by point in first-appearance order, as
// make_quotient_args does); what depends on the sampled values and on the randomness is computed by k_quot_prepare.
void Context::enqueue_quotient_tables(ProofRun& r, const QM31* d_vals) {
  LMN_RUN_ALIASES(r);
  ProofRun::FriPlan& fp = r.fri;
  struct Col {
    const uint32_t* lde;
    int lde_log;
    std::vector<std::pair<int, uint32_t>> samples;   // (point, index of the sampled value)
  };
  std::vector<Col> flat;
  uint32_t n_samples = 0;
  for (int t = 0; t < 4; ++t)
    for (size_t c = 0; c < trees[t]->cols.size(); ++c) {
      Col f{trees[t]->cols[c].lde, trees[t]->cols[c].log_size + lb, {}};
      for (size_t p = 0; p < spoints[t][c].size(); ++p) f.samples.push_back({spoints[t][c][p], n_samples++});
      flat.push_back(f);
    }
  std::set<int, std::greater<int>> size_set;
  for (auto& f : flat) size_set.insert(f.lde_log);
  sizes.assign(size_set.begin(), size_set.end());
  QuotPrepPlan* plan = (QuotPrepPlan*)pin_alloc(sizeof(QuotPrepPlan));
  memset(plan, 0, sizeof *plan);
  QuotPrepEntry* entries = (QuotPrepEntry*)pin_alloc((size_t)std::max<uint32_t>(n_samples, 1) * sizeof(QuotPrepEntry));
  plan->n_sizes = (int)sizes.size();
  plan->n_samples = (int)n_samples;
  plan->n_maps = std::max(r.comp_log, EVAL_LB);
  plan->entries = entries;
  plan->vals = d_vals;
  QM31* h_vals = (QM31*)result_block(((size_t)n_samples + 1) * sizeof(QM31));
  plan->vals_host = h_vals;
  r.h_vals = h_vals;
  plan->maps = r.d_maps;
  uint32_t at = 0;
  for (size_t si = 0; si < sizes.size(); ++si) {
    const int ls = sizes[si];
    QuotPrepSize& sz = plan->size[si];
    std::vector<int> batch_point;
    std::vector<std::vector<std::pair<const uint32_t*, uint32_t>>> batch_cols;
    for (auto& f : flat)
      if (f.lde_log == ls)
        for (auto& sm : f.samples) {
          size_t b = 0;
          while (b < batch_point.size() && batch_point[b] != sm.first) ++b;
          if (b == batch_point.size()) {
            batch_point.push_back(sm.first);
            batch_cols.emplace_back();
          }
          batch_cols[b].push_back({f.lde, sm.second});
        }
    if (batch_point.size() > (size_t)QUOT_MAX_BATCH) throw LmnError(LMN_ERR_INTERNAL, "too many sample batches");
    sz.n_batch = (int)batch_point.size();
    sz.first_entry = at;
    int local = 0;
    for (size_t b = 0; b < batch_point.size(); ++b) {
      sz.batch_start[b] = local;
      sz.point[b] = batch_point[b];
      for (auto& cv : batch_cols[b]) {
        entries[at++] = QuotPrepEntry{cv.first, cv.second, (uint16_t)si, (uint16_t)b};
        ++local;
      }
    }
    sz.batch_start[batch_point.size()] = local;
    sz.n_entries = local;
    if (local > QUOT_MAX_ENTRIES) throw LmnError(LMN_ERR_INTERNAL, "too many column samples");
    sz.entries_out = (QuotEntry*)arena_.alloc_bytes((size_t)std::max(local, 1) * sizeof(QuotEntry));
    sz.dev_out = (QuotDev*)arena_.alloc_bytes(sizeof(QuotDev));
    QuotientArgs a{};
    a.log_size = ls;
    a.nbatch = sz.n_batch;
    for (int b = 0; b <= sz.n_batch; ++b) a.batch_start[b] = sz.batch_start[b];
    a.entries = sz.entries_out;
    a.dev = sz.dev_out;
    a.tw_y = twY_[ls];
    a.tw_x = ls >= 2 ? twX_[ls] : nullptr;
    a.row0 = 0;
    a.log_rows = ls;
    a.out_stride = 1ull << ls;
    a.out = arena_.alloc_words(4ull << ls);
    r.qargs.push_back(a);
    quots.push_back({ls, a.out, false});
  }
  plan->n_entries = (int)at;
  // the FRI loop's channel is the device's own from here on; the tail's layer table rides along (page-locked -> device)
  if (fp.planned_ls0 != quots[0].log) plan_fri_layout(r, quots[0].log, quots.back().log);
  fp.d_chan = r.d_chan;
  fp.d_tail = nullptr;
  if (!fp.tail_table.empty()) {
    const size_t bytes = fp.tail_table.size() * sizeof(FriTailLayer);
    static_assert(sizeof(FriTailLayer) % 4 == 0, "the tail's table is copied word by word");
    FriTailLayer* pinned = (FriTailLayer*)pin_alloc(bytes);
    memcpy(pinned, fp.tail_table.data(), bytes);
    fp.d_tail = (FriTailLayer*)arena_.alloc_bytes(bytes);
    plan->copy_src = reinterpret_cast<const uint32_t*>(pinned);
    plan->copy_dst = reinterpret_cast<uint32_t*>(fp.d_tail);
    plan->copy_words = (uint32_t)(bytes / 4);
  }
  launch_quot_prepare(r.d_chan, plan, stream_);
}

// The wait for the FRI results is over (run_fri_commit) and the host has not replayed anything yet: roots, relation
// elements, claimed sums, composition randomness, OODS point (replay_device_transcript), then the sampled values, the
// composition identity and the quotient randomness - every draw checked against the device's.
void Context::finish_oods_on_host(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  replay_device_transcript(r, [&](QM31 tt) { set_sample_points(r, tt); });
  size_t k = 0;
  sampled.assign(4, {});
  std::vector<QM31> flat;
  for (int t = 0; t < 4; ++t) {
    sampled[t].resize(trees[t]->cols.size());
    for (size_t c = 0; c < trees[t]->cols.size(); ++c)
      for (size_t p = 0; p < spoints[t][c].size(); ++p) {
        sampled[t][c].push_back(r.h_vals[k]);
        flat.push_back(r.h_vals[k++]);
      }
  }
  proof.sampled_values = sampled;
  channel.mix_felts(flat);
  check_composition_identity(r);
  quot_alpha = channel.draw_felt();
  if (!q_eq(quot_alpha, r.h_vals[k]))
    throw LmnError(LMN_ERR_INTERNAL, "device/host transcript divergence at the quotient randomness");
  hm.mark("replay: sampled values, composition identity, quotient randomness");
}

void Context::run_oods(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  plan_sample_points(r);
  std::map<int, int>& prev_point_of_log = r.prev_point_of_log;
  const std::vector<Pt>& neg_step = r.neg_step;
  auto set_points = [&](QM31 tt) { set_sample_points(r, tt); };
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
  // unsharded proofs with the device-resident transcript do not wait here: the sampled values stay on the device, where
  // k_quot_prepare makes the transcript step and the quotient kernels' tables (LMN_HOST_QUOT=1: the wait of round 5)
  r.quot_dev = r.dev_fs && !shard_.active && getenv("LMN_HOST_QUOT") == nullptr && r.eval_jobs.size() <= (size_t)QUOT_PREP_MAX_SAMPLES &&
               r.eval_jobs.size() <= (size_t)QUOT_MAX_ENTRIES && !r.eval_jobs.empty();
  if (r.quot_dev) {
    std::set<int> lde_sizes;
    for (int t = 0; t < 4; ++t)
      for (auto& c : trees[t]->cols) lde_sizes.insert(c.log_size + lb);
    r.quot_dev = lde_sizes.size() <= (size_t)QUOT_PREP_MAX_SIZES;
  }
  if (r.quot_dev) {
    const QM31* d_vals = nullptr;
    {
      StageTimer st(this, log, stream_, C_OODS);
      eval_at_points(r.eval_jobs, points, comp_log, false, r.d_maps, (int)neg_step.size(), r.d_eval_jobs, &d_vals);
    }
    StageTimer st(this, log, stream_, C_QUOT);
    enqueue_quotient_tables(r, d_vals);
    hm.mark("oods + quotient tables enqueued (device transcript)");
    return;
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
  check_composition_identity(r);
}

void Context::run_quotients(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  if (r.quot_dev) {   // tables, randomness and the FRI loop's channel are the device's (enqueue_quotient_tables)
    sh = false;
    g = 0;
    fri_T = std::max(shard_.fri_min_log, (int)cfg.log_last_layer + lb);
    StageTimer st(this, log, stream_, C_QUOT);
    for (auto& a : r.qargs) launch_quotients(a, stream_);
    hm.mark("quotients enqueued");
    return;
  }
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
    stage_group_begin(n_samples * sizeof(QuotEntry) + (64 + sizeof(QuotDev)) * sizes.size() + 4096);
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
