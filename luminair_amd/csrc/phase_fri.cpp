// prove(), inside stwo::prover::prove: the FRI commit phase.  The layer loop runs on the device
// without host round trips (device-resident channel; the host replays and cross-checks it), folds are computed inside
// the next layer's leaf hashing where nothing else needs the values first; then the last layer's polynomial.
#include "prove_run.h"

namespace lmn {

// The FRI commit loop's buffers, in two steps.  plan_fri_layout needs only the quotient columns' sizes: run_oods calls it
// BEFORE its wait for the sampled values, while the host is a millisecond ahead of the device anyway; plan_fri_buffers
// (run_quotients, inside its upload group, on the proof's critical path) then only adds the channel's state and stages the
// tail's table.
void Context::plan_fri_layout(ProofRun& r, int ls0, int smallest_quot_log) {
  ProofRun::FriPlan& fp = r.fri;
  const int lb = r.lb;
  fp.last_size_log = (int)cfg.log_last_layer + lb;   // <= ls0 - 1: run_setup refuses a trace smaller than the last layer
  if (fp.last_size_log > ls0 - 1) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "FRI: first line layer smaller than the last layer");
  fp.max_layers = ls0 + 1;
  const size_t root_words = (size_t)fp.max_layers * 8, alpha_words = (size_t)fp.max_layers * 4;
  const size_t last_words = 4ull << fp.last_size_log;
  fp.d_out = arena_.alloc_words(root_words + alpha_words + last_words);
  fp.d_roots = fp.d_out;
  fp.d_alphas = (QM31*)(fp.d_out + root_words);
  fp.d_last = fp.d_out + root_words + alpha_words;
  fp.out_bytes = (root_words + alpha_words + last_words) * 4;
  fp.tail_log = -1;
  fp.tail_table.clear();
  fp.planned_ls0 = ls0;
  if (shard_.active) return;
  // unsharded: the layers from min(2^10, the size at which the last quotient column has joined) down are one launch
  const int tail_log = std::min(std::min(10, ls0 - 1), smallest_quot_log - 1);
  if (tail_log <= fp.last_size_log) return;
  const int n_tail = tail_log - fp.last_size_log;
  fp.tail_log = tail_log;
  fp.tail_first = arena_.alloc_words(4ull << tail_log);
  fp.tail_table.resize(n_tail);
  fp.tail_layers.reserve(n_tail);
  uint32_t* layer = fp.tail_first;
  for (int li = 0; li < n_tail; ++li) {
    const int L = tail_log - li;
    fp.tail_layers.emplace_back();
    FriLayer& fl = fp.tail_layers.back();
    fl.log = L;
    fl.vals = layer;
    fl.sharded = false;
    fl.merkle.max_log = L;
    fl.merkle.layers.assign(L + 1, nullptr);
    for (int l = 0; l <= L; ++l) fl.merkle.layers[l] = arena_.alloc_words((size_t)8 << l);
    uint32_t* next = L - 1 == fp.last_size_log ? fp.d_last : arena_.alloc_words(4ull << (L - 1));
    FriTailLayer& t = fp.tail_table[li];
    t.vals = layer;
    t.next = next;
    t.itw = itwX_[L + 1];
    for (int l = 0; l <= L; ++l) t.merkle[l] = fl.merkle.layers[l];
    layer = next;
  }
}

void Context::plan_fri_buffers(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  ProofRun::FriPlan& fp = r.fri;
  if (fp.planned_ls0 != quots[0].log) plan_fri_layout(r, quots[0].log, quots.back().log);   // (sharded proofs, level-2 callers)
  DevChannel hc{};
  memcpy(hc.digest, channel.digest().w, 32);   // nothing is mixed between the draw of the quotient randomness and the first layer's root
  hc.n_sent = 0;
  hc.variant = (cfg.protocol_variant & LMN_PV_DRAW_CTR_U32) ? 1u : 0u;   // the only encoding the FRI loop's channel ops depend on
  fp.d_chan = (DevChannel*)stage_upload(&hc, sizeof hc);
  fp.d_tail = fp.tail_table.empty() ? nullptr : upload_vec(fp.tail_table);
}

void Context::run_fri_commit(ProofRun& r) {
  LMN_RUN_ALIASES(r);
  // ---- FRI commit
  auto secure_cols = [&](const uint32_t* vals, int lg, bool s, std::vector<ColRef>& out) { secure_columns(vals, lg, s, g, out); };
  auto sharded_log = [&](int lg) { return r.sharded_log(lg); };
  {
    StageTimer st(this, log, stream_, C_FRI);
    for (auto& q : quots) secure_cols(q.vals, q.log, q.sharded, first_cols);
    // The FRI commit loop runs without host round trips: a device-resident copy of the channel
    // mixes each layer root and draws the folding alpha; the host replays the same steps afterwards.
    int ls0 = quots[0].log;
    ProofRun::FriPlan& fp = r.fri;   // buffers and tables: plan_fri_buffers (uploaded with the quotient phase's tables)
    const int last_size_log = (int)cfg.log_last_layer + lb;
    DevChannel* d_ch = fp.d_chan;
    QM31* d_alphas = fp.d_alphas;
    uint32_t* d_roots = fp.d_roots;
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
    // the tail's first layer and the last layer have their places already (FriPlan)
    auto layer_alloc = [&](int lg, bool s) {
      if (!s && lg == fp.last_size_log) return fp.d_last;
      if (!s && lg == fp.tail_log) return fp.tail_first;
      return arena_.alloc_words(s ? (4ull << (lg - g)) : (4ull << lg));
    };
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
    const bool fuse_folds = getenv("LMN_NO_FOLD_FUSION") == nullptr;   // (read per proof: the tests toggle it)
    if (!sh && fuse_folds) {
      pend = {true, true, quots[0].vals, ls0, d_alphas + (n_roots - 1)};
    } else {
      fold(true, layer, lay_sh, quots[0].vals, ls0, quots[0].sharded, d_alphas + (n_roots - 1), 0);
      if (quots[0].sharded && !lay_sh) gather_columns(layer, 1ull << layer_log, 4, (1ull << layer_log) >> g);
    }
    size_t qi = 1;
    // the line fold that produces the tail's first layer is left to the tail's launch (FriTailIo)
    FriTailIo tail_pre{nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    const uint32_t* h_out = nullptr;   // the result block on the host: written by the tail itself, or downloaded
    while (layer_log > last_size_log) {
      if (pend.on && layer_log <= 10) materialise(layer);
      const bool into_tail = !lay_sh && layer_log <= 10 && qi == quots.size();
      if (tail_pre.src && !(into_tail && layer_log == fp.tail_log && layer == fp.tail_first))
        throw LmnError(LMN_ERR_INTERNAL, "FRI: a fold left to the tail, and no tail where it was planned");
      if (into_tail) {
        // all remaining layers fit one block: commit + fold them in a single launch
        int n_tail = layer_log - last_size_log;
        FriTailLayer* d_tl = fp.d_tail;
        if (layer_log == fp.tail_log && layer == fp.tail_first) {
          for (auto& fl : fp.tail_layers) inner.push_back(fl);
          layer = fp.d_last;
        } else {
          // sharded proofs (the replicated part starts where the shard plan says)
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
            uint32_t* next = layer_alloc(L - 1, false);
            tl[li].vals = layer;
            tl[li].next = next;
            tl[li].itw = itwX_[L + 1];
            for (int l = 0; l <= L; ++l) tl[li].merkle[l] = fl.merkle.layers[l];
            inner.push_back(fl);
            layer = next;
          }
          d_tl = upload_vec(tl);
        }
        {
          StageTimer t(this, log, stream_, C_MERKLE);
          if (!sh && layer == fp.d_last) {   // the tail ends the loop: it hands the result block over itself
            uint32_t* mirror = (uint32_t*)result_block(fp.out_bytes);
            tail_pre.out_block = fp.d_out;
            tail_pre.out_mirror = mirror;
            tail_pre.out_words = (uint32_t)(fp.out_bytes / 4);
            h_out = mirror;
          }
          launch_fri_tail(d_ch, d_tl, n_tail, layer_log, d_alphas + n_roots, d_roots + 8 * n_roots, stream_, &tail_pre);
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
      } else if (!sh && fuse_folds && !joins && qi == quots.size() && next_log == fp.tail_log && next == fp.tail_first &&
                 next_log > last_size_log) {
        tail_pre.src = layer;   // the tail starts with this fold: no launch for it
        tail_pre.itw = itwX_[layer_log + 1];
        tail_pre.alpha = d_alpha;
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
    if (qi != quots.size()) throw LmnError(LMN_ERR_INTERNAL, "FRI: unconsumed columns");
    if (n_roots > fp.max_layers || layer_log != fp.last_size_log || layer != fp.d_last)
      throw LmnError(LMN_ERR_INTERNAL, "FRI: the layer plan does not match the layers committed");
    last_log = layer_log;
    // roots | alphas | last layer: one block, one download
    if (!h_out) h_out = (const uint32_t*)stage_download(fp.d_out, fp.out_bytes);
    const uint32_t* h_roots = h_out;
    const QM31* h_alphas = (const QM31*)(h_out + ((const uint32_t*)fp.d_alphas - fp.d_out));
    const uint32_t* raw = h_out + (fp.d_last - fp.d_out);
    lmn_sync(stream_);
    // The proof's first wait (unsharded proofs): the host's own transcript still stands at the claims - replay root 1 .. the
    // quotient randomness now, then the FRI roots below.  (Replaying early, on a word k_quot_prepare stores into page-locked
    // memory ~1 ms before the FRI loop is through, was measured: no latency gain - 2.197 vs 2.182 ms - and the polling costs
    // the host CPU this path saves: 2.1 instead of 3.7 ms of CPU per proof with 8 proofs in flight.)
    if (r.quot_dev) finish_oods_on_host(r);
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
}

}  // namespace lmn
