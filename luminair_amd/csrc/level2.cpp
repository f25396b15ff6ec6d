// Level-2 ops on device handles (include/luminair_hip.h, `lmn_col_*` / `lmn_tree_*`): the stwo `Backend`-shaped
// surface a Rust `HipBackend` would bind (/root/reference/crates/prover/src/prover.rs:38-46,312,
// /root/reference/crates/air/src/utils.rs:112-128).  Columns live in HBM from lmn_col_from_cpu to lmn_col_to_cpu;
// every op is one or a few launches of the same gfx950 kernels `lmn_prove` uses, on the context's stream.
#include "capi_internal.h"

#include <algorithm>

struct lmn_col {
  uint32_t* d;
  uint32_t ncols, log_size;
  bool view = false;  // aliases columns of another handle (lmn_col_view): does not own `d`
  uint64_t words() const { return (uint64_t)ncols << log_size; }
};
struct lmn_tree {
  uint32_t* slab;                 // all layers, root first
  std::vector<uint32_t*> layers;  // layers[k]: 2^k hashes
  int max_log;
  lmn::Hash32 root;
};

namespace lmn {

constexpr uint32_t COL_MAX_LOG = 27;  // 2^26-row traces have 2^27-row LDEs

static lmn_col* new_col(uint32_t ncols, uint32_t log_size) {
  if (ncols == 0 || ncols > 4096 || log_size > COL_MAX_LOG)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "column handle: bad shape");
  lmn_col* c = new lmn_col{nullptr, ncols, log_size, false};
  try {
    c->d = (uint32_t*)lmn_dev_malloc(c->words() * 4);
  } catch (const LmnError& e) {
    delete c;
    throw LmnError(LMN_ERR_OUT_OF_MEMORY, std::string("column allocation failed: ") + e.what());
  }
  return c;
}

// field words crossing the C ABI must be canonical M31 (< 2^31 - 1), as the verifier enforces for proof bytes
static void check_canonical(const uint32_t* w, uint64_t n, const char* what) {
  for (uint64_t i = 0; i < n; ++i)
    if (w[i] >= P31) throw LmnError(LMN_ERR_INVALID_ARGUMENT, std::string(what) + " is not a canonical M31 word");
}

// frees a freshly allocated handle when the op that fills it throws
struct ColGuard {
  lmn_col* c;
  explicit ColGuard(lmn_col* c_) : c(c_) {}
  ~ColGuard() {
    if (c) {
      lmn_dev_free(c->d);
      delete c;
    }
  }
  lmn_col* release() {
    lmn_col* r = c;
    c = nullptr;
    return r;
  }
};

lmn_col* Context::col_alloc(uint32_t ncols, uint32_t log_size, bool zero) {
  set_device();
  ColGuard c(new_col(ncols, log_size));
  if (zero) lmn_memset(c.c->d, 0, c.c->words() * 4, stream_);
  return c.release();
}
lmn_col* Context::col_from_cpu(const uint32_t* host, uint32_t ncols, uint32_t log_size) {
  set_device();
  ColGuard c(new_col(ncols, log_size));
  lmn_h2d(c.c->d, host, c.c->words() * 4, stream_);
  lmn_sync(stream_);  // the host buffer is borrowed only for the duration of the call
  return c.release();
}
void Context::col_to_cpu(const lmn_col* c, uint32_t* host) {
  set_device();
  lmn_d2h(host, c->d, c->words() * 4, stream_);
  lmn_sync(stream_);
}
void Context::col_free(lmn_col* c) {
  if (!c) return;
  if (c->view) {
    delete c;
    return;
  }
  set_device();
  lmn_sync(stream_);  // stream-ordered ops may still read it
  lmn_dev_free(c->d);
  delete c;
}
lmn_col* Context::col_view(const lmn_col* c, uint32_t first, uint32_t n) {
  if (n == 0 || first >= c->ncols || n > c->ncols - first) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: column range out of bounds");
  return new lmn_col{c->d + ((uint64_t)first << c->log_size), n, c->log_size, true};
}

void Context::col_bit_reverse(lmn_col* c) {
  set_device();
  launch_bit_reverse(c->d, 1ull << c->log_size, (int)c->ncols, (int)c->log_size, stream_);
}
void Context::col_precompute_twiddles(uint32_t log_size) {
  set_device();
  if (log_size > COL_MAX_LOG) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "precompute_twiddles: log size too large");
  ensure_twiddles((int)log_size);
}
void Context::col_interpolate(lmn_col* c) {
  set_device();
  if (c->log_size < 1) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "interpolate: log_size < 1");
  ensure_twiddles((int)c->log_size);
  const uint64_t n = 1ull << c->log_size;
  launch_ifft(c->d, n, c->d, n, (int)c->ncols, (int)c->log_size, itw((int)c->log_size), stream_);
}
lmn_col* Context::col_evaluate(const lmn_col* co, uint32_t log_domain) {
  set_device();
  if (log_domain < co->log_size || log_domain < 1) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "evaluate: domain smaller than the polynomial");
  lmn_col* out = new_col(co->ncols, log_domain);
  try {
    ensure_twiddles((int)log_domain);
    launch_fft(out->d, 1ull << log_domain, co->d, 1ull << co->log_size, (int)co->log_size, (int)co->ncols, (int)log_domain,
               tw((int)log_domain), stream_);
  } catch (...) {
    col_free(out);
    throw;
  }
  return out;
}
lmn_col* Context::col_evaluate_block(const lmn_col* co, uint32_t log_domain, uint32_t log_blocks, uint32_t block) {
  set_device();
  if (log_domain < co->log_size) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "evaluate_block: domain smaller than the polynomial");
  if (log_blocks < 1 || log_blocks > 3 || log_blocks >= log_domain || block >= (1u << log_blocks))
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad block specification");
  const uint32_t lb = log_domain - log_blocks;
  lmn_col* out = new_col(co->ncols, lb);
  try {
    ensure_twiddles((int)log_domain);
    launch_fft_block(out->d, 1ull << lb, co->d, 1ull << co->log_size, (int)co->log_size, (int)co->ncols, (int)log_domain,
                     (int)log_blocks, block, tw((int)log_domain), stream_);
  } catch (...) {
    col_free(out);
    throw;
  }
  return out;
}
lmn_col* Context::col_extend(const lmn_col* co, uint32_t log_size) {
  set_device();
  if (log_size < co->log_size) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "extend: target smaller than the polynomial");
  ColGuard out(new_col(co->ncols, log_size));
  launch_extend(co->d, 1ull << co->log_size, (int)co->log_size, out.c->d, 1ull << log_size, (int)log_size, (int)co->ncols, stream_);
  return out.release();
}
void Context::col_eval_at_point(const lmn_col* co, uint32_t column, const uint32_t pt[8], uint32_t out[4]) {
  check_canonical(pt, 8, "eval_at_point: point");
  if (column >= co->ncols) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "eval_at_point: column index out of range");
  arena_.reserve(8u << 20);
  begin_op();
  QPt p{{pt[0], pt[1], pt[2], pt[3]}, {pt[4], pt[5], pt[6], pt[7]}};
  std::vector<QM31> r = eval_at_points({{co->d + ((uint64_t)column << co->log_size), (int)co->log_size, 0}}, {p},
                                       (int)co->log_size);
  out[0] = r[0].a;
  out[1] = r[0].b;
  out[2] = r[0].c;
  out[3] = r[0].d;
}

// frees a tree handle (and its device slab) when the op that fills it throws
struct TreeGuard {
  lmn_tree* t;
  explicit TreeGuard(lmn_tree* t_) : t(t_) {}
  ~TreeGuard() {
    if (t) {
      if (t->slab) lmn_dev_free(t->slab);
      delete t;
    }
  }
  lmn_tree* release() {
    lmn_tree* r = t;
    t = nullptr;
    return r;
  }
};

lmn_tree* Context::col_commit(const lmn_col* const* cols, uint32_t n) {
  set_device();
  std::vector<ColRef> sorted;
  uint32_t max_log = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const lmn_col* c = cols[k];
    if (!c) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "commit: null column handle");
    for (uint32_t j = 0; j < c->ncols; ++j) sorted.push_back({c->d + ((uint64_t)j << c->log_size), (int)c->log_size, false});
    max_log = std::max(max_log, c->log_size);
  }
  if (sorted.empty()) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "commit: no columns");
  std::stable_sort(sorted.begin(), sorted.end(), [](auto& a, auto& b) { return a.log > b.log; });
  // the tree is hashed in the arena, then its layers move to a slab the handle owns
  arena_.reserve((16ull << max_log) * 4 + (4u << 20));
  begin_op();
  reset_event_log();
  DevMerkle m;
  build_merkle(m, sorted);
  TreeGuard tg(new lmn_tree{nullptr, {}, m.max_log, {}});
  lmn_tree* t = tg.t;
  try {
    t->slab = (uint32_t*)lmn_dev_malloc((16ull << m.max_log) * 4);
  } catch (const LmnError& e) {
    throw LmnError(LMN_ERR_OUT_OF_MEMORY, std::string("tree allocation failed: ") + e.what());
  }
  t->layers.assign(m.max_log + 1, nullptr);
  uint64_t off = 0;
  for (int l = 0; l <= m.max_log; ++l) {
    t->layers[l] = t->slab + off;
    lmn_d2d(t->layers[l], m.layers[l], (32ull << l), stream_);
    off += 8ull << l;
  }
  fetch_root_async(m);
  lmn_sync(stream_);
  m.finish_root();
  t->root = m.root;
  return tg.release();
}
void Context::tree_layer_to_cpu(const lmn_tree* t, uint32_t layer_log, uint8_t* out) {
  set_device();
  if ((int)layer_log > t->max_log) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tree: no such layer");
  lmn_d2h(out, t->layers[layer_log], 32ull << layer_log, stream_);
  lmn_sync(stream_);
}
void Context::tree_free(lmn_tree* t) {
  if (!t) return;
  set_device();
  lmn_sync(stream_);
  lmn_dev_free(t->slab);
  delete t;
}

void Context::col_accumulate(lmn_col* dst, const lmn_col* src) {
  set_device();
  if (dst->ncols != src->ncols || dst->log_size != src->log_size)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "accumulate: shapes differ");
  launch_secure_add(dst->d, src->d, dst->words(), stream_);
}

lmn_col* Context::col_accumulate_quotients(const lmn_col* const* cols, uint32_t n, const uint32_t* sample_col,
                                           const uint32_t* sample_point, const uint32_t* sample_values, uint32_t nsamples,
                                           const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4]) {
  set_device();
  if (n == 0 || !cols[0]) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "accumulate_quotients: no columns / null column handle");
  const uint32_t log_size = cols[0]->log_size;
  if (log_size < 2) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "accumulate_quotients: domain too small");
  std::vector<const uint32_t*> d_cols;
  for (uint32_t k = 0; k < n; ++k) {
    if (!cols[k] || cols[k]->log_size != log_size) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "accumulate_quotients: columns of one size only");
    for (uint32_t j = 0; j < cols[k]->ncols; ++j) d_cols.push_back(cols[k]->d + ((uint64_t)j << log_size));
  }
  const uint32_t ncols = (uint32_t)d_cols.size();
  check_canonical(points_xy, 8ull * npoints, "accumulate_quotients: point");
  check_canonical(sample_values, 4ull * nsamples, "accumulate_quotients: sample value");
  check_canonical(alpha, 4, "accumulate_quotients: alpha");
  std::vector<QPt> pts(npoints);
  for (uint32_t p = 0; p < npoints; ++p) {
    const uint32_t* w = points_xy + 8 * p;
    pts[p] = {{w[0], w[1], w[2], w[3]}, {w[4], w[5], w[6], w[7]}};
  }
  std::vector<std::vector<std::pair<int, QM31>>> smp(ncols);
  for (uint32_t i = 0; i < nsamples; ++i) {
    if (sample_col[i] >= ncols || sample_point[i] >= npoints) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad sample index");
    const uint32_t* v = sample_values + 4 * i;
    smp[sample_col[i]].push_back({(int)sample_point[i], QM31{v[0], v[1], v[2], v[3]}});
  }
  ensure_twiddles((int)log_size);
  arena_.reserve(8u << 20);
  begin_op();
  lmn_col* out = new_col(4, log_size);
  try {
    QuotientArgs a = make_quotient_args((int)log_size, d_cols, smp, pts, QM31{alpha[0], alpha[1], alpha[2], alpha[3]}, false);
    a.out = out->d;
    launch_quotients(a, stream_);
    lmn_sync(stream_);  // the (pointer, coefficient) table lives in the arena, which the next op resets
  } catch (...) {
    col_free(out);
    throw;
  }
  return out;
}

static void check_secure(const lmn_col* c, const char* what) {
  if (c->ncols != 4) throw LmnError(LMN_ERR_INVALID_ARGUMENT, std::string(what) + ": a secure column has 4 coordinate columns");
}
lmn_col* Context::col_fold_line(const lmn_col* src, const uint32_t alpha[4]) {
  check_canonical(alpha, 4, "fold_line: alpha");
  check_secure(src, "fold_line");
  if (src->log_size < 1) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "fold_line: nothing to fold");
  ensure_twiddles((int)src->log_size + 1);
  arena_.reserve(8u << 20);
  begin_op();
  std::vector<QM31> av{QM31{alpha[0], alpha[1], alpha[2], alpha[3]}};
  QM31* d_alpha = upload_vec(av);
  ColGuard out(new_col(4, src->log_size - 1));
  launch_fold_line(out.c->d, src->d, 1u << src->log_size, itwX_[src->log_size + 1], d_alpha, stream_);
  lmn_sync(stream_);  // alpha lives in the arena
  return out.release();
}
void Context::col_fold_circle_into_line(lmn_col* dst, const lmn_col* src, const uint32_t alpha[4]) {
  check_canonical(alpha, 4, "fold_circle_into_line: alpha");
  check_secure(src, "fold_circle_into_line");
  check_secure(dst, "fold_circle_into_line");
  if (src->log_size < 1 || dst->log_size + 1 != src->log_size)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "fold_circle_into_line: dst must be half the size of src");
  ensure_twiddles((int)src->log_size);
  arena_.reserve(8u << 20);
  begin_op();
  std::vector<QM31> av{QM31{alpha[0], alpha[1], alpha[2], alpha[3]}};
  QM31* d_alpha = upload_vec(av);
  launch_fold_circle_into_line(dst->d, src->d, 1u << src->log_size, itwY_[src->log_size], d_alpha, 1, stream_);
  lmn_sync(stream_);
}
lmn_col* Context::col_decompose(const lmn_col* f, uint32_t lambda_out[4]) {
  check_secure(f, "decompose");
  if (f->log_size < 1) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "decompose: a circle domain has at least two points");
  arena_.reserve(8u << 20);
  begin_op();
  QM31* d_lambda = (QM31*)arena_.alloc_bytes(sizeof(QM31));
  QM31* scratch = (QM31*)arena_.alloc_bytes((size_t)decompose_num_blocks((int)f->log_size) * sizeof(QM31));
  ColGuard gg(new_col(4, f->log_size));
  lmn_col* g = gg.c;
  launch_decompose(f->d, (int)f->log_size, g->d, d_lambda, scratch, stream_);
  const QM31* l = (const QM31*)stage_download(d_lambda, sizeof(QM31));
  lmn_sync(stream_);
  lambda_out[0] = l->a;
  lambda_out[1] = l->b;
  lambda_out[2] = l->c;
  lambda_out[3] = l->d;
  return gg.release();
}

// ---- the per-component stages on handles (the same launches Context::prove makes)
static const ComponentSpec* spec_or_throw(uint32_t kind, const char* what) {
  const ComponentSpec* sp = component_spec((int)kind);
  if (!sp) throw LmnError(LMN_ERR_INVALID_ARGUMENT, std::string(what) + ": unknown component kind");
  return sp;
}
static QM31 q_words(const uint32_t* w) { return QM31{w[0], w[1], w[2], w[3]}; }

lmn_col* Context::col_logup(uint32_t kind, const lmn_col* main, const lmn_col* pre, const uint32_t* elems,
                            uint32_t claimed_out[4]) {
  set_device();
  const ComponentSpec* sp = spec_or_throw(kind, "logup");
  if ((int)main->ncols != sp->n_cols) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "logup: wrong number of trace columns for this kind");
  if (main->log_size < 4 || main->log_size > 26) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "logup: trace log size must be 4..26");
  if (sp->n_pre && (!pre || (int)pre->ncols != sp->n_pre || pre->log_size != main->log_size))
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "logup: this kind needs its preprocessed columns (same size as the trace)");
  check_canonical(elems, 8ull * N_ELEMS, "logup: relation element");
  const uint64_t n = 1ull << main->log_size;
  const int nic = 4 * sp->n_rel;
  const int nb = logup_num_blocks((uint32_t)n);
  arena_.reserve(n * sizeof(QM31) + (size_t)nb * 16 + (size_t)logup_scan_num_blocks((int)main->log_size) * sizeof(QM31) + (4u << 20));
  begin_op();
  ColGuard out(new_col((uint32_t)nic, main->log_size));
  LogupArgs a{};
  a.k = sp->n_rel;
  for (int j = 0; j < sp->n_rel; ++j) {
    auto column = [&](int idx) -> const uint32_t* { return (sp->rel_pre[j] ? pre->d : main->d) + (uint64_t)idx * n; };
    a.val[j] = column(sp->rel_val[j]);
    a.id[j] = sp->rel_id[j] >= 0 ? column(sp->rel_id[j]) : nullptr;
    a.mult[j] = main->d + (uint64_t)sp->rel_mult[j] * n;
    a.neg[j] = sp->rel_neg[j];
    a.z[j] = q_words(elems + 8 * sp->rel_elems[j]);
    a.alpha[j] = q_words(elems + 8 * sp->rel_elems[j] + 4);
  }
  a.inter = out.c->d;
  a.last_tmp = (QM31*)arena_.alloc_bytes(n * sizeof(QM31));
  a.partials = arena_.alloc_words((size_t)nb * 4);
  a.n = (uint32_t)n;
  launch_logup_fracs(a, stream_);
  QM31* d_cs = (QM31*)arena_.alloc_bytes(2 * sizeof(QM31));
  QM31* bsums = (QM31*)arena_.alloc_bytes((size_t)logup_scan_num_blocks((int)main->log_size) * sizeof(QM31));
  launch_logup_scan(a.last_tmp, d_cs, (int)main->log_size, out.c->d + (uint64_t)(nic - 4) * n, bsums, stream_, true,
                    m_inv((uint32_t)(n % P31)));
  const QM31* cs = (const QM31*)stage_download(d_cs, 2 * sizeof(QM31));
  lmn_sync(stream_);
  claimed_out[0] = cs[0].a;
  claimed_out[1] = cs[0].b;
  claimed_out[2] = cs[0].c;
  claimed_out[3] = cs[0].d;
  return out.release();
}

void Context::col_composition(uint32_t kind, const lmn_col* main_lde, const lmn_col* inter_lde, const lmn_col* pre_lde,
                              const uint32_t* elems, const uint32_t claimed[4], const uint32_t* coeffs, uint32_t n_coeffs,
                              lmn_col* acc) {
  set_device();
  const ComponentSpec* sp = spec_or_throw(kind, "composition");
  const int e = (int)main_lde->log_size, ls = e - 1;
  if (ls < 4 || ls > 26) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "composition: evaluation domain log size must be 5..27");
  if ((int)main_lde->ncols != sp->n_cols || (int)inter_lde->ncols != 4 * sp->n_rel || (int)inter_lde->log_size != e)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "composition: trace / interaction columns do not match this kind");
  if (sp->n_pre && (!pre_lde || (int)pre_lde->ncols != sp->n_pre || (int)pre_lde->log_size != e))
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "composition: this kind needs its preprocessed columns on the evaluation domain");
  if ((int)n_coeffs != sp->n_local + sp->n_rel) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "composition: one coefficient per constraint");
  if (acc->ncols != 4 || (int)acc->log_size != e) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "composition: accumulator must be 4 x 2^(k+1)");
  check_canonical(elems, 8ull * N_ELEMS, "composition: relation element");
  check_canonical(claimed, 4, "composition: claimed sum");
  check_canonical(coeffs, 4ull * n_coeffs, "composition: coefficient");
  arena_.reserve(4u << 20);
  begin_op();
  const uint64_t E = 1ull << e;
  CompositionArgs a{};
  a.kind = sp->kind;
  a.log_size = ls;
  a.eval_log = e;
  a.main = main_lde->d;
  a.inter = inter_lde->d;
  a.row0 = 0;
  a.n_rows = (uint32_t)E;
  a.stride = E;
  a.prev_last = a.inter + (uint64_t)(4 * (sp->n_rel - 1)) * E;
  a.out = acc->d;
  a.accumulate = 1;
  a.z = q_words(elems + 8 * ELEMS_NODE);
  a.alpha = q_words(elems + 8 * ELEMS_NODE + 4);
  for (int j = 0; j < sp->n_rel; ++j)
    if (sp->rel_elems[j] != ELEMS_NODE) {
      a.z2 = q_words(elems + 8 * sp->rel_elems[j]);
      a.alpha2 = q_words(elems + 8 * sp->rel_elems[j] + 4);
    }
  a.pre = sp->n_pre >= 1 ? pre_lde->d : nullptr;
  a.pre2 = sp->n_pre >= 2 ? pre_lde->d + E : nullptr;
  const QM31 cl = q_words(claimed);
  std::vector<QM31> cs{cl, q_mul_m(cl, m_inv((uint32_t)((1ull << ls) % P31)))};
  a.claimed_shift = upload_vec(cs);
  for (uint32_t k = 0; k < n_coeffs; ++k) a.coeff[k] = q_words(coeffs + 4 * k);
  for (int b = 0; b < 2; ++b) {
    Pt p = domain_point(e, (uint32_t)b << ls);
    uint32_t x = p.x;
    for (int k = 0; k < ls - 1; ++k) x = m_sub(m_dbl(m_sqr(x)), 1u);
    a.zinv[b] = m_inv(x);
  }
  launch_composition(a, stream_);
  lmn_sync(stream_);  // [claimed, shift] lives in the arena, which the next op resets
}

}  // namespace lmn

// ------------------------------------------------------------------------------------ extern "C"
namespace {
template <typename F>
int guard2(lmn_ctx* ctx, F&& f) {
  return lmn::capi_guard(ctx, std::forward<F>(f));
}
}  // namespace

extern "C" {
int lmn_col_alloc(lmn_ctx* ctx, uint32_t ncols, uint32_t log_size, lmn_col** out) {
  if (!ctx || !out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] { *out = ctx->impl->col_alloc(ncols, log_size, true); });
}
int lmn_col_from_cpu(lmn_ctx* ctx, const uint32_t* host, uint32_t ncols, uint32_t log_size, lmn_col** out) {
  if (!ctx || !host || !out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] { *out = ctx->impl->col_from_cpu(host, ncols, log_size); });
}
int lmn_col_to_cpu(lmn_ctx* ctx, const lmn_col* col, uint32_t* host) {
  if (!ctx || !col || !host) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->col_to_cpu(col, host); });
}
void lmn_col_free(lmn_ctx* ctx, lmn_col* col) {
  if (ctx && col) guard2(ctx, [&] { ctx->impl->col_free(col); });
}
uint32_t lmn_col_ncols(const lmn_col* col) { return col ? col->ncols : 0; }
uint32_t lmn_col_log_size(const lmn_col* col) { return col ? col->log_size : 0; }
void* lmn_col_device_ptr(const lmn_col* col) { return col ? col->d : nullptr; }
int lmn_col_bit_reverse(lmn_ctx* ctx, lmn_col* col) {
  if (!ctx || !col) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->col_bit_reverse(col); });
}
int lmn_col_precompute_twiddles(lmn_ctx* ctx, uint32_t log_size) {
  if (!ctx) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->col_precompute_twiddles(log_size); });
}
int lmn_col_interpolate(lmn_ctx* ctx, lmn_col* c) {
  if (!ctx || !c) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->col_interpolate(c); });
}
int lmn_col_evaluate(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t log_domain, lmn_col** out) {
  if (!ctx || !coeffs || !out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] { *out = ctx->impl->col_evaluate(coeffs, log_domain); });
}
int lmn_col_evaluate_block(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t log_domain, uint32_t log_blocks, uint32_t block,
                           lmn_col** out) {
  if (!ctx || !coeffs || !out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] { *out = ctx->impl->col_evaluate_block(coeffs, log_domain, log_blocks, block); });
}
int lmn_col_extend(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t log_size, lmn_col** out) {
  if (!ctx || !coeffs || !out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] { *out = ctx->impl->col_extend(coeffs, log_size); });
}
int lmn_col_eval_at_point(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t column, const uint32_t point_xy[8],
                          uint32_t value_out[4]) {
  if (!ctx || !coeffs || !point_xy || !value_out) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->col_eval_at_point(coeffs, column, point_xy, value_out); });
}
int lmn_col_commit(lmn_ctx* ctx, const lmn_col* const* cols, uint32_t n, lmn_tree** tree_out) {
  if (!ctx || !cols || !tree_out) return LMN_ERR_INVALID_ARGUMENT;
  *tree_out = nullptr;
  return guard2(ctx, [&] { *tree_out = ctx->impl->col_commit(cols, n); });
}
int lmn_tree_root(lmn_ctx* ctx, const lmn_tree* tree, uint8_t root_out[32]) {
  if (!ctx || !tree || !root_out) return LMN_ERR_INVALID_ARGUMENT;
  memcpy(root_out, tree->root.w, 32);
  return LMN_OK;
}
uint32_t lmn_tree_log_size(const lmn_tree* tree) { return tree ? (uint32_t)tree->max_log : 0; }
int lmn_tree_layer_to_cpu(lmn_ctx* ctx, const lmn_tree* tree, uint32_t layer_log, uint8_t* hashes_out) {
  if (!ctx || !tree || !hashes_out) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->tree_layer_to_cpu(tree, layer_log, hashes_out); });
}
void lmn_tree_free(lmn_ctx* ctx, lmn_tree* tree) {
  if (ctx && tree) guard2(ctx, [&] { ctx->impl->tree_free(tree); });
}
int lmn_col_accumulate(lmn_ctx* ctx, lmn_col* dst, const lmn_col* src) {
  if (!ctx || !dst || !src) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->col_accumulate(dst, src); });
}
int lmn_col_accumulate_quotients(lmn_ctx* ctx, const lmn_col* const* cols, uint32_t n, const uint32_t* sample_col,
                                 const uint32_t* sample_point, const uint32_t* sample_values, uint32_t nsamples,
                                 const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4], lmn_col** out) {
  if (!ctx || !cols || !sample_col || !sample_point || !sample_values || !points_xy || !alpha || !out)
    return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] {
    *out = ctx->impl->col_accumulate_quotients(cols, n, sample_col, sample_point, sample_values, nsamples, points_xy, npoints, alpha);
  });
}
int lmn_col_fold_line(lmn_ctx* ctx, const lmn_col* src, const uint32_t alpha[4], lmn_col** out) {
  if (!ctx || !src || !alpha || !out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] { *out = ctx->impl->col_fold_line(src, alpha); });
}
int lmn_col_fold_circle_into_line(lmn_ctx* ctx, lmn_col* dst, const lmn_col* src, const uint32_t alpha[4]) {
  if (!ctx || !dst || !src || !alpha) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] { ctx->impl->col_fold_circle_into_line(dst, src, alpha); });
}
int lmn_col_view(lmn_ctx* ctx, const lmn_col* col, uint32_t first, uint32_t n, lmn_col** out) {
  if (!ctx || !col || !out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  return guard2(ctx, [&] { *out = ctx->impl->col_view(col, first, n); });
}
int lmn_col_logup(lmn_ctx* ctx, uint32_t kind, const lmn_col* main, const lmn_col* pre, const uint32_t* elems,
                  lmn_col** interaction_out, uint32_t claimed_sum_out[4]) {
  if (!ctx || !main || !elems || !interaction_out || !claimed_sum_out) return LMN_ERR_INVALID_ARGUMENT;
  *interaction_out = nullptr;
  return guard2(ctx, [&] { *interaction_out = ctx->impl->col_logup(kind, main, pre, elems, claimed_sum_out); });
}
int lmn_col_composition(lmn_ctx* ctx, uint32_t kind, const lmn_col* main_lde, const lmn_col* inter_lde,
                        const lmn_col* pre_lde, const uint32_t* elems, const uint32_t claimed_sum[4],
                        const uint32_t* coeffs, uint32_t n_coeffs, lmn_col* acc) {
  if (!ctx || !main_lde || !inter_lde || !elems || !claimed_sum || !coeffs || !acc) return LMN_ERR_INVALID_ARGUMENT;
  return guard2(ctx, [&] {
    ctx->impl->col_composition(kind, main_lde, inter_lde, pre_lde, elems, claimed_sum, coeffs, n_coeffs, acc);
  });
}
uint32_t lmn_kind_constraints(uint32_t kind) {
  const lmn::ComponentSpec* s = lmn::component_spec((int)kind);
  return s ? (uint32_t)(s->n_local + s->n_rel) : 0u;
}
uint32_t lmn_kind_constraint_layout(uint32_t kind, uint32_t protocol_flags, int32_t proto_index_out[16], int32_t sign_out[16]) {
  const lmn::ComponentSpec* s = lmn::component_spec((int)kind);
  if (!s || !proto_index_out || !sign_out || (protocol_flags & ~LMN_PV_ALL)) return 0u;
  const lmn::ConstraintLayout L = lmn::constraint_layout(*s, protocol_flags);
  for (int k = 0; k < 16; ++k) {
    proto_index_out[k] = k < L.n_kernel ? L.proto_index[k] : -1;
    sign_out[k] = k < L.n_kernel && L.neg[k] ? -1 : 1;
  }
  return (uint32_t)L.n_protocol;
}
uint32_t lmn_kind_relations(uint32_t kind) {
  const lmn::ComponentSpec* s = lmn::component_spec((int)kind);
  return s ? (uint32_t)s->n_rel : 0u;
}
int lmn_col_decompose(lmn_ctx* ctx, const lmn_col* f, lmn_col** g_out, uint32_t lambda_out[4]) {
  if (!ctx || !f || !g_out || !lambda_out) return LMN_ERR_INVALID_ARGUMENT;
  *g_out = nullptr;
  return guard2(ctx, [&] { *g_out = ctx->impl->col_decompose(f, lambda_out); });
}
}  // extern "C"
