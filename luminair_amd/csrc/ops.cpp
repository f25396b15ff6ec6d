// Context::op_*: the host-buffer convenience forms of the level-2 operations (include/luminair_hip.h lmn_op_*):
// upload, one kernel family, download.  The device-handle forms are in level2.cpp.
#include "prover_internal.h"

namespace lmn {

// ------------------------------------------------------------------------------------ level-2 ops
constexpr uint32_t OP_MAX_LOG = 26;  // largest column a level-2 op accepts (as lmn_prove: 2^26 rows)
static void check_op_log(uint32_t log_size, const char* what) {
  if (log_size > OP_MAX_LOG) throw LmnError(LMN_ERR_INVALID_ARGUMENT, std::string(what) + ": log size above 26");
}
void Context::set_device() {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
}
void Context::reset_event_log() { g_log(this)->reset(); }
// every op starts from an empty device arena AND an empty pinned staging buffer
void Context::begin_op() {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  arena_.reset();
  pin_off_ = 0;
}

void Context::op_interpolate(uint32_t* cols, uint32_t ncols, uint32_t log_size) {
  check_op_log(log_size, "interpolate");
  ensure_twiddles((int)log_size);
  size_t bytes = ((size_t)ncols << log_size) * 4;
  arena_.reserve(bytes + (1u << 20));
  begin_op();
  uint32_t* d = arena_.alloc_words((size_t)ncols << log_size);
  lmn_h2d(d, cols, bytes, stream_);
  launch_ifft(d, 1ull << log_size, d, 1ull << log_size, (int)ncols, (int)log_size, itw((int)log_size), stream_);
  lmn_d2h(cols, d, bytes, stream_);
  lmn_sync(stream_);
}

void Context::op_evaluate(const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                          uint32_t* out) {
  if (log_coeffs > log_domain) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "log_coeffs > log_domain");
  check_op_log(log_domain, "evaluate");
  ensure_twiddles((int)log_domain);
  size_t in_w = (size_t)ncols << log_coeffs, out_w = (size_t)ncols << log_domain;
  arena_.reserve((in_w + out_w) * 4 + (1u << 20));
  begin_op();
  uint32_t* d_in = arena_.alloc_words(in_w);
  uint32_t* d_out = arena_.alloc_words(out_w);
  lmn_h2d(d_in, coeffs, in_w * 4, stream_);
  launch_fft(d_out, 1ull << log_domain, d_in, 1ull << log_coeffs, (int)log_coeffs, (int)ncols, (int)log_domain,
             tw((int)log_domain), stream_);
  lmn_d2h(out, d_out, out_w * 4, stream_);
  lmn_sync(stream_);
}

void Context::op_evaluate_block(const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                                uint32_t log_blocks, uint32_t block, uint32_t* out) {
  if (log_coeffs > log_domain) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "log_coeffs > log_domain");
  if (log_blocks < 1 || log_blocks > 3 || log_blocks >= log_domain || block >= (1u << log_blocks))
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad block specification");
  check_op_log(log_domain, "evaluate_block");
  ensure_twiddles((int)log_domain);
  const uint32_t lb = log_domain - log_blocks;
  size_t in_w = (size_t)ncols << log_coeffs, out_w = (size_t)ncols << lb;
  arena_.reserve((in_w + out_w) * 4 + (1u << 20));
  begin_op();
  uint32_t* d_in = arena_.alloc_words(in_w);
  uint32_t* d_out = arena_.alloc_words(out_w);
  lmn_h2d(d_in, coeffs, in_w * 4, stream_);
  launch_fft_block(d_out, 1ull << lb, d_in, 1ull << log_coeffs, (int)log_coeffs, (int)ncols, (int)log_domain,
                   (int)log_blocks, block, tw((int)log_domain), stream_);
  lmn_d2h(out, d_out, out_w * 4, stream_);
  lmn_sync(stream_);
}

void Context::op_merkle_root(const uint32_t* const* cols, const uint32_t* log_sizes, uint32_t ncols, uint8_t root[32]) {
  size_t words = 0;
  uint32_t max_log = 0;
  for (uint32_t c = 0; c < ncols; ++c) {
    check_op_log(log_sizes[c], "merkle_root");
    words += 1ull << log_sizes[c];
    max_log = std::max(max_log, log_sizes[c]);
  }
  arena_.reserve((words + (16ull << max_log)) * 4 + (1u << 20));
  begin_op();
  std::vector<ColRef> sorted;
  for (uint32_t c = 0; c < ncols; ++c) {
    uint32_t* d = arena_.alloc_words(1ull << log_sizes[c]);
    lmn_h2d(d, cols[c], (4ull << log_sizes[c]), stream_);
    sorted.push_back({d, (int)log_sizes[c], false});
  }
  std::stable_sort(sorted.begin(), sorted.end(), [](auto& a, auto& b) { return a.log > b.log; });
  g_log(this)->reset();
  DevMerkle m;
  build_merkle(m, sorted);
  fetch_root_async(m);
  lmn_sync(stream_);
  m.finish_root();
  memcpy(root, m.root.w, 32);
}

void Context::op_eval_at_point(const uint32_t* coeffs, uint32_t log_size, const uint32_t pt[8], uint32_t out[4]) {
  check_op_log(log_size, "eval_at_point");
  arena_.reserve((4ull << log_size) + (8u << 20));
  begin_op();
  uint32_t* d = arena_.alloc_words(1ull << log_size);
  lmn_h2d(d, coeffs, 4ull << log_size, stream_);
  QPt p{{pt[0], pt[1], pt[2], pt[3]}, {pt[4], pt[5], pt[6], pt[7]}};
  std::vector<QM31> r = eval_at_points({{d, (int)log_size, 0}}, {p}, (int)log_size);
  out[0] = r[0].a;
  out[1] = r[0].b;
  out[2] = r[0].c;
  out[3] = r[0].d;
}

// QuotientOps::accumulate_quotients for the columns of one LDE size
void Context::op_accumulate_quotients(uint32_t log_size, const uint32_t* const* cols, uint32_t ncols,
                                      const uint32_t* sample_col, const uint32_t* sample_point, const uint32_t* sample_values,
                                      uint32_t nsamples, const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4],
                                      uint32_t* out) {
  if (log_size < 2 || log_size > 26) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad log_size");
  ensure_twiddles((int)log_size);
  const uint64_t L = 1ull << log_size;
  arena_.reserve(((uint64_t)ncols + 4) * L * 4 + (8u << 20));
  begin_op();
  std::vector<const uint32_t*> d_cols(ncols);
  for (uint32_t c = 0; c < ncols; ++c) {
    uint32_t* d = arena_.alloc_words(L);
    lmn_h2d(d, cols[c], L * 4, stream_);
    d_cols[c] = d;
  }
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
  QuotientArgs a = make_quotient_args((int)log_size, d_cols, smp, pts, QM31{alpha[0], alpha[1], alpha[2], alpha[3]});
  launch_quotients(a, stream_);
  lmn_d2h(out, a.out, 16 * L, stream_);
  lmn_sync(stream_);
}

// FriOps::fold_line (circle == 0) / FriOps::fold_circle_into_line (circle == 1; dst is accumulated:
// dst = dst * alpha^2 + fold(src), as the FRI commit loop does)
void Context::op_fold(int circle, uint32_t* dst, const uint32_t* src, uint32_t log_src, const uint32_t alpha[4]) {
  if (log_src < 1 || log_src > OP_MAX_LOG) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad log size");
  ensure_twiddles((int)log_src + 1);
  const uint64_t L = 1ull << log_src;
  arena_.reserve(6 * L * 4 + (1u << 20));
  begin_op();
  uint32_t* d_src = arena_.alloc_words(4 * L);
  uint32_t* d_dst = arena_.alloc_words(2 * L);
  lmn_h2d(d_src, src, 16 * L, stream_);
  std::vector<QM31> av{QM31{alpha[0], alpha[1], alpha[2], alpha[3]}};
  QM31* d_alpha = upload_vec(av);
  if (circle) {
    lmn_h2d(d_dst, dst, 8 * L, stream_);
    launch_fold_circle_into_line(d_dst, d_src, (uint32_t)L, itwY_[log_src], d_alpha, 1, stream_);
  } else {
    launch_fold_line(d_dst, d_src, (uint32_t)L, itwX_[log_src + 1], d_alpha, stream_);
  }
  lmn_d2h(dst, d_dst, 8 * L, stream_);
  lmn_sync(stream_);
}

// tiled FFT vs one-layer-per-launch kernels on pseudo-random data (device-side differential check)
void Context::op_fft_selftest(uint32_t log_size, uint32_t ncols) {
  check_op_log(log_size, "fft_selftest");
  ensure_twiddles((int)log_size + 1);
  size_t w = (size_t)ncols << log_size;
  arena_.reserve(w * 32 + (1u << 20));
  begin_op();
  std::vector<uint32_t> h(w);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    v = (uint32_t)(st >> 33) % P31;
  }
  uint32_t* a = arena_.alloc_words(w);
  uint32_t* b = arena_.alloc_words(w);
  uint64_t n = 1ull << log_size;
  for (int inverse = 0; inverse < 2; ++inverse) {
    lmn_h2d(a, h.data(), w * 4, stream_);
    lmn_h2d(b, h.data(), w * 4, stream_);
    if (inverse) {
      launch_ifft(a, n, a, n, (int)ncols, (int)log_size, itw((int)log_size), stream_);
      launch_fft_simple(b, n, (int)ncols, (int)log_size, itw((int)log_size), true, stream_);
    } else {
      launch_fft(a, n, a, n, (int)log_size, (int)ncols, (int)log_size, tw((int)log_size), stream_);
      launch_fft_simple(b, n, (int)ncols, (int)log_size, tw((int)log_size), false, stream_);
    }
    std::vector<uint32_t> ra(w), rb(w);
    lmn_d2h(ra.data(), a, w * 4, stream_);
    lmn_d2h(rb.data(), b, w * 4, stream_);
    lmn_sync(stream_);
    for (size_t i = 0; i < w; ++i)
      if (ra[i] != rb[i])
        throw LmnError(LMN_ERR_INTERNAL, std::string("fft selftest mismatch (inverse=") + std::to_string(inverse) +
                                             ") at word " + std::to_string(i));
  }
  // the fused interpolate + extend path of the commitments against the separate transforms
  if (fft_interp_extend_supported((int)log_size) && (int)log_size + 1 <= tw_max_log_) {
    begin_op();
    uint32_t* ev = arena_.alloc_words(w);
    uint32_t* co = arena_.alloc_words(w);
    uint32_t* lde = arena_.alloc_words(2 * w);
    uint32_t* co2 = arena_.alloc_words(w);
    uint32_t* lde2 = arena_.alloc_words(2 * w);
    lmn_h2d(ev, h.data(), w * 4, stream_);
    launch_interp_extend(co, n, ev, n, lde, 2 * n, (int)ncols, (int)log_size, itw((int)log_size), tw((int)log_size + 1), stream_);
    lmn_d2d(co2, ev, w * 4, stream_);
    launch_fft_simple(co2, n, (int)ncols, (int)log_size, itw((int)log_size), true, stream_);
    launch_extend(co2, n, (int)log_size, lde2, 2 * n, (int)log_size + 1, (int)ncols, stream_);
    launch_fft_simple(lde2, 2 * n, (int)ncols, (int)log_size + 1, tw((int)log_size + 1), false, stream_);
    std::vector<uint32_t> ra(3 * w), rb(3 * w);
    lmn_d2h(ra.data(), co, w * 4, stream_);
    lmn_d2h(ra.data() + w, lde, 2 * w * 4, stream_);
    lmn_d2h(rb.data(), co2, w * 4, stream_);
    lmn_d2h(rb.data() + w, lde2, 2 * w * 4, stream_);
    lmn_sync(stream_);
    for (size_t i = 0; i < 3 * w; ++i)
      if (ra[i] != rb[i])
        throw LmnError(LMN_ERR_INTERNAL, std::string("fft selftest mismatch (interpolate + extend) at word ") + std::to_string(i));
  }
}

}  // namespace lmn
