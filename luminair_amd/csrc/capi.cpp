// extern "C" boundary (include/luminair_hip.h): the entry points a Rust `extern "C"` block would
// bind in place of /root/reference/crates/prover/src/prover.rs:28-31.
#include "../../include/luminair_hip.h"

#include "capi_internal.h"

#include <algorithm>
#include <cmath>

using lmn::Context;

namespace {
// for the entry points that need no context
template <class F>
static int host_guard(F&& f) {
  try {
    f();
    return LMN_OK;
  } catch (const LmnError& e) {
    return e.code;
  } catch (const std::bad_alloc&) {
    return LMN_ERR_OUT_OF_MEMORY;
  } catch (...) {
    return LMN_ERR_INTERNAL;
  }
}

template <typename F>
int guard(lmn_ctx* ctx, F&& f) {
  return lmn::capi_guard(ctx, std::forward<F>(f));
}
thread_local std::string g_create_error;
}  // namespace

extern "C" {

const char* lmn_strerror(int code) {
  switch (code) {
    case LMN_OK: return "ok";
    case LMN_ERR_EMPTY_TRACE: return "TraceError(EmptyTrace)";
    case LMN_ERR_MAIN_TRACE: return "MainTraceEvalGenError";
    case LMN_ERR_INTERACTION_TRACE: return "InteractionTraceEvalGenError";
    case LMN_ERR_CONSTRAINTS: return "ProverError(ConstraintsNotSatisfied)";
    case LMN_ERR_SERIALIZATION: return "SerializationError";
    case LMN_ERR_INVALID_ARGUMENT: return "invalid argument";
    case LMN_ERR_OUT_OF_MEMORY: return "out of memory";
    case LMN_ERR_NO_DEVICE: return "no HIP device (no CPU fallback exists)";
    case LMN_ERR_VERIFICATION: return "StwoVerifierError";
    case LMN_ERR_INVALID_LOGUP: return "InvalidLogUp";
    default: return "internal error";
  }
}

uint32_t lmn_abi_version(void) { return LMN_API_VERSION; }

const char* lmn_last_error(const lmn_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

void lmn_default_config(lmn_config* cfg) {
  if (!cfg) return;
  cfg->pow_bits = 5;
  cfg->log_blowup = 1;
  cfg->log_last_layer = 0;
  cfg->n_queries = 3;
  cfg->fp_scale = 12;
  cfg->protocol_variant = LMN_VARIANT_KAT;
}

uint32_t lmn_kind_columns(uint32_t kind) {
  const lmn::ComponentSpec* s = lmn::component_spec((int)kind);
  return s ? (uint32_t)s->n_cols : 0u;
}

int lmn_kind_padding_row(uint32_t kind, uint32_t* out) {
  const lmn::ComponentSpec* s = lmn::component_spec((int)kind);
  if (!s || !out) return LMN_ERR_INVALID_ARGUMENT;
  for (int c = 0; c < s->n_cols; ++c) out[c] = 0u;
  if (s->is_last_col >= 0) out[s->is_last_col] = 1u;
  for (int k = 0; k < s->n_pad; ++k) out[s->pad_col[k]] = s->pad_val[k];
  return LMN_OK;
}

int lmn_ctx_create(int device, const lmn_config* cfg, lmn_ctx** out) {
  if (!out) return LMN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  lmn_config c;
  lmn_default_config(&c);
  if (cfg) c = *cfg;
  lmn_ctx* ctx = new lmn_ctx{nullptr, {}};
  int rc = guard(ctx, [&] { ctx->impl = new Context(device, c); });
  if (rc != LMN_OK) {
    g_create_error = ctx->last_error;
    delete ctx;
    return rc;
  }
  *out = ctx;
  return LMN_OK;
}

// error text written by the CALLING thread: under the context's own lock, like the worker's guard()
static int set_error(lmn_ctx* ctx, int code, const char* msg) {
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  ctx->last_error = msg;
  return code;
}

static void async_stop(lmn_ctx* ctx) {
  lmn_async* a = ctx->async;
  if (!a) return;
  {
    std::unique_lock<std::mutex> lk(a->m);
    a->cv.wait(lk, [&] { return a->state != lmn_async::SUBMITTED; });  // let a running proof finish
    a->state = lmn_async::QUIT;
  }
  a->cv.notify_all();
  if (a->worker.joinable()) a->worker.join();
  delete a;
  ctx->async = nullptr;
}

void lmn_ctx_destroy(lmn_ctx* ctx) {
  if (!ctx) return;
  async_stop(ctx);
  delete ctx->impl;
  delete ctx;
}

// ---- asynchronous form of lmn_prove: the reference's callers are single-threaded (SURVEY.md §8b "threading"); with
// submit / wait one thread keeps a proof in flight on each of several contexts
int lmn_prove_submit(lmn_ctx* ctx, const lmn_table* tables, size_t n_tables, const lmn_settings* settings) {
  if (!ctx) return LMN_ERR_INVALID_ARGUMENT;
  std::unique_lock<std::mutex> create_lock(ctx->async_mu);
  if (!ctx->async) {
    lmn_async* a = new lmn_async();
    ctx->async = a;
    a->worker = std::thread([ctx, a] {
      for (;;) {
        std::unique_lock<std::mutex> lk(a->m);
        a->cv.wait(lk, [&] { return a->state == lmn_async::SUBMITTED || a->state == lmn_async::QUIT; });
        if (a->state == lmn_async::QUIT) return;
        const lmn_table* t = a->tables;
        const size_t n = a->n_tables;
        const lmn_settings* st = a->settings;
        lk.unlock();
        std::vector<uint8_t> bytes;
        const int rc = guard(ctx, [&] { bytes = ctx->impl->prove(t, n, st); });
        lk.lock();
        a->rc = rc;
        a->proof.swap(bytes);
        a->state = lmn_async::DONE;
        lk.unlock();
        a->cv.notify_all();
      }
    });
  }
  lmn_async* a = ctx->async;
  create_lock.unlock();
  bool busy = false;
  {
    std::lock_guard<std::mutex> lk(a->m);
    if (a->state != lmn_async::IDLE) {
      busy = true;
    } else {
      a->tables = tables;
      a->n_tables = n_tables;
      a->settings = settings;
      a->state = lmn_async::SUBMITTED;
    }
  }
  if (busy) return set_error(ctx, LMN_ERR_INVALID_ARGUMENT,
                             "lmn_prove_submit: a submitted proof has not been collected with lmn_prove_wait");
  a->cv.notify_all();
  return LMN_OK;
}

int lmn_prove_wait(lmn_ctx* ctx, uint8_t** proof_bincode, size_t* proof_len) {
  if (!ctx || !proof_bincode || !proof_len) return LMN_ERR_INVALID_ARGUMENT;
  *proof_bincode = nullptr;
  *proof_len = 0;
  lmn_async* a;
  {
    std::lock_guard<std::mutex> create_lock(ctx->async_mu);
    a = ctx->async;
  }
  if (!a) return set_error(ctx, LMN_ERR_INVALID_ARGUMENT, "lmn_prove_wait: nothing was submitted");
  std::unique_lock<std::mutex> lk(a->m);
  if (a->state == lmn_async::IDLE) {
    lk.unlock();
    return set_error(ctx, LMN_ERR_INVALID_ARGUMENT, "lmn_prove_wait: nothing was submitted");
  }
  a->cv.wait(lk, [&] { return a->state == lmn_async::DONE; });
  const int rc = a->rc;
  if (rc == LMN_OK) {
    uint8_t* p = (uint8_t*)malloc(a->proof.size() ? a->proof.size() : 1);
    if (!p) {
      a->proof.clear();
      a->state = lmn_async::IDLE;
      return LMN_ERR_OUT_OF_MEMORY;
    }
    memcpy(p, a->proof.data(), a->proof.size());
    *proof_bincode = p;
    *proof_len = a->proof.size();
  }
  a->proof.clear();
  a->state = lmn_async::IDLE;
  return rc;
}

int lmn_prove(lmn_ctx* ctx, const lmn_table* tables, size_t n_tables, const lmn_settings* settings,
              uint8_t** proof_bincode, size_t* proof_len) {
  if (!ctx || !proof_bincode || !proof_len) return LMN_ERR_INVALID_ARGUMENT;
  *proof_bincode = nullptr;
  *proof_len = 0;
  return guard(ctx, [&] {
    std::vector<uint8_t> bytes = ctx->impl->prove(tables, n_tables, settings);
    uint8_t* p = (uint8_t*)malloc(bytes.size());
    if (!p) throw std::bad_alloc();
    memcpy(p, bytes.data(), bytes.size());
    *proof_bincode = p;
    *proof_len = bytes.size();
  });
}

void lmn_free(void* p) { free(p); }

int lmn_set_profiling(lmn_ctx* ctx, int enabled) {
  if (!ctx) return LMN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  ctx->impl->profiling = enabled != 0;
  return LMN_OK;
}

int lmn_get_timings(const lmn_ctx* ctx, lmn_timings* out) {
  if (!ctx || !out) return LMN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> lock(const_cast<lmn_ctx*>(ctx)->mu);
  *out = ctx->impl->timings;
  return LMN_OK;
}

int lmn_verify_with_config(const uint8_t* proof_bincode, size_t proof_len, const lmn_settings* settings,
                           const lmn_config* expected) {
  if (!proof_bincode || !expected) return LMN_ERR_INVALID_ARGUMENT;
  lmn_ctx tmp{nullptr, {}};
  int rc = guard(&tmp, [&] { lmn::verify_proof(proof_bincode, proof_len, *expected, settings); });
  g_create_error = tmp.last_error;
  return rc;
}

int lmn_verify_diagnose(const uint8_t* proof_bincode, size_t proof_len, const lmn_settings* settings,
                        const lmn_config* expected, lmn_verify_report* report) {
  if (!proof_bincode || !expected || !report) return LMN_ERR_INVALID_ARGUMENT;
  memset(report, 0, sizeof *report);
  lmn_ctx tmp{nullptr, {}};
  int rc = guard(&tmp, [&] { lmn::verify_proof(proof_bincode, proof_len, *expected, settings, report); });
  if (rc != LMN_OK && !report->first_failure[0]) snprintf(report->first_failure, sizeof report->first_failure, "%s", tmp.last_error.c_str());
  g_create_error = tmp.last_error;
  return rc;
}

int lmn_verify(const uint8_t* proof_bincode, size_t proof_len, const lmn_settings* settings, uint32_t protocol_variant) {
  lmn_config c;
  lmn_default_config(&c);  // PcsConfig::default(), as the reference verifier hard-codes it
  c.protocol_variant = protocol_variant;
  return lmn_verify_with_config(proof_bincode, proof_len, settings, &c);
}

// page-locked host memory for host-resident trace rows (no context: usable by every context of the process)
int lmn_host_alloc(size_t bytes, void** host_out) {
  if (!host_out || bytes == 0) return LMN_ERR_INVALID_ARGUMENT;
  *host_out = nullptr;
  return host_guard([&] { *host_out = lmn_host_alloc_pinned(bytes); });
}
void lmn_host_free(void* host) {
  if (host) lmn_host_free_pinned(host);
}
int lmn_host_register(void* host, size_t bytes) {
  if (!host || bytes == 0) return LMN_ERR_INVALID_ARGUMENT;
  return host_guard([&] { lmn_host_register_range(host, bytes); });
}
int lmn_host_unregister(void* host) {
  if (!host) return LMN_ERR_INVALID_ARGUMENT;
  return host_guard([&] { lmn_host_unregister_range(host); });
}

int lmn_upload(lmn_ctx* ctx, const void* host, size_t bytes, void** device_out) {
  if (!ctx || !host || !device_out) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { *device_out = ctx->impl->upload(host, bytes); });
}

void lmn_device_free(lmn_ctx* ctx, void* p) {
  if (ctx && p) ctx->impl->device_free(p);
}

int lmn_op_interpolate(lmn_ctx* ctx, uint32_t* cols, uint32_t ncols, uint32_t log_size) {
  if (!ctx || !cols || ncols == 0 || log_size < 1) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_interpolate(cols, ncols, log_size); });
}

int lmn_op_evaluate(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                    uint32_t* evals_out) {
  if (!ctx || !coeffs || !evals_out || ncols == 0 || log_domain < 1) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_evaluate(coeffs, ncols, log_coeffs, log_domain, evals_out); });
}

int lmn_op_merkle_root(lmn_ctx* ctx, const uint32_t* const* cols, const uint32_t* log_sizes, uint32_t ncols,
                       uint8_t root_out[32]) {
  if (!ctx || !root_out || (ncols && (!cols || !log_sizes))) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_merkle_root(cols, log_sizes, ncols, root_out); });
}

int lmn_op_eval_at_point(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t log_size, const uint32_t point_xy[8],
                         uint32_t value_out[4]) {
  if (!ctx || !coeffs || !point_xy || !value_out) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_eval_at_point(coeffs, log_size, point_xy, value_out); });
}

int lmn_op_fft_selftest(lmn_ctx* ctx, uint32_t log_size, uint32_t ncols) {
  if (!ctx || log_size < 1 || ncols == 0) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_fft_selftest(log_size, ncols); });
}

}  // extern "C"

int lmn_op_accumulate_quotients(lmn_ctx* ctx, uint32_t log_size, const uint32_t* const* cols, uint32_t ncols,
                                const uint32_t* sample_col, const uint32_t* sample_point, const uint32_t* sample_values,
                                uint32_t nsamples, const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4],
                                uint32_t* out) {
  if (!ctx || !cols || !sample_col || !sample_point || !sample_values || !points_xy || !alpha || !out)
    return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    ctx->impl->op_accumulate_quotients(log_size, cols, ncols, sample_col, sample_point, sample_values, nsamples, points_xy,
                                       npoints, alpha, out);
  });
}

int lmn_op_fold_line(lmn_ctx* ctx, const uint32_t* src, uint32_t log_src, const uint32_t alpha[4], uint32_t* dst) {
  if (!ctx || !src || !alpha || !dst) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_fold(0, dst, src, log_src, alpha); });
}

int lmn_op_fold_circle_into_line(lmn_ctx* ctx, uint32_t* dst, const uint32_t* src, uint32_t log_src, const uint32_t alpha[4]) {
  if (!ctx || !src || !alpha || !dst) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_fold(1, dst, src, log_src, alpha); });
}

int lmn_op_grind(const uint8_t digest[32], uint32_t pow_bits, uint32_t protocol_variant, uint64_t* nonce_out) {
  if (!digest || !nonce_out || pow_bits > 40 || (protocol_variant & ~LMN_PV_ALL)) return LMN_ERR_INVALID_ARGUMENT;
  lmn::Channel ch(protocol_variant);
  lmn::Hash32 d;
  memcpy(d.w, digest, 32);
  ch.set_digest(d);
  *nonce_out = ch.grind(pow_bits);
  return LMN_OK;
}

int lmn_device_alloc(lmn_ctx* ctx, size_t bytes, void** device_out) {
  if (!ctx || !device_out) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { *device_out = ctx->impl->device_alloc(bytes); });
}

int lmn_download(lmn_ctx* ctx, const void* device, void* host, size_t bytes) {
  if (!ctx || !device || !host) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->download(device, host, bytes); });
}

int lmn_trace_elementwise(lmn_ctx* ctx, uint32_t kind, const int32_t* lhs_dev, const int32_t* rhs_dev, uint64_t n,
                          const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !lhs_dev || !info || !rows_dev || kind == LMN_KIND_LESS_THAN) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->trace_elementwise(kind, lhs_dev, nullptr, rhs_dev, nullptr, n, *info, rows_dev, row_offset, out_dev); });
}

int lmn_trace_sum_reduce(lmn_ctx* ctx, const int32_t* input_dev, uint64_t front, uint64_t dim, uint64_t back,
                         const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !input_dev || !info || !rows_dev) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->trace_reduce(false, input_dev, front, dim, back, *info, rows_dev, row_offset, out_dev); });
}

int lmn_trace_elementwise_v(lmn_ctx* ctx, uint32_t kind, const int32_t* lhs_dev, const lmn_view* lhs_view,
                            const int32_t* rhs_dev, const lmn_view* rhs_view, uint64_t n, const lmn_node_info* info,
                            uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !lhs_dev || !info || !rows_dev || kind == LMN_KIND_LESS_THAN) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    ctx->impl->trace_elementwise(kind, lhs_dev, lhs_view, rhs_dev, rhs_view, n, *info, rows_dev, row_offset, out_dev);
  });
}

int lmn_trace_contiguous(lmn_ctx* ctx, const int32_t* input_dev, uint64_t in_size, const lmn_view* view, uint64_t out_size,
                         const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !input_dev || !info || !rows_dev) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->trace_contiguous(input_dev, in_size, view, out_size, *info, rows_dev, row_offset, out_dev); });
}

int lmn_trace_lut(lmn_ctx* ctx, uint32_t kind, const int32_t* input_dev, const lmn_view* view, uint64_t n,
                  const lmn_node_info* info, const uint32_t* lut_col1_dev, int32_t lo, uint32_t lut_len,
                  uint32_t* mult_dev, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !input_dev || !info || !lut_col1_dev || !mult_dev || !rows_dev || lut_len == 0) return LMN_ERR_INVALID_ARGUMENT;
  const lmn_range one{(int64_t)lo, (int64_t)lo + (int64_t)lut_len - 1};
  return guard(ctx, [&] {
    ctx->impl->trace_lut(kind, input_dev, view, n, *info, lut_col1_dev, &one, 1, mult_dev, rows_dev, row_offset, out_dev);
  });
}

int lmn_trace_lut_ranges(lmn_ctx* ctx, uint32_t kind, const int32_t* input_dev, const lmn_view* view, uint64_t n,
                         const lmn_node_info* info, const uint32_t* lut_col1_dev, const lmn_range* ranges, uint32_t n_ranges,
                         uint32_t* mult_dev, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !input_dev || !info || !lut_col1_dev || !mult_dev || !rows_dev) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    ctx->impl->trace_lut(kind, input_dev, view, n, *info, lut_col1_dev, ranges, n_ranges, mult_dev, rows_dev, row_offset, out_dev);
  });
}

int lmn_trace_less_than(lmn_ctx* ctx, const int32_t* lhs_dev, const lmn_view* lhs_view, const int32_t* rhs_dev,
                        const lmn_view* rhs_view, uint64_t n, const lmn_node_info* info, uint32_t* range_check_mult_dev,
                        uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !lhs_dev || !rhs_dev || !info || !range_check_mult_dev || !rows_dev) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    ctx->impl->trace_elementwise(LMN_KIND_LESS_THAN, lhs_dev, lhs_view, rhs_dev, rhs_view, n, *info, rows_dev, row_offset,
                                 out_dev, range_check_mult_dev);
  });
}

int lmn_trace_max_reduce(lmn_ctx* ctx, const int32_t* input_dev, uint64_t front, uint64_t dim, uint64_t back,
                         const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev) {
  if (!ctx || !input_dev || !info || !rows_dev) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->trace_reduce(true, input_dev, front, dim, back, *info, rows_dev, row_offset, out_dev); });
}

namespace lmn {
void rccl_unique_id(uint8_t* out);
}
int lmn_ctx_set_shard(lmn_ctx* ctx, uint32_t rank, uint32_t world, uint32_t fri_min_log, const lmn_collective* coll) {
  if (!ctx) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] {
    // validate first: a rejected call leaves the context's current sharding (and its RCCL transport) untouched
    lmn::Context::check_shard_args(rank, world, fri_min_log, coll);
    ctx->impl->clear_shard();  // releases a previous built-in RCCL transport, if any
    ctx->impl->set_shard(rank, world, fri_min_log, coll);
  });
}
int lmn_rccl_unique_id(uint8_t id_out[LMN_RCCL_ID_BYTES]) {
  if (!id_out) return LMN_ERR_INVALID_ARGUMENT;
  lmn_ctx tmp{nullptr, {}};
  int rc = guard(&tmp, [&] { lmn::rccl_unique_id(id_out); });
  g_create_error = tmp.last_error;
  return rc;
}
int lmn_ctx_set_shard_rccl(lmn_ctx* ctx, uint32_t rank, uint32_t world, uint32_t fri_min_log,
                           const uint8_t id[LMN_RCCL_ID_BYTES]) {
  if (!ctx || !id) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->set_shard_rccl(rank, world, fri_min_log, id); });
}
int lmn_ctx_clear_shard(lmn_ctx* ctx) {
  if (!ctx) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->clear_shard(); });
}

// ---- LUT columns from the reference's layouts (preprocessed.rs:34-46, 351-383, 434-466, 517-549)
static int lut_values(const lmn_range* ranges, uint32_t n, std::vector<int64_t>& vals) {
  if (!ranges || n == 0) return LMN_ERR_INVALID_ARGUMENT;
  uint64_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (ranges[i].hi < ranges[i].lo || ranges[i].lo <= -(1ll << 30) || ranges[i].hi >= (1ll << 30)) return LMN_ERR_INVALID_ARGUMENT;
    total += (uint64_t)(ranges[i].hi - ranges[i].lo + 1);
    if (total > (1ull << 26)) return LMN_ERR_INVALID_ARGUMENT;
  }
  vals.reserve(total);
  for (uint32_t i = 0; i < n; ++i)
    for (int64_t v = ranges[i].lo; v <= ranges[i].hi; ++v) vals.push_back(v);
  std::sort(vals.begin(), vals.end());
  vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
  return LMN_OK;
}
int lmn_lut_log_size(const lmn_range* ranges, uint32_t n_ranges, uint32_t* log_size_out) {
  if (!log_size_out) return LMN_ERR_INVALID_ARGUMENT;
  // LookupLayout::new counts the values of the (already coalesced) ranges: value_count, preprocessed.rs:117-119
  uint64_t count = 0;
  if (!ranges || n_ranges == 0) return LMN_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < n_ranges; ++i) {
    // same bounds as lut_values(): Fixed<12> values inside (-2^30, 2^30), so hi - lo + 1 cannot overflow
    if (ranges[i].hi < ranges[i].lo || ranges[i].lo <= -(1ll << 30) || ranges[i].hi >= (1ll << 30)) return LMN_ERR_INVALID_ARGUMENT;
    count += (uint64_t)(ranges[i].hi - ranges[i].lo + 1);
    if (count > (1ull << 26)) return LMN_ERR_INVALID_ARGUMENT;
  }
  // calculate_log_size: ceil(count / 16) rounded up to a power of two, times 16 (LOG_N_LANES = 4)
  uint64_t packs = (count + 15) >> 4, p2 = 1;
  uint32_t lg = 0;
  while (p2 < packs) {
    p2 <<= 1;
    ++lg;
  }
  *log_size_out = lg + 4;
  return LMN_OK;
}
int lmn_lut_from_ranges(uint32_t lut_kind, const lmn_range* ranges, uint32_t n_ranges, uint32_t log_size, uint32_t* col0_out,
                        uint32_t* col1_out) {
  return lmn_lut_from_ranges_r(lut_kind, ranges, n_ranges, log_size, LMN_ROUND_HALF_AWAY, col0_out, col1_out);
}

int lmn_lut_from_ranges_r(uint32_t lut_kind, const lmn_range* ranges, uint32_t n_ranges, uint32_t log_size, uint32_t rounding,
                          uint32_t* col0_out, uint32_t* col1_out) {
  if (!col0_out || !col1_out || lut_kind > LMN_LUT_LOG2 || log_size > 26 || rounding > LMN_ROUND_FLOOR)
    return LMN_ERR_INVALID_ARGUMENT;
  std::vector<int64_t> vals;
  int rc = lut_values(ranges, n_ranges, vals);
  if (rc != LMN_OK) return rc;
  const uint64_t n = 1ull << log_size;
  if (vals.size() > n) return LMN_ERR_INVALID_ARGUMENT;
  const double scale = 4096.0;
  auto to_m31 = [](int64_t v) { return (uint32_t)(v >= 0 ? v : (int64_t)lmn::P31 + v); };  // Fixed::to_m31
  for (uint64_t i = 0; i < n; ++i) col0_out[i] = col1_out[i] = 0u;
  for (size_t i = 0; i < vals.size(); ++i) {
    const double x = (double)vals[i] / scale;
    double y;
    if (lut_kind == LMN_LUT_SIN) {
      y = std::sin(x);
    } else if (lut_kind == LMN_LUT_EXP2) {
      y = std::exp2(x);
    } else {
      // Divergence from the reference, on purpose: Log2PreProcessed::gen_column (preprocessed.rs:517-549) feeds
      // x <= 0 to f64::log2 and stores the saturated cast of NaN / -inf; such a row can never be looked up by a
      // valid trace (log2 of a non-positive input has no fixed-point value), so a layout that contains one is
      // rejected here instead of committing to a meaningless LUT row.
      if (vals[i] <= 0) return LMN_ERR_INVALID_ARGUMENT;
      y = std::log2(x);
    }
    const double ys = y * scale;
    const double r = rounding == LMN_ROUND_HALF_AWAY   ? std::round(ys)
                     : rounding == LMN_ROUND_HALF_EVEN ? std::nearbyint(ys)   // default FE_TONEAREST = ties to even
                     : rounding == LMN_ROUND_TRUNC     ? std::trunc(ys)
                                                       : std::floor(ys);
    if (!(std::fabs(r) < (double)(1ll << 30))) return LMN_ERR_INVALID_ARGUMENT;
    col0_out[i] = to_m31(vals[i]);
    col1_out[i] = to_m31((int64_t)r);
  }
  return LMN_OK;
}

int lmn_device_copy(lmn_ctx* ctx, void* device_dst, const void* device_src, size_t bytes) {
  if (!ctx || !device_dst || !device_src) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->device_copy(device_dst, device_src, bytes); });
}

int lmn_upload_to(lmn_ctx* ctx, const void* host, size_t bytes, void* device_dst) {
  if (!ctx || !host || !device_dst) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->upload_to(host, bytes, device_dst); });
}

int lmn_op_evaluate_block(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                          uint32_t log_blocks, uint32_t block, uint32_t* evals_out) {
  if (!ctx || !coeffs || !evals_out || ncols == 0) return LMN_ERR_INVALID_ARGUMENT;
  return guard(ctx, [&] { ctx->impl->op_evaluate_block(coeffs, ncols, log_coeffs, log_domain, log_blocks, block, evals_out); });
}
