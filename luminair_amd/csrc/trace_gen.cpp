// The step before the path: `process_trace` of the reference's operators on device-resident tensors
// (crates/graph/src/op/prim.rs), filling trace-table rows in HBM.  Host side of the lmn_trace_* entry points.
#include "prover_internal.h"

namespace lmn {

static TraceNode trace_node(const lmn_node_info& info) {
  auto m31 = [](int64_t v) { return (uint32_t)(((v % (int64_t)P31) + (int64_t)P31) % (int64_t)P31); };
  TraceNode nd{};
  nd.node_id = info.node_id;
  nd.lhs_id = info.input_ids[0];
  nd.rhs_id = info.input_ids[1];
  nd.lhs_mult = m31(info.input_mults[0]);
  nd.rhs_mult = m31(info.input_mults[1]);
  nd.out_mult = info.is_final_output ? 0u : m31(info.num_consumers);
  return nd;
}

// `LuminairSumReduce::process_trace` (prim.rs:1450-1565) on a contiguous (front, dim, back) device tensor
void Context::trace_reduce(bool is_max, const int32_t* input, uint64_t front, uint64_t dim, uint64_t back,
                           const lmn_node_info& info, uint32_t* rows, uint64_t row_offset, int32_t* out) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (front == 0 || dim == 0 || back == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (front * back * dim >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tensor too large");
  launch_trace_reduce(is_max, input, front, dim, back, trace_node(info), rows + row_offset * (is_max ? 15ull : 14ull), out,
                      stream_);  // stream-ordered with every later call on this context (lmn_prove, lmn_download)
}

// `process_trace` of one Add / Mul / Recip node on device tensors (prim.rs:967-1013, :1090-1139, :388-431)
static TraceView trace_view(const lmn_view* v, uint64_t n) {
  TraceView t{};
  if (!v) return t;
  if (v->ndim < 1 || v->ndim > 4) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: ndim must be 1..4");
  uint64_t prod = 1;
  t.ndim = v->ndim;
  for (uint32_t k = 0; k < v->ndim; ++k) {
    if (v->shape[k] == 0) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: empty dimension");
    t.shape[k] = v->shape[k];
    t.strides[k] = v->strides[k];
    prod *= v->shape[k];
  }
  if (v->offset < 0) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: negative offset");
  t.offset = v->offset;
  if (prod != n) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "view: shape does not match the element count");
  return t;
}

// `process_trace` of a Sin / Exp2 / Log2 node on a device tensor; fills the LUT multiplicity column too
void Context::trace_lut(uint32_t kind, const int32_t* input, const lmn_view* view, uint64_t n, const lmn_node_info& info,
                        const uint32_t* lut_col1, const lmn_range* ranges, uint32_t n_ranges, uint32_t* mult, uint32_t* rows,
                        uint64_t row_offset, int32_t* out) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (kind != LMN_KIND_SIN && kind != LMN_KIND_EXP2 && kind != LMN_KIND_LOG2)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: kind must be Sin, Exp2 or Log2");
  if (n == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (n >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "bad sizes");
  if (!ranges || n_ranges == 0 || n_ranges > (uint32_t)LUT_MAX_RANGES)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: 1..16 value ranges");
  LutRanges rg{};
  rg.n = (int)n_ranges;
  uint64_t base = 0;
  for (uint32_t k = 0; k < n_ranges; ++k) {
    // ascending, disjoint, inside the Fixed<12> range the LUT generator accepts (coalesce_ranges' output)
    if (ranges[k].hi < ranges[k].lo || ranges[k].lo <= -(1ll << 30) || ranges[k].hi >= (1ll << 30) ||
        (k > 0 && ranges[k].lo <= ranges[k - 1].hi))
      throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: ranges must be ascending, disjoint and inside (-2^30, 2^30)");
    rg.lo[k] = (int32_t)ranges[k].lo;
    rg.hi[k] = (int32_t)ranges[k].hi;
    rg.base[k] = (uint32_t)base;
    base += (uint64_t)(ranges[k].hi - ranges[k].lo + 1);
    if (base > (1ull << 26)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: LUT larger than 2^26 rows");
  }
  const TraceView tv = trace_view(view, n);
  // bad_flag_[1] is zero between calls; the kernel sets it when an input misses every range
  launch_trace_lut(input, tv, n, trace_node(info), lut_col1, rg, mult, rows + row_offset * 12ull, out, bad_flag_ + 1, stream_);
  uint32_t err = 0;
  lmn_d2h(&err, bad_flag_ + 1, 4, stream_);
  lmn_sync(stream_);
  if (err) {
    const uint32_t zero = 0u;
    lmn_h2d(bad_flag_ + 1, &zero, 4, stream_);
    lmn_sync(stream_);
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_lut: an input value lies outside the LUT's range");
  }
}

void Context::trace_elementwise(uint32_t kind, const int32_t* lhs, const lmn_view* lv, const int32_t* rhs,
                                const lmn_view* rv, uint64_t n, const lmn_node_info& info, uint32_t* rows,
                                uint64_t row_offset, int32_t* out, uint32_t* aux) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  const ComponentSpec* sp = component_spec((int)kind);
  const bool binary = kind == LMN_KIND_ADD || kind == LMN_KIND_MUL || kind == LMN_KIND_REM || kind == LMN_KIND_LESS_THAN;
  const bool unary = kind == LMN_KIND_RECIP || kind == LMN_KIND_SQRT || kind == LMN_KIND_CONTIGUOUS || kind == LMN_KIND_INPUTS;
  if (!sp || !(binary || unary))
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_elementwise: not an elementwise kind");
  if (binary && !rhs) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_elementwise: missing right operand");
  if (kind == LMN_KIND_LESS_THAN && !aux)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "trace_elementwise: LessThan needs the range-check multiplicity table");
  if (n == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (n >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tensor too large");
  const TraceNode nd = trace_node(info);
  const TraceView tlv = trace_view(lv, n), trv = trace_view(rv, n);
  launch_trace_elementwise((int)kind, lhs, tlv, rhs, trv, n, nd, rows + row_offset * (uint64_t)sp->n_cols, out, aux,
                           stream_);  // stream-ordered with every later call on this context
}

// `LuminairContiguous::process_trace` in the reference's own row rule (prim.rs:229-301): max(in_size, out_size) rows
void Context::trace_contiguous(const int32_t* input, uint64_t in_size, const lmn_view* view, uint64_t out_size,
                               const lmn_node_info& info, uint32_t* rows, uint64_t row_offset, int32_t* out) {
#ifndef LMN_EMU
  LMN_HIP_CHECK(hipSetDevice(device_));
#endif
  if (in_size == 0 || out_size == 0) throw LmnError(LMN_ERR_EMPTY_TRACE, "TraceError::EmptyTrace");
  if (in_size >= (1ull << 31) || out_size >= (1ull << 31)) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "tensor too large");
  TraceNode nd = trace_node(info);
  nd.phys_n = in_size;
  nd.out_n = out_size;
  const TraceView tv = trace_view(view, out_size);
  if (view)  // every element the view addresses must lie inside the buffer
    for (uint64_t corner = 0; corner < (1ull << view->ndim); ++corner) {
      int64_t off = view->offset;
      for (uint32_t k = 0; k < view->ndim; ++k)
        if (corner >> k & 1) off += (int64_t)(view->shape[k] - 1) * view->strides[k];
      if (off < 0 || (uint64_t)off >= in_size) throw LmnError(LMN_ERR_INVALID_ARGUMENT, "contiguous: the view leaves the input buffer");
    }
  else if (out_size > in_size)
    throw LmnError(LMN_ERR_INVALID_ARGUMENT, "contiguous: output larger than the input buffer without a view");
  const uint64_t n = std::max(in_size, out_size);
  launch_trace_elementwise(LMN_KIND_CONTIGUOUS, input, tv, nullptr, TraceView{}, n, nd, rows + row_offset * 11ull, out, nullptr,
                           stream_);
}

}  // namespace lmn
