// gfx950 kernels, part 1: the AoS -> SoA transpose of `write_trace`, device-side trace
// generation (`process_trace`, section 8f-3) and the level-2 column ops (bit reversal, decompose).  One wavefront = 64 lanes;
// global accesses are laid out so that consecutive lanes touch consecutive 4-byte words of a column.
#include "kernels_common.h"

namespace lmn {

// =============================================================================================
// a3  AoS -> SoA transpose with padding rows (is_last_idx = 1, everything else 0)
// =============================================================================================
// Rows [blk_row0, blk_row0 + blk_rows) of the padded table are produced (the whole table, or one rank's row block of a
// sharded proof); row r of column c lands at cols[c * out_stride + (r - blk_row0)].
// ROWS rows per workgroup: 256 (a lane has `ncols` independent loads in flight and every column gets a 1 KB run) wherever
// the row block allows, 64 for smaller blocks.
template <int ROWS>
LMN_KERNEL k_transpose_pad(const uint32_t* __restrict__ rows, uint64_t n_rows, int ncols, uint64_t size,
                           uint32_t* __restrict__ cols, PadRow pad, uint32_t* __restrict__ bad_flag, uint32_t magic,
                           uint64_t out_stride, uint64_t blk_row0, uint32_t bad_value) {
  LMN_DYN_SMEM(uint32_t, tile);  // ROWS x stride
  // odd row stride: the column-major read below walks rows at that stride, and an even one (16 words for the 15 columns of
  // Add) maps the rows of a column onto 2 of the 32 LDS banks
  const int stride = ncols | 1;
  const uint64_t row0 = blk_row0 + (uint64_t)blockIdx.x * ROWS;
  const int total = ROWS * ncols;
  constexpr int BATCH = 8;   // loads issued before the first of them is used
  for (int k0 = threadIdx.x; k0 < total; k0 += BATCH * TPB) {
    uint32_t v[BATCH];
    int rr[BATCH], cc[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int k = k0 + j * TPB;
      // k / ncols by the precomputed reciprocal (exact for k < 2^16): a runtime integer division is ~30 VALU ops
      const int r = magic ? (int)(((uint64_t)(uint32_t)k * magic) >> 32) : k, c = k - r * ncols;  // magic 0: one column
      rr[j] = r;
      cc[j] = c;
      const uint64_t gr = row0 + r;
      v[j] = 0u;
      if (k < total) v[j] = gr < n_rows ? rows[gr * (uint64_t)ncols + c] : pad.v[c];
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      if (k0 + j * TPB >= total) break;
      if (v[j] >= P31) *bad_flag = bad_value;  // the boundary takes raw u32 words: reject non-canonical M31 values
      tile[rr[j] * stride + cc[j]] = v[j];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < total; k += TPB) {
    const int c = k / ROWS, r = k - c * ROWS;
    if (row0 + r < size) cols[(uint64_t)c * out_stride + (row0 - blk_row0) + r] = tile[r * stride + c];
  }
}

void launch_transpose_pad_rows(const uint32_t* rows, uint64_t n_rows, int ncols, int log_size, uint32_t* cols,
                               uint64_t out_stride, uint64_t blk_row0, uint64_t blk_rows, const PadRow& pad, uint32_t* bad_flag,
                               lmn_stream_t s, uint32_t bad_value) {
  if (LMN_ABLATED(64u)) return;   // (experiment build: the whole cost of the transpose launch = the upper bound of fusing it away)
  uint64_t size = 1ull << log_size;
  const bool big = blk_row0 % 256 == 0 && blk_rows % 256 == 0;
  const unsigned tr_rows = big ? 256u : 64u;
  if (blk_row0 % tr_rows || blk_row0 + blk_rows > size) throw LmnError(-100, "transpose: bad row block");
  unsigned grid = cdiv(blk_rows, tr_rows);
  size_t smem = (size_t)tr_rows * (ncols + 1) * 4;
  if (ncols > 32 || ncols < 1) throw LmnError(-100, "transpose: bad column count");
  const uint32_t magic = ncols == 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)ncols - 1) / (uint64_t)ncols);  // ceil(2^32 / ncols)
  // rows beyond the block are cut off by treating its end as the table's size
  if (big)
    LMN_LAUNCH(k_transpose_pad<256>, dim3(grid), dim3(TPB), smem, s, rows, n_rows, ncols, blk_row0 + blk_rows, cols, pad, bad_flag,
               magic, out_stride, blk_row0, bad_value);
  else
    LMN_LAUNCH(k_transpose_pad<64>, dim3(grid), dim3(TPB), smem, s, rows, n_rows, ncols, blk_row0 + blk_rows, cols, pad, bad_flag,
               magic, out_stride, blk_row0, bad_value);
}
void launch_transpose_pad(const uint32_t* rows, uint64_t n_rows, int ncols, int log_size, uint32_t* cols,
                          const PadRow& pad, uint32_t* bad_flag, lmn_stream_t s) {
  launch_transpose_pad_rows(rows, n_rows, ncols, log_size, cols, 1ull << log_size, 0, 1ull << log_size, pad, bad_flag, s);
}

// =============================================================================================
// gen_trace for Add / Mul / Recip nodes (crates/graph/src/op/prim.rs:967-1013, :1090-1139, :388-431):
// one lane per tensor element computes the fixed-point op and its row; the block stages its rows in LDS
// and writes them out as one contiguous, coalesced run of words.
// =============================================================================================
LMN_HD uint32_t fixed_to_m31(int64_t v) { return v >= 0 ? (uint32_t)v : (uint32_t)((int64_t)P31 + v); }

LMN_D uint64_t view_offset(const TraceView& v, uint64_t r) {
  if (v.ndim == 0) return r;
  int64_t off = v.offset;
  for (int k = (int)v.ndim - 1; k >= 0; --k) {
    const uint64_t d = v.shape[k];
    off += (int64_t)(r % d) * v.strides[k];
    r /= d;
  }
  return (uint64_t)off;
}

LMN_HD constexpr int trace_ncols(int kind) {
  return kind == 0 ? 15 : kind == 1 ? 16 : kind == 2 ? 13 : kind == 7 ? 13 : kind == 8 ? 16 : kind == 13 ? 22
                                                                                       : kind == 16 ? 11 : 7;
}
// floor(sqrt(v)) for v < 2^44, exact (double sqrt + one correction step each way)
LMN_D int64_t isqrt_u64(int64_t v) {
  int64_t r = (int64_t)sqrt((double)v);
  while (r * r > v) --r;
  while ((r + 1) * (r + 1) <= v) ++r;
  return r;
}

template <int KIND>
LMN_KERNEL k_trace_elementwise(const int32_t* __restrict__ lhs, TraceView lv, const int32_t* __restrict__ rhs,
                               TraceView rv, uint64_t n, TraceNode nd, uint32_t* __restrict__ rows,
                               int32_t* __restrict__ out, uint32_t* __restrict__ aux) {
  constexpr int NC = trace_ncols(KIND);
  constexpr int ST = NC | 1;  // odd LDS row stride: conflict-free column writes
  LMN_SHARED uint32_t tile[TPB * ST];
  const uint64_t row0 = (uint64_t)blockIdx.x * TPB;
  const uint64_t r = row0 + threadIdx.x;
  if (r < n) {
    uint32_t* t = tile + threadIdx.x * ST;
    const bool ref_contig = KIND == 16 && nd.phys_n != 0;
    const int64_t a = lhs[view_offset(lv, ref_contig ? r % nd.out_n : r)];
    const uint32_t idx = (uint32_t)r, last = r + 1 == (ref_contig ? nd.phys_n : n) ? 1u : 0u;
    if (ref_contig) {
      // LuminairContiguous::process_trace as the reference writes it (prim.rs:253-296): row idx pairs the idx-th
      // element of the input BUFFER (zero past its end) with the idx-th element of the OUTPUT (the view; past the
      // output's end the index expression wraps), is_last_idx marks the buffer's last element.  Every buffer
      // element is consumed exactly once, so slices and permutations of the input balance.
      t[0] = nd.node_id; t[1] = nd.lhs_id; t[2] = idx; t[3] = last;
      t[4] = nd.node_id; t[5] = nd.lhs_id; t[6] = idx + 1u;
      t[7] = r < nd.phys_n ? fixed_to_m31((int64_t)lhs[r]) : 0u;
      t[8] = fixed_to_m31(a); t[9] = nd.lhs_mult; t[10] = nd.out_mult;
      if (out && r < nd.out_n) out[r] = (int32_t)a;
    } else if (KIND == 16 || KIND == 7) {
      // Contiguous (prim.rs:229-301): out = input.  Sqrt (prim.rs:573-660): out = floor(sqrt(input * scale)),
      // rem = input * scale - out^2 (natural identity; numerair's form is unpinned)
      t[0] = nd.node_id; t[1] = nd.lhs_id; t[2] = idx; t[3] = last;
      t[4] = nd.node_id; t[5] = nd.lhs_id; t[6] = idx + 1u;
      t[7] = fixed_to_m31(a);
      if (KIND == 16) {
        t[8] = fixed_to_m31(a); t[9] = nd.lhs_mult; t[10] = nd.out_mult;
        if (out) out[r] = (int32_t)a;
      } else {
        const int64_t o = isqrt_u64(a * 4096ll);
        t[8] = fixed_to_m31(o); t[9] = fixed_to_m31(a * 4096ll - o * o); t[10] = 4096u;
        t[11] = nd.lhs_mult; t[12] = nd.out_mult;
        if (out) out[r] = (int32_t)o;
      }
    } else if (KIND == 8 || KIND == 13) {
      const int64_t b = rhs[view_offset(rv, r)];
      t[0] = nd.node_id; t[1] = nd.lhs_id; t[2] = nd.rhs_id; t[3] = idx; t[4] = last;
      t[5] = nd.node_id; t[6] = nd.lhs_id; t[7] = nd.rhs_id; t[8] = idx + 1u;
      t[9] = fixed_to_m31(a); t[10] = fixed_to_m31(b);
      if (KIND == 8) {
        // Rem (prim.rs:1323-1421), operands > 0: lhs = rhs * quotient + rem; the out relation carries rem
        const int64_t quo = a / b, rem = a % b;
        t[11] = fixed_to_m31(rem); t[12] = fixed_to_m31(quo);
        t[13] = nd.lhs_mult; t[14] = nd.rhs_mult; t[15] = nd.out_mult;
        if (out) out[r] = (int32_t)rem;
      } else {
        // LessThan (prim.rs:1203-1295): out = 1.0 iff lhs < rhs; diff = rhs - lhs (+ P with borrow) in four
        // range-checked 8-bit limbs; aux = the RangeCheckLookup multiplicity column (256 entries)
        const bool lt = a < b;
        const int64_t diff = b - a + (lt ? 0 : (int64_t)P31);
        t[11] = lt ? 4096u : 0u;
        t[12] = (uint32_t)(diff % (int64_t)P31);
        t[13] = lt ? 0u : 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t limb = (uint32_t)(diff >> (8 * k)) & 0xFFu;
          t[14 + k] = limb;
          atomicAdd(&aux[limb], 1u);
        }
        t[18] = nd.lhs_mult; t[19] = nd.rhs_mult; t[20] = nd.out_mult; t[21] = 1u;
        if (out) out[r] = lt ? 4096 : 0;
      }
    } else if (KIND == 15) {
      // CopyToStwo / Inputs (prim.rs:52-88): node, idx, is_last, next_node, next_idx, val, multiplicity
      t[0] = nd.node_id; t[1] = idx; t[2] = last; t[3] = nd.node_id; t[4] = idx + 1u;
      t[5] = fixed_to_m31(a); t[6] = nd.out_mult;
      if (out) out[r] = (int32_t)a;
    } else if (KIND == 2) {
      // node, input, idx, is_last, next_node, next_input, next_idx, input, out, rem, scale, in_mult, out_mult
      const int64_t sc2 = 4096ll * 4096ll;
      const int64_t o = sc2 / a, rem = sc2 - a * o;
      t[0] = nd.node_id; t[1] = nd.lhs_id; t[2] = idx; t[3] = last;
      t[4] = nd.node_id; t[5] = nd.lhs_id; t[6] = idx + 1u;
      t[7] = fixed_to_m31(a); t[8] = fixed_to_m31(o); t[9] = fixed_to_m31(rem); t[10] = 4096u;
      t[11] = nd.lhs_mult; t[12] = nd.out_mult;
      if (out) out[r] = (int32_t)o;
    } else {
      const int64_t b = rhs[view_offset(rv, r)];
      t[0] = nd.node_id; t[1] = nd.lhs_id; t[2] = nd.rhs_id; t[3] = idx; t[4] = last;
      t[5] = nd.node_id; t[6] = nd.lhs_id; t[7] = nd.rhs_id; t[8] = idx + 1u;
      t[9] = fixed_to_m31(a); t[10] = fixed_to_m31(b);
      int64_t o;
      if (KIND == 0) {
        o = a + b;
        t[11] = fixed_to_m31(o);
        t[12] = nd.lhs_mult; t[13] = nd.rhs_mult; t[14] = nd.out_mult;
      } else {
        const int64_t prod = a * b;
        o = prod >> 12;  // floor
        t[11] = fixed_to_m31(o);
        t[12] = (uint32_t)(prod & 4095);
        t[13] = nd.lhs_mult; t[14] = nd.rhs_mult; t[15] = nd.out_mult;
      }
      if (out) out[r] = (int32_t)o;
    }
  }
  __syncthreads();
  const uint64_t rows_here = n - row0 < (uint64_t)TPB ? n - row0 : (uint64_t)TPB;
  const uint32_t words = (uint32_t)rows_here * NC;
  uint32_t* dst = rows + row0 * NC;
  for (uint32_t w = threadIdx.x; w < words; w += TPB) dst[w] = tile[(w / NC) * ST + (w % NC)];
}

// SumReduce rows (prim.rs:1486-1510, 1536-1561): row r = (i*back + j)*dim + k holds input[i, k, j], the
// running sum before and after it, and the output on the group's last step.  One lane per row; the prefix
// inside a group comes from a block-wide segmented scan (plus one cooperative carry-in for the group that straddles
// the block's first row), rows leave through LDS.
// MAX: MaxReduce rows (prim.rs:1591-1734): the running maximum starts at the group's first element, is_max marks
// the rows whose input becomes the new maximum (strict comparison).
template <bool MAX>
LMN_KERNEL k_trace_reduce(const int32_t* __restrict__ input, uint64_t dim, uint64_t back, uint64_t n_rows,
                          uint64_t n_out, TraceNode nd, uint32_t* __restrict__ rows, int32_t* __restrict__ out) {
  constexpr int NC = MAX ? 15 : 14, ST = MAX ? 17 : 15;
  LMN_SHARED uint32_t tile[TPB * ST];
  LMN_SHARED int64_t scan[TPB];   // inclusive segmented scan of the block's inputs (segment = reduction group)
  LMN_SHARED uint32_t head[TPB];  // distance (in lanes) back to the segment's first lane inside this block
  const uint64_t row0 = (uint64_t)blockIdx.x * TPB;
  const uint64_t r = row0 + threadIdx.x;
  const bool on = r < n_rows;
  const uint64_t g = on ? r / dim : 0, k = on ? r % dim : 0;  // output index (i*back + j), reduction step
  const uint64_t i = g / back, j = g % back;
  const int32_t* p = input + i * dim * back + j;
  const int64_t v = on ? (int64_t)p[k * back] : 0;
  auto op = [](int64_t a, int64_t b) { return MAX ? (a > b ? a : b) : a + b; };
  // carry-in of the group that straddles the block's first row: steps [0, k0) of that group, reduced cooperatively
  // (the block's first lane has k = k0); every other group starts inside the block
  const uint64_t k0 = row0 % dim;
  {
    const uint64_t g0 = row0 / dim, i0 = g0 / back, j0 = g0 % back;
    const int32_t* p0 = input + i0 * dim * back + j0;
    int64_t part = MAX ? INT64_MIN : 0;
    for (uint64_t kk = threadIdx.x; kk < k0; kk += TPB) part = op(part, (int64_t)p0[kk * back]);
    scan[threadIdx.x] = part;
    __syncthreads();
    for (int st = TPB / 2; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) scan[threadIdx.x] = op(scan[threadIdx.x], scan[threadIdx.x + st]);
      __syncthreads();
    }
  }
  const int64_t carry = scan[0];
  __syncthreads();
  // Hillis-Steele segmented inclusive scan: lane t combines with lane t - d only while that lane is still inside
  // the same group (head[t] = lanes back to the group's first lane in this block)
  const uint32_t dist = (uint32_t)(k < (uint64_t)threadIdx.x ? k : threadIdx.x);
  scan[threadIdx.x] = v;
  head[threadIdx.x] = dist;
  __syncthreads();
  for (uint32_t d = 1; d < TPB; d <<= 1) {
    int64_t add = 0;
    const bool take = d <= dist;
    if (take) add = scan[threadIdx.x - d];
    __syncthreads();
    if (take) scan[threadIdx.x] = op(scan[threadIdx.x], add);
    __syncthreads();
  }
  if (on) {
    // exclusive value: everything of the group before step k (inside the block, plus the carry for the straddling group)
    const bool first_seg = k == k0 + threadIdx.x;  // this lane's group began before the block
    int64_t acc;
    if (k == 0) {
      acc = MAX ? v : 0;
    } else {
      const bool has_prev = dist > 0;
      const int64_t inside = has_prev ? scan[threadIdx.x - 1] : (MAX ? INT64_MIN : 0);
      acc = first_seg ? (has_prev ? op(carry, inside) : carry) : inside;
    }
    const int64_t next = op(acc, v);
    const bool last_step = k + 1 == dim;
    uint32_t* t = tile + threadIdx.x * ST;
    t[0] = nd.node_id; t[1] = nd.lhs_id; t[2] = (uint32_t)g; t[3] = g + 1 == n_out ? 1u : 0u;
    t[4] = nd.node_id; t[5] = nd.lhs_id; t[6] = (uint32_t)g + 1u;
    t[7] = fixed_to_m31(v); t[8] = last_step ? fixed_to_m31(next) : 0u;
    t[9] = fixed_to_m31(acc); t[10] = fixed_to_m31(next); t[11] = last_step ? 1u : 0u;
    if (MAX) {
      t[12] = v > acc ? 1u : 0u; t[13] = nd.lhs_mult; t[14] = last_step ? nd.out_mult : 0u;
    } else {
      t[12] = nd.lhs_mult; t[13] = last_step ? nd.out_mult : 0u;
    }
    if (last_step && out) out[g] = (int32_t)next;
  }
  __syncthreads();
  const uint64_t rows_here = n_rows - row0 < (uint64_t)TPB ? n_rows - row0 : (uint64_t)TPB;
  const uint32_t words = (uint32_t)rows_here * NC;
  uint32_t* dst = rows + row0 * NC;
  for (uint32_t w = threadIdx.x; w < words; w += TPB) dst[w] = tile[(w / NC) * ST + (w % NC)];
}

void launch_trace_reduce(bool is_max, const int32_t* input, uint64_t front, uint64_t dim, uint64_t back,
                         const TraceNode& nd, uint32_t* rows, int32_t* out, lmn_stream_t s) {
  const uint64_t n_out = front * back, n_rows = n_out * dim;
  if (is_max)
    LMN_LAUNCH(k_trace_reduce<true>, dim3(cdiv(n_rows, TPB)), dim3(TPB), 0, s, input, dim, back, n_rows, n_out, nd, rows, out);
  else
    LMN_LAUNCH(k_trace_reduce<false>, dim3(cdiv(n_rows, TPB)), dim3(TPB), 0, s, input, dim, back, n_rows, n_out, nd, rows, out);
}

void launch_trace_elementwise(int kind, const int32_t* lhs, const TraceView& lv, const int32_t* rhs, const TraceView& rv,
                              uint64_t n, const TraceNode& nd, uint32_t* rows, int32_t* out, uint32_t* aux,
                              lmn_stream_t s) {
  dim3 g(cdiv(n, TPB)), b(TPB);
  switch (kind) {
    case 0: LMN_LAUNCH(k_trace_elementwise<0>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    case 1: LMN_LAUNCH(k_trace_elementwise<1>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    case 2: LMN_LAUNCH(k_trace_elementwise<2>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    case 7: LMN_LAUNCH(k_trace_elementwise<7>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    case 8: LMN_LAUNCH(k_trace_elementwise<8>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    case 13: LMN_LAUNCH(k_trace_elementwise<13>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    case 15: LMN_LAUNCH(k_trace_elementwise<15>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    case 16: LMN_LAUNCH(k_trace_elementwise<16>, g, b, 0, s, lhs, lv, rhs, rv, n, nd, rows, out, aux); break;
    default: throw LmnError(-100, "trace_elementwise: unsupported kind");
  }
}

// Sin / Exp2 / Log2 rows (sin/table.rs: node, input, idx, is_last, next_node, next_input, next_idx, input, out,
// input_mult, out_mult, lookup_mult) with out read from the LUT's output column, plus the LUT multiplicities.
LMN_KERNEL k_trace_lut(const int32_t* __restrict__ input, TraceView view, uint64_t n, TraceNode nd,
                       const uint32_t* __restrict__ lut1, LutRanges rg, uint32_t* __restrict__ mult,
                       uint32_t* __restrict__ rows, int32_t* __restrict__ out, uint32_t* __restrict__ err_flag) {
  constexpr int NC = 12, ST = 13;
  LMN_SHARED uint32_t tile[TPB * ST];
  const uint64_t row0 = (uint64_t)blockIdx.x * TPB;
  const uint64_t r = row0 + threadIdx.x;
  if (r < n) {
    const int64_t a = input[view_offset(view, r)];
    // LookupLayout::find_index: the range that holds `a` (few ranges: a linear scan of block-uniform bounds)
    int64_t li = -1;
    for (int k = 0; k < rg.n; ++k)
      if (a >= (int64_t)rg.lo[k] && a <= (int64_t)rg.hi[k]) li = (int64_t)rg.base[k] + (a - (int64_t)rg.lo[k]);
    uint32_t ow = 0u;
    if (li < 0) {
      *err_flag = 1u;
    } else {
      ow = lut1[li];
      atomicAdd(&mult[li], 1u);
    }
    uint32_t* t = tile + threadIdx.x * ST;
    t[0] = nd.node_id; t[1] = nd.lhs_id; t[2] = (uint32_t)r; t[3] = r + 1 == n ? 1u : 0u;
    t[4] = nd.node_id; t[5] = nd.lhs_id; t[6] = (uint32_t)r + 1u;
    t[7] = fixed_to_m31(a); t[8] = ow; t[9] = nd.lhs_mult; t[10] = nd.out_mult; t[11] = 1u;
    if (out) out[r] = ow > (P31 >> 1) ? (int32_t)ow - (int32_t)P31 : (int32_t)ow;
  }
  __syncthreads();
  const uint64_t rows_here = n - row0 < (uint64_t)TPB ? n - row0 : (uint64_t)TPB;
  const uint32_t words = (uint32_t)rows_here * NC;
  uint32_t* dst = rows + row0 * NC;
  for (uint32_t w = threadIdx.x; w < words; w += TPB) dst[w] = tile[(w / NC) * ST + (w % NC)];
}

void launch_trace_lut(const int32_t* input, const TraceView& view, uint64_t n, const TraceNode& nd,
                      const uint32_t* lut_col1, const LutRanges& ranges, uint32_t* mult, uint32_t* rows,
                      int32_t* out, uint32_t* err_flag, lmn_stream_t s) {
  LMN_LAUNCH(k_trace_lut, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, input, view, n, nd, lut_col1, ranges, mult, rows, out,
             err_flag);
}

// =============================================================================================
// level-2 column ops: bit reversal, FriOps::decompose
// =============================================================================================
LMN_KERNEL k_bit_reverse(uint32_t* __restrict__ data, uint64_t col_stride, int log_n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1ull << log_n)) return;
  const uint64_t j = log_n == 0 ? 0 : (uint64_t)(__brev((uint32_t)i) >> (32 - log_n));
  if (i >= j) return;  // each unordered pair is swapped once, by its smaller index
  uint32_t* col = data + (uint64_t)blockIdx.y * col_stride;
  const uint32_t a = col[i], b = col[j];
  col[i] = b;
  col[j] = a;
}
void launch_bit_reverse(uint32_t* data, uint64_t col_stride, int ncols, int log_n, lmn_stream_t s) {
  if (log_n > 32) throw LmnError(-100, "bit_reverse: column too large");
  LMN_LAUNCH(k_bit_reverse, dim3(cdiv(1ull << log_n, TPB), ncols), dim3(TPB), 0, s, data, col_stride, log_n);
}

int decompose_num_blocks(int log_n) { return log_n < 1 ? 1 : (int)cdiv(1ull << (log_n - 1), TPB); }

// partial[b] = sum over the block's i < n/2 of f[i] - f[i + n/2]
LMN_KERNEL k_decompose_partial(const uint32_t* __restrict__ f, int log_n, QM31* __restrict__ partial) {
  LMN_SHARED QM31 red[TPB];
  const uint64_t n = 1ull << log_n, half = n >> 1;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  QM31 v = q_zero();
  if (i < half) {
    const QM31 a = load_secure_col(f, n, i), b = load_secure_col(f, n, i + half);
    v = q_sub(a, b);
  }
  red[threadIdx.x] = v;
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = q_add(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// lambda = (sum of partials) * n_inv
LMN_KERNEL k_decompose_lambda(const QM31* __restrict__ partial, int nblocks, uint32_t n_inv, QM31* __restrict__ lambda) {
  LMN_SHARED QM31 red[TPB];
  QM31 acc = q_zero();
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) acc = q_add(acc, partial[b]);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = q_add(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *lambda = q_mul_m(red[0], n_inv);
}
LMN_KERNEL k_decompose_apply(const uint32_t* __restrict__ f, int log_n, uint32_t* __restrict__ g,
                             const QM31* __restrict__ lambda) {
  const uint64_t n = 1ull << log_n;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const QM31 l = *lambda, v = load_secure_col(f, n, i);
  const QM31 r = i >= (n >> 1) ? q_add(v, l) : q_sub(v, l);
  g[i] = r.a;
  g[n + i] = r.b;
  g[2 * n + i] = r.c;
  g[3 * n + i] = r.d;
}
void launch_decompose(const uint32_t* f, int log_n, uint32_t* g, QM31* lambda_out, QM31* scratch, lmn_stream_t s) {
  if (log_n < 1) throw LmnError(-100, "decompose: a circle domain has at least two points");
  const int nb = decompose_num_blocks(log_n);
  LMN_LAUNCH(k_decompose_partial, dim3(nb), dim3(TPB), 0, s, f, log_n, scratch);
  int e = (31 - (log_n % 31)) % 31;  // 2^-log_n mod P
  LMN_LAUNCH(k_decompose_lambda, dim3(1), dim3(TPB), 0, s, scratch, nb, 1u << e, lambda_out);
  LMN_LAUNCH(k_decompose_apply, dim3(cdiv(1ull << log_n, TPB)), dim3(TPB), 0, s, f, log_n, g, lambda_out);
}

}  // namespace lmn
