// gfx950 kernels for the LuminAIR prove hot path.  One wavefront = 64 lanes; all global accesses
// are laid out so consecutive lanes touch consecutive 4-byte words of a column (column-major
// trace data).  See DESIGN.md for the per-kernel roofline notes.
#include <cmath>
#include "kernels.h"
#include "issue_phases.h"
#include "fft_fixed.h"
#include "launch_util.h"

#include <algorithm>

namespace lmn {

constexpr int TPB = 256;

// Wave priority of the short kernels that sit on a proof's serial path (tree tops, FRI tail, scans, small reductions):
// with several proofs in flight their few waves otherwise wait behind every older wave of the big kernels.
#ifndef LMN_SERIAL_PRIO
#define LMN_SERIAL_PRIO 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
#define LMN_SERIAL_KERNEL()                                              \
  do {                                                                   \
    if (LMN_SERIAL_PRIO) __builtin_amdgcn_s_setprio(LMN_SERIAL_PRIO);    \
  } while (0)
#else
#define LMN_SERIAL_KERNEL() do { } while (0)
#endif

static inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// Experiment switch (tools/build_variants.sh qphase "-DLMN_QM31_PHASES"): the 16 multiply-accumulates of a QM31 product and
// the multiply-accumulate runs of the lazy dot products issued as first-port phases (issue_phases.h), as the Blake2s and
// butterfly code does.
#if defined(LMN_QM31_PHASES) && defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
LMN_D QM31 q_mul_phased(QM31 x, QM31 y) {
  const uint32_t nb = P31 - x.b, nd = P31 - x.d;
  const uint32_t c2 = m_dbl(x.c), d2 = m_dbl(x.d);
  const uint32_t e1 = m_sub(c2, x.d);
  const uint32_t e2 = P31 - m_add(d2, x.c);
  const uint32_t f1 = m_add(x.c, d2);
  LMN_PHASE_PORT0();
  const uint64_t lre = (uint64_t)x.a * y.a + (uint64_t)nb * y.b + (uint64_t)e1 * y.c + (uint64_t)e2 * y.d;
  const uint64_t lim = (uint64_t)x.a * y.b + (uint64_t)x.b * y.a + (uint64_t)f1 * y.c + (uint64_t)e1 * y.d;
  const uint64_t hre = (uint64_t)x.a * y.c + (uint64_t)nb * y.d + (uint64_t)x.c * y.a + (uint64_t)nd * y.b;
  const uint64_t him = (uint64_t)x.a * y.d + (uint64_t)x.b * y.c + (uint64_t)x.c * y.b + (uint64_t)x.d * y.a;
  LMN_PHASE_ANY();
  return {m_red64(lre), m_red64(lim), m_red64(hre), m_red64(him)};
}
#define q_mul q_mul_phased
#define LMN_QPHASE_PORT0() LMN_PHASE_PORT0()
#define LMN_QPHASE_ANY() LMN_PHASE_ANY()
#else
#define LMN_QPHASE_PORT0() do { } while (0)
#define LMN_QPHASE_ANY() do { } while (0)
#endif

LMN_D QM31 load_secure_col(const uint32_t* __restrict__ base, uint64_t stride, uint64_t i) {
  return QM31{base[i], base[stride + i], base[2 * stride + i], base[3 * stride + i]};
}

// =============================================================================================
// a3  AoS -> SoA transpose with padding rows (is_last_idx = 1, everything else 0)
// =============================================================================================
constexpr int TR_ROWS = 64;
// Rows [blk_row0, blk_row0 + blk_rows) of the padded table are produced (the whole table, or one rank's row block of a
// sharded proof); row r of column c lands at cols[c * out_stride + (r - blk_row0)].
LMN_KERNEL k_transpose_pad(const uint32_t* __restrict__ rows, uint64_t n_rows, int ncols, uint64_t size,
                           uint32_t* __restrict__ cols, PadRow pad, uint32_t* __restrict__ bad_flag, uint32_t magic,
                           uint64_t out_stride, uint64_t blk_row0, uint32_t bad_value) {
  LMN_DYN_SMEM(uint32_t, tile);  // TR_ROWS x stride
  // odd row stride: the column-major read below walks rows at that stride, and an even one (16 words for the 15 columns of
  // Add) maps the 64 rows of a column onto 2 of the 32 LDS banks
  const int stride = ncols | 1;
  const uint64_t row0 = blk_row0 + (uint64_t)blockIdx.x * TR_ROWS;
  const int total = TR_ROWS * ncols;
  for (int k = threadIdx.x; k < total; k += blockDim.x) {
    // k / ncols by the precomputed reciprocal (exact for k < 2^16): a runtime integer division is ~30 VALU ops
    const int r = magic ? (int)(((uint64_t)(uint32_t)k * magic) >> 32) : k, c = k - r * ncols;  // magic 0: one column
    uint64_t gr = row0 + r;
    uint32_t v;
    if (gr < n_rows)
      v = rows[gr * (uint64_t)ncols + c];
    else
      v = pad.v[c];
    if (v >= P31) *bad_flag = bad_value;  // the boundary takes raw u32 words: reject non-canonical M31 values
    tile[r * stride + c] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < total; k += blockDim.x) {
    int c = k / TR_ROWS, r = k - c * TR_ROWS;
    if (row0 + r < size) cols[(uint64_t)c * out_stride + (row0 - blk_row0) + r] = tile[r * stride + c];
  }
}

void launch_transpose_pad_rows(const uint32_t* rows, uint64_t n_rows, int ncols, int log_size, uint32_t* cols,
                               uint64_t out_stride, uint64_t blk_row0, uint64_t blk_rows, const PadRow& pad, uint32_t* bad_flag,
                               lmn_stream_t s, uint32_t bad_value) {
  uint64_t size = 1ull << log_size;
  if (blk_row0 % TR_ROWS || blk_row0 + blk_rows > size) throw LmnError(-100, "transpose: bad row block");
  unsigned grid = cdiv(blk_rows, TR_ROWS);
  size_t smem = (size_t)TR_ROWS * (ncols + 1) * 4;
  if (ncols > 32 || ncols < 1) throw LmnError(-100, "transpose: bad column count");
  const uint32_t magic = ncols == 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)ncols - 1) / (uint64_t)ncols);  // ceil(2^32 / ncols)
  // rows beyond the block are cut off by treating its end as the table's size
  LMN_LAUNCH(k_transpose_pad, dim3(grid), dim3(TPB), smem, s, rows, n_rows, ncols, blk_row0 + blk_rows, cols, pad, bad_flag, magic,
             out_stride, blk_row0, bad_value);
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
// a4  Circle FFT.  Layer i pairs indices differing in bit i; twiddle index = idx >> (i+1).
//     A pass runs layers [lo, hi) on LDS tiles of 2^(hi-lo) rows x 2^cb contiguous words.
// =============================================================================================
// src may differ from data (out-of-place first pass); words at index >= src_len read as zero
// (zero-extension of a coefficient vector onto a larger domain, i.e. the LDE).
template <bool INV>
LMN_KERNEL k_fft_pass(uint32_t* data, uint64_t col_stride, const uint32_t* src,
                      uint64_t src_stride, uint64_t src_len, int lo, int hi, int cb, TwPtrs tw, uint32_t scale) {
  LMN_DYN_SMEM(uint32_t, sm);
  const int rbits = hi - lo;
  const int C = 1 << cb;
  const int tile_elems = 1 << (rbits + cb);
  uint32_t* col = data + (uint64_t)blockIdx.y * col_stride;
  const uint32_t* scol = src + (uint64_t)blockIdx.y * src_stride;
  const uint32_t tile = blockIdx.x;
  const uint32_t q = tile & ((1u << (lo - cb)) - 1u);
  const uint32_t H = tile >> (lo - cb);
  const uint64_t base = ((uint64_t)H << hi) + ((uint64_t)q << cb);
  for (int e = threadIdx.x; e < tile_elems; e += blockDim.x) {
    int m = e >> cb, c = e & (C - 1);
    uint64_t gi = base + ((uint64_t)m << lo) + c;
    sm[e] = gi < src_len ? scol[gi] : 0u;
  }
  __syncthreads();
  for (int step = 0; step < rbits; ++step) {
    const int i = INV ? lo + step : hi - 1 - step;
    const int bit = i - lo;
    const uint32_t* __restrict__ t = tw.l[i];
    const uint32_t hbase = H << (hi - i - 1);
    for (int b = threadIdx.x; b < tile_elems / 2; b += blockDim.x) {
      int c = b & (C - 1);
      int p = b >> cb;
      int m0 = ((p >> bit) << (bit + 1)) | (p & ((1 << bit) - 1));
      int m1 = m0 | (1 << bit);
      uint32_t w = t[hbase + (uint32_t)(m0 >> (bit + 1))];
      int i0 = (m0 << cb) | c, i1 = (m1 << cb) | c;
      uint32_t v0 = sm[i0], v1 = sm[i1];
      if (INV) {
        sm[i0] = m_add(v0, v1);
        sm[i1] = m_mul(m_sub(v0, v1), w);
      } else {
        uint32_t x = m_mul(v1, w);
        sm[i0] = m_add(v0, x);
        sm[i1] = m_sub(v0, x);
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < tile_elems; e += blockDim.x) {
    int m = e >> cb, c = e & (C - 1);
    uint32_t v = sm[e];
    if (INV && scale != 1u) v = m_mul(v, scale);
    col[base + ((uint64_t)m << lo) + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Staged variant: each lane keeps 2^R points (R <= 4) in registers and runs R butterfly layers
// on them before exchanging through LDS, so a 12-layer tile needs 2 LDS exchanges instead of 12
// and the per-layer index arithmetic disappears.  Tile geometry as in k_fft_pass.
// ---------------------------------------------------------------------------------------------
constexpr int FFT_MAX_STAGES = 4;
struct FftStagePlan {
  int lo, hi, cb;
  int nst;
  int first[FFT_MAX_STAGES];  // first layer of each stage, ascending
  int R[FFT_MAX_STAGES];      // layers per stage (1..4)
  int xcd_swizzle;
};

LMN_HD uint32_t fft_lds_pad(uint32_t e) { return e + (e >> 5); }

// out[k] = x[k] * w[k] in M31 for N independent products, issued in phases (field.h m_mul, same arithmetic).  Leaves the
// wave in the first-port phase.
template <int N>
LMN_D void m_mul_phased(const uint32_t (&x)[N], const uint32_t (&w)[N], uint32_t (&out)[N]) {
  uint64_t pr[N];
  uint32_t hi[N], s[N], s2[N];
  LMN_PHASE_PORT0();
#pragma unroll
  for (int k = 0; k < N; ++k) pr[k] = (uint64_t)x[k] * (uint64_t)w[k];
#pragma unroll
  for (int k = 0; k < N; ++k) hi[k] = (uint32_t)(pr[k] >> 31);
  LMN_PHASE_ANY();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    s[k] = ((uint32_t)pr[k] & P31) + hi[k];
    s2[k] = s[k] - P31;
  }
  LMN_PHASE_PORT0();
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = s[k] < s2[k] ? s[k] : s2[k];
}

// The 2^R - 1 twiddles of a register stage (layer r needs 2^(R-1-r) of them: butterfly k of the layer uses entry k >> r).
// Loaded in one go BEFORE the stage's data so that a single memory latency covers all R layers - loading them layer by
// layer put one exposed L2 round trip in front of every layer (the kernels run at 2.6 - 4 waves per SIMD).
template <int R>
struct StageTwiddles {
  uint32_t t[(1 << R) - 1];  // layers 0 .. R-1 one after the other
  static constexpr int at(int r) { return (1 << R) - (1 << (R - r)); }  // first entry of layer r
};

template <int R, bool INV>
LMN_D void load_stage_twiddles(StageTwiddles<R>& T, const TwPtrs& tw, int first_layer, int hi, uint32_t H, uint32_t mhigh) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int L = first_layer + r;
    const uint32_t* __restrict__ t = tw.l[L];
    const uint32_t hb = (H << (hi - L - 1)) + (mhigh << (R - 1 - r));
    LMN_ASSUME(hb < (1u << 28));  // lets the compiler use 32-bit offsets from the uniform table pointer
#pragma unroll
    for (int q = 0; q < (1 << (R - 1 - r)); ++q) T.t[StageTwiddles<R>::at(r) + q] = t[hb + (uint32_t)q];
  }
}

template <int R, bool INV>
LMN_D void radix_butterflies(uint32_t (&v)[1 << R], const StageTwiddles<R>& T) {
  constexpr int NB = 1 << (R - 1);  // independent butterflies per layer
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const int r = INV ? rr : R - 1 - rr;
    uint32_t w[NB], a[NB], b[NB], x[NB], u[NB], u2[NB], d[NB], d2[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));  // k-th index with bit r clear; j >> (r+1) == k >> r
      w[k] = T.t[StageTwiddles<R>::at(r) + (k >> r)];
      a[k] = v[j];
      b[k] = v[j | (1 << r)];
    }
    if (INV) {
      LMN_PHASE_ANY();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        u[k] = a[k] + b[k];
        u2[k] = u[k] - P31;
        d[k] = a[k] - b[k];
        d2[k] = d[k] + P31;
      }
      LMN_PHASE_PORT0();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        a[k] = u[k] < u2[k] ? u[k] : u2[k];   // m_add(a, b)
        d[k] = d[k] < d2[k] ? d[k] : d2[k];   // m_sub(a, b)
      }
      m_mul_phased<NB>(d, w, x);
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));
        v[j] = a[k];
        v[j | (1 << r)] = x[k];
      }
    } else {
      m_mul_phased<NB>(b, w, x);
      LMN_PHASE_ANY();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        u[k] = a[k] + x[k];
        u2[k] = u[k] - P31;
        d[k] = a[k] - x[k];
        d2[k] = d[k] + P31;
      }
      LMN_PHASE_PORT0();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));
        v[j] = u[k] < u2[k] ? u[k] : u2[k];              // m_add(a, x)
        v[j | (1 << r)] = d[k] < d2[k] ? d[k] : d2[k];   // m_sub(a, x)
      }
    }
  }
  LMN_PHASE_ANY();
}

// sm_in / sm_out: LDS tile read by a stage that does not load from global / written by one that does not store to
// global (the same buffer in the plain passes).  keep != nullptr: a to_global stage also leaves its (scaled) values in
// that LDS tile (the fused interpolate + extend pass continues from them).
template <int R, bool INV>
LMN_D void fft_stage(const uint32_t* sm_in, uint32_t* sm_out, uint32_t* col, const uint32_t* scol, uint64_t src_len,
                     uint64_t base, int lo, int hi, int cb, int first_layer, uint32_t H, bool from_global, bool to_global,
                     const TwPtrs& tw, uint32_t scale, uint32_t* keep = nullptr) {
  const int p = first_layer - lo + cb;           // bit position of the stage's first layer in the tile index
  const uint32_t tile_elems = 1u << (hi - lo + cb);
  const uint32_t ngroups = tile_elems >> R;
  const uint32_t cmask = (1u << cb) - 1u;
  // 32-bit offsets from the (block-uniform) tile base keep the address arithmetic off the 64-bit path
  const uint32_t* __restrict__ tsrc = scol + base;
  uint32_t* __restrict__ tdst = col + base;
  const uint64_t span = src_len > base ? src_len - base : 0;
  const uint32_t lim = span > 0x10000000ull ? 0x10000000u : (uint32_t)span;  // readable words from tsrc
  // The 2^R points of a group sit at tile index e0 + (j << p) (bits [p, p+R) of e0 are zero), so both
  // address maps split into a per-lane base plus a wave-uniform term in j: global offset
  // ((e >> cb) << lo) + (e & cmask) has stride 2^(p - cb + lo) (p >= cb always), and the padded LDS index
  // pad(e) = e + (e >> 5) satisfies pad(e0 + (j << p)) = pad(e0) + pad(j << p) because the two addends
  // occupy disjoint bits (no carry into bit 5).  One add per point instead of re-deriving each address.
  const uint32_t gstride = 1u << (p - cb + lo);
  const uint32_t tile_span = (((tile_elems - 1u) >> cb) << lo) + cmask + 1u;  // words of [tsrc, ...) the tile touches
  const bool full = tile_span <= lim;                                           // block-uniform: no zero-extension here
  for (uint32_t g = threadIdx.x; g < ngroups; g += blockDim.x) {
    const uint32_t e0 = ((g >> p) << (p + R)) | (g & ((1u << p) - 1u));
    const uint32_t off0 = ((e0 >> cb) << lo) + (e0 & cmask);
    LMN_ASSUME(off0 < 0x10000000u);
    const uint32_t m0 = e0 >> cb;
    StageTwiddles<R> T;
    load_stage_twiddles<R, INV>(T, tw, first_layer, hi, H, m0 >> (first_layer - lo + R));
    uint32_t v[1 << R];
    if (from_global) {
      if (p == 0 && cb == 0 && R >= 2) {
        if (e0 < lim) {
          const uint4* q = reinterpret_cast<const uint4*>(tsrc + e0);
#pragma unroll
          for (int k = 0; k < (1 << R) / 4; ++k) {
            uint4 x = q[k];
            v[4 * k] = x.x;
            v[4 * k + 1] = x.y;
            v[4 * k + 2] = x.z;
            v[4 * k + 3] = x.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < (1 << R); ++j) v[j] = 0u;
        }
      } else if (full) {
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) v[j] = tsrc[off0 + (uint32_t)j * gstride];
      } else {
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) {
          const uint32_t off = off0 + (uint32_t)j * gstride;
          v[j] = off < lim ? tsrc[off] : 0u;
        }
      }
    } else {
      const uint32_t pb = fft_lds_pad(e0);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) v[j] = sm_in[pb + fft_lds_pad((uint32_t)j << p)];
    }
    radix_butterflies<R, INV>(v, T);
    if (to_global) {
      if (INV && scale != 1u) {
        uint32_t sc[1 << R], pr[1 << R];
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) sc[j] = scale;
        m_mul_phased<(1 << R)>(v, sc, pr);
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) v[j] = pr[j];
        LMN_PHASE_ANY();
      }
      if (p == 0 && cb == 0 && R >= 2) {
        uint4* q = reinterpret_cast<uint4*>(tdst + e0);
#pragma unroll
        for (int k = 0; k < (1 << R) / 4; ++k) q[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
      } else {
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) tdst[off0 + (uint32_t)j * gstride] = v[j];
      }
      if (keep) {
        const uint32_t pb = fft_lds_pad(e0);
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) keep[pb + fft_lds_pad((uint32_t)j << p)] = v[j];
      }
    } else {
      const uint32_t pb = fft_lds_pad(e0);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) sm_out[pb + fft_lds_pad((uint32_t)j << p)] = v[j];
    }
  }
}

template <bool INV>
LMN_D void fft_stage_dispatch(int R, const uint32_t* sm_in, uint32_t* sm_out, uint32_t* col, const uint32_t* scol,
                              uint64_t src_len, uint64_t base, int lo, int hi, int cb, int first_layer, uint32_t H,
                              bool from_global, bool to_global, const TwPtrs& tw, uint32_t scale, uint32_t* keep = nullptr) {
  switch (R) {
    case 1: fft_stage<1, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
    case 2: fft_stage<2, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
    case 3: fft_stage<3, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
    default: fft_stage<4, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
  }
}

template <bool INV>
LMN_KERNEL k_fft_staged(uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride,
                        uint64_t src_len, FftStagePlan pl, TwPtrs tw, uint32_t scale, int ncols, int cpb,
                        uint32_t h_off) {
  LMN_DYN_SMEM(uint32_t, sm);
  // XCD-aware tile order: consecutive workgroups are dealt round-robin to the 8 XCDs (observed, used
  // for speed only), so give each XCD a contiguous run of tiles: neighbouring strided tiles share
  // 128-byte lines and then hit the same L2.
  uint32_t tile = blockIdx.x;
  if (pl.xcd_swizzle && (gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t q = tile & ((1u << (pl.lo - pl.cb)) - 1u);
  const uint32_t Hl = tile >> (pl.lo - pl.cb);
  const uint64_t base = ((uint64_t)Hl << pl.hi) + ((uint64_t)q << pl.cb);
  // h_off != 0: the data is one aligned block of a larger domain (row-block sharding): addresses stay local, the
  // twiddle index carries the block's position
  const uint32_t H = Hl + h_off;
  for (int cc = 0; cc < cpb; ++cc) {
    const int c = blockIdx.y * cpb + cc;
    if (c >= ncols) break;
    uint32_t* col = data + (uint64_t)c * col_stride;
    const uint32_t* scol = src + (uint64_t)c * src_stride;
    for (int k = 0; k < pl.nst; ++k) {
      const int s = INV ? k : pl.nst - 1 - k;
      const bool fg = k == 0, tg = k == pl.nst - 1;
      fft_stage_dispatch<INV>(pl.R[s], sm, sm, col, scol, src_len, base, pl.lo, pl.hi, pl.cb, pl.first[s], H, fg, tg, tw, scale);
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused "interpolate then extend" pass (PolyOps::interpolate followed by PolyOps::evaluate on the blown-up domain, as
// every committed column goes through: prover.rs:56-59,179,298).  The last pass of the inverse transform on 2^n points
// (layers [lo, n), strided tile) and the first pass of the forward transform onto 2^(n+1) points touch the SAME
// coefficient positions: the zero-extended coefficient vector makes the forward layer n the identity on both halves,
// and layers [lo, n) of each half h are the tile's own layers with twiddle row H = h.  So one workgroup loads the
// tile once, finishes the interpolation (coefficients go to HBM - the OODS evaluation needs them - and stay in LDS),
// and runs the forward layers twice from LDS, writing both halves of the extended evaluation: one launch and one
// read of the coefficients less per column than two separate passes.
// ---------------------------------------------------------------------------------------------
LMN_KERNEL k_fft_interp_extend(uint32_t* coeffs, uint64_t coeff_stride, const uint32_t* src, uint64_t src_stride,
                               uint32_t* lde, uint64_t lde_stride, FftStagePlan pl, TwPtrs itw, TwPtrs tw, uint32_t scale,
                               int ncols, int cpb) {
  LMN_DYN_SMEM(uint32_t, sm);
  const uint32_t tile_elems = 1u << (pl.hi - pl.lo + pl.cb);
  uint32_t* A = sm;                                   // the coefficient tile (kept for the second half)
  uint32_t* B = sm + fft_lds_pad(tile_elems) + 1u;    // exchange buffer of the stages
  uint32_t tile = blockIdx.x;
  if (pl.xcd_swizzle && (gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint64_t base = (uint64_t)tile << pl.cb;      // hi = n: the tile spans the whole column in its strided rows
  const uint64_t n_words = 1ull << pl.hi;
  for (int cc = 0; cc < cpb; ++cc) {
    const int c = blockIdx.y * cpb + cc;
    if (c >= ncols) break;
    uint32_t* ccol = coeffs + (uint64_t)c * coeff_stride;
    const uint32_t* scol = src + (uint64_t)c * src_stride;
    uint32_t* lcol = lde + (uint64_t)c * lde_stride;
    // inverse layers [lo, n): stages ascending; the last one scales, stores the coefficients and keeps them in A
    for (int k = 0; k < pl.nst; ++k) {
      const bool fg = k == 0, tg = k == pl.nst - 1;
      fft_stage_dispatch<true>(pl.R[k], B, B, ccol, scol, n_words, base, pl.lo, pl.hi, pl.cb, pl.first[k], 0u, fg, tg, itw, scale,
                               tg ? A : nullptr);
      __syncthreads();
    }
    // forward layers [lo, n) of half h of the 2^(n+1)-point transform: stages descending, first one reads A
    for (uint32_t h = 0; h < 2; ++h) {
      for (int k = 0; k < pl.nst; ++k) {
        const int s = pl.nst - 1 - k;
        const bool tg = k == pl.nst - 1;
        fft_stage_dispatch<false>(pl.R[s], k == 0 ? A : B, B, lcol + h * n_words, lcol, 0, base, pl.lo, pl.hi, pl.cb, pl.first[s], h,
                                  false, tg, tw, 1u);
        __syncthreads();
      }
    }
  }
}

struct FftPass {
  int lo, hi, cb;
};
constexpr int FFT_LOW_BITS = 12;    // contiguous low pass: 2^12 words = 16 KiB LDS
constexpr int FFT_HIGH_BITS = 10;   // strided high passes: 2^10 rows x 16 words = 64 KiB LDS
constexpr int FFT_HIGH_CB = 4;

static int plan_passes(int log_n, FftPass* out) {
  // LMN_FFT_SPLIT (experiment): 0 = strided passes of equal depth (default), 1 = deepest first (10, then the rest),
  // 2 = deepest last
  static const int split = getenv("LMN_FFT_SPLIT") ? atoi(getenv("LMN_FFT_SPLIT")) : 0;
  int n = 0;
  int lo = 0;
  int hi = log_n < FFT_LOW_BITS ? log_n : FFT_LOW_BITS;
  out[n++] = {0, hi, 0};
  lo = hi;
  while (lo < log_n) {
    int rem = log_n - lo;
    int npass = (rem + FFT_HIGH_BITS - 1) / FFT_HIGH_BITS;
    int take = (rem + npass - 1) / npass;
    if (split == 1 && npass > 1) take = FFT_HIGH_BITS;
    if (split == 2 && npass > 1) take = rem - (npass - 1) * FFT_HIGH_BITS;
    // three-pass sizes (2^23 points and more) have strided passes of 5 - 7 layers: 32-word runs (whole 128-byte lines)
    // keep their tiles at 4 - 16 KiB; the single strided pass of the smaller sizes keeps 16-word runs (up to 2^10 rows)
    static const int cb3 = getenv("LMN_FFT_CB3") ? atoi(getenv("LMN_FFT_CB3")) : 5;
    out[n++] = {lo, lo + take, take == 5 ? 5 : (npass > 1 && take <= 7 ? cb3 : FFT_HIGH_CB)};
    lo += take;
  }
  return n;
}

static uint32_t inv_pow2(int log_n) {
  // 2^-log_n mod P = 2^(31 - log_n mod 31)
  int e = (31 - (log_n % 31)) % 31;
  return 1u << e;
}

static void split_stages(FftStagePlan& pl) {
  int rbits = pl.hi - pl.lo;
  int nst = (rbits + 3) / 4;
  pl.nst = nst;
  int f = pl.lo;
  for (int k = 0; k < nst; ++k) {
    int r = (rbits - (f - pl.lo) + (nst - k) - 1) / (nst - k);  // balanced split, each <= 4
    pl.first[k] = f;
    pl.R[k] = r;
    f += r;
  }
}

// one k_fft_staged launch: layers [p.lo, p.hi) of a 2^log_n transform
template <bool INV>
static void launch_staged_pass(uint32_t* data, uint64_t col_stride, const uint32_t* psrc, uint64_t pstride, uint64_t plen,
                               const FftPass& p, int log_n, const TwPtrs& tw, uint32_t scale, int ncols, lmn_stream_t s,
                               uint32_t block_index = 0) {
  const int rbits = p.hi - p.lo;
  const unsigned tiles = 1u << (log_n - rbits - p.cb);
  FftStagePlan pl{};
  pl.lo = p.lo;
  pl.hi = p.hi;
  pl.cb = p.cb;
  split_stages(pl);
  static const int env_xcd = getenv("LMN_FFT_XCD") ? atoi(getenv("LMN_FFT_XCD")) : 1;
  pl.xcd_swizzle = (env_xcd && p.cb > 0) ? 1 : 0;
  uint32_t tile_elems = 1u << (rbits + p.cb);
  size_t smem = (size_t)4 * (tile_elems + (tile_elems >> 5) + 1);
  // several columns per block when there are plenty of tiles: twiddles stay hot in L1/L2
  int cpb = tiles >= 2048 ? 3 : (tiles >= 512 ? 2 : 1);
  static const int env_cpb = getenv("LMN_FFT_CPB") ? atoi(getenv("LMN_FFT_CPB")) : 0;
  static const int env_thr = getenv("LMN_FFT_THREADS") ? atoi(getenv("LMN_FFT_THREADS")) : 0;
  if (env_cpb > 0) cpb = env_cpb;
  if (cpb > ncols) cpb = ncols;
  // the shapes of the prover's committed columns have compile-time-specialised kernels (fft_fixed.hip)
  const bool full = plen >= (1ull << log_n);
  const bool half_top = !INV && plen == (1ull << (log_n - 1)) && p.hi == log_n && p.lo > 0;   // LDE by one bit, top pass
  if (full || half_top) {
    const uint32_t scale_log = scale == 1u ? 0u : (uint32_t)__builtin_ctz(scale);
    if (launch_fft_fixed_pass(INV, data, col_stride, psrc, pstride, p.lo, rbits, p.cb, log_n, tw, scale_log, ncols, cpb,
                              block_index << (log_n - p.hi), pl.xcd_swizzle, !full, s))
      return;
  }
  unsigned gy = (unsigned)((ncols + cpb - 1) / cpb);
  int threads = (int)std::min<uint32_t>(TPB, std::max<uint32_t>(64u, tile_elems >> 4));
  if (env_thr > 0) threads = env_thr;
  LMN_LAUNCH(k_fft_staged<INV>, dim3(tiles, gy), dim3(threads), smem, s, data, col_stride, psrc, pstride, plen, pl, tw,
             scale, ncols, cpb, block_index << (log_n - p.hi));
}

template <bool INV>
static int run_fft(uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride, int log_src,
                    int ncols, int log_n, const TwPtrs& tw, lmn_stream_t s, uint32_t block_index = 0) {
  if (log_n < 1) throw LmnError(-100, "fft: log_n < 1");
  static const bool use_v1 = getenv("LMN_FFT_V1") != nullptr;
  FftPass passes[8];
  int np = plan_passes(log_n, passes);
  for (int k = 0; k < np; ++k) {
    const FftPass& p = INV ? passes[k] : passes[np - 1 - k];
    int rbits = p.hi - p.lo;
    unsigned tiles = 1u << (log_n - rbits - p.cb);
    bool last = INV && k == np - 1;
    uint32_t scale = last ? inv_pow2(log_n) : 1u;
    const uint32_t* psrc = k == 0 ? src : data;
    uint64_t pstride = k == 0 ? src_stride : col_stride;
    uint64_t plen = k == 0 ? (1ull << log_src) : (1ull << log_n);
    if (use_v1) {
      size_t smem = (size_t)4 << (rbits + p.cb);
      LMN_LAUNCH(k_fft_pass<INV>, dim3(tiles, ncols), dim3(TPB), smem, s, data, col_stride, psrc, pstride, plen,
                 p.lo, p.hi, p.cb, tw, scale);
      continue;
    }
    launch_staged_pass<INV>(data, col_stride, psrc, pstride, plen, p, log_n, tw, scale, ncols, s, block_index);
  }
  return np;
}

// interpolate (2^log_n evaluations -> coefficients, kept) + extend onto the 2^(log_n + 1) domain in three launches:
// inverse low pass, the fused strided pass (k_fft_interp_extend), forward low pass.  Applies when both transforms
// have exactly one strided pass of at most FFT_HIGH_BITS layers.
bool fft_interp_extend_supported(int log_n) {
  static const bool off = getenv("LMN_NO_FFT_FUSION") != nullptr;
  return !off && log_n > FFT_LOW_BITS && log_n - FFT_LOW_BITS <= FFT_HIGH_BITS - 1;
}
int launch_interp_extend(uint32_t* coeffs, uint64_t coeff_stride, const uint32_t* evals, uint64_t evals_stride, uint32_t* lde,
                         uint64_t lde_stride, int ncols, int log_n, const TwPtrs& itw, const TwPtrs& tw_ext, lmn_stream_t s) {
  if (LMN_ABLATED(2u)) return 3;
  if (!fft_interp_extend_supported(log_n)) throw LmnError(-100, "interp_extend: unsupported size");
  const FftPass low{0, FFT_LOW_BITS, 0};
  launch_staged_pass<true>(coeffs, coeff_stride, evals, evals_stride, 1ull << log_n, low, log_n, itw, 1u, ncols, s);
  FftStagePlan pl{};
  pl.lo = FFT_LOW_BITS;
  pl.hi = log_n;
  pl.cb = FFT_HIGH_CB;
  split_stages(pl);
  pl.xcd_swizzle = 1;
  const uint32_t tile_elems = 1u << (log_n - FFT_LOW_BITS + FFT_HIGH_CB);
  const size_t smem = (size_t)8 * (tile_elems + (tile_elems >> 5) + 1);
  const unsigned tiles = 1u << (FFT_LOW_BITS - FFT_HIGH_CB);
  int cpb = 1;
  const int threads = (int)std::min<uint32_t>(TPB, std::max<uint32_t>(64u, tile_elems >> 4));
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
  if (smem > 64 * 1024) allow_big_lds((const void*)k_fft_interp_extend, 160 * 1024);
#endif
  if (!launch_interp_extend_fixed(coeffs, coeff_stride, lde, lde_stride, log_n, itw, tw_ext, ncols, s))
    LMN_LAUNCH(k_fft_interp_extend, dim3(tiles, (unsigned)((ncols + cpb - 1) / cpb)), dim3(threads), smem, s, coeffs,
               coeff_stride, coeffs, coeff_stride, lde, lde_stride, pl, itw, tw_ext, inv_pow2(log_n), ncols, cpb);
  launch_staged_pass<false>(lde, lde_stride, lde, lde_stride, 2ull << log_n, low, log_n + 1, tw_ext, 1u, ncols, s);
  return 3;
}

int launch_ifft(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int ncols, int log_n,
                const TwPtrs& itw, lmn_stream_t s) {
  if (LMN_ABLATED(2u)) return 1;
  return run_fft<true>(dst, dst_stride, src, src_stride, log_n, ncols, log_n, itw, s);
}
int launch_fft(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int log_src, int ncols,
               int log_n, const TwPtrs& tw, lmn_stream_t s) {
  if (LMN_ABLATED(2u)) return 1;
  return run_fft<false>(dst, dst_stride, src, src_stride, log_src, ncols, log_n, tw, s);
}

// Row-block LDE (single-commitment sharding): block `b` of 2^g equal blocks of the forward transform
// onto a 2^n domain.  The top g layers pair indices that differ in the block bits only and their twiddle index
// (idx >> (i+1)) depends on the block bits only, so element t of block b after those layers is a 2^g-point
// transform across the blocks at fixed t - computed here from the (zero-extended) coefficients; the remaining
// layers then run inside the block (k_fft_staged with the block's twiddle offset).
template <int G>
LMN_KERNEL k_fft_top_block(uint32_t* __restrict__ dst, uint64_t dst_stride, const uint32_t* __restrict__ src,
                           uint64_t src_stride, uint64_t src_len, int log_n, uint32_t block, TwPtrs tw) {
  const uint64_t S = 1ull << (log_n - G);
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= S) return;
  const uint32_t* scol = src + (uint64_t)blockIdx.y * src_stride;
  uint32_t v[1 << G];
#pragma unroll
  for (int c = 0; c < (1 << G); ++c) {
    const uint64_t idx = (uint64_t)c * S + t;
    v[c] = idx < src_len ? scol[idx] : 0u;
  }
#pragma unroll
  for (int k = G - 1; k >= 0; --k) {
    const uint32_t* __restrict__ tl = tw.l[log_n - G + k];
#pragma unroll
    for (int c = 0; c < (1 << G); ++c) {
      if (c & (1 << k)) continue;
      const uint32_t w = tl[c >> (k + 1)];
      const uint32_t a = v[c], x = m_mul(v[c | (1 << k)], w);
      v[c] = m_add(a, x);
      v[c | (1 << k)] = m_sub(a, x);
    }
  }
  uint32_t out = v[0];
#pragma unroll
  for (int c = 1; c < (1 << G); ++c) out = block == (uint32_t)c ? v[c] : out;
  dst[(uint64_t)blockIdx.y * dst_stride + t] = out;
}

int launch_fft_block(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int log_src, int ncols,
                     int log_n, int log_blocks, uint32_t block, const TwPtrs& tw, lmn_stream_t s) {
  if (log_blocks < 1 || log_blocks > 3 || log_n - log_blocks < 1 || block >= (1u << log_blocks))
    throw LmnError(-100, "fft_block: bad arguments");
  const int lb = log_n - log_blocks;
  dim3 g(cdiv(1ull << lb, TPB), ncols), b(TPB);
  const uint64_t slen = 1ull << log_src;
  switch (log_blocks) {
    case 1: LMN_LAUNCH(k_fft_top_block<1>, g, b, 0, s, dst, dst_stride, src, src_stride, slen, log_n, block, tw); break;
    case 2: LMN_LAUNCH(k_fft_top_block<2>, g, b, 0, s, dst, dst_stride, src, src_stride, slen, log_n, block, tw); break;
    default: LMN_LAUNCH(k_fft_top_block<3>, g, b, 0, s, dst, dst_stride, src, src_stride, slen, log_n, block, tw); break;
  }
  return 1 + run_fft<false>(dst, dst_stride, dst, dst_stride, lb, ncols, lb, tw, s, block);
}

// one global-memory layer per launch: the obviously-correct reference used by the self-test
LMN_KERNEL k_fft_layer_simple(uint32_t* __restrict__ data, uint64_t col_stride, int log_n, int i,
                              const uint32_t* __restrict__ t, int inverse, uint32_t scale) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t half = 1ull << (log_n - 1);
  if (b >= half) return;
  uint32_t* col = data + (uint64_t)blockIdx.y * col_stride;
  uint64_t h = b >> i, l = b & ((1ull << i) - 1);
  uint64_t i0 = (h << (i + 1)) + l, i1 = i0 + (1ull << i);
  uint32_t w = t[h];
  uint32_t v0 = col[i0], v1 = col[i1];
  if (inverse) {
    uint32_t a = m_add(v0, v1), d = m_mul(m_sub(v0, v1), w);
    if (scale != 1u) {
      a = m_mul(a, scale);
      d = m_mul(d, scale);
    }
    col[i0] = a;
    col[i1] = d;
  } else {
    uint32_t x = m_mul(v1, w);
    col[i0] = m_add(v0, x);
    col[i1] = m_sub(v0, x);
  }
}

LMN_KERNEL k_pack_blocks(uint32_t* __restrict__ cols, uint64_t col_stride, uint32_t* __restrict__ packed, uint32_t block_rows,
                         int ncols, int nsel, PackSel sel, int unpack) {
  const uint32_t h = blockIdx.y % (uint32_t)nsel, c = (blockIdx.y / (uint32_t)nsel) % (uint32_t)ncols;
  const uint32_t s = blockIdx.y / (uint32_t)(nsel * ncols);
  uint32_t* cp = cols + (uint64_t)c * col_stride + (uint64_t)sel.blk[s][h] * block_rows;
  uint32_t* pp = packed + (((uint64_t)s * ncols + c) * nsel + h) * block_rows;
  if (unpack) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < block_rows; i += gridDim.x * blockDim.x) cp[i] = pp[i];
  } else {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < block_rows; i += gridDim.x * blockDim.x) pp[i] = cp[i];
  }
}
void launch_pack_blocks(const uint32_t* src, uint64_t src_stride, uint32_t* dst, uint32_t block_rows, int ncols, int nsel,
                        int world, const PackSel& sel, lmn_stream_t s) {
  if (ncols <= 0) return;
  LMN_LAUNCH(k_pack_blocks, dim3(std::min<unsigned>(cdiv(block_rows, TPB), 64u), (unsigned)(world * ncols * nsel)), dim3(TPB),
             0, s, const_cast<uint32_t*>(src), src_stride, dst, block_rows, ncols, nsel, sel, 0);
}
void launch_unpack_blocks(const uint32_t* packed, uint32_t* cols, uint64_t col_stride, uint32_t block_rows, int ncols, int world,
                          lmn_stream_t s) {
  if (ncols <= 0) return;
  PackSel sel{};
  for (int r = 0; r < world; ++r) sel.blk[r][0] = (uint32_t)r;
  LMN_LAUNCH(k_pack_blocks, dim3(std::min<unsigned>(cdiv(block_rows, TPB), 64u), (unsigned)(world * ncols)), dim3(TPB), 0, s,
             cols, col_stride, const_cast<uint32_t*>(packed), block_rows, ncols, 1, sel, 1);
}

void launch_fft_simple(uint32_t* data, uint64_t col_stride, int ncols, int log_n, const TwPtrs& tw, bool inverse,
                       lmn_stream_t s) {
  uint64_t half = 1ull << (log_n - 1);
  for (int k = 0; k < log_n; ++k) {
    int i = inverse ? k : log_n - 1 - k;
    uint32_t scale = (inverse && k == log_n - 1) ? inv_pow2(log_n) : 1u;
    LMN_LAUNCH(k_fft_layer_simple, dim3(cdiv(half, TPB), ncols), dim3(TPB), 0, s, data, col_stride, log_n, i,
               tw.l[i], inverse ? 1 : 0, scale);
  }
}

LMN_KERNEL k_extend(const uint32_t* __restrict__ src, uint64_t src_stride, uint64_t src_len,
                    uint32_t* __restrict__ dst, uint64_t dst_stride, uint64_t dst_len) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dst_len) return;
  uint32_t v = i < src_len ? src[(uint64_t)blockIdx.y * src_stride + i] : 0u;
  dst[(uint64_t)blockIdx.y * dst_stride + i] = v;
}

void launch_extend(const uint32_t* src, uint64_t src_stride, int log_src, uint32_t* dst, uint64_t dst_stride,
                   int log_dst, int ncols, lmn_stream_t s) {
  uint64_t dl = 1ull << log_dst;
  LMN_LAUNCH(k_extend, dim3(cdiv(dl, TPB), ncols), dim3(TPB), 0, s, src, src_stride, 1ull << log_src, dst,
             dst_stride, dl);
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

// =============================================================================================
// a4  Blake2s Merkle layer: one lane per node; a wave reads 64 consecutive rows of each column.
// =============================================================================================
LMN_KERNEL k_merkle_layer(const uint32_t* __restrict__ prev, const uint32_t* const* __restrict__ cols, int ncols,
                          uint32_t size, uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  const int npre = prev ? 16 : 0;
  const int nwords = npre + ncols;
  const int nblocks = (nwords + 15) / 16;
  uint32_t h[8];
  b2_init(h);
  for (int b = 0; b < nblocks; ++b) {
    uint32_t m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int w = b * 16 + k;
      uint32_t v = 0u;
      if (w < npre)
        v = prev[(uint64_t)i * 16 + w];
      else if (w < nwords)
        v = cols[w - npre][i];
      m[k] = v;
    }
    bool last = b + 1 == nblocks;
    b2_compress(h, m, last ? (uint32_t)(4 * nwords) : (uint32_t)(64 * (b + 1)), last ? 0xffffffffu : 0u);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) out[(uint64_t)i * 8 + k] = h[k];
}

void launch_merkle_layer(const uint32_t* prev, const uint32_t* const* cols, int ncols, uint32_t size, uint32_t* out,
                         lmn_stream_t s) {
  if (!prev && ncols == 0) throw LmnError(-100, "merkle layer with no input");
  LMN_LAUNCH(k_merkle_layer, dim3(cdiv(size, TPB)), dim3(TPB), 0, s, prev, cols, ncols, size, out);
}

// Fused Merkle subtree.  Every lane owns 2^sub consecutive start-level nodes and reduces them to one
// subtree root in registers (post-order, private LDS slots as the merge stack: all 64 lanes of a
// wave stay busy on every compression); the block's subtree roots then climb further levels
// through LDS.  All levels are written to HBM (decommitment needs them).
// Start-level columns are described as up to MERKLE_MAX_SEG runs of contiguous columns
// (column c of a run lives at base + c*size), so no per-column pointer loads are needed.
LMN_D const uint32_t* merkle_col_ptr(const MerkleSegs& sg, int c, uint64_t size) {
  // c is compile-time after unrolling; the comparisons are wave-uniform scalar work
  int n0 = sg.n[0], n1 = n0 + sg.n[1], n2 = n1 + sg.n[2];
  if (c < n0) return sg.base[0] + (uint64_t)c * size;
  if (c < n1) return sg.base[1] + (uint64_t)(c - n0) * size;
  if (c < n2) return sg.base[2] + (uint64_t)(c - n1) * size;
  return sg.base[3] + (uint64_t)(c - n2) * size;
}

// First 16 message words of start-level node i: the two child hashes when the level has a `prev`
// layer, else its first 16 columns (zero padded).  Split from the hashing so that callers can issue the
// loads of the next node before compressing the current one.
LMN_D void merkle_load_first(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                             uint32_t i, uint32_t m[16]) {
  if (prev) {
    const uint4* p4 = reinterpret_cast<const uint4*>(prev) + (uint64_t)i * 4;
    uint4 a = p4[0], b = p4[1], c = p4[2], d = p4[3];
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
    m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
    m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
    m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = k < ncols ? merkle_col_ptr(sg, k, size)[i] : 0u;
  }
}

// Hash of start-level node i given its first 16 message words (merkle_load_first).
LMN_D void merkle_hash_from(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                            uint32_t i, uint32_t m[16], uint32_t h[8]) {
  b2_init(h);
  const uint32_t total = (prev ? 64u : 0u) + 4u * (uint32_t)ncols;
  int c0 = prev ? 0 : 16;  // first column not yet consumed
  if (c0 >= ncols) {
    b2_compress(h, m, total, 0xffffffffu);
    return;
  }
  b2_compress(h, m, 64u, 0u);
  uint32_t done = 64u;
  while (c0 < ncols) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int c = c0 + k;
      m[k] = c < ncols ? merkle_col_ptr(sg, c, size)[i] : 0u;
    }
    c0 += 16;
    bool last = c0 >= ncols;
    done += 64u;
    b2_compress(h, m, last ? total : done, last ? 0xffffffffu : 0u);
  }
}

LMN_D void merkle_hash_start(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                             uint32_t i, uint32_t h[8]) {
  uint32_t m[16];
  merkle_load_first(prev, sg, ncols, size, i, m);
  merkle_hash_from(prev, sg, ncols, size, i, m, h);
}

// MerkleFold::below: raw words of leaves 2i (m[0..7]) and 2i+1 (m[8..15]) of the level under the start level
LMN_D void merkle_load_below(const uint32_t* __restrict__ below, int below_ncols, uint32_t size, uint32_t i, uint32_t m[16]) {
  const uint64_t L = 2ull * size;
  const uint32_t* __restrict__ bp = below + 2ull * i;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < below_ncols) {
      m[k] = bp[(uint64_t)k * L];
      m[8 + k] = bp[(uint64_t)k * L + 1];
    } else {
      m[k] = 0u;
      m[8 + k] = 0u;
    }
  }
}
// start node i = H(H(leaf 2i) || H(leaf 2i+1) || own columns) from the raw leaf words of merkle_load_below
LMN_D void merkle_hash_below(const MerkleSegs& sg, int ncols, uint32_t size, int below_ncols, uint32_t i, const uint32_t m[16],
                             uint32_t h[8]) {
  uint32_t ml[16], mr[16], hl[8], hr[8];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    ml[k] = k < 8 ? m[k] : 0u;
    mr[k] = k < 8 ? m[8 + k] : 0u;
  }
  b2_compress_fresh(hl, ml, 4u * (uint32_t)below_ncols);
  b2_compress_fresh(hr, mr, 4u * (uint32_t)below_ncols);
  uint32_t m2[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    m2[k] = hl[k];
    m2[8 + k] = hr[k];
  }
  // merkle_hash_from only asks whether the node HAS children; their hashes are in m2
  merkle_hash_from(reinterpret_cast<const uint32_t*>(sg.base[0]), sg, ncols, size, i, m2, h);
}

LMN_D void store_hash(uint32_t* __restrict__ o, const uint32_t h[8]) {
  uint4* o4 = reinterpret_cast<uint4*>(o);
  o4[0] = make_uint4(h[0], h[1], h[2], h[3]);
  o4[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// Quad-cooperative Blake2s of one 64-byte final block (a Merkle parent): lane q of a quad owns column q
// of the 4x4 state (a, b, c, d = rows), so the four G functions of a half-round run on four lanes; the
// diagonal step rotates rows b, c, d by 1, 2, 3 lanes with DPP quad_perm.  Message words are fetched from
// LDS by per-lane sigma offsets.  ~400 issue slots instead of ~1000: single-hash latency 0.93 us vs 2.0 us
// on MI355X (tools/microbench4.hip) - used where a level is too narrow to fill lanes anyway.
// Returns words q and 4+q of the digest.  Must be executed by all four lanes of the quad.
// WS = 0: the 16 message words are contiguous at msg; WS > 0: the two child hashes live in a word-major node array
// (word k of node j at base[k * WS + j]): message word w = msg[(w & 7) * WS + (w >> 3)] with msg = base + 2 * parent.
template <uint32_t WS = 0u>
LMN_D void b2_quad_parent(const uint32_t* msg, uint32_t q, uint32_t& o_lo, uint32_t& o_hi, uint32_t t0 = 64u) {
#define LMN_B2_QW(w) (WS ? msg[((w) & 7u) * WS + ((w) >> 3)] : msg[(w)])
#ifdef LMN_EMU
  // CPU emulation (tests only): a cross-lane rendezvous per DPP move would be a block-wide fiber switch;
  // every lane hashes the block alone and keeps its two words.  The DPP path is checked on the GPU.
  uint32_t h[8], m[16];
  for (uint32_t k = 0; k < 16; ++k) m[k] = LMN_B2_QW(k);
  b2_compress_fresh(h, m, t0);
  o_lo = h[q];
  o_hi = h[4 + q];
  return;
#endif
  const uint32_t iv_lo = q == 0 ? 0x6A09E667u : q == 1 ? 0xBB67AE85u : q == 2 ? 0x3C6EF372u : 0xA54FF53Au;
  const uint32_t iv_hi = q == 0 ? 0x510E527Fu : q == 1 ? 0x9B05688Cu : q == 2 ? 0x1F83D9ABu : 0x5BE0CD19u;
  const uint32_t h_lo = q == 0 ? (0x6A09E667u ^ 0x01010020u) : iv_lo;
  uint32_t a = h_lo, b = iv_hi, c = iv_lo, d = iv_hi ^ (q == 0 ? t0 : q == 2 ? 0xffffffffu : 0u);
  const uint32_t sh8 = 8u * q;
#define LMN_B2_QUAD_ROUND(...)                                            \
  {                                                                       \
    const uint64_t S = LMN_B2_SIGMA_PACK(__VA_ARGS__);                    \
    const uint32_t lo = (uint32_t)(S >> sh8), hi = (uint32_t)(S >> (32u + sh8)); \
    const uint32_t x0 = LMN_B2_QW(lo & 15u), y0 = LMN_B2_QW((lo >> 4) & 15u); \
    const uint32_t x1 = LMN_B2_QW(hi & 15u), y1 = LMN_B2_QW((hi >> 4) & 15u); \
    LMN_B2_G(a, b, c, d, x0, y0)                                          \
    b = lmn_quad_perm(b, 0x39);                                           \
    c = lmn_quad_perm(c, 0x4E);                                           \
    d = lmn_quad_perm(d, 0x93);                                           \
    LMN_B2_G(a, b, c, d, x1, y1)                                          \
    b = lmn_quad_perm(b, 0x93);                                           \
    c = lmn_quad_perm(c, 0x4E);                                           \
    d = lmn_quad_perm(d, 0x39);                                           \
  }
  LMN_B2_QUAD_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  LMN_B2_QUAD_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  LMN_B2_QUAD_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  LMN_B2_QUAD_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  LMN_B2_QUAD_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  LMN_B2_QUAD_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  LMN_B2_QUAD_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  LMN_B2_QUAD_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  LMN_B2_QUAD_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  LMN_B2_QUAD_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef LMN_B2_QUAD_ROUND
  o_lo = h_lo ^ a ^ c;
  o_hi = iv_hi ^ b ^ d;
#undef LMN_B2_QW
}

// One level of an in-LDS Merkle climb.  The block's nodes live WORD-MAJOR in sh (word k of node j at sh[k * BLOCK + j]:
// lanes that walk nodes touch consecutive banks - the node-major form, 8 or 16 words per lane, put a whole wave on two
// banks): the children of this block's `n_par` parents are nodes 0 .. 2 * n_par - 1, parent j replaces node j and is
// written to out[(node0 + j)*8 ..].  Levels with at most 128 parents (two quad-waves per SIMD of the CU) use four lanes
// per hash, which is faster there; wider levels are throughput-bound inside the CU and keep one lane per hash.
// Block-uniform arguments; ends WITHOUT a barrier.
template <int BLOCK>
LMN_D void merkle_lds_level(uint32_t* sh, uint32_t* __restrict__ out, uint32_t node0, uint32_t n_par) {
  __syncthreads();
  if (n_par * 4u <= (uint32_t)BLOCK && n_par <= 128u) {
    const uint32_t g = threadIdx.x >> 2, q = threadIdx.x & 3u;
    const bool on = g < n_par;
    uint32_t o_lo = 0u, o_hi = 0u;
    const bool wave_on = ((threadIdx.x & ~63u) >> 2) < n_par;  // wave-uniform
    if (wave_on) b2_quad_parent<(uint32_t)BLOCK>(sh + 2u * g, q, o_lo, o_hi);
    __syncthreads();
    if (on) {
      sh[q * BLOCK + g] = o_lo;
      sh[(4u + q) * BLOCK + g] = o_hi;
      uint32_t* o = out + (uint64_t)(node0 + g) * 8;
      o[q] = o_lo;
      o[4u + q] = o_hi;
    }
  } else {
    const bool on = threadIdx.x < n_par;
    uint32_t cur[8];
    if (on) {
      uint32_t m[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        m[k] = sh[k * BLOCK + 2u * threadIdx.x];
        m[8 + k] = sh[k * BLOCK + 2u * threadIdx.x + 1u];
      }
      b2_compress_fresh(cur, m, 64u);
      store_hash(out + (uint64_t)(node0 + threadIdx.x) * 8, cur);
    }
    __syncthreads();
    if (on) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[k * BLOCK + threadIdx.x] = cur[k];
    }
  }
}

// LDS climb shared by the Merkle kernels: `lvl_size` hashes of the whole level exist, this block's share
// sits in sh[idx*8..]; levels first..last are produced.
template <int BLOCK>
LMN_D void merkle_lds_climb(uint32_t* sh, const MerkleLevels& outs, int first, int last, uint32_t lvl_size) {
  for (int l = first; l <= last; ++l) {
    lvl_size >>= 1;
    const uint32_t active = (uint32_t)BLOCK >> (l - first + 1);
    const uint32_t node0 = blockIdx.x * active;
    const uint32_t n_par = node0 >= lvl_size ? 0u : (lvl_size - node0 < active ? lvl_size - node0 : active);
    merkle_lds_level<BLOCK>(sh, outs.p[l], node0, n_par);
  }
}

// MODE 1: leaf level of a single-size tree (no child layer, one contiguous run of <= 16 columns: one
// compression per leaf); MODE 2: pure inner level (children only); MODE 0: anything else.  The special modes
// drop the run selection and the multi-block loop from the hot loop.
// ZT: the caller keeps the message words this mode never loads at zero (set once, outside its leaf loop), so they are not
// rewritten for every leaf
template <int MODE, bool ZT = false>
LMN_D void merkle_load_mode(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                            uint32_t i, uint32_t m[16], const MerkleFold& fold = MerkleFold{}) {
  if (MODE == 3) {
    // leaf i of a FRI layer = fold of the pair (2i, 2i+1) of the previous layer (same arithmetic as k_fold)
    const uint64_t L = 2ull * size;
    const uint32_t* __restrict__ sp = fold.src + 2ull * i;
    const QM31 a{sp[0], sp[L], sp[2 * L], sp[3 * L]};
    const QM31 b{sp[1], sp[L + 1], sp[2 * L + 1], sp[3 * L + 1]};
    const QM31 alpha = *fold.alpha;
    QM31 r = q_add(q_add(a, b), q_mul(alpha, q_mul_m(q_sub(a, b), fold.itw[i])));
    if (fold.src2) {   // block-uniform: a quotient column joins this layer (k_fold with accumulate = 1)
      const uint32_t* __restrict__ sq = fold.src2 + 2ull * i;
      const QM31 c{sq[0], sq[L], sq[2 * L], sq[3 * L]};
      const QM31 d{sq[1], sq[L + 1], sq[2 * L + 1], sq[3 * L + 1]};
      const QM31 rc = q_add(q_add(c, d), q_mul(alpha, q_mul_m(q_sub(c, d), fold.itw2[i])));
      r = q_add(q_mul(r, q_mul(alpha, alpha)), rc);
    }
    uint32_t* __restrict__ o = fold.dst + i;
    o[0] = r.a;
    o[(uint64_t)size] = r.b;
    o[2ull * size] = r.c;
    o[3ull * size] = r.d;
    m[0] = r.a;
    m[1] = r.b;
    m[2] = r.c;
    m[3] = r.d;
    if (!ZT) {
#pragma unroll
      for (int k = 4; k < 16; ++k) m[k] = 0u;
    }
  } else if (MODE == 1) {
    const uint32_t* __restrict__ base = sg.base[0] + i;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < ncols)
        m[k] = base[(uint64_t)k * size];
      else if (!ZT)
        m[k] = 0u;
    }
  } else if (MODE == 2) {
    const uint4* p4 = reinterpret_cast<const uint4*>(prev) + (uint64_t)i * 4;
    uint4 a = p4[0], b = p4[1], c = p4[2], d = p4[3];
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
    m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
    m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
    m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
  } else if (MODE == 4) {
    merkle_load_below(fold.below, fold.below_ncols, size, i, m);
  } else {
    merkle_load_first(prev, sg, ncols, size, i, m);
  }
}
template <int MODE>
LMN_D void merkle_hash_mode(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                            uint32_t i, uint32_t m[16], uint32_t h[8], const MerkleFold& fold = MerkleFold{}) {
  if (MODE == 4) {
    merkle_hash_below(sg, ncols, size, fold.below_ncols, i, m, h);
  } else if (MODE == 3) {
    b2_compress_fresh(h, m, 16u);
  } else if (MODE == 1) {
    b2_compress_fresh(h, m, 4u * (uint32_t)ncols);
  } else if (MODE == 2) {
    b2_compress_fresh(h, m, 64u);
  } else {
    merkle_hash_from(prev, sg, ncols, size, i, m, h);
  }
}

template <int MODE>
LMN_KERNEL k_merkle_fused(const uint32_t* __restrict__ prev, MerkleSegs sg, int ncols, uint32_t size,
                          MerkleLevels outs, int sub, int nfused, MerkleFold fold) {
  // Wave-cooperative subtree: in batch j lane l hashes start node W0 + 64*j + l (coalesced column
  // loads and hash stores).  Siblings sit in neighbouring lanes, so after every second batch the
  // lanes swap one hash with lane^1 and ALL 64 lanes compress one level-1 parent (even lanes for the
  // older batch, odd lanes for the newer one); after every fourth batch the same with lane^2, etc.
  // Each lane keeps its pending hash per level in private LDS slots (word 8 = node index).
  LMN_SHARED uint32_t stack[MERKLE_MAX_SUB * 8 * TPB];  // [level][word][thread]
  uint32_t* sh = stack;  // the climb buffer reuses the stack storage once the batch loop is over
  const uint32_t per = 1u << sub;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t W0 = (t >> 6) * (64u << sub);
  uint32_t cur[8];
  uint32_t cur_idx = 0;
  // software pipeline: the next batch's loads are in flight while this batch is compressed (only ~2 waves share a SIMD
  // here, too few to hide HBM latency by occupancy alone).  Two message buffers alternate (the loop body handles an even and
  // an odd batch), so the prefetched words are hashed where they were loaded - no 16-register copy per leaf.
  uint32_t mA[16], mB[16];
  constexpr bool ZT = MODE == 1 || MODE == 3;   // the words beyond a leaf's columns stay zero for the whole kernel
  if (ZT) {
#pragma unroll
    for (int k = 0; k < 16; ++k) mA[k] = mB[k] = 0u;
  }
  merkle_load_mode<MODE, ZT>(prev, sg, ncols, size, W0 + lane, mA, fold);
  auto leaf = [&](uint32_t j, uint32_t (&mc)[16], uint32_t (&mn)[16]) {
    const uint32_t node = W0 + 64u * j + lane;
    cur_idx = node;
    if (j + 1 < per) merkle_load_mode<MODE, ZT>(prev, sg, ncols, size, node + 64u, mn, fold);
    merkle_hash_mode<MODE>(prev, sg, ncols, size, node, mc, cur, fold);
    if (outs.p[0]) store_hash(outs.p[0] + (uint64_t)node * 8, cur);
  };
  for (uint32_t j = 0; j < per; j += 2) {
    leaf(j, mA, mB);
    if (per == 1) break;                       // sub = 0: one leaf per lane, nothing to merge in registers
#pragma unroll
    for (int k = 0; k < 8; ++k) stack[k * TPB + threadIdx.x] = cur[k];   // level-0 slot: the even batch waits for its sibling
    leaf(j + 1, mB, mA);
    uint32_t jj = j + 1;
    int lvl = 0;
    while (jj & 1u) {
      const bool b = ((lane >> lvl) & 1u) != 0u;
      uint32_t m[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t st = stack[(lvl * 8 + k) * TPB + threadIdx.x];
        const uint32_t recv = lmn_shfl_xor(b ? st : cur[k], 1 << lvl);
        m[k] = b ? recv : st;
        m[8 + k] = b ? cur[k] : recv;
      }
      // the pending (older) node of this lane sits exactly 64 nodes before the newer one at every level
      cur_idx = (b ? cur_idx : cur_idx - 64u) >> 1;
      b2_compress_fresh(cur, m, 64u);
      jj >>= 1;
      ++lvl;
      if (outs.p[lvl]) store_hash(outs.p[lvl] + (uint64_t)cur_idx * 8, cur);
    }
    if (lvl < sub) {
#pragma unroll
      for (int k = 0; k < 8; ++k) stack[(lvl * 8 + k) * TPB + threadIdx.x] = cur[k];
    }
  }
  __syncthreads();  // every lane is done with its stack slots before they are overwritten
  {
    const uint32_t local = cur_idx - blockIdx.x * TPB;  // this block owns level-`sub` nodes [b*TPB, (b+1)*TPB)
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[k * TPB + local] = cur[k];   // word-major (merkle_lds_level)
  }
  merkle_lds_climb<TPB>(sh, outs, sub + 1, nfused, size >> sub);
}

LMN_D void chan_draw_words(DevChannel* ch, uint32_t out[8]) {
  uint32_t m[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = ch->digest[k];
  m[8] = ch->n_sent;
#pragma unroll
  for (int k = 9; k < 16; ++k) m[k] = 0u;
  b2_init(out);
  // KAT encoding: digest || u64 counter zero-padded to 32 bytes (64-byte message);
  // LMN_PV_DRAW_CTR_U32: digest || u32 counter || 0x00 (37-byte message)
  b2_compress(out, m, ch->variant == 0u ? 64u : 37u, 0xffffffffu);
  ch->n_sent += 1u;
}

// digest <- H(digest || root); alpha <- draw_felt(); executed by ONE lane
LMN_D void chan_mix_root_draw(DevChannel* ch, const uint32_t* root, QM31* out_alpha, uint32_t* root_copy) {
  uint32_t m[16], h[8];
  for (int k = 0; k < 8; ++k) {
    m[k] = ch->digest[k];
    m[8 + k] = root[k];
    root_copy[k] = root[k];
  }
  b2_compress_fresh(h, m, 64u);
  for (int k = 0; k < 8; ++k) ch->digest[k] = h[k];
  ch->n_sent = 0u;
  for (;;) {
    uint32_t w[8];
    chan_draw_words(ch, w);
    bool ok = true;
    for (int k = 0; k < 8; ++k) ok = ok && (w[k] < 2u * P31);
    if (!ok) continue;
    QM31 a;
    a.a = w[0] >= P31 ? w[0] - P31 : w[0];
    a.b = w[1] >= P31 ? w[1] - P31 : w[1];
    a.c = w[2] >= P31 ? w[2] - P31 : w[2];
    a.d = w[3] >= P31 ? w[3] - P31 : w[3];
    *out_alpha = a;
    break;
  }
}

// Block-cooperative form of chan_mix_root_draw for kernels that already hold the root in LDS: called by ALL
// threads of the block (block-uniform control flow); the hashing runs on the first quad with the
// quad-cooperative Blake2s (0.9 us per hash instead of 2 us).  `scratch`: 28 words of LDS that do not
// overlap the root's 8 words.  Ends with a barrier and returns the drawn alpha to every thread.
LMN_D QM31 chan_mix_root_draw_block(DevChannel* ch, const uint32_t* root, uint32_t root_stride, uint32_t* scratch,
                                    QM31* out_alpha, uint32_t* root_copy) {
  uint32_t* msg = scratch;       // 16 words: digest || root, then digest || counter
  uint32_t* wbuf = scratch + 16;  // 8 words: drawn words
  const uint32_t tid = threadIdx.x, q = tid & 3u;
  const uint32_t t_draw = ch->variant == 0u ? 64u : 37u;
  __syncthreads();
  if (tid < 8u) {
    msg[tid] = ch->digest[tid];
    const uint32_t r = root[tid * root_stride];   // word k of the root at root[k * root_stride] (word-major node array)
    msg[8u + tid] = r;
    root_copy[tid] = r;
  }
  __syncthreads();
  uint32_t lo = 0u, hi = 0u;
  if (tid < 64u) b2_quad_parent(msg, q, lo, hi);
  __syncthreads();
  if (tid < 4u) {
    msg[q] = lo;
    msg[4u + q] = hi;
    ch->digest[q] = lo;
    ch->digest[4u + q] = hi;
  }
  for (uint32_t n_sent = 0;; ++n_sent) {
    if (tid >= 8u && tid < 16u) msg[tid] = tid == 8u ? n_sent : 0u;
    __syncthreads();
    if (tid < 64u) b2_quad_parent(msg, q, lo, hi, t_draw);
    if (tid < 4u) {
      wbuf[q] = lo;
      wbuf[4u + q] = hi;
    }
    __syncthreads();
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) ok = ok && (wbuf[k] < 2u * P31);
    if (ok) {  // block-uniform
      if (tid == 0u) {
        QM31 a;
        a.a = wbuf[0] >= P31 ? wbuf[0] - P31 : wbuf[0];
        a.b = wbuf[1] >= P31 ? wbuf[1] - P31 : wbuf[1];
        a.c = wbuf[2] >= P31 ? wbuf[2] - P31 : wbuf[2];
        a.d = wbuf[3] >= P31 ? wbuf[3] - P31 : wbuf[3];
        *out_alpha = a;
        ch->n_sent = n_sent + 1u;
        scratch[24] = a.a;
        scratch[25] = a.b;
        scratch[26] = a.c;
        scratch[27] = a.d;
      }
      break;
    }
    __syncthreads();  // everyone has read wbuf before the next draw overwrites it
  }
  __syncthreads();
  return QM31{scratch[24], scratch[25], scratch[26], scratch[27]};
}

// Small trees / tree tops: one node per lane, one block of up to 1024 lanes, up to 10 LDS levels.
constexpr int MERKLE_SMALL_BLOCK = 1024;
template <int MODE>
LMN_KERNEL k_merkle_small(const uint32_t* __restrict__ prev, MerkleSegs sg, int ncols, uint32_t size,
                          MerkleLevels outs, int nfused, DevChannel* ch, QM31* alpha_out, uint32_t* root_copy) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED uint32_t sh[MERKLE_SMALL_BLOCK * 8];
  const uint32_t i = threadIdx.x;
  uint32_t cur[8];
  if (i < size) {
    uint32_t m[16];
    merkle_load_mode<MODE>(prev, sg, ncols, size, i, m);
    merkle_hash_mode<MODE>(prev, sg, ncols, size, i, m, cur);
    store_hash(outs.p[0] + (uint64_t)i * 8, cur);
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[k * MERKLE_SMALL_BLOCK + i] = cur[k];   // word-major (merkle_lds_level)
  }
  LMN_SERIAL_KERNEL();  // the leaf compression left the wave at its low phase priority
  merkle_lds_climb<MERKLE_SMALL_BLOCK>(sh, outs, 1, nfused, size);
  // when this launch produced the root, it can also run the device-resident Fiat-Shamir step
  if (ch != nullptr && (size >> nfused) == 1u)
    chan_mix_root_draw_block(ch, sh, MERKLE_SMALL_BLOCK, sh + 16, alpha_out, root_copy);
}

void launch_merkle_fused(const uint32_t* prev, const MerkleSegs& sg, int ncols, uint32_t size,
                         const MerkleLevels& outs, int sub, int nfused, lmn_stream_t s, const MerkleFold* fold) {
  if (LMN_ABLATED(1u)) return;
  if (!prev && ncols == 0) throw LmnError(-100, "merkle level with no input");
  if (nfused > MERKLE_MAX_FUSED || sub > MERKLE_MAX_SUB || sub > nfused || nfused - sub > 8 ||
      size % ((uint32_t)TPB << sub) != 0)
    throw LmnError(-100, "merkle_fused: bad arguments");
  const dim3 g(cdiv(size >> sub, TPB)), b(TPB);
  const MerkleFold none{};
  // a null p[l] (l < sub only: the levels a lane keeps in registers) is a level the caller does not want written
  for (int l = sub; l <= nfused; ++l)
    if (!outs.p[l]) throw LmnError(-100, "merkle_fused: only the per-lane levels may be left unwritten");
  if (fold && fold->below) {
    if (prev || fold->src || ncols < 1 || fold->below_ncols < 1 || fold->below_ncols > 8)
      throw LmnError(-100, "merkle_fused: a start level over its own leaf level has columns, no stored children and <= 8 leaf columns");
    LMN_LAUNCH(k_merkle_fused<4>, g, b, 0, s, prev, sg, ncols, size, outs, sub, nfused, *fold);
  } else if (fold) {
    if (prev || ncols != 4) throw LmnError(-100, "merkle_fused: a folded level is a leaf level of 4 coordinate columns");
    LMN_LAUNCH(k_merkle_fused<3>, g, b, 0, s, prev, sg, ncols, size, outs, sub, nfused, *fold);
  } else if (!prev && ncols <= 16 && sg.n[0] == ncols)
    LMN_LAUNCH(k_merkle_fused<1>, g, b, 0, s, prev, sg, ncols, size, outs, sub, nfused, none);
  else if (prev && ncols == 0)
    LMN_LAUNCH(k_merkle_fused<2>, g, b, 0, s, prev, sg, ncols, size, outs, sub, nfused, none);
  else
    LMN_LAUNCH(k_merkle_fused<0>, g, b, 0, s, prev, sg, ncols, size, outs, sub, nfused, none);
}

void launch_merkle_small(const uint32_t* prev, const MerkleSegs& sg, int ncols, uint32_t size,
                         const MerkleLevels& outs, int nfused, DevChannel* ch, QM31* alpha_out, uint32_t* root_copy,
                         lmn_stream_t s) {
  if (!prev && ncols == 0) throw LmnError(-100, "merkle level with no input");
  if (size > (uint32_t)MERKLE_SMALL_BLOCK || nfused > 10) throw LmnError(-100, "merkle_small: bad arguments");
  const dim3 g(1), b(MERKLE_SMALL_BLOCK);
  if (!prev && ncols <= 16 && sg.n[0] == ncols)
    LMN_LAUNCH(k_merkle_small<1>, g, b, 0, s, prev, sg, ncols, size, outs, nfused, ch, alpha_out, root_copy);
  else if (prev && ncols == 0)
    LMN_LAUNCH(k_merkle_small<2>, g, b, 0, s, prev, sg, ncols, size, outs, nfused, ch, alpha_out, root_copy);
  else
    LMN_LAUNCH(k_merkle_small<0>, g, b, 0, s, prev, sg, ncols, size, outs, nfused, ch, alpha_out, root_copy);
}

// =============================================================================================
// Device-resident Fiat-Shamir steps for the FRI commit loop (no host round trip per layer)
// =============================================================================================
LMN_KERNEL k_chan_mix_root_draw(DevChannel* ch, const uint32_t* __restrict__ root, QM31* out_alpha,
                                uint32_t* root_copy) {
  LMN_SERIAL_KERNEL();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  chan_mix_root_draw(ch, root, out_alpha, root_copy);
}

void launch_chan_mix_root_draw(DevChannel* ch, const uint32_t* root, QM31* out_alpha, uint32_t* root_copy,
                               lmn_stream_t s) {
  LMN_LAUNCH(k_chan_mix_root_draw, dim3(1), dim3(64), 0, s, ch, root, out_alpha, root_copy);
}

// =============================================================================================
// FRI tail: all layers of size <= 1024 in ONE single-block launch: per layer Merkle-commit the
// line evaluation (LDS tree), mix the root into the device-resident channel, draw the folding
// alpha, fold.  Every layer's evaluations and tree levels still go to HBM for decommitment.
// =============================================================================================
LMN_KERNEL k_fri_tail(DevChannel* ch, const FriTailLayer* __restrict__ layers, int n_layers, int first_log,
                      QM31* alphas_out, uint32_t* roots_out) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED uint32_t sh[MERKLE_SMALL_BLOCK * 8];
  const uint32_t i = threadIdx.x;
  for (int li = 0; li < n_layers; ++li) {
    const FriTailLayer ly = layers[li];
    const int L = first_log - li;
    const uint32_t size = 1u << L;
    uint32_t cur[8];
    if (i < size) {
      uint32_t m[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) m[k] = 0u;
      m[0] = ly.vals[i];
      m[1] = ly.vals[size + i];
      m[2] = ly.vals[2 * size + i];
      m[3] = ly.vals[3 * size + i];
      b2_compress_fresh(cur, m, 16u);
      store_hash(ly.merkle[L] + (uint64_t)i * 8, cur);
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[k * MERKLE_SMALL_BLOCK + i] = cur[k];   // word-major (merkle_lds_level)
    }
    for (int l = L - 1; l >= 0; --l) merkle_lds_level<MERKLE_SMALL_BLOCK>(sh, ly.merkle[l], 0u, 1u << l);
    const QM31 alpha = chan_mix_root_draw_block(ch, sh, MERKLE_SMALL_BLOCK, sh + 16, &alphas_out[li], roots_out + li * 8);
    const uint32_t n = size >> 1;
    if (i < n) {
      QM31 a{ly.vals[2 * i], ly.vals[size + 2 * i], ly.vals[2 * size + 2 * i], ly.vals[3 * size + 2 * i]};
      QM31 b{ly.vals[2 * i + 1], ly.vals[size + 2 * i + 1], ly.vals[2 * size + 2 * i + 1],
             ly.vals[3 * size + 2 * i + 1]};
      QM31 f0 = q_add(a, b);
      QM31 f1 = q_mul_m(q_sub(a, b), ly.itw[i]);
      QM31 r = q_add(f0, q_mul(alpha, f1));
      ly.next[i] = r.a;
      ly.next[n + i] = r.b;
      ly.next[2 * n + i] = r.c;
      ly.next[3 * n + i] = r.d;
    }
    __syncthreads();
  }
}

void launch_fri_tail(DevChannel* ch, const FriTailLayer* layers, int n_layers, int first_log, QM31* alphas_out,
                     uint32_t* roots_out, lmn_stream_t s) {
  if (first_log > 10 || n_layers < 1 || n_layers > first_log) throw LmnError(-100, "fri_tail: bad arguments");
  LMN_LAUNCH(k_fri_tail, dim3(1), dim3(MERKLE_SMALL_BLOCK), 0, s, ch, layers, n_layers, first_log, alphas_out,
             roots_out);
}

// =============================================================================================
// a11  twiddle tables: the points of a half coset by double-and-add from the step's doublings (<= 26 group additions per
// entry) instead of a serial walk on the host - 2^26 entries and their inverses in milliseconds
// =============================================================================================
LMN_KERNEL k_twiddles(int bits, TwGen g, int coord, uint32_t* __restrict__ tw, uint32_t* __restrict__ itw,
                      uint32_t* __restrict__ tw2, uint32_t* __restrict__ itw2) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= (1u << bits)) return;
  const uint32_t j = bits ? (__brev(h) >> (32 - bits)) : 0u;
  uint32_t x = g.ix, y = g.iy;
  for (int k = 0; k < bits; ++k) {
    if ((j >> k) & 1u) {
      const uint32_t nx = m_sub(m_mul(x, g.sx[k]), m_mul(y, g.sy[k]));
      y = m_add(m_mul(x, g.sy[k]), m_mul(y, g.sx[k]));
      x = nx;
    }
  }
  const uint32_t v = coord ? x : y, vi = m_inv(v);
  tw[h] = v;
  itw[h] = vi;
  tw2[h] = 2u * v;
  itw2[h] = 2u * vi;
}

void launch_twiddles(int bits, const TwGen& g, int coord, uint32_t* tw, uint32_t* itw, uint32_t* tw2, uint32_t* itw2,
                     lmn_stream_t s) {
  if (bits < 0 || bits > 29) throw LmnError(-100, "twiddles: bad size");
  LMN_LAUNCH(k_twiddles, dim3(cdiv(1ull << bits, TPB)), dim3(TPB), 0, s, bits, g, coord, tw, itw, tw2, itw2);
}

// =============================================================================================
// gather
// =============================================================================================
LMN_KERNEL k_gather(const uint32_t* __restrict__ arena, const GatherEntry* __restrict__ entries, uint32_t n,
                    const MerkleRecompute* __restrict__ jobs, uint32_t n_jobs, uint32_t* __restrict__ out) {
  LMN_SERIAL_KERNEL();
  uint32_t e = blockIdx.x;
  if (e < n) {
    GatherEntry g = entries[e];
    for (uint32_t k = threadIdx.x; k < g.len; k += blockDim.x) out[g.dst_off + k] = arena[g.src_off + k];
    return;
  }
  e -= n;
  if (e >= n_jobs) return;
  // A tree node the fused launch kept in registers only: lane q of a quad hashes start node (node << depth) + q, the
  // quad reduces the 2^depth hashes pairwise (every quad of the block does the same; block-uniform control flow).
  const MerkleRecompute j = jobs[e];
  const uint32_t q = threadIdx.x & 3u;
  uint32_t h[8];
  const uint32_t i0 = (j.node << j.depth) + (q & ((1u << j.depth) - 1u));
  if (j.below) {
    uint32_t mb[16];
    merkle_load_below(j.below, j.below_ncols, j.size, i0, mb);
    merkle_hash_below(j.sg, j.ncols, j.size, j.below_ncols, i0, mb, h);
  } else {
    merkle_hash_start(j.prev, j.sg, j.ncols, j.size, i0, h);
  }
  for (int s = 0; s < j.depth; ++s) {
    const bool hi = ((q >> s) & 1u) != 0u;
    uint32_t m[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t other = lmn_shfl_xor(h[k], 1 << s);
      m[k] = hi ? other : h[k];
      m[8 + k] = hi ? h[k] : other;
    }
    b2_compress_fresh(h, m, 64u);
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) out[j.dst_off + k] = h[k];
  }
}

void launch_gather(const uint32_t* arena, const GatherEntry* entries, uint32_t n_entries, const MerkleRecompute* jobs,
                   uint32_t n_jobs, uint32_t* out, lmn_stream_t s) {
  if (n_entries + n_jobs == 0) return;
  LMN_LAUNCH(k_gather, dim3(n_entries + n_jobs), dim3(64), 0, s, arena, entries, n_entries, jobs, n_jobs, out);
}

// =============================================================================================
// a6  LogUp: S_j[r] = S_{j-1}[r] + mult_j[r] / (val_j[r] + alpha*id_j[r] - z)
// =============================================================================================
int logup_num_blocks(uint32_t n) { return (int)cdiv(n, TPB); }

template <int K>
LMN_KERNEL k_logup_fracs(LogupArgs a) {
  LMN_SHARED uint64_t red[TPB * 4];
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  QM31 S = q_zero();
  if (r < a.n) {
    QM31 den[K], pre[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      QM31 d = q_from_m(a.val[j][r]);
      if (a.id[j]) d = q_add(d, q_mul_m(a.alpha[j], a.id[j][r]));
      d = q_sub(d, a.z[j]);
      den[j] = d;
      pre[j] = j == 0 ? d : q_mul(pre[j - 1], d);
    }
    QM31 inv = q_inv(pre[K - 1]);
    QM31 invs[K];
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
      invs[j] = j == 0 ? inv : q_mul(inv, pre[j - 1]);
      inv = q_mul(inv, den[j]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint32_t mlt = a.mult[j][r];
      if (a.neg[j]) mlt = m_neg(mlt);
      S = q_add(S, q_mul_m(invs[j], mlt));
      if (j < K - 1) {
        uint32_t* o = a.inter + (uint64_t)(4 * j) * a.n + r;
        o[0] = S.a;
        o[(uint64_t)a.n] = S.b;
        o[(uint64_t)2 * a.n] = S.c;
        o[(uint64_t)3 * a.n] = S.d;
      }
    }
    a.last_tmp[r] = S;
  }
  red[threadIdx.x * 4 + 0] = S.a;
  red[threadIdx.x * 4 + 1] = S.b;
  red[threadIdx.x * 4 + 2] = S.c;
  red[threadIdx.x * 4 + 3] = S.d;
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] += red[(threadIdx.x + st) * 4 + k];
    __syncthreads();
  }
  if (threadIdx.x < 4) a.partials[blockIdx.x * 4 + threadIdx.x] = m_red64(red[threadIdx.x]);
}

void launch_logup_fracs(const LogupArgs& a, lmn_stream_t s) {
  if (LMN_ABLATED(32u)) return;
  dim3 g(logup_num_blocks(a.n)), b(TPB);
  switch (a.k) {
    case 1: LMN_LAUNCH(k_logup_fracs<1>, g, b, 0, s, a); break;
    case 2: LMN_LAUNCH(k_logup_fracs<2>, g, b, 0, s, a); break;
    case 3: LMN_LAUNCH(k_logup_fracs<3>, g, b, 0, s, a); break;
    case 7: LMN_LAUNCH(k_logup_fracs<7>, g, b, 0, s, a); break;
    default: throw LmnError(-100, "logup: unsupported relation count");
  }
}

LMN_KERNEL k_logup_reduce(const uint32_t* __restrict__ partials, int nblocks, uint32_t n_inv, QM31* out) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED uint64_t red[TPB * 4];
  uint64_t acc[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
    for (int k = 0; k < 4; ++k) acc[k] += partials[b * 4 + k];
  for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] = m_red64(acc[k]);
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] += red[(threadIdx.x + st) * 4 + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    QM31 c{m_red64(red[0]), m_red64(red[1]), m_red64(red[2]), m_red64(red[3])};
    out[0] = c;
    out[1] = q_mul_m(c, n_inv);
  }
}

void launch_logup_reduce(const uint32_t* partials, int nblocks, uint32_t n_inv, QM31* claimed_out, lmn_stream_t s) {
  LMN_LAUNCH(k_logup_reduce, dim3(1), dim3(TPB), 0, s, partials, nblocks, n_inv, claimed_out);
}

// coset-order position -> storage index (bit-reversed circle-domain order), SURVEY Appendix A.2
LMN_D uint32_t coset_pos_to_storage(uint32_t i, int log_size) {
  uint32_t n = 1u << log_size;
  uint32_t cd = (i & 1u) ? n - ((i + 1u) >> 1) : (i >> 1);
  return log_size == 0 ? 0u : (__brev(cd) >> (32 - log_size));
}

constexpr int SCAN_PER_THREAD = 4;
constexpr int SCAN_PER_BLOCK = TPB * SCAN_PER_THREAD;
int logup_scan_num_blocks(int log_size);

// block-local inclusive scan of thread sums in LDS (Hillis-Steele over TPB QM31 values)
LMN_D QM31 block_scan_inclusive(QM31 v, QM31* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < TPB; off <<= 1) {
    QM31 add = q_zero();
    if ((int)threadIdx.x >= off) add = sh[threadIdx.x - off];
    __syncthreads();
    sh[threadIdx.x] = q_add(sh[threadIdx.x], add);
    __syncthreads();
  }
  return sh[threadIdx.x];
}

// mode 0: write block totals; mode 1: write scanned values (+ exclusive block offsets)
LMN_KERNEL k_logup_scan(const QM31* __restrict__ last_tmp, const QM31* __restrict__ claimed_shift, int log_size,
                        uint32_t* __restrict__ out_cols, QM31* blocksums, int mode) {
  LMN_SHARED QM31 sh[TPB];
  const uint32_t n = 1u << log_size;
  const QM31 shift = claimed_shift[1];
  uint32_t i0 = (blockIdx.x * TPB + threadIdx.x) * SCAN_PER_THREAD;
  QM31 v[SCAN_PER_THREAD];
  uint32_t st[SCAN_PER_THREAD];
  QM31 run = q_zero();
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    uint32_t i = i0 + k;
    if (i < n) {
      st[k] = coset_pos_to_storage(i, log_size);
      run = q_add(run, q_sub(last_tmp[st[k]], shift));
    } else {
      st[k] = 0xffffffffu;
    }
    v[k] = run;
  }
  QM31 incl = block_scan_inclusive(run, sh);
  if (mode == 0) {
    if (threadIdx.x == TPB - 1) blocksums[blockIdx.x] = incl;
    return;
  }
  QM31 offset = q_sub(incl, run);  // exclusive prefix of this thread within the block
  if (blockIdx.x > 0) offset = q_add(offset, blocksums[blockIdx.x - 1]);
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    if (st[k] == 0xffffffffu) continue;
    QM31 t = q_add(v[k], offset);
    uint32_t* o = out_cols + st[k];
    o[0] = t.a;
    o[(uint64_t)n] = t.b;
    o[(uint64_t)2 * n] = t.c;
    o[(uint64_t)3 * n] = t.d;
  }
}

// inclusive scan of the block totals in place (single block of up to 1024 lanes; lane t owns a contiguous run)
constexpr int SCAN_SUMS_THREADS = 1024;
LMN_KERNEL k_scan_blocksums(QM31* blocksums, int nblocks) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED QM31 sh[SCAN_SUMS_THREADS];
  const int T = (int)blockDim.x;
  const int per = (nblocks + T - 1) / T;
  const int b0 = threadIdx.x * per;
  // pass 1: the lane's total (loads in independent batches of 8, so that they overlap)
  QM31 run = q_zero();
  for (int k0 = 0; k0 < per; k0 += 8) {
    QM31 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = b0 + k0 + j;
      v[j] = (k0 + j < per && b < nblocks) ? blocksums[b] : q_zero();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) run = q_add(run, v[j]);
  }
  // Hillis-Steele over the lanes' totals
  sh[threadIdx.x] = run;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {
    QM31 add = q_zero();
    if ((int)threadIdx.x >= off) add = sh[threadIdx.x - off];
    __syncthreads();
    sh[threadIdx.x] = q_add(sh[threadIdx.x], add);
    __syncthreads();
  }
  // pass 2: inclusive prefix inside the lane's run, starting from the lanes before it
  QM31 acc = q_sub(sh[threadIdx.x], run);
  for (int k0 = 0; k0 < per; k0 += 8) {
    QM31 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = b0 + k0 + j;
      v[j] = (k0 + j < per && b < nblocks) ? blocksums[b] : q_zero();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = b0 + k0 + j;
      acc = q_add(acc, v[j]);
      if (k0 + j < per && b < nblocks) blocksums[b] = acc;
    }
  }
}

// ---- coalesced coset-order scan (log_size >= SCAN2_MIN_LOG)
// Coset positions 2m and 2m+1 hold circle-domain indices m and n-1-m = ~m, i.e. storage indices s = brev(m) (even)
// and ~s (odd).  With m = mh * 2^A + ml, u = s >> 1 = brev_A(ml) << (k-1-A) | brev(mh): a block of 2^A consecutive
// m (fixed mh) is a stride-2^(k-1-A) comb in storage.  A workgroup therefore takes the 2 * 2^C blocks whose brev(mh)
// is (G << C) | x for G in {g, ~g} and all x < 2^C: their even elements are the even words of 2^A contiguous runs
// of 2^(C+1) storage indices in region g, their odd partners the odd words of the runs in region ~g, and vice versa -
// every line the workgroup touches is used completely, loads (AoS QM31) and stores (4 SoA columns) are contiguous
// runs of 2^(C+1) elements.  Three launches: block totals, scan of the 2^(k-1-A) totals, prefix + write
// (48 B per row of HBM traffic for 32 B of algorithmic bytes; the scattered version moved ~160 B per row).
constexpr int SCAN2_A = 6, SCAN2_C = 4;
constexpr int SCAN2_MIN_LOG = SCAN2_A + SCAN2_C + 2;
constexpr int SCAN2_ELEMS = 2 << (SCAN2_A + SCAN2_C + 1);  // QM31 values per workgroup (4096 = 64 KB)
static_assert(SCAN2_ELEMS == TPB * 16 && (1 << (SCAN2_C + 1)) * 8 == TPB, "one 8-lane group per block of positions");

template <int MODE>
LMN_KERNEL k_logup_scan2(const QM31* __restrict__ last_tmp, const QM31* __restrict__ claimed_shift, int log_size,
                         uint32_t* __restrict__ out_cols, QM31* __restrict__ blocksums) {
  LMN_SERIAL_KERNEL();
  LMN_DYN_SMEM(QM31, T);
  constexpr int A = SCAN2_A, C = SCAN2_C;
  const int gbits = log_size - 1 - A - C;                 // bits of the region index G
  const uint32_t n = 1u << log_size;
  const uint32_t g = blockIdx.x, gmask = (1u << gbits) - 1u;
  const QM31 shift = claimed_shift[1];
  // element e of the tile: ((Gi * 2^A + r) * 2^C + x) * 2 + parity  <->  storage 2u + parity,
  // u = r << (k-1-A) | G << C | x   (r = brev_A(ml))
  auto storage_of = [&](uint32_t e) {
    const uint32_t par = e & 1u, x = (e >> 1) & ((1u << C) - 1u), r = (e >> (1 + C)) & ((1u << A) - 1u), gi = e >> (1 + C + A);
    const uint32_t G = gi ? (~g & gmask) : g;
    const uint32_t u = (r << (log_size - 1 - A)) | (G << C) | x;
    return 2u * u + par;
  };
  for (int i = 0; i < 16; ++i) {
    const uint32_t e = (uint32_t)i * TPB + threadIdx.x;
    T[e] = q_sub(last_tmp[storage_of(e)], shift);
  }
  __syncthreads();
  // 8 lanes per block of 2^A positions pairs; lane `part` owns m = part*8 .. part*8+7 (16 elements)
  const uint32_t blk = threadIdx.x >> 3, part = threadIdx.x & 7u;
  const uint32_t gi = blk >> C, x = blk & ((1u << C) - 1u);
  uint32_t slot[16];
  QM31 v[16];
  QM31 run = q_zero();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t ml = part * 8u + (uint32_t)j;
    const uint32_t r = __brev(ml) >> (32 - A);
    slot[2 * j] = (((gi << A) + r) << C | x) << 1;                                                  // storage 2u
    slot[2 * j + 1] = ((((1u - gi) << A) + ((1u << A) - 1u - r)) << C | ((1u << C) - 1u - x)) << 1 | 1u;  // storage ~(2u)
    run = q_add(run, T[slot[2 * j]]);
    v[2 * j] = run;
    run = q_add(run, T[slot[2 * j + 1]]);
    v[2 * j + 1] = run;
  }
  // inclusive scan of the 8 lanes' sums (xor butterfly inside the 8-lane group)
  QM31 pre = run, tot = run;
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    QM31 o;
    o.a = lmn_shfl_xor(tot.a, d);
    o.b = lmn_shfl_xor(tot.b, d);
    o.c = lmn_shfl_xor(tot.c, d);
    o.d = lmn_shfl_xor(tot.d, d);
    if (part & (uint32_t)d) pre = q_add(pre, o);
    tot = q_add(tot, o);
  }
  const uint32_t G = gi ? (~g & gmask) : g;
  const uint32_t mh = __brev((G << C) | x) >> (32 - (gbits + C));
  if (MODE == 0) {
    if (part == 0) blocksums[mh] = tot;
    return;
  }
  QM31 off = q_sub(pre, run);
  if (mh > 0) off = q_add(off, blocksums[mh - 1]);
#pragma unroll
  for (int j = 0; j < 16; ++j) T[slot[j]] = q_add(v[j], off);
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    const uint32_t e = (uint32_t)i * TPB + threadIdx.x;
    const QM31 t = T[e];
    uint32_t* o = out_cols + storage_of(e);
    o[0] = t.a;
    o[(uint64_t)n] = t.b;
    o[(uint64_t)2 * n] = t.c;
    o[(uint64_t)3 * n] = t.d;
  }
}

int logup_scan_num_blocks(int log_size) {
  if (log_size >= SCAN2_MIN_LOG) return 1 << (log_size - 1 - SCAN2_A);   // one total per block of 2^A position pairs
  return (int)cdiv(1ull << log_size, SCAN_PER_BLOCK);
}

void launch_logup_scan(const QM31* last_tmp, const QM31* claimed_shift, int log_size, uint32_t* out_cols,
                       QM31* blocksums, lmn_stream_t s) {
  static const bool scattered = getenv("LMN_LOGUP_SCAN_V1") != nullptr;   // ablation: the round-1 kernel
  int nb = logup_scan_num_blocks(log_size);
  if (log_size >= SCAN2_MIN_LOG && !scattered) {
    const dim3 grid(1u << (log_size - 2 - SCAN2_A - SCAN2_C));
    const size_t smem = (size_t)SCAN2_ELEMS * sizeof(QM31);
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
    allow_big_lds((const void*)k_logup_scan2<0>, (int)smem);
    allow_big_lds((const void*)k_logup_scan2<1>, (int)smem);
#endif
    LMN_LAUNCH(k_logup_scan2<0>, grid, dim3(TPB), smem, s, last_tmp, claimed_shift, log_size, out_cols, blocksums);
    LMN_LAUNCH(k_scan_blocksums, dim3(1), dim3(nb > 2048 ? SCAN_SUMS_THREADS : TPB), 0, s, blocksums, nb);
    LMN_LAUNCH(k_logup_scan2<1>, grid, dim3(TPB), smem, s, last_tmp, claimed_shift, log_size, out_cols, blocksums);
    return;
  }
  if (scattered) nb = (int)cdiv(1ull << log_size, SCAN_PER_BLOCK);
  LMN_LAUNCH(k_logup_scan, dim3(nb), dim3(TPB), 0, s, last_tmp, claimed_shift, log_size, out_cols, blocksums, 0);
  LMN_LAUNCH(k_scan_blocksums, dim3(1), dim3(nb > 2048 ? SCAN_SUMS_THREADS : TPB), 0, s, blocksums, nb);
  LMN_LAUNCH(k_logup_scan, dim3(nb), dim3(TPB), 0, s, last_tmp, claimed_shift, log_size, out_cols, blocksums, 1);
}

// =============================================================================================
// a7  Constraint quotients (composition polynomial) on the eval domain
// =============================================================================================
// storage index of the point p_s - 2^(eval_log - log_size) coset steps (mask offset -1)
LMN_D uint32_t prev_row_storage(uint32_t s, int eval_log, int log_size) {
  const int hb = eval_log - 1;
  const uint32_t low = s & 1u;
  uint32_t t = s >> 1;
  if (hb == 0) return s;
  const uint32_t mask = (1u << hb) - 1u;
  const uint32_t d = 1u << (eval_log - log_size - 1);
  uint32_t j = __brev(t) >> (32 - hb);
  j = (low ? j + d : j - d) & mask;
  t = __brev(j) >> (32 - hb);
  return (t << 1) | low;
}

// sum_k coeff[k] * constraint_k, accumulated lazily (QAcc: one v_mad_u64_u32 per coordinate for the
// M31-valued local constraints, folded every third term; k is compile-time after unrolling)
struct ConsAcc {
  QAcc acc;
  const QM31* coeff;
  int k;
  LMN_HD void bump() {
    ++k;
    if (k % 3 == 0) qacc_fold(acc);
  }
  LMN_HD void add_m(uint32_t c) {
    qacc_mad(acc, coeff[k], c);
    bump();
  }
  LMN_HD void add_q(QM31 c) {
    const QM31 t = q_mul(coeff[k], c);
    acc.a += t.a;
    acc.b += t.b;
    acc.c += t.c;
    acc.d += t.d;
    bump();
  }
};

template <int NCOLS>
LMN_D void load_row(const uint32_t* __restrict__ base, uint64_t stride, uint32_t s, uint32_t* c) {
#pragma unroll
  for (int k = 0; k < NCOLS; ++k) c[k] = base[(uint64_t)k * stride + s];
}

LMN_D QM31 load_secure(const uint32_t* __restrict__ base, uint64_t stride, uint32_t s) {
  return QM31{base[s], base[stride + s], base[2 * stride + s], base[3 * stride + s]};
}

// logup constraints for NREL relations; values are passed by value (no indexed private arrays:
// those get promoted to LDS and cost occupancy).  rc[j]: 0 = NodeElements (z, alpha); 1 = width-1
// LUT relation val - z2 (range check); 2 = width-2 LUT relation val + alpha2*id - z2 (sin/exp2/log2).
// neg: numerator is -mult.
template <int NREL>
LMN_D void logup_constraints(ConsAcc& ca, const CompositionArgs& a, const uint32_t (&mult)[NREL],
                             const uint32_t (&val)[NREL], const uint32_t (&id)[NREL], const int (&rc)[NREL], bool neg,
                             uint32_t s, uint32_t t, uint64_t E) {
  QM31 prev = q_zero();
#pragma unroll
  for (int j = 0; j < NREL; ++j) {
    QM31 cur = load_secure(a.inter + (uint64_t)(4 * j) * a.stride, a.stride, t);
    QM31 den = rc[j] == 1   ? q_sub(q_from_m(val[j]), a.z2)
               : rc[j] == 2 ? q_sub(q_add_m(q_mul_m(a.alpha2, id[j]), val[j]), a.z2)
                            : q_sub(q_add_m(q_mul_m(a.alpha, id[j]), val[j]), a.z);
    QM31 diff;
    if (j < NREL - 1) {
      diff = q_sub(cur, prev);
    } else {
      uint32_t ps = prev_row_storage(s, a.eval_log, a.log_size);
      QM31 pr = load_secure(a.prev_last, E, ps);
      diff = q_add(q_sub(q_sub(cur, pr), prev), a.claimed_shift[1]);
    }
    ca.add_q(q_sub_m(q_mul(diff, den), neg ? m_neg(mult[j]) : mult[j]));
    prev = cur;
  }
}

template <int KIND>
LMN_KERNEL k_composition(CompositionArgs a) {
  const uint64_t E = 1ull << a.eval_log;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // row inside the block handled by this launch
  if (t >= a.n_rows) return;
  const uint32_t s = a.row0 + t;                             // storage index on the whole eval domain
  ConsAcc ca{qacc_zero(), a.coeff, 0};
  const uint32_t* __restrict__ mn = a.main + t;
  const uint64_t cstride = a.stride;
#define LMN_COL(k) mn[(uint64_t)(k) * cstride]
  if (KIND == 0 || KIND == 1) {
    // Add (15 cols) / Mul (16 cols: rem inserted at 12)
    constexpr bool mul = KIND == 1;
    constexpr int mo = mul ? 13 : 12;
    const uint32_t node = LMN_COL(0), lhs_id = LMN_COL(1), rhs_id = LMN_COL(2), idx = LMN_COL(3), is_last = LMN_COL(4);
    const uint32_t n_node = LMN_COL(5), n_lhs = LMN_COL(6), n_rhs = LMN_COL(7), n_idx = LMN_COL(8);
    const uint32_t lhs = LMN_COL(9), rhs = LMN_COL(10), out = LMN_COL(11);
    const uint32_t m0 = LMN_COL(mo), m1 = LMN_COL(mo + 1), m2 = LMN_COL(mo + 2);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    if (mul) {
      const uint32_t rem = LMN_COL(12);
      ca.add_m(m_sub(m_mul(lhs, rhs), m_add(m_mul(out, 4096u), rem)));
      ca.add_m(0u);  // second eval_fixed_mul slot: zero on rem == 0 (KAT-pinned form)
    } else {
      ca.add_m(m_sub(out, m_add(lhs, rhs)));
    }
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(n_lhs, lhs_id)));
    ca.add_m(m_mul(not_last, m_sub(n_rhs, rhs_id)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[3] = {m0, m1, m2}, rv[3] = {lhs, rhs, out}, ri[3] = {lhs_id, rhs_id, node};
    const int rc[3] = {0, 0, 0};
    logup_constraints<3>(ca, a, rm, rv, ri, rc, false, s, t, E);
  } else if (KIND == 2 || KIND == 7) {
    // Recip / Sqrt (13 cols; the eval_fixed_* forms are unpinned natural identities)
    const uint32_t node = LMN_COL(0), in_id = LMN_COL(1), idx = LMN_COL(2), is_last = LMN_COL(3);
    const uint32_t n_node = LMN_COL(4), n_in = LMN_COL(5), n_idx = LMN_COL(6);
    const uint32_t inp = LMN_COL(7), out = LMN_COL(8), rem = LMN_COL(9), scale = LMN_COL(10);
    const uint32_t m0 = LMN_COL(11), m1 = LMN_COL(12);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    if (KIND == 2)
      ca.add_m(m_sub(m_sqr(scale), m_add(m_mul(inp, out), rem)));
    else
      ca.add_m(m_sub(m_mul(inp, scale), m_add(m_sqr(out), rem)));
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(n_in, in_id)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[2] = {m0, m1}, rv[2] = {inp, out}, ri[2] = {in_id, node};
    const int rc[2] = {0, 0};
    logup_constraints<2>(ca, a, rm, rv, ri, rc, false, s, t, E);
  } else if (KIND == 8) {
    // Rem (16 cols): lhs = rhs*quotient + rem (unpinned form); the out relation carries `rem`
    const uint32_t node = LMN_COL(0), lhs_id = LMN_COL(1), rhs_id = LMN_COL(2), idx = LMN_COL(3), is_last = LMN_COL(4);
    const uint32_t n_node = LMN_COL(5), n_lhs = LMN_COL(6), n_rhs = LMN_COL(7), n_idx = LMN_COL(8);
    const uint32_t lhs = LMN_COL(9), rhs = LMN_COL(10), rem = LMN_COL(11), quo = LMN_COL(12);
    const uint32_t m0 = LMN_COL(13), m1 = LMN_COL(14), m2 = LMN_COL(15);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    ca.add_m(m_sub(lhs, m_add(m_mul(rhs, quo), rem)));
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(n_lhs, lhs_id)));
    ca.add_m(m_mul(not_last, m_sub(n_rhs, rhs_id)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[3] = {m0, m1, m2}, rv[3] = {lhs, rhs, rem}, ri[3] = {lhs_id, rhs_id, node};
    const int rc[3] = {0, 0, 0};
    logup_constraints<3>(ca, a, rm, rv, ri, rc, false, s, t, E);
  } else if (KIND == 13) {
    // LessThan (22 cols; less_than/component.rs:48-185): 9 local constraints, 3 node relations +
    // 4 range-check relations on the 8-bit limbs of diff
    const uint32_t node = LMN_COL(0), lhs_id = LMN_COL(1), rhs_id = LMN_COL(2), idx = LMN_COL(3), is_last = LMN_COL(4);
    const uint32_t n_node = LMN_COL(5), n_lhs = LMN_COL(6), n_rhs = LMN_COL(7), n_idx = LMN_COL(8);
    const uint32_t lhs = LMN_COL(9), rhs = LMN_COL(10), out = LMN_COL(11), diff = LMN_COL(12), borrow = LMN_COL(13);
    const uint32_t l0 = LMN_COL(14), l1 = LMN_COL(15), l2 = LMN_COL(16), l3 = LMN_COL(17);
    const uint32_t m0 = LMN_COL(18), m1 = LMN_COL(19), m2 = LMN_COL(20), md = LMN_COL(21);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    ca.add_m(m_mul(borrow, m_sub(borrow, 1u)));
    ca.add_m(m_sub(out, m_mul(m_sub(1u, borrow), 4096u)));
    ca.add_m(m_sub(m_add(lhs, diff), rhs));  // - borrow * (2^31 - 1), which is 0 in M31
    ca.add_m(m_sub(diff, m_add(m_add(m_mul(l3, 1u << 24), m_mul(l2, 1u << 16)), m_add(m_mul(l1, 1u << 8), l0))));
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(n_lhs, lhs_id)));
    ca.add_m(m_mul(not_last, m_sub(n_rhs, rhs_id)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[7] = {m0, m1, m2, md, md, md, md}, rv[7] = {lhs, rhs, out, l0, l1, l2, l3};
    const uint32_t ri[7] = {lhs_id, rhs_id, node, 0u, 0u, 0u, 0u};
    const int rc[7] = {0, 0, 0, 1, 1, 1, 1};
    logup_constraints<7>(ca, a, rm, rv, ri, rc, false, s, t, E);
  } else if (KIND == 14) {
    // RangeCheckLookup: multiplicity column + preprocessed LUT column, relation (-multiplicity, [lut])
    const uint32_t rm[1] = {LMN_COL(0)}, rv[1] = {a.pre[t]}, ri[1] = {0u};
    const int rc[1] = {1};
    logup_constraints<1>(ca, a, rm, rv, ri, rc, true, s, t, E);
  } else if (KIND == 4) {
    // SinLookup / Exp2Lookup / Log2Lookup (lookups/sin/component.rs:40-59): multiplicity column + the two
    // preprocessed LUT columns, relation (-multiplicity, [lut_0, lut_1])
    const uint32_t rm[1] = {LMN_COL(0)}, rv[1] = {a.pre[t]}, ri[1] = {a.pre2[t]};
    const int rc[1] = {2};
    logup_constraints<1>(ca, a, rm, rv, ri, rc, true, s, t, E);
  } else if (KIND == 3) {
    // Sin / Exp2 / Log2 (12 cols; sin/component.rs:50-122): the function value is enforced by the LUT
    // relation (lookup_mult, [input, out]) only
    const uint32_t node = LMN_COL(0), in_id = LMN_COL(1), idx = LMN_COL(2), is_last = LMN_COL(3);
    const uint32_t n_node = LMN_COL(4), n_in = LMN_COL(5), n_idx = LMN_COL(6);
    const uint32_t inp = LMN_COL(7), out = LMN_COL(8);
    const uint32_t m0 = LMN_COL(9), m1 = LMN_COL(10), m2 = LMN_COL(11);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(n_in, in_id)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[3] = {m0, m1, m2}, rv[3] = {inp, out, inp}, ri[3] = {in_id, node, out};
    const int rc[3] = {0, 0, 2};
    logup_constraints<3>(ca, a, rm, rv, ri, rc, false, s, t, E);
  } else if (KIND == 5 || KIND == 6 || KIND == 16) {
    // SumReduce (14 cols) / MaxReduce (15) / Contiguous (11): shared id/idx prefix, 2 relations
    const uint32_t node = LMN_COL(0), in_id = LMN_COL(1), idx = LMN_COL(2), is_last = LMN_COL(3);
    const uint32_t n_node = LMN_COL(4), n_in = LMN_COL(5), n_idx = LMN_COL(6);
    const uint32_t inp = LMN_COL(7), out = LMN_COL(8);
    constexpr int mo = KIND == 5 ? 12 : (KIND == 6 ? 13 : 9);
    const uint32_t m0 = LMN_COL(mo), m1 = LMN_COL(mo + 1);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    if (KIND == 5) {
      const uint32_t acc = LMN_COL(9), next_acc = LMN_COL(10), ils = LMN_COL(11);
      ca.add_m(m_mul(ils, m_sub(ils, 1u)));
      ca.add_m(m_sub(next_acc, m_add(acc, inp)));
      ca.add_m(m_mul(m_sub(out, next_acc), ils));
    } else if (KIND == 6) {
      const uint32_t mx = LMN_COL(9), next_mx = LMN_COL(10), ils = LMN_COL(11), im = LMN_COL(12);
      ca.add_m(m_mul(ils, m_sub(ils, 1u)));
      ca.add_m(m_mul(im, m_sub(im, 1u)));
      ca.add_m(m_mul(im, m_sub(next_mx, inp)));
      ca.add_m(m_mul(m_sub(1u, im), m_sub(next_mx, mx)));
      ca.add_m(m_mul(m_sub(out, next_mx), ils));
    }
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(n_in, in_id)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[2] = {m0, m1}, rv[2] = {inp, out}, ri[2] = {in_id, node};
    const int rc[2] = {0, 0};
    logup_constraints<2>(ca, a, rm, rv, ri, rc, false, s, t, E);
  } else {
    const uint32_t node = LMN_COL(0), idx = LMN_COL(1), is_last = LMN_COL(2), n_node = LMN_COL(3), n_idx = LMN_COL(4);
    const uint32_t val = LMN_COL(5), mult = LMN_COL(6);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[1] = {mult}, rv[1] = {val}, ri[1] = {node};
    const int rc[1] = {0};
    logup_constraints<1>(ca, a, rm, rv, ri, rc, false, s, t, E);
  }
#undef LMN_COL
  QM31 r = q_mul_m(qacc_reduce(ca.acc), a.zinv[(s >> a.log_size) & 1u]);
  uint32_t* o = a.out + s;
  if (a.accumulate) {
    r.a = m_add(r.a, o[0]);
    r.b = m_add(r.b, o[E]);
    r.c = m_add(r.c, o[2 * E]);
    r.d = m_add(r.d, o[3 * E]);
  }
  o[0] = r.a;
  o[E] = r.b;
  o[2 * E] = r.c;
  o[3 * E] = r.d;
}

void launch_composition(const CompositionArgs& a, lmn_stream_t s) {
  if (LMN_ABLATED(8u)) return;
  if (a.eval_log != a.log_size + 1) throw LmnError(-100, "composition: eval domain must be log_size+1");
  if (a.n_rows == 0 || (uint64_t)a.row0 + a.n_rows > (1ull << a.eval_log) || a.stride < a.n_rows || !a.prev_last)
    throw LmnError(-100, "composition: bad row block");
  dim3 g(cdiv(a.n_rows, TPB)), b(TPB);
  switch (a.kind) {
    case 0: LMN_LAUNCH(k_composition<0>, g, b, 0, s, a); break;
    case 1: LMN_LAUNCH(k_composition<1>, g, b, 0, s, a); break;
    case 2: LMN_LAUNCH(k_composition<2>, g, b, 0, s, a); break;
    case 3:
    case 9:
    case 11: LMN_LAUNCH(k_composition<3>, g, b, 0, s, a); break;
    case 4:
    case 10:
    case 12: LMN_LAUNCH(k_composition<4>, g, b, 0, s, a); break;
    case 5: LMN_LAUNCH(k_composition<5>, g, b, 0, s, a); break;
    case 6: LMN_LAUNCH(k_composition<6>, g, b, 0, s, a); break;
    case 7: LMN_LAUNCH(k_composition<7>, g, b, 0, s, a); break;
    case 8: LMN_LAUNCH(k_composition<8>, g, b, 0, s, a); break;
    case 13: LMN_LAUNCH(k_composition<13>, g, b, 0, s, a); break;
    case 14: LMN_LAUNCH(k_composition<14>, g, b, 0, s, a); break;
    case 15: LMN_LAUNCH(k_composition<15>, g, b, 0, s, a); break;
    case 16: LMN_LAUNCH(k_composition<16>, g, b, 0, s, a); break;
    default: throw LmnError(-100, "composition: unsupported component kind");
  }
}

LMN_KERNEL k_secure_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = m_add(out[i], in[i]);
}
void launch_secure_add(uint32_t* out, const uint32_t* in, uint64_t n_words, lmn_stream_t s) {
  LMN_LAUNCH(k_secure_add, dim3(cdiv(n_words, TPB)), dim3(TPB), 0, s, out, in, n_words);
}

// =============================================================================================
// a9  eval_at_point: sum_j coeff_j * basis_j(point), basis factored as lo-table x hi-table
// =============================================================================================
constexpr int EVAL_HI_PER_CHUNK = 8;
LMN_HD int eval_num_chunks_hd(int log_n) {
  if (log_n <= EVAL_LB) return 1;
  int total_hi = 1 << (log_n - EVAL_LB);
  int hpc = total_hi < EVAL_HI_PER_CHUNK ? total_hi : EVAL_HI_PER_CHUNK;
  return total_hi / hpc;
}
int eval_num_chunks(int log_n) { return eval_num_chunks_hd(log_n); }

// shard_world > 1 (single-proof sharding): the chunks of every job are dealt to the ranks in contiguous runs (a job
// with fewer chunks than ranks gives one chunk to each of the first ranks); a rank writes zero for chunks it does
// not own, so that the per-job reduction is this rank's PARTIAL sum - the ranks' partials are all-gathered (16 B per
// job and rank) and added on the host.
LMN_KERNEL k_eval_at_point(const EvalJob* __restrict__ jobs, const QM31* __restrict__ lo_tab,
                           const QM31* __restrict__ hi_tab, uint32_t hi_stride, QM31* __restrict__ partial_out,
                           int max_chunks, uint32_t shard_rank, uint32_t shard_world) {
  LMN_SHARED QM31 red[TPB];
  const EvalJob job = jobs[blockIdx.y];
  const int chunk = blockIdx.x;
  const int nchunks = eval_num_chunks_hd(job.log_n);
  if (chunk >= nchunks) return;
  if (shard_world > 1) {
    // job.owner >= 0: the coefficients exist on that rank only, which evaluates every chunk
    const uint32_t owner = job.owner >= 0 ? (uint32_t)job.owner
                           : (uint32_t)nchunks >= shard_world ? (uint32_t)chunk / ((uint32_t)nchunks / shard_world) : (uint32_t)chunk;
    if (owner != shard_rank) {
      if (threadIdx.x == 0) partial_out[(uint64_t)blockIdx.y * max_chunks + chunk] = q_zero();
      return;
    }
  }
  const int lb = job.log_n < EVAL_LB ? job.log_n : EVAL_LB;
  const uint32_t lo_n = 1u << lb;
  const uint32_t total_hi = 1u << (job.log_n - lb);
  const uint32_t hpc = total_hi / (uint32_t)nchunks;
  const QM31* L = lo_tab + ((uint64_t)job.point << EVAL_LB);
  const QM31* Hh = hi_tab + (uint64_t)job.point * hi_stride;
  // each lane owns at most 4 lo positions (tid + 256k): keep their basis values in registers
  QM31 Lr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t lo = threadIdx.x + (uint32_t)k * TPB;
    Lr[k] = lo < lo_n ? L[lo] : q_zero();
  }
  // sum_hi H[hi] * (sum_lo L[lo] * c[hi, lo]) regrouped as sum_lo L[lo] * (sum_hi H[hi] * c[hi, lo]): the
  // inner sums are QM31 (wave-uniform H) x M31 products, accumulated lazily in 64-bit lanes; one full
  // QM31 product per owned lo position closes the chunk.
  QAcc in0 = qacc_zero(), in1 = qacc_zero(), in2 = qacc_zero(), in3 = qacc_zero();
  uint32_t pending = 0;
  for (uint32_t hh = 0; hh < hpc; ++hh) {
    const uint32_t hi = chunk * hpc + hh;
    const uint32_t* __restrict__ cp = job.coeffs + ((uint64_t)hi << lb);
    const QM31 Hv = Hh[hi];
    uint32_t cv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t lo = threadIdx.x + (uint32_t)k * TPB;
      cv[k] = lo < lo_n ? cp[lo] : 0u;
    }
    LMN_QPHASE_PORT0();
    qacc_mad(in0, Hv, cv[0]);
    qacc_mad(in1, Hv, cv[1]);
    qacc_mad(in2, Hv, cv[2]);
    qacc_mad(in3, Hv, cv[3]);
    LMN_QPHASE_ANY();
    if (++pending == 3) {
      pending = 0;
      qacc_fold(in0);
      qacc_fold(in1);
      qacc_fold(in2);
      qacc_fold(in3);
    }
  }
  QM31 acc = q_mul(Lr[0], qacc_reduce(in0));
  acc = q_add(acc, q_mul(Lr[1], qacc_reduce(in1)));
  acc = q_add(acc, q_mul(Lr[2], qacc_reduce(in2)));
  acc = q_add(acc, q_mul(Lr[3], qacc_reduce(in3)));
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = q_add(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial_out[(uint64_t)blockIdx.y * max_chunks + chunk] = red[0];
}

void launch_eval_at_point(const EvalJob* jobs, int njobs, const QM31* lo_tab, const QM31* hi_tab, uint32_t hi_stride,
                          int max_log, QM31* partial_out, int max_chunks, lmn_stream_t s, uint32_t shard_rank,
                          uint32_t shard_world) {
  (void)max_log;
  if (LMN_ABLATED(16u)) return;
  LMN_LAUNCH(k_eval_at_point, dim3(max_chunks, njobs), dim3(TPB), 0, s, jobs, lo_tab, hi_tab, hi_stride, partial_out,
             max_chunks, shard_rank, shard_world);
}

// basis tables on the device: lo[p][j] = prod_{k<EVAL_LB} maps[p][k]^(bit k of j); hi[p][j] likewise
// with maps[p][EVAL_LB + k].  maps = [y, x, pi(x), pi^2(x), ...] per sample point.
LMN_KERNEL k_eval_tables(const QM31* __restrict__ maps, int maps_stride, QM31* __restrict__ lo_tab,
                         QM31* __restrict__ hi_tab, uint32_t hi_n, int hi_bits) {
  LMN_SERIAL_KERNEL();
  const int p = blockIdx.y;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const QM31* mp = maps + (uint64_t)p * maps_stride;
  const uint32_t lo_n = 1u << EVAL_LB;
  if (j < lo_n) {
    QM31 acc = q_one();
    for (int k = 0; k < EVAL_LB; ++k)
      if ((j >> k) & 1u) acc = q_mul(acc, mp[k]);
    lo_tab[(uint64_t)p * lo_n + j] = acc;
  }
  if (j < hi_n) {
    QM31 acc = q_one();
    for (int k = 0; k < hi_bits; ++k)
      if ((j >> k) & 1u) acc = q_mul(acc, mp[EVAL_LB + k]);
    hi_tab[(uint64_t)p * hi_n + j] = acc;
  }
}

void launch_eval_tables(const QM31* maps, int maps_stride, int npoints, QM31* lo_tab, QM31* hi_tab, uint32_t hi_n,
                        int hi_bits, lmn_stream_t s) {
  uint32_t n = hi_n > (1u << EVAL_LB) ? hi_n : (1u << EVAL_LB);
  LMN_LAUNCH(k_eval_tables, dim3(cdiv(n, TPB), npoints), dim3(TPB), 0, s, maps, maps_stride, lo_tab, hi_tab, hi_n,
             hi_bits);
}

// out[job] = sum over that job's chunks of partial[job][chunk]
LMN_KERNEL k_eval_reduce(const EvalJob* __restrict__ jobs, const QM31* __restrict__ partial, int max_chunks,
                         QM31* __restrict__ out) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED QM31 red[TPB];
  const int job = blockIdx.x;
  const int nc = eval_num_chunks_hd(jobs[job].log_n);
  QM31 acc = q_zero();
  for (int c = threadIdx.x; c < nc; c += blockDim.x) acc = q_add(acc, partial[(uint64_t)job * max_chunks + c]);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = q_add(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[job] = red[0];
}

void launch_eval_reduce(const EvalJob* jobs, int njobs, const QM31* partial, int max_chunks, QM31* out,
                        lmn_stream_t s) {
  LMN_LAUNCH(k_eval_reduce, dim3(njobs), dim3(TPB), 0, s, jobs, partial, max_chunks, out);
}

// =============================================================================================
// a9  FRI quotients: row = sum_batches [ row*alpha^|batch| + (sum_cols c*f(q) - (A*q.y + B)) / den ]
// =============================================================================================
LMN_HD uint32_t domain_x(const uint32_t* tw_x, uint32_t s) {
  uint32_t x = tw_x[s >> 2];
  return (s & 2u) ? m_neg(x) : x;
}
LMN_HD uint32_t domain_y(const uint32_t* tw_y, uint32_t s) {
  uint32_t y = tw_y[s >> 1];
  return (s & 1u) ? m_neg(y) : y;
}

// NB = number of sample-point batches (compile-time: exact loops, no dummy products); every lane owns
// QUOT_ROWS rows a quarter of the domain apart and inverts all their denominator norms with ONE field
// inversion (Montgomery batching: 3 products per element instead of a 37-product exponentiation per row).
// Rows per lane: 4 for one batch, 2 for two batches (the headline shape) - with the register budget of 8 waves per SIMD
// (64 VGPRs, no spills) those two launches fill the chip in whole rounds (16 waves per SIMD in two rounds of 8; at 87
// VGPRs the 8 waves per SIMD of the two-batch launch ran as 5 + 3): 0.167 -> 0.150 ms per proof, + 2 % proofs/s.
// Three and four batches keep 4 rows at the compiler's own budget (they would spill at 64).
#ifndef LMN_QUOT_ROWS1
#define LMN_QUOT_ROWS1 4
#endif
#ifndef LMN_QUOT_ROWS2
#define LMN_QUOT_ROWS2 2
#endif
template <int NB>
constexpr int quot_rows() { return NB == 1 ? LMN_QUOT_ROWS1 : (NB == 2 ? LMN_QUOT_ROWS2 : 4); }   // rows per lane
template <int NB>
LMN_D void quotients_body(const QuotientArgs& a);
template <int NB>
LMN_KERNEL k_quotients(QuotientArgs a) { quotients_body<NB>(a); }
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
template <int NB>
__attribute__((amdgpu_waves_per_eu(8, 8))) LMN_KERNEL k_quotients_occ(QuotientArgs a) { quotients_body<NB>(a); }
#else
template <int NB>
LMN_KERNEL k_quotients_occ(QuotientArgs a) { quotients_body<NB>(a); }
#endif
template <int NB>
LMN_D void quotients_body(const QuotientArgs& a) {
  constexpr int QUOT_ROWS = quot_rows<NB>();
  // (column pointer, alpha^k * c) table staged once per block in LDS: the per-column loop then
  // reads wave-uniform LDS words instead of chasing pointers through global memory
  LMN_SHARED QuotEntry tab[QUOT_MAX_ENTRIES];
  const int nent = a.batch_start[NB];
  for (int e = threadIdx.x; e < nent; e += blockDim.x) tab[e] = a.entries[e];
  __syncthreads();
  const uint32_t Q = (1u << a.log_rows) / QUOT_ROWS;
  const uint32_t s0 = blockIdx.x * blockDim.x + threadIdx.x;  // row inside the block handled by this launch
  if (s0 >= Q) return;
  constexpr int NE = QUOT_ROWS * NB;
  CM31 den[NE];
  uint32_t nrm[NE], pre[NE], ys[QUOT_ROWS];
#pragma unroll
  for (int k = 0; k < QUOT_ROWS; ++k) {
    const uint32_t s = a.row0 + s0 + (uint32_t)k * Q;  // storage index on the whole domain
    const uint32_t x = a.log_size >= 2 ? domain_x(a.tw_x, s) : 0u;
    const uint32_t y = domain_y(a.tw_y, s);
    ys[k] = y;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int e = k * NB + b;
      CM31 dx{m_sub(a.prx[b].a, x), a.prx[b].b};
      CM31 dy{m_sub(a.pry[b].a, y), a.pry[b].b};
      den[e] = c_sub(c_mul(dx, a.piy[b]), c_mul(dy, a.pix[b]));
      nrm[e] = c_norm(den[e]);
      pre[e] = e == 0 ? nrm[e] : m_mul(pre[e - 1], nrm[e]);
    }
  }
  uint32_t inv = m_inv(pre[NE - 1]);
  CM31 dinv[NE];
#pragma unroll
  for (int e = NE - 1; e >= 0; --e) {
    const uint32_t ni = e == 0 ? inv : m_mul(inv, pre[e - 1]);
    inv = m_mul(inv, nrm[e]);
    dinv[e] = CM31{m_mul(den[e].a, ni), m_mul(m_neg(den[e].b), ni)};
  }
#pragma unroll
  for (int k = 0; k < QUOT_ROWS; ++k) {
    const uint32_t s = s0 + (uint32_t)k * Q;
    QM31 row = q_zero();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      // sum_k c_k * f_k(s) accumulated lazily in 64-bit lanes (three products per fold)
      QAcc acc = qacc_zero();
      const int k1 = a.batch_start[b + 1];
      int kk = a.batch_start[b];
      for (; kk + 6 <= k1; kk += 6) {
        uint32_t f0 = tab[kk].col[s], f1 = tab[kk + 1].col[s], f2 = tab[kk + 2].col[s];
        uint32_t f3 = tab[kk + 3].col[s], f4 = tab[kk + 4].col[s], f5 = tab[kk + 5].col[s];
        LMN_QPHASE_PORT0();
        qacc_mad(acc, tab[kk].c, f0);
        qacc_mad(acc, tab[kk + 1].c, f1);
        qacc_mad(acc, tab[kk + 2].c, f2);
        LMN_QPHASE_ANY();
        qacc_fold(acc);
        LMN_QPHASE_PORT0();
        qacc_mad(acc, tab[kk + 3].c, f3);
        qacc_mad(acc, tab[kk + 4].c, f4);
        qacc_mad(acc, tab[kk + 5].c, f5);
        LMN_QPHASE_ANY();
        qacc_fold(acc);
      }
      for (; kk < k1; ++kk) {
        qacc_mad(acc, tab[kk].c, tab[kk].col[s]);
        qacc_fold(acc);
      }
      QM31 num = qacc_reduce(acc);
      num = q_sub(num, q_add(q_mul_m(a.A[b], ys[k]), a.B[b]));
      const QM31 term = q_mul_c(num, dinv[k * NB + b]);
      row = b == 0 ? term : q_add(q_mul(row, a.batch_coeff[b]), term);  // no 0 * coeff product for the first batch
    }
    uint32_t* o = a.out + s;
    o[0] = row.a;
    o[a.out_stride] = row.b;
    o[2ull * a.out_stride] = row.c;
    o[3ull * a.out_stride] = row.d;
  }
}

void launch_quotients(const QuotientArgs& a, lmn_stream_t s) {
  if (LMN_ABLATED(4u)) return;
  if (a.nbatch < 1 || a.nbatch > QUOT_MAX_BATCH) throw LmnError(-100, "quotients: bad batch count");
  if (a.batch_start[a.nbatch] > QUOT_MAX_ENTRIES) throw LmnError(-100, "quotients: too many column samples");
  if (a.log_size < 2 || a.log_rows < 2 || a.log_rows > a.log_size || (a.row0 & ((1u << a.log_rows) - 1u)) ||
      a.out_stride < (1ull << a.log_rows))
    throw LmnError(-100, "quotients: bad row block");
  if (a.nbatch == 1 && (1u << a.log_rows) < (unsigned)quot_rows<1>()) throw LmnError(-100, "quotients: row block too small");
  dim3 g(cdiv((1ull << a.log_rows) / 4, TPB)), g1(cdiv((1ull << a.log_rows) / quot_rows<1>(), TPB)),
      g2(cdiv((1ull << a.log_rows) / quot_rows<2>(), TPB)), b(TPB);
  switch (a.nbatch) {
    case 1: LMN_LAUNCH(k_quotients_occ<1>, g1, b, 0, s, a); break;
    case 2: LMN_LAUNCH(k_quotients_occ<2>, g2, b, 0, s, a); break;
    case 3: LMN_LAUNCH(k_quotients<3>, g, b, 0, s, a); break;
    default: LMN_LAUNCH(k_quotients<4>, g, b, 0, s, a); break;
  }
}

// =============================================================================================
// a9  FRI folds
// =============================================================================================
LMN_KERNEL k_fold(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t src_len,
                  const uint32_t* __restrict__ itw, const QM31* __restrict__ alpha_ptr, int accumulate,
                  uint64_t dst_stride) {
  LMN_SERIAL_KERNEL();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (src_len >> 1)) return;
  const uint64_t n = dst_stride;
  const QM31 alpha = *alpha_ptr;
  const uint64_t L = src_len;
  QM31 a{src[2 * i], src[L + 2 * i], src[2 * L + 2 * i], src[3 * L + 2 * i]};
  QM31 b{src[2 * i + 1], src[L + 2 * i + 1], src[2 * L + 2 * i + 1], src[3 * L + 2 * i + 1]};
  QM31 f0 = q_add(a, b);
  QM31 f1 = q_mul_m(q_sub(a, b), itw[i]);
  QM31 r = q_add(f0, q_mul(alpha, f1));
  if (accumulate) {
    QM31 d{dst[i], dst[n + i], dst[2ull * n + i], dst[3ull * n + i]};
    r = q_add(q_mul(d, q_mul(alpha, alpha)), r);
  }
  dst[i] = r.a;
  dst[n + i] = r.b;
  dst[2ull * n + i] = r.c;
  dst[3ull * n + i] = r.d;
}

void launch_fold_circle_into_line(uint32_t* dst, const uint32_t* src, uint32_t src_len, const uint32_t* itw_y,
                                  const QM31* alpha, int accumulate, lmn_stream_t s, uint64_t dst_stride) {
  LMN_LAUNCH(k_fold, dim3(cdiv(src_len / 2, TPB)), dim3(TPB), 0, s, dst, src, src_len, itw_y, alpha, accumulate,
             dst_stride ? dst_stride : (uint64_t)(src_len / 2));
}
void launch_fold_line(uint32_t* dst, const uint32_t* src, uint32_t src_len, const uint32_t* itw_x, const QM31* alpha,
                      lmn_stream_t s, uint64_t dst_stride) {
  LMN_LAUNCH(k_fold, dim3(cdiv(src_len / 2, TPB)), dim3(TPB), 0, s, dst, src, src_len, itw_x, alpha, 0,
             dst_stride ? dst_stride : (uint64_t)(src_len / 2));
}

}  // namespace lmn
