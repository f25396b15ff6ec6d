// gfx950 kernels, part 5: constraint quotients of all 17 components, point evaluation, FRI quotients and folds
// (SURVEY.md section 8a rows a7, a9).
#include "kernels_common.h"

namespace lmn {

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
struct CompElems {
  QM31 z, alpha, z2, alpha2;
};
template <int NREL>
LMN_D void logup_constraints(ConsAcc& ca, const CompositionArgs& a, const CompElems& ce, const uint32_t (&mult)[NREL],
                             const uint32_t (&val)[NREL], const uint32_t (&id)[NREL], const int (&rc)[NREL], bool neg,
                             uint32_t s, uint32_t t, uint64_t E) {
  QM31 prev = q_zero();
#pragma unroll
  for (int j = 0; j < NREL; ++j) {
    QM31 cur = load_secure_ub(a.inter + (uint64_t)(4 * j) * a.stride, a.stride, t);
    QM31 den = rc[j] == 1   ? q_sub(q_from_m(val[j]), ce.z2)
               : rc[j] == 2 ? q_sub(q_add_m(q_mul_m(ce.alpha2, id[j]), val[j]), ce.z2)
                            : q_sub(q_add_m(q_mul_m(ce.alpha, id[j]), val[j]), ce.z);
    QM31 diff;
    if (j < NREL - 1) {
      diff = q_sub(cur, prev);
    } else {
      uint32_t ps = prev_row_storage(s, a.eval_log, a.log_size);
      QM31 pr = load_secure_ub(a.prev_last, E, ps);
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
  // coefficients and relation elements: kernel arguments, or what the device-resident channel left in memory
  ConsAcc ca{qacc_zero(), a.d_coeff ? a.d_coeff : a.coeff, 0};
  const CompElems ce = a.d_elems ? CompElems{a.d_elems->z[0], a.d_elems->alpha[0], a.d_elems->z[a.es2], a.d_elems->alpha[a.es2]}
                                 : CompElems{a.z, a.alpha, a.z2, a.alpha2};
  const uint32_t* __restrict__ mn = a.main;   // wave-uniform column bases, the row as a 32-bit lane offset (ld_ub)
  const uint64_t cstride = a.stride;
#define LMN_COL(k) ld_col(mn, (k), cstride, t)
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
    logup_constraints<3>(ca, a, ce, rm, rv, ri, rc, false, s, t, E);
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
    logup_constraints<2>(ca, a, ce, rm, rv, ri, rc, false, s, t, E);
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
    logup_constraints<3>(ca, a, ce, rm, rv, ri, rc, false, s, t, E);
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
    logup_constraints<7>(ca, a, ce, rm, rv, ri, rc, false, s, t, E);
  } else if (KIND == 14) {
    // RangeCheckLookup: multiplicity column + preprocessed LUT column, relation (-multiplicity, [lut])
    const uint32_t rm[1] = {LMN_COL(0)}, rv[1] = {ld_ub(a.pre, t)}, ri[1] = {0u};
    const int rc[1] = {1};
    logup_constraints<1>(ca, a, ce, rm, rv, ri, rc, true, s, t, E);
  } else if (KIND == 4) {
    // SinLookup / Exp2Lookup / Log2Lookup (lookups/sin/component.rs:40-59): multiplicity column + the two
    // preprocessed LUT columns, relation (-multiplicity, [lut_0, lut_1])
    const uint32_t rm[1] = {LMN_COL(0)}, rv[1] = {ld_ub(a.pre, t)}, ri[1] = {ld_ub(a.pre2, t)};
    const int rc[1] = {2};
    logup_constraints<1>(ca, a, ce, rm, rv, ri, rc, true, s, t, E);
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
    logup_constraints<3>(ca, a, ce, rm, rv, ri, rc, false, s, t, E);
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
    logup_constraints<2>(ca, a, ce, rm, rv, ri, rc, false, s, t, E);
  } else {
    const uint32_t node = LMN_COL(0), idx = LMN_COL(1), is_last = LMN_COL(2), n_node = LMN_COL(3), n_idx = LMN_COL(4);
    const uint32_t val = LMN_COL(5), mult = LMN_COL(6);
    ca.add_m(m_mul(is_last, m_sub(is_last, 1u)));
    const uint32_t not_last = m_sub(1u, is_last);
    ca.add_m(m_mul(not_last, m_sub(n_node, node)));
    ca.add_m(m_mul(not_last, m_sub(m_sub(n_idx, idx), 1u)));
    const uint32_t rm[1] = {mult}, rv[1] = {val}, ri[1] = {node};
    const int rc[1] = {0};
    logup_constraints<1>(ca, a, ce, rm, rv, ri, rc, false, s, t, E);
  }
#undef LMN_COL
  QM31 r = q_mul_m(qacc_reduce(ca.acc), a.zinv[(s >> a.log_size) & 1u]);
  if (a.accumulate) {
    r.a = m_add(r.a, ld_ub(a.out, s));
    r.b = m_add(r.b, ld_ub(a.out + E, s));
    r.c = m_add(r.c, ld_ub(a.out + 2 * E, s));
    r.d = m_add(r.d, ld_ub(a.out + 3 * E, s));
  }
  st_ub(a.out, s, r.a);
  st_ub(a.out + E, s, r.b);
  st_ub(a.out + 2 * E, s, r.c);
  st_ub(a.out + 3 * E, s, r.d);
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
  // a chunk is at most EVAL_HI_PER_CHUNK runs of coefficients: all their loads are issued before the first product
  uint32_t cv[EVAL_HI_PER_CHUNK][4];
  QM31 Hv[EVAL_HI_PER_CHUNK];
#pragma unroll
  for (uint32_t hh = 0; hh < (uint32_t)EVAL_HI_PER_CHUNK; ++hh) {
    const uint32_t hi = chunk * hpc + hh;
    const uint32_t* __restrict__ cp = job.coeffs + ((uint64_t)hi << lb);
    Hv[hh] = hh < hpc ? Hh[hi] : q_zero();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t lo = threadIdx.x + (uint32_t)k * TPB;
      cv[hh][k] = (hh < hpc && lo < lo_n) ? ld_ub(cp, lo) : 0u;   // (block-uniform run of coefficients: kernels_common.h)
    }
  }
#pragma unroll
  for (uint32_t hh = 0; hh < (uint32_t)EVAL_HI_PER_CHUNK; ++hh) {
    if (hh >= hpc) break;
    LMN_QPHASE_PORT0();
    qacc_mad(in0, Hv[hh], cv[hh][0]);
    qacc_mad(in1, Hv[hh], cv[hh][1]);
    qacc_mad(in2, Hv[hh], cv[hh][2]);
    qacc_mad(in3, Hv[hh], cv[hh][3]);
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
#if defined(LMN_QUOT_TAB_LDS) || defined(LMN_EMU) || !defined(__HIP_DEVICE_COMPILE__)
  // (column pointer, alpha^k * c) table staged once per block in LDS: the per-column loop then
  // reads wave-uniform LDS words instead of chasing pointers through global memory
  LMN_SHARED QuotEntry tab[QUOT_MAX_ENTRIES];
  const int nent = a.batch_start[NB];
  for (int e = threadIdx.x; e < nent; e += blockDim.x) tab[e] = a.entries[e];
  __syncthreads();
#define LMN_QCOL(e, row) tab[e].col[row]
#else
#define LMN_QCOL(e, row) ld_ub(tab[e].col, row)
  // The (column pointer, alpha^k * c) table is read through the SCALAR cache (constant address space: nothing writes it
  // during the launch): a column's base address and its coefficient arrive in SGPRs, so the column load takes an SGPR
  // base + the row as a 32-bit lane offset (no 64-bit vector address per load, one v_lshl_add_u64 of ~10 instructions per
  // column and row before) and the multiply-accumulates take the coefficient as their scalar operand; the LDS copy of
  // the table (a ds_read_b64 + ds_read_b128 per column and row group) is gone.
  typedef const QuotEntry __attribute__((address_space(4))) * ConstTab;
  const ConstTab tab = (ConstTab)(uintptr_t)a.entries;
#endif
#if defined(LMN_EMU) || !defined(__HIP_DEVICE_COMPILE__)
  const QuotDev& qd = *a.dev;
#else
  // (the same for the batches' sums and points: written before this launch, never during it)
  const QuotDev __attribute__((address_space(4)))& qd = *(const QuotDev __attribute__((address_space(4)))*)(uintptr_t)a.dev;
#endif
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
      CM31 dx{m_sub(qd.prx[b].a, x), qd.prx[b].b};
      CM31 dy{m_sub(qd.pry[b].a, y), qd.pry[b].b};
      den[e] = c_sub(c_mul(dx, CM31{qd.piy[b].a, qd.piy[b].b}), c_mul(dy, CM31{qd.pix[b].a, qd.pix[b].b}));
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
        uint32_t f0 = LMN_QCOL(kk, s), f1 = LMN_QCOL(kk + 1, s), f2 = LMN_QCOL(kk + 2, s);
        uint32_t f3 = LMN_QCOL(kk + 3, s), f4 = LMN_QCOL(kk + 4, s), f5 = LMN_QCOL(kk + 5, s);
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
        qacc_mad(acc, tab[kk].c, LMN_QCOL(kk, s));
        qacc_fold(acc);
      }
      QM31 num = qacc_reduce(acc);
      num = q_sub(num, q_add(q_mul_m(QM31{qd.A[b].a, qd.A[b].b, qd.A[b].c, qd.A[b].d}, ys[k]),
                             QM31{qd.B[b].a, qd.B[b].b, qd.B[b].c, qd.B[b].d}));
      const QM31 term = q_mul_c(num, dinv[k * NB + b]);
      row = b == 0 ? term   // no 0 * coeff product for the first batch
                   : q_add(q_mul(row, QM31{qd.batch_coeff[b].a, qd.batch_coeff[b].b, qd.batch_coeff[b].c, qd.batch_coeff[b].d}), term);
    }
    uint32_t* o = a.out + s;
    o[0] = row.a;
    o[a.out_stride] = row.b;
    o[2ull * a.out_stride] = row.c;
    o[3ull * a.out_stride] = row.d;
  }
#undef LMN_QCOL
}

void launch_quotients(const QuotientArgs& a, lmn_stream_t s) {
  if (LMN_ABLATED(4u)) return;
  if (a.nbatch < 1 || a.nbatch > QUOT_MAX_BATCH) throw LmnError(-100, "quotients: bad batch count");
  if (a.batch_start[a.nbatch] > QUOT_MAX_ENTRIES) throw LmnError(-100, "quotients: too many column samples");
  if (!a.dev || !a.entries) throw LmnError(-100, "quotients: no batch table");
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
