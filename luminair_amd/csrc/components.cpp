// The 17 AIR components as data (column layouts, relation wiring, padding rows: crates/air/src/components/**/table.rs,
// component.rs), the constraint slots under the protocol's constraint-form bits, the relation-element draws
// (components/mod.rs:227-235, lookups/mod.rs:44-51) and the host-side evaluation of the composition polynomial at the OODS
// point from sampled mask values (shared by the prover's self-check and the verifier).
#include "prover_internal.h"

namespace lmn {

// ------------------------------------------------------------------------------------ components
// Column layouts / relation wiring: crates/air/src/components/{add,mul,recip,inputs}/{table,component}.rs
static const ComponentSpec kSpecs[] = {
    // kind, n_cols, is_last, n_rel, rel_mult, rel_val, rel_id, n_local, rel_elems, rel_neg, rel_pre, n_pre, pre_id, n_pad, pad_col, pad_val
    {LMN_KIND_ADD, 15, 4, 3, {12, 13, 14}, {9, 10, 11}, {1, 2, 0}, 6, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_MUL, 16, 4, 3, {13, 14, 15}, {9, 10, 11}, {1, 2, 0}, 7, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_RECIP, 13, 3, 2, {11, 12}, {7, 8}, {1, 0}, 5, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_INPUTS, 7, 2, 1, {6}, {5}, {0}, 3, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    // constraint forms fully visible in the reference (no numerair helper):
    {LMN_KIND_SUM_REDUCE, 14, 3, 2, {12, 13}, {7, 8}, {1, 0}, 7, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},   // sum_reduce/component.rs:36-110
    {LMN_KIND_MAX_REDUCE, 15, 3, 2, {13, 14}, {7, 8}, {1, 0}, 9, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},   // max_reduce/component.rs
    {LMN_KIND_CONTIGUOUS, 11, 3, 2, {9, 10}, {7, 8}, {1, 0}, 4, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},    // contiguous/component.rs
    // numerair's eval_fixed_sqrt / eval_fixed_rem are un-vendored: natural fixed-point identities (unpinned)
    {LMN_KIND_SQRT, 13, 3, 2, {11, 12}, {7, 8}, {1, 0}, 5, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_REM, 16, 4, 3, {13, 14, 15}, {9, 10, 11}, {1, 2, 0}, 6, {0}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    // less_than/component.rs:48-185; padding row less_than/table.rs:47-72 (rhs=1, out=4096, diff=1, limb0=1)
    {LMN_KIND_LESS_THAN, 22, 4, 7, {18, 19, 20, 21, 21, 21, 21}, {9, 10, 11, 14, 15, 16, 17}, {1, 2, 0, -1, -1, -1, -1}, 9,
     {0, 0, 0, 1, 1, 1, 1}, {0}, {0}, 0, {0, 0}, 4, {10, 11, 12, 14}, {1u, 4096u, 1u, 1u}},
    // lookups/range_check/component.rs: (-multiplicity, [range_check_8_column_0])
    {LMN_KIND_RANGE_CHECK_LOOKUP, 1, -1, 1, {0}, {0}, {-1}, 0, {ELEMS_RANGE_CHECK}, {1}, {1}, 1, {PRE_RANGE_CHECK, 0}, 0, {0}, {0}},
    // sin/component.rs:50-122 (exp2, log2 alike): node relations on input/out + LUT relation (lookup_mult, [input, out])
    {LMN_KIND_SIN, 12, 3, 3, {9, 10, 11}, {7, 8, 7}, {1, 0, 8}, 4, {0, 0, ELEMS_SIN}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_EXP2, 12, 3, 3, {9, 10, 11}, {7, 8, 7}, {1, 0, 8}, 4, {0, 0, ELEMS_EXP2}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    {LMN_KIND_LOG2, 12, 3, 3, {9, 10, 11}, {7, 8, 7}, {1, 0, 8}, 4, {0, 0, ELEMS_LOG2}, {0}, {0}, 0, {0, 0}, 0, {0}, {0}},
    // lookups/sin/component.rs:40-59: (-multiplicity, [lut_0, lut_1]) over the two preprocessed columns
    {LMN_KIND_SIN_LOOKUP, 1, -1, 1, {0}, {0}, {1}, 0, {ELEMS_SIN}, {1}, {1}, 2, {PRE_SIN0, PRE_SIN0 + 1}, 0, {0}, {0}},
    {LMN_KIND_EXP2_LOOKUP, 1, -1, 1, {0}, {0}, {1}, 0, {ELEMS_EXP2}, {1}, {1}, 2, {PRE_EXP20, PRE_EXP20 + 1}, 0, {0}, {0}},
    {LMN_KIND_LOG2_LOOKUP, 1, -1, 1, {0}, {0}, {1}, 0, {ELEMS_LOG2}, {1}, {1}, 2, {PRE_LOG20, PRE_LOG20 + 1}, 0, {0}, {0}},
};
const ComponentSpec* component_spec(int kind) {
  for (auto& s : kSpecs)
    if (s.kind == kind) return &s;
  return nullptr;
}

ConstraintLayout constraint_layout(const ComponentSpec& sp, uint32_t flags) {
  ConstraintLayout L;
  L.n_kernel = sp.n_local + sp.n_rel;
  // kernel slot 1 is the eval_fixed_* constraint of Mul / Recip / Sqrt / Rem; Mul's kernel slot 2 is its zero slot
  bool drop_slot2 = false, extra_after1 = false, neg1 = false;
  switch (sp.kind) {
    case LMN_KIND_MUL: drop_slot2 = (flags & LMN_PV_MUL_ONE_SLOT) != 0; break;
    case LMN_KIND_RECIP: extra_after1 = (flags & LMN_PV_RECIP_TWO_SLOTS) != 0; neg1 = (flags & LMN_PV_RECIP_NEG) != 0; break;
    case LMN_KIND_SQRT: extra_after1 = (flags & LMN_PV_SQRT_TWO_SLOTS) != 0; neg1 = (flags & LMN_PV_SQRT_NEG) != 0; break;
    case LMN_KIND_REM: extra_after1 = (flags & LMN_PV_REM_TWO_SLOTS) != 0; neg1 = (flags & LMN_PV_REM_NEG) != 0; break;
    default: break;
  }
  int p = 0;
  for (int k = 0; k < L.n_kernel; ++k) {
    L.neg[k] = k == 1 && neg1;
    if (k == 2 && drop_slot2) {
      L.proto_index[k] = -1;
      continue;
    }
    L.proto_index[k] = p++;
    if (k == 1 && extra_after1) ++p;   // the helper's second (zero) slot
  }
  L.n_protocol = p;
  return L;
}

int relation_draw_sets(uint32_t protocol_flags, int sets_out[5]) {
  int n = 0;
  sets_out[n++] = ELEMS_NODE;
  sets_out[n++] = ELEMS_SIN;  // the KAT era drew a single LUT relation; HEAD: sin, exp2, log2, range_check
  if (protocol_flags & LMN_PV_LUT_DRAWS4) {
    sets_out[n++] = ELEMS_EXP2;
    sets_out[n++] = ELEMS_LOG2;
    sets_out[n++] = ELEMS_RANGE_CHECK;
  }
  return n;
}

RelElems draw_relation_elements(Channel& channel, uint32_t protocol_flags) {
  RelElems e;
  int sets[5];
  const int n = relation_draw_sets(protocol_flags, sets);
  for (int i = 0; i < n; ++i) {
    std::vector<QM31> d = channel.draw_felts(2);
    e.z[sets[i]] = d[0];
    e.alpha[sets[i]] = d[1];
    e.drawn[sets[i]] = true;
  }
  return e;
}

std::vector<int> assign_preprocessed(std::vector<Instance>& inst) {
  int log_of[N_PRE_IDS];
  for (int& l : log_of) l = -1;
  for (auto& ci : inst)
    for (int k = 0; k < ci.spec->n_pre; ++k) log_of[ci.spec->pre_id[k]] = ci.log_size;
  std::vector<int> order;
  for (int id = 0; id < N_PRE_IDS; ++id)
    if (log_of[id] >= 0) order.push_back(id);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return log_of[a] > log_of[b]; });
  int pos[N_PRE_IDS];
  std::vector<int> logs;
  for (size_t i = 0; i < order.size(); ++i) {
    pos[order[i]] = (int)i;
    logs.push_back(log_of[order[i]]);
  }
  for (auto& ci : inst)
    for (int k = 0; k < ci.spec->n_pre; ++k) ci.pre_idx[k] = pos[ci.spec->pre_id[k]];
  return logs;
}

// ------------------------------------------------------------------------------------ host-side AIR at a point
static QM31 qsub1(QM31 a) { return q_sub_m(a, 1u); }
static QM31 one_minus(QM31 a) { return q_sub(q_one(), a); }

// local constraints at a point, in `evaluate` order (crates/air/src/components/*/component.rs)
static std::vector<QM31> local_constraints(int kind, const std::vector<QM31>& c) {
  std::vector<QM31> out;
  if (kind == LMN_KIND_ADD || kind == LMN_KIND_MUL) {
    QM31 is_last = c[4], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    if (kind == LMN_KIND_ADD) {
      out.push_back(q_sub(c[11], q_add(c[9], c[10])));
    } else {
      out.push_back(q_sub(q_mul(c[9], c[10]), q_add(q_mul_m(c[11], 4096u), c[12])));
      out.push_back(q_zero());
    }
    out.push_back(q_mul(not_last, q_sub(c[5], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[6], c[1])));
    out.push_back(q_mul(not_last, q_sub(c[7], c[2])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[8], c[3]))));
  } else if (kind == LMN_KIND_RECIP) {
    QM31 is_last = c[3], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_sub(q_sqr(c[10]), q_add(q_mul(c[7], c[8]), c[9])));
    out.push_back(q_mul(not_last, q_sub(c[4], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[1])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[6], c[2]))));
  } else if (kind == LMN_KIND_SQRT) {
    QM31 is_last = c[3], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_sub(q_mul(c[7], c[10]), q_add(q_sqr(c[8]), c[9])));
    out.push_back(q_mul(not_last, q_sub(c[4], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[1])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[6], c[2]))));
  } else if (kind == LMN_KIND_REM) {
    QM31 is_last = c[4], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_sub(c[9], q_add(q_mul(c[10], c[12]), c[11])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[6], c[1])));
    out.push_back(q_mul(not_last, q_sub(c[7], c[2])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[8], c[3]))));
  } else if (kind == LMN_KIND_RANGE_CHECK_LOOKUP || kind == LMN_KIND_SIN_LOOKUP || kind == LMN_KIND_EXP2_LOOKUP ||
             kind == LMN_KIND_LOG2_LOOKUP) {
    // no local constraints
  } else if (kind == LMN_KIND_LESS_THAN) {
    QM31 is_last = c[4], not_last = one_minus(is_last), borrow = c[13];
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_mul(borrow, qsub1(borrow)));
    out.push_back(q_sub(c[11], q_mul_m(one_minus(borrow), 4096u)));
    out.push_back(q_sub(q_add(c[9], c[12]), c[10]));  // - borrow * (2^31 - 1) == 0 in M31
    out.push_back(q_sub(c[12], q_add(q_add(q_mul_m(c[17], 1u << 24), q_mul_m(c[16], 1u << 16)),
                                     q_add(q_mul_m(c[15], 1u << 8), c[14]))));
    out.push_back(q_mul(not_last, q_sub(c[5], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[6], c[1])));
    out.push_back(q_mul(not_last, q_sub(c[7], c[2])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[8], c[3]))));
  } else if (kind == LMN_KIND_INPUTS) {
    QM31 is_last = c[2], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    out.push_back(q_mul(not_last, q_sub(c[3], c[0])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[4], c[1]))));
  } else {  // SumReduce / MaxReduce / Contiguous / Sin / Exp2 / Log2 share the id/idx prefix (columns 0..6)
    QM31 is_last = c[3], not_last = one_minus(is_last);
    out.push_back(q_mul(is_last, qsub1(is_last)));
    if (kind == LMN_KIND_SUM_REDUCE) {
      QM31 ils = c[11];
      out.push_back(q_mul(ils, qsub1(ils)));
      out.push_back(q_sub(c[10], q_add(c[9], c[7])));
      out.push_back(q_mul(q_sub(c[8], c[10]), ils));
    } else if (kind == LMN_KIND_MAX_REDUCE) {
      QM31 ils = c[11], im = c[12];
      out.push_back(q_mul(ils, qsub1(ils)));
      out.push_back(q_mul(im, qsub1(im)));
      out.push_back(q_mul(im, q_sub(c[10], c[7])));
      out.push_back(q_mul(one_minus(im), q_sub(c[10], c[9])));
      out.push_back(q_mul(q_sub(c[8], c[10]), ils));
    }
    out.push_back(q_mul(not_last, q_sub(c[4], c[0])));
    out.push_back(q_mul(not_last, q_sub(c[5], c[1])));
    out.push_back(q_mul(not_last, qsub1(q_sub(c[6], c[2]))));
  }
  return out;
}

QM31 eval_composition_at_point(const std::vector<Instance>& inst,
                               const std::vector<std::vector<std::vector<QM31>>>& sv, QPt oods, const RelElems& elems,
                               QM31 comp_alpha, uint32_t protocol_flags) {
  QM31 acc = q_zero();
  for (auto& ci : inst) {
    const ComponentSpec* sp = ci.spec;
    std::vector<QM31> main(sp->n_cols);
    for (int c = 0; c < sp->n_cols; ++c) main[c] = sv[1][ci.main_start + c][0];
    std::vector<QM31> cons = local_constraints(sp->kind, main);
    QM31 prev = q_zero();
    QM31 shift = q_mul_m(ci.claimed, m_inv((uint32_t)((1ull << ci.log_size) % P31)));
    for (int j = 0; j < sp->n_rel; ++j) {
      const auto* cols = &sv[2][ci.inter_start + 4 * j];
      auto cell = [&](int idx) { return sp->rel_pre[j] ? sv[0][ci.pre_idx[idx]][0] : main[idx]; };
      const int es = sp->rel_elems[j];
      QM31 den = q_sub(cell(sp->rel_val[j]), elems.z[es]);
      if (sp->rel_id[j] >= 0) den = q_add(den, q_mul(elems.alpha[es], cell(sp->rel_id[j])));
      QM31 num = sp->rel_neg[j] ? q_neg(main[sp->rel_mult[j]]) : main[sp->rel_mult[j]];
      QM31 cur, diff;
      if (j < sp->n_rel - 1) {
        cur = q_from_partial_evals(cols[0][0], cols[1][0], cols[2][0], cols[3][0]);
        diff = q_sub(cur, prev);
      } else {
        QM31 prev_row = q_from_partial_evals(cols[0][0], cols[1][0], cols[2][0], cols[3][0]);
        cur = q_from_partial_evals(cols[0][1], cols[1][1], cols[2][1], cols[3][1]);
        diff = q_add(q_sub(q_sub(cur, prev_row), prev), shift);
      }
      cons.push_back(q_sub(q_mul(diff, den), num));
      prev = cur;
    }
    QM31 x = oods.x;
    for (int k = 0; k < ci.log_size - 1; ++k) x = q_sub_m(q_add(q_sqr(x), q_sqr(x)), 1u);
    QM31 zinv = q_inv(x);
    // kernel-slot values -> the protocol's constraint list (constraint-form bits: slots added / dropped, signs)
    const ConstraintLayout L = constraint_layout(*sp, protocol_flags);
    std::vector<QM31> proto(L.n_protocol, q_zero());
    for (int k = 0; k < L.n_kernel; ++k)
      if (L.proto_index[k] >= 0) proto[L.proto_index[k]] = L.neg[k] ? q_neg(cons[k]) : cons[k];
    for (auto& c : proto) acc = q_add(q_mul(acc, comp_alpha), q_mul(c, zinv));
  }
  return acc;
}

}  // namespace lmn
