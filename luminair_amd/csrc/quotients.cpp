// Host side of `QuotientOps::accumulate_quotients`: sample batches and line coefficients of the
// columns of one LDE size.
#include "prover_internal.h"

namespace lmn {

// ------------------------------------------------------------------------------------ FRI quotients
// Host side of `accumulate_quotients` for the columns of one LDE size: ColumnSampleBatch::new_vec groups
// the (column, point, value) samples by point in first-appearance order; per sample the line through
// (p.y, v) and (conj p.y, conj v) gives coefficients a, b, c scaled by alpha^k.
QuotientArgs Context::make_quotient_args(int ls, const std::vector<const uint32_t*>& cols,
                                         const std::vector<std::vector<std::pair<int, QM31>>>& samples,
                                         const std::vector<QPt>& points, QM31 quot_alpha, bool alloc_out) {
  std::vector<int> batch_point;
  std::vector<std::vector<std::pair<int, QM31>>> batch_cols;
  batch_point.reserve(QUOT_MAX_BATCH + 1);
  batch_cols.reserve(QUOT_MAX_BATCH + 1);
  for (size_t c = 0; c < cols.size(); ++c)
    for (auto& sm : samples[c]) {
      size_t b = 0;
      while (b < batch_point.size() && batch_point[b] != sm.first) ++b;
      if (b == batch_point.size()) {
        batch_point.push_back(sm.first);
        batch_cols.emplace_back();
        batch_cols.back().reserve(cols.size());
      }
      batch_cols[b].push_back({(int)c, sm.second});
    }
  if (batch_point.size() > (size_t)QUOT_MAX_BATCH) throw LmnError(LMN_ERR_INTERNAL, "too many sample batches");
  QuotientArgs a{};
  QuotDev qd{};
  a.log_size = ls;
  a.nbatch = (int)batch_point.size();
  std::vector<int> col_idx;
  std::vector<QM31> coeff_c;
  col_idx.reserve(2 * cols.size());
  coeff_c.reserve(2 * cols.size());
  for (size_t b = 0; b < batch_point.size(); ++b) {
    QPt pt = points[batch_point[b]];
    a.batch_start[b] = (int)col_idx.size();
    QM31 alpha = q_one(), A = q_zero(), B = q_zero();
    for (auto& cv : batch_cols[b]) {
      alpha = q_mul(alpha, quot_alpha);
      QM31 val = cv.second;
      QM31 la = q_sub(q_conj(val), val);
      QM31 lc = q_sub(q_conj(pt.y), pt.y);
      QM31 lbb = q_sub(q_mul(val, lc), q_mul(la, pt.y));
      A = q_add(A, q_mul(alpha, la));
      B = q_add(B, q_mul(alpha, lbb));
      col_idx.push_back(cv.first);
      coeff_c.push_back(q_mul(alpha, lc));
    }
    qd.A[b] = A;
    qd.B[b] = B;
    qd.batch_coeff[b] = q_pow(quot_alpha, batch_cols[b].size());
    qd.prx[b] = {pt.x.a, pt.x.b};
    qd.pix[b] = {pt.x.c, pt.x.d};
    qd.pry[b] = {pt.y.a, pt.y.b};
    qd.piy[b] = {pt.y.c, pt.y.d};
  }
  a.batch_start[batch_point.size()] = (int)col_idx.size();
  if (col_idx.size() > (size_t)QUOT_MAX_ENTRIES) throw LmnError(LMN_ERR_INTERNAL, "too many column samples");
  std::vector<QuotEntry> entries(col_idx.size());
  for (size_t k = 0; k < col_idx.size(); ++k) entries[k] = {cols[col_idx[k]], coeff_c[k]};
  a.entries = upload_vec(entries);
  a.dev = (const QuotDev*)stage_upload(&qd, sizeof qd);
  a.tw_y = twY_[ls];
  a.tw_x = ls >= 2 ? twX_[ls] : nullptr;
  a.row0 = 0;
  a.log_rows = ls;
  a.out_stride = 1ull << ls;
  a.out = alloc_out ? arena_.alloc_words(4ull << ls) : nullptr;
  return a;
}

}  // namespace lmn
