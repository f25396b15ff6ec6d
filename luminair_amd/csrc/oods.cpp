// `PolyOps::eval_at_point` for every (column, point) of a proof at once (behind stwo::prover::prove,
// /root/reference/crates/prover/src/prover.rs:312): host side of k_eval_tables / k_eval_at_point / k_eval_reduce.
#include "prover_internal.h"

namespace lmn {

// ------------------------------------------------------------------------------------ OODS evaluation
std::vector<QM31> Context::eval_at_points(const std::vector<EvalJob>& jobs, const std::vector<QPt>& points,
                                          int max_log, bool split, const QM31* d_maps_in, int n_points,
                                          const EvalJob* d_jobs_in, const QM31** device_out) {
  const int np = d_maps_in ? n_points : (int)points.size();
  const uint32_t lo_n = 1u << EVAL_LB;
  const int hi_bits = max_log > EVAL_LB ? max_log - EVAL_LB : 0;
  const uint32_t hi_n = 1u << hi_bits;
  const int nmaps = std::max(max_log, EVAL_LB);
  // mappings per point: y, x, pi(x), pi^2(x), ...  (tables themselves are expanded on the device)
  std::vector<QM31> maps(d_maps_in ? 0 : (size_t)np * nmaps);
  for (int p = 0; p < np && !d_maps_in; ++p) {
    QM31* mp = &maps[(size_t)p * nmaps];
    mp[0] = points[p].y;
    mp[1] = points[p].x;
    QM31 cur = points[p].x;
    for (int k = 2; k < nmaps; ++k) {
      cur = q_sub_m(q_add(q_sqr(cur), q_sqr(cur)), 1u);
      mp[k] = cur;
    }
  }
  int max_chunks = eval_num_chunks(max_log);
  const EvalJob* d_jobs = d_jobs_in ? d_jobs_in : upload_vec(jobs);
  const QM31* d_maps = d_maps_in ? d_maps_in : upload_vec(maps);
  QM31* d_lo = (QM31*)arena_.alloc_bytes((size_t)np * lo_n * sizeof(QM31));
  QM31* d_hi = (QM31*)arena_.alloc_bytes((size_t)np * hi_n * sizeof(QM31));
  QM31* d_part = (QM31*)arena_.alloc_bytes(jobs.size() * (size_t)max_chunks * sizeof(QM31));
  split = split && shard_.active && shard_.world > 1;
  const uint32_t W = split ? shard_.world : 1u, R = split ? shard_.rank : 0u;
  const size_t nj = jobs.size();
  // slot r: rank r's partial sums.  An unsharded proof's reduce kernel writes the values straight to page-locked memory -
  // or, device_out, leaves them in device memory for k_quot_prepare and nobody waits here
  if (device_out && split) throw LmnError(LMN_ERR_INTERNAL, "eval_at_points: device-resident values of a sharded proof");
  QM31* d_out = (split || device_out) ? (QM31*)arena_.alloc_bytes((size_t)W * nj * sizeof(QM31)) : (QM31*)result_block(nj * sizeof(QM31));
  launch_eval_tables(d_maps, nmaps, np, d_lo, d_hi, hi_n, hi_bits, stream_);
  launch_eval_at_point(d_jobs, (int)nj, d_lo, d_hi, hi_n, max_log, d_part, max_chunks, stream_, R, W);
  launch_eval_reduce(d_jobs, (int)nj, d_part, max_chunks, d_out + (size_t)R * nj, stream_);
  if (split) gather_columns((uint32_t*)d_out, 0, 1, nj * 4);
  if (device_out) {
    *device_out = d_out;
    return {};
  }
  const QM31* res = split ? (const QM31*)stage_download(d_out, (size_t)W * nj * sizeof(QM31)) : d_out;
  lmn_sync(stream_);
  std::vector<QM31> out(res, res + nj);
  for (uint32_t r = 1; r < W; ++r)
    for (size_t j = 0; j < nj; ++j) out[j] = q_add(out[j], res[(size_t)r * nj + j]);
  return out;
}

}  // namespace lmn
