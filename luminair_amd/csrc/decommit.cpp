// Decommitment planning (stwo MerkleProver::decommit / FRI witness order): turns the query
// positions into an ordered list of device references that ONE gather launch fetches (k_gather), recomputing the tree
// nodes a fused launch never wrote.
#include "prover_internal.h"

namespace lmn {

// ------------------------------------------------------------------------------------ decommit planning
// A run of device words to fetch.  owner < 0: every rank holds it; otherwise only rank `owner` does (row-block
// sharded column or Merkle layer) and ptr is meaningful on that rank alone.
static Ref col_ref(const ColRef& c, uint64_t row, int g) {
  if (!c.sharded) return {c.ptr + row, 1, -1};
  const int sh = c.log - g;
  return {c.ptr + (row & ((1ull << sh) - 1)), 1, (int)(row >> sh)};
}
static Ref node_ref(const DevMerkle& m, int layer, uint64_t node, std::vector<MerkleRecompute>& jobs) {
  if (!m.layers[layer]) {   // a level its launch kept in registers (MerkleCut)
    for (auto& c : m.cuts)
      if (layer <= c.start_log && layer > c.start_log - c.depth) {
        jobs.push_back({c.prev, c.sg, c.ncols, 1u << c.start_log, c.below, c.below_ncols, (uint32_t)node, c.start_log - layer, 0u});
        return {nullptr, 8, -1, (int)jobs.size() - 1};
      }
    throw LmnError(LMN_ERR_INTERNAL, "merkle: layer without storage");
  }
  if (m.g == 0 || layer <= m.g) return {m.layers[layer] + node * 8, 8, -1};
  const int sh = layer - m.g;
  return {m.layers[layer] + (node & ((1ull << sh) - 1)) * 8, 8, (int)(node >> sh)};
}

void release_host_scratch(void* p) { delete static_cast<HostScratch*>(p); }

// MerkleProver::decommit: emits device references in output order
void plan_merkle_decommit(const DevMerkle& m, const std::vector<ColRef>& cols_sorted, int g,
                                 const std::map<int, std::vector<uint32_t>>& queries, std::vector<Ref>& queried,
                                 std::vector<Ref>& hash_wit, std::vector<Ref>& col_wit, std::vector<MerkleRecompute>& jobs) {
  size_t pos = 0;
  std::vector<uint32_t> last, total;
  last.reserve(16);
  total.reserve(16);
  for (int log = m.max_log; log >= 0; --log) {
    size_t start = pos;
    while (pos < cols_sorted.size() && cols_sorted[pos].log == log) ++pos;
    bool have_prev = log < m.max_log;
    static const std::vector<uint32_t> kNone;
    auto it = queries.find(log);
    const std::vector<uint32_t>& colq = it != queries.end() ? it->second : kNone;
    size_t pi = 0, ci = 0;
    total.clear();
    while (pi < last.size() || ci < colq.size()) {
      uint32_t node;
      if (pi < last.size() && ci < colq.size())
        node = std::min(last[pi] / 2, colq[ci]);
      else if (pi < last.size())
        node = last[pi] / 2;
      else
        node = colq[ci];
      if (have_prev) {
        if (pi < last.size() && last[pi] == 2 * node)
          ++pi;
        else
          hash_wit.push_back(node_ref(m, log + 1, 2ull * node, jobs));
        if (pi < last.size() && last[pi] == 2 * node + 1)
          ++pi;
        else
          hash_wit.push_back(node_ref(m, log + 1, 2ull * node + 1, jobs));
      }
      bool is_q = ci < colq.size() && colq[ci] == node;
      if (is_q) ++ci;
      for (size_t c = start; c < pos; ++c) (is_q ? queried : col_wit).push_back(col_ref(cols_sorted[c], node, g));
      total.push_back(node);
    }
    last.swap(total);
  }
}

std::vector<uint32_t> fold_positions(const std::vector<uint32_t>& p, int n) {
  std::vector<uint32_t> out;
  out.reserve(p.size());
  for (auto v : p) {
    uint32_t q = v >> n;
    if (out.empty() || out.back() != q) out.push_back(q);
  }
  return out;
}

// compute_decommitment_positions_and_witness_evals with fold_step = 1; `cols` = the 4 coordinate columns
void plan_fri_witness(const ColRef (&cols)[4], int g, const std::vector<uint32_t>& qpos,
                             std::vector<uint32_t>& dec_pos, std::vector<Ref>& wit) {
  // (planning runs between the proof's last two waits: no allocation per pair)
  dec_pos.reserve(dec_pos.size() + 2 * qpos.size());
  size_t i = 0;
  while (i < qpos.size()) {
    const uint32_t start = (qpos[i] >> 1) << 1;
    bool have[2] = {false, false};   // which of the pair's two positions are queried (sorted, distinct positions)
    while (i < qpos.size() && ((qpos[i] >> 1) << 1) == start) have[qpos[i++] - start] = true;
    for (uint32_t pos = start; pos < start + 2; ++pos) {
      dec_pos.push_back(pos);
      if (have[pos - start]) continue;
      for (int k = 0; k < 4; ++k) wit.push_back(col_ref(cols[k], pos, g));
    }
  }
}

}  // namespace lmn
