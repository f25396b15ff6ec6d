// State of one proof while Context::prove walks its phases (prove.cpp drives; the phases live in phase_*.cpp).
// The phases read and write it through LMN_RUN_ALIASES, so their bodies read like the single function they were
// carved out of.  Not part of any boundary.
#pragma once
#include "prover_internal.h"

namespace lmn {

struct TableInfo {
  const ComponentSpec* spec;
  uint64_t n_rows;
  int log_size;
  const uint32_t* rows;
  bool on_device;
};
// one FRI quotient column (the columns of one LDE size accumulated)
struct Quot {
  int log;
  uint32_t* vals;  // 4 x 2^log, or 4 x 2^(log-g) (this rank's rows) when sharded
  bool sharded;
};
// one committed inner FRI layer
struct FriLayer {
  int log;
  uint32_t* vals;  // 4 x 2^log (line evaluation), or this rank's 4 x 2^(log-g) rows when sharded
  bool sharded;
  DevMerkle merkle;
};

struct ProofRun {
  explicit ProofRun(uint32_t protocol_flags) : channel(protocol_flags) {}
  // inputs (borrowed)
  const lmn_table* tables = nullptr;
  size_t n_tables = 0;
  const lmn_settings* settings = nullptr;
  // set-up
  int lb = 0, n_slots = 0, comp_log = 0;
  HostMarks hm;
  EventLog* log = nullptr;
  std::vector<TableInfo> infos;
  Channel channel;
  Proof proof;
  std::unique_ptr<StageTimer> total_guard;
  // commitments: preprocessed, main trace, interaction trace, composition
  DevTree tree0, tree1, tree2, tree3;
  DevTree* trees[4] = {&tree0, &tree1, &tree2, &tree3};
  std::vector<Instance> inst;
  std::vector<uint32_t*> pre_evals;  // tree-0 columns on their trace domain (logup denominators)
  RelElems elems;
  QM31 comp_alpha{};
  // Device-resident Fiat-Shamir of the commitment phases (unsharded proofs): the host enqueues everything up to the
  // sampled values without waiting; the k_chan_* kernels do the transcript steps in between and report back through
  // `d_report`, which the host reads at its first wait (run_oods) to replay and cross-check the same steps.
  bool dev_fs = false;
  DevChannel* d_chan = nullptr;
  DevReport* d_report = nullptr;
  const DevReport* h_report = nullptr;   // page-locked copy, valid after the wait in run_oods
  QM31* d_coeff = nullptr;               // 16 constraint-slot coefficients per component
  QM31* d_maps = nullptr;                // point mappings for launch_eval_tables
  std::vector<EvalJob> eval_jobs;        // one per (column, sample point), sampled_values order (plan_eval_jobs)
  EvalJob* d_eval_jobs = nullptr;        // device copy made by the OODS step's workgroup (device-resident transcript)
  uint32_t bad_mark = 0;                 // this proof's mark of the non-canonical-word verdict
  // Unsharded proofs go on without the sampled values as well (round 6): k_quot_prepare mixes them, draws the quotient
  // randomness and writes the quotient kernels' tables on the device; the host's first wait is the one for the FRI results,
  // where it replays everything from root 1 on (finish_oods_on_host).  LMN_HOST_QUOT=1 keeps the wait in run_oods.
  bool quot_dev = false;
  const QM31* h_vals = nullptr;          // page-locked: the sampled values (eval_jobs order) + the device's quotient randomness
  std::vector<QuotientArgs> qargs;       // the quotient launches, one per LDE size (descending), laid out by run_oods
  // OODS: point 0 = the OODS point, then per trace size the point one trace step before it (plan_sample_points)
  std::map<int, int> prev_point_of_log;
  std::vector<Pt> neg_step;
  QPt oods{};
  std::vector<QPt> points;
  std::vector<std::vector<std::vector<int>>> spoints;     // sample point indices per tree / column, sampled_values order
  std::vector<std::vector<std::vector<QM31>>> sampled;
  // FRI
  QM31 quot_alpha{};
  bool sh = false;   // sharded proof
  int g = 0;         // log2 ranks
  int fri_T = 0;     // quotient columns / layers of more than 2^fri_T rows are split into row blocks
  std::vector<int> sizes;   // LDE log sizes present, descending
  std::vector<Quot> quots;
  DevMerkle first_merkle;
  std::vector<ColRef> first_cols;
  std::vector<FriLayer> inner;
  // What the FRI commit loop reads from and leaves on the device, laid out before the quotient kernels are launched
  // (plan_fri_buffers) so that its tables travel in the quotient phase's one upload and its results come back in one
  // download: roots | alphas | last layer in one block; the device channel's start state; for unsharded proofs the
  // layers of the single-block tail (k_fri_tail) with their trees and the table that names them.
  struct FriPlan {
    int max_layers = 0, last_size_log = 0;
    uint32_t* d_out = nullptr;      // max_layers x 8 root words | max_layers x QM31 alphas | 4 x 2^last_size_log last layer
    uint32_t* d_roots = nullptr;
    QM31* d_alphas = nullptr;
    uint32_t* d_last = nullptr;
    size_t out_bytes = 0;
    DevChannel* d_chan = nullptr;
    int tail_log = -1;              // first layer of the tail (-1: none planned)
    uint32_t* tail_first = nullptr; // that layer's values
    std::vector<FriLayer> tail_layers;
    std::vector<FriTailLayer> tail_table;   // host copy, staged by plan_fri_buffers
    FriTailLayer* d_tail = nullptr;
    int planned_ls0 = -1;           // plan_fri_layout ran for a first layer of this size
  } fri;
  std::vector<QM31> last_vals;
  int last_log = 0;
  std::vector<uint32_t> queries;
  std::map<int, std::vector<uint32_t>> pos_by_log;
  bool sharded_log(int lg) const { return sh && lg > fri_T; }
  ProofRun(const ProofRun&) = delete;
  ProofRun& operator=(const ProofRun&) = delete;
};

// the four coordinate columns of a secure column as tree columns
inline void secure_columns(const uint32_t* vals, int lg, bool s, int g, std::vector<ColRef>& out) {
  const uint64_t stride = s ? (1ull << (lg - g)) : (1ull << lg);
  for (int k = 0; k < 4; ++k) out.push_back({vals + (uint64_t)k * stride, lg, s});
}

#define LMN_RUN_ALIASES(r)                                                                                              \
  HostMarks& hm = (r).hm; EventLog* const log = (r).log; const int lb = (r).lb, n_slots = (r).n_slots; int& comp_log = (r).comp_log; \
  std::vector<TableInfo>& infos = (r).infos; Channel& channel = (r).channel; Proof& proof = (r).proof;                  \
  DevTree &tree0 = (r).tree0, &tree1 = (r).tree1, &tree2 = (r).tree2, &tree3 = (r).tree3; DevTree* (&trees)[4] = (r).trees; \
  std::vector<Instance>& inst = (r).inst; std::vector<uint32_t*>& pre_evals = (r).pre_evals; RelElems& elems = (r).elems; \
  QM31& comp_alpha = (r).comp_alpha; QPt& oods = (r).oods; std::vector<QPt>& points = (r).points;                       \
  std::vector<std::vector<std::vector<int>>>& spoints = (r).spoints;                                                    \
  std::vector<std::vector<std::vector<QM31>>>& sampled = (r).sampled; QM31& quot_alpha = (r).quot_alpha;                \
  bool& sh = (r).sh; int &g = (r).g, &fri_T = (r).fri_T; std::vector<int>& sizes = (r).sizes;                           \
  std::vector<Quot>& quots = (r).quots; DevMerkle& first_merkle = (r).first_merkle;                                     \
  std::vector<ColRef>& first_cols = (r).first_cols; std::vector<FriLayer>& inner = (r).inner;                           \
  std::vector<QM31>& last_vals = (r).last_vals; int& last_log = (r).last_log; std::vector<uint32_t>& queries = (r).queries; \
  std::map<int, std::vector<uint32_t>>& pos_by_log = (r).pos_by_log;                                                    \
  (void)hm; (void)log; (void)lb; (void)n_slots; (void)comp_log; (void)infos; (void)channel; (void)proof; (void)tree0;   \
  (void)tree1; (void)tree2; (void)tree3; (void)trees; (void)inst; (void)pre_evals; (void)elems; (void)comp_alpha;       \
  (void)oods; (void)points; (void)spoints; (void)sampled; (void)quot_alpha; (void)sh; (void)g; (void)fri_T; (void)sizes; \
  (void)quots; (void)first_merkle; (void)first_cols; (void)inner; (void)last_vals; (void)last_log; (void)queries;       \
  (void)pos_by_log

}  // namespace lmn
