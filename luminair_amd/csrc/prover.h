// Device-resident prover context and the `prove` orchestration that replaces
// /root/reference/crates/prover/src/prover.rs:28-319 on MI355X.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/luminair_hip.h"
#include "host.h"
#include "kernels.h"

namespace lmn {

constexpr int MAX_REL = 7;
struct ComponentSpec {
  int kind;
  int n_cols;
  int is_last_col;
  int n_rel;
  int rel_mult[MAX_REL], rel_val[MAX_REL], rel_id[MAX_REL];  // rel_id < 0: width-1 relation (value only)
  int n_local;                // number of local constraints (including zero slots)
  int rel_elems[MAX_REL];     // ELEMS_*: 0 NodeElements, 1 RangeCheckLookup, 2 SinLookup, 3 Exp2Lookup, 4 Log2Lookup
  int rel_neg[MAX_REL];       // numerator is -mult
  int rel_pre[MAX_REL];       // rel_val / rel_id index the component's preprocessed columns
  int n_pre;                  // preprocessed (tree 0) columns read (0..2)
  int pre_id[2];              // PRE_*: position in PreProcessedTrace order (preprocessed.rs:157-179)
  int n_pad;                  // extra non-zero padding cells besides is_last_col = 1
  int pad_col[4];
  uint32_t pad_val[4];
};
const ComponentSpec* component_spec(int kind);
enum { ELEMS_NODE = 0, ELEMS_RANGE_CHECK = 1, ELEMS_SIN = 2, ELEMS_EXP2 = 3, ELEMS_LOG2 = 4, N_ELEMS = 5 };
// tree-0 column order before the stable size sort: sin_lut_0/1, exp2_lut_0/1, log2_lut_0/1, range_check_8
enum { PRE_SIN0 = 0, PRE_EXP20 = 2, PRE_LOG20 = 4, PRE_RANGE_CHECK = 6, N_PRE_IDS = 7 };
// relation element sets drawn after the main commitment (components/mod.rs:227-235, lookups/mod.rs:44-51)
struct RelElems {
  QM31 z[N_ELEMS], alpha[N_ELEMS];
  bool drawn[N_ELEMS] = {false, false, false, false, false};
};
class Channel;
RelElems draw_relation_elements(Channel& channel, uint32_t protocol_flags);
// the element set each draw_felts(2) of LuminairInteractionElements::draw fills, in draw order; returns the number of draws
int relation_draw_sets(uint32_t protocol_flags, int sets_out[5]);
inline int claim_slots(uint32_t protocol_flags) { return (protocol_flags & LMN_PV_CLAIM17) ? 17 : 8; }

// Constraint slots of a component under the protocol's constraint-form bits (LMN_PV_*_SLOT(S) / _NEG, luminair_hip.h).
// "Kernel slot" k = the k-th value k_composition<KIND> (and local_constraints() on the host) emits: the local constraints
// in `evaluate` order with the KAT-era shape (eval_fixed_mul: two slots; recip / sqrt / rem: one), then one per relation.
// The protocol may give a helper one slot more or less and the opposite sign; that only changes WHICH power of the
// composition randomness multiplies a kernel slot, so it is decided here on the host and the kernels never see the bits.
struct ConstraintLayout {
  int n_kernel = 0;      // n_local + n_rel
  int n_protocol = 0;    // constraints the component contributes to the composition polynomial (zero slots included)
  int proto_index[16];   // per kernel slot: index among the component's protocol constraints; -1 = no such constraint
  bool neg[16];          // the protocol's constraint is minus the kernel's value
};
ConstraintLayout constraint_layout(const ComponentSpec& sp, uint32_t protocol_flags);

// one component of a proof: shared by the prover and the host-side verifier
struct Instance {
  const ComponentSpec* spec;
  int log_size;
  int main_start, inter_start;  // column offsets inside trees 1 / 2
  QM31 claimed;
  uint32_t* halo = nullptr;               // sharded proofs with all_to_all: neighbour blocks of the last logup group
  const QM31* d_claimed_shift = nullptr;  // device [claimed, shift] (prover only)
  uint32_t* trace_evals = nullptr;        // device, n_cols x 2^log_size (prover only)
  bool rows_sharded = false;              // sharded proofs: trace_evals holds this rank's block of 2^(log_size - g) rows only
  // trace_evals == nullptr: the evaluations were never stored column-major (the transpose ran inside the interpolation,
  // fft_fixed.hip k_fft_rows_fx): readers take them from the table's rows on the device, padded beyond rows_n
  const uint32_t* rows_dev = nullptr;
  uint64_t rows_n = 0;
  PadRow pad{};
  int pre_idx[2] = {-1, -1};    // tree-0 column indices of the component's preprocessed columns
};
// sum_k c_k(oods)/Z_k(oods) * alpha^(N-1-k) from the sampled mask values
QM31 eval_composition_at_point(const std::vector<Instance>& inst, const std::vector<std::vector<std::vector<QM31>>>& sv,
                               QPt oods, const RelElems& elems, QM31 comp_alpha, uint32_t protocol_flags);
// tree-0 layout implied by the components present: fills Instance::pre_idx, returns the columns' log
// sizes in tree order (a LUT column has the log size of its lookup component)
std::vector<int> assign_preprocessed(std::vector<Instance>& inst);
// verify(proof, settings): crates/verifiers/rust/src/verifier.rs:21-143 (host only, no GPU work)
// `expect`: the verifier's own PcsConfig + protocol variant (a proof announcing another config is rejected);
// `settings` (may be null): cross-checked against the tree-0 layout the claim implies
void verify_proof(const uint8_t* data, size_t len, const lmn_config& expect, const lmn_settings* settings,
                  lmn_verify_report* report = nullptr);

// bump allocator over one device slab; reset per proof
class Arena {
 public:
  ~Arena();
  void reserve(size_t bytes);
  void reset() { off_ = 0; }
  void* alloc_bytes(size_t bytes);
  uint32_t* alloc_words(size_t words) { return (uint32_t*)alloc_bytes(words * 4); }
  uint64_t word_offset(const void* p) const { return (uint64_t)((const char*)p - base_) / 4; }
  const uint32_t* base_words() const { return (const uint32_t*)base_; }
  size_t capacity() const { return cap_; }
  size_t used() const { return off_; }

 private:
  char* base_ = nullptr;
  size_t cap_ = 0, off_ = 0;
};

struct DevColumn {
  int log_size;      // polynomial (coefficient) log size
  uint32_t* coeffs;  // 2^log_size
  uint32_t* lde;     // 2^(log_size + log_blowup), bit-reversed canonic-domain evaluations
  bool sharded = false;  // lde holds only this rank's block of 2^(log_size + log_blowup - g) rows
  int owner = -1;        // >= 0: the coefficients exist on that rank only (column-parallel interpolation)
};

// A column as the Merkle / decommitment code sees it.  sharded: ptr is this rank's aligned block of
// 2^(log - g) rows (row r of the column lives on rank r >> (log - g)); otherwise ptr holds all 2^log rows.
struct ColRef {
  const uint32_t* ptr;
  int log;
  bool sharded;
};

// A fused Merkle launch whose lanes kept their per-lane subtree in registers only: layers start_log, start_log - 1, ...,
// start_log - depth + 1 of the tree do not exist in HBM (7/8 of a big tree's 32-byte nodes, 1 GB of writes per 2^20-row
// proof, of which the decommitment reads a few dozen).  What the decommitment needs of them is recomputed from the
// launch's start level (MerkleRecompute, k_gather).
struct MerkleCut {
  int start_log, depth;
  const uint32_t* prev;
  MerkleSegs sg;
  int ncols;
  const uint32_t* below = nullptr;   // the launch hashed the leaf level under its start level itself (MerkleFold::below)
  int below_ncols = 0;
};

struct DevMerkle {
  int max_log = -1;
  // layers[k]: 2^k hashes of 8 words.  In a sharded tree (g > 0) the layers k > g hold only this rank's
  // 2^(k-g) nodes (node n lives on rank n >> (k - g)); layers k <= g are complete on every rank.
  // A null layer was not written: see `cuts`.
  std::vector<uint32_t*> layers;
  std::vector<MerkleCut> cuts;
  int g = 0;
  Hash32 root;
  const uint32_t* root_pinned = nullptr;  // pending async download of layers[0]
  void finish_root() {
    if (root_pinned) memcpy(root.w, root_pinned, 32);
    root_pinned = nullptr;
  }
};

struct DevTree {
  std::vector<DevColumn> cols;
  DevMerkle merkle;
};

struct StageTimer;

struct ProofRun;   // state of one proof across the phases of Context::prove (prove_run.h)

class Context {
 public:
  Context(int device, const lmn_config& cfg);
  ~Context();
  std::vector<uint8_t> prove(const lmn_table* tables, size_t n_tables, const lmn_settings* settings);
#ifdef LMN_BATCH
  // lock-step batches (batch.h): every member context issues its work on the group's one stream
  void adopt_stream(lmn_stream_t s);
  // the set-up `prove` would do for these tables on its first call with such a shape (twiddle tables), done outside
  // lock-step; malformed tables are left for `prove` to report
  void prepare_for(const lmn_table* tables, size_t n_tables);
#endif

  // level-2 ops
  void op_interpolate(uint32_t* cols, uint32_t ncols, uint32_t log_size);
  void op_evaluate(const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain, uint32_t* out);
  void op_evaluate_block(const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                         uint32_t log_blocks, uint32_t block, uint32_t* out);
  void op_merkle_root(const uint32_t* const* cols, const uint32_t* log_sizes, uint32_t ncols, uint8_t root[32]);
  void op_eval_at_point(const uint32_t* coeffs, uint32_t log_size, const uint32_t pt[8], uint32_t out[4]);
  void op_fft_selftest(uint32_t log_size, uint32_t ncols);
  void op_accumulate_quotients(uint32_t log_size, const uint32_t* const* cols, uint32_t ncols, const uint32_t* sample_col,
                               const uint32_t* sample_point, const uint32_t* sample_values, uint32_t nsamples,
                               const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4], uint32_t* out);
  void op_fold(int circle, uint32_t* dst, const uint32_t* src, uint32_t log_src, const uint32_t alpha[4]);

  // single-proof sharding (lmn_ctx_set_shard / lmn_ctx_set_shard_rccl)
  static void check_shard_args(uint32_t rank, uint32_t world, uint32_t fri_min_log, const lmn_collective* coll);
  void set_shard(uint32_t rank, uint32_t world, uint32_t fri_min_log, const lmn_collective* coll);
  void set_shard_rccl(uint32_t rank, uint32_t world, uint32_t fri_min_log, const uint8_t* id);
  void clear_shard();

  // level-2 ops on device handles (level2.cpp)
  lmn_col* col_alloc(uint32_t ncols, uint32_t log_size, bool zero);
  lmn_col* col_from_cpu(const uint32_t* host, uint32_t ncols, uint32_t log_size);
  void col_to_cpu(const lmn_col* c, uint32_t* host);
  void col_free(lmn_col* c);
  lmn_col* col_view(const lmn_col* c, uint32_t first, uint32_t n);
  void col_bit_reverse(lmn_col* c);
  void col_precompute_twiddles(uint32_t log_size);
  void col_interpolate(lmn_col* c);
  lmn_col* col_evaluate(const lmn_col* coeffs, uint32_t log_domain);
  lmn_col* col_evaluate_block(const lmn_col* coeffs, uint32_t log_domain, uint32_t log_blocks, uint32_t block);
  lmn_col* col_extend(const lmn_col* coeffs, uint32_t log_size);
  void col_eval_at_point(const lmn_col* coeffs, uint32_t column, const uint32_t pt[8], uint32_t out[4]);
  lmn_tree* col_commit(const lmn_col* const* cols, uint32_t n);
  void tree_layer_to_cpu(const lmn_tree* t, uint32_t layer_log, uint8_t* out);
  void tree_free(lmn_tree* t);
  void col_accumulate(lmn_col* dst, const lmn_col* src);
  lmn_col* col_accumulate_quotients(const lmn_col* const* cols, uint32_t n, const uint32_t* sample_col,
                                    const uint32_t* sample_point, const uint32_t* sample_values, uint32_t nsamples,
                                    const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4]);
  lmn_col* col_fold_line(const lmn_col* src, const uint32_t alpha[4]);
  void col_fold_circle_into_line(lmn_col* dst, const lmn_col* src, const uint32_t alpha[4]);
  lmn_col* col_decompose(const lmn_col* f, uint32_t lambda_out[4]);
  lmn_col* col_logup(uint32_t kind, const lmn_col* main, const lmn_col* pre, const uint32_t* elems, uint32_t claimed_out[4]);
  void col_composition(uint32_t kind, const lmn_col* main_lde, const lmn_col* inter_lde, const lmn_col* pre_lde,
                       const uint32_t* elems, const uint32_t claimed[4], const uint32_t* coeffs, uint32_t n_coeffs,
                       lmn_col* acc);

  void* upload(const void* host, size_t bytes);
  void* device_alloc(size_t bytes);
  void upload_to(const void* host, size_t bytes, void* dst);
  void device_copy(void* dst, const void* src, size_t bytes);
  void download(const void* device, void* host, size_t bytes);
  void trace_reduce(bool is_max, const int32_t* input, uint64_t front, uint64_t dim, uint64_t back,
                    const lmn_node_info& info, uint32_t* rows, uint64_t row_offset, int32_t* out);
  void trace_elementwise(uint32_t kind, const int32_t* lhs, const lmn_view* lv, const int32_t* rhs, const lmn_view* rv,
                         uint64_t n, const lmn_node_info& info, uint32_t* rows, uint64_t row_offset, int32_t* out,
                         uint32_t* aux = nullptr);
  void trace_contiguous(const int32_t* input, uint64_t in_size, const lmn_view* view, uint64_t out_size,
                        const lmn_node_info& info, uint32_t* rows, uint64_t row_offset, int32_t* out);
  void trace_lut(uint32_t kind, const int32_t* input, const lmn_view* view, uint64_t n, const lmn_node_info& info,
                 const uint32_t* lut_col1, const lmn_range* ranges, uint32_t n_ranges, uint32_t* mult, uint32_t* rows,
                 uint64_t row_offset, int32_t* out);
  void device_free(void* p);

  lmn_config cfg;
  lmn_timings timings{};
  bool profiling = false;  // record HIP events around stages/kernels (lmn_set_profiling)
  void* event_log = nullptr;  // EventLog (prover.cpp)
  void* host_scratch = nullptr;  // HostScratch (prover.cpp): decommit-planning storage reused across proofs
  std::string last_error;

 private:
  // the phases of prove(), in transcript order (prove.cpp, phase_*.cpp)
  void run_setup(ProofRun& r);
  void run_preprocessed(ProofRun& r);
  void run_main_trace(ProofRun& r);
  void run_interaction(ProofRun& r);
  void run_composition(ProofRun& r);
  void run_oods(ProofRun& r);
  void replay_device_transcript(ProofRun& r, const std::function<void(QM31)>& set_points);
  void set_sample_points(ProofRun& r, QM31 t);          // phase_oods.cpp: the OODS point of the draw t and the mask points
  void enqueue_quotient_tables(ProofRun& r, const QM31* d_vals);   // phase_oods.cpp: k_quot_prepare + the quotient launches' arguments
  void finish_oods_on_host(ProofRun& r);                 // phase_oods.cpp: the host's replay of everything from root 1 to the quotient randomness
  void check_composition_identity(ProofRun& r);          // phase_oods.cpp: stwo's OODS sanity check on the sampled values
  void run_quotients(ProofRun& r);
  void run_fri_commit(ProofRun& r);
  void run_queries(ProofRun& r);
  void run_decommit(ProofRun& r);
  std::vector<uint8_t> run_finish(ProofRun& r);

  void ensure_twiddles(int max_domain_log);
  TwPtrs tw(int domain_log) const;
  TwPtrs itw(int domain_log) const;
  // evaluations -> coefficients (`coeffs` may equal `evals`).  Unsharded proofs of a supported size get the blown-up
  // evaluation in the same three launches (launch_interp_extend): returns the LDE (ncols x 2^(log + log_blowup),
  // arena) or nullptr when lde_and_merkle has to produce it.
  //   Sharded proofs whose collective offers all_to_all (SURVEY.md section 8e stages A / B): this rank interpolates and
  //   extends only its contiguous share of the columns and receives its row block of every column's LDE through one
  //   all-to-all; coefficients then exist on the owning rank only (CommitOut::first), and with halo_first >= 0 the two
  //   neighbouring row blocks of the 4 columns halo_first .. halo_first + 3 (mask offset -1 of the last logup column
  //   group) arrive through a second, small all-to-all.
  struct CommitOut {
    uint32_t* lde = nullptr;      // nullptr: lde_and_merkle produces it
    uint64_t stride = 0;          // words between the columns of `lde`
    bool sharded = false;         // `lde` holds this rank's row block only
    bool owned = false;           // column c's coefficients live on rank `owner_of(c)` only
    int first[9] = {0};           // columns [first[r], first[r + 1]) belong to rank r
    uint32_t* halo = nullptr;     // 4 columns x 2^(log + 1) words, the two neighbour blocks filled in
    int owner_of(int c) const {
      if (!owned) return -1;
      int r = 0;
      while (first[r + 1] <= c) ++r;
      return r;
    }
  };
  // evals_row_blocks: `evals` holds this rank's row block of every column (row-parallel transposes / logup); the blocks
  // travel to the columns' owners through one more all-to-all before stage A
  CommitOut interpolate_for_commit(uint32_t* coeffs, const uint32_t* evals, int ncols, int log_size, int halo_first = -1,
                                   bool evals_row_blocks = false);
  bool shard_all_to_all() const;
  bool shard_a2a_columns(int log_size) const;   // stage A / B for columns of this size
  bool shard_rows_front(int log_size) const;    // row-parallel transposes and logup fractions for tables of this size
  // commit `cols` (coefficients already in place) -> LDE + Merkle (row-block sharded when a shard is set)
  // `step` (device-resident transcript): made by the launch that produces the root, or by launch_chan_step behind it
  void lde_and_merkle(DevTree& tree, bool fetch_root = true, DevChannel* step_ch = nullptr, const ChanStep* step = nullptr);
  // Merkle tree over columns sorted by size (descending, stable).  sharded: every rank hashes the subtree over its
  // row block, the subtree roots are all-gathered and the top log2(world) levels are hashed on every rank.
  // `fold` (unsharded FRI layers of more than 2^10 rows only): the leaf level computes the layer as the fold of the
  // previous one while hashing it (MerkleFold, kernels.h); cols_sorted then names the 4 columns it writes.
  void build_merkle(DevMerkle& m, const std::vector<ColRef>& cols_sorted, DevChannel* ch = nullptr,
                    QM31* alpha_out = nullptr, uint32_t* root_copy = nullptr, bool sharded = false,
                    const MerkleFold* fold = nullptr, const ChanStep* step = nullptr);
  // null entries of `layers` are allocated from the arena as the launches reach them; with `cuts` given, the levels a
  // fused launch keeps in registers stay null and are recorded there instead
  void build_merkle_levels(std::vector<uint32_t*>& layers, int max_log,
                           const std::vector<std::vector<const uint32_t*>>& per_level, DevChannel* ch, QM31* alpha_out,
                           uint32_t* root_copy, const MerkleFold* fold = nullptr, std::vector<MerkleCut>* cuts = nullptr,
                           const ChanStep* step = nullptr);
  // in-place all-gather of `ncols` columns `col_stride` words apart: rank r owns words [r*w, (r+1)*w) of each
  void gather_columns(uint32_t* base, uint64_t col_stride, int ncols, uint64_t words_per_rank);
  void merkle_layer_timed(const uint32_t* prev, const uint32_t* const* cols, int ncols, uint32_t size, uint32_t* out);
  void plan_fri_layout(struct ProofRun& r, int ls0, int smallest_quot_log);   // phase_fri.cpp
  void plan_fri_buffers(struct ProofRun& r);   // phase_fri.cpp
  void plan_sample_points(struct ProofRun& r);   // phase_oods.cpp
  void plan_eval_jobs(struct ProofRun& r);       // phase_oods.cpp
  void plan_oods_step(struct ProofRun& r, ChanStep& step);   // phase_oods.cpp: ChanStep kind 3 for the composition tree's root
  QuotientArgs make_quotient_args(int ls, const std::vector<const uint32_t*>& cols,
                                  const std::vector<std::vector<std::pair<int, QM31>>>& samples,
                                  const std::vector<QPt>& points, QM31 quot_alpha, bool alloc_out = true);
  // split: a sharded proof evaluates 1/world of every polynomial's coefficient chunks per rank and all-gathers the
  // partial sums (16 B per sample and rank)
  // d_maps (device, n_points x max(max_log, EVAL_LB) mappings, written by the ChanStep of kind 3) replaces `points` when given
  // d_jobs (device copy of `jobs`, already in place) saves the upload
  std::vector<QM31> eval_at_points(const std::vector<EvalJob>& jobs, const std::vector<QPt>& points, int max_log,
                                   bool split = false, const QM31* d_maps = nullptr, int n_points = 0,
                                   const EvalJob* d_jobs = nullptr, const QM31** device_out = nullptr);

  // pinned host staging (bump allocator, reset per proof): async H2D sources / D2H targets
  void begin_op();
  void set_device();
  void reset_event_log();
  void* pin_alloc(size_t bytes);
  void* stage_upload(const void* host, size_t bytes);           // -> device pointer (arena), async
  // Several small uploads as ONE transfer: between stage_group_begin and stage_group_end, stage_upload places its data
  // in a block reserved up front (same offsets on both sides) and nothing is transferred until the group ends.  What
  // does not fit the reservation is uploaded on its own as before.
  void stage_group_begin(size_t reserve_bytes);
  void stage_group_end();
  // a page-locked block that kernels write their (small) results into directly: no transfer behind the kernel, valid
  // after the next wait.  Unsharded proofs; the pointer is a device pointer as well (gather's entry table already
  // relies on that)
  void* result_block(size_t bytes) { return pin_alloc(bytes ? bytes : 4); }
  const void* stage_download(const void* dev, size_t bytes);    // -> pinned host pointer, valid after sync
  template <class T>
  T* upload_vec(const std::vector<T>& v) {
    return (T*)stage_upload(v.data(), v.size() * sizeof(T));
  }
  void fetch_root_async(DevMerkle& m);   // m.root_pinned valid after the next sync
  uint32_t* bad_flag_ = nullptr;  // two device words: [0] marked when a trace table holds a non-canonical M31 word, [1] trace_lut's
  uint32_t bad_epoch_ = 1;        // the mark of the current unsharded proof (see Context::prove)
  char* pin_base_ = nullptr;
  size_t pin_cap_ = 0, pin_off_ = 0;
  char *grp_pin_ = nullptr, *grp_dev_ = nullptr;   // open upload group (stage_group_begin)
  size_t grp_cap_ = 0, grp_off_ = 0;

  struct Shard {
    bool active = false;
    uint32_t rank = 0, world = 1;
    int g = 0;             // log2(world)
    int fri_min_log = 16;  // FRI layers / quotient columns of at most 2^fri_min_log rows are replicated
    lmn_collective coll{};
    void* rccl = nullptr;  // built-in RCCL transport (RcclTransport in prover.cpp)
  } shard_;
  uint32_t block_rows(int log) const { return 1u << (log - shard_.g); }  // rows of a 2^log column held per rank

  int device_;
  lmn_stream_t stream_{};
  bool owns_stream_ = true;
  // second stream + ordering events: the quotient kernel of the smaller LDE size runs next to the leaf hashing of
  // the larger one in the first FRI layer (LMN_FRI_OVERLAP=1, experiment)
  lmn_stream_t stream2_{};
  lmn_event_t ev_fork_{}, ev_join_{};
  bool have_stream2_ = false;
  int fri_overlap_mode_ = 0;          // 0 never, 1 always, 2 while this is the process's only proof in flight
  bool second_stream_wanted();
  lmn_event_t wait_before_level_ev_{};
  int wait_before_level_ = -1;    // build_merkle_levels: make stream_ wait for wait_before_level_ev_ before this level
  bool merkle_cut_ = false;       // inside prove() of an unsharded proof: trees are stored without their register levels
  Arena arena_;
  int tw_max_log_ = 0;
  // twiddle tables: Y[m] (m>=1), X[k] (k>=2), forward + inverse, device pointers
  std::vector<uint32_t*> twY_, twX_, itwY_, itwX_;
  std::vector<uint32_t*> twY2_, twX2_, itwY2_, itwX2_;  // the same tables, entries doubled (TwPtrs::d)
  std::vector<void*> tw_allocs_;
  std::shared_ptr<void> tw_shared_;   // the device's twiddle set, shared by the contexts of a process (prover.cpp)
  friend struct StageTimer;
};

}  // namespace lmn
