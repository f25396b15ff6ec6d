// Launch wrappers for the gfx950 kernels of the prove hot path.
// All pointers are device pointers; all launches are asynchronous on `s`.
#pragma once
#include "blake2s.h"
#include "field.h"

namespace lmn {

constexpr int MAX_LOG = 30;

// Per-layer twiddle pointers of one canonic circle domain (layer 0 = circle/y layer).  d[i] = the same table with
// every entry doubled (2w < 2^32): the fixed-shape FFT kernels multiply by 2w so that the 64-bit product splits into
// floor(x w / 2^31) (high word) and 2 (x w mod 2^31) (low word) without a funnel shift (fft_fixed.hip).
struct TwPtrs {
  const uint32_t* l[MAX_LOG];
  const uint32_t* d[MAX_LOG];
};

// ---- a3: AoS rows -> padded SoA columns (write_trace / pack_values)
struct PadRow {
  uint32_t v[32];  // the component's padding row (e.g. add/table.rs:40-58)
};
void launch_transpose_pad(const uint32_t* rows, uint64_t n_rows, int ncols, int log_size, uint32_t* cols,
                          const PadRow& pad, uint32_t* bad_flag /* device word, set when a value >= P */, lmn_stream_t s);
// only rows [blk_row0, blk_row0 + blk_rows) of the padded table (blk_row0 a multiple of 64); columns out_stride apart
void launch_transpose_pad_rows(const uint32_t* rows, uint64_t n_rows, int ncols, int log_size, uint32_t* cols,
                               uint64_t out_stride, uint64_t blk_row0, uint64_t blk_rows, const PadRow& pad, uint32_t* bad_flag,
                               lmn_stream_t s,
                               uint32_t bad_value = 1u /* what a non-canonical word writes to *bad_flag */);

// ---- a4: circle FFT passes.  data = ncols columns of 2^log_n words at stride col_stride.
// dst may equal src (in place).  launch_fft zero-extends src (2^log_src words) to 2^log_n (LDE).
// Both return the number of pass kernels launched.
int launch_ifft(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int ncols, int log_n,
                const TwPtrs& itw, lmn_stream_t s);
int launch_fft(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int log_src, int ncols,
               int log_n, const TwPtrs& tw, lmn_stream_t s);
// interpolate + extend by one bit in three launches (the strided passes of both transforms fused): evals (2^log_n) ->
// coeffs (2^log_n, kept) and lde (2^(log_n+1)).  itw: inverse twiddles of domain log_n; tw_ext: twiddles of log_n + 1.
bool fft_interp_extend_supported(int log_n);
int launch_interp_extend(uint32_t* coeffs, uint64_t coeff_stride, const uint32_t* evals, uint64_t evals_stride, uint32_t* lde,
                         uint64_t lde_stride, int ncols, int log_n, const TwPtrs& itw, const TwPtrs& tw_ext, lmn_stream_t s);
// The same from the table's AoS rows (transpose fused into the inverse low pass; launch_transpose_pad's padding and
// canonical-word check included).  false: not applicable to this size / build (transpose first, then launch_interp_extend).
struct PadRow;
bool launch_interp_extend_rows(uint32_t* coeffs, uint64_t coeff_stride, const uint32_t* rows, uint64_t n_rows, const PadRow& pad,
                               uint32_t* bad_flag, uint32_t bad_value, uint32_t* lde, uint64_t lde_stride, int ncols, int log_n,
                               const TwPtrs& itw, const TwPtrs& tw_ext, lmn_stream_t s);
// forward transform restricted to block `block` of 2^log_blocks equal row blocks of the 2^log_n domain (tw = the
// twiddles of the whole domain); dst receives 2^(log_n - log_blocks) words per column
int launch_fft_block(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int log_src, int ncols,
                     int log_n, int log_blocks, uint32_t block, const TwPtrs& tw, lmn_stream_t s);
// Row blocks of `ncols` columns (column stride src_stride, blocks of block_rows rows) packed per destination rank for
// the all-to-all of a sharded commitment: dst[((s * ncols + c) * nsel + h) * block_rows + i] =
// src[c * src_stride + sel.blk[s][h] * block_rows + i] for s < world, h < nsel.
struct PackSel {
  uint32_t blk[8][2];
};
void launch_pack_blocks(const uint32_t* src, uint64_t src_stride, uint32_t* dst, uint32_t block_rows, int ncols, int nsel,
                        int world, const PackSel& sel, lmn_stream_t s);
// the inverse with one block per rank: cols[c * col_stride + r * block_rows + i] = packed[(r * ncols + c) * block_rows + i]
void launch_unpack_blocks(const uint32_t* packed, uint32_t* cols, uint64_t col_stride, uint32_t block_rows, int ncols, int world,
                          lmn_stream_t s);
// single-layer reference kernels (debug / self-test only)
void launch_fft_simple(uint32_t* data, uint64_t col_stride, int ncols, int log_n, const TwPtrs& tw, bool inverse,
                       lmn_stream_t s);
// dst (2^log_dst per column) <- src coefficients (2^log_src per column) zero-extended
void launch_extend(const uint32_t* src, uint64_t src_stride, int log_src, uint32_t* dst, uint64_t dst_stride,
                   int log_dst, int ncols, lmn_stream_t s);

// ---- level-2 column ops (stwo ColumnOps / FriOps pieces that `prove` itself never needs as separate passes)
// in-place bit-reversal permutation of every column (ColumnOps::bit_reverse_column)
void launch_bit_reverse(uint32_t* data, uint64_t col_stride, int ncols, int log_n, lmn_stream_t s);
// FriOps::decompose of a secure column f (4 x 2^log_n): lambda_out[0] = (sum of first half - sum of second half) / 2^log_n;
// g = f - lambda on the first half, f + lambda on the second.  scratch: decompose_num_blocks(log_n) QM31 values.
int decompose_num_blocks(int log_n);
void launch_decompose(const uint32_t* f, int log_n, uint32_t* g, QM31* lambda_out, QM31* scratch, lmn_stream_t s);

// ---- a4: Blake2s Merkle layer.  out[i] = H(prev[2i] || prev[2i+1] || cols[0][i] .. cols[ncols-1][i])
void launch_merkle_layer(const uint32_t* prev, const uint32_t* const* cols, int ncols, uint32_t size, uint32_t* out,
                         lmn_stream_t s);

// ---- device-resident channel (FRI commit loop): digest <- H(digest || root); alpha <- draw_felt()
struct DevChannel {
  uint32_t digest[8];
  uint32_t n_sent;
  uint32_t variant;   // draw encoding: 0 = digest || counter padded to 32 bytes (KAT), 1 = digest || u32 counter || 0x00 (LMN_PV_DRAW_CTR_U32)
};
void launch_chan_mix_root_draw(DevChannel* ch, const uint32_t* root, QM31* out_alpha, uint32_t* root_copy,
                               lmn_stream_t s);

// ---- device-resident Fiat-Shamir of the commitment phases (phase_trace / _logup / _composition / _oods.cpp): the host
// enqueues a whole proof up to the sampled values without waiting; these single-workgroup kernels do the transcript steps
// in between on a DevChannel and leave what the next kernels need in device memory.  The host replays the same steps on
// its own Channel once the values have arrived and cross-checks every draw.
constexpr int CHAN_N_ELEMS = 5;              // relation element sets (prover.h ELEMS_*)
struct DevElems {                            // z / alpha of every set, written by the ChanStep of kind 1 (or uploaded by the host)
  QM31 z[CHAN_N_ELEMS], alpha[CHAN_N_ELEMS];
};
// Everything the host needs back from the device-resident steps, in one place: ONE download at the proof's first wait
struct DevReport {
  DevElems elems;
  QM31 comp_alpha, t;
  QM31 claimed[17];
  uint32_t roots[3][8];                      // main, interaction, composition tree
  uint32_t bad;                              // copy of the transposes' non-canonical-word verdict
  uint32_t pad[3];
};
struct ChanElemSets {
  int n;
  int set[CHAN_N_ELEMS];                     // relation element set of each draw_felts(2); negative: drawn and discarded
};
constexpr int CHAN_MAX_INST = 17;
struct ChanCoeffPlan {
  int n_inst, n_total;
  const QM31* claimed[CHAN_MAX_INST];        // device [claimed, shift] of each component, struct order
  int16_t k0[CHAN_MAX_INST];
  int8_t n_kernel[CHAN_MAX_INST];
  int8_t proto_index[CHAN_MAX_INST][16];
  uint16_t neg[CHAN_MAX_INST];               // bit k: the protocol's constraint is minus kernel slot k
};
constexpr int CHAN_MAX_POINTS = 24;
struct ChanOodsPlan {
  int n_points, n_maps;
  uint32_t step_x[CHAN_MAX_POINTS], step_y[CHAN_MAX_POINTS];
};
// One transcript step of the commitment phases, made by the launch that produces the tree's root (launch_merkle_small)
// or by a launch of its own (launch_chan_step):
//  kind 1  the channel starts at `start` (a launch argument instead of an upload); mix_root(root 1); one draw_felts(2) per
//          entry of `sets` -> rep->elems; rep->bad = *bad_word
//  kind 2  mix_felts([claimed_i]) per component, mix_root(root 2), draw_felt() = the composition randomness alpha; then the
//          coefficient of every kernel constraint slot of every component: sign * alpha^(n_total - 1 - (k0 + proto_index)),
//          0 for a slot the protocol lacks (Context's constraint_layout) - 16 per component at coeff_out + 16 * i
//  kind 3  mix_root(root 3), t = draw_felt(), the OODS point ((1 - t^2) / (1 + t^2), 2t / (1 + t^2)), the points
//          oods + step_i (i >= 1; step_0 unused) and per point the mappings y, x, pi(x), pi^2(x), ... that
//          launch_eval_tables expands -> maps_out (n_points x n_maps); the finished report is copied to rep_host
//          (page-locked memory) by the kernel itself - no download behind it
//  kind 0  none (launch_merkle_small with a channel: mix_root + the draw of a folding alpha, the FRI layers' step)
struct ChanStep {
  int kind;
  DevChannel start;
  const uint32_t* bad_word;
  ChanElemSets sets;
  ChanCoeffPlan coeff;
  QM31* coeff_out;
  ChanOodsPlan oods;
  QM31* maps_out;
  uint32_t* rep_host;
  DevReport* rep;
  // kind 3 also copies `copy_words` words from page-locked memory to device memory (the evaluation kernels' job table)
  const uint32_t* copy_src;
  uint32_t* copy_dst;
  uint32_t copy_words, pad_;
};
// The kernels take the step as a device-visible pointer (+ its kind by value) and fetch it into LDS with one load per lane:
// page-locked host memory serves (one round trip, behind the launch's first loads).  check_chan_step validates a plan.
void check_chan_step(const ChanStep& step);
void launch_chan_step(DevChannel* ch, const ChanStep* step, int step_kind, const uint32_t* root, lmn_stream_t s);

// fused subtree variants: start level (children hashes and/or its own columns) + plain levels above.
// The start level's columns are given as runs of contiguous equal-size columns.
constexpr int MERKLE_MAX_SEG = 4;
// Experiment builds only (-DLMN_ABLATE, tools/build_variants.sh): env LMN_ABLATE is a mask of kernel families whose
// launches are skipped, to read a family's MARGINAL cost under concurrent load off the change in proofs/s (the proofs
// are garbage; the prover's OODS self-check is off in such a build).  1 merkle_fused, 2 transforms, 4 FRI quotients,
// 8 constraint quotients, 16 OODS evaluation, 32 logup; traffic experiments (round 6): 64 the transpose launch, 128 the
// composition tree's leaf loads served from a 16 KB stand-in.
#ifdef LMN_ABLATE
inline unsigned ablate_mask() {
  static const unsigned m = getenv("LMN_ABLATE") ? (unsigned)atoi(getenv("LMN_ABLATE")) : 0u;
  return m;
}
#define LMN_ABLATED(bit) (::lmn::ablate_mask() & (bit))
#else
#define LMN_ABLATED(bit) false
#endif
constexpr int MERKLE_MAX_SUB = 3;     // per-lane register subtree: 2^3 start nodes
constexpr int MERKLE_MAX_FUSED = 11;  // levels above the start level covered by one launch
struct MerkleSegs {
  const uint32_t* base[MERKLE_MAX_SEG];
  int n[MERKLE_MAX_SEG];
};
struct MerkleLevels {
  uint32_t* p[MERKLE_MAX_FUSED + 1];  // p[0] = start level output, p[l] = l levels above it
};
// FRI layers: the leaves of a layer's tree are the fold of the previous layer.  With `fold` given, the start level
// computes leaf i = fold(src[2i], src[2i+1]) itself (FriOps::fold_line / fold_circle_into_line without an
// accumulator), writes it to dst (the layer's 4 coordinate columns, which later steps read) and hashes it - one
// launch and one pass over the layer less than fold kernel + leaf hashing.
struct MerkleFold {
  const uint32_t* src;   // previous layer / quotient column: 4 coordinate columns of 2*size words, stride 2*size
  const uint32_t* itw;   // inverse twiddles of the fold (one per leaf)
  const QM31* alpha;     // device: the folding randomness drawn by the previous layer's channel step
  uint32_t* dst;         // this layer: 4 coordinate columns of `size` words, stride `size`
  // A quotient column of 2*size rows that joins this layer (fold_circle_into_line with accumulation, same alpha):
  // leaf i = fold(src)[i] * alpha^2 + fold(src2)[i]; null if none
  const uint32_t* src2;
  const uint32_t* itw2;
  // Not a fold - a tree whose largest level (2*size leaves of `below_ncols` <= 8 contiguous columns, no children) sits
  // directly under a level with columns of its own: the start level hashes its two leaves itself instead of reading
  // their hashes, so the leaf level is neither written nor read back (2 x 32 B per leaf) and needs no launch.
  const uint32_t* below;
  int below_ncols;
};
void launch_merkle_fused(const uint32_t* prev, const MerkleSegs& sg, int ncols, uint32_t size,
                         const MerkleLevels& outs, int sub, int nfused, lmn_stream_t s, const MerkleFold* fold = nullptr);
// one block, size <= 1024 start nodes, nfused <= 10
// If `ch` is given and this launch reaches the root, it also makes the transcript step that consumes the root: without
// `step`, mix_root and the draw of the next felt (saves a launch per FRI layer); with it, that ChanStep.
void launch_merkle_small(const uint32_t* prev, const MerkleSegs& sg, int ncols, uint32_t size,
                         const MerkleLevels& outs, int nfused, DevChannel* ch, QM31* alpha_out, uint32_t* root_copy,
                         lmn_stream_t s, const ChanStep* step = nullptr, int step_kind = 0);

// FRI tail: layers of log size first_log, first_log-1, ... (n_layers of them, all <= 2^10) committed
// and folded in one single-block launch.  layers[li].next is the evaluation buffer of the next layer.
struct FriTailLayer {
  const uint32_t* vals;        // 4 x 2^L line evaluation (coordinate-major)
  uint32_t* next;              // 4 x 2^(L-1): fold output
  const uint32_t* itw;         // 1/x twiddles of the line domain of log size L
  uint32_t* merkle[11];        // merkle[l]: 2^l hashes, l = 0..L
};
// What the tail does at its two ends besides its layers.
// In front (src != null): its first layer does not exist yet - it is the line fold of `src` (2^(first_log+1) values,
// 1/x twiddles `itw`, folding randomness *alpha), computed by the tail itself (and written to layers[0].vals).
// Behind (out_mirror != null): when the tail is done it copies the FRI loop's result block (`out_words` words at out_block:
// roots | alphas | last layer, written by this and the earlier launches) to page-locked memory - no download behind it
struct FriTailIo {
  const uint32_t* src;
  const uint32_t* itw;
  const QM31* alpha;
  const uint32_t* out_block;
  uint32_t* out_mirror;
  uint32_t out_words;
};
void launch_fri_tail(DevChannel* ch, const FriTailLayer* layers, int n_layers, int first_log, QM31* alphas_out,
                     uint32_t* roots_out, lmn_stream_t s, const FriTailIo* io = nullptr);

// ---- twiddle tables on the device (a11): entry h of a table of 2^bits entries is the y (coord 0) or x (coord 1)
// coordinate of the point init + bit_reverse(h, bits) * step of a half coset; sx / sy = step * 2^k.  Four forms are written:
// the value, its inverse, and both doubled (TwPtrs::d).
struct TwGen {
  uint32_t ix, iy;
  uint32_t sx[30], sy[30];
};
void launch_twiddles(int bits, const TwGen& g, int coord, uint32_t* tw, uint32_t* itw, uint32_t* tw2, uint32_t* itw2,
                     lmn_stream_t s);

// ---- gather: out[dst_off[e] + k] = arena[src_off[e] + k], k < len[e]
struct GatherEntry {
  uint64_t src_off;  // word offset into arena
  uint32_t len;      // words
  uint32_t dst_off;  // word offset into out
};
// A Merkle node that was never written (a fused launch leaves the levels its lanes keep in registers - 7/8 of a big
// tree's hashes, of which a proof reads a few dozen - out of HBM): recomputed for the decommitment from the launch's
// start level.  node = index at `depth` (0..2) levels above that start level; the 8 words go to out[dst_off..].
struct MerkleRecompute {
  const uint32_t* prev;  // start level of the launch that skipped the node (as given to launch_merkle_fused)
  MerkleSegs sg;
  int ncols;
  uint32_t size;
  const uint32_t* below; // MerkleFold::below of that launch (children = hashes of two leaves), or null
  int below_ncols;
  uint32_t node;
  int depth;
  uint32_t dst_off;
};
void launch_gather(const uint32_t* arena, const GatherEntry* entries, uint32_t n_entries, const MerkleRecompute* jobs,
                   uint32_t n_jobs, uint32_t* out, lmn_stream_t s);

// ---- trace generation for the elementwise primitives (the producer of the hot path's input)
struct TraceNode {
  uint32_t node_id, lhs_id, rhs_id;
  uint32_t lhs_mult, rhs_mult, out_mult;  // canonical M31
  // Contiguous in the reference's own row rule (prim.rs:253-296, zip_longest over the input BUFFER and the output):
  // phys_n = elements of the input buffer (0: the view rule), out_n = elements of the output
  uint64_t phys_n = 0, out_n = 0;
};
struct TraceView {  // strided view; ndim == 0: contiguous
  uint32_t ndim;
  uint32_t shape[4];
  int64_t strides[4];
  int64_t offset;
};
void launch_trace_elementwise(int kind, const int32_t* lhs, const TraceView& lv, const int32_t* rhs, const TraceView& rv,
                              uint64_t n, const TraceNode& nd, uint32_t* rows, int32_t* out, uint32_t* aux,
                              lmn_stream_t s);
// The value ranges a LUT enumerates, ascending and disjoint (`LookupLayout::ranges` after `coalesce_ranges`,
// crates/graph/src/graph.rs:665-691): LUT row of value v in range r = base[r] + (v - lo[r])
// (`LookupLayout::find_index`, crates/air/src/preprocessed.rs:60-77).
constexpr int LUT_MAX_RANGES = 16;
struct LutRanges {
  int n;
  int32_t lo[LUT_MAX_RANGES], hi[LUT_MAX_RANGES];
  uint32_t base[LUT_MAX_RANGES];
};
// LUT op rows + LUT multiplicities; *err_flag (device word) is set when an input falls outside every range
void launch_trace_lut(const int32_t* input, const TraceView& view, uint64_t n, const TraceNode& nd,
                      const uint32_t* lut_col1, const LutRanges& ranges, uint32_t* mult, uint32_t* rows,
                      int32_t* out, uint32_t* err_flag, lmn_stream_t s);

void launch_trace_reduce(bool is_max, const int32_t* input, uint64_t front, uint64_t dim, uint64_t back,
                         const TraceNode& nd, uint32_t* rows, int32_t* out, lmn_stream_t s);

// ---- a6: logup
constexpr int LOGUP_MAX_REL = 7;
struct LogupArgs {
  int k;                       // number of relations (1..7)
  const uint32_t* val[LOGUP_MAX_REL];   // value column
  const uint32_t* id[LOGUP_MAX_REL];    // tensor-id column, nullptr for width-1 relations
  const uint32_t* mult[LOGUP_MAX_REL];  // multiplicity column
  int neg[LOGUP_MAX_REL];      // numerator is -mult
  QM31 z[LOGUP_MAX_REL], alpha[LOGUP_MAX_REL];  // element set of each relation (used when d_elems is null)
  const DevElems* d_elems;     // device-resident draws (ChanStep kind 1): relation j uses set es[j]
  int es[LOGUP_MAX_REL];
  // rows != nullptr: the component's evaluations were never stored column-major (the transpose ran inside the
  // interpolation): the relation's cells are read from the table's rows - row r at rows + r * row_words, columns
  // vcol / icol (-1: none) / mcol - and from the padding row's values beyond n_real
  const uint32_t* rows;
  uint32_t row_words, n_real;
  int vcol[LOGUP_MAX_REL], icol[LOGUP_MAX_REL], mcol[LOGUP_MAX_REL];
  uint32_t pad_val[LOGUP_MAX_REL], pad_id[LOGUP_MAX_REL], pad_mult[LOGUP_MAX_REL];
  uint32_t* inter;             // interaction eval columns (4k columns, stride n)
  QM31* last_tmp;              // S_{k-1} per row (AoS), n entries
  uint32_t* partials;          // per-block partial sums, 4 words each
  uint32_t n;                  // rows
};
int logup_num_blocks(uint32_t n);
void launch_logup_fracs(const LogupArgs& a, lmn_stream_t s);
// claimed_out[0] = sum of partials; claimed_out[1] = shift = claimed * n_inv
void launch_logup_reduce(const uint32_t* partials, int nblocks, uint32_t n_inv, QM31* claimed_out, lmn_stream_t s);
// prefix sum of (last_tmp - shift) in coset order, written to the last 4 interaction columns.
// derive_claim: claimed_shift is OUTPUT - the claimed sum (= sum of last_tmp) and shift = claimed * n_inv come out of the
// scan of the block totals itself (no launch_logup_reduce in front); otherwise it is read.
void launch_logup_scan(const QM31* last_tmp, QM31* claimed_shift, int log_size, uint32_t* out_cols /*4 x n*/,
                       QM31* blocksums, lmn_stream_t s, bool derive_claim = false, uint32_t n_inv = 0);
int logup_scan_num_blocks(int log_size);

// ---- a7: constraint quotients on the eval domain (log_size + 1)
struct CompositionArgs {
  int kind;                    // TraceTable kind (LMN_KIND_*)
  int log_size;                // trace log size
  int eval_log;                // eval domain log size
  // Rows [row0, row0 + n_rows) of the eval domain are evaluated (the whole domain, or one rank's block of a
  // sharded proof).  main / inter / pre / pre2 hold exactly those rows, columns `stride` words apart; `out` and
  // `prev_last` span the whole domain (stride 2^eval_log, indexed by the global storage index).
  uint32_t row0, n_rows;
  uint64_t stride;
  const uint32_t* main;        // main columns on the eval domain
  const uint32_t* inter;       // interaction columns on the eval domain
  const uint32_t* prev_last;   // last 4 interaction columns where the mask offset -1 reads them (other row blocks)
  uint32_t* out;               // 4 coordinate columns, stride 2^eval_log
  int accumulate;              // out += instead of out =
  QM31 z, alpha;               // NodeElements
  QM31 z2, alpha2;             // the component's LUT element set (range check / sin / exp2 / log2)
  const DevElems* d_elems;     // when set: the four values above are read from here (sets 0 and es2) instead
  int es2;
  const uint32_t* pre;         // preprocessed columns on the eval domain (lookup components)
  const uint32_t* pre2;
  const QM31* claimed_shift;   // device: [claimed, shift]
  QM31 coeff[16];              // alpha^(N-1-k) for this component's constraints, in order
  const QM31* d_coeff;         // when set: the 16 coefficients are read from device memory (ChanStep kind 2) instead
  uint32_t zinv[2];            // 1/Z for rows with (s >> log_size) == 0 / 1
};
void launch_composition(const CompositionArgs& a, lmn_stream_t s);
// out (4 x n) += in (4 x n)
void launch_secure_add(uint32_t* out, const uint32_t* in, uint64_t n_words, lmn_stream_t s);

// ---- a9: OODS evaluation of coefficient columns
struct EvalJob {
  const uint32_t* coeffs;
  int log_n;
  int point;                   // which point table (0 = oods, 1.. = shifted points)
  int owner = -1;              // sharded proofs: -1 = the coefficient chunks are split over the ranks; r = only rank r
                               // holds the coefficients (column-parallel interpolation) and evaluates all chunks
};
constexpr int EVAL_LB = 10;
// tables: for point p: lo table at lo_tab + p*2^EVAL_LB, hi table at hi_tab + p*hi_stride
// shard_world > 1: only the chunks this rank owns are evaluated (zeros elsewhere): the reduction is a partial sum
void launch_eval_at_point(const EvalJob* jobs, int njobs, const QM31* lo_tab, const QM31* hi_tab, uint32_t hi_stride,
                          int max_log, QM31* partial_out /* njobs x max_chunks */, int max_chunks, lmn_stream_t s,
                          uint32_t shard_rank = 0, uint32_t shard_world = 1);
int eval_num_chunks(int log_n);
void launch_eval_tables(const QM31* maps, int maps_stride, int npoints, QM31* lo_tab, QM31* hi_tab, uint32_t hi_n,
                        int hi_bits, lmn_stream_t s);
void launch_eval_reduce(const EvalJob* jobs, int njobs, const QM31* partial, int max_chunks, QM31* out,
                        lmn_stream_t s);

// ---- a9: FRI quotients
constexpr int QUOT_MAX_BATCH = 4;
constexpr int QUOT_MAX_ENTRIES = 512;
struct QuotEntry {
  const uint32_t* col;         // column of 2^log_size words
  QM31 c;                      // alpha^k * c for this (batch, column) sample
};
// Per sample batch b (the columns sampled at one point p_b): A = sum_k alpha^k a_k, B = sum_k alpha^k b_k of the batch's
// line coefficients, batch_coeff = alpha^|batch|, and the point's coordinates split into their CM31 halves
// (x = prx + u pix, y = pry + u piy).  In device memory: uploaded by the host, or - unsharded proofs - written by
// k_quot_prepare behind the point evaluations, so that no host round trip stands in front of the quotient kernels.
struct QuotDev {
  QM31 A[QUOT_MAX_BATCH], B[QUOT_MAX_BATCH], batch_coeff[QUOT_MAX_BATCH];
  CM31 prx[QUOT_MAX_BATCH], pry[QUOT_MAX_BATCH], pix[QUOT_MAX_BATCH], piy[QUOT_MAX_BATCH];
};
struct QuotientArgs {
  int log_size;                // log size of the whole LDE domain
  // rows [row0, row0 + 2^log_rows) are computed; entries[].col and out hold exactly those rows
  uint32_t row0;
  int log_rows;
  uint64_t out_stride;         // words between the 4 coordinate columns of out
  int nbatch;
  int batch_start[QUOT_MAX_BATCH + 1];  // range into entries
  const QuotEntry* entries;    // device
  const QuotDev* dev;          // device: the batches' line sums, powers and sample points (uploaded, or written by k_quot_prepare)
  const uint32_t* tw_y;        // layer-0 twiddles of the domain (y at storage 2h)
  const uint32_t* tw_x;        // layer-1 twiddles (x at storage 4h)
  uint32_t* out;               // 4 coordinate columns
};
void launch_quotients(const QuotientArgs& a, lmn_stream_t s);

// ---- a9: the transcript step and the tables in front of the FRI quotient kernels, on the device (unsharded proofs).
// Behind the point evaluations one workgroup mixes the sampled values into the device-resident channel
// (Channel::mix_felts: one Blake2s over digest || values, quad-cooperative), draws the quotient randomness alpha, and
// writes what QuotientOps::accumulate_quotients' host side (quotients.cpp make_quotient_args) would have uploaded: per LDE
// size the (column, alpha^k * c_k) entries and the batches' QuotDev.  The host enqueues the quotient kernels and the whole
// FRI commit loop without waiting for the sampled values; it replays this step (and checks the composition identity) when
// the FRI results arrive.
constexpr int QUOT_PREP_MAX_SIZES = 8;
constexpr int QUOT_PREP_MAX_SAMPLES = 496;   // 8 + 4 x 496 message words = 125 Blake2s blocks
struct QuotPrepEntry {
  const uint32_t* col;         // the column's LDE on the device
  uint32_t sample;             // index of its sampled value (sampled_values order)
  uint16_t size, batch;        // LDE size (index into QuotPrepPlan::size) and sample batch inside it
};
struct QuotPrepSize {
  int n_entries, n_batch;
  int batch_start[QUOT_MAX_BATCH + 1];   // into this size's entries
  int point[QUOT_MAX_BATCH];             // sample point of each batch (row of `maps`)
  uint32_t first_entry, pad_;            // the size's first entry in the plan's list
  QuotEntry* entries_out;                // device, n_entries
  QuotDev* dev_out;                      // device
};
struct QuotPrepPlan {
  int n_sizes, n_samples, n_maps, n_entries;
  QuotPrepSize size[QUOT_PREP_MAX_SIZES];
  const QuotPrepEntry* entries;          // page-locked, n_entries (sizes in order, batches in order inside a size)
  const QM31* vals;                      // device: the sampled values (k_eval_reduce)
  QM31* vals_host;                       // page-locked: n_samples values + the drawn alpha, for the host's replay
  const QM31* maps;                      // device: per sample point y, x, pi(x), ... (ChanStep kind 3), n_maps apart
  const uint32_t* copy_src;              // also copies copy_words words from page-locked to device memory (the FRI tail's table)
  uint32_t* copy_dst;
  uint32_t copy_words, pad_;
};
void launch_quot_prepare(DevChannel* ch, const QuotPrepPlan* plan /* device-visible */, lmn_stream_t s);

// ---- a9: FRI folds.  Secure columns are 4 coordinate arrays at stride = length.
// alpha is read from device memory (written by the device-resident channel).
// src holds src_len rows (a whole layer or one rank's block; itw_* then points at the block's first twiddle);
// dst_stride = words between dst's coordinate columns (0: src_len / 2, i.e. dst is exactly the folded rows).
void launch_fold_circle_into_line(uint32_t* dst, const uint32_t* src, uint32_t src_len, const uint32_t* itw_y,
                                  const QM31* alpha, int accumulate, lmn_stream_t s, uint64_t dst_stride = 0);
void launch_fold_line(uint32_t* dst, const uint32_t* src, uint32_t src_len, const uint32_t* itw_x, const QM31* alpha,
                      lmn_stream_t s, uint64_t dst_stride = 0);

}  // namespace lmn
