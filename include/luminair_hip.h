/* luminair_hip.h — C ABI of the MI355X-native LuminAIR prover backend.
 *
 * Drop-in boundary for the reference's
 *     pub fn prove(pie: LuminairPie, settings: CircuitSettings)
 *         -> Result<LuminairProof<Blake2sMerkleHasher>, LuminairError>
 * (/root/reference/crates/prover/src/prover.rs:28-31).  A Rust shim flattens
 * `pie.trace_tables` (crates/air/src/pie.rs:31-66) into `lmn_table`s and deserialises the returned
 * bincode bytes with `LuminairProof::from_bincode` (crates/prover/src/lib.rs:36-43).
 *
 * Plain pointers and sizes only; no C++/torch types.  One context per GPU (or several per GPU to keep
 * proofs in flight).  Different contexts are independent and run concurrently.  Calls on ONE context
 * from several threads are serialised by an internal lock (a context owns one HIP stream, one device
 * arena and one pinned staging buffer): safe, but they do not overlap - use one context per thread.
 * Device buffers handed to a context (LMN_TABLE_ROWS_ON_DEVICE, lmn_col handles) must not be written
 * by another context while a call that reads them is running.
 */
#ifndef LUMINAIR_HIP_H
#define LUMINAIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header.  It is bumped whenever a struct crossing the boundary changes size or layout (4: lmn_view
 * gained `offset`, 56 -> 64 bytes; 6: `protocol_variant` became a set of LMN_PV_* bits, LMN_VARIANT_PINNED 1 -> 0x1f).  A caller compares LMN_API_VERSION (what it was compiled against) with
 * lmn_abi_version() (what the loaded library implements) once at start-up; luminair_amd/backend.py does.
 * Structs passed in (lmn_view, lmn_node_info, lmn_settings, lmn_config) must be zero-initialised before their fields
 * are set, so that fields added by a later version read as 0. */
#define LMN_API_VERSION 6
uint32_t lmn_abi_version(void);

/* Error codes mirror LuminairError (/root/reference/crates/utils/src/lib.rs:5-34). */
#define LMN_OK 0
#define LMN_ERR_EMPTY_TRACE (-1)          /* TraceError::EmptyTrace (add/witness.rs:39-41)      */
#define LMN_ERR_MAIN_TRACE (-2)           /* MainTraceEvalGenError                                */
#define LMN_ERR_INTERACTION_TRACE (-3)    /* InteractionTraceEvalGenError                         */
#define LMN_ERR_CONSTRAINTS (-4)          /* ProverError(ConstraintsNotSatisfied)                 */
#define LMN_ERR_SERIALIZATION (-5)        /* SerializationError                                   */
#define LMN_ERR_INVALID_ARGUMENT (-6)     /* malformed table / unsupported component / config     */
#define LMN_ERR_OUT_OF_MEMORY (-7)
#define LMN_ERR_NO_DEVICE (-8)            /* no HIP device: the library never falls back to CPU   */
#define LMN_ERR_VERIFICATION (-9)         /* StwoVerifierError (verify only)                      */
#define LMN_ERR_INVALID_LOGUP (-10)       /* InvalidLogUp: claimed sums do not cancel (verify)    */
#define LMN_ERR_INTERNAL (-100)

/* TraceTable kinds, in `enum TraceTable` order (crates/air/src/pie.rs:31-66). */
#define LMN_KIND_ADD 0
#define LMN_KIND_MUL 1
#define LMN_KIND_RECIP 2
#define LMN_KIND_SIN 3
#define LMN_KIND_SIN_LOOKUP 4
#define LMN_KIND_SUM_REDUCE 5
#define LMN_KIND_MAX_REDUCE 6
#define LMN_KIND_SQRT 7
#define LMN_KIND_REM 8
#define LMN_KIND_EXP2 9
#define LMN_KIND_EXP2_LOOKUP 10
#define LMN_KIND_LOG2 11
#define LMN_KIND_LOG2_LOOKUP 12
#define LMN_KIND_LESS_THAN 13
#define LMN_KIND_RANGE_CHECK_LOOKUP 14
#define LMN_KIND_INPUTS 15
#define LMN_KIND_CONTIGUOUS 16

/* Protocol flags (SURVEY.md §8c "known deltas KAT-era -> pinned rev"; Appendix A.3).  The reference's only known-answer
 * proof (ui/demo/public/proof) was made by an older LuminAIR / stwo than the sources at HEAD; every observable
 * difference between the two is ONE independent bit of `lmn_config.protocol_variant` (since ABI version 6; before, the
 * field was an enum 0 / 1), so that a proof made by any build of the reference can be pinned by search
 * (tools/pin_variant.py tries every combination through lmn_verify_diagnose).  All bits clear = the KAT protocol, every
 * byte of which is pinned.  LMN_VARIANT_PINNED = what the reference at HEAD is believed to run; its transcript bits are
 * from memory of the un-vendored stwo ("parity unpinned").
 *
 * Transcript bits:
 *   CLAIM17        LuminairClaim / LuminairInteractionClaim have 17 Option slots (crates/air/src/lib.rs:30-48) instead of 8
 *   LUT_DRAWS4     LookupElements::draw draws sin, exp2, log2, range_check (lookups/mod.rs:44-51) instead of one LUT relation
 *   MIX_U64_HASHED mix_u64(v) = blake2s(digest || lo32 LE || hi32 LE) instead of the bare compression function on [lo, hi, 0..]
 *   DRAW_CTR_U32   draw = blake2s(digest || counter u32 LE || 0x00) instead of digest || counter zero-padded to 32 bytes
 *   POW_PREFIXED   proof of work = trailing zeros of blake2s(blake2s(0x12345678 LE || 12 zero bytes || digest || pow_bits LE)
 *                  || nonce u64 LE), nonce then mixed with mix_u64 - instead of the trailing zeros of the digest after
 *                  mix_u64(nonce)
 * Constraint-form bits (numerair's `eval_fixed_*` helpers are un-vendored; the KAT pins eval_fixed_add and the first
 * eval_fixed_mul constraint, and that eval_fixed_mul occupies two constraint slots):
 *   MUL_ONE_SLOT   eval_fixed_mul emits one constraint (KAT: two, the second contributing zero whenever rem == 0)
 *   RECIP_/SQRT_/REM_TWO_SLOTS  the helper emits a second (zero) slot as KAT-era eval_fixed_mul does (default: one)
 *   RECIP_/SQRT_/REM_NEG        the helper's constraint has the opposite sign: input*out + rem - scale^2,
 *                  out^2 + rem - input*scale, rhs*quotient + rem - lhs (default: scale^2 - (input*out + rem), ...) */
#define LMN_PV_CLAIM17 0x0001u
#define LMN_PV_LUT_DRAWS4 0x0002u
#define LMN_PV_MIX_U64_HASHED 0x0004u
#define LMN_PV_DRAW_CTR_U32 0x0008u
#define LMN_PV_POW_PREFIXED 0x0010u
#define LMN_PV_MUL_ONE_SLOT 0x0100u
#define LMN_PV_RECIP_TWO_SLOTS 0x0200u
#define LMN_PV_RECIP_NEG 0x0400u
#define LMN_PV_SQRT_TWO_SLOTS 0x0800u
#define LMN_PV_SQRT_NEG 0x1000u
#define LMN_PV_REM_TWO_SLOTS 0x2000u
#define LMN_PV_REM_NEG 0x4000u
#define LMN_PV_TRANSCRIPT_MASK 0x001fu
#define LMN_PV_FORMS_MASK 0x7f00u
#define LMN_PV_ALL (LMN_PV_TRANSCRIPT_MASK | LMN_PV_FORMS_MASK)
#define LMN_VARIANT_KAT 0u
#define LMN_VARIANT_PINNED LMN_PV_TRANSCRIPT_MASK

/* Replaces PcsConfig::default() (prover.rs:36) + DEFAULT_FP_SCALE (crates/air/src/lib.rs:23-24). */
typedef struct lmn_config {
  uint32_t pow_bits;         /* default 5 */
  uint32_t log_blowup;       /* default 1 (= blow-up 2, PcsConfig::default()); 1..3 accepted, sharded proofs: 1 only */
  uint32_t log_last_layer;   /* default 0 */
  uint32_t n_queries;        /* default 3 */
  uint32_t fp_scale;         /* default 12 */
  uint32_t protocol_variant; /* OR of LMN_PV_* bits: LMN_VARIANT_KAT (0), LMN_VARIANT_PINNED, or any combination */
} lmn_config;

#define LMN_TABLE_ROWS_ON_DEVICE 1u

/* One trace table of a LuminairPie: AoS rows, one u32 (canonical M31) per column in the
 * component's `Column::index()` order (e.g. crates/air/src/components/add/table.rs:191-211). */
typedef struct lmn_table {
  uint32_t kind;        /* LMN_KIND_* */
  uint32_t flags;       /* LMN_TABLE_ROWS_ON_DEVICE: `rows` is a device pointer (lmn_upload) */
  uint64_t n_rows;
  const uint32_t* rows; /* n_rows * n_columns(kind) words */
} lmn_table;

/* CircuitSettings (crates/air/src/settings.rs).  `has_lookups` is a bit per `Lookups` field in
 * draw order (crates/air/src/components/lookups/mod.rs:20-51).  The 8-bit range-check LUT
 * (`lookups.range_check`, crates/graph/src/graph.rs:142-146) is generated by the library (row r holds
 * r, crates/air/src/preprocessed.rs:289-296).  The sin / exp2 / log2 LUT columns come from f64 math on
 * the reference's host side (`SinPreProcessed::gen_column`, preprocessed.rs:351-383 and siblings), so
 * the caller hands them over as data: `luts[i]` carries the two preprocessed columns
 * (`<name>_lut_0` = inputs, `<name>_lut_1` = outputs), 2^log_size canonical M31 words each, exactly
 * as `lookups_to_preprocessed_column` + `gen_column_simd` produce them.  A LUT is required iff its
 * lookup table (LMN_KIND_*_LOOKUP) is in the pie; the lookup table must have 2^log_size rows. */
#define LMN_LOOKUP_SIN 1u
#define LMN_LOOKUP_EXP2 2u
#define LMN_LOOKUP_LOG2 4u
#define LMN_LOOKUP_RANGE_CHECK 8u
#define LMN_LUT_SIN 0u
#define LMN_LUT_EXP2 1u
#define LMN_LUT_LOG2 2u
typedef struct lmn_lut {
  uint32_t kind;        /* LMN_LUT_* */
  uint32_t log_size;
  const uint32_t* col0; /* host pointers, borrowed for the duration of the call */
  const uint32_t* col1;
} lmn_lut;
typedef struct lmn_settings {
  uint32_t has_lookups; /* OR of LMN_LOOKUP_* (informational; checked against the pie's tables) */
  uint32_t n_luts;
  const lmn_lut* luts;
} lmn_settings;

/* `LookupLayout` (crates/air/src/preprocessed.rs:34-46): inclusive ranges of Fixed<12> values (numerair's
 * `Fixed` is a newtype over i64) and the LUT's log size.  lmn_lut_log_size restates `LookupLayout::new` /
 * `calculate_log_size` (crates/air/src/utils.rs:22-27: the value count rounded up to a power of two, at least 16
 * rows).  lmn_lut_from_ranges restates `SinPreProcessed::gen_column` (preprocessed.rs:351-383; Exp2 :434-466, Log2
 * :517-549): all values of the ranges, sorted and de-duplicated, go to col0 as M31 words and f(value / 2^12) * 2^12
 * rounded to the nearest integer (ties away from zero, `f64::round`; numerair's `Fixed::from_f64` is un-vendored:
 * unpinned) to col1; the remaining rows are zero.  Host code (f64 libm, like the reference); needs no context.
 * col0_out / col1_out: 2^log_size words each.  Returns LMN_ERR_INVALID_ARGUMENT for an empty / inverted range, a
 * value outside (-2^30, 2^30), a log2 argument <= 0, or a log_size too small for the values. */
typedef struct lmn_range {
  int64_t lo, hi; /* Range(Fixed, Fixed), inclusive */
} lmn_range;
int lmn_lut_log_size(const lmn_range* ranges, uint32_t n_ranges, uint32_t* log_size_out);
int lmn_lut_from_ranges(uint32_t lut_kind /* LMN_LUT_* */, const lmn_range* ranges, uint32_t n_ranges, uint32_t log_size,
                        uint32_t* col0_out, uint32_t* col1_out);
/* The same with the f64 -> Fixed rounding stated (since ABI version 6; `Fixed::from_f64` is in the un-vendored numerair,
 * so the rule is one of the axes tools/pin_variant.py searches when it is given the reference's preprocessed root):
 * HALF_AWAY = `f64::round` (what lmn_lut_from_ranges uses), HALF_EVEN = round-half-to-even, TRUNC = `as i64` (towards
 * zero), FLOOR = `f64::floor`. */
#define LMN_ROUND_HALF_AWAY 0u
#define LMN_ROUND_HALF_EVEN 1u
#define LMN_ROUND_TRUNC 2u
#define LMN_ROUND_FLOOR 3u
int lmn_lut_from_ranges_r(uint32_t lut_kind, const lmn_range* ranges, uint32_t n_ranges, uint32_t log_size, uint32_t rounding,
                          uint32_t* col0_out, uint32_t* col1_out);

typedef struct lmn_ctx lmn_ctx;

/* Per-stage timings of the last lmn_prove call, milliseconds (HIP events on the prover stream). */
typedef struct lmn_timings {
  float total_ms;
  float transpose_ms, main_commit_ms, logup_ms, interaction_commit_ms, composition_ms, composition_commit_ms,
      oods_ms, quotients_ms, fri_ms, decommit_ms;
  float fft_ms;     /* all circle (i)FFT passes */
  float merkle_ms;  /* all Blake2s Merkle layer kernels */
  uint64_t fft_bytes;    /* algorithmic bytes moved by those FFT launches (8 B per element per transform) */
  uint64_t merkle_bytes; /* algorithmic bytes of the Merkle launches */
  uint32_t fft_launches, merkle_launches;
  uint64_t fft_butterflies;      /* M31 butterflies (1 mul + 1 add + 1 sub) executed by those FFT launches */
  uint64_t merkle_compressions;  /* Blake2s compression-function calls executed by the Merkle launches */
  /* the k_merkle_fused launches alone (the dominant kernel; excludes k_merkle_small / k_fri_tail) */
  float merkle_fused_ms;
  uint32_t merkle_fused_launches;
  uint64_t merkle_fused_bytes, merkle_fused_compressions;
  /* sharded proofs (since ABI version 5): bytes this rank RECEIVED from other ranks through all_to_all / all_gather */
  uint64_t shard_a2a_bytes, shard_gather_bytes;
  uint32_t shard_a2a_calls, shard_gather_calls;
} lmn_timings;

const char* lmn_strerror(int code);
const char* lmn_last_error(const lmn_ctx* ctx);
void lmn_default_config(lmn_config* cfg);
uint32_t lmn_kind_columns(uint32_t kind); /* 0 if the kind is not supported */
/* The rest of a component's boundary data, for bindings and for tests/test_reference_layout.py: the number of logup
 * relations (`TraceColumn::count().1`, e.g. add/table.rs:214-216: lmn_kind_relations, declared with level 2 below) and the padding row `write_trace` appends up to
 * the next power of two (`*TraceTableRow::padding()`, e.g. add/table.rs:40-58; less_than/table.rs:47-72):
 * out receives lmn_kind_columns(kind) words. */
int lmn_kind_padding_row(uint32_t kind, uint32_t* out);

int lmn_ctx_create(int device, const lmn_config* cfg, lmn_ctx** out);
void lmn_ctx_destroy(lmn_ctx* ctx);

/* Replaces prove(pie, settings).  On success *proof_bincode holds `LuminairProof::to_bincode()`
 * bytes owned by the library until lmn_free. */
int lmn_prove(lmn_ctx* ctx, const lmn_table* tables, size_t n_tables, const lmn_settings* settings,
              uint8_t** proof_bincode, size_t* proof_len);
void lmn_free(void* p);
/* Asynchronous form: the reference's callers are single-threaded (prover.rs:28 is a plain function call), and one
 * MI355X wants about eight proofs in flight.  lmn_prove_submit hands the proof to the context's
 * own worker thread and returns; lmn_prove_wait blocks until it is done and returns what lmn_prove would have.  One
 * outstanding submission per context; `tables`, the row buffers and `settings` are borrowed until lmn_prove_wait
 * returns.  A caller keeps N proofs in flight from one thread with N contexts: submit on each, then wait on each. */
int lmn_prove_submit(lmn_ctx* ctx, const lmn_table* tables, size_t n_tables, const lmn_settings* settings);
int lmn_prove_wait(lmn_ctx* ctx, uint8_t** proof_bincode, size_t* proof_len);
/* Per-stage / per-kernel HIP-event timing is off by default (every event record costs a few
 * microseconds between dependent kernels); enable it for the proofs whose lmn_timings you want. */
int lmn_set_profiling(lmn_ctx* ctx, int enabled);
int lmn_get_timings(const lmn_ctx* ctx, lmn_timings* out);

/* Replaces `verify(proof, settings)` (/root/reference/crates/verifiers/rust/src/verifier.rs:21-143).
 * Host-only (the reference verifier is CPU code too); needs no context and no GPU.  Returns LMN_OK,
 * LMN_ERR_INVALID_LOGUP, LMN_ERR_VERIFICATION or LMN_ERR_SERIALIZATION; the message of the last
 * failure on this thread is available through lmn_last_error(NULL). */
int lmn_verify(const uint8_t* proof_bincode, size_t proof_len, const lmn_settings* settings, uint32_t protocol_variant /* LMN_PV_* bits */);
/* lmn_verify checks the proof against PcsConfig::default() (pow 5, blow-up 2, 3 queries, last layer degree 0), as
 * verifier.rs:36 does; a deployment with other parameters states them here.  The security parameters are the
 * verifier's: a proof whose embedded config differs from `expected` is rejected (LMN_ERR_VERIFICATION).  Field
 * elements that are not canonical M31 words, Option tags other than 0/1 and claim log sizes above 26 are
 * LMN_ERR_SERIALIZATION.  `settings` may be NULL; when given, `has_lookups` (if non-zero) and every LUT's
 * log_size must agree with the lookup components in the proof's claim (the LUT columns themselves are not read:
 * like the reference, the verifier takes the preprocessed root from the proof). */
int lmn_verify_with_config(const uint8_t* proof_bincode, size_t proof_len, const lmn_settings* settings,
                           const lmn_config* expected);

/* lmn_verify that does not stop at the first failed check (since ABI version 6): the replay goes on as far as the
 * proof's shape allows and reports which checks passed.  The checks depend on different parts of the protocol, which
 * is what makes a proof from an unknown build of the reference diagnosable (tools/pin_variant.py runs this for every
 * combination of LMN_PV_* bits):
 *   PARSE          the bytes deserialize under the claim layout (CLAIM17) - LMN_ERR_SERIALIZATION otherwise
 *   SHAPE          tree / sampled-value / FRI-layer counts are what the claim implies
 *   LOGUP_SUM      the claimed sums cancel (no transcript dependence)
 *   OODS           composition identity at the OODS point: transcript up to the OODS draw (claim mix, relation draws,
 *                  composition randomness) AND every constraint form of the components present
 *   POW            proof-of-work nonce: the whole transcript up to the nonce; pow_bits bits of evidence only
 *   TREE_DECOMMIT  the four trace trees' decommitments at the drawn query positions: the whole transcript including
 *                  the query draw, and the Merkle hashing; independent of the constraint forms
 *   FRI_DECOMMIT   the FRI layers' decommitments (same dependence)
 *   FRI_FOLDS      FRI quotients from sampled + queried values fold to the last layer: OODS point, quotient and fold
 *                  randomness; independent of the constraint forms
 * `steps` records the channel digest after every mix of the replay, in order (a maintainer prints the same digests from
 * the Rust prover to find the first diverging step).  Returns LMN_OK when the replay ran to the end (whatever the checks
 * said), else the error that stopped it (its text in `first_failure` if no check had failed before).  `report` is
 * always filled as far as the replay got. */
#define LMN_CHECK_PARSE 0x0001u
#define LMN_CHECK_SHAPE 0x0002u
#define LMN_CHECK_LOGUP_SUM 0x0004u
#define LMN_CHECK_OODS 0x0008u
#define LMN_CHECK_POW 0x0010u
#define LMN_CHECK_TREE_DECOMMIT 0x0020u
#define LMN_CHECK_FRI_DECOMMIT 0x0040u
#define LMN_CHECK_FRI_FOLDS 0x0080u
#define LMN_CHECK_ALL 0x00ffu
#define LMN_STEP_ROOT_PREPROCESSED 0u /* mix_root(commitments[0]), prover.rs:59 */
#define LMN_STEP_CLAIM 1u             /* LuminairClaim::mix_into, crates/air/src/lib.rs:52-104 */
#define LMN_STEP_ROOT_MAIN 2u         /* prover.rs:179 */
#define LMN_STEP_INTERACTION_CLAIM 3u /* mix_felts(claimed_sum) per component, components/mod.rs:209-211 */
#define LMN_STEP_ROOT_INTERACTION 4u  /* prover.rs:298 */
#define LMN_STEP_ROOT_COMPOSITION 5u  /* inside stwo::prover::prove, prover.rs:312 */
#define LMN_STEP_SAMPLED_VALUES 6u
#define LMN_STEP_FRI_FIRST_LAYER 7u
#define LMN_STEP_FRI_INNER_LAYER 8u   /* index = layer number */
#define LMN_STEP_FRI_LAST_LAYER 9u
#define LMN_STEP_POW_NONCE 10u
#define LMN_MAX_TRANSCRIPT_STEPS 48
typedef struct lmn_transcript_step {
  uint32_t step;  /* LMN_STEP_* */
  uint32_t index;
  uint8_t digest[32];
} lmn_transcript_step;
typedef struct lmn_verify_report {
  uint32_t checks_run;    /* LMN_CHECK_* bits of the checks the replay reached */
  uint32_t checks_passed; /* ... that held */
  uint32_t checks_failed; /* ... that did not (a check with several parts fails if any part does) */
  uint32_t n_steps;
  lmn_transcript_step steps[LMN_MAX_TRANSCRIPT_STEPS];
  char first_failure[128]; /* text of the first failed check or of the error that stopped the replay */
} lmn_verify_report;
int lmn_verify_diagnose(const uint8_t* proof_bincode, size_t proof_len, const lmn_settings* settings,
                        const lmn_config* expected, lmn_verify_report* report);

/* Page-locked host memory for tables handed over as HOST buffers (`lmn_table.rows` without LMN_TABLE_ROWS_ON_DEVICE -
 * the reference's own calling convention: `prove(pie, settings)` takes the pie by value in host memory,
 * /root/reference/crates/prover/src/prover.rs:28-31,70).  Rows in ordinary pageable memory are copied through the
 * runtime's staging buffers (≈ 20 - 28 GB/s, the calling thread blocked meanwhile); rows in page-locked memory are
 * fetched by the GPU's DMA engines directly at PCIe rate and the call returns at once.  The binding's `flatten`
 *  can write the pie's rows straight into an `lmn_host_alloc` buffer, or page-lock the vectors it
 * already owns with `lmn_host_register`.  No context needed; the memory is usable by every context of the process. */
int lmn_host_alloc(size_t bytes, void** host_out);
void lmn_host_free(void* host);
int lmn_host_register(void* host, size_t bytes);   /* page-lock [host, host + bytes) in place */
int lmn_host_unregister(void* host);

/* Device residency helpers for callers that keep trace tables in HBM. */
int lmn_upload(lmn_ctx* ctx, const void* host, size_t bytes, void** device_out);
void lmn_device_free(lmn_ctx* ctx, void* device_ptr);

int lmn_device_alloc(lmn_ctx* ctx, size_t bytes, void** device_out);
int lmn_download(lmn_ctx* ctx, const void* device, void* host, size_t bytes);
int lmn_upload_to(lmn_ctx* ctx, const void* host, size_t bytes, void* device_dst); /* into an existing allocation */
/* device-to-device copy enqueued on the context's stream (ordered with every other call on this context; returns
 * without waiting) */
int lmn_device_copy(lmn_ctx* ctx, void* device_dst, const void* device_src, size_t bytes);

/* ---- The step before the path: `LuminairOperator::process_trace` of the
 * elementwise primitives on device-resident tensors (crates/graph/src/op/prim.rs:967-1013 Add,
 * :1090-1139 Mul, :388-431 Recip): executes the fixed-point op and fills the node's trace-table rows in
 * HBM, ready for lmn_prove with LMN_TABLE_ROWS_ON_DEVICE.  Tensors are contiguous int32 arrays of
 * Fixed<12> values; Mul: out = floor(lhs*rhs / 4096), rem = lhs*rhs - out*4096; Recip (input > 0):
 * out = floor(4096^2 / input), rem = 4096^2 - input*out (numerair's sign conventions are unpinned,
 * DESIGN.md §2).  out_mult = is_final_output ? 0 : num_consumers; input_mults are -1 at HEAD
 * (prim.rs:1005-1006), 0 for graph initializers in the KAT era. */
typedef struct lmn_node_info {
  uint32_t node_id;
  uint32_t input_ids[2];
  uint32_t num_consumers;
  uint32_t is_final_output;
  int32_t input_mults[2];
} lmn_node_info;
/* kind: LMN_KIND_ADD | LMN_KIND_MUL | LMN_KIND_REM (prim.rs:1323-1421, operands > 0) | unary (rhs_dev ignored):
 * LMN_KIND_RECIP | LMN_KIND_SQRT (prim.rs:573-660) | LMN_KIND_CONTIGUOUS (prim.rs:229-301) | LMN_KIND_INPUTS
 * (`CopyToStwo`, prim.rs:52-88: rows of a graph input, multiplicity = num_consumers).  rows_dev receives
 * n * lmn_kind_columns(kind) words starting at row `row_offset` of the node's table (several nodes of one
 * kind share a table: prim.rs appends); out_dev (may be NULL) receives the n output values. */
int lmn_trace_elementwise(lmn_ctx* ctx, uint32_t kind, const int32_t* lhs_dev, const int32_t* rhs_dev, uint64_t n,
                          const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);

/* Strided view of a device tensor, the part of luminal's ShapeTracker the elementwise ops need: element
 * (i_0 .. i_{ndim-1}) of the op's (row-major) output shape reads tensor[offset + sum_k i_k * strides[k]]; an
 * expanded ("fake") dimension has stride 0, a slice a non-zero offset.  NULL = contiguous. */
typedef struct lmn_view {
  uint32_t ndim; /* 1..4 */
  uint32_t shape[4];
  int64_t strides[4]; /* in elements */
  int64_t offset;     /* in elements: first element of the view (slices) */
} lmn_view;
/* lmn_trace_elementwise with a view per operand (crates/graph/src/op/prim.rs `get_index` with the
 * ShapeTracker's index expression); the product of view shapes must equal n. */
int lmn_trace_elementwise_v(lmn_ctx* ctx, uint32_t kind, const int32_t* lhs_dev, const lmn_view* lhs_view,
                            const int32_t* rhs_dev, const lmn_view* rhs_view, uint64_t n, const lmn_node_info* info,
                            uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);

/* `LuminairContiguous::process_trace` in the reference's own row rule (crates/graph/src/op/prim.rs:229-301): the
 * reference zips the input BUFFER with the output, so row idx carries the idx-th buffer element (consumed with
 * multiplicity -1; zero past the buffer's end) and the idx-th output element (the view's element; past the output's
 * end the index wraps - luminal's own out-of-range index expression is un-vendored, unpinned), max(in_size, out_size)
 * rows, is_last_idx on the buffer's last element.  This is the rule under which a slice or permutation of the input
 * balances its logup (crates/graph/src/tests/ops.rs test_contiguous).  `view` NULL = identity.  For a view with
 * expanded dimensions (out_size > in_size) the reference's rule cannot balance; use lmn_trace_elementwise_v
 * (LMN_KIND_CONTIGUOUS), which consumes the view's element per output row. */
int lmn_trace_contiguous(lmn_ctx* ctx, const int32_t* input_dev, uint64_t in_size, const lmn_view* view, uint64_t out_size,
                         const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);

/* `process_trace` of the LUT ops Sin / Exp2 / Log2 (prim.rs:663-760 and siblings; rows per
 * crates/air/src/components/sin/table.rs): out = LUT[input] for a LUT that enumerates the single value range
 * lo..lo+lut_len-1 (`lut_col1_dev` = its preprocessed output column, M31 words), one lookup per row
 * (lookup multiplicity 1), and the LUT's multiplicity column `mult_dev[input - lo]` incremented per use -
 * `mult_dev` (2^k words, zero-initialised by the caller) IS the SinLookup / Exp2Lookup / Log2Lookup trace table.
 * kind: LMN_KIND_SIN | LMN_KIND_EXP2 | LMN_KIND_LOG2.  Inputs outside the range fail with LMN_ERR_INVALID_ARGUMENT. */
int lmn_trace_lut(lmn_ctx* ctx, uint32_t kind, const int32_t* input_dev, const lmn_view* view, uint64_t n,
                  const lmn_node_info* info, const uint32_t* lut_col1_dev, int32_t lo, uint32_t lut_len,
                  uint32_t* mult_dev, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);
/* The same for a LUT that enumerates SEVERAL value ranges - what `gen_circuit_settings` produces when a graph applies
 * the function on disjoint input ranges (one padded range per op, `coalesce_ranges`, crates/graph/src/graph.rs:61-159,
 * 665-691): `ranges` ascending and disjoint (at most 16), LUT row of a value = `LookupLayout::find_index`
 * (crates/air/src/preprocessed.rs:60-77); `lut_col1_dev` / `mult_dev` are the columns lmn_lut_from_ranges lays out
 * for the same ranges. */
int lmn_trace_lut_ranges(lmn_ctx* ctx, uint32_t kind, const int32_t* input_dev, const lmn_view* view, uint64_t n,
                         const lmn_node_info* info, const uint32_t* lut_col1_dev, const lmn_range* ranges, uint32_t n_ranges,
                         uint32_t* mult_dev, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);

/* `LuminairLessThan::process_trace` (prim.rs:1203-1295): out = 1.0 iff lhs < rhs, diff = rhs - lhs (+ P with
 * borrow) split into four 8-bit limbs; `range_check_mult_dev` (256 words, zero-initialised by the caller) is the
 * RangeCheckLookup trace table and is incremented once per limb. */
int lmn_trace_less_than(lmn_ctx* ctx, const int32_t* lhs_dev, const lmn_view* lhs_view, const int32_t* rhs_dev,
                        const lmn_view* rhs_view, uint64_t n, const lmn_node_info* info, uint32_t* range_check_mult_dev,
                        uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);
/* `LuminairMaxReduce::process_trace` (prim.rs:1591-1734): as lmn_trace_sum_reduce with the running maximum. */
int lmn_trace_max_reduce(lmn_ctx* ctx, const int32_t* input_dev, uint64_t front, uint64_t dim, uint64_t back,
                         const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);

/* `LuminairSumReduce::process_trace` (prim.rs:1450-1565) on a contiguous device tensor of shape
 * (front, dim, back), reducing `dim`: front*back*dim rows in (i, j, k) order with the running sums
 * acc / next_acc; out_dev (may be NULL) receives the front*back sums. */
int lmn_trace_sum_reduce(lmn_ctx* ctx, const int32_t* input_dev, uint64_t front, uint64_t dim, uint64_t back,
                         const lmn_node_info* info, uint32_t* rows_dev, uint64_t row_offset, int32_t* out_dev);

/* ---- Single-proof sharding over the GPUs of one node (BASELINE.json configs 4 and 5).
 * One context per GPU, world = 1, 2, 4 or 8 ranks.  Every rank calls lmn_prove with the SAME tables and gets the
 * SAME proof bytes as an unsharded context would produce.  What is split: every rank evaluates only its aligned
 * block of rows of each LDE (the commitment `tree_builder.commit`, /root/reference/crates/prover/src/prover.rs:179,
 * 298, and stwo's commit of the composition polynomial behind prover.rs:312), hashes the Merkle subtree over that
 * block, evaluates constraint quotients, FRI quotients and FRI pair-folds on its rows.  What is exchanged, all
 * through ONE primitive, an in-place all-gather on device memory: the world subtree roots of every committed tree
 * (32 B per rank), the composition evaluations (16 B per eval-domain row in total) before their interpolation,
 * FRI layers / quotient columns of at most 2^fri_min_log rows (finished on every rank), and the few KB of
 * decommitted values.  The trace-sized transposes / interpolations / logup columns are recomputed on every rank:
 * cheaper than moving the coefficients over xGMI.
 *
 * `all_gather` must be stream-ordered on `stream` (a hipStream_t): `buf_dev` holds world * bytes_per_rank bytes,
 * this rank's part is already at offset rank * bytes_per_rank, and on completion every part is filled in (the
 * in-place form of ncclAllGather).  Return 0 on success. */
typedef struct lmn_collective {
  void* user;
  int (*all_gather)(void* user, void* buf_dev, size_t bytes_per_rank, void* stream);
  /* optional (may be NULL): bracket a batch of all_gather calls that may be fused into one launch
   * (ncclGroupStart / ncclGroupEnd); the 4 coordinate columns of a secure column are gathered as one batch */
  int (*group_begin)(void* user);
  int (*group_end)(void* user);
  /* optional (may be NULL; since ABI version 5): the second exchange primitive, SURVEY.md section 8(e) stage B.  With it
   * every rank interpolates and extends only 1/world of the columns of a commitment (stage A) and the extended columns
   * are re-partitioned into row blocks by ONE stream-ordered all-to-all: this rank sends send_bytes[p] bytes at
   * send_dev + send_off[p] to every rank p and receives recv_bytes[p] bytes from p at recv_dev + recv_off[p]
   * (arrays of `world` entries, p = rank included: a local copy).  Grouped ncclSend / ncclRecv in the built-in
   * transport.  Without it the columns are interpolated on every rank (round 2-3 behaviour). */
  int (*all_to_all)(void* user, const void* send_dev, const size_t* send_off, const size_t* send_bytes, void* recv_dev,
                    const size_t* recv_off, const size_t* recv_bytes, void* stream);
} lmn_collective;
/* fri_min_log = 0 selects the default (16: every all-gather of a small FRI layer's subtree roots costs a
 * collective's latency, hashing a 2^16-row layer on every rank does not).  world = 1 is allowed (the collective is
 * still called). */
int lmn_ctx_set_shard(lmn_ctx* ctx, uint32_t rank, uint32_t world, uint32_t fri_min_log, const lmn_collective* coll);
/* Built-in transport: RCCL over xGMI (librccl is dlopen'ed on first use, the library has no link-time dependency
 * on it).  Rank 0 creates the id, the caller hands the 128 bytes to the other ranks by any means. */
#define LMN_RCCL_ID_BYTES 128
int lmn_rccl_unique_id(uint8_t id_out[LMN_RCCL_ID_BYTES]);
int lmn_ctx_set_shard_rccl(lmn_ctx* ctx, uint32_t rank, uint32_t world, uint32_t fri_min_log,
                           const uint8_t id[LMN_RCCL_ID_BYTES]);
int lmn_ctx_clear_shard(lmn_ctx* ctx);

/* ---- Level 2, host-buffer convenience forms (each call: H2D -> kernel -> D2H; the device-handle forms a
 * `HipBackend` would use are the lmn_col_* functions at the end of this header).  Columns are `ncols` arrays of
 * 2^log_size words. */
int lmn_op_interpolate(lmn_ctx* ctx, uint32_t* cols, uint32_t ncols, uint32_t log_size);           /* PolyOps::interpolate */
int lmn_op_evaluate(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs,
                    uint32_t log_domain, uint32_t* evals_out);                                     /* PolyOps::evaluate */
int lmn_op_merkle_root(lmn_ctx* ctx, const uint32_t* const* cols, const uint32_t* log_sizes, uint32_t ncols,
                       uint8_t root_out[32]);                                                      /* MerkleOps::commit_on_layer chain */
int lmn_op_eval_at_point(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t log_size, const uint32_t point_xy[8],
                         uint32_t value_out[4]);                                                   /* PolyOps::eval_at_point */
/* QuotientOps::accumulate_quotients for the columns of ONE LDE size: `cols[c]` = 2^log_size evaluations
 * (bit-reversed canonic domain); sample i = (column sample_col[i], point sample_point[i], value
 * sample_values[4i..4i+4)), in (column, mask-position) order; points_xy = npoints x 8 words (x then y);
 * out = 4 coordinate columns x 2^log_size. */
int lmn_op_accumulate_quotients(lmn_ctx* ctx, uint32_t log_size, const uint32_t* const* cols, uint32_t ncols,
                                const uint32_t* sample_col, const uint32_t* sample_point, const uint32_t* sample_values,
                                uint32_t nsamples, const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4],
                                uint32_t* out);
/* FriOps::fold_line: src = 4 coordinate columns x 2^log_src on LineDomain(half_odds(log_src)); dst gets 2^(log_src-1). */
int lmn_op_fold_line(lmn_ctx* ctx, const uint32_t* src, uint32_t log_src, const uint32_t alpha[4], uint32_t* dst);
/* FriOps::fold_circle_into_line: dst (4 x 2^(log_src-1), in/out) = dst * alpha^2 + fold(src), src on the
 * canonic circle domain of log_src. */
int lmn_op_fold_circle_into_line(lmn_ctx* ctx, uint32_t* dst, const uint32_t* src, uint32_t log_src, const uint32_t alpha[4]);
/* GrindOps::grind: smallest nonce accepted by the proof-of-work check of `protocol_variant` (LMN_PV_POW_PREFIXED,
 * LMN_PV_MIX_U64_HASHED) on channel state `digest` (host). */
int lmn_op_grind(const uint8_t digest[32], uint32_t pow_bits, uint32_t protocol_variant, uint64_t* nonce_out);
/* PolyOps::evaluate restricted to one aligned block of rows (single-commitment sharding over GPUs):
 * rows [block * 2^(log_domain - log_blocks), (block + 1) * 2^(log_domain - log_blocks)) of every column's
 * evaluation on the 2^log_domain domain, computed from the coefficients alone (1 <= log_blocks <= 3).
 * evals_out: ncols x 2^(log_domain - log_blocks) words. */
int lmn_op_evaluate_block(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs, uint32_t log_domain,
                          uint32_t log_blocks, uint32_t block, uint32_t* evals_out);
int lmn_op_fft_selftest(lmn_ctx* ctx, uint32_t log_size, uint32_t ncols);  /* tiled vs single-layer kernels */

/* ---- Level 2 on device handles ("Each takes device handles (lmn_col*) so columns stay resident").
 * This is what a Rust `HipBackend` would bind in place of `SimdBackend` behind
 * /root/reference/crates/prover/src/prover.rs:38-46,312 and the `TreeBuilder` shim
 * /root/reference/crates/air/src/utils.rs:112-128: stwo's `Col<B, BaseField>` becomes an `lmn_col` with ncols = 1,
 * a `SecureColumnByCoords<B>` an `lmn_col` with ncols = 4 (coordinate-major), a batch of equal-size columns one
 * `lmn_col` with ncols = n.  Data moves between host and HBM only in lmn_col_from_cpu / lmn_col_to_cpu (stwo's
 * `Column::to_cpu` / `FromIterator`); every op below runs on HBM-resident data on the context's stream and returns
 * after enqueueing (ops on one context are stream-ordered; lmn_col_to_cpu and the ops that return values to the
 * host synchronise).  stwo trait and method names are from the un-vendored dependency (unverified here, SURVEY.md
 * §8b); each op is checked against the oracle's restatement of the same operation (tests/level2_checks.py). */
typedef struct lmn_col lmn_col;   /* ncols columns of 2^log_size canonical-M31 words, contiguous, stride 2^log_size */
typedef struct lmn_tree lmn_tree; /* a committed Merkle tree: every layer stays in HBM for decommitment */

int lmn_col_alloc(lmn_ctx* ctx, uint32_t ncols, uint32_t log_size, lmn_col** out);            /* Column::zeros       */
int lmn_col_from_cpu(lmn_ctx* ctx, const uint32_t* host, uint32_t ncols, uint32_t log_size, lmn_col** out);
int lmn_col_to_cpu(lmn_ctx* ctx, const lmn_col* col, uint32_t* host);                          /* Column::to_cpu      */
void lmn_col_free(lmn_ctx* ctx, lmn_col* col);
uint32_t lmn_col_ncols(const lmn_col* col);
uint32_t lmn_col_log_size(const lmn_col* col);
void* lmn_col_device_ptr(const lmn_col* col);                                                  /* for lmn_table.rows etc. */
/* Non-owning handle over columns [first, first + n) of `col` (stwo holds one `Col` per column; here a batch of
 * equal-size columns is one allocation, and a single column / one component's columns are views of it).  Valid
 * while `col` lives; released with lmn_col_free (which then frees nothing on the device). */
int lmn_col_view(lmn_ctx* ctx, const lmn_col* col, uint32_t first, uint32_t n, lmn_col** out);

int lmn_col_bit_reverse(lmn_ctx* ctx, lmn_col* col);                         /* ColumnOps::bit_reverse_column, in place  */
int lmn_col_precompute_twiddles(lmn_ctx* ctx, uint32_t log_size);            /* PolyOps::precompute_twiddles (cached in the ctx) */
int lmn_col_interpolate(lmn_ctx* ctx, lmn_col* evals_to_coeffs);             /* PolyOps::interpolate(_columns), in place */
int lmn_col_evaluate(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t log_domain, lmn_col** evals_out);  /* PolyOps::evaluate(_polynomials) */
int lmn_col_evaluate_block(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t log_domain, uint32_t log_blocks, uint32_t block,
                           lmn_col** evals_out);                             /* row block `block` of 2^log_blocks (sharded commit) */
int lmn_col_extend(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t log_size, lmn_col** out);   /* PolyOps::extend: zero-padded coefficients */
int lmn_col_eval_at_point(lmn_ctx* ctx, const lmn_col* coeffs, uint32_t column, const uint32_t point_xy[8],
                          uint32_t value_out[4]);                            /* PolyOps::eval_at_point */
/* MerkleOps::commit_on_layer for every layer: columns of any mix of sizes (each lmn_col contributes its ncols
 * columns), sorted by size descending (stable) as `MerkleProver::commit` does. */
int lmn_col_commit(lmn_ctx* ctx, const lmn_col* const* cols, uint32_t n, lmn_tree** tree_out);
int lmn_tree_root(lmn_ctx* ctx, const lmn_tree* tree, uint8_t root_out[32]);
uint32_t lmn_tree_log_size(const lmn_tree* tree);
int lmn_tree_layer_to_cpu(lmn_ctx* ctx, const lmn_tree* tree, uint32_t layer_log, uint8_t* hashes_out); /* 32 * 2^layer_log bytes */
void lmn_tree_free(lmn_ctx* ctx, lmn_tree* tree);
int lmn_col_accumulate(lmn_ctx* ctx, lmn_col* dst, const lmn_col* src);     /* AccumulationOps::accumulate: dst += src (same shape) */
/* QuotientOps::accumulate_quotients (one LDE size; samples as in lmn_op_accumulate_quotients): out = 4 coordinate columns */
int lmn_col_accumulate_quotients(lmn_ctx* ctx, const lmn_col* const* cols, uint32_t n, const uint32_t* sample_col,
                                 const uint32_t* sample_point, const uint32_t* sample_values, uint32_t nsamples,
                                 const uint32_t* points_xy, uint32_t npoints, const uint32_t alpha[4], lmn_col** out);
int lmn_col_fold_line(lmn_ctx* ctx, const lmn_col* src, const uint32_t alpha[4], lmn_col** out);       /* FriOps::fold_line */
int lmn_col_fold_circle_into_line(lmn_ctx* ctx, lmn_col* dst, const lmn_col* src, const uint32_t alpha[4]); /* FriOps::fold_circle_into_line */
/* FriOps::decompose: f = g + lambda * v_n on a circle domain in bit-reversed order; lambda = (sum of the first
 * half - sum of the second half) / 2^log_size, g = f - lambda on the first half and f + lambda on the second. */
int lmn_col_decompose(lmn_ctx* ctx, const lmn_col* f, lmn_col** g_out, uint32_t lambda_out[4]);

/* ---- The two per-component stages of `prove` that stwo's constraint framework runs on the backend's columns:
 * with these, the level-2 surface alone reproduces a proof (tests/test_level2_only_prove.py proves examples/simple
 * with lmn_col_* / lmn_tree_* and a host channel and gets the reference's 4 876 known-answer bytes).
 *
 * `elems`: the relation element sets drawn after the main commitment, LMN_N_ELEMS x 8 words: for set e (0 NodeElements,
 * 1 RangeCheck, 2 Sin, 3 Exp2, 4 Log2 - crates/air/src/components/mod.rs:216-235, lookups/mod.rs:44-51) z at
 * elems[8e .. 8e+4) and alpha at elems[8e+4 .. 8e+8).  Sets the component does not use are ignored. */
#define LMN_N_ELEMS 5
/* InteractionClaimGenerator::write_interaction_trace (/root/reference/crates/air/src/components/add/witness.rs:126-167
 * and its 16 siblings) = stwo `LogupTraceGenerator::{new_col, write_frac, finalize_col, finalize_last}` for the
 * relations of component `kind` (LMN_KIND_*): `main` = the component's trace columns on the trace domain (n_cols x 2^k,
 * `Column::index()` order), `pre` = its preprocessed columns (lookup components; NULL otherwise).  Returns the
 * interaction trace (4 base columns per relation, the last group prefix-summed in coset order) and its claimed sum. */
int lmn_col_logup(lmn_ctx* ctx, uint32_t kind, const lmn_col* main, const lmn_col* pre, const uint32_t* elems,
                  lmn_col** interaction_out, uint32_t claimed_sum_out[4]);
/* FrameworkComponent::evaluate_constraint_quotients_on_domain for component `kind`
 * (/root/reference/crates/air/src/components/add/component.rs:38-116 and siblings; components/mod.rs:122,530): on the
 * evaluation domain of log size k + 1 (= the committed LDE at blow-up 2), every constraint of the component times its
 * power of the composition randomness times 1/Z, accumulated into `acc` (4 coordinate columns x 2^(k+1), `acc += ...`:
 * DomainEvaluationAccumulator's column for that size).  `main_lde` / `inter_lde` / `pre_lde`: the component's trace,
 * interaction and preprocessed columns on that domain; `coeffs`: n_coeffs x 4 words, the power of alpha that multiplies
 * constraint i of this component (local constraints in `evaluate` order, then one per relation);
 * `claimed_sum`: the component's claimed logup sum. */
int lmn_col_composition(lmn_ctx* ctx, uint32_t kind, const lmn_col* main_lde, const lmn_col* inter_lde,
                        const lmn_col* pre_lde, const uint32_t* elems, const uint32_t claimed_sum[4],
                        const uint32_t* coeffs, uint32_t n_coeffs, lmn_col* acc);
/* number of constraint slots the kernels of component `kind` emit (local + one per relation; the KAT-era shape:
 * eval_fixed_mul two slots, eval_fixed_recip / _sqrt / _rem one) and of its relations */
uint32_t lmn_kind_constraints(uint32_t kind);
uint32_t lmn_kind_relations(uint32_t kind);
/* Those slots under the constraint-form bits of `protocol_flags` (LMN_PV_*_SLOT(S), LMN_PV_*_NEG; since ABI version 6):
 * proto_index_out[k] = index of kernel slot k among the component's constraints in the protocol (-1: the protocol has no
 * such constraint, its coefficient is 0), sign_out[k] = +1 / -1 (the coefficient of slot k is sign * alpha^(N-1-i) for
 * global constraint index i); 16 entries each.  Returns the number of constraints the component contributes to the
 * composition polynomial under these flags (0: unsupported kind).  What lmn_col_composition's caller builds `coeffs` from. */
uint32_t lmn_kind_constraint_layout(uint32_t kind, uint32_t protocol_flags, int32_t proto_index_out[16], int32_t sign_out[16]);

#ifdef __cplusplus
}
#endif
#endif /* LUMINAIR_HIP_H */
