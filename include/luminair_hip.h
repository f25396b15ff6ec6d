/* luminair_hip.h — C ABI of the MI355X-native LuminAIR prover backend.
 *
 * Drop-in boundary for the reference's
 *     pub fn prove(pie: LuminairPie, settings: CircuitSettings)
 *         -> Result<LuminairProof<Blake2sMerkleHasher>, LuminairError>
 * (/root/reference/crates/prover/src/prover.rs:28-31).  A Rust shim flattens
 * `pie.trace_tables` (crates/air/src/pie.rs:31-66) into `lmn_table`s and deserialises the returned
 * bincode bytes with `LuminairProof::from_bincode` (crates/prover/src/lib.rs:36-43).
 *
 * Plain pointers and sizes only; no C++/torch types.  One context per GPU.  A context is not
 * thread-safe; different contexts are independent.
 */
#ifndef LUMINAIR_HIP_H
#define LUMINAIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

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

/* Protocol variants (SURVEY.md §8c "known deltas"): 0 = the variant pinned by the reference's only
 * known-answer proof (ui/demo/public/proof); 1 = LuminAIR HEAD claim layout + Inputs component,
 * channel encodings unverified ("parity unpinned"). */
#define LMN_VARIANT_KAT 0
#define LMN_VARIANT_PINNED 1

/* Replaces PcsConfig::default() (prover.rs:36) + DEFAULT_FP_SCALE (crates/air/src/lib.rs:23-24). */
typedef struct lmn_config {
  uint32_t pow_bits;         /* default 5 */
  uint32_t log_blowup;       /* default 1 (only 1 is supported: eval domain == LDE domain) */
  uint32_t log_last_layer;   /* default 0 */
  uint32_t n_queries;        /* default 3 */
  uint32_t fp_scale;         /* default 12 */
  uint32_t protocol_variant; /* LMN_VARIANT_* */
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

/* CircuitSettings (crates/air/src/settings.rs): `lookups.{sin,exp2,log2}` must be None; the 8-bit
 * range-check LUT (`lookups.range_check`, crates/graph/src/graph.rs:142-146) is supported and is
 * implied by the presence of a RangeCheckLookup table. */
#define LMN_LOOKUP_RANGE_CHECK 8u
typedef struct lmn_settings {
  uint32_t has_lookups; /* 0 or LMN_LOOKUP_RANGE_CHECK */
} lmn_settings;

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
} lmn_timings;

const char* lmn_strerror(int code);
const char* lmn_last_error(const lmn_ctx* ctx);
void lmn_default_config(lmn_config* cfg);
uint32_t lmn_kind_columns(uint32_t kind); /* 0 if the kind is not supported */

int lmn_ctx_create(int device, const lmn_config* cfg, lmn_ctx** out);
void lmn_ctx_destroy(lmn_ctx* ctx);

/* Replaces prove(pie, settings).  On success *proof_bincode holds `LuminairProof::to_bincode()`
 * bytes owned by the library until lmn_free. */
int lmn_prove(lmn_ctx* ctx, const lmn_table* tables, size_t n_tables, const lmn_settings* settings,
              uint8_t** proof_bincode, size_t* proof_len);
void lmn_free(void* p);
/* Per-stage / per-kernel HIP-event timing is off by default (every event record costs a few
 * microseconds between dependent kernels); enable it for the proofs whose lmn_timings you want. */
int lmn_set_profiling(lmn_ctx* ctx, int enabled);
int lmn_get_timings(const lmn_ctx* ctx, lmn_timings* out);

/* Replaces `verify(proof, settings)` (/root/reference/crates/verifiers/rust/src/verifier.rs:21-143).
 * Host-only (the reference verifier is CPU code too); needs no context and no GPU.  Returns LMN_OK,
 * LMN_ERR_INVALID_LOGUP, LMN_ERR_VERIFICATION or LMN_ERR_SERIALIZATION; the message of the last
 * failure on this thread is available through lmn_last_error(NULL). */
int lmn_verify(const uint8_t* proof_bincode, size_t proof_len, const lmn_settings* settings, uint32_t protocol_variant);

/* Device residency helpers for callers that keep trace tables in HBM. */
int lmn_upload(lmn_ctx* ctx, const void* host, size_t bytes, void** device_out);
void lmn_device_free(lmn_ctx* ctx, void* device_ptr);

/* ---- Level 2: stwo `Backend`-shaped column ops (host buffers in/out; used by the parity tests
 * and by a future Rust HipBackend).  Columns are `ncols` arrays of 2^log_size words. */
int lmn_op_interpolate(lmn_ctx* ctx, uint32_t* cols, uint32_t ncols, uint32_t log_size);           /* PolyOps::interpolate */
int lmn_op_evaluate(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t ncols, uint32_t log_coeffs,
                    uint32_t log_domain, uint32_t* evals_out);                                     /* PolyOps::evaluate */
int lmn_op_merkle_root(lmn_ctx* ctx, const uint32_t* const* cols, const uint32_t* log_sizes, uint32_t ncols,
                       uint8_t root_out[32]);                                                      /* MerkleOps::commit_on_layer chain */
int lmn_op_eval_at_point(lmn_ctx* ctx, const uint32_t* coeffs, uint32_t log_size, const uint32_t point_xy[8],
                         uint32_t value_out[4]);                                                   /* PolyOps::eval_at_point */
int lmn_op_fft_selftest(lmn_ctx* ctx, uint32_t log_size, uint32_t ncols);  /* tiled vs single-layer kernels */

#ifdef __cplusplus
}
#endif
#endif /* LUMINAIR_HIP_H */
