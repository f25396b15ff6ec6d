"""ctypes binding of the C ABI in include/luminair_hip.h.

The product library is `luminair_amd/csrc/libluminair_hip.so` (hipcc, gfx950).  There is no CPU
fallback: loading fails loudly if the library was not built, and `Context()` fails with
LMN_ERR_NO_DEVICE when no HIP device is present.  (`Library(path)` accepts an explicit path so
the test-suite can load the test-only emulation build; the package itself never does.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libluminair_hip.so")

LMN_OK = 0
ERR_EMPTY_TRACE, ERR_MAIN_TRACE, ERR_INTERACTION_TRACE, ERR_CONSTRAINTS = -1, -2, -3, -4
ERR_SERIALIZATION, ERR_INVALID_ARGUMENT, ERR_OUT_OF_MEMORY, ERR_NO_DEVICE, ERR_INTERNAL = -5, -6, -7, -8, -100
ERR_VERIFICATION, ERR_INVALID_LOGUP = -9, -10
# lmn_config.protocol_variant: OR of LMN_PV_* bits (include/luminair_hip.h).  All clear = the protocol of the reference's
# known-answer proof; VARIANT_PINNED = what the reference at HEAD is believed to run (transcript bits from memory of the
# un-vendored stwo: unpinned).  tools/pin_variant.py finds the combination a given proof was made with.
PV_CLAIM17, PV_LUT_DRAWS4, PV_MIX_U64_HASHED, PV_DRAW_CTR_U32, PV_POW_PREFIXED = 0x1, 0x2, 0x4, 0x8, 0x10
PV_MUL_ONE_SLOT, PV_RECIP_TWO_SLOTS, PV_RECIP_NEG, PV_SQRT_TWO_SLOTS, PV_SQRT_NEG, PV_REM_TWO_SLOTS, PV_REM_NEG = (
    0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000)
PV_TRANSCRIPT_MASK, PV_FORMS_MASK = 0x1f, 0x7f00
PV_NAMES = {PV_CLAIM17: "claim17", PV_LUT_DRAWS4: "lut_draws4", PV_MIX_U64_HASHED: "mix_u64_hashed",
            PV_DRAW_CTR_U32: "draw_ctr_u32", PV_POW_PREFIXED: "pow_prefixed", PV_MUL_ONE_SLOT: "mul_one_slot",
            PV_RECIP_TWO_SLOTS: "recip_two_slots", PV_RECIP_NEG: "recip_neg", PV_SQRT_TWO_SLOTS: "sqrt_two_slots",
            PV_SQRT_NEG: "sqrt_neg", PV_REM_TWO_SLOTS: "rem_two_slots", PV_REM_NEG: "rem_neg"}
VARIANT_KAT, VARIANT_PINNED = 0, PV_TRANSCRIPT_MASK
CHECK_PARSE, CHECK_SHAPE, CHECK_LOGUP_SUM, CHECK_OODS, CHECK_POW, CHECK_TREE_DECOMMIT, CHECK_FRI_DECOMMIT, CHECK_FRI_FOLDS = (
    1, 2, 4, 8, 16, 32, 64, 128)
CHECK_ALL = 0xff
CHECK_NAMES = {CHECK_PARSE: "parse", CHECK_SHAPE: "shape", CHECK_LOGUP_SUM: "logup_sum", CHECK_OODS: "oods",
               CHECK_POW: "pow", CHECK_TREE_DECOMMIT: "tree_decommit", CHECK_FRI_DECOMMIT: "fri_decommit",
               CHECK_FRI_FOLDS: "fri_folds"}
STEP_NAMES = ["root_preprocessed", "claim", "root_main", "interaction_claim", "root_interaction", "root_composition",
              "sampled_values", "fri_first_layer", "fri_inner_layer", "fri_last_layer", "pow_nonce"]
ROUND_HALF_AWAY, ROUND_HALF_EVEN, ROUND_TRUNC, ROUND_FLOOR = 0, 1, 2, 3
MAX_TRANSCRIPT_STEPS = 48
TABLE_ROWS_ON_DEVICE = 1


class LmnConfig(C.Structure):
    _fields_ = [("pow_bits", C.c_uint32), ("log_blowup", C.c_uint32), ("log_last_layer", C.c_uint32),
                ("n_queries", C.c_uint32), ("fp_scale", C.c_uint32), ("protocol_variant", C.c_uint32)]


class LmnTable(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("flags", C.c_uint32), ("n_rows", C.c_uint64), ("rows", C.c_void_p)]


class LmnLut(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("log_size", C.c_uint32), ("col0", C.c_void_p), ("col1", C.c_void_p)]


class LmnSettings(C.Structure):
    _fields_ = [("has_lookups", C.c_uint32), ("n_luts", C.c_uint32), ("luts", C.POINTER(LmnLut))]


LUT_KINDS = {"sin": 0, "exp2": 1, "log2": 2}   # LMN_LUT_*


class LmnView(C.Structure):
    """Strided view of a device tensor (`lmn_view`): shape of the op's output, strides in elements, 0 = expanded."""
    _fields_ = [("ndim", C.c_uint32), ("shape", C.c_uint32 * 4), ("strides", C.c_int64 * 4), ("offset", C.c_int64)]

    @staticmethod
    def make(shape, strides, offset: int = 0) -> "LmnView":
        n = len(shape)
        if not 1 <= n <= 4:
            raise ValueError("views have 1..4 dimensions")
        return LmnView(n, (C.c_uint32 * 4)(*(list(shape) + [0] * (4 - n))),
                       (C.c_int64 * 4)(*(list(strides) + [0] * (4 - n))), int(offset))


class LmnNodeInfo(C.Structure):
    """`NodeInfo` fields `process_trace` reads (crates/graph/src/utils.rs / op/prim.rs:980-990)."""
    _fields_ = [("node_id", C.c_uint32), ("input_ids", C.c_uint32 * 2), ("num_consumers", C.c_uint32),
                ("is_final_output", C.c_uint32), ("input_mults", C.c_int32 * 2)]


class LmnRange(C.Structure):
    """`Range(Fixed, Fixed)`: inclusive range of Fixed<12> values (i64)."""
    _fields_ = [("lo", C.c_int64), ("hi", C.c_int64)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p,
                            C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p)


class LmnCollective(C.Structure):
    """`lmn_collective`: the exchange primitives of a sharded proof - an in-place all-gather on device memory and
    (optional) the all-to-all that re-partitions column-parallel LDEs into row blocks."""
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("group_begin", C.c_void_p), ("group_end", C.c_void_p),
                ("all_to_all", ALL_TO_ALL_FN)]


RCCL_ID_BYTES = 128


class LmnTimings(C.Structure):
    _fields_ = [("total_ms", C.c_float)] + [(n, C.c_float) for n in (
        "transpose_ms", "main_commit_ms", "logup_ms", "interaction_commit_ms", "composition_ms",
        "composition_commit_ms", "oods_ms", "quotients_ms", "fri_ms", "decommit_ms", "fft_ms", "merkle_ms")] + [
        ("fft_bytes", C.c_uint64), ("merkle_bytes", C.c_uint64), ("fft_launches", C.c_uint32),
        ("merkle_launches", C.c_uint32), ("fft_butterflies", C.c_uint64), ("merkle_compressions", C.c_uint64),
        ("merkle_fused_ms", C.c_float), ("merkle_fused_launches", C.c_uint32), ("merkle_fused_bytes", C.c_uint64),
        ("merkle_fused_compressions", C.c_uint64), ("shard_a2a_bytes", C.c_uint64), ("shard_gather_bytes", C.c_uint64),
        ("shard_a2a_calls", C.c_uint32), ("shard_gather_calls", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class LmnTranscriptStep(C.Structure):
    _fields_ = [("step", C.c_uint32), ("index", C.c_uint32), ("digest", C.c_uint8 * 32)]


class LmnVerifyReport(C.Structure):
    """`lmn_verify_report`: which checks of the verifier a proof passes under one set of protocol flags, and the
    channel digest after every mix of the replay."""
    _fields_ = [("checks_run", C.c_uint32), ("checks_passed", C.c_uint32), ("checks_failed", C.c_uint32),
                ("n_steps", C.c_uint32), ("steps", LmnTranscriptStep * MAX_TRANSCRIPT_STEPS),
                ("first_failure", C.c_char * 128)]


API_VERSION = 6   # LMN_API_VERSION of include/luminair_hip.h

EXPORTS = ["lmn_abi_version", "lmn_kind_padding_row", "lmn_strerror", "lmn_last_error", "lmn_default_config", "lmn_kind_columns", "lmn_ctx_create",
           "lmn_ctx_destroy", "lmn_prove", "lmn_prove_submit", "lmn_prove_wait", "lmn_free", "lmn_get_timings", "lmn_set_profiling", "lmn_upload", "lmn_device_free", "lmn_verify",
           "lmn_host_alloc", "lmn_host_free", "lmn_host_register", "lmn_host_unregister",
           "lmn_op_interpolate", "lmn_op_evaluate", "lmn_op_merkle_root", "lmn_op_eval_at_point",
           "lmn_op_fft_selftest", "lmn_op_accumulate_quotients", "lmn_op_fold_line", "lmn_op_fold_circle_into_line",
           "lmn_op_grind", "lmn_device_alloc", "lmn_download", "lmn_trace_elementwise", "lmn_trace_sum_reduce",
           "lmn_trace_elementwise_v", "lmn_trace_contiguous", "lmn_trace_lut", "lmn_trace_lut_ranges", "lmn_trace_less_than", "lmn_trace_max_reduce", "lmn_upload_to", "lmn_device_copy", "lmn_op_evaluate_block",
           "lmn_verify_with_config", "lmn_verify_diagnose", "lmn_kind_constraint_layout", "lmn_lut_log_size", "lmn_lut_from_ranges", "lmn_lut_from_ranges_r", "lmn_col_alloc", "lmn_col_from_cpu", "lmn_col_to_cpu", "lmn_col_free", "lmn_col_ncols",
           "lmn_col_log_size", "lmn_col_device_ptr", "lmn_col_view", "lmn_col_bit_reverse", "lmn_col_precompute_twiddles",
           "lmn_col_interpolate", "lmn_col_evaluate", "lmn_col_evaluate_block", "lmn_col_extend", "lmn_col_eval_at_point",
           "lmn_col_commit", "lmn_tree_root", "lmn_tree_log_size", "lmn_tree_layer_to_cpu", "lmn_tree_free",
           "lmn_col_accumulate", "lmn_col_accumulate_quotients", "lmn_col_fold_line", "lmn_col_fold_circle_into_line",
           "lmn_col_decompose", "lmn_col_logup", "lmn_col_composition", "lmn_kind_constraints", "lmn_kind_relations", "lmn_ctx_set_shard", "lmn_rccl_unique_id", "lmn_ctx_set_shard_rccl", "lmn_ctx_clear_shard"]


class LuminairBackendError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("%s (code %d)" % (message, code))
        self.code = code


class Library:
    def __init__(self, path: Optional[str] = None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise LuminairBackendError(ERR_NO_DEVICE, "HIP backend library not built: %s (run __graft_entry__.build())"
                                       % path)
        self.path = path
        lib = C.CDLL(path)
        self.lib = lib
        # the ctypes structures below mirror include/luminair_hip.h at this ABI version; a library built from another
        # header (e.g. a 56-byte lmn_view without `offset`) must be refused, not fed garbage
        try:
            lib.lmn_abi_version.restype = C.c_uint32
            got = int(lib.lmn_abi_version())
        except AttributeError:
            got = None
        if got != API_VERSION:
            raise LuminairBackendError(ERR_INVALID_ARGUMENT, "%s implements ABI version %s, this package binds version %d"
                                       % (path, got, API_VERSION))
        lib.lmn_strerror.restype = C.c_char_p
        lib.lmn_strerror.argtypes = [C.c_int]
        lib.lmn_last_error.restype = C.c_char_p
        lib.lmn_last_error.argtypes = [C.c_void_p]
        lib.lmn_default_config.argtypes = [C.POINTER(LmnConfig)]
        lib.lmn_kind_columns.restype = C.c_uint32
        lib.lmn_kind_columns.argtypes = [C.c_uint32]
        lib.lmn_kind_relations.restype = C.c_uint32
        lib.lmn_kind_relations.argtypes = [C.c_uint32]
        lib.lmn_kind_padding_row.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        lib.lmn_ctx_create.argtypes = [C.c_int, C.POINTER(LmnConfig), C.POINTER(C.c_void_p)]
        lib.lmn_ctx_destroy.argtypes = [C.c_void_p]
        lib.lmn_prove.argtypes = [C.c_void_p, C.POINTER(LmnTable), C.c_size_t, C.POINTER(LmnSettings),
                                  C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        lib.lmn_prove_submit.argtypes = [C.c_void_p, C.POINTER(LmnTable), C.c_size_t, C.POINTER(LmnSettings)]
        lib.lmn_prove_wait.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        lib.lmn_free.argtypes = [C.c_void_p]
        lib.lmn_get_timings.argtypes = [C.c_void_p, C.POINTER(LmnTimings)]
        lib.lmn_set_profiling.argtypes = [C.c_void_p, C.c_int]
        lib.lmn_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.lmn_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        lib.lmn_host_free.argtypes = [C.c_void_p]
        lib.lmn_host_register.argtypes = [C.c_void_p, C.c_size_t]
        lib.lmn_host_unregister.argtypes = [C.c_void_p]
        lib.lmn_device_free.argtypes = [C.c_void_p, C.c_void_p]
        lib.lmn_verify.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(LmnSettings), C.c_uint32]
        lib.lmn_verify_with_config.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(LmnSettings), C.POINTER(LmnConfig)]
        lib.lmn_op_interpolate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.lmn_op_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.lmn_op_evaluate_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_void_p]
        lib.lmn_op_merkle_root.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32,
                                           C.c_void_p]
        lib.lmn_op_eval_at_point.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lmn_op_fft_selftest.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        lib.lmn_op_accumulate_quotients.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                    C.c_void_p, C.c_void_p]
        lib.lmn_op_fold_line.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lmn_op_fold_circle_into_line.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.lmn_op_grind.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        lib.lmn_device_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.lmn_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.lmn_upload_to.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.lmn_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.lmn_lut_log_size.argtypes = [C.POINTER(LmnRange), C.c_uint32, C.POINTER(C.c_uint32)]
        lib.lmn_lut_from_ranges.argtypes = [C.c_uint32, C.POINTER(LmnRange), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lmn_lut_from_ranges_r.argtypes = [C.c_uint32, C.POINTER(LmnRange), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_void_p]
        lib.lmn_verify_diagnose.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(LmnSettings), C.POINTER(LmnConfig),
                                            C.POINTER(LmnVerifyReport)]
        lib.lmn_kind_constraint_layout.restype = C.c_uint32
        lib.lmn_kind_constraint_layout.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        VP, U32 = C.c_void_p, C.c_uint32
        lib.lmn_col_alloc.argtypes = [VP, U32, U32, C.POINTER(VP)]
        lib.lmn_col_from_cpu.argtypes = [VP, VP, U32, U32, C.POINTER(VP)]
        lib.lmn_col_to_cpu.argtypes = [VP, VP, VP]
        lib.lmn_col_free.argtypes = [VP, VP]
        lib.lmn_col_free.restype = None
        lib.lmn_col_ncols.argtypes = [VP]
        lib.lmn_col_ncols.restype = U32
        lib.lmn_col_log_size.argtypes = [VP]
        lib.lmn_col_log_size.restype = U32
        lib.lmn_col_device_ptr.argtypes = [VP]
        lib.lmn_col_device_ptr.restype = VP
        lib.lmn_col_view.argtypes = [VP, VP, U32, U32, C.POINTER(VP)]
        lib.lmn_col_bit_reverse.argtypes = [VP, VP]
        lib.lmn_col_precompute_twiddles.argtypes = [VP, U32]
        lib.lmn_col_interpolate.argtypes = [VP, VP]
        lib.lmn_col_evaluate.argtypes = [VP, VP, U32, C.POINTER(VP)]
        lib.lmn_col_evaluate_block.argtypes = [VP, VP, U32, U32, U32, C.POINTER(VP)]
        lib.lmn_col_extend.argtypes = [VP, VP, U32, C.POINTER(VP)]
        lib.lmn_col_eval_at_point.argtypes = [VP, VP, U32, VP, VP]
        lib.lmn_col_commit.argtypes = [VP, C.POINTER(VP), U32, C.POINTER(VP)]
        lib.lmn_tree_root.argtypes = [VP, VP, VP]
        lib.lmn_tree_log_size.argtypes = [VP]
        lib.lmn_tree_log_size.restype = U32
        lib.lmn_tree_layer_to_cpu.argtypes = [VP, VP, U32, VP]
        lib.lmn_tree_free.argtypes = [VP, VP]
        lib.lmn_tree_free.restype = None
        lib.lmn_col_accumulate.argtypes = [VP, VP, VP]
        lib.lmn_col_accumulate_quotients.argtypes = [VP, C.POINTER(VP), U32, VP, VP, VP, U32, VP, U32, VP, C.POINTER(VP)]
        lib.lmn_col_fold_line.argtypes = [VP, VP, VP, C.POINTER(VP)]
        lib.lmn_col_fold_circle_into_line.argtypes = [VP, VP, VP, VP]
        lib.lmn_col_decompose.argtypes = [VP, VP, C.POINTER(VP), VP]
        lib.lmn_col_logup.argtypes = [VP, U32, VP, VP, VP, C.POINTER(VP), VP]
        lib.lmn_col_composition.argtypes = [VP, U32, VP, VP, VP, VP, VP, VP, U32, VP]
        lib.lmn_kind_constraints.argtypes = [U32]
        lib.lmn_kind_constraints.restype = U32
        lib.lmn_kind_relations.argtypes = [U32]
        lib.lmn_kind_relations.restype = U32
        lib.lmn_ctx_set_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(LmnCollective)]
        lib.lmn_rccl_unique_id.argtypes = [C.c_void_p]
        lib.lmn_ctx_set_shard_rccl.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.lmn_ctx_clear_shard.argtypes = [C.c_void_p]
        lib.lmn_trace_elementwise_v.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(LmnView), C.c_void_p,
                                                C.POINTER(LmnView), C.c_uint64, C.POINTER(LmnNodeInfo), C.c_void_p,
                                                C.c_uint64, C.c_void_p]
        lib.lmn_trace_contiguous.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(LmnView), C.c_uint64,
                                             C.POINTER(LmnNodeInfo), C.c_void_p, C.c_uint64, C.c_void_p]
        lib.lmn_trace_lut.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(LmnView), C.c_uint64,
                                      C.POINTER(LmnNodeInfo), C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_void_p]
        lib.lmn_trace_lut_ranges.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(LmnView), C.c_uint64,
                                             C.POINTER(LmnNodeInfo), C.c_void_p, C.POINTER(LmnRange), C.c_uint32, C.c_void_p,
                                             C.c_void_p, C.c_uint64, C.c_void_p]
        lib.lmn_trace_less_than.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(LmnView), C.c_void_p, C.POINTER(LmnView),
                                            C.c_uint64, C.POINTER(LmnNodeInfo), C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.c_void_p]
        lib.lmn_trace_max_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                             C.POINTER(LmnNodeInfo), C.c_void_p, C.c_uint64, C.c_void_p]
        lib.lmn_trace_sum_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                             C.POINTER(LmnNodeInfo), C.c_void_p, C.c_uint64, C.c_void_p]
        lib.lmn_trace_elementwise.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                              C.POINTER(LmnNodeInfo), C.c_void_p, C.c_uint64, C.c_void_p]

    def host_rows(self, shape, dtype=np.uint32) -> "PinnedArray":
        """A page-locked numpy array (`lmn_host_alloc`): trace rows written here reach the GPU by direct DMA."""
        return PinnedArray(self, shape, dtype)

    def host_register(self, arr: np.ndarray):
        """Page-lock an existing C-contiguous array in place (`lmn_host_register`); pair with host_unregister."""
        assert arr.flags["C_CONTIGUOUS"]
        rc = self.lib.lmn_host_register(arr.ctypes.data_as(C.c_void_p), arr.nbytes)
        if rc != LMN_OK:
            raise LuminairBackendError(rc, self.lib.lmn_strerror(rc).decode())

    def host_unregister(self, arr: np.ndarray):
        rc = self.lib.lmn_host_unregister(arr.ctypes.data_as(C.c_void_p))
        if rc != LMN_OK:
            raise LuminairBackendError(rc, self.lib.lmn_strerror(rc).decode())

    def default_config(self) -> LmnConfig:
        cfg = LmnConfig()
        self.lib.lmn_default_config(C.byref(cfg))
        return cfg

    def kind_columns(self, kind: int) -> int:
        return int(self.lib.lmn_kind_columns(kind))

    def lut_log_size(self, ranges: Sequence[Tuple[int, int]]) -> int:
        """`LookupLayout::new(ranges).log_size` (crates/air/src/preprocessed.rs:49-52)."""
        arr = (LmnRange * len(ranges))(*[LmnRange(int(a), int(b)) for a, b in ranges])
        out = C.c_uint32()
        rc = self.lib.lmn_lut_log_size(arr, len(ranges), C.byref(out))
        if rc != LMN_OK:
            raise LuminairBackendError(rc, "bad LUT ranges")
        return int(out.value)

    def lut_from_ranges(self, name: str, ranges: Sequence[Tuple[int, int]], log_size: Optional[int] = None):
        """The two preprocessed columns of a sin / exp2 / log2 LUT from the reference's `LookupLayout`
        (`SinPreProcessed::gen_column` and siblings).  Returns (col0, col1), uint32 arrays of 2^log_size words."""
        if log_size is None:
            log_size = self.lut_log_size(ranges)
        arr = (LmnRange * len(ranges))(*[LmnRange(int(a), int(b)) for a, b in ranges])
        c0 = np.empty(1 << log_size, dtype=np.uint32)
        c1 = np.empty(1 << log_size, dtype=np.uint32)
        rc = self.lib.lmn_lut_from_ranges(LUT_KINDS[name], arr, len(ranges), log_size, c0.ctypes.data, c1.ctypes.data)
        if rc != LMN_OK:
            raise LuminairBackendError(rc, "bad LUT ranges")
        return c0, c1

    def grind(self, digest: bytes, pow_bits: int, variant: int = VARIANT_KAT) -> int:
        """GrindOps::grind on a 32-byte channel digest."""
        out = C.c_uint64()
        rc = self.lib.lmn_op_grind(bytes(digest), pow_bits, variant, C.byref(out))
        if rc != LMN_OK:
            raise LuminairBackendError(rc, self.lib.lmn_strerror(rc).decode())
        return int(out.value)

    def verify(self, proof: bytes, variant: int = VARIANT_KAT, config: Optional[LmnConfig] = None,
               settings: Optional[LmnSettings] = None) -> None:
        """`verify(proof, settings)` on the host; raises LuminairBackendError on rejection.  `config` = the
        verifier's own PcsConfig (default: PcsConfig::default(), as the reference hard-codes)."""
        sp = C.byref(settings) if settings is not None else None
        if config is not None:
            rc = self.lib.lmn_verify_with_config(proof, len(proof), sp, C.byref(config))
        else:
            rc = self.lib.lmn_verify(proof, len(proof), sp, variant)
        if rc != LMN_OK:
            msg = self.lib.lmn_last_error(None).decode() or self.lib.lmn_strerror(rc).decode()
            raise LuminairBackendError(rc, msg)


    def diagnose(self, proof: bytes, variant: int, config: Optional[LmnConfig] = None,
                 settings: Optional[LmnSettings] = None):
        """`lmn_verify_diagnose`: (return code, LmnVerifyReport) - which verifier checks `proof` passes under the
        protocol flags `variant`, without stopping at the first failure."""
        cfg = LmnConfig()
        if config is not None:
            C.memmove(C.byref(cfg), C.byref(config), C.sizeof(LmnConfig))
        else:
            self.lib.lmn_default_config(C.byref(cfg))
        cfg.protocol_variant = variant
        rep = LmnVerifyReport()
        sp = C.byref(settings) if settings is not None else None
        rc = self.lib.lmn_verify_diagnose(proof, len(proof), sp, C.byref(cfg), C.byref(rep))
        return rc, rep

    def constraint_layout(self, kind: int, variant: int):
        """(n_protocol, proto_index[16], sign[16]) of `lmn_kind_constraint_layout`."""
        pi, sg = (C.c_int32 * 16)(), (C.c_int32 * 16)()
        n = int(self.lib.lmn_kind_constraint_layout(kind, variant, pi, sg))
        return n, list(pi), list(sg)


_default_library: Optional[Library] = None


def default_library() -> Library:
    global _default_library
    if _default_library is None:
        _default_library = Library()
    return _default_library


class PinnedArray:
    """numpy view of an `lmn_host_alloc` buffer; `.array` is valid until `free()` (or garbage collection of this object)."""

    def __init__(self, library: "Library", shape, dtype=np.uint32):
        self.library = library
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        rc = library.lib.lmn_host_alloc(nbytes, C.byref(p))
        if rc != LMN_OK:
            raise LuminairBackendError(rc, library.lib.lmn_strerror(rc).decode())
        self.ptr = p.value
        buf = (C.c_uint8 * nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.library.lib.lmn_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    def __init__(self, ctx: "Context", ptr: int, nbytes: int, owned: bool = True):
        self.ctx, self.ptr, self.nbytes, self.owned = ctx, ptr, nbytes, owned

    def view(self, offset: int, nbytes: int) -> "DeviceBuffer":
        """A sub-range of this allocation (not owned: freeing it is a no-op)."""
        if offset < 0 or offset + nbytes > self.nbytes:
            raise ValueError("view outside the allocation")
        return DeviceBuffer(self.ctx, self.ptr + offset, nbytes, owned=False)

    def free(self):
        if self.ptr and self.owned:
            self.ctx.lib.lib.lmn_device_free(self.ctx.handle, self.ptr)
        self.ptr = 0


class Col:
    """`lmn_col`: ncols columns of 2^log_size words resident in HBM (stwo `Col<B, _>` / `SecureColumnByCoords<B>`).
    Every method is one level-2 op on the device data; only `from_cpu` / `to_cpu` move bytes over PCIe."""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.handle = ctx, handle

    @property
    def ncols(self) -> int:
        return int(self.ctx.lib.lib.lmn_col_ncols(self.handle))

    @property
    def log_size(self) -> int:
        return int(self.ctx.lib.lib.lmn_col_log_size(self.handle))

    @property
    def device_ptr(self) -> int:
        return int(self.ctx.lib.lib.lmn_col_device_ptr(self.handle))

    def _new(self, fn, *args) -> "Col":
        out = C.c_void_p()
        self.ctx._check(fn(self.ctx.handle, self.handle, *args, C.byref(out)))
        return Col(self.ctx, out)

    def to_cpu(self) -> np.ndarray:
        host = np.empty((self.ncols, 1 << self.log_size), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lib.lmn_col_to_cpu(self.ctx.handle, self.handle, host.ctypes.data))
        return host

    def free(self):
        if self.handle:
            self.ctx.lib.lib.lmn_col_free(self.ctx.handle, self.handle)
            self.handle = None

    def view(self, first: int, n: int) -> "Col":
        """Non-owning handle over columns [first, first + n) (valid while this handle lives)."""
        return self._new(self.ctx.lib.lib.lmn_col_view, first, n)

    def bit_reverse(self) -> "Col":
        self.ctx._check(self.ctx.lib.lib.lmn_col_bit_reverse(self.ctx.handle, self.handle))
        return self

    def interpolate(self) -> "Col":
        """evaluations -> coefficients, in place"""
        self.ctx._check(self.ctx.lib.lib.lmn_col_interpolate(self.ctx.handle, self.handle))
        return self

    def evaluate(self, log_domain: int) -> "Col":
        return self._new(self.ctx.lib.lib.lmn_col_evaluate, log_domain)

    def evaluate_block(self, log_domain: int, log_blocks: int, block: int) -> "Col":
        return self._new(self.ctx.lib.lib.lmn_col_evaluate_block, log_domain, log_blocks, block)

    def extend(self, log_size: int) -> "Col":
        return self._new(self.ctx.lib.lib.lmn_col_extend, log_size)

    def eval_at_point(self, column: int, point_xy: Sequence[int]) -> Tuple[int, int, int, int]:
        pt = (C.c_uint32 * 8)(*[int(v) for v in point_xy])
        out = (C.c_uint32 * 4)()
        self.ctx._check(self.ctx.lib.lib.lmn_col_eval_at_point(self.ctx.handle, self.handle, column, pt, out))
        return tuple(int(v) for v in out)

    def accumulate(self, other: "Col") -> "Col":
        """self += other"""
        self.ctx._check(self.ctx.lib.lib.lmn_col_accumulate(self.ctx.handle, self.handle, other.handle))
        return self

    def fold_line(self, alpha) -> "Col":
        al = (C.c_uint32 * 4)(*[int(v) for v in alpha])
        return self._new(self.ctx.lib.lib.lmn_col_fold_line, al)

    def fold_circle_into_line(self, src: "Col", alpha) -> "Col":
        """self = self * alpha^2 + fold(src)"""
        al = (C.c_uint32 * 4)(*[int(v) for v in alpha])
        self.ctx._check(self.ctx.lib.lib.lmn_col_fold_circle_into_line(self.ctx.handle, self.handle, src.handle, al))
        return self

    def decompose(self):
        """FriOps::decompose -> (g, lambda)"""
        out = C.c_void_p()
        lam = (C.c_uint32 * 4)()
        self.ctx._check(self.ctx.lib.lib.lmn_col_decompose(self.ctx.handle, self.handle, C.byref(out), lam))
        return Col(self.ctx, out), tuple(int(v) for v in lam)


class Tree:
    """`lmn_tree`: a committed Merkle tree whose layers stay in HBM."""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.handle = ctx, handle

    def root(self) -> bytes:
        out = (C.c_uint8 * 32)()
        self.ctx._check(self.ctx.lib.lib.lmn_tree_root(self.ctx.handle, self.handle, out))
        return bytes(out)

    @property
    def log_size(self) -> int:
        return int(self.ctx.lib.lib.lmn_tree_log_size(self.handle))

    def layer(self, layer_log: int) -> List[bytes]:
        buf = (C.c_uint8 * (32 << layer_log))()
        self.ctx._check(self.ctx.lib.lib.lmn_tree_layer_to_cpu(self.ctx.handle, self.handle, layer_log, buf))
        raw = bytes(buf)
        return [raw[32 * i:32 * i + 32] for i in range(1 << layer_log)]

    def free(self):
        if self.handle:
            self.ctx.lib.lib.lmn_tree_free(self.ctx.handle, self.handle)
            self.handle = None


class Context:
    """One prover context per GPU (`lmn_ctx`)."""

    def __init__(self, device: int = 0, config: Optional[LmnConfig] = None, library: Optional[Library] = None):
        self.lib = library or default_library()
        self.config = config or self.lib.default_config()
        h = C.c_void_p()
        rc = self.lib.lib.lmn_ctx_create(device, C.byref(self.config), C.byref(h))
        if rc != LMN_OK:
            msg = self.lib.lib.lmn_last_error(None).decode() or self.lib.lib.lmn_strerror(rc).decode()
            raise LuminairBackendError(rc, msg)
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            slab = getattr(self, "_graph_slab", None)      # DeviceGraph's cached allocation (luminair_amd/graph.py)
            if slab is not None:
                slab.free()
                self._graph_slab = None
            self.lib.lib.lmn_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != LMN_OK:
            msg = self.lib.lib.lmn_last_error(self.handle).decode() or self.lib.lib.lmn_strerror(rc).decode()
            raise LuminairBackendError(rc, msg)

    # ---- level-2 ops on device handles (lmn_col_* / lmn_tree_*)
    def col_from_cpu(self, cols: np.ndarray) -> Col:
        a = np.ascontiguousarray(cols, dtype=np.uint32)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        ncols, n = a.shape
        if n & (n - 1) or n == 0:
            raise ValueError("column length must be a power of two")
        out = C.c_void_p()
        self._check(self.lib.lib.lmn_col_from_cpu(self.handle, a.ctypes.data, ncols, n.bit_length() - 1, C.byref(out)))
        return Col(self, out)

    def col_zeros(self, ncols: int, log_size: int) -> Col:
        out = C.c_void_p()
        self._check(self.lib.lib.lmn_col_alloc(self.handle, ncols, log_size, C.byref(out)))
        return Col(self, out)

    def precompute_twiddles(self, log_size: int):
        self._check(self.lib.lib.lmn_col_precompute_twiddles(self.handle, log_size))

    def commit(self, cols: Sequence[Col]) -> Tree:
        arr = (C.c_void_p * max(len(cols), 1))(*[c.handle for c in cols])
        out = C.c_void_p()
        self._check(self.lib.lib.lmn_col_commit(self.handle, arr, len(cols), C.byref(out)))
        return Tree(self, out)

    def col_accumulate_quotients(self, cols: Sequence[Col], samples, points, alpha) -> Col:
        """As `accumulate_quotients`, on resident columns; returns the secure column (4 coordinate columns)."""
        arr = (C.c_void_p * len(cols))(*[c.handle for c in cols])
        sc = np.array([s[0] for s in samples], dtype=np.uint32)
        sp = np.array([s[1] for s in samples], dtype=np.uint32)
        sv = np.array([list(s[2]) for s in samples], dtype=np.uint32).reshape(-1)
        pts = np.array([list(p) for p in points], dtype=np.uint32).reshape(-1)
        al = (C.c_uint32 * 4)(*[int(v) for v in alpha])
        out = C.c_void_p()
        self._check(self.lib.lib.lmn_col_accumulate_quotients(
            self.handle, arr, len(cols), sc.ctypes.data, sp.ctypes.data, sv.ctypes.data, len(samples), pts.ctypes.data,
            len(points), al, C.byref(out)))
        return Col(self, out)

    N_ELEMS = 5   # LMN_N_ELEMS: NodeElements, RangeCheck, Sin, Exp2, Log2

    @staticmethod
    def _elems_words(elems):
        """elems: {set index: (z words, alpha words)} or a flat sequence of N_ELEMS * 8 words"""
        if isinstance(elems, dict):
            flat = [0] * (8 * Context.N_ELEMS)
            for e, (z, a) in elems.items():
                flat[8 * e:8 * e + 4] = [int(v) for v in z]
                flat[8 * e + 4:8 * e + 8] = [int(v) for v in a]
            elems = flat
        if len(elems) != 8 * Context.N_ELEMS:
            raise ValueError("elems: %d words expected" % (8 * Context.N_ELEMS))
        return (C.c_uint32 * (8 * Context.N_ELEMS))(*[int(v) for v in elems])

    def col_logup(self, kind: int, main: Col, pre: Optional[Col], elems) -> Tuple[Col, Tuple[int, int, int, int]]:
        """`write_interaction_trace` of component `kind` on resident columns -> (interaction columns, claimed sum)."""
        out = C.c_void_p()
        claimed = (C.c_uint32 * 4)()
        self._check(self.lib.lib.lmn_col_logup(self.handle, kind, main.handle, pre.handle if pre is not None else None,
                                               self._elems_words(elems), C.byref(out), claimed))
        return Col(self, out), tuple(int(v) for v in claimed)

    def col_composition(self, kind: int, main_lde: Col, inter_lde: Col, pre_lde: Optional[Col], elems, claimed, coeffs,
                        acc: Col) -> Col:
        """acc += sum_k constraint_k * coeffs[k] / Z of component `kind` on its evaluation domain."""
        cw = (C.c_uint32 * (4 * len(coeffs)))(*[int(v) for c in coeffs for v in c])
        cl = (C.c_uint32 * 4)(*[int(v) for v in claimed])
        self._check(self.lib.lib.lmn_col_composition(
            self.handle, kind, main_lde.handle, inter_lde.handle, pre_lde.handle if pre_lde is not None else None,
            self._elems_words(elems), cl, cw, len(coeffs), acc.handle))
        return acc

    # ---- single-proof sharding (lmn_ctx_set_shard*)
    def set_shard(self, rank: int, world: int, all_gather, fri_min_log: int = 0, all_to_all=None):
        """Shard every later `prove` over `world` contexts (one per GPU, one per process).  `all_gather(buf_ptr,
        bytes_per_rank, stream)` is the in-place all-gather of `lmn_collective` (device pointer as an int);
        `all_to_all(send_ptr, send_off, send_bytes, recv_ptr, recv_off, recv_bytes, stream)` (lists of `world` ints) the
        optional second primitive."""
        def _cb(_user, buf, nbytes, stream):
            try:
                all_gather(buf, nbytes, stream)
                return 0
            except Exception:   # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        def _a2a(_user, send, so, sb, recv, ro, rb, stream):
            try:
                all_to_all(send, [so[p] for p in range(world)], [sb[p] for p in range(world)], recv,
                           [ro[p] for p in range(world)], [rb[p] for p in range(world)], stream)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = ALL_GATHER_FN(_cb)
        cb2 = ALL_TO_ALL_FN(_a2a) if all_to_all is not None else ALL_TO_ALL_FN()
        coll = LmnCollective(None, cb, None, None, cb2)
        self._check(self.lib.lib.lmn_ctx_set_shard(self.handle, rank, world, fri_min_log, C.byref(coll)))
        # keep the trampolines alive as long as the context uses them (a rejected call keeps the previous ones)
        self._shard_cb, self._shard_coll = (cb, cb2), coll

    def rccl_unique_id(self) -> bytes:
        buf = (C.c_uint8 * RCCL_ID_BYTES)()
        rc = self.lib.lib.lmn_rccl_unique_id(buf)
        if rc != LMN_OK:
            raise LuminairBackendError(rc, self.lib.lib.lmn_last_error(None).decode() or "lmn_rccl_unique_id failed")
        return bytes(buf)

    def set_shard_rccl(self, rank: int, world: int, unique_id: bytes, fri_min_log: int = 0):
        """Built-in transport: RCCL over xGMI on the prover's own stream (`unique_id` from rank 0's rccl_unique_id)."""
        if len(unique_id) != RCCL_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % RCCL_ID_BYTES)
        buf = (C.c_uint8 * RCCL_ID_BYTES)(*unique_id)
        self._check(self.lib.lib.lmn_ctx_set_shard_rccl(self.handle, rank, world, fri_min_log, buf))

    def clear_shard(self):
        self._check(self.lib.lib.lmn_ctx_clear_shard(self.handle))
        self._shard_cb = self._shard_coll = None

    def upload(self, arr: np.ndarray) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        out = C.c_void_p()
        self._check(self.lib.lib.lmn_upload(self.handle, arr.ctypes.data, arr.nbytes, C.byref(out)))
        return DeviceBuffer(self, out.value, arr.nbytes)

    def alloc(self, nbytes: int) -> DeviceBuffer:
        out = C.c_void_p()
        self._check(self.lib.lib.lmn_device_alloc(self.handle, nbytes, C.byref(out)))
        return DeviceBuffer(self, out.value, nbytes)

    def upload_to(self, buf: DeviceBuffer, arr: np.ndarray) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > buf.nbytes:
            raise ValueError("destination too small")
        self._check(self.lib.lib.lmn_upload_to(self.handle, arr.ctypes.data, arr.nbytes, buf.ptr))
        return buf

    def device_copy(self, dst_ptr: int, src_ptr: int, nbytes: int):
        """Stream-ordered device-to-device copy (returns without waiting)."""
        self._check(self.lib.lib.lmn_device_copy(self.handle, dst_ptr, src_ptr, nbytes))

    def download(self, buf: DeviceBuffer, dtype=np.uint32) -> np.ndarray:
        host = np.empty(buf.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._check(self.lib.lib.lmn_download(self.handle, buf.ptr, host.ctypes.data, buf.nbytes))
        return host

    def trace_elementwise(self, kind: int, lhs: DeviceBuffer, rhs: Optional[DeviceBuffer], n: int, node_id: int,
                          input_ids, num_consumers: int, is_final_output: bool = False, input_mults=(-1, -1),
                          rows: Optional[DeviceBuffer] = None, row_offset: int = 0,
                          lhs_view: Optional[LmnView] = None, rhs_view: Optional[LmnView] = None,
                          out: Optional[DeviceBuffer] = None):
        """`process_trace` of one Add / Mul / Recip node on device tensors (int32 Fixed<12> values).
        Returns (rows DeviceBuffer, out DeviceBuffer)."""
        ncols = self.lib.kind_columns(kind)
        if rows is None:
            rows = self.alloc((row_offset + n) * ncols * 4)
        out = out or self.alloc(n * 4)
        ids = list(input_ids) + [0] * (2 - len(input_ids))
        mults = list(input_mults) + [0] * (2 - len(input_mults))
        info = LmnNodeInfo(node_id, (C.c_uint32 * 2)(*ids), num_consumers, 1 if is_final_output else 0,
                           (C.c_int32 * 2)(*mults))
        if lhs_view is None and rhs_view is None:
            self._check(self.lib.lib.lmn_trace_elementwise(self.handle, kind, lhs.ptr, rhs.ptr if rhs is not None else None,
                                                           n, C.byref(info), rows.ptr, row_offset, out.ptr))
        else:
            self._check(self.lib.lib.lmn_trace_elementwise_v(
                self.handle, kind, lhs.ptr, C.byref(lhs_view) if lhs_view is not None else None,
                rhs.ptr if rhs is not None else None, C.byref(rhs_view) if rhs_view is not None else None, n,
                C.byref(info), rows.ptr, row_offset, out.ptr))
        return rows, out

    def trace_contiguous(self, inp: DeviceBuffer, in_size: int, out_size: int, node_id: int, input_id: int,
                         num_consumers: int, is_final_output: bool = False, input_mult: int = -1,
                         view: Optional[LmnView] = None, rows: Optional[DeviceBuffer] = None, row_offset: int = 0,
                         out: Optional[DeviceBuffer] = None):
        """`LuminairContiguous::process_trace` in the reference's row rule: max(in_size, out_size) rows."""
        n_rows = max(in_size, out_size)
        if rows is None:
            rows = self.alloc((row_offset + n_rows) * 11 * 4)
        out = out or self.alloc(out_size * 4)
        info = LmnNodeInfo(node_id, (C.c_uint32 * 2)(input_id, 0), num_consumers, 1 if is_final_output else 0,
                           (C.c_int32 * 2)(input_mult, 0))
        self._check(self.lib.lib.lmn_trace_contiguous(self.handle, inp.ptr, in_size, C.byref(view) if view is not None else None,
                                                      out_size, C.byref(info), rows.ptr, row_offset, out.ptr))
        return rows, out

    def trace_lut(self, kind: int, inp: DeviceBuffer, n: int, node_id: int, input_id: int, num_consumers: int,
                  lut_col1: DeviceBuffer, lo: int = 0, lut_len: int = 0, mult: DeviceBuffer = None,
                  is_final_output: bool = False, input_mult: int = -1, view: Optional[LmnView] = None,
                  rows: Optional[DeviceBuffer] = None, row_offset: int = 0, out: Optional[DeviceBuffer] = None,
                  ranges: Optional[Sequence[Tuple[int, int]]] = None):
        """`process_trace` of a Sin / Exp2 / Log2 node; `mult` (the lookup component's table) is updated in place.
        The LUT enumerates either the single range lo .. lo + lut_len - 1 or `ranges` (ascending, disjoint)."""
        if rows is None:
            rows = self.alloc((row_offset + n) * 12 * 4)
        out = out or self.alloc(n * 4)
        info = LmnNodeInfo(node_id, (C.c_uint32 * 2)(input_id, 0), num_consumers, 1 if is_final_output else 0,
                           (C.c_int32 * 2)(input_mult, 0))
        vp = C.byref(view) if view is not None else None
        if ranges is None:
            self._check(self.lib.lib.lmn_trace_lut(self.handle, kind, inp.ptr, vp, n, C.byref(info), lut_col1.ptr, lo,
                                                   lut_len, mult.ptr, rows.ptr, row_offset, out.ptr))
        else:
            arr = (LmnRange * len(ranges))(*[LmnRange(int(a), int(b)) for a, b in ranges])
            self._check(self.lib.lib.lmn_trace_lut_ranges(self.handle, kind, inp.ptr, vp, n, C.byref(info), lut_col1.ptr,
                                                          arr, len(ranges), mult.ptr, rows.ptr, row_offset, out.ptr))
        return rows, out

    def trace_sum_reduce(self, inp: DeviceBuffer, front: int, dim: int, back: int, node_id: int, input_id: int,
                         num_consumers: int, is_final_output: bool = False, input_mult: int = -1,
                         rows: Optional[DeviceBuffer] = None, row_offset: int = 0, maximum: bool = False,
                         out: Optional[DeviceBuffer] = None):
        """`LuminairSumReduce::process_trace` (or MaxReduce with maximum=True) on a contiguous
        (front, dim, back) int32 device tensor."""
        n_rows, n_out = front * dim * back, front * back
        if rows is None:
            rows = self.alloc((row_offset + n_rows) * (15 if maximum else 14) * 4)
        out = out or self.alloc(n_out * 4)
        info = LmnNodeInfo(node_id, (C.c_uint32 * 2)(input_id, 0), num_consumers, 1 if is_final_output else 0,
                           (C.c_int32 * 2)(input_mult, 0))
        fn = self.lib.lib.lmn_trace_max_reduce if maximum else self.lib.lib.lmn_trace_sum_reduce
        self._check(fn(self.handle, inp.ptr, front, dim, back, C.byref(info), rows.ptr, row_offset, out.ptr))
        return rows, out

    def trace_less_than(self, lhs: DeviceBuffer, rhs: DeviceBuffer, n: int, node_id: int, input_ids, num_consumers: int,
                        range_check_mult: DeviceBuffer, is_final_output: bool = False, input_mults=(-1, -1),
                        rows: Optional[DeviceBuffer] = None, row_offset: int = 0,
                        lhs_view: Optional[LmnView] = None, rhs_view: Optional[LmnView] = None,
                        out: Optional[DeviceBuffer] = None):
        """`LuminairLessThan::process_trace`; `range_check_mult` (256 words) is the RangeCheckLookup table."""
        if rows is None:
            rows = self.alloc((row_offset + n) * 22 * 4)
        out = out or self.alloc(n * 4)
        info = LmnNodeInfo(node_id, (C.c_uint32 * 2)(*input_ids), num_consumers, 1 if is_final_output else 0,
                           (C.c_int32 * 2)(*input_mults))
        self._check(self.lib.lib.lmn_trace_less_than(
            self.handle, lhs.ptr, C.byref(lhs_view) if lhs_view is not None else None, rhs.ptr,
            C.byref(rhs_view) if rhs_view is not None else None, n, C.byref(info), range_check_mult.ptr, rows.ptr,
            row_offset, out.ptr))
        return rows, out

    def _marshal_tables(self, tables, luts):
        """-> (LmnTable array, n, LmnSettings, objects that must stay alive while the library reads them)"""
        n = len(tables)
        arr = (LmnTable * max(n, 1))()
        keep = []
        for i, (kind, rows, n_rows) in enumerate(tables):
            arr[i].kind = kind
            arr[i].n_rows = n_rows
            if isinstance(rows, DeviceBuffer):
                arr[i].flags = TABLE_ROWS_ON_DEVICE
                arr[i].rows = rows.ptr
            else:
                a = np.ascontiguousarray(rows, dtype=np.uint32)
                keep.append(a)
                arr[i].flags = 0
                arr[i].rows = a.ctypes.data
        luts = luts or {}
        lut_arr = (LmnLut * max(len(luts), 1))()
        for i, (name, (c0, c1)) in enumerate(luts.items()):
            c0 = np.ascontiguousarray(c0, dtype=np.uint32)
            c1 = np.ascontiguousarray(c1, dtype=np.uint32)
            if len(c0) != len(c1) or len(c0) & (len(c0) - 1) or len(c0) == 0:
                raise LuminairBackendError(-2, "LUT columns must have equal power-of-two lengths")
            keep += [c0, c1]
            lut_arr[i] = LmnLut(LUT_KINDS[name], len(c0).bit_length() - 1, c0.ctypes.data, c1.ctypes.data)
        settings = LmnSettings(0, len(luts), lut_arr)
        keep += [arr, lut_arr, settings]
        return arr, n, settings, keep

    def prove_tables(self, tables: Sequence[Tuple[int, object, int]], luts=None) -> bytes:
        """tables: [(kind, rows, n_rows)] where rows is a uint32 ndarray (host) or a DeviceBuffer;
        luts: {"sin" | "exp2" | "log2": (col0, col1)} preprocessed LUT columns (uint32, 2^k words each)."""
        arr, n, settings, _keep = self._marshal_tables(tables, luts)
        out = C.POINTER(C.c_uint8)()
        out_len = C.c_size_t()
        self._check(self.lib.lib.lmn_prove(self.handle, arr, n, C.byref(settings), C.byref(out), C.byref(out_len)))
        data = C.string_at(out, out_len.value)
        self.lib.lib.lmn_free(out)
        return data

    def prove_submit(self, tables: Sequence[Tuple[int, object, int]], luts=None):
        """`lmn_prove_submit`: start the proof on the context's own worker thread and return; collect it with
        `prove_wait()`.  One thread keeps N proofs in flight with N contexts."""
        arr, n, settings, keep = self._marshal_tables(tables, luts)
        self._check(self.lib.lib.lmn_prove_submit(self.handle, arr, n, C.byref(settings)))
        self._pending = keep          # borrowed by the library until prove_wait returns

    def prove_wait(self) -> bytes:
        out = C.POINTER(C.c_uint8)()
        out_len = C.c_size_t()
        try:
            self._check(self.lib.lib.lmn_prove_wait(self.handle, C.byref(out), C.byref(out_len)))
        finally:
            self._pending = None
        data = C.string_at(out, out_len.value)
        self.lib.lib.lmn_free(out)
        return data

    def set_profiling(self, enabled: bool):
        self._check(self.lib.lib.lmn_set_profiling(self.handle, 1 if enabled else 0))

    def timings(self) -> dict:
        t = LmnTimings()
        self._check(self.lib.lib.lmn_get_timings(self.handle, C.byref(t)))
        return t.as_dict()

    # ---- level-2 ops (host buffers)
    def interpolate(self, cols: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(cols, dtype=np.uint32).copy()
        ncols, n = a.shape
        self._check(self.lib.lib.lmn_op_interpolate(self.handle, a.ctypes.data, ncols, n.bit_length() - 1))
        return a

    def evaluate(self, coeffs: np.ndarray, log_domain: int) -> np.ndarray:
        a = np.ascontiguousarray(coeffs, dtype=np.uint32)
        ncols, n = a.shape
        out = np.empty((ncols, 1 << log_domain), dtype=np.uint32)
        self._check(self.lib.lib.lmn_op_evaluate(self.handle, a.ctypes.data, ncols, n.bit_length() - 1, log_domain,
                                                 out.ctypes.data))
        return out

    def evaluate_block(self, coeffs: np.ndarray, log_domain: int, log_blocks: int, block: int) -> np.ndarray:
        """Rows of block `block` (of 2^log_blocks equal row blocks) of the evaluation of (ncols, 2^k) coefficient
        columns on the 2^log_domain domain."""
        a = np.ascontiguousarray(coeffs, dtype=np.uint32)
        ncols, n = a.shape
        out = np.empty((ncols, 1 << (log_domain - log_blocks)), dtype=np.uint32)
        self._check(self.lib.lib.lmn_op_evaluate_block(self.handle, a.ctypes.data, ncols, n.bit_length() - 1, log_domain,
                                                       log_blocks, block, out.ctypes.data))
        return out

    def merkle_root(self, cols: List[np.ndarray]) -> bytes:
        arrs = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        logs = (C.c_uint32 * max(len(arrs), 1))(*[len(a).bit_length() - 1 for a in arrs])
        root = (C.c_uint8 * 32)()
        self._check(self.lib.lib.lmn_op_merkle_root(self.handle, ptrs, logs, len(arrs), root))
        return bytes(root)

    def eval_at_point(self, coeffs: np.ndarray, point_xy: Sequence[int]) -> Tuple[int, int, int, int]:
        a = np.ascontiguousarray(coeffs, dtype=np.uint32)
        pt = (C.c_uint32 * 8)(*[int(v) for v in point_xy])
        out = (C.c_uint32 * 4)()
        self._check(self.lib.lib.lmn_op_eval_at_point(self.handle, a.ctypes.data, len(a).bit_length() - 1, pt, out))
        return tuple(int(v) for v in out)

    def accumulate_quotients(self, cols, samples, points, alpha) -> np.ndarray:
        """QuotientOps::accumulate_quotients for equal-size columns.  cols: [2^k uint32 arrays];
        samples: [(col index, point index, (4 words))] in (column, mask position) order; points: [(8 words)];
        alpha: 4 words.  Returns the 4 coordinate columns, shape (4, 2^k)."""
        keep = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        L = len(keep[0])
        ptrs = (C.c_void_p * len(keep))(*[c.ctypes.data for c in keep])
        sc = np.array([s[0] for s in samples], dtype=np.uint32)
        sp = np.array([s[1] for s in samples], dtype=np.uint32)
        sv = np.array([list(s[2]) for s in samples], dtype=np.uint32).reshape(-1)
        pts = np.array([list(p) for p in points], dtype=np.uint32).reshape(-1)
        al = (C.c_uint32 * 4)(*[int(v) for v in alpha])
        out = np.empty((4, L), dtype=np.uint32)
        self._check(self.lib.lib.lmn_op_accumulate_quotients(
            self.handle, L.bit_length() - 1, ptrs, len(keep), sc.ctypes.data, sp.ctypes.data, sv.ctypes.data, len(samples),
            pts.ctypes.data, len(points), al, out.ctypes.data))
        return out

    def fold_line(self, src: np.ndarray, alpha) -> np.ndarray:
        """FriOps::fold_line on 4 coordinate columns (4, 2^k) -> (4, 2^(k-1))."""
        a = np.ascontiguousarray(src, dtype=np.uint32)
        L = a.shape[1]
        out = np.empty((4, L // 2), dtype=np.uint32)
        al = (C.c_uint32 * 4)(*[int(v) for v in alpha])
        self._check(self.lib.lib.lmn_op_fold_line(self.handle, a.ctypes.data, L.bit_length() - 1, al, out.ctypes.data))
        return out

    def fold_circle_into_line(self, dst: np.ndarray, src: np.ndarray, alpha) -> np.ndarray:
        """FriOps::fold_circle_into_line: returns dst * alpha^2 + fold(src)."""
        a = np.ascontiguousarray(src, dtype=np.uint32)
        d = np.ascontiguousarray(dst, dtype=np.uint32).copy()
        al = (C.c_uint32 * 4)(*[int(v) for v in alpha])
        self._check(self.lib.lib.lmn_op_fold_circle_into_line(self.handle, d.ctypes.data, a.ctypes.data,
                                                              a.shape[1].bit_length() - 1, al))
        return d

    def fft_selftest(self, log_size: int, ncols: int = 2):
        self._check(self.lib.lib.lmn_op_fft_selftest(self.handle, log_size, ncols))
