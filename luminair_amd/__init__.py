"""luminair_amd — MI355X-native Circle-STARK prover backend behind LuminAIR's `prove` API.

Host-side mirror of the reference interface for the hot path only
(`/root/reference/crates/prover/src/prover.rs:28-31`, `crates/air/src/pie.rs:31-66,143-210`,
`crates/prover/src/lib.rs:15-32`).  All proving work happens in hand-written HIP kernels for
gfx950 behind the C ABI of `include/luminair_hip.h`; there is no CPU fallback.
"""
from .pie import (CircuitSettings, ExecutionResources, Lookup, LookupLayout, LuminairError, LuminairPie, LuminairProof,
                  Metadata, RangeCheckLookup, TraceTable, TraceTableKind)
from .graph import DeviceGraph
from .prover import Prover, ProverPool, prove, verify

__all__ = ["Lookup", "LookupLayout", "RangeCheckLookup", "CircuitSettings", "ExecutionResources", "LuminairError", "LuminairPie", "LuminairProof", "Metadata",
           "TraceTable", "TraceTableKind", "Prover", "ProverPool", "prove", "verify", "DeviceGraph"]
