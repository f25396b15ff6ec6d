"""`prove(pie, settings)` — same name, argument meaning and error behaviour as
`/root/reference/crates/prover/src/prover.rs:28-31`, executed by the HIP backend."""
from __future__ import annotations

from typing import Optional

from . import backend
from .pie import CircuitSettings, LuminairError, LuminairPie, LuminairProof

_ERR_VARIANT = {
    backend.ERR_EMPTY_TRACE: "TraceError(EmptyTrace)",
    backend.ERR_MAIN_TRACE: "MainTraceEvalGenError",
    backend.ERR_INTERACTION_TRACE: "InteractionTraceEvalGenError",
    backend.ERR_CONSTRAINTS: "ProverError(ConstraintsNotSatisfied)",
    backend.ERR_SERIALIZATION: "SerializationError",
    backend.ERR_INVALID_ARGUMENT: "InvalidArgument",
    backend.ERR_OUT_OF_MEMORY: "OutOfMemory",
    backend.ERR_NO_DEVICE: "NoDevice",
    backend.ERR_INTERNAL: "Internal",
    backend.ERR_VERIFICATION: "StwoVerifierError",
    backend.ERR_INVALID_LOGUP: "InvalidLogUp",
}


class Prover:
    """A prover bound to one GPU; keeps twiddles and the device arena cached across proofs
    (the reference recomputes twiddles per call, prover.rs:38-42)."""

    def __init__(self, device: int = 0, protocol_variant: int = backend.VARIANT_KAT, library=None, **pcs):
        lib = library or backend.default_library()
        cfg = lib.default_config()
        cfg.protocol_variant = protocol_variant
        for k, v in pcs.items():
            setattr(cfg, k, v)
        try:
            self.ctx = backend.Context(device, cfg, lib)
        except backend.LuminairBackendError as e:
            raise LuminairError(_ERR_VARIANT.get(e.code, "Internal"), str(e), e.code) from e

    def prove(self, pie: LuminairPie, settings: Optional[CircuitSettings] = None) -> LuminairProof:
        tables = [(int(t.kind), t.rows, t.n_rows) for t in pie.trace_tables]
        luts = settings.lut_columns(self.ctx.lib) if settings is not None else None
        try:
            return LuminairProof(self.ctx.prove_tables(tables, luts))
        except backend.LuminairBackendError as e:
            raise LuminairError(_ERR_VARIANT.get(e.code, "Internal"), str(e), e.code) from e

    def timings(self) -> dict:
        return self.ctx.timings()


class ProverPool:
    """`n` prover contexts on one GPU driven from ONE thread with `lmn_prove_submit` / `lmn_prove_wait`: the way a
    single-threaded caller of the reference's `prove` keeps the chip busy (eight proofs in flight saturate an MI355X; 12 make
    2 %, 24 make 3 % more proofs/s at 1.9 GB of device memory per context and 2^20 rows).  `prove_many` returns the proofs in input order."""

    def __init__(self, device: int = 0, n: int = 12, protocol_variant: int = backend.VARIANT_KAT, library=None, **pcs):
        self.provers = [Prover(device, protocol_variant, library, **pcs) for _ in range(max(1, n))]

    def prove_many(self, pies, settings: Optional[CircuitSettings] = None):
        out, in_flight = [], []          # in_flight: context indices in submit order
        n = len(self.provers)

        def collect(k):
            try:
                return LuminairProof(self.provers[k].ctx.prove_wait())
            except backend.LuminairBackendError as e:
                raise LuminairError(_ERR_VARIANT.get(e.code, "Internal"), str(e), e.code) from e

        try:
            for i, pie in enumerate(pies):
                k = i % n
                if len(in_flight) == n:
                    out.append(collect(in_flight.pop(0)))        # context k holds the oldest submission
                tables = [(int(t.kind), t.rows, t.n_rows) for t in pie.trace_tables]
                luts = settings.lut_columns(self.provers[k].ctx.lib) if settings is not None else None
                try:
                    self.provers[k].ctx.prove_submit(tables, luts)
                except backend.LuminairBackendError as e:
                    raise LuminairError(_ERR_VARIANT.get(e.code, "Internal"), str(e), e.code) from e
                in_flight.append(k)
            while in_flight:
                out.append(collect(in_flight.pop(0)))
        finally:
            for k in in_flight:              # an error above: drain what is still running before the buffers go away
                try:
                    self.provers[k].ctx.prove_wait()
                except backend.LuminairBackendError:
                    pass
        return out

    def close(self):
        for p in self.provers:
            p.ctx.close()


_default: Optional[Prover] = None


def prove(pie: LuminairPie, settings: Optional[CircuitSettings] = None) -> LuminairProof:
    """Drop-in for the reference's free function `prove(pie, settings)` on GPU 0."""
    global _default
    if _default is None:
        _default = Prover(0)
    return _default.prove(pie, settings)


def verify(proof: LuminairProof, settings: Optional[CircuitSettings] = None,
           protocol_variant: int = backend.VARIANT_KAT, library=None) -> None:
    """Drop-in for the reference's `verify(proof, settings)` (crates/verifiers/rust/src/verifier.rs:21-143):
    host-side check of a proof's bincode bytes; raises LuminairError(StwoVerifierError | InvalidLogUp | ...)."""
    lib = library or backend.default_library()
    c_settings, keep = None, None
    if settings is not None and (settings.layouts or settings.lookups or settings.range_check):
        # what verifier.rs:47-57 reads from the settings: which lookups exist and how large their LUTs are
        mask, luts = 0, []
        for bit, name in enumerate(("sin", "exp2", "log2")):
            if settings.layouts and name in settings.layouts:
                mask |= 1 << bit
                luts.append(backend.LmnLut(backend.LUT_KINDS[name], settings.layouts[name].layout.log_size, None, None))
            elif settings.lookups and name in settings.lookups:
                mask |= 1 << bit
                luts.append(backend.LmnLut(backend.LUT_KINDS[name], len(settings.lookups[name][0]).bit_length() - 1, None, None))
        if settings.range_check is not None:
            mask |= 8
        keep = (backend.LmnLut * max(len(luts), 1))(*luts)
        c_settings = backend.LmnSettings(mask, len(luts), keep)
    try:
        lib.verify(proof.to_bincode() if isinstance(proof, LuminairProof) else bytes(proof), protocol_variant,
                   settings=c_settings)
    except backend.LuminairBackendError as e:
        raise LuminairError(_ERR_VARIANT.get(e.code, "Internal"), str(e), e.code) from e
