"""Lock-step batches of SMALL proofs (`lmn_batch_*`, libluminair_hip_batch.so; luminair_amd/csrc/batch.h).

The reference proves one pie per `prove(pie, settings)` call (crates/prover/src/prover.rs:28-31); its own benchmark
shape (crates/graph/benches/ops.rs:92-166: 32x32 tensors) and BASELINE config 4 (examples/black-schole-nn) are bound
by kernel launches and host round trips on a GPU.  `BatchProver.prove_batch` proves up to `slots` pies of identical
shape with one launch per pipeline step; every proof is byte-identical to `Prover.prove`'s."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

from . import backend
from .backend import LmnConfig, LmnSettings, LmnTable, LuminairBackendError

BATCH_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libluminair_hip_batch.so")


class BatchProver:
    def __init__(self, device: int = 0, slots: int = 32, protocol_variant: int = backend.VARIANT_KAT,
                 library_path: Optional[str] = None, **pcs):
        self.lib = backend.Library(library_path or BATCH_LIB)
        lib = self.lib.lib
        lib.lmn_batch_create.argtypes = [C.c_int, C.POINTER(LmnConfig), C.c_uint32, C.POINTER(C.c_void_p)]
        lib.lmn_batch_prove.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(LmnTable)), C.c_size_t,
                                        C.POINTER(LmnSettings), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_int)]
        lib.lmn_batch_last_error.restype = C.c_char_p
        lib.lmn_batch_last_error.argtypes = [C.c_void_p]
        lib.lmn_batch_counter.restype = C.c_uint64
        lib.lmn_batch_counter.argtypes = [C.c_void_p, C.c_int]
        lib.lmn_batch_destroy.argtypes = [C.c_void_p]
        cfg = LmnConfig()
        lib.lmn_default_config(C.byref(cfg))
        cfg.protocol_variant = protocol_variant
        for k, v in pcs.items():
            setattr(cfg, k, v)
        self.slots = slots
        self.handle = C.c_void_p()
        rc = lib.lmn_batch_create(device, C.byref(cfg), slots, C.byref(self.handle))
        if rc != 0:
            raise LuminairBackendError(rc, "lmn_batch_create failed: %s" % lib.lmn_strerror(rc).decode())

    def marshal(self, pies: Sequence[Sequence[Tuple[int, object, int]]], luts=None):
        """the C argument arrays of a batch, built once: `prove_batch(marshalled)` then costs no Python per pie (a ctypes caller
        spends ~20 us per pie on the table structs - 4 ms for 192 pies, under the interpreter lock; a C or Rust caller does not)"""
        n = len(pies)
        if n == 0 or n > self.slots:
            raise ValueError("a batch holds 1..%d pies" % self.slots)
        keep, settings = [], None
        arrs = (C.POINTER(LmnTable) * n)()
        n_tables = len(pies[0])
        for i, tables in enumerate(pies):
            arr, nt, st, k = backend.Context._marshal_tables(None, tables, luts)
            if nt != n_tables:
                raise ValueError("the pies of a batch must have the same tables")
            keep.append((arr, k))
            arrs[i] = C.cast(arr, C.POINTER(LmnTable))
            settings = settings or st
        return ("marshalled", n, arrs, n_tables, settings, keep)

    def prove_batch(self, pies, luts=None) -> List[bytes]:
        """pies[i] = [(kind, rows, n_rows)] like `Context.prove_tables`; all pies must have the same kinds and row counts.
        Or what `marshal(pies, luts)` returned."""
        if not (isinstance(pies, tuple) and len(pies) == 6 and pies[0] == "marshalled"):
            pies = self.marshal(pies, luts)
        _, n, arrs, n_tables, settings, keep = pies
        proofs = (C.POINTER(C.c_uint8) * n)()
        lens = (C.c_size_t * n)()
        rcs = (C.c_int * n)()
        lib = self.lib.lib
        rc = lib.lmn_batch_prove(self.handle, n, arrs, n_tables, C.byref(settings), proofs, lens, rcs)
        out = []
        for i in range(n):
            if proofs[i]:
                out.append(C.string_at(proofs[i], lens[i]))
                lib.lmn_free(proofs[i])
            else:
                out.append(None)
        if rc != 0:
            raise LuminairBackendError(rc, "lmn_batch_prove: %s (per pie: %s)"
                                       % ((lib.lmn_batch_last_error(self.handle) or b"").decode(), list(rcs)))
        return out

    def counters(self):
        c = self.lib.lib.lmn_batch_counter
        return {"launches": int(c(self.handle, 0)), "host_waits": int(c(self.handle, 1)),
                "copy_launches": int(c(self.handle, 2)), "direct_copies": int(c(self.handle, 3)),
                "arrival_skew_ms": int(c(self.handle, 4)) / 1e6, "leader_ms": int(c(self.handle, 5)) / 1e6,
                "member_busy_ms": int(c(self.handle, 6)) / 1e6}

    def close(self):
        if self.handle:
            self.lib.lib.lmn_batch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchPool:
    """`groups` lock-step batch groups of `slots` pies each, driven concurrently (one thread per group inside `prove_many`):
    while the members of one group run their host code - a third of a 64-pie batch's 2.9 ms on the reference's benchmark
    shape - the launches of another group use the GPU.  Measured on MI355X, 32x32 Add pies: 1 / 2 / 3 groups of 64 =
    21 / 24 / 30 k proofs/s, 3 groups of 192: 40 k, 46 - 49 k since the workers back off while idle
    (tools/small_proof_groups.py).  Every proof is byte-identical to
    `Prover.prove`'s; `prove_many` returns the proofs in input order."""

    def __init__(self, device: int = 0, groups: int = 3, slots: int = 64, protocol_variant: int = backend.VARIANT_KAT,
                 library_path: Optional[str] = None, **pcs):
        # worker threads per group (LMN_BATCH_THREADS, read by lmn_batch_create; 8 when a group is alone): the throughput collapses
        # when the process has many of them - 3 groups of 192 pies make 46 - 49 k proofs/s with 8 workers each, 43 - 44 k with 4,
        # 20 - 22 k with 16, 12 k with 24 (tools/small_proof_groups.py, profiles/r6_small_proof_groups_threads.txt): at most 24 in all
        groups = max(1, groups)
        own_env = "LMN_BATCH_THREADS" not in os.environ
        if own_env:
            os.environ["LMN_BATCH_THREADS"] = str(max(2, min(8, 24 // groups)))
        try:
            self.groups = [BatchProver(device, slots, protocol_variant, library_path, **pcs) for _ in range(groups)]
        finally:
            if own_env:
                del os.environ["LMN_BATCH_THREADS"]
        self.slots = slots

    def prove_many(self, pies: Sequence[Sequence[Tuple[int, object, int]]], luts=None) -> List[bytes]:
        """all pies of identical shape (as in `BatchProver.prove_batch`); any number of them"""
        import threading
        chunks = [(i, pies[i:i + self.slots]) for i in range(0, len(pies), self.slots)]
        out: List[Optional[bytes]] = [None] * len(pies)
        errors: List[BaseException] = []
        lock = threading.Lock()

        def drive(bp):
            while True:
                with lock:
                    if not chunks or errors:
                        return
                    at, chunk = chunks.pop(0)
                try:
                    out[at:at + len(chunk)] = bp.prove_batch(chunk, luts)
                except BaseException as e:  # noqa: BLE001 - re-raised by the caller's thread
                    with lock:
                        errors.append(e)
                    return

        ths = [threading.Thread(target=drive, args=(bp,)) for bp in self.groups[:max(1, min(len(self.groups), len(chunks)))]]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise errors[0]
        return out

    def close(self):
        for bp in self.groups:
            bp.close()
        self.groups = []
