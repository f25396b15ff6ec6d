"""CPU oracle for the LuminAIR `prove` hot path (TEST INFRASTRUCTURE ONLY).

This package is a numpy/pure-Python restatement of the Circle-STARK protocol that
`/root/reference/crates/prover/src/prover.rs:28-319` drives through the un-vendored
`stwo` crate (rev 0790eba4, `Cargo.toml:21-23`).  It is the *checker* for the HIP
product path in `luminair_amd/`: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product path never does.

Two restatements live here: vectorised numpy (`oracle.prover.NumpyKernels`, the readable one) and
plain C with OpenMP (`oracle/c/stark_kernels.c` via `oracle.cbackend.CKernels`, fast enough to prove
the full 2^20-row workload in seconds: used for full-size byte comparisons and as `cpu_baseline`).
Both are driven by the same `oracle.prover.prove` and must agree byte-for-byte (tests/test_oracle_c.py).

Parity pin: `tests/golden/kat_simple/{proof,settings,graph.dot}` — the one cryptographic
known-answer test the reference ships (`ui/demo/public/proof`, a proof of the
`examples/simple` graph made by an older LuminAIR/stwo).  `oracle.prover.prove` with
`ProtocolVariant.KAT` reproduces those 4 876 bytes bit-for-bit from the three input
tensors (tests/test_oracle_kat.py).  For the *pinned* rev the reference holds no
byte-level vectors at all, so the `PINNED` variant is **parity unpinned**.
"""
