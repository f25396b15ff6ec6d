#!/usr/bin/env python3
"""The reference's OWN benchmark matrix (crates/graph/benches/ops.rs:92-884: 10 operators x {Trace Generation, Proving,
Verification} at 32x32, seed 42, graph construction excluded by iter_with_setup) on the MI355X path:
  trace generation = DeviceGraph.gen_trace (process_trace kernels on device tensors, tables stay in HBM),
  proving          = lmn_prove on those device-resident tables (PINNED variant = what gen_trace at HEAD emits),
  verification     = lmn_verify (host).
Median of `reps` runs each, in ms.  The reference publishes Add / Mul only (docs/snippets/benchmark-component.mdx:165-179,
GitHub Actions CPU runner): trace 0.0959 / 1.2587 ms, proving 13.05 / 13.12 ms, verification 0.258 / 0.256 ms."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from luminair_amd import backend
from luminair_amd.graph import DeviceGraph

S = 4096
PUBLISHED = {"add": (0.0959, 13.051, 0.2581), "mul": (1.2587, 13.123, 0.2556)}


def build(op, g, rng):
    any_ = lambda: g.input(rng.integers(-2048, 2048, size=(32, 32)))
    pos = lambda: g.input(rng.integers(5, 2048, size=(32, 32)))      # nonzero = true in the reference macro
    if op == "add":
        return g.add(any_(), any_())
    if op == "mul":
        return g.mul(any_(), any_())
    if op == "recip":
        return g.recip(pos())
    if op == "sum_reduce":
        return g.sum_reduce(pos(), 0)
    if op == "max_reduce":
        return g.max_reduce(pos(), 0)
    if op == "sin":
        g.set_lut("sin", 0, 2047)
        return g.sin(pos())
    if op == "sqrt":
        return g.sqrt(pos())
    if op == "exp2":
        g.set_lut("exp2", 0, 2047)
        return g.exp2(pos())
    if op == "less_than":
        return g.less_than(any_(), any_())
    if op == "rem":
        return g.rem(pos(), pos())
    raise ValueError(op)


def med(xs):
    return sorted(xs)[len(xs) // 2]


def main(reps=15):
    lib = backend.default_library()
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(0, cfg, lib)
    out = {}
    for op in ("add", "mul", "recip", "sum_reduce", "max_reduce", "sin", "sqrt", "exp2", "less_than", "rem"):
        tg, pr, vf = [], [], []
        for r in range(reps + 2):
            g = DeviceGraph(ctx)
            g.output(build(op, g, np.random.default_rng(42)))
            t0 = time.perf_counter()
            tables, luts, bufs = g.gen_trace()
            ctx.download(tables[0][1].view(0, 4))                 # the trace calls are stream-ordered: wait for them
            t1 = time.perf_counter()
            proof = ctx.prove_tables(tables, luts)
            t2 = time.perf_counter()
            lib.verify(proof, backend.VARIANT_PINNED)
            t3 = time.perf_counter()
            for b in bufs:
                b.free()
            if r >= 2:
                tg.append(1e3 * (t1 - t0)); pr.append(1e3 * (t2 - t1)); vf.append(1e3 * (t3 - t2))
        out[op] = {"trace_generation_ms": round(med(tg), 4), "proving_ms": round(med(pr), 4), "verification_ms": round(med(vf), 4),
                   "tables": [(int(k), int(n)) for k, _, n in tables], "proof_bytes": len(proof)}
        if op in PUBLISHED:
            out[op]["reference_published_ms"] = dict(zip(("trace_generation", "proving", "verification"), PUBLISHED[op]))
    ctx.close()
    print(json.dumps({"benchmark": "crates/graph/benches/ops.rs at 32x32 on MI355X (device path), median of %d" % reps,
                      "ops": out}))


if __name__ == "__main__":
    main()
