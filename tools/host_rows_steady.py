#!/usr/bin/env python3
"""Steady-state proofs/s of the headline workload with the trace rows handed over as HOST buffers (what the reference's
`prove(pie, settings)` boundary does, prover.rs:28-31): device-resident rows vs page-locked rows vs pageable rows, for
several region lengths and numbers of contexts.  The bench line's `host_rows*` sub-results use one of these settings;
this tool shows how much of their distance to the headline is ramp (all contexts start with an upload at once) and how
much is steady state.  One JSON line per case.
Usage: python tools/host_rows_steady.py [--log-rows 20] [--steps 48,192,384] [--contexts 8,12]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--steps", default="48,192,384")
    ap.add_argument("--contexts", default="8,12")
    ap.add_argument("--modes", default="device,pinned,pageable")
    args = ap.parse_args()
    import luminair_amd
    from luminair_amd import synthetic as syn
    import bench

    tabs = syn.config2_add_only(1 << args.log_rows, 42)
    for n_ctx in [int(x) for x in args.contexts.split(",")]:
        provers = [luminair_amd.Prover(0) for _ in range(n_ctx)]
        lib = provers[0].ctx.lib
        dev = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]
        for p, b in zip(provers, dev):
            p.ctx.prove_tables(b)
        pins = []
        for k, r in tabs:
            a = lib.host_rows(r.shape, r.dtype)
            a.array[...] = r
            pins.append((k, a))
        bufs = {"device": dev,
                "pinned": [[(k, a.array, len(a.array)) for k, a in pins] for _ in provers],
                "pageable": [[(k, r, len(r)) for k, r in tabs] for _ in provers]}
        for mode in args.modes.split(","):
            for steps in [int(x) for x in args.steps.split(",")]:
                r = bench.throughput(provers, bufs[mode], steps, n_ctx)
                print(json.dumps({"rows": mode, "contexts": n_ctx, "proofs": steps, "proofs_per_s": round(r["value"], 1),
                                  "solo_latency_ms": round(bench.solo_latency(provers[0].ctx, bufs[mode][0]), 3)}), flush=True)
        for _, a in pins:
            a.free()
        for bb in dev:
            for _, b_, _ in bb:
                b_.free()
        for p in provers:
            p.ctx.close()


if __name__ == "__main__":
    main()
