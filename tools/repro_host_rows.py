#!/usr/bin/env python3
"""Reproduce / rule out the driver-observed `host_rows` failure (BENCH_r02.json: a valid 2^20-row Add trace handed
over as HOST buffers to four concurrent contexts was rejected with ProverError(ConstraintsNotSatisfied)).

Cases (each: `rounds` x `per_round` proofs on `inflight` contexts driven by one thread each, every proof compared
with the SHA-256 of the device-rows proof of the same table):
  shared       all contexts are handed the SAME host numpy buffer (what bench.py r2 did)
  private      every context has its own copy of the rows
  fresh        contexts created anew each round, first proof of each is a host-rows proof (arena grows inside the
               concurrent region)
Prints one JSON line per case: proofs, failures by kind (error text or "bytes differ").
Usage: python tools/repro_host_rows.py [--rounds 50] [--per-round 48] [--inflight 4] [--log-rows 20]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def drive(provers, bufs, want, per_ctx):
    """one thread per context (a context is not thread-safe), `per_ctx` proofs each; returns the failure kinds"""
    def worker(i):
        out = []
        for _ in range(per_ctx):
            try:
                got = provers[i].ctx.prove_tables(bufs[i])
                out.append(None if hashlib.sha256(got).digest() == want else "bytes differ")
            except Exception as e:  # noqa: BLE001 - the failure kind is the result
                out.append(str(e))
        return out
    with ThreadPoolExecutor(max_workers=len(provers)) as pool:
        res = [f.result() for f in [pool.submit(worker, i) for i in range(len(provers))]]
    return [r for rs in res for r in rs]


def run_case(name, provers, bufs, want, rounds, per_round):
    fails = {}
    done = 0
    t0 = time.perf_counter()
    for _ in range(rounds):
        for r in drive(provers, bufs, want, max(1, per_round // len(provers))):
            done += 1
            if r is not None:
                fails[r] = fails.get(r, 0) + 1
    print(json.dumps({"case": name, "proofs": done, "failures": fails, "seconds": round(time.perf_counter() - t0, 2)}),
          flush=True)
    return sum(fails.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=50)
    ap.add_argument("--per-round", type=int, default=48)
    ap.add_argument("--inflight", type=int, default=4)
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--cases", default="shared,private,fresh")
    ap.add_argument("--warm", action="store_true",
                    help="initialise every context serially with a device-rows proof first (as bench.py does)")
    args = ap.parse_args()
    import numpy as np
    import luminair_amd
    from luminair_amd import synthetic as syn

    tabs = syn.config2_add_only(1 << args.log_rows, 42)
    p0 = luminair_amd.Prover(0)
    dev = [(k, p0.ctx.upload(r), len(r)) for k, r in tabs]
    want = hashlib.sha256(p0.ctx.prove_tables(dev)).digest()
    bad = 0
    cases = args.cases.split(",")
    if "shared" in cases or "private" in cases:
        provers = [luminair_amd.Prover(0) for _ in range(args.inflight)]
        if args.warm:
            for p in provers:
                p.ctx.prove_tables(dev)
        if "shared" in cases:
            hb = [[(k, r, len(r)) for k, r in tabs] for _ in provers]
            bad += run_case("shared", provers, hb, want, args.rounds, args.per_round)
        if "private" in cases:
            hb = [[(k, np.array(r, copy=True), len(r)) for k, r in tabs] for _ in provers]
            bad += run_case("private", provers, hb, want, args.rounds, args.per_round)
        for p in provers:
            p.ctx.close()
    if "fresh" in cases:
        # as bench.py r2: contexts warmed with device rows (arena sized without host staging), then host rows
        n_fresh = 0
        fails = {}
        t0 = time.perf_counter()
        for _ in range(max(1, args.rounds // 5)):
            provers = [luminair_amd.Prover(0) for _ in range(args.inflight)]
            for p in provers:
                p.ctx.prove_tables(dev)
            hb = [[(k, r, len(r)) for k, r in tabs] for _ in provers]
            for r in drive(provers, hb, want, 4):
                n_fresh += 1
                if r is not None:
                    fails[r] = fails.get(r, 0) + 1
            for p in provers:
                p.ctx.close()
        print(json.dumps({"case": "fresh (arena regrowth inside the concurrent region)", "proofs": n_fresh,
                          "failures": fails, "seconds": round(time.perf_counter() - t0, 2)}), flush=True)
        bad += sum(fails.values())
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
