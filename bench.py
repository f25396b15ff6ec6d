#!/usr/bin/env python3
"""bench.py — proofs/sec of the MI355X prove hot path on BASELINE.json config 2
(single Add-op AIR, 2^20 trace rows per proof), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one whole proof (AoS trace rows resident in HBM -> bincode proof bytes on the host).
Proofs are independent, so ranks shard proofs with no data-path collective ("weak" scaling);
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks timing.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak
# ALU ceilings measured on MI355X with tools/microbench.hip; informational only
BLAKE2S_PEAK_GCOMP = 39.0                                  # G compressions/s, chip-wide
BUTTERFLY_PEAK_G = 1.0 / (1.0 / 6326.0 + 2.0 / 12800.0)    # G butterflies/s = 1 M31 mul + 2 add|sub


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--log-rows", type=int, default=20, help="log2 rows of the Add trace (default 20 = BASELINE config 2)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="independent proofs in flight per GPU (one prover context + HIP stream each)")
    ap.add_argument("--host-rows", action="store_true",
                    help="hand the trace rows over as host buffers (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-anchor", action="store_true", help="skip the 32x32 reference-shape latency anchor")
    ap.add_argument("--cpu-sample-log", type=int, default=20)
    return ap.parse_args(argv)


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def timed_region(step_fn, steps, warmup, barrier, device_sync):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides."""
    for _ in range(warmup):
        step_fn()
    device_sync()
    barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    device_sync()
    barrier()
    device_sync()
    return time.perf_counter() - t0


def aggregate(elapsed, world, steps_per_rank, reduce_max):
    """value = units all ranks processed / max-over-ranks time."""
    tmax = reduce_max(elapsed)
    return {"seconds": tmax, "value": world * steps_per_rank / tmax, "ms_per_step": 1e3 * tmax / steps_per_rank}


def cpu_baseline(sample_log, full_log):
    """The oracle's plain-C restatement (oracle/c/stark_kernels.c, OpenMP over the host cores, driven by
    oracle/prover.py) timed on this box on the same workload.  Bounded: one cold proof (builds the
    twiddle/domain tables, as the reference does per proof) plus one warm proof."""
    from luminair_amd import synthetic as syn
    from oracle.cbackend import CKernels
    from oracle.prover import prove as oracle_prove
    K = CKernels()
    tabs = syn.config2_add_only(1 << sample_log, 42)
    t0 = time.perf_counter()
    oracle_prove(tabs, kernels=K)
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle_prove(tabs, kernels=K)
    warm = time.perf_counter() - t0
    scale = float(1 << (full_log - sample_log))
    small = syn.config2_graph_faithful(1024, 42)
    from oracle.channel import ProtocolVariant
    oracle_prove(small, kernels=K, variant=ProtocolVariant.PINNED)
    t0 = time.perf_counter()
    oracle_prove(small, kernels=K, variant=ProtocolVariant.PINNED)
    small_ms = 1e3 * (time.perf_counter() - t0)
    return {"reference_shape_32x32_add_ms": small_ms,"value": 1.0 / (warm * scale), "unit": "proofs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "C/OpenMP oracle proof of a 2^%d-row Add trace: %.2f s warm (tables cached), %.2f s cold%s"
                      % (sample_log, warm, cold, "" if scale == 1 else "; scaled x%d to 2^%d rows" % (scale, full_log))}


def main(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    import numpy as np
    import torch
    import luminair_amd
    from luminair_amd import synthetic as syn

    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = local_rank if torch.cuda.is_available() else 0

    # `inflight` independent prover contexts per GPU (own HIP stream + device arena each); a step is
    # still one whole proof, steps are dealt round-robin to the contexts and run concurrently
    from concurrent.futures import ThreadPoolExecutor
    inflight = max(1, min(args.inflight, args.steps))
    provers = [luminair_amd.Prover(dev) for _ in range(inflight)]
    prover = provers[0]
    tabs = syn.config2_add_only(1 << args.log_rows, 42 + rank)   # each rank proves its own trace
    if args.host_rows:
        bufs = [[(k, r, len(r)) for k, r in tabs] for p in provers]             # PCIe-inclusive variant
    else:
        bufs = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]   # trace rows resident in HBM
    out = {}
    pool = ThreadPoolExecutor(max_workers=inflight)
    pending = []

    def one(i):
        out["proof"] = provers[i].ctx.prove_tables(bufs[i])

    counter = {"n": 0}

    def step():
        # keep at most `inflight` proofs outstanding (ctypes releases the GIL inside lmn_prove)
        i = counter["n"] % inflight
        counter["n"] += 1
        if len(pending) >= inflight:
            pending.pop(0).result()
        pending.append(pool.submit(one, i))

    def drain():
        while pending:
            pending.pop(0).result()

    # one-time context initialisation (twiddle tables, device arena) for EVERY in-flight context, so the
    # W warmup + K timed steps below never include a context's first-use setup whatever W is
    for i in range(inflight):
        one(i)

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])

    def device_sync():
        drain()
        torch.cuda.synchronize()

    def reduce_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = timed_region(step, args.steps, args.warmup, barrier, device_sync)
    agg = aggregate(elapsed, world, args.steps, reduce_max)
    # single-proof latency (one proof alone on the GPU) and the per-kernel timings behind `roofline`
    lat = []
    for _ in range(21):
        t0 = time.perf_counter()
        one(0)
        lat.append(1e3 * (time.perf_counter() - t0))
    lat.sort()
    latency_ms = lat[len(lat) // 2]                      # p50 of 21 solo proofs
    latency_p95_ms = lat[int(0.95 * (len(lat) - 1))]
    # one more solo proof with HIP-event profiling switched on: the source of `roofline` and `stage_ms`
    prover.ctx.set_profiling(True)
    one(0)
    prover.ctx.set_profiling(False)

    # roofline per kernel family, from HIP events recorded by the library on the prover's own stream
    # around every transform / tree of the solo proof above (lmn_timings).  `traffic` comes from the
    # committed rocprofv3 PMC summary (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction
    # applied: 2*FETCH_SIZE + WRITE_SIZE), per launch like `achieved`.
    tm = prover.timings()
    pmc = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_summary.json")) as f:
            pmc = json.load(f)["kernels"]
    except Exception:
        pass
    fams = {
        "k_fft_staged": (tm["fft_ms"], tm["fft_bytes"], tm["fft_launches"], ["k_fft_staged<false>", "k_fft_staged<true>"]),
        "k_merkle_fused": (tm["merkle_fused_ms"], tm["merkle_fused_bytes"], tm["merkle_fused_launches"],
                           ["k_merkle_fused", "k_merkle_fused<0>", "k_merkle_fused<1>", "k_merkle_fused<2>"]),
    }
    alu = {
        "k_fft_staged": (tm["fft_butterflies"], BUTTERFLY_PEAK_G, "G butterflies/s"),
        "k_merkle_fused": (tm["merkle_fused_compressions"], BLAKE2S_PEAK_GCOMP, "G Blake2s compressions/s"),
    }

    def roof(name):
        ms, nbytes, launches, pmc_names = fams[name]
        launches = max(launches, 1)
        achieved = nbytes / (1e-3 * ms) / 1e9 if ms > 0 else 0.0
        traffic = None
        got = [pmc[k] for k in pmc_names if k in pmc]
        if got:
            tot_l = sum(g["launches"] for g in got)
            traffic = sum(g["hbm_bytes_per_launch_corrected"] * g["launches"] for g in got) / max(tot_l, 1)
        ops, alu_peak, alu_unit = alu[name]
        alu_achieved = ops / (1e-3 * ms) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launches_per_proof": launches,
                "avg_launch_ms": ms / launches, "algorithmic_bytes_per_launch": nbytes / launches,
                "alu_ceiling": {"achieved": alu_achieved, "peak_measured": alu_peak, "unit": alu_unit,
                                "frac": alu_achieved / alu_peak}}

    dom = max(fams, key=lambda k: fams[k][0])
    roofline = roof(dom)
    roofline_other = [roof(k) for k in fams if k != dom]

    # whole-proof figure against SURVEY.md §8d's minimum-traffic model (48*C*N + 1500*N bytes, C = 27 columns)
    model_bytes = (48 * 27 + 1500) * float(1 << args.log_rows)
    whole = {"model_bytes_per_proof": model_bytes, "achieved": model_bytes / (1e-3 * agg["ms_per_step"]) / 1e9 * world,
             "peak": HBM_PEAK_GBS * world, "unit": "GB/s"}
    whole["frac"] = whole["achieved"] / whole["peak"]

    # the reference's own published shape (32x32 Add, 1 024 Add rows + 2 048 Inputs rows,
    # 13.05 ms on a GitHub Actions runner) as a sanity anchor: solo GPU latency, median of 9
    anchor = None
    if rank == 0 and world == 1 and not args.no_anchor:
        from luminair_amd import backend as _bk
        ap = luminair_amd.Prover(dev, protocol_variant=_bk.VARIANT_PINNED)
        atabs = [(k, r, len(r)) for k, r in syn.config2_graph_faithful(1024, 42)]
        ap.ctx.prove_tables(atabs)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            ap.ctx.prove_tables(atabs)
            ts.append(1e3 * (time.perf_counter() - t0))
        anchor = {"workload": "32x32 Add graph: Add 2^10 rows + Inputs 2^11 rows, host rows (PCIe-inclusive)",
                  "gpu_latency_ms": sorted(ts)[4], "reference_published_ms": 13.05,
                  "reference_hardware": "GitHub Actions ubuntu-latest CPU (docs/snippets/benchmark-component.mdx:172)"}

    # the step before the path: `process_trace` of the Add node on device tensors
    trace_gen = None
    if rank == 0:
        import numpy as _np
        n_rows = 1 << args.log_rows
        rng = _np.random.default_rng(7)
        dl = prover.ctx.upload(rng.integers(-2048, 2048, size=n_rows).astype(_np.int32))
        dr = prover.ctx.upload(rng.integers(-2048, 2048, size=n_rows).astype(_np.int32))
        rows_buf = prover.ctx.alloc(n_rows * 15 * 4)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            _, ob = prover.ctx.trace_elementwise(0, dl, dr, n_rows, node_id=2, input_ids=(0, 1), num_consumers=0,
                                                 is_final_output=True, input_mults=(0, 0), rows=rows_buf)
            ts.append(1e3 * (time.perf_counter() - t0))
            ob.free()
        for b_ in (dl, dr, rows_buf):
            b_.free()
        trace_gen = {"workload": "Add node process_trace on device tensors, 2^%d elements -> 15-column rows in HBM"
                                 % args.log_rows, "ms": sorted(ts)[2],
                     "reference_published_ms": 0.0959, "reference_workload": "32x32 Add trace generation (BASELINE.md §1)"}

    line = {
        "metric": "proofs/sec, 2^%d-row Add trace" % args.log_rows, "value": agg["value"], "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": agg["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (M31/QM31 field arithmetic)",
        "data": "synthetic" + (" (host rows: PCIe-inclusive)" if args.host_rows else ""),
        "config": {"workload": "BASELINE config 2a: single Add-op AIR, 2^%d trace rows per proof, PcsConfig default "
                               "(pow 5, blowup 2x, 3 queries), KAT protocol variant" % args.log_rows,
                   "rows": 1 << args.log_rows, "proofs_per_rank": args.steps, "parallelism": "proof-sharded x%d" % world,
                   "proofs_in_flight_per_gpu": inflight,
                   "proof_bytes": len(out["proof"])},
        "prove_latency_ms": latency_ms,
        "prove_latency_p95_ms": latency_p95_ms,
        "stage_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
        "roofline": roofline,
        "roofline_other": roofline_other,
        "whole_proof_vs_traffic_model": whole,
    }
    if anchor:
        line["reference_shape_anchor"] = anchor
    if trace_gen:
        line["device_trace_generation"] = trace_gen
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_log, args.log_rows), args.log_rows)
    if rank == 0:
        print(json.dumps(line))
    pool.shutdown()
    for bl in bufs:
        for _, b, _ in bl:
            if hasattr(b, "free"):
                b.free()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
